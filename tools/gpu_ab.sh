cd $GRAFT_REPO_ROOT
python tools/profile_kernels.py 10 0 0x2000 0x100 0 0x2000 2>&1 | tail -5
