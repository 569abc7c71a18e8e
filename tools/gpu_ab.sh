# scratch: the command list of the most recent `gpurun -- 'bash tools/gpu_ab.sh'` call; edited per experiment
cd $GRAFT_REPO_ROOT
for r in 1 2; do python tools/time_proposal.py 2>&1 | tail -2; TT_LIB_VARIANT=s768 python tools/time_proposal.py 2>&1 | tail -2 | sed "s/^/s768 /"; done
