cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-170
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
