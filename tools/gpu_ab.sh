cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pair.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-200
(cd _r3 && bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-200 | sed "s/^/r3 /")
done
