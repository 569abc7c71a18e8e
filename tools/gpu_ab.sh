cd $GRAFT_REPO_ROOT
for r in 1 2; do
bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-200
(cd _r3 && bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-200 | sed "s/^/r3 /")
done
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_pair.py tests/test_gpu_points.py -m gpu -x -q 2>&1 | tail -2
