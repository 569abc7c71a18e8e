cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_points_backward.py -m gpu -x -q 2>&1 | tail -4)
bash tools/abn.sh 2 "--steps 100" 2>&1 | cut -c1-210
