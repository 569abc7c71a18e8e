cd $GRAFT_REPO_ROOT
TT_FUZZ_SEEDS=400 timeout 1700 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --tb=line -k "point_query or eval_render" 2>&1 | grep -v "^$" | cut -c1-700 | tail -25
