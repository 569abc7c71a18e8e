cd $GRAFT_REPO_ROOT
bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-200
TT_FUZZ_SEEDS=400 timeout 1700 python -m pytest "tests/test_gpu_fuzz.py::test_random_configuration_matches_oracle" -m gpu -q --tb=line 2>&1 | grep -v "^$" | cut -c1-600 | tail -25 > gpurun_out/fuzz400.txt
cat gpurun_out/fuzz400.txt
