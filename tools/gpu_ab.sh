# scratch: the command list of the most recent `gpurun -- 'bash tools/gpu_ab.sh'` call; edited per experiment
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py tests/test_gpu_determinism.py tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
bash tools/abn.sh 1 "--steps 100" 2>&1 | cut -c1-170
