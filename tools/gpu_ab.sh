# scratch: the command list of the most recent `gpurun -- 'bash tools/gpu_ab.sh'` call; edited per experiment
cd $GRAFT_REPO_ROOT
timeout 600 python tools/stress_export.py 2>&1 | tail -4
timeout 600 python tools/stress_backward.py 20 2>&1 | tail -6
