# scratch: the command list of the most recent `gpurun -- 'bash tools/gpu_ab.sh'` call (A/B timings of library variants,
# tools/abn.sh); edited per experiment
cd $GRAFT_REPO_ROOT
bash tools/abn.sh 2 "--steps 100" 2>&1 | cut -c1-170
