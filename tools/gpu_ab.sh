# scratch: the command list of the most recent `gpurun -- 'bash tools/gpu_ab.sh'` call; edited per experiment
cd $GRAFT_REPO_ROOT
bash tools/abn.sh 2 "--steps 100" "--steps 100 --lib-variant t256" "--steps 100 --lib-variant t384" 2>&1 | cut -c1-170
