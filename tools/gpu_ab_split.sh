cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
bash tools/abn.sh 2 "--steps 100 --lib-variant rtz" "--steps 100" "--steps 100 --lib-variant asm" "--steps 100 --lib-variant loasm" 2>&1 | tee gpurun_out/r4a/ab_split2.txt
