cd $GRAFT_REPO_ROOT
bash tools/abn.sh 2 "--steps 100" "--steps 100 --lib-variant r1" "--steps 100 --lib-variant r3" "--steps 100 --lib-variant r4" "--steps 100 --lib-variant r5" 2>&1 | cut -c1-170
