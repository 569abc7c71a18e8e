cd $GRAFT_REPO_ROOT
bash tools/abn.sh 3 "--steps 100" "--steps 100 --lib-variant base" 2>&1 | cut -c1-170
