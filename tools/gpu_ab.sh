cd $GRAFT_REPO_ROOT
bash tools/abn.sh 2 "--steps 100" "--steps 100 --lib-variant fwA" "--steps 100 --lib-variant fwB" "--steps 100 --lib-variant fwC" 2>&1 | cut -c1-160
