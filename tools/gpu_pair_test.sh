cd $GRAFT_REPO_ROOT
bash tools/abn.sh 1 "--steps 50" "--steps 50 --lib-variant noat" "--steps 50 --lib-variant nosc" "--steps 50 --bwd-solo" 2>&1 | cut -c1-200
