cd $GRAFT_REPO_ROOT
bash tools/abn.sh 1 "--steps 50" "--steps 50 --bwd-pair" 2>&1 | cut -c1-160
