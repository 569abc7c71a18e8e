cd $GRAFT_REPO_ROOT
bash tools/kstats.sh gpurun_out/ks python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras
