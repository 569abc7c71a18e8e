# scratch: the command list of the most recent `gpurun -- 'bash tools/gpu_ab.sh'` call; edited per experiment
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -30
(python bench.py --steps 600 --no-cpu-baseline --no-pmc --no-extras > /dev/null 2>&1 &) ; sleep 25; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|power\|temp\|mclk" | head -12
