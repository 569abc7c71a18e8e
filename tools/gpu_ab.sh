# scratch: the command list of the most recent `gpurun -- 'bash tools/gpu_ab.sh'` call; edited per experiment
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
cp gpurun_out/parity_report.jsonl gpurun_out/parity_report_full.jsonl
