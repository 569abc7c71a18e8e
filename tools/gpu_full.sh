cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
cp tests/../gpurun_out/parity_report.jsonl gpurun_out/parity_report_full.jsonl 2>/dev/null
