# round 5, first GPU call: the three-piece products (probe, parity subset, fuzz seeds, A/B/C of the three modes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 120 tools/split3_probe > $O/split3_probe.txt 2>&1; echo "probe rc=$?"; cat $O/split3_probe.txt
rm -f gpurun_out/parity_report.jsonl
(time timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_forward.py tests/test_gpu_points_backward.py tests/test_gpu_eval.py tests/test_gpu_work_accounting.py tests/test_gpu_plugin.py -x -q) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_subset.log
cp gpurun_out/parity_report.jsonl $O/ 2>/dev/null
timeout 600 python tools/fuzz_seeds.py 176 235 288 295 99 391 198 229 > $O/fuzz_named.txt 2>&1; tail -40 $O/fuzz_named.txt
timeout 900 python tools/fuzz_seeds.py range:0:120 > $O/fuzz_0_120.txt 2>&1; tail -25 $O/fuzz_0_120.txt
bash tools/abn.sh 2 "--precision split3 --steps 100" "--precision f32 --steps 100" "--precision split2 --steps 100" 2>&1 | tee $O/abc.txt
