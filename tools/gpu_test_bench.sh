# usage (GPU box): bash tools/gpu_test_bench.sh TAG [pytest-args...]   -- the GPU suite + a short bench, logs under gpurun_out/TAG
cd $GRAFT_REPO_ROOT
TAG=${1:-run}; shift
mkdir -p gpurun_out/$TAG
rm -f gpurun_out/parity_report.jsonl
(time timeout 1500 python -m pytest tests -m gpu -x -q "$@") > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -4 gpurun_out/$TAG/pytest.log
grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/$TAG/pytest.log | head -20
cp gpurun_out/parity_report.jsonl gpurun_out/$TAG/ 2>/dev/null
bash tools/abn.sh 2 "--steps 100" "--steps 100 --exact-f32" 2>&1 | tee gpurun_out/$TAG/bench_short.txt
