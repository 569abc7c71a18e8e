cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
rm -f gpurun_out/parity_report.jsonl
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest.log
tail -5 gpurun_out/r4a/pytest.log
./tools/mfma16_probe > gpurun_out/r4a/mfma16_probe.txt 2>&1
./tools/ds_tr_probe > gpurun_out/r4a/ds_tr_probe.txt 2>&1
timeout 60 ./tools/pair_sync_probe > gpurun_out/r4a/pair_sync_probe.txt 2>&1
cat gpurun_out/r4a/mfma16_probe.txt gpurun_out/r4a/pair_sync_probe.txt
(time timeout 900 python bench.py) > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r4a/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r4a/bench.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','timed_region_s')}, d.get('sustained'))
    for k,v in d['kernels'].items(): print(k, v['avg_ms'], v.get('live_tile_frac'), v.get('frac_8d'), v.get('frac_8d_executed'), v.get('sq'))
    print('exact', d.get('exact_f32',{}).get('ms_per_step'))
    print('secondary', json.dumps(d.get('secondary'))[:3000])
except Exception as e: print('parse fail', e)
PY
