cd $GRAFT_REPO_ROOT
python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_config3.json
python bench.py --exact-f32 --steps 50 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_exact.json
python - <<'PY'
import json
for f in ('gpurun_out/bench_config3.json','gpurun_out/bench_exact.json'):
    d=json.loads(open(f).read())
    print(f, d['value'], d['ms_per_step'], d['config'].get('workload','')[:80], {k:v['avg_ms'] for k,v in d['kernels'].items()})
PY
python tools/time_field_query.py 2>&1 | tail -2
python tools/time_eval_paths.py 2>&1 | tail -4
