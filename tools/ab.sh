#!/bin/bash
# Dev tool: A/B of two bench.py variants inside ONE gpurun call (box-to-box spread is +-1 %).
# usage: bash tools/ab.sh "<flags A>" "<flags B>" [rounds]
A="$1"; B="$2"; N=${3:-2}
for r in $(seq 1 $N); do
  for v in "$A" "$B"; do
    python bench.py --no-cpu-baseline --no-pmc --no-extras $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s' % sys.argv[1], 'ms/step %.3f' % d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'glue %.3f' % d['stages']['glue_ms'])" "[$v]"
  done
done
