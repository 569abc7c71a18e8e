#!/bin/bash
# Dev tool: A/B/C... of several bench.py flag sets inside ONE gpurun call.  usage: bash tools/abn.sh ROUNDS "flags1" "flags2" ...
N=$1; shift
for r in $(seq 1 $N); do
  for v in "$@"; do
    python bench.py --no-cpu-baseline --no-pmc --no-extras $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % sys.argv[1], 'ms/step %.3f' % d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'glue %.3f' % d['stages']['glue_ms'], 'loss %.6f' % d['config']['loss'])" "[$v]"
  done
done
