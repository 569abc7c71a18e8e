#!/bin/bash
# Dev tool: rocprofv3 kernel stats (our kernels only) of a python command.  usage: bash tools/kstats.sh OUTDIR cmd...
out=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -- "$@" > $R/$out/log.txt 2>&1
f=$(find $R/$out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_march", "k_decode", "k_planes", "k_query", "k_sample")):
        print(f"{r['Name'].split('(')[0][:50]:50s} calls {r['Calls']:>5s}  avg_us {float(r['AverageNs']) / 1e3:10.1f}")
PY
