#!/bin/bash
# Dev tool: rocprofv3 kernel stats (our kernels only) of a python command.  usage: bash tools/kstats.sh OUTDIR cmd...
out=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -- "$@" > $R/$out/log.txt 2>&1
f=$(find $R/$out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E "k_march|k_decode|k_planes|k_query" "$f" | awk -F'","' '{printf "%-60s calls %5s avg_us %10.1f\n", substr($1,2,58), $2, $4/1000}'
