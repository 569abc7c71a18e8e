#!/bin/bash
# Round profile of bench.py on the GPU box: one --stats pass + separate --pmc passes (never combined with
# sys/hip/hsa traces).   usage: bash tools/profile_round.sh gpurun_out/prof_r01
out=${1:-gpurun_out/prof}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- $CMD > $R/$out/stats.log 2>&1
echo "stats rc=$?"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$out/pmc$i -- $CMD > $R/$out/pmc$i.log 2>&1
  echo "pmc pass $i ($grp) rc=$?"
done
timeout 600 python $R/bench.py > $R/$out/bench_n1.json 2> $R/$out/bench_n1.log
echo "bench rc=$?"
