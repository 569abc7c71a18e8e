R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras"
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcx$i -- $CMD > $R/gpurun_out/pmcx$i.log 2>&1
  echo "pass $i rc=$?"
done
