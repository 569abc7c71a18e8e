"""Dev tool: instruction-fetch counters of the decode kernels (bench step, one pass per counter group): is the ~54 KB loop
body of the backward kernels served by the 64 KB instruction cache two CUs share?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
groups = [["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"],
          ["SQ_IFETCH", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INST_CYCLES_VMEM", "SQ_WAVES"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"],
          ["SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES"]]
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for g in groups:
    res, err = bench.pmc_pass(g, cfg)
    if res is None:
        print("pmc failed:", g, err)
        continue
    kernels = sorted({k for c in res.values() for k in c if "decode" in k or "render" in k})
    for k in kernels:
        print(k, json.dumps({c: res[c].get(k) for c in g if c in res}))
