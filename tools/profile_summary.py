"""Fold the output of tools/profile_round.sh into the tracked round summary.
usage: python tools/profile_summary.py gpurun_out/prof_r01 r01
writes profiles/<tag>_summary.md, <tag>_kernel_stats.csv, <tag>_traffic_bytes.json, <tag>_bench_n1.json"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
CLOCK_GHZ, SIMDS, SES = 2.1, 1024, 32  # clock assumed under load; 256 CUs x 4 SIMDs; 32 shader engines


def short(name):
    return name.split("(")[0].replace("void ", "")[:44]


ours = lambda n: short(n).startswith(("k_", "k_decode", "k_march", "k_planes", "k_sample"))
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
L = [f"# Round {tag[1:]} profile summary (1x MI355X, bench.py workload: 256x256 rays x 128 samples, fwd+bwd)", "",
     "command: rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 "
     "--no-cpu-baseline --no-pmc --no-extras   (recipe: tools/profile_round.sh, folded by tools/profile_summary.py)", "",
     "## rocprofv3 --stats (top kernels)", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:14]:
    L.append(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
             f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if not ours(r["Kernel_Name"]):
            continue
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows if ours(r["Name"])}
m = lambda k, c: acc[k][c] / cnt[k][c] if cnt[k][c] else float("nan")

L += ["", "## SQ counters (separate --pmc pass), per launch", "",
      f"MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (duration x {CLOCK_GHZ} GHz x {SIMDS} SIMDs)  [clock assumed under "
      "load]; wave life = SQ_WAVE_CYCLES x 4 / SQ_WAVES / (SQ_BUSY_CYCLES / 32 SEs): the average fraction of the "
      "kernel a wave is resident (load balance; counters in quad-cycles / per-SE cycles).  The decode kernels (template argument 2 = the default three-piece "
      "precision mode, 0 = two pieces, 1 = fp32 MFMA) run their mat-vec chains as split-fp16 MFMAs: 6 (three-piece) or 3 "
      "(two-piece) x v_mfma_f32_32x32x16_f16 = 192 / 96 busy cycles per 16-deep k-step instead of 8 x "
      "v_mfma_f32_32x32x2_f32 = 512, so their MfmaUtil is LOW BY DESIGN: the same algorithmic FLOPs need 2.7x / 5.3x less "
      "matrix-pipe time (the bench line's `modes.f32.kernels.*.sq.mfma_util` has the fp32-MFMA mode: 0.62 - 0.67).", "",
      "| kernel | avg us | MFMA busy cycles | MFMA busy ms / SIMD | MfmaUtil | wave life | WAIT_ANY/WAVE | WAIT_INST/WAVE |",
      "|---|---|---|---|---|---|---|---|"]
for k in sorted(acc, key=lambda k: -avg_us.get(k, 0)):
    us = avg_us.get(k, float("nan"))
    busy = m(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    util = busy / (us * 1e-6 * CLOCK_GHZ * 1e9 * SIMDS)
    wc, wv, bc = m(k, "SQ_WAVE_CYCLES"), m(k, "SQ_WAVES"), m(k, "SQ_BUSY_CYCLES")
    life = (wc * 4 / wv) / (bc / SES) if wv and bc else float("nan")
    L.append(f"| {k} | {us:.1f} | {busy:.3e} | {busy / SIMDS / (CLOCK_GHZ * 1e6):.2f} | {util:.2f} | {life:.2f} | "
             f"{m(k, 'SQ_WAIT_ANY') / wc:.2f} | {m(k, 'SQ_WAIT_INST_ANY') / wc:.2f} |")

L += ["", "## HBM-side traffic (separate --pmc passes; FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE x2 = the gfx950 "
      "correction of MI355X_MICROARCH.md, calibrated there for wide coalesced reads only)", "",
      "| kernel | FETCH_SIZE KiB | corrected read MB | WRITE_SIZE KiB | write MB | L2 hit rate |", "|---|---|---|---|---|---|"]
traffic = {}
for k in sorted(acc, key=lambda k: -avg_us.get(k, 0)):
    f, w = m(k, "FETCH_SIZE"), m(k, "WRITE_SIZE")
    hit, miss = m(k, "TCC_HIT_sum"), m(k, "TCC_MISS_sum")
    traffic[k] = (2 * f + w) * 1024
    L.append(f"| {k} | {f:.0f} | {2 * f * 1024 / 1e6:.1f} | {w:.0f} | {w * 1024 / 1e6:.1f} | {hit / (hit + miss):.3f} |")
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic_bytes.json"), "w"), indent=1)

L += ["", "## Instruction mix per launch (SQ_INSTS_*)", "",
      "| kernel | VALU (incl. MFMA) | LDS | VMEM rd | VMEM wr | LDS bank-conflict cycles / LDS active |", "|---|---|---|---|---|---|"]
for k in sorted(acc, key=lambda k: -avg_us.get(k, 0)):
    L.append(f"| {k} | {m(k, 'SQ_INSTS_VALU'):.3e} | {m(k, 'SQ_INSTS_LDS'):.3e} | {m(k, 'SQ_INSTS_VMEM_RD'):.3e} | "
             f"{m(k, 'SQ_INSTS_VMEM_WR'):.3e} | {m(k, 'SQ_LDS_BANK_CONFLICT') / max(m(k, 'SQ_LDS_IDX_ACTIVE'), 1):.3f} |")

bj = os.path.join(src, "bench_n1.json")
if os.path.exists(bj) and os.path.getsize(bj) > 0:
    line = open(bj).read().strip().splitlines()[-1]
    open(os.path.join(dst, f"{tag}_bench_n1.json"), "w").write(line + "\n")
    d = json.loads(line)
    L += ["", "## bench.py line of this build", "",
          f"{d['value'] / 1e6:.3f} M rays/s, {d['ms_per_step']:.2f} ms/step; roofline {d['roofline']['kernel']} "
          f"{d['roofline']['achieved']} / {d['roofline']['peak']} {d['roofline']['unit']} = {d['roofline']['frac']}; ray march "
          f"{d['roofline_hbm']['achieved']} GB/s = {d['roofline_hbm']['frac']} of HBM peak; cpu_baseline "
          f"{d.get('cpu_baseline', {}).get('value', float('nan')):.0f} rays/s on "
          f"{d.get('cpu_baseline', {}).get('cores', '?')} threads."]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
