"""Dev tool: the tracked copy of a full `pytest -m gpu` parity report.  The 1 400 fuzz records keep, per gradient, the norm
ratios, the measured fp32 order sensitivity, the rule that let the gradient pass and the element-wise violating fraction against the fp32 oracle
(the worst-element details and the per-alternative lists stay in gpurun_out/); every other record is copied unchanged.
usage: python tools/compact_parity_report.py gpurun_out/r06final/parity_report.jsonl profiles/r06_parity_report.jsonl"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
KEEP = ("hip_vs_fp32", "hip_vs_fp64", "fp32_vs_fp64", "fp32_order_sensitivity", "passed_by", "bar", "nearest", "nearest_is")
n = 0
with open(dst, "w") as out:
    for line in open(src):
        r = json.loads(line)
        if str(r.get("case", "")).startswith("test_gpu_fuzz"):
            c = {}
            for k, v in r.items():
                if isinstance(v, dict) and "hip_vs_fp32" in v:
                    g = {x: (float(f"{v[x]:.3g}") if isinstance(v[x], float) else v[x]) for x in KEEP if x in v}
                    for e in ("elem_hip_vs_fp32",):
                        if e in v:
                            g[e + "_viol_frac"] = float(f"{v[e]['viol_frac']:.4g}")
                    c[k] = g
                else:
                    c[k] = v
            r = c
        out.write(json.dumps(r) + "\n")
        n += 1
print(n, "records")
