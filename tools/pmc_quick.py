"""Dev tool: the SQ pipe-utilisation pass of bench.py (pmc_pipes) for every decode kernel of the bench step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res, err = bench.pmc_pipes(1)
if res is None:
    print("pmc failed:", err)
else:
    for k, v in res.items():
        if "decode" in k or "march" in k:
            print(k, json.dumps(v))
