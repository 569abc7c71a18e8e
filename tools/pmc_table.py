"""Dev tool: fold the counter_collection csv files written by tools/pmc_passes.sh into one per-kernel table.
usage: python tools/pmc_table.py gpurun_out/pmc"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        if not k.startswith(("k_", "void k_")):
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f"   {c:34s} {acc[k][c] / cnt[k][c]:14.4g}  (per launch, {cnt[k][c]} launches)")
