"""Dev tool: per-kernel registers / spills / LDS / occupancy as the compiler reports them (no GPU needed).
usage: python tools/kernel_resources.py [file.hip ...]   (default: every csrc/*.hip)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "triplaneturbo_amd", "csrc", "*.hip")))
sys.path.insert(0, ROOT)
from triplaneturbo_amd._lib import SOURCE_FLAGS  # noqa: E402  (the per-file flags the product library is built with)
# EXTRA_FLAGS="..." replaces the per-file flags (A/B of scheduler options); unset = exactly the product build's flags
extra_env = os.environ.get("EXTRA_FLAGS")
for f in files:
    extra = extra_env.split() if extra_env is not None else list(SOURCE_FLAGS.get(os.path.basename(f), []))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-gpu-rdc",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "triplaneturbo_amd", "csrc"),
           "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for k, r in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout
        name = name.strip().split("(")[0].replace("void ", "")
        g = lambda key: r.get(key, "?")
        print(f"{name[:46]:46s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>4s} spill {g('VGPRs Spill'):>4s} "
              f"scratch {g('ScratchSize [bytes/lane]'):>5s} sgpr {g('TotalSGPRs'):>4s} lds {g('LDS Size [bytes/block]'):>7s} "
              f"occ {g('Occupancy [waves/SIMD]')}")
