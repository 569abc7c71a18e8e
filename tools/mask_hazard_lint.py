"""Lint for the lane-mask hazard of DESIGN.md section 6 ("stale lane-mask bits"): disassembles the device code of a
built library and reports every place where

    v_cmp*   -> writes a lane mask (an SGPR pair or vcc)                               (VALU)
    s_and_b64 / s_or_b64 / s_andn2_b64 / s_orn2_b64 / s_xor_b64 / s_xnor_b64
             -> combines such a FRESH mask (written by a v_cmp at most FRESH instructions earlier) on the scalar ALU
    v_cndmask_b32 ... , <the combined pair>     within WINDOW instructions of the combination, with no VALU write of
                                                that pair in between                   (VALU select on the SALU result)

which is the shape of `in = bx && by; w = in ? a : b` that delivered stale mask bits for lanes 48..63 on MI355X when a
SIMD ran a single wave (tools/stress_export.py found it; corners_setup in csrc/tt_device.h is written with 0/1 float
factors so that the compiler cannot form it).  The lint looks at the texel-weight code paths only when `--functions`
narrows it; by default every kernel is scanned and the findings are listed per kernel.

usage: python tools/mask_hazard_lint.py path/to/libtt_hip.so [--window 4] [--fresh 6] [--json]
exit status 1 if any finding is inside a guarded kernel (see GUARDED below), else 0."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
# kernels whose bilinear weights / scatter coefficients go through corners_setup: a finding here fails the lint
GUARDED = ("k_decode_rays", "k_decode_bwd_geo", "k_decode_bwd_tex", "k_query_points", "k_query_field", "k_render_eval",
           "k_points_bwd_x")
SALU_COMBINE = ("s_and_b64", "s_or_b64", "s_andn2_b64", "s_orn2_b64", "s_xor_b64", "s_xnor_b64", "s_nand_b64", "s_nor_b64")
PAIR = re.compile(r"\b(vcc|s\[\d+:\d+\])")


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="tt_lint_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" in f:
                r = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], cwd=tmp, check=True, capture_output=True, text=True)
                out.append(r.stdout)
        return "\n".join(out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(asm):
    """yield (symbol, [instruction text, ...])"""
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name and body:
                yield name, body
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.split("//")[0].strip()
        if t and not t.endswith(":"):
            body.append(t)
    if name and body:
        yield name, body


def is_valu(op):
    return op.startswith("v_")


def operands(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return []
    return [o.strip() for o in parts[1].split(",")]


def scan(body, window, fresh):
    """findings: (index of the v_cndmask, index of the s_* combine, index of the v_cmp, pair)"""
    found = []
    cmp_at = {}       # pair -> index of the last v_cmp that wrote it
    combined = {}     # pair -> (index of the combine, index of the v_cmp feeding it)
    for i, ins in enumerate(body):
        op = ins.split()[0]
        ops = operands(ins)
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            dst = ops[0] if ops and PAIR.fullmatch(ops[0]) else "vcc"  # e32 forms write vcc implicitly
            cmp_at[dst] = i
            combined.pop(dst, None)
            continue
        if op in SALU_COMBINE and len(ops) >= 3:
            srcs = [o for o in ops[1:] if PAIR.fullmatch(o)]
            # fresh compare masks, directly or through an earlier combination (in = (a && b) && (c && d))
            feeding = [cmp_at[s] for s in srcs if s in cmp_at and i - cmp_at[s] <= fresh]
            feeding += [combined[s][1] for s in srcs if s in combined and i - combined[s][1] <= fresh]
            combined.pop(ops[0], None)
            cmp_at.pop(ops[0], None)
            if feeding:
                combined[ops[0]] = (i, max(feeding))
            continue
        if op.startswith("v_cndmask") and ops:
            sel = ops[-1] if PAIR.fullmatch(ops[-1]) else "vcc"
            if sel in combined and i - combined[sel][0] <= window:
                found.append((i, combined[sel][0], combined[sel][1], sel))
            continue
        # any other write to a tracked pair ends its tracking (VALU writes are what makes it safe; be conservative
        # and drop on every write)
        if ops and PAIR.fullmatch(ops[0]) and not op.startswith(("s_cbranch", "s_branch", "v_cndmask")):
            combined.pop(ops[0], None)
            cmp_at.pop(ops[0], None)
    return found


def lint(lib, window=4, fresh=6):
    res = {}
    for name, body in functions(disassemble(lib)):
        f = scan(body, window, fresh)
        if f:
            res[name] = [{"cndmask": body[a], "combine": body[b], "cmp": body[c], "pair": p, "distance": a - b}
                         for a, b, c, p in f]
    return res


def guarded(name):
    return any(g in name for g in GUARDED)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--window", type=int, default=4)
    ap.add_argument("--fresh", type=int, default=6)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    r = lint(a.lib, a.window, a.fresh)
    if a.json:
        print(json.dumps(r, indent=1))
    else:
        for k, v in r.items():
            print(f"{'GUARDED ' if guarded(k) else '        '}{k[:90]}: {len(v)} finding(s)")
            for f in v[:6]:
                print(f"      {f['cmp']}  ->  {f['combine']}  ->  {f['cndmask']}   (distance {f['distance']})")
    sys.exit(1 if any(guarded(k) for k in r) else 0)
