"""Build + load libtt_hip.so (the C-ABI HIP library, include/tt_abi.h) with ctypes.

There is NO fallback: if the shared library is missing or a symbol is absent the import of
any op fails loudly.  ``build()`` cross-compiles for gfx950 with hipcc (works without a GPU).
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import re
import subprocess
import tempfile
from typing import List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIB_PATH = os.path.join(_HERE, "libtt_hip.so")
# dev-only variant built with -DTT_TUNING: honours the TT_DEBUG_FLAGS / TT_SB / TT_CHUNK / TT_UNIT / TT_ORDER
# environment variables (profiling ablations and tuning sweeps, tools/).  The product library above never calls getenv.
_DEFAULT_LIB_PATH = LIB_PATH
TUNING_LIB_PATH = os.path.join(_HERE, "libtt_hip_tuning.so")
SOURCES = ["tt_forward.hip", "tt_march.hip", "tt_backward.hip", "tt_backward_tex.hip", "tt_points.hip", "tt_composite.hip", "tt_grad2.hip", "tt_sampler.hip", "tt_hashgrid.hip", "tt_host.cpp"]
# per-translation-unit flags: the texture backward is faster under hipcc's max-ILP scheduling strategy (3.11 -> 3.02 ms;
# the other kernels are not); the geometry backward is faster with its transient MFMA results in VGPRs rather than AGPRs
# (-amdgpu-mfma-vgpr-form: 471 -> 248 v_accvgpr_read, 3.045 -> 2.995 ms; texture backward slower, forward neutral) and
# without the scheduler's register-pressure rescheduling stage, which has nothing to win at one wave per SIMD
# (-amdgpu-disable-unclustered-high-rp-reschedule: 2.945 -> 2.865 ms; forward slower, texture neutral):
# profiles/experiments/README.md
# Round 6 (dense-rank scatter): the texture unit is now faster WITHOUT max-ILP and, like the geometry unit, without the
# register-pressure rescheduling stage (same-box A/B, texture backward 3.36 max-ilp / 3.32 no flags / 3.30 this).
SOURCE_FLAGS = {"tt_backward_tex.hip": ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"],
                "tt_backward.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form",
                                    "-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"]}
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared", "-fno-gpu-rdc"]

# every symbol include/tt_abi.h declares (tests check the header and this list agree)
SYMBOLS = [
    "tt_strerror", "tt_abi_version", "tt_planes_pack", "tt_planes_unpack_grad", "tt_query_points",
    "tt_query_field", "tt_decode_rays", "tt_render_fwd", "tt_render_bwd_geo", "tt_render_bwd_tex", "tt_grid_sample_2d_grad2", "tt_grid_sample_2d_grad2_typed",
    "tt_march_fwd", "tt_march_bwd", "tt_sample_uniform", "tt_sample_importance",
    "tt_points_bwd_geo", "tt_points_bwd_tex", "tt_points_bwd_x", "tt_hashgrid_n_params", "tt_hashgrid_fwd", "tt_hashgrid_bwd",
    "tt_debug_poison_queue", "tt_patch_composite_fwd", "tt_patch_composite_bwd", "tt_render_eval",
    "tt_composite_fwd", "tt_composite_bwd", "tt_eikonal_fwd", "tt_eikonal_bwd", "tt_source_hash",
]


def _sources() -> List[str]:
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _expected_abi() -> int:
    m = re.search(r"#define\s+TT_ABI_VERSION\s+(\d+)", open(os.path.join(INCLUDE, "tt_abi.h")).read())
    if not m:
        raise RuntimeError("include/tt_abi.h has no TT_ABI_VERSION")
    return int(m.group(1))


def _deps() -> List[str]:
    deps = _sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    deps.append(os.path.join(INCLUDE, "tt_abi.h"))
    return deps


def source_hash(tuning: bool = False, defines: Optional[List[str]] = None, source_flags: Optional[dict] = None) -> str:
    """sha256 over every source / header of the library AND the flags it is built with.  build() embeds it in the
    binary (`tt_source_hash()`, marker string TT_SOURCE_HASH=...); needs_build() compares the embedded value with the
    tree, so a .so that travelled with equal or newer mtimes than its sources (a pushed tree, a checkout) is never
    silently reused for different sources."""
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
        h.update(b"\0")
    flags = dict(SOURCE_FLAGS if source_flags is None else source_flags)
    h.update(repr((HIPCC_FLAGS, sorted(flags.items()), bool(tuning), list(defines or []))).encode())
    return h.hexdigest()


_HASH_RE = re.compile(rb"TT_SOURCE_HASH=([0-9a-f]{64})")


def embedded_hash(path: str) -> Optional[str]:
    """The source hash a built library carries (read from the file, no dlopen), or None."""
    try:
        m = _HASH_RE.search(open(path, "rb").read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def needs_build(path: str = LIB_PATH, tuning: bool = False, defines: Optional[List[str]] = None,
                source_flags: Optional[dict] = None) -> bool:
    if not os.path.exists(path):
        return True
    return embedded_hash(path) != source_hash(tuning, defines, source_flags)


_FLAG_OK: dict = {}


def _flag_supported(hipcc: str, pair: List[str]) -> bool:
    """Performance-only `-mllvm` options are hidden LLVM flags: probe each once on an empty translation unit, so that a
    hipcc without one of them builds the library without it (with a warning) instead of failing the whole build."""
    key = " ".join(pair)
    if key not in _FLAG_OK:
        with tempfile.TemporaryDirectory(prefix="tt_flag_") as d:
            src = os.path.join(d, "e.hip")
            open(src, "w").write("#include <hip/hip_runtime.h>\n__global__ void k() {}\n")
            base = [hipcc, "--offload-arch=gfx950", "-c", src, "-o", os.path.join(d, "e.o")]
            if subprocess.run(base, capture_output=True, text=True).returncode != 0:
                return True  # the probe itself does not compile here: keep the flag and let the real build speak
            r = subprocess.run(base + pair, capture_output=True, text=True)
        _FLAG_OK[key] = r.returncode == 0
        if r.returncode != 0:
            print(f"triplaneturbo_amd: hipcc rejects '{key}' (performance-only flag): building without it")
    return _FLAG_OK[key]


def _usable_flags(hipcc: str, flags: List[str]) -> List[str]:
    out, i = [], 0
    while i < len(flags):
        pair = flags[i:i + 2] if flags[i] == "-mllvm" else flags[i:i + 1]
        if flags[i] != "-mllvm" or _flag_supported(hipcc, pair):
            out += pair
        i += len(pair)
    return out


def build(force: bool = False, verbose: bool = False, tuning: bool = False, variant: Optional[str] = None,
          defines: Optional[List[str]] = None, source_flags: Optional[dict] = None) -> str:
    """hipcc --offload-arch=gfx950 ... -shared -> triplaneturbo_amd/libtt_hip.so (in-tree).  One hipcc process per
    translation unit, in parallel, then a link.  tuning=True builds the dev variant libtt_hip_tuning.so instead;
    variant="x" + defines=["-DFOO"] builds an experiment library libtt_hip_x.so (dev A/B runs, tools/ab.sh).
    Up to date = the source hash embedded in the library equals the hash of the tree (source_hash), not mtimes."""
    out = TUNING_LIB_PATH if tuning else LIB_PATH
    if variant:
        out = os.path.join(_HERE, f"libtt_hip_{variant}.so")
    if not force and not needs_build(out, tuning, defines, source_flags):
        return out
    # One builder at a time per output file: a multi-rank launch on a stale checkout (every rank's load() lands here) must not
    # compile and link concurrently into the same paths.  The ranks serialise on an fcntl lock next to the library; whoever
    # gets it SECOND finds the library up to date and returns.  Objects go to a per-process directory, the link to a
    # per-process temporary that os.replace() moves into place atomically.
    import fcntl
    import shutil
    os.makedirs(os.path.join(_HERE, "build"), exist_ok=True)
    with open(os.path.join(_HERE, "build", os.path.basename(out) + ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build(out, tuning, defines, source_flags):
                return out
            objdir = os.path.join(_HERE, "build", (variant or ("tuning" if tuning else "release")) + f".{os.getpid()}")
            try:
                return _build_locked(out, objdir, verbose, tuning, defines, source_flags)
            finally:
                shutil.rmtree(objdir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(out: str, objdir: str, verbose: bool, tuning: bool, defines: Optional[List[str]],
                  source_flags: Optional[dict]) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(objdir, exist_ok=True)
    digest = source_hash(tuning, defines, source_flags)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + (["-DTT_TUNING"] if tuning else []) + list(defines or [])
    flags.append(f'-DTT_SOURCE_HASH_STR="{digest}"')
    procs = []
    for s in _sources():
        obj = os.path.join(objdir, os.path.basename(s) + ".o")
        per_unit = _usable_flags(hipcc, (SOURCE_FLAGS if source_flags is None else source_flags).get(os.path.basename(s), []))
        cmd = [hipcc] + flags + per_unit + ["-I", INCLUDE, "-I", CSRC, "-x", "hip", "-c", s, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    objs = []
    for obj, pr in procs:
        so, se = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed ({pr.returncode}):\n{so}\n{se}")
        objs.append(obj)
    tmp = f"{out}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    if embedded_hash(tmp) != digest:
        os.unlink(tmp)
        raise RuntimeError(f"{out} does not carry the source hash it was built with")
    os.replace(tmp, out)
    return out


def use_tuning_build() -> None:
    """Dev tools only: build and bind the -DTT_TUNING variant for the rest of this process (before the first op)."""
    global _lib, LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_tuning_build() must be called before the library is first loaded")
    LIB_PATH = build(tuning=True)


def use_variant(name: str) -> None:
    """Dev tools only: bind an experiment library built by build(variant=name, defines=[...]) (must exist already)."""
    global _lib, LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_variant() must be called before the library is first loaded")
    path = os.path.join(_HERE, f"libtt_hip_{name}.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is not built")
    LIB_PATH = path


_lib: Optional[ctypes.CDLL] = None

_P = ctypes.c_void_p
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64
_F = ctypes.c_float


class MlpWeights(ctypes.Structure):
    _fields_ = [(n, _P) for n in ("w1", "w2", "w3", "v1", "v2", "v3")]


class RenderCfg(ctypes.Structure):
    _fields_ = [
        ("n_prompts", _I32), ("views_per_prompt", _I32), ("plane_h", _I32), ("plane_w", _I32),
        ("rays_per_view", _I32), ("n_samples", _I32), ("n_rays", _I64), ("radius", _F),
        ("sdf_bias_radius", _F), ("inv_std", _F), ("cos_anneal_ratio", _F), ("rgb_grad_shrink", _F),
        ("flags", _I32), ("image_w", _I32), ("tile_sb", _I32), ("grad_copies", _I32), ("tile_chunk", _I32),
        ("skip_eps_tex", _F), ("skip_eps_geo", _F), ("inv_std_dev", _P), ("stats", _P),
    ]


class HashGridCfg(ctypes.Structure):  # tt_hashgrid_cfg
    _fields_ = [("n_levels", _I32), ("n_features_per_level", _I32), ("log2_hashmap_size", _I32),
                ("base_resolution", _I32), ("per_level_scale", _F)]


TT_R_PER_SAMPLE = 1
TT_R_EXACT_F32 = 2
TT_R_WGRAD_F32 = 4
TT_R_BWD_SOLO = 8
TT_R_BWD_PAIR = 16
TT_R_SPLIT2 = 32
TT_R_SPLIT3 = 64
TT_R_VOLSDF = 128
TT_Q_NORMAL = 1
TT_Q_TEX = 2
TT_Q_EXACT_F32 = 4
TT_Q_SPLIT2 = 8
TT_Q_SPLIT3 = 16

# Precision of the per-point MLP products (include/tt_abi.h, "precision of the matrix products"):
#   "split3" (default)  fp32-grade: three fp16 pieces per operand, six product terms on the fp16 matrix pipe
#   "f32"               every product on the fp32-input MFMA (the A/B reference)
#   "split2"            the FAST mode: two pieces, three terms, ~2^-21.5 per product (tolerance-bounded)
DEFAULT_PRECISION = "split3"
_PRECISION_FLAGS = {"split3": (TT_R_SPLIT3, TT_Q_SPLIT3), "split2": (TT_R_SPLIT2, TT_Q_SPLIT2),
                    "f32": (TT_R_EXACT_F32, TT_Q_EXACT_F32)}
_PRECISION_ALIASES = {"fast": "split2", "exact_f32": "f32", "fp32_mfma": "f32"}


def resolve_precision(precision: Optional[str] = None, exact_f32: bool = False) -> str:
    """Canonical precision name.  `exact_f32=True` (the switch of rounds 2-4) means "f32"; naming both is an error
    unless they agree."""
    if precision is None:
        return "f32" if exact_f32 else DEFAULT_PRECISION
    name = _PRECISION_ALIASES.get(precision, precision)
    if name not in _PRECISION_FLAGS:
        raise ValueError(f"unknown precision {precision!r}: one of {sorted(_PRECISION_FLAGS)} / {sorted(_PRECISION_ALIASES)}")
    if exact_f32 and name != "f32":
        raise ValueError(f"exact_f32=True contradicts precision={precision!r}")
    return name


def r_flag(name: str) -> int:
    return _PRECISION_FLAGS[name][0]


def q_flag(name: str) -> int:
    return _PRECISION_FLAGS[name][1]
PLACEMENTS = {"tt": 0, "center": 1}  # enum tt_sample_placement
TT_PLACE_VOLSDF = 0x100  # OR-ed into the placement of tt_sample_importance: VolSDF proposal density


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950). triplaneturbo_amd has no CPU/PyTorch fallback.")
    if LIB_PATH == _DEFAULT_LIB_PATH and embedded_hash(LIB_PATH) != source_hash():
        # a library built from OTHER sources (an edited checkout, a stale copy): never bind it silently -- rebuild in-tree,
        # or say so when there is no compiler (dev variants bound by use_variant / use_tuning_build carry their own hash)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not os.path.exists(hipcc):
            raise RuntimeError(f"{LIB_PATH} was built from different sources than this checkout (embedded hash "
                               f"{embedded_hash(LIB_PATH)}, tree {source_hash()}) and {hipcc} is not there to rebuild it")
        build()
    lib = ctypes.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} lacks symbols {missing}; rebuild it")
    lib.tt_strerror.restype = ctypes.c_char_p
    lib.tt_strerror.argtypes = [ctypes.c_int]
    lib.tt_abi_version.restype = ctypes.c_int
    lib.tt_source_hash.restype = ctypes.c_char_p
    lib.tt_source_hash.argtypes = []
    if lib.tt_abi_version() != _expected_abi():
        # argument lists moved between ABI versions: calling a stale library would shift pointer arguments
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.tt_abi_version()}, include/tt_abi.h declares "
                           f"{_expected_abi()}: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    lib.tt_planes_pack.argtypes = [_P, _P, _I32, _I32, _I32, _P]
    lib.tt_planes_unpack_grad.argtypes = [_P, _P, _I32, _I32, _I32, _I32, _P]
    lib.tt_query_points.argtypes = [_P, ctypes.POINTER(MlpWeights), _P, _I32, _I64, _I32, _I32, _I32, _I32, _F, _F,
                                    _I32, _P, _P, _P, _P]
    lib.tt_render_fwd.argtypes = [_P, ctypes.POINTER(MlpWeights), _P, _P, _P, _P, ctypes.POINTER(RenderCfg)] + [_P] * 11
    _cfgp, _wp = ctypes.POINTER(RenderCfg), ctypes.POINTER(MlpWeights)
    lib.tt_query_field.argtypes = [_P, _wp, _P, _I32, _I64, _I32, _I32, _I32, _I32, _F, _F, _I32, _P, _P, _P]
    lib.tt_decode_rays.argtypes = [_P, _wp, _P, _P, _P, _P, _cfgp, _I32, _P, _P, _P, _P]
    optional = {
        "tt_render_bwd_geo": [_P, _wp, _P, _P, _P, _P, _cfgp] + [_P] * 14 + [_P, _P, _P, _wp, _P],
        "tt_render_bwd_tex": [_P, _wp, _P, _P, _P, _P, _cfgp] + [_P] * 4 + [_P, _wp, _P],
        "tt_march_fwd": [_P, _P, _P, _cfgp] + [_P] * 11,
        "tt_march_bwd": [_P, _P, _P, _cfgp] + [_P] * 17,
        "tt_points_bwd_geo": [_P, _wp, _P, _I32, _I64, _I32, _I32, _I32, _I32, _F, _F, _I32, _P, _P, _P, _P, _wp, _P],
        "tt_points_bwd_tex": [_P, _wp, _P, _I32, _I64, _I32, _I32, _I32, _I32, _F, _I32, _I32, _P, _P, _wp, _P],
        "tt_points_bwd_x": [_P, _wp, _P, _I32, _I64, _I32, _I32, _I32, _I32, _F, _I32, _P, _P, _P, _P, _P],
        "tt_hashgrid_n_params": [ctypes.POINTER(HashGridCfg)],
        "tt_hashgrid_fwd": [_P, _I64, _P, ctypes.POINTER(HashGridCfg), _P, _P],
        "tt_hashgrid_bwd": [_P, _I64, _P, ctypes.POINTER(HashGridCfg), _P, _P],
        "tt_sample_uniform": [_I64, _I32, _F, _F, _P, _I32, _P, _P, _P],
        "tt_sample_importance": [_P, _P, _P, _I64, _I32, _I32, _F, _P, _F, _P, _I32, _P, _P, _P],
        "tt_grid_sample_2d_grad2": [_P] * 5 + [_I32] * 4 + [_I64, _I32, _I32] + [_P] * 3 + [_P],
        "tt_grid_sample_2d_grad2_typed": [_I32] + [_P] * 5 + [_I32] * 4 + [_I64, _I32, _I32] + [_P] * 3 + [_P],
        "tt_debug_poison_queue": [_P],
        "tt_composite_fwd": [_P, _P, _P, _P, _P, _I32, _P, _P, _I64, _I32, _I32, _I32] + [_P] * 6,
        "tt_composite_bwd": [_P, _P, _P, _P, _P, _I32, _P, _P, _I64, _I32, _I32, _I32] + [_P] * 11,
        "tt_render_eval": [_P, _wp, _P, _P, _P, _P, _cfgp, _F, _F] + [_P] * 7,
        "tt_eikonal_fwd": [_P, _I64, _P, _P],
        "tt_eikonal_bwd": [_P, _P, _I64, _P, _P],
        "tt_patch_composite_fwd": [_P, _P, _P] + [_I32] * 9 + [_P],
        "tt_patch_composite_bwd": [_P, _P, _P] + [_I32] * 9 + [_P],
    }
    for name, argtypes in optional.items():
        if name in SYMBOLS:
            getattr(lib, name).argtypes = argtypes
    for name in SYMBOLS[2:]:
        if name != "tt_source_hash":
            getattr(lib, name).restype = ctypes.c_int
    lib.tt_hashgrid_n_params.restype = ctypes.c_int64
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"{what} failed: {load().tt_strerror(status).decode()} ({status})")
