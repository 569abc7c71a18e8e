"""CPU tests of the host side: plugin registry / configs / schedules, sampler contract, synthetic inputs,
C-ABI surface (header <-> library <-> ctypes agreement).  No GPU, no compute calls into the library."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

import triplaneturbo_amd as tt
from oracle import cpu_ref as O
from triplaneturbo_amd import _lib, parallel, registry, sampler, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_names_and_find():
    for name in ("few-step-triplane-dual-stable-diffusion", "generative-space-sdf-volume-renderer", "patch-renderer",
                 "no-material"):
        assert isinstance(tt.find(name), type)
    with pytest.raises(ValueError):
        tt.register("patch-renderer")(object)  # names of extensions conflict (threestudio/__init__.py:8-11)
    with pytest.raises(KeyError):
        tt.find("does-not-exist")


def test_config_fields_match_reference_names():
    R = tt.find("generative-space-sdf-volume-renderer")
    want = {"radius", "num_samples_per_ray", "randomized", "eval_chunk_size", "learned_variance_init",
            "cos_anneal_end_steps", "use_volsdf", "near_plane", "far_plane", "trainable_variance", "estimator",
            "grid_prune", "prune_alpha_threshold", "num_samples_per_ray_importance", "train_chunk_size",
            "rgb_grad_shrink", "normal_direction", "weights"}  # renderer :40-71 + base classes
    assert want == {f.name for f in R.Config.__dataclass_fields__.values()}
    d = R.Config()
    assert (d.num_samples_per_ray, d.learned_variance_init, d.estimator, d.near_plane) == (512, 0.3, "occgrid", 0.0)
    G = tt.find("few-step-triplane-dual-stable-diffusion")
    for f in ("n_feature_dims", "mlp_network_config", "normal_type", "sdf_bias", "sdf_bias_params", "rotate_planes",
              "split_channels", "geo_interpolate", "tex_interpolate", "radius", "isosurface_deformable_grid"):
        assert f in G.Config.__dataclass_fields__
    with pytest.raises(KeyError):
        registry.parse_structured(R.Config, {"not_a_field": 1})


def _modules(renderer_cfg=None):
    g = tt.find("few-step-triplane-dual-stable-diffusion")({})
    m = tt.find("no-material")({})
    b = tt.find("solid-color-background")({})
    cfg = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
               num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0, rgb_grad_shrink=[0, 1, 0.01, 20000])
    cfg.update(renderer_cfg or {})
    return g, m, b, cfg


def test_renderer_construction_and_schedules():
    g, m, b, cfg = _modules()
    r = tt.find("generative-space-sdf-volume-renderer")(cfg, geometry=g, material=m, background=b)
    assert abs(float(r.variance.inv_std) - 100.0) < 0.05  # exp(10 * 0.4605), yaml :137
    assert abs(r.render_step_size - 1.732 * 2 / 64) < 1e-12
    r.update_step(0, 0)
    assert r.rgb_grad_shrink == 1
    r.update_step(0, 10000)
    assert abs(r.rgb_grad_shrink - 0.505) < 1e-9
    r.update_step(0, 50000)
    assert abs(r.rgb_grad_shrink - 0.01) < 1e-12
    assert r.train().randomized is True and r.eval().randomized is False
    assert list(g.state_dict().keys())[1:] == [f"{n}.layers.{i}.weight" for n in ("sdf_network", "feature_network")
                                                for i in (0, 2, 4)]
    _, _, _, c = _modules({"trainable_variance": True})  # the reference class default (renderer :53,82): a trained parameter
    rt = tt.find("generative-space-sdf-volume-renderer")(c, geometry=g, material=m, background=b)
    assert rt.variance._inv_std.requires_grad and not r.variance._inv_std.requires_grad
    _, _, _, c = _modules({"estimator": "occgrid"})
    with pytest.raises(NotImplementedError):
        tt.find("generative-space-sdf-volume-renderer")(c, geometry=g, material=m, background=b)
    _, _, _, c = _modules({"use_volsdf": True})  # neus_volume_renderer.py:95-96 -> TT_R_VOLSDF (tests/test_volsdf.py)
    rv = tt.find("generative-space-sdf-volume-renderer")(c, geometry=g, material=m, background=b)
    assert rv._render_config().use_volsdf and not r._render_config().use_volsdf
    p = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                                   "base_renderer_type": "generative-space-sdf-volume-renderer",
                                   "base_renderer": cfg}, geometry=g, material=m, background=b)
    p.update_step(0, 10000)
    assert abs(p.base_renderer.rgb_grad_shrink - 0.505) < 1e-9


def test_C_schedule_matches_reference_semantics():
    assert registry.C(0.3, 0, 5) == 0.3
    assert registry.C([0, 1.0, 0.0, 20000], 0, 5000) == 0.75
    assert registry.C([1.0, 0.0, 100], 0, 50) == 0.5  # 3-element form gets a 0 start step
    assert registry.C([0, 0.0, 1.0, 10, 3.0, 20], 0, 15) == 2.0  # piecewise form
    assert registry.C([0, 1.0, 0.0, 2.0], 1, 999) == 0.5  # float end_step => epochs
    # three chained ramps (values of threestudio/utils/misc.py:69-104 on the same list), incl. the hand-over steps
    spec = [0, 0.0, 1.0, 10, 3.0, 20, 2.0, 40]
    want = {0: 0.0, 5: 0.5, 10: 1.0, 15: 2.0, 20: 3.0, 30: 2.5, 40: 2.0, 100: 2.0}
    for step, v in want.items():
        assert registry.C(spec, 0, step) == v, step
    assert abs(registry.C([10, 0.1, 1.0, 110], 0, 60, "exp") - 0.1 ** 0.5) < 1e-15
    with pytest.raises(ValueError):
        registry.C([0, 1.0, 2.0, 10], 0, 5, "cubic")
    with pytest.raises(AssertionError):
        registry.C([0, 1.0, 2.0, 10, 5.0], 0, 5)


def test_sampler_has_no_cpu_path():
    """The samplers are HIP kernels (tests/test_gpu_sampler.py checks them against the oracle's contract); asking
    for CPU intervals must fail loudly rather than fall back."""
    with pytest.raises(RuntimeError):
        sampler.uniform_intervals(5, 32, 0.1, 4.0, device="cpu")
    with pytest.raises(RuntimeError):
        sampler.importance_sampling(lambda a, b: a, 5, 32, 16, 0.1, 4.0, 100.0, 0.05, device="cpu")


def test_reference_training_yaml_loads_into_the_plugins():
    """Drop-in check at the config level: the reference's own training config (configs/TriplaneTurbo_v1.yaml) must
    instantiate our geometry / material / background / patch renderer with every key it sets (unknown keys raise).
    Reads /root/reference, so it only runs in the build container (skipped on the GPU box)."""
    path = "/root/reference/configs/TriplaneTurbo_v1.yaml"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    import yaml
    txt = re.sub(r"\$\{[^}]*\}", "1", open(path).read())  # OmegaConf interpolations -> a plain scalar
    s = yaml.safe_load(txt)["system"]
    g = tt.find(s["geometry_type"])(s["geometry"])
    m = tt.find(s["material_type"])(s["material"])
    b = tt.find(s["background_type"])(s["background"])
    r = tt.find(s["renderer_2nd_type"])(s["renderer_2nd"], geometry=g, material=m, background=b)
    base = r.base_renderer
    assert type(base).__name__ == "GenerativeSpaceSDFVolumeRenderer" and base.cfg.estimator == "importance"
    assert (base.cfg.num_samples_per_ray, base.cfg.num_samples_per_ray_importance) == (64, 128)
    assert r.cfg.patch_size == 40 and r.cfg.global_downsample == 3
    assert abs(float(base.variance.inv_std) - 100.0) < 0.1  # exp(10 * 0.4605)
    assert b.cfg.color_activation == "sigmoid-mipnerf" and b.encoding.n_output_dims == 16
    assert g.cfg.isosurface_deformable_grid and hasattr(g, "deformation_network")


def test_hashgrid_table_size_matches_oracle_levels():
    """tt_hashgrid_n_params is a host-only function: per-level sizes follow tcnn's rule restated in the oracle."""
    lib = _lib.load()
    for cfg in ((8, 2, 19, 4, 1.8114473285278132), (5, 4, 10, 3, 2.0), (16, 2, 19, 16, 1.3819), (3, 1, 14, 16, 1.5)):
        c = _lib.HashGridCfg(*cfg)
        _, total = O.hashgrid_levels(cfg[0], cfg[2], cfg[3], cfg[4])
        assert lib.tt_hashgrid_n_params(ctypes.byref(c)) == total * cfg[1], cfg
    assert lib.tt_hashgrid_n_params(ctypes.byref(_lib.HashGridCfg(17, 2, 19, 4, 2.0))) == -1  # TT_ERR_BAD_ARG
    assert lib.tt_hashgrid_n_params(ctypes.byref(_lib.HashGridCfg(8, 3, 19, 4, 2.0))) == -2  # TT_ERR_UNSUPPORTED


def test_synthetic_inputs_match_oracle_conventions():
    a = synthetic.make_cameras(4, 6, 10, azimuth_start_deg=20.0)
    b = O.make_cameras(4, 6, 10, azimuth_start_deg=20.0)
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a[1].norm(dim=-1), torch.ones(4, 6, 10))
    assert abs(float(a[3][0]) - 0.9 / torch.tan(torch.tensor(torch.pi / 6)).item()) < 1e-6
    ts, te = synthetic.uniform_intervals(3, 128, 0.1, 4.0)
    to, _ = O.uniform_intervals(3, 128, 0.1, 4.0)
    assert torch.equal(ts, to)


def test_shard_prompts_partition():
    for n, w in ((64, 8), (10, 4), (3, 8)):
        parts = [list(parallel.shard_prompts(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


# ---------------- C ABI surface ----------------
def _header_symbols():
    src = open(os.path.join(ROOT, "include", "tt_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    path = _lib.build()  # hipcc cross-compiles for gfx950 without a GPU; no-op when up to date
    lib = ctypes.CDLL(path)
    syms = _header_symbols()
    assert set(syms) == set(_lib.SYMBOLS), (syms, _lib.SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), s
    lib.tt_abi_version.restype = ctypes.c_int
    lib.tt_strerror.restype = ctypes.c_char_p
    assert lib.tt_abi_version() == _lib._expected_abi() == 17
    assert b"bad argument" in lib.tt_strerror(-1)
    # the binary carries the hash of the sources + flags it was built from, and that is what "up to date" means
    lib.tt_source_hash.restype = ctypes.c_char_p
    assert lib.tt_source_hash().decode() == _lib.source_hash() == _lib.embedded_hash(path)
    assert not _lib.needs_build(path)


def test_stale_library_is_rebuilt_whatever_its_mtime(tmp_path, monkeypatch):
    """needs_build compares the embedded source hash, not file times: a library whose sources changed is stale even when
    it is NEWER than every source (a pushed tree, a checkout), and a library without a hash is always stale."""
    path = _lib.build()
    fake = tmp_path / "libtt_fake.so"
    blob = open(path, "rb").read()
    fake.write_bytes(blob)
    assert not _lib.needs_build(str(fake))
    good = _lib.embedded_hash(path).encode()
    fake.write_bytes(blob.replace(good, b"0" * 64))  # same file, other sources
    os.utime(fake, (2e9, 2e9))                       # ... and far newer than the tree
    assert _lib.needs_build(str(fake))
    fake.write_bytes(b"no hash in here")
    assert _lib.needs_build(str(fake))
    # a change of the build flags is a change of the binary too
    assert _lib.source_hash(defines=["-DX"]) != _lib.source_hash()
    assert _lib.source_hash(tuning=True) != _lib.source_hash()


def test_load_refuses_an_abi_mismatch(monkeypatch):
    _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_expected_abi", lambda: 999)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.load()


def test_ctypes_struct_layout_matches_header(tmp_path):
    code = r'''
#include <stdio.h>
#include <stddef.h>
#include "tt_abi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(tt_render_cfg), offsetof(tt_render_cfg, n_rays),
         offsetof(tt_render_cfg, radius), offsetof(tt_render_cfg, flags), offsetof(tt_render_cfg, image_w),
         offsetof(tt_render_cfg, tile_chunk), sizeof(tt_mlp_weights), sizeof(tt_mlp_grads),
         offsetof(tt_render_cfg, skip_eps_geo));
  return 0; }'''
    src = tmp_path / "t.c"
    src.write_text(code)
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    c = _lib.RenderCfg
    got = [ctypes.sizeof(c), c.n_rays.offset, c.radius.offset, c.flags.offset, c.image_w.offset, c.tile_chunk.offset,
           ctypes.sizeof(_lib.MlpWeights), ctypes.sizeof(_lib.MlpWeights), c.skip_eps_geo.offset]
    assert [int(x) for x in out] == got


def test_ops_refuse_cpu_tensors():
    from triplaneturbo_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.planes_pack(torch.zeros(1, 6, 32, 8, 8))


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Argument validation happens before any HIP call, so the error paths are testable without a device:
    null pointers / inconsistent shapes -> TT_ERR_BAD_ARG (-1), unsupported shapes -> TT_ERR_UNSUPPORTED (-2)."""
    _lib.build()
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(8)  # never dereferenced: validation fails first
    assert lib.tt_planes_pack(null, one, 1, 8, 8, null) == -1
    assert lib.tt_planes_pack(one, one, 1, 8, 16, null) == -2  # non-square planes (rotation v1 transposes)
    assert lib.tt_planes_unpack_grad(one, null, 1, 8, 8, 1, null) == -1
    w = _lib.MlpWeights(one, one, one, one, one, one)
    assert lib.tt_query_points(one, ctypes.byref(w), one, 3, 10, 2, 2, 8, 8, 1.0, 0.5, 3, one, one, one, null) == -1
    cfg = _lib.RenderCfg(1, 1, 8, 8, 16, 4, 16, 1.0, 0.5, 100.0, 1.0, 1.0, 0, 0, 0, 1)
    bad = _lib.RenderCfg(1, 1, 8, 8, 16, 4, 17, 1.0, 0.5, 100.0, 1.0, 1.0, 0, 0, 0, 1)  # n_rays != views*rays_per_view
    args = [one] * 11
    assert lib.tt_render_fwd(one, ctypes.byref(w), one, one, one, one, ctypes.byref(bad), *args, null) == -1
    assert lib.tt_render_fwd(one, ctypes.byref(w), one, one, one, one, ctypes.byref(cfg), *([null] + [one] * 10),
                             null) == -1
    assert lib.tt_decode_rays(one, ctypes.byref(w), one, one, one, one, ctypes.byref(cfg), 1, one, null, null,
                              null) == -1  # TT_Q_NORMAL without an sdf_grad buffer
    assert lib.tt_grid_sample_2d_grad2(one, one, one, one, one, 1, 4, 8, 8, 5, 2, 0, one, one, one, null) == -2  # reflection
    assert lib.tt_grid_sample_2d_grad2_typed(7, one, one, one, one, one, 1, 4, 8, 8, 5, 0, 0, one, one, one, null) == -2
    # packed planes of 4 GB and more (texels are addressed with 32-bit byte offsets): 86 prompts x 6 x 256^2 x 128 B
    big = _lib.RenderCfg(86, 1, 256, 256, 16, 4, 86 * 16, 1.0, 0.5, 100.0, 1.0, 1.0, 0, 0, 0, 1)
    ok = _lib.RenderCfg(85, 1, 256, 256, 16, 4, 85 * 16, 1.0, 0.5, 100.0, 1.0, 1.0, 0, 0, 0, 1)
    assert lib.tt_render_fwd(one, ctypes.byref(w), one, one, one, one, ctypes.byref(big), *args, null) == -2
    assert lib.tt_render_fwd(one, ctypes.byref(w), one, one, one, one, ctypes.byref(ok), *([null] + [one] * 10),
                             null) == -1  # passes the size check, then fails on the null output
    assert b"unsupported" in lib.tt_strerror(-2)
    # sampler placement enum: only TT_PLACE_TT (0) / TT_PLACE_CENTER (1)
    assert lib.tt_sample_uniform(4, 8, 0.1, 4.0, null, 2, one, one, null) == -1
    assert lib.tt_sample_importance(one, one, one, 4, 8, 4, 100.0, null, 0.05, null, 7, one, one, null) == -1
    assert lib.tt_sample_importance(one, one, one, 4, 8, 4, 0.0, null, 0.05, null, 0, one, one, null) == -1  # no inv_std at all


def test_product_library_never_reads_the_environment():
    """The tuning / ablation hooks (TT_DEBUG_FLAGS, TT_SB, ...) exist only in the -DTT_TUNING dev build: the product
    library must not import getenv at all."""
    path = _lib.build()
    out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out


def test_geometry_decode_applies_split_channels_v1():
    """few_step...:187-196: geo planes keep the first C channels, tex planes the last C of the generator's 2C."""
    from triplaneturbo_amd.geometry import StableDiffusionTriplaneDualAttention

    class Gen(torch.nn.Module):
        def forward_decode(self, latents):
            return latents

    g = StableDiffusionTriplaneDualAttention({}, space_generator=Gen())
    x = torch.randn(2, 6, 64, 4, 4)
    y = g.decode(x)
    assert y.shape == (2, 6, 32, 4, 4)
    assert torch.equal(y[:, :3], x[:, :3, :32]) and torch.equal(y[:, 3:], x[:, 3:, 32:])


def test_no_kernel_selects_on_a_scalar_alu_combination_of_fresh_compare_masks(tmp_path):
    """Mechanical guard of DESIGN.md section 6 (stale lane-mask bits after v_cmp -> s_and_b64 / s_or_b64 -> v_cndmask):
    tools/mask_hazard_lint.py disassembles the device code of the PRODUCT library and must find no such sequence in any
    kernel (window of 4 instructions behind the combination, masks at most 16 instructions old) -- and it must FLAG the
    negative control, the textbook in-bounds logic `in = bx && by; w = in ? wx * wy : 0` that corners_setup replaced
    (tools/mask_hazard_probe.hip), so a revert of the 0/1-factor convention (csrc/tt_mask.h) cannot pass silently."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mask_hazard_lint as L
    if not os.path.exists(L.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    found = L.lint(_lib.build(), window=4, fresh=16)
    assert not found, {k: v[:2] for k, v in found.items()}
    probe = tmp_path / "probe.so"
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                    "-fno-gpu-rdc", os.path.join(ROOT, "tools", "mask_hazard_probe.hip"), "-o", str(probe)], check=True,
                   cwd=str(tmp_path))
    flagged = L.lint(str(probe), window=4, fresh=16)
    assert any("k_textbook_corners" in k for k in flagged), flagged


def test_load_never_binds_a_stale_library_silently(monkeypatch):
    """_lib.load() compares the source hash embedded in libtt_hip.so with the hash of the checkout and rebuilds (or raises
    when there is no hipcc) on a mismatch -- a plugin user who only imports the package must not get a library built from
    other sources."""
    _lib.build()
    calls = []
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "source_hash", lambda *a, **k: "0" * 64)
    monkeypatch.setattr(_lib, "build", lambda *a, **k: calls.append(1) or _lib.LIB_PATH)
    _lib.load()
    assert calls == [1]
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    with pytest.raises(RuntimeError, match="different sources"):
        _lib.load()


def test_concurrent_builds_are_serialised(tmp_path):
    """load() rebuilds a stale library, so a multi-rank launch on an edited checkout sends every rank into
    build() at once.  They must serialise on the lock file and only ONE of them may compile: the second finds the library
    up to date.  (A stand-in for hipcc that logs its invocations and copies the hash marker; no GPU, no real compile.)"""
    import subprocess
    import sys
    import textwrap
    from triplaneturbo_amd import _lib
    fake = tmp_path / "fake_hipcc"
    log = tmp_path / "calls.log"
    fake.write_text(textwrap.dedent(f"""\
        #!{sys.executable}
        import sys, time, re
        a = sys.argv[1:]
        open({str(log)!r}, "a").write("call\\n")
        time.sleep(0.3)
        out = a[a.index("-o") + 1]
        if "-c" in a:
            h = [x for x in a if x.startswith("-DTT_SOURCE_HASH_STR=")][0].split("=", 1)[1].strip('"')
            open(out, "w").write("TT_SOURCE_HASH=" + h)
        else:
            objs = [x for x in a if x.endswith(".o")]
            open(out, "w").write(open(objs[0]).read())
        """))
    fake.chmod(0o755)
    variant = f"locktest{os.getpid()}"
    code = (f"from triplaneturbo_amd import _lib; print(_lib.build(variant={variant!r}, defines=['-DLOCKTEST'], "
            f"source_flags={{}}))")
    env = dict(os.environ, HIPCC=str(fake), PYTHONPATH=ROOT)
    out_path = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libtt_hip_{variant}.so")
    try:
        procs = [subprocess.Popen([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True) for _ in range(3)]
        res = [p.communicate(timeout=120) for p in procs]
        assert all(p.returncode == 0 for p in procs), res
        n_calls = len(open(log).read().split())
        assert n_calls == len(_lib._sources()) + 1, (n_calls, res)  # one compile per unit + one link, ONCE
        assert _lib.embedded_hash(out_path) == _lib.source_hash(False, ["-DLOCKTEST"], {})
        leftovers = [f for f in os.listdir(os.path.dirname(out_path)) if f.endswith(".tmp")]
        assert not leftovers, leftovers
    finally:
        for f in (out_path, os.path.join(os.path.dirname(out_path), "build", os.path.basename(out_path) + ".lock")):
            if os.path.exists(f):
                os.unlink(f)


def test_bench_stdout_is_one_json_line():
    """bench.py: claim_stdout() re-points file descriptor 1 to stderr and keeps a private duplicate for the result line, so
    whatever native libraries write to fd 1 (`[Gloo] Rank ...`, RCCL debug output) cannot reach the caller's stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "os.write(1, b'[Gloo] Rank 0 is connected to 1 peer ranks\\n'); print('python-level chatter'); "
            "os.system('echo from a child process'); bench.emit({'metric': 'm', 'value': 1.5})" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("\n") == 1 and json.loads(r.stdout) == {"metric": "m", "value": 1.5}, r.stdout
    for s in ("[Gloo] Rank 0", "python-level chatter", "from a child process"):
        assert s in r.stderr
