"""Dev tool (CPU): what does the plane-gradient scatter of the backward kernels see at the reference's TRAINING shape?

Replays the tile decomposition of the kernels (csrc/tt_device.h: TileGeom -- bw x bh pixel blocks x sb consecutive samples,
32 samples per tile) over PatchRenderer's two renders (42 x 42 global rays + a 40 x 40 patch of a 128 x 128 image, 193
importance samples from the oracle's sampler) and counts, per (plane, tile): active corner references, DISTINCT texels
(= the fewest 128-byte atomics any in-tile combine can issue), and the references that lose their slot under a slot
window of a given shape (torus hash of the texel coordinates, first claim wins, as scatter_claim does).

    python tools/scatter_sim.py            # table over tile shapes and window shapes
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O  # noqa: E402


def make_scene(R=256, n_view=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    cache = torch.randn(1, 6, 32, R, R, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    ro, rd, c2w, cd = O.make_cameras(n_view, 128, 128)
    return cache, sw, fw, ro, rd


def sample(cache, sw, fw, ro, rd, n_view):
    """193 importance-sampled intervals per ray (oracle sampler, not stratified)."""
    B, Hh, Ww, _ = ro.shape
    n_rays = B * Hh * Ww
    o, d = ro.reshape(-1, 3), rd.reshape(-1, 3)

    def sdf_fn(ts, te):
        tm = (ts + te) * 0.5
        pts = (o[:, None, :] + d[:, None, :] * tm[..., None]).reshape(B, -1, 3)
        with torch.no_grad():
            out = O.geometry_forward(pts, cache.repeat_interleave(n_view, 0), sw, fw, output_normal=False)
        return out["sdf"].reshape(n_rays, -1)

    ts, te = O.importance_sampling(sdf_fn, n_rays, 128, 64, 0.1, 4.0, 100.0, 1.732 * 2 / 64)
    return ts, te


def tile_ids(B, Hh, Ww, S, sb):
    """tile id of every (ray, sample): pixel block bw x bh (32 / sb rays) x sb consecutive samples."""
    nr = 32 // sb
    bw = {1: 1, 2: 2, 4: 2, 8: 4, 16: 4, 32: 8}[nr]
    bh = nr // bw
    y, x = np.meshgrid(np.arange(Hh), np.arange(Ww), indexing="ij")
    blk = (y // bh) * ((Ww + bw - 1) // bw) + (x // bw)
    nblk = ((Hh + bh - 1) // bh) * ((Ww + bw - 1) // bw)
    blk = (np.arange(B)[:, None, None] * nblk + blk[None]).reshape(-1)  # per ray
    k = np.arange(S) // sb
    nk = (S + sb - 1) // sb
    return (blk[:, None] * nk + k[None, :]).astype(np.int64)  # (n_rays, S)


def corner_texels(pts, R):
    """per plane: texel coordinates (x0, y0) of the top-left corner and in-bounds flags of the 4 corners"""
    X, Y, Z = pts[..., 0], pts[..., 1], pts[..., 2]
    out = []
    for (u, v) in ((X, Y), (X, Z), (Z, Y)):
        ix = ((u + 1) * R - 1) / 2
        iy = ((v + 1) * R - 1) / 2
        x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
        out.append((x0, y0))
    return out


def analyse(name, ro, rd, ts, te, R, sb, windows):
    B, Hh, Ww, _ = ro.shape
    S = ts.shape[1]
    tm = ((ts + te) * 0.5).numpy()
    pts = ro.reshape(-1, 1, 3).numpy() + rd.reshape(-1, 1, 3).numpy() * tm[..., None]
    tid = tile_ids(B, Hh, Ww, S, sb)
    res = {"refs": 0, "distinct": 0, "tiles": 0}
    for w in windows:
        res[w] = 0
    for (x0, y0) in corner_texels(pts, R):
        keys, tids = [], []
        for dy in (0, 1):
            for dx in (0, 1):
                x, y = x0 + dx, y0 + dy
                inb = (x >= 0) & (x < R) & (y >= 0) & (y < R)
                keys.append((y * R + x)[inb])
                tids.append(tid[inb])
        tex = np.concatenate(keys)
        t = np.concatenate(tids)
        res["refs"] += tex.size
        res["tiles"] += np.unique(t).size
        pair = t * (R * R) + tex
        upair, cnt = np.unique(pair, return_counts=True)
        res["distinct"] += upair.size
        ut, ux = upair // (R * R), upair % (R * R)
        uy, uxx = ux // R, ux % R
        for w in windows:
            wx, wy = w
            slot = (uy % wy) * wx + (uxx % wx)
            # per (tile, slot): the texel with the most references wins (upper bound of what first-claim-wins keeps)
            ts_key = ut * (wx * wy) + slot
            order = np.lexsort((-cnt, ts_key))
            first = np.ones(order.size, bool)
            first[1:] = ts_key[order][1:] != ts_key[order][:-1]
            lost = cnt[order][~first].sum()
            res[w] += lost
    r = res
    line = f"{name:28s} sb={sb:2d}  refs/plane-tile {r['refs'] / r['tiles']:6.1f}  distinct {r['distinct'] / r['tiles']:6.1f}"
    for w in windows:
        line += f"  lost[{w[0]}x{w[1]}] {100.0 * r[w] / r['refs']:5.1f}%"
    print(line, flush=True)
    return r


def main():
    R, n_view = 256, 4
    cache, sw, fw, ro, rd = make_scene(R, n_view)
    ds = 3
    g_o = torch.nn.functional.interpolate(ro.permute(0, 3, 1, 2), (128 // ds, 128 // ds), mode="bilinear").permute(0, 2, 3, 1).contiguous()
    g_d = torch.nn.functional.interpolate(rd.permute(0, 3, 1, 2), (128 // ds, 128 // ds), mode="bilinear").permute(0, 2, 3, 1).contiguous()
    p_o, p_d = ro[:, 44:84, 30:70].contiguous(), rd[:, 44:84, 30:70].contiguous()
    windows = [(8, 8), (16, 8), (16, 16)]
    for name, o, d in (("global 42x42", g_o, g_d), ("patch 40x40", p_o, p_d)):
        ts, te = sample(cache, sw, fw, o, d, n_view)
        for sb in (2, 4, 8, 16, 32):
            analyse(name, o, d, ts, te, R, sb, windows)


if __name__ == "__main__":
    main()
