// tt_device.h -- device-side building blocks shared by the forward and backward kernels.
//
// gfx950 / CDNA4 only.  Everything here is wave64 code built around one convention:
//
//   A wave owns a TILE of 32 samples.  Lane l = (j, hi) with j = l & 31 (sample within the tile) and
//   hi = l >> 5 (which half of every feature/hidden vector the lane holds).  A length-32n vector v is
//   held as 16n registers per lane with   reg r  <->  element  LIDX(r, hi) = (r&3) + 8*(r>>2) + 4*hi.
//
// That map is exactly the C/D register layout of v_mfma_f32_32x32x2_f32 (col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so when the per-point MLP is written "transposed"
// (D[hidden x samples] = W[hidden x k] * X[k x samples], weights as the A operand, samples as the 32
// columns of B) the output registers of one layer ARE the B operand registers of the next layer:
// k-step r of the next layer pairs element LIDX(r,0) (lanes 0-31) with LIDX(r,1) (lanes 32-63), and the
// A operand simply reads the matching weight column.  No shuffles, no LDS round trip between layers,
// exact fp32 (the f32 MFMA is a k-ordered fmaf chain).  The plane gather uses the same map: lane (j,hi)
// loads the 16-byte chunks {hi, hi+2, hi+4, hi+6} of each 128-byte channels-last texel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// profiling-only ablations (results are then WRONG by construction).  They exist only in a -DTT_TUNING build of the
// library (tools/: `_lib.build(tuning=True)`), which also reads the TT_DEBUG_FLAGS / TT_SB / TT_CHUNK / TT_UNIT /
// TT_ORDER environment variables; the product build compiles every one of these tests to `false` and never calls
// getenv.
#ifdef TT_TUNING
#define TT_DBG(flags, bit) (((flags) & (bit)) != 0)
#else
#define TT_DBG(flags, bit) false
#endif
#define TT_DBG_NO_SCATTER 0x100
#define TT_DBG_NO_WGRAD 0x200
#define TT_DBG_NO_GATHER 0x400
#define TT_DBG_NO_MLP 0x800
#define TT_DBG_NO_STORE 0x1000
#define TT_DBG_NO_ATOMICS 0x2000 /* plane-gradient flush atomics issued but dropped (buffer range check) */
// (0x2000 no combine, 0x4000 no global atomics, 0x8000 no scatter MFMA, 0x10000 no claims: ablations of the
// pre-pipelining scatter, results in profiles/r02_ablations.txt; the straight-line scatter_planes has no switches)

#define TT_C 32
#define TT_HID 64
#define TT_TILE 32

#define LIDX(r, hi) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hi))

#include "tt_mask.h"

// ---- LDS image of the MLP weights: row-major, row stride = cols + 4 floats -----------------------
// (+4 keeps 16-byte alignment and makes the per-lane-row ds_read_b128 of mv_fwd conflict-free:
//  36*i, 68*i, 100*i mod 64 hit 16 distinct 4-bank slots for the 16 lanes of a b128 lane group.)
#define W1S 36
#define W2S 68
#define V1S 100
#define V2S 68
#define OFF_W1 0
#define OFF_W2 (OFF_W1 + 64 * W1S)
#define OFF_W3 (OFF_W2 + 64 * W2S)
#define OFF_V1 (OFF_W3 + 64)
#define OFF_V2 (OFF_V1 + 64 * V1S)
#define OFF_V3 (OFF_V2 + 64 * V2S)
#define LDS_W_FLOATS (OFF_V3 + 3 * 64)
#define LDS_GEO_FLOATS (OFF_W3 + 64) /* W1, W2, w3 only */

struct MlpPtrs {
    const float* w1;
    const float* w2;
    const float* w3;
    const float* v1;
    const float* v2;
    const float* v3;
};

__device__ __forceinline__ void lds_load_matrix(float* dst, const float* __restrict__ src, int rows, int cols,
                                                int stride) {
    for (int e = threadIdx.x; e < rows * cols; e += blockDim.x) {
        int r = e / cols, c = e - r * cols;
        dst[r * stride + c] = src[e];
    }
}

__device__ __forceinline__ void lds_load_geo_weights(float* L, const MlpPtrs& w) {
    lds_load_matrix(L + OFF_W1, w.w1, 64, 32, W1S);
    lds_load_matrix(L + OFF_W2, w.w2, 64, 64, W2S);
    lds_load_matrix(L + OFF_W3, w.w3, 1, 64, 64);
}
__device__ __forceinline__ void lds_load_tex_weights(float* L, const MlpPtrs& w) {
    lds_load_matrix(L + OFF_V1, w.v1, 64, 96, V1S);
    lds_load_matrix(L + OFF_V2, w.v2, 64, 64, V2S);
    lds_load_matrix(L + OFF_V3, w.v3, 3, 64, 64);
}

// ---- MFMA mat-vec products on a 32-sample tile -----------------------------------------------------
// y[NOUT] = W[NOUT][NIN] * x[NIN]   (W in LDS, row stride NIN+4).  x, y in the LIDX register layout.
// All NOUT/32 row tiles advance together: their accumulator chains are independent, so one wave keeps the
// matrix pipe busy (a single dependent chain stalls on the 64-cycle result latency plus the LDS wait), and the
// A operands of k-group g+1 are fetched while group g multiplies.
template <int NOUT, int NIN>
__device__ __forceinline__ void mv_fwd(const float* Wl, const float (&x)[NIN / 2], float (&y)[NOUT / 2], int i,
                                       int hi) {
    constexpr int MT = NOUT / 32;
    f32x16 acc[MT];
    f32x4 a_cur[MT], a_nxt[MT];
    const float* row = Wl + i * (NIN + 4) + 4 * hi;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        a_cur[m] = *reinterpret_cast<const f32x4*>(row + 32 * m * (NIN + 4));
    }
#pragma unroll
    for (int g = 0; g < NIN / 8; ++g) {
        if (g + 1 < NIN / 8) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                a_nxt[m] = *reinterpret_cast<const f32x4*>(row + 32 * m * (NIN + 4) + 8 * (g + 1));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m][k], x[4 * g + k], acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) y[16 * m + k] = acc[m][k];
}

// y[NOUT] = W[NIN][NOUT]^T * x[NIN]   (W in LDS as stored, row stride STRIDE = NOUT+4).
// STRIDE != NOUT+4 selects a 32m-column slice of a wider stored matrix (Wl already offset to its first column).
// With a single row tile (NOUT = 32) the k range is split into two independent accumulator chains.
template <int NOUT, int NIN, int STRIDE = NOUT + 4>
__device__ __forceinline__ void mv_bwd(const float* Wl, const float (&x)[NIN / 2], float (&y)[NOUT / 2], int i,
                                       int hi) {
    constexpr int MT = NOUT / 32;
    const float* base = Wl + 4 * hi * STRIDE + i;
    if (MT == 1) {
        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc1 = acc0;
#pragma unroll
        for (int r = 0; r < NIN / 4; ++r) {
            const float a0 = base[LIDX(r, 0) * STRIDE], a1 = base[LIDX(r + NIN / 4, 0) * STRIDE];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x[r], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x[r + NIN / 4], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = acc0[k] + acc1[k];
    } else {
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
            acc[m] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NIN / 2; ++r) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float a = base[LIDX(r, 0) * STRIDE + 32 * m];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[r], acc[m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) y[16 * m + k] = acc[m][k];
    }
}

// dot of a register vector with an LDS vector w[N] (same LIDX layout), summed over both halves.
template <int N>
__device__ __forceinline__ float dot_lds(const float* wl, const float (&x)[N / 2], int hi) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < N / 8; ++g) {
        f32x4 a = *reinterpret_cast<const f32x4*>(wl + 8 * g + 4 * hi);
        s = fmaf(a[0], x[4 * g + 0], s);
        s = fmaf(a[1], x[4 * g + 1], s);
        s = fmaf(a[2], x[4 * g + 2], s);
        s = fmaf(a[3], x[4 * g + 3], s);
    }
    return s + __shfl_xor(s, 32);
}

// ---- bilinear corner set-up (ATen GridSampler.cuh:23-31; zeros padding, align_corners=False) -------
struct Corners {
    int off[4];   // texel index y*W+x (of the nearest in-bounds texel when out of bounds: weight 0)
    float w[4];   // bilinear weights nw, ne, sw, se (0 when out of bounds)
    float du[4];  // d w / d ix   (0 when out of bounds)
    float dv[4];  // d w / d iy
    int hs[4];    // 16x16 torus hash of the texel (y&15)*16 + (x&15): entry of the scatter's texel table
    bool any;     // any corner in bounds
    int inmask;   // bit k: corner k in bounds (d2 w_k / d ix d iy = +1, -1, -1, +1 there, 0 elsewhere)
};

#pragma clang fp contract(off)
// In-bounds handling WITHOUT lane-mask logic: the textbook form
//     in_k = bx && by ;  w_k = in_k ? wx * wy : 0 ;  off_k = in_k ? y * W + x : 0
// compiles to v_cmp -> s_and_b64 (SALU combination of lane masks) -> v_cndmask, and on MI355X that sequence was
// observed to deliver STALE mask bits for lanes 48..63 to the first select when the SIMD runs a single wave (tail of a
// kernel): w[2] of the upper half-wave wrong in ~5 of 9375 tiles per launch, nondeterministically, while the offset
// selected by the same mask two instructions later was right (tools/stress_export.py; gone with
// -mllvm -amdgpu-waitcnt-forcezero and whenever the masks are re-materialised by VALU compares before the selects, with
// or without s_nops; isolated sequences in tools/sgpr_hazard_probe.hip do not reproduce it).  Here every flag is a float 0/1 made by ONE compare + select (VALU -> VALU through VCC, a hazard hipcc
// handles) and combined by multiplication; results are bit-identical to the textbook form (x * 1 = x, finite * 0 = 0).
__device__ __forceinline__ void corners_setup(float gx, float gy, int H, int W, bool valid, Corners& c) {
    // a non-finite coordinate is out of bounds (ATen's within_bounds_2d is false for NaN: the sample contributes 0; with
    // 0/1 factors NaN * 0 would be NaN): send it far outside, one compare + select per axis
    gx = __builtin_fabsf(gx) < 1e30f ? gx : 4.f;
    gy = __builtin_fabsf(gy) < 1e30f ? gy : 4.f;
    // same op order as the reference: ((coord + 1) * size - 1) / 2
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    float fx = floorf(ix), fy = floorf(iy);
    float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
    float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
    int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f);
    int y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    x0 = valid ? x0 : -2;  // an invalid lane is out of bounds in x
    const float bx0 = (unsigned)x0 < (unsigned)W ? 1.f : 0.f, bx1 = (unsigned)(x0 + 1) < (unsigned)W ? 1.f : 0.f;
    const float by0 = (unsigned)y0 < (unsigned)H ? 1.f : 0.f, by1 = (unsigned)(y0 + 1) < (unsigned)H ? 1.f : 0.f;
    const float mx0 = wx0 * bx0, mx1 = wx1 * bx1, my0 = wy0 * by0, my1 = wy1 * by1;  // weights masked per axis
    c.w[0] = mx0 * my0;
    c.w[1] = mx1 * my0;
    c.w[2] = mx0 * my1;
    c.w[3] = mx1 * my1;
    c.du[0] = -(my0 * bx0);
    c.du[1] = my0 * bx1;
    c.du[2] = -(my1 * bx0);
    c.du[3] = my1 * bx1;
    c.dv[0] = -(mx0 * by0);
    c.dv[1] = -(mx1 * by0);
    c.dv[2] = mx0 * by1;
    c.dv[3] = mx1 * by1;
    // texel index: any VALID texel will do for an out-of-bounds corner (its weight, du and dv are 0 and nothing else
    // looks at it): clamp instead of select
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1) * W, yc1 = min(max(y0 + 1, 0), H - 1) * W;
    c.off[0] = yc0 + xc0;
    c.off[1] = yc0 + xc1;
    c.off[2] = yc1 + xc0;
    c.off[3] = yc1 + xc1;
    const float i0 = bx0 * by0, i1 = bx1 * by0, i2 = bx0 * by1, i3 = bx1 * by1;  // (only the points-gradient kernel)
    c.hs[0] = ((y0 & 15) << 4) | (x0 & 15);
    c.hs[1] = ((y0 & 15) << 4) | ((x0 + 1) & 15);
    c.hs[2] = (((y0 + 1) & 15) << 4) | (x0 & 15);
    c.hs[3] = (((y0 + 1) & 15) << 4) | ((x0 + 1) & 15);
    c.any = (bx0 + bx1) * (by0 + by1) != 0.f;
    c.inmask = (int)i0 | ((int)i1 << 1) | ((int)i2 << 2) | ((int)i3 << 3);
}

// world position -> plane-sampling coordinates, mirroring the reference's fp32 op order:
//   scale_tensor(x, (-radius, radius), (-1, 1))  (threestudio/utils/ops.py:27-38)  then * (2/box_warp) = * 1
__device__ __forceinline__ float scale_coord(float x, float radius) {
    float t = (x - (-radius)) / (radius - (-radius));
    return t * 2.f + (-1.f);
}

__device__ __forceinline__ void sample_position(float ox, float oy, float oz, float dx, float dy, float dz, float ts,
                                                float te, float& tm, float& px, float& py, float& pz) {
    tm = (ts + te) / 2.f;  // renderer :337
    px = ox + dx * tm;     // renderer :338
    py = oy + dy * tm;
    pz = oz + dz * tm;
}

__device__ __forceinline__ float sphere_bias(float px, float py, float pz, float bias_radius, float& nrm) {
    nrm = sqrtf((px * px + py * py) + pz * pz);  // few_step...:146-149
    return nrm - bias_radius;
}
#pragma clang fp contract(fast)

// plane p uses (u,v): p0 (x,y), p1 (x,z), p2 (z,y)  (geometry/utils.py:46-63,111-125)
#define PLANE_U(p, X, Y, Z) ((p) == 2 ? (Z) : (X))
#define PLANE_V(p, X, Y, Z) ((p) == 1 ? (Z) : (Y))

// ---- gathers, one lane per sample (points-gradient and small kernels; the render kernels use the coalesced ones below) ----
// geometry planes (v1 = sum over planes): f[16] and, if NEED_J, J = d f / d(world xyz) (3 x 16)
template <bool NEED_J>
__device__ __forceinline__ bool gather_geo(const float* __restrict__ planes, int H, int W, float X, float Y, float Z,
                                           bool valid, float jscale_u, float jscale_v, int hi, float (&f)[16],
                                           float (&jx)[16], float (&jy)[16], float (&jz)[16], int dbg = 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        f[r] = 0.f;
        jx[r] = 0.f;
        jy[r] = 0.f;
        jz[r] = 0.f;
    }
    bool any = false;
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        Corners c;
        corners_setup(PLANE_U(p, X, Y, Z), PLANE_V(p, X, Y, Z), H, W, valid, c);
        if (!__any(c.any)) continue;  // exact: every contribution of this plane is 0 for the whole tile
        any = any || c.any;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4* t = reinterpret_cast<const f32x4*>(planes + (p * HW + (size_t)c.off[k]) * TT_C) + hi;
            f32x4 v[4];
            if (TT_DBG(dbg, TT_DBG_NO_GATHER)) {
                const f32x4 z = {c.w[k], c.du[k], c.dv[k], X};
                v[0] = v[1] = v[2] = v[3] = z;
            } else {
                v[0] = t[0];
                v[1] = t[2];
                v[2] = t[4];
                v[3] = t[6];
            }
            const float wk = c.w[k], a = c.du[k] * jscale_u, b = c.dv[k] * jscale_v;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tv = v[q][e];
                    f[4 * q + e] = fmaf(wk, tv, f[4 * q + e]);
                    if (NEED_J) {
                        if (p == 0) {
                            jx[4 * q + e] = fmaf(a, tv, jx[4 * q + e]);
                            jy[4 * q + e] = fmaf(b, tv, jy[4 * q + e]);
                        } else if (p == 1) {
                            jx[4 * q + e] = fmaf(a, tv, jx[4 * q + e]);
                            jz[4 * q + e] = fmaf(b, tv, jz[4 * q + e]);
                        } else {
                            jz[4 * q + e] = fmaf(a, tv, jz[4 * q + e]);
                            jy[4 * q + e] = fmaf(b, tv, jy[4 * q + e]);
                        }
                    }
                }
        }
    }
    return any;
}

// texture planes (v2 = concat over planes): e[48], e[16p + r] <-> channel LIDX(r,hi) of plane p
__device__ __forceinline__ bool gather_tex(const float* __restrict__ planes, int H, int W, float X, float Y, float Z,
                                           bool valid, int hi, float (&e)[48], int dbg = 0) {
#pragma unroll
    for (int r = 0; r < 48; ++r) e[r] = 0.f;
    bool any = false;
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        Corners c;
        corners_setup(PLANE_U(p, X, Y, Z), PLANE_V(p, X, Y, Z), H, W, valid, c);
        if (!__any(c.any)) continue;
        any = any || c.any;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4* t = reinterpret_cast<const f32x4*>(planes + ((3 + p) * HW + (size_t)c.off[k]) * TT_C) + hi;
            f32x4 v[4];
            if (TT_DBG(dbg, TT_DBG_NO_GATHER)) {
                const f32x4 z = {c.w[k], Y, Z, X};
                v[0] = v[1] = v[2] = v[3] = z;
            } else {
                v[0] = t[0];
                v[1] = t[2];
                v[2] = t[4];
                v[3] = t[6];
            }
            const float wk = c.w[k];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) e[16 * p + 4 * q + ee] = fmaf(wk, v[q][ee], e[16 * p + 4 * q + ee]);
        }
    }
    return any;
}

// Work accounting (tt_render_cfg.stats, measurement only).  NOTHING is kept in registers: the pointer is a kernel argument
// (wave-uniform), so a production launch (null) pays one scalar branch per counting site; a measuring launch counts in a
// wave-private LDS slot (one lane, plain read-modify-write) and adds the slot to the caller's counters with three atomics
// per wave at kernel end.  (Round 4 first kept per-wave counters in registers: 0.09 ms of the 7.85 ms step -- the decode
// kernels have no registers to spare; then one global atomic per site: free when off, but 15 ms per measuring launch.)
struct TileStats {
    unsigned long long* p;  // null, or the caller's 4 counters: [0] visited, [1] executed, [2] in-bounds pairs
    unsigned* w;            // this wave's LDS slot (4 ints) when p != null
};
enum { TT_STAT_VISITED = 0, TT_STAT_EXECUTED = 1, TT_STAT_INBOUNDS = 2 };
__device__ __forceinline__ TileStats tile_stats(uint64_t* stats64) {
    TileStats st = {nullptr, nullptr};
#ifndef TT_NO_STATS  // (dev A/B only, tools/build_variants.py: what the counting costs)
    __shared__ unsigned tt_stat_slots[16 * 4];  // <= 16 waves per workgroup
    st.p = reinterpret_cast<unsigned long long*>(stats64);
    if (st.p) {
        st.w = tt_stat_slots + 4 * __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if ((threadIdx.x & 63) < 4) st.w[threadIdx.x & 63] = 0u;
    }
#endif
    return st;
}
// wave-uniform `v`; one lane counts
__device__ __forceinline__ void tile_stat(const TileStats& st, int which, unsigned v = 1) {
    if (st.p) {
        if ((threadIdx.x & 63) == 0) st.w[which] += v;
    }
}
__device__ __forceinline__ unsigned* tile_stat_ptr(const TileStats& st, int which) { return st.p ? st.w + which : nullptr; }
__device__ __forceinline__ void tile_stats_flush(const TileStats& st) {
    if (st.p) {
        const int l = threadIdx.x & 63;
        if (l < 3) atomicAdd(st.p + l, (unsigned long long)st.w[l]);
    }
}

// ---- coalesced gathers ------------------------------------------------------------------------------------------------
// The gathers above give every lane its own sample: one wave-instruction touches 32 different 128-byte texel lines, 32
// bytes of each, and a line is fetched by four instructions.  The L1 / texture-address path, not the latency, is what
// bounds them (a timing experiment with this access pattern shortened the gather phase by 23 %; issuing all loads at once
// did nothing).  Here 8 CONSECUTIVE LANES read one texel (lane & 7 = 16-byte chunk): an instruction touches 8 whole lines.
// The bilinear set-up is still done once per sample in the (sample, half) lane layout and handed over through a small
// LDS table; lane (js, c) = (lane >> 3, lane & 7) then accumulates channels 4c..4c+3 of samples 8n + js, n = 0..3, in the
// same corner order as before (bit-identical sums), and the result goes back to the LIDX register layout through a
// [sample][36]-float LDS tile (conflict-free both ways).
typedef int ti32x4 __attribute__((ext_vector_type(4)));
#define GC_TABLE_FLOATS(NW) (3 * 32 * 4 * (1 + (NW))) /* 3 planes: offsets + NW weight sets */
#define GC_TILE_FLOATS (32 * 36)

#define GC_PLANE_TABLE_FLOATS (32 * 16) /* one plane: offsets + up to 3 weight sets */
#define GC_SCRATCH_FLOATS (GC_PLANE_TABLE_FLOATS + GC_TILE_FLOATS) /* forward kernels: 6.5 KB per wave */

// texel index -> address of this lane's 16-byte chunk: uniform base + 32-bit BYTE offset (texel << 7 | chunk byte), one
// VALU instruction and the scalar-base form of the load instead of 64-bit address arithmetic per load (the host
// refuses packed plane buffers of 4 GB and more: tt_planes_too_large).  Backward kernels only (A/B on one box:
// backward -0.03 / -0.05 ms, forward +0.09 ms).
__device__ __forceinline__ const f32x4* gc_addr(const float* __restrict__ planes, int texel, unsigned cbyte) {
    return reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(planes) + (((unsigned)texel << 7) | cbyte));
}

// lane (js, c) loads channels 4c..4c+3 of the 16 texels (4 samples x 4 corners) it serves in plane-table order:
// the 16 loads are issued back to back -- one memory round trip per plane
__device__ __forceinline__ void gc_load16(const float* __restrict__ pl, const int* Toff, int js, f32x4 (&t)[4][4]) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const ti32x4 o4 = *reinterpret_cast<const ti32x4*>(Toff + (8 * n + js) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) t[n][k] = *reinterpret_cast<const f32x4*>(pl + (size_t)(unsigned)o4[k] * TT_C);
    }
}

// (js, c) accumulator layout -> LIDX register layout through the [sample][36] tile
__device__ __forceinline__ void gc_transpose(float* R, const f32x4 (&acc)[4], int i, int hi, int js, int c,
                                             float* out16) {
#pragma unroll
    for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(R + (8 * n + js) * 36 + 4 * c) = acc[n];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(R + i * 36 + 4 * (hi + 2 * q));
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) out16[4 * q + ee] = v[ee];
    }
}

// Forward variants: ONE plane's table at a time (measured faster in the forward kernels, where two waves per SIMD
// overlap each other's set-up; the backward kernels -- one wave per SIMD -- are faster with all three tables first).
// T: GC_SCRATCH_FLOATS floats.
// geometry planes, forward: f[16] and, if NEED_J, J = d f / d(world xyz) (3 x 16), as gather_geo (same sums)
template <bool NEED_J>
__device__ __forceinline__ bool gather_geo_c(const float* __restrict__ planes, unsigned tex0, int H, int W, float X,
                                             float Y, float Z, bool valid, float jscale_u, float jscale_v, int lane,
                                             float* T, float (&f)[16], float (&jx)[16], float (&jy)[16],
                                             float (&jz)[16], unsigned* inb = nullptr) {
    const int i = lane & 31, hi = lane >> 5, js = lane >> 3, c = lane & 7;
    int* Toff = reinterpret_cast<int*>(T);
    float* Tw = T + 32 * 4;
    float* Ta = T + 2 * 32 * 4;
    float* Tb = T + 3 * 32 * 4;
    float* R = T + GC_PLANE_TABLE_FLOATS;
    const size_t HW = (size_t)H * W;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 af[4] = {z4, z4, z4, z4}, ax[4] = {z4, z4, z4, z4}, ay[4] = {z4, z4, z4, z4}, az[4] = {z4, z4, z4, z4};
    bool any = false;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        Corners cn;
        corners_setup(PLANE_U(p, X, Y, Z), PLANE_V(p, X, Y, Z), H, W, valid, cn);
        any = any || cn.any;
        const unsigned long long inmask = __ballot(cn.any);
        if (inb) {
            if ((threadIdx.x & 63) == 0) *inb += (unsigned)__popcll(inmask & 0xffffffffull);
        }
        if (inmask == 0) continue;  // exact: every contribution of this plane is 0 for the whole tile
        if (hi == 0) {
            const unsigned b = tex0 + (unsigned)(p * HW);
            const ti32x4 o = {(int)(b + cn.off[0]), (int)(b + cn.off[1]), (int)(b + cn.off[2]), (int)(b + cn.off[3])};
            *reinterpret_cast<ti32x4*>(Toff + i * 4) = o;
            const f32x4 w = {cn.w[0], cn.w[1], cn.w[2], cn.w[3]};
            *reinterpret_cast<f32x4*>(Tw + i * 4) = w;
        } else if (NEED_J) {
            const f32x4 a = {cn.du[0] * jscale_u, cn.du[1] * jscale_u, cn.du[2] * jscale_u, cn.du[3] * jscale_u};
            const f32x4 b = {cn.dv[0] * jscale_v, cn.dv[1] * jscale_v, cn.dv[2] * jscale_v, cn.dv[3] * jscale_v};
            *reinterpret_cast<f32x4*>(Ta + i * 4) = a;
            *reinterpret_cast<f32x4*>(Tb + i * 4) = b;
        }
        f32x4 t[4][4];
        gc_load16(planes + 4 * c, Toff, js, t);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(Tw + (8 * n + js) * 4);
            f32x4 a4 = z4, b4 = z4;
            if (NEED_J) {
                a4 = *reinterpret_cast<const f32x4*>(Ta + (8 * n + js) * 4);
                b4 = *reinterpret_cast<const f32x4*>(Tb + (8 * n + js) * 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const float tv = t[n][k][ee];
                    af[n][ee] = fmaf(w4[k], tv, af[n][ee]);
                    if (NEED_J) {
                        if (p == 0) {
                            ax[n][ee] = fmaf(a4[k], tv, ax[n][ee]);
                            ay[n][ee] = fmaf(b4[k], tv, ay[n][ee]);
                        } else if (p == 1) {
                            ax[n][ee] = fmaf(a4[k], tv, ax[n][ee]);
                            az[n][ee] = fmaf(b4[k], tv, az[n][ee]);
                        } else {
                            az[n][ee] = fmaf(a4[k], tv, az[n][ee]);
                            ay[n][ee] = fmaf(b4[k], tv, ay[n][ee]);
                        }
                    }
                }
        }
    }
    gc_transpose(R, af, i, hi, js, c, f);
    if (NEED_J) {
        gc_transpose(R, ax, i, hi, js, c, jx);
        gc_transpose(R, ay, i, hi, js, c, jy);
        gc_transpose(R, az, i, hi, js, c, jz);
    }
    return any;
}

// texture planes, forward: e[48] as gather_tex
__device__ __forceinline__ bool gather_tex_cp(const float* __restrict__ planes, unsigned tex0, int H, int W, float X,
                                              float Y, float Z, bool valid, int lane, float* T, float (&e)[48]) {
    const int i = lane & 31, hi = lane >> 5, js = lane >> 3, c = lane & 7;
    int* Toff = reinterpret_cast<int*>(T);
    float* Tw = T + 32 * 4;
    float* R = T + GC_PLANE_TABLE_FLOATS;
    const size_t HW = (size_t)H * W;
    bool any = false;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        Corners cn;
        corners_setup(PLANE_U(p, X, Y, Z), PLANE_V(p, X, Y, Z), H, W, valid, cn);
        any = any || cn.any;
        if (!__any(cn.any)) {  // exact: every weight of this plane is 0 for the whole tile
#pragma unroll
            for (int r = 0; r < 16; ++r) e[16 * p + r] = 0.f;
            continue;
        }
        if (hi == 0) {
            const unsigned b = tex0 + (unsigned)((3 + p) * HW);
            const ti32x4 o = {(int)(b + cn.off[0]), (int)(b + cn.off[1]), (int)(b + cn.off[2]), (int)(b + cn.off[3])};
            *reinterpret_cast<ti32x4*>(Toff + i * 4) = o;
        } else {
            const f32x4 w = {cn.w[0], cn.w[1], cn.w[2], cn.w[3]};
            *reinterpret_cast<f32x4*>(Tw + i * 4) = w;
        }
        f32x4 t[4][4], acc[4];
        gc_load16(planes + 4 * c, Toff, js, t);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(Tw + (8 * n + js) * 4);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) a[ee] = fmaf(w4[k], t[n][k][ee], a[ee]);
            acc[n] = a;
        }
        gc_transpose(R, acc, i, hi, js, c, &e[16 * p]);
    }
    return any;
}

// Backward variants: all three planes' tables first.
// texture planes: e[48] as gather_tex.  T: wave-private LDS scratch, GC_TABLE_FLOATS(1) + GC_TILE_FLOATS floats.
// `planes` is the base of the packed buffer and `tex0` the texel index of this lane's prompt (a tile may straddle
// prompts, and the lane that loads a texel is not the lane that owns the sample: the table holds absolute indices).
__device__ __forceinline__ bool gather_tex_c(const float* __restrict__ planes, unsigned tex0, int H, int W, float X,
                                             float Y, float Z, bool valid, int lane, float* T, float (&e)[48],
                                             unsigned* inb = nullptr) {
    const int i = lane & 31, hi = lane >> 5, js = lane >> 3, c = lane & 7;
    int* Toff = reinterpret_cast<int*>(T);
    float* Tw = T + 3 * 32 * 4;
    float* R = T + GC_TABLE_FLOATS(1);
    const size_t HW = (size_t)H * W;
    bool any = false, anyp[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        Corners cn;
        corners_setup(PLANE_U(p, X, Y, Z), PLANE_V(p, X, Y, Z), H, W, valid, cn);
        const unsigned long long inmask = __ballot(cn.any);
        if (inb) {
            if ((threadIdx.x & 63) == 0) *inb += (unsigned)__popcll(inmask & 0xffffffffull);
        }
        anyp[p] = inmask != 0;
        any = any || cn.any;
        if (hi == 0) {
            const unsigned b = tex0 + (unsigned)((3 + p) * HW);
            const ti32x4 o = {(int)(b + cn.off[0]), (int)(b + cn.off[1]), (int)(b + cn.off[2]), (int)(b + cn.off[3])};
            *reinterpret_cast<ti32x4*>(Toff + (p * 32 + i) * 4) = o;
        } else {
            const f32x4 w = {cn.w[0], cn.w[1], cn.w[2], cn.w[3]};
            *reinterpret_cast<f32x4*>(Tw + (p * 32 + i) * 4) = w;
        }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (!anyp[p]) {  // exact: every weight of this plane is 0 for the whole tile
#pragma unroll
            for (int r = 0; r < 16; ++r) e[16 * p + r] = 0.f;
            continue;
        }
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int sidx = (p * 32 + 8 * n + js) * 4;
            const ti32x4 o4 = *reinterpret_cast<const ti32x4*>(Toff + sidx);
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(Tw + sidx);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 t = *gc_addr(planes, o4[k], 16u * (unsigned)c);
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) a[ee] = fmaf(w4[k], t[ee], a[ee]);
            }
            acc[n] = a;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(R + (8 * n + js) * 36 + 4 * c) = acc[n];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(R + i * 36 + 4 * (hi + 2 * q));
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) e[16 * p + 4 * q + ee] = v[ee];
        }
    }
    return any;
}

// corner set-up of one geometry plane and the coefficient  coef_c = w_c sbar + dw_c/dx . gbar  that both the backward
// gather (u = sum coef_c texel_c) and the gradient scatter use
__device__ __forceinline__ void geo_corner_coefs(int p, int H, int W, float X, float Y, float Z, bool valid, float sbar,
                                                 float gux, float guy, float guz, float jscale_u, float jscale_v,
                                                 Corners& cn, float (&coef)[4]) {
    corners_setup(PLANE_U(p, X, Y, Z), PLANE_V(p, X, Y, Z), H, W, valid, cn);
    const float gu = (p == 2 ? guz : gux) * jscale_u, gv = (p == 1 ? guz : guy) * jscale_v;
#pragma unroll
    for (int k = 0; k < 4; ++k) coef[k] = fmaf(cn.w[k], sbar, fmaf(cn.du[k], gu, cn.dv[k] * gv));
}

// geometry planes, backward variant: the upstream (sbar, gbar) of the sample is known before the gather, so the only
// two combinations of the texels the backward needs are f (for the MLP recompute) and
//     u = sbar * f + J gbar = sum_corners coef_c * texel_c,   coef_c = w_c sbar + dw_c/dx . gbar
// (the same coefficient the gradient scatter uses): 32 registers and 2 FMAs per texel value instead of 64 / 4.
// Corner order of the sums as in gather_geo.  T: GC_TABLE_FLOATS(2) +
// GC_TILE_FLOATS floats.  Nothing of the corner set-up is kept: the scatter at the end of the tile step re-derives it
// with geo_corner_coefs (cheap VALU) instead of holding ~40 registers across the MLP chain.
__device__ __forceinline__ bool gather_geo_bwd_c(const float* __restrict__ planes, unsigned tex0, int H, int W, float X,
                                                 float Y, float Z, bool valid, float sbar, float gux, float guy,
                                                 float guz, float jscale_u, float jscale_v, int lane, float* T,
                                                 float (&f)[16], float (&u)[16], bool (&anyp)[3],
                                                 unsigned* inb = nullptr) {
    const int i = lane & 31, hi = lane >> 5, js = lane >> 3, c = lane & 7;
    int* Toff = reinterpret_cast<int*>(T);
    float* Tw = T + 3 * 32 * 4;
    float* Tc = T + 2 * 3 * 32 * 4;
    float* R = T + GC_TABLE_FLOATS(2);
    const size_t HW = (size_t)H * W;
    bool any = false;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        Corners cn;
        float coef[4];
        geo_corner_coefs(p, H, W, X, Y, Z, valid, sbar, gux, guy, guz, jscale_u, jscale_v, cn, coef);
        const unsigned long long inmask = __ballot(cn.any);
        if (inb) {
            if ((threadIdx.x & 63) == 0) *inb += (unsigned)__popcll(inmask & 0xffffffffull);
        }
        anyp[p] = inmask != 0;
        any = any || cn.any;
        if (hi == 0) {
            const unsigned b = tex0 + (unsigned)(p * HW);
            const ti32x4 o = {(int)(b + cn.off[0]), (int)(b + cn.off[1]), (int)(b + cn.off[2]), (int)(b + cn.off[3])};
            *reinterpret_cast<ti32x4*>(Toff + (p * 32 + i) * 4) = o;
            const f32x4 w = {cn.w[0], cn.w[1], cn.w[2], cn.w[3]};
            *reinterpret_cast<f32x4*>(Tw + (p * 32 + i) * 4) = w;
        } else {
            const f32x4 w = {coef[0], coef[1], coef[2], coef[3]};
            *reinterpret_cast<f32x4*>(Tc + (p * 32 + i) * 4) = w;
        }
    }
    f32x4 af[4], au[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        af[n] = z;
        au[n] = z;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (!anyp[p]) continue;  // exact: every weight of this plane is 0 for the whole tile
        f32x4 t[4][4];  // the 16 loads of a plane are issued back to back (one memory round trip, not four)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const ti32x4 o4 = *reinterpret_cast<const ti32x4*>(Toff + (p * 32 + 8 * n + js) * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t[n][k] = *gc_addr(planes, o4[k], 16u * (unsigned)c);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int sidx = (p * 32 + 8 * n + js) * 4;
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(Tw + sidx);
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(Tc + sidx);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    af[n][ee] = fmaf(w4[k], t[n][k][ee], af[n][ee]);
                    au[n][ee] = fmaf(c4[k], t[n][k][ee], au[n][ee]);
                }
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(R + (8 * n + js) * 36 + 4 * c) = af[n];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(R + i * 36 + 4 * (hi + 2 * q));
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) f[4 * q + ee] = v[ee];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(R + (8 * n + js) * 36 + 4 * c) = au[n];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(R + i * 36 + 4 * (hi + 2 * q));
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) u[4 * q + ee] = v[ee];
    }
    return any;
}

// ---- work decomposition of the per-sample kernels ----------------------------------------------------------
// A TILE is 32 samples = RB ADJACENT RAYS x SB CONSECUTIVE SAMPLE INDICES (RB * SB = 32).  SB = 1 (uniform
// sampling): an 8x4 pixel block at one sample index -- adjacent rays hit neighbouring texels at equal depth, so
// gathers hit L1 and the plane-gradient scatter of a tile touches few distinct texels.  SB = 4 (importance
// sampling: consecutive samples of a ray crowd into the same texels near the surface): a 4x2 pixel block x 4
// consecutive samples, so those repeats are summed inside the tile instead of hammering one address with
// atomics.  Without an image width the rays of a tile are RB consecutive rays.  A WORK ITEM is (ray block, chunk of
// CH sample indices); items are independent (the ray march is a separate kernel), so the grid is balanced
// whatever the image size.
// ---- work distribution -------------------------------------------------------------------------------------
// Ray blocks are dealt to the 8 XCDs round-robin in UNITS of consecutive blocks (unit u -> XCD u % 8, workgroup b
// runs on XCD b % 8): every XCD sees the same mix of cheap (outside the volume: no gather, no scatter) and expensive
// (through the object) image regions, and still works on compact groups of pixel blocks (its texels stay in its
// 4 MB L2).  An XCD's items form a queue ordered chunk-major (local item li -> chunk li / nb, local block li % nb:
// the waves running concurrently on an XCD work on neighbouring pixel blocks at the SAME depth range) or
// block-major.  Waves POP items from their XCD's queue (one int32 atomic per item on a counter zeroed on the stream
// in front of the launch, tt_queue_counters) and, once it is empty, steal from the other XCDs' queues: with static assignment the waves
// were alive only ~75 % of the kernel (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES) because item cost varies ~3x and
// items-per-wave does not divide evenly.
struct ItemQueue {
    int* ctr;          // [8] next local item per XCD
    long long unit;    // blocks per deal unit
    long long nb;      // local ray blocks per XCD (incl. the padding of a ragged last deal round)
    long long n_local; // items per XCD queue = nb * n_chunks
    int home;          // this workgroup's XCD
    int hop;           // queues already found empty (0..8)
};
__device__ __forceinline__ ItemQueue item_queue(int* ctr, long long n_blocks, int n_chunks, long long unit) {
    ItemQueue q;
    q.ctr = ctr;
    q.unit = unit;
    const long long n_units = (n_blocks + unit - 1) / unit;
    q.nb = ((n_units + 7) / 8) * unit;
    q.n_local = q.nb * n_chunks;
    q.home = blockIdx.x & 7;
    q.hop = 0;
    return q;
}
// next item for this wave: returns false when all queues are empty.  b may be >= n_blocks in the ragged last deal
// round (caller skips it).  Wave-uniform.
__device__ __forceinline__ bool item_pop(ItemQueue& q, int order, int n_chunks, long long& b, int& ck) {
    while (q.hop < 8) {
        const int xcd = (q.home + q.hop) & 7;
        int li = 0;
        if ((threadIdx.x & 63) == 0) li = atomicAdd(q.ctr + xcd, 1);
        li = __builtin_amdgcn_readfirstlane(li);
        if (li < q.n_local) {
            long long bl;
            if (order == 0) {  // chunk-major
                ck = (int)(li / q.nb);
                bl = li - (long long)ck * q.nb;
            } else {  // block-major
                bl = li / n_chunks;
                ck = (int)(li - bl * n_chunks);
            }
            const long long round = bl / q.unit;
            b = (round * 8 + xcd) * q.unit + (bl - round * q.unit);
            return true;
        }
        ++q.hop;
    }
    return false;
}

struct TileGeom {
    long long n_rays;
    int rays_per_view;
    int image_w, image_h;  // image_w == 0: linear strips of RB rays
    int sb, bw, bh;        // samples per ray per tile; pixel block bw x bh = 32 / sb rays
    int bpr, bpv;          // pixel blocks per image row / per view
    int n_samples, chunk, n_chunks;
    long long n_blocks;
    long long unit;  // blocks per XCD deal unit (item_range)
    int order;  // 0 chunk-major within an XCD, 1 block-major (TT_ORDER, tuning only)
};

// ray handled by lane j (0..31) of ray block b; the lane's sample offset inside a tile step is j % sb
__device__ __forceinline__ long long tile_ray(const TileGeom& g, long long b, int j, bool& rvalid) {
    // The returned index is ALWAYS a valid ray (clamped arithmetically, not selected on the validity mask): a lane
    // outside the image / past the last ray works on a duplicate whose contributions `rvalid` switches off.
    // Validity is built from single compares as 0/1 factors and tested ONCE (lane-mask hygiene, tt_mask.h).
    long long ray;
    float okf = 1.f;
    const int jr = j / g.sb;
    if (g.image_w > 0) {
        const long long view = b / g.bpv;
        const int rem = (int)(b - view * g.bpv);
        const int by = rem / g.bpr, bx = rem - by * g.bpr;
        const int x = bx * g.bw + (jr % g.bw), y = by * g.bh + (jr / g.bw);
        okf = tt_opaque(x < g.image_w ? 1.f : 0.f) * (y < g.image_h ? 1.f : 0.f);
        ray = view * g.rays_per_view + (long long)min(y, g.image_h - 1) * g.image_w + min(x, g.image_w - 1);
    } else {
        ray = b * (32 / g.sb) + jr;
    }
    okf = tt_opaque(okf) * (ray < g.n_rays ? 1.f : 0.f);
    rvalid = okf != 0.f;
    return ray < g.n_rays - 1 ? ray : g.n_rays - 1;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
