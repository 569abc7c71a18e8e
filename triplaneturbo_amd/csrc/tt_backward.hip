// tt_backward.hip -- backward of the fused render, split by network so each kernel's persistent
// weight-gradient accumulators fit the register file beside its working set:
//
//   k_decode_bwd_geo : consumes (d/d sdf, d/d sdf_grad) per sample from k_march_bwd (tt_march.hip), recomputes
//       the geometry decode per 32-sample tile and back-propagates through BOTH the value chain and the
//       input-gradient chain of the sdf net (the reference's second-order path: gridsample_cuda.cu:27-210 +
//       aten grid_sampler_2d_backward + transposed GEMMs), accumulates dW1/dW2 with MFMA outer products
//       (K = the 32 samples of the tile) and scatters d/d planes 0..2.
//   k_decode_bwd_tex : feature net backward, dV1/dV2/dV3, scatter of d/d planes 3..5.
// Both are purely per-sample: tiles are 32 adjacent rays at one sample index (see tt_device.h).
//
// Everything per-sample is RECOMPUTED from the planes; the forward saves only trans / weights / features.
#include "tt_device.h"
#include "tt_mfma16.h"
#include "tt_alpha.h"
#include "tt_host.h"
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// tuning build only: cycles per phase of the texture backward, summed over waves into p.phase_cycles[16]
#ifdef TT_TUNING
#define TT_PHASE(k)                                              \
    do {                                                         \
        __builtin_amdgcn_sched_barrier(0);                       \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph_acc[k] += t_now - ph_t;                               \
        ph_t = t_now;                                            \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#else
#define TT_PHASE(k) \
    do {            \
    } while (0)
#endif
#define XS 36  // row stride (floats) of the [index][sample] transposition scratch

// ---- LDS transposition helpers (wave-private scratch; DS ops of one wave execute in order) ----------
template <int N>
__device__ __forceinline__ void stage_rows(float* S, const float (&v)[N / 2], int j, int hi) {
#pragma unroll
    for (int r = 0; r < N / 2; ++r) S[LIDX(r, hi) * XS + j] = v[r];
}
// stage the N-element slice of a longer register vector that starts at register OFF
template <int N, int OFF, int TOT>
__device__ __forceinline__ void stage_rows_sub(float* S, const float (&v)[TOT], int j, int hi) {
#pragma unroll
    for (int r = 0; r < N / 2; ++r) S[LIDX(r, hi) * XS + j] = v[OFF + r];
}

// acc[m][n] += X[32m.., s] * Y[32n.., s]^T summed over the 32 samples s of the tile
template <int NX, int NY>
__device__ __forceinline__ void wgrad(f32x16 (&acc)[NX / 32][NY / 32], const float* Xs, const float* Ys, int i,
                                      int hi) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        f32x4 xa[NX / 32], yb[NY / 32];
#pragma unroll
        for (int m = 0; m < NX / 32; ++m)
            xa[m] = *reinterpret_cast<const f32x4*>(Xs + (32 * m + i) * XS + 16 * hi + 4 * t4);
#pragma unroll
        for (int n = 0; n < NY / 32; ++n)
            yb[n] = *reinterpret_cast<const f32x4*>(Ys + (32 * n + i) * XS + 16 * hi + 4 * t4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < NX / 32; ++m)
#pragma unroll
                for (int n = 0; n < NY / 32; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[m][k], yb[n][k], acc[m][n], 0, 0, 0);
    }
}

// one 32-row slice of the left operand: acc[n] += X[0..31, s] * Y[32n.., s]^T
template <int NY>
__device__ __forceinline__ void wgrad_row(f32x16 (&acc)[NY / 32], const float* Xs, const float* Ys, int i, int hi) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        f32x4 yb[NY / 32];
        const f32x4 xa = *reinterpret_cast<const f32x4*>(Xs + i * XS + 16 * hi + 4 * t4);
#pragma unroll
        for (int n = 0; n < NY / 32; ++n)
            yb[n] = *reinterpret_cast<const f32x4*>(Ys + (32 * n + i) * XS + 16 * hi + 4 * t4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int n = 0; n < NY / 32; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[k], yb[n][k], acc[n], 0, 0, 0);
    }
}

// row sum over the 32 samples of scratch row `lane` (lane <-> index 0..63)
__device__ __forceinline__ float rowsum32(const float* Xs, int lane) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        f32x4 a = *reinterpret_cast<const f32x4*>(Xs + lane * XS + 4 * g);
        s += (a[0] + a[1]) + (a[2] + a[3]);
    }
    return s;
}

// ---- weight-gradient outer products on the fp16 pipe -----------------------------------------------------------------
// acc[m][n] += X[32m.., s] Y[32n.., s]^T over the 32 samples of the tile, as 2-term split-fp16 products (tt_mfma16.h)
// with PER-LAUNCH operand scales (powers of two from rigorous magnitude bounds, wg16_scale below): nothing is ever
// rescaled inside the sample loop, so the persistent accumulators are touched by MFMAs only (per-tile or per-wave
// "sticky" scales need in-loop arithmetic on the 96-160 accumulator registers, which makes the allocator spill: measured
// in round 2).  An operand entry v is staged as ONE dword (hi | lo << 16), hi = f16(v sc), lo = f16(v sc - hi), in the
// same [index][sample] scratch as the fp32 form.  The two halves of a dword are fed to the MFMA as two ADJACENT k-slots:
// a k-step of 16 slots is 8 samples, slot 2d = hi, slot 2d + 1 = lo of the lane's d-th sample, for both operands -- so
//     mfma(A, B)          = sum_s (hi_x hi_y + lo_x lo_y)
//     mfma(A, rot16(B))   = sum_s (hi_x lo_y + lo_x hi_y)
// together the FULL product of the two split numbers: 8 MFMAs of 32 cycles per 32 x 32 tile instead of 16 fp32 MFMAs of
// 64, no de-interleaving, one v_alignbit per B dword.  Error per product term <= 2^-22 of the operands' global maxima:
// fp32-grade for a sum over all samples (the fp32 accumulator itself resolves 2^-24 of the running sum).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 2^(141 - E) for a bound with biased exponent E: maps [0, bound] into the fp16 range (bound -> [2^14, 2^15))
__device__ __forceinline__ float wg16_scale(float bound) {
    int E = (int)(__builtin_bit_cast(unsigned, bound * 1.0001f) >> 23) & 0xff;
    E = E < 16 ? 16 : (E > 240 ? 240 : E);
    return __builtin_bit_cast(float, (unsigned)(268 - E) << 23);
}
__device__ __forceinline__ unsigned wg16_pack(float x) {  // (hi | lo << 16), both round-toward-zero: hi + lo ~ x
    // (hi by masking the fp32 significand to 11 bits instead of the convert / convert-back pair: same speed, measured)
    const unsigned p = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x, 0.f));
    const float hf = (float)__builtin_bit_cast(h2_t, p).x;
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x, x - hf));
}
template <int N>
__device__ __forceinline__ void stage_rows16(float* S, const float (&v)[N / 2], int j, int hi, float sc) {
    unsigned* U = reinterpret_cast<unsigned*>(S);
#pragma unroll
    for (int r = 0; r < N / 2; ++r) U[LIDX(r, hi) * XS + j] = wg16_pack(v[r] * sc);
}
template <int N, int OFF, int TOT>
__device__ __forceinline__ void stage_rows16_sub(float* S, const float (&v)[TOT], int j, int hi, float sc) {
    unsigned* U = reinterpret_cast<unsigned*>(S);
#pragma unroll
    for (int r = 0; r < N / 2; ++r) U[LIDX(r, hi) * XS + j] = wg16_pack(v[OFF + r] * sc);
}
__device__ __forceinline__ h8_t wg16_frag(const float* S, int row, int t, int hi) {
    return __builtin_bit_cast(h8_t, *reinterpret_cast<const u32x4*>(S + row * XS + 8 * t + 4 * hi));
}
__device__ __forceinline__ h8_t wg16_swap(h8_t v) {
    u32x4 u = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int d = 0; d < 4; ++d) u[d] = __builtin_amdgcn_alignbit(u[d], u[d], 16);
    return __builtin_bit_cast(h8_t, u);
}
template <int NX, int NY>
__device__ __forceinline__ void wgrad16(f32x16 (&acc)[NX / 32][NY / 32], const float* Xs, const float* Ys, int i,
                                        int hi) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // k-step: samples 8 t .. 8 t + 7 (this half-wave: 8 t + 4 hi .. + 3)
        h8_t xa[NX / 32], yb[NY / 32], ys[NY / 32];
#pragma unroll
        for (int m = 0; m < NX / 32; ++m) xa[m] = wg16_frag(Xs, 32 * m + i, t, hi);
#pragma unroll
        for (int n = 0; n < NY / 32; ++n) {
            yb[n] = wg16_frag(Ys, 32 * n + i, t, hi);
            ys[n] = wg16_swap(yb[n]);
        }
#pragma unroll
        for (int m = 0; m < NX / 32; ++m)
#pragma unroll
            for (int n = 0; n < NY / 32; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[m], yb[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NX / 32; ++m)
#pragma unroll
            for (int n = 0; n < NY / 32; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[m], ys[n], acc[m][n], 0, 0, 0);
    }
}
// one 32-row slice of the left operand (see wgrad_row)
template <int NY>
__device__ __forceinline__ void wgrad16_row(f32x16 (&acc)[NY / 32], const float* Xs, const float* Ys, int i, int hi) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const h8_t xa = wg16_frag(Xs, i, t, hi);
        h8_t yb[NY / 32], ys[NY / 32];
#pragma unroll
        for (int n = 0; n < NY / 32; ++n) {
            yb[n] = wg16_frag(Ys, 32 * n + i, t, hi);
            ys[n] = wg16_swap(yb[n]);
        }
#pragma unroll
        for (int n = 0; n < NY / 32; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, yb[n], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NY / 32; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, ys[n], acc[n], 0, 0, 0);
    }
}

// workgroup-wide max of a per-thread value through a shared word (all threads call; v >= 0)
__device__ __forceinline__ float block_max(float v, unsigned* word) {
    __syncthreads();
    if (threadIdx.x == 0) *word = 0u;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMax(word, __builtin_bit_cast(unsigned, v));
    __syncthreads();
    const float r = __builtin_bit_cast(float, *word);
    __syncthreads();
    return r;
}

// dst += acc * ux * uy  (ux, uy: the inverse operand scales of the fp16 outer products, 1 for the fp32 ones; two
// factors so that extreme scales cannot overflow their product)
template <int NX, int NY>
__device__ __forceinline__ void flush_wgrad(const f32x16 (&acc)[NX / 32][NY / 32], float* __restrict__ dst, int i,
                                            int hi, float ux = 1.f, float uy = 1.f) {
#pragma unroll
    for (int m = 0; m < NX / 32; ++m)
#pragma unroll
        for (int n = 0; n < NY / 32; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                atomicAdd(dst + (32 * m + LIDX(r, hi)) * NY + 32 * n + i, (acc[m][n][r] * ux) * uy);
}

#define ZERO16 \
    { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }

// ---- plane-gradient scatter, combined on the matrix cores --------------------------------------------------
// fp32 global atomics are THE bottleneck of the backward on MI355X (~325 G atomic float-adds/s chip-wide,
// pattern-independent; LDS fp32 atomics are even slower: one ds_add_f32 wave-instruction per ~190 cycles per CU,
// both measured with tools/atomic_bench.hip / tools/lds_atomic_bench.hip).  A tile is 32 adjacent rays at one
// depth, so its 128 (sample, corner) references per plane touch only ~40-50 distinct texels.  Per plane the
// tile's gradient is
//        G[slot][ch] = sum_j M[slot][j] * Q[j][ch]        (64 texel slots x 32 samples x 32 channels)
// with M the sparse matrix of corner coefficients -- a GEMM, done exactly in fp32 with 32 MFMAs.  slot = 8x8
// torus hash of the texel coordinates (the 4 corners of one sample never collide, so M is filled with plain
// stores); slot ownership is claimed with one integer LDS CAS per reference, and a reference that loses its slot
// to a different texel (footprint wider than 8 texels) falls back to direct global atomics.  The MFMA C/D layout
// (lane <-> channel, register <-> slot) is exactly what a coalesced 128-byte global atomic needs, so every
// occupied slot is flushed with ONE atomic instruction per half-wave straight from the accumulator registers.
#define MS XS  // row stride of M (floats): same conflict-free stride as the transposition scratch

// M region: either the fp32 matrix (65 rows x MS floats, EXACT) or its split-fp16 image -- two planes (hi, lo) of
// 65 rows x M16_RS halves (32 samples + pad: 80-byte rows keep the 16-byte A-operand reads spread over the banks)
#define M16_RS 40
#define M16_PLANE (65 * M16_RS)
#define SCATTER_M_FLOATS 2624 /* >= max(65 * MS, 2 * M16_PLANE / 2), multiple of 64 */
template <bool EXACT>
__device__ __forceinline__ void scatter_clear(float* M, int lane) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (EXACT) {  // 64 rows of the fp32 matrix (the dump row is never read)
#pragma unroll
        for (int g = 0; g < MS / 4; ++g) *reinterpret_cast<f32x4*>(M + lane * MS + 4 * g) = z;
    } else {
#pragma unroll
        for (int g = 0; g < SCATTER_M_FLOATS / 256; ++g) *reinterpret_cast<f32x4*>(M + (g * 64 + lane) * 4) = z;
        if (lane < (SCATTER_M_FLOATS % 256) / 4)
            *reinterpret_cast<f32x4*>(M + ((SCATTER_M_FLOATS / 256) * 64 + lane) * 4) = z;
    }
}

// The three planes of one tile step, software-pipelined.  Everything except the rare lost-reference path is
// straight-line code (no per-reference branches: inactive references CAS a per-lane dummy tag and store to a dump
// row of M; empty slots are dropped by the buffer range check), so that plane p's 32 MFMAs (2048 matrix-pipe cycles,
// one wave per SIMD: nothing else would fill them) run over plane p+1's corner set-up, slot claims and M fill:
//   operands of plane p -> registers (A = M rows, B = Q columns) ; M back to zero
//   prep(p+1) ; GEMM(p) || claim(p+1) ; flush(p) from the accumulators ; tags(p) back to empty
// Tags are double-buffered (the flush of plane p reads them after plane p+1 claimed its slots).
// LDS per wave: M = 64 rows + 1 dump row (stride MS), tags = 2 x 64 slots + 32 dummies (SCATTER_TAG_INTS).
#define SCATTER_TAG_INTS 160

struct PlaneRefs {  // the two corners (2hi, 2hi+1) of this lane's sample in one plane
    float c0, c1;   // coefficient (0: no reference), normalised per sample unless EXACT
    int o0, o1;     // absolute texel index (prompt and plane included)
    int h0, h1;     // slot: 8x8 torus hash of the texel coordinates
    float qs;       // factor the sample's row of Q must be staged with (inverse of the coefficient normalisation)
};
// NORM: the sample's four coefficients are scaled by the power of two that brings the largest into [2^14, 2^15) -- the
// top of the fp16 range, as tt_mfma16.h does for every split operand -- and the sample's row of Q by its inverse:
// M Q is unchanged (exactly), column j of M and row j of Q belong to the same sample.
template <bool NORM>
__device__ __forceinline__ PlaneRefs plane_refs(const float (&coef)[4], const int (&aoff)[4], const int (&hs)[4],
                                                int hi) {
    PlaneRefs r;
    float cn = 1.f;
    r.qs = 1.f;
    if (NORM) {
        const float m = fmaxf(fmaxf(__builtin_fabsf(coef[0]), __builtin_fabsf(coef[1])),
                              fmaxf(__builtin_fabsf(coef[2]), __builtin_fabsf(coef[3])));
        int E = (int)(__builtin_bit_cast(unsigned, m) >> 23);
        E = E < 16 ? 16 : (E > 240 ? 240 : E);
        cn = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);    // 2^(141 - E)
        r.qs = __builtin_bit_cast(float, (unsigned)(E - 14) << 23);   // 1 / cn
    }
    r.c0 = (hi ? coef[2] : coef[0]) * cn;
    r.c1 = (hi ? coef[3] : coef[1]) * cn;
    r.o0 = hi ? aoff[2] : aoff[0];
    r.o1 = hi ? aoff[3] : aoff[1];
    r.h0 = hi ? hs[2] : hs[0];
    r.h1 = hi ? hs[3] : hs[1];
    return r;
}
struct ClaimState {
    bool w0, w1;  // wrote M (slot won or shared with the same texel)
    bool m0, m1;  // won the slot: this lane resets the tag
    bool l0, l1;  // lost the slot to a different texel: direct atomics
};

__device__ __forceinline__ void scatter_init_tags(int* tags, int lane) {
    tags[lane] = -1;
    tags[64 + lane] = -1;
    if (lane < 32) tags[128 + lane] = -2;  // dummies: never empty, never equal to a texel index
}

// store / clear one coefficient of M (column i = this lane's sample; row 64 = dump row)
template <bool EXACT>
__device__ __forceinline__ void m_store(float* M, int row, int i, float c) {
    if (EXACT) {
        M[row * MS + i] = c;
    } else {
        half_t h, l;
        split16(c, h, l);
        half_t* Mh = reinterpret_cast<half_t*>(M);
        Mh[row * M16_RS + i] = h;
        Mh[M16_PLANE + row * M16_RS + i] = l;
    }
}
template <bool EXACT>
__device__ __forceinline__ void m_zero(float* M, int row, int i) {
    if (EXACT) {
        M[row * MS + i] = 0.f;
    } else {
        half_t* Mh = reinterpret_cast<half_t*>(M);
        Mh[row * M16_RS + i] = (half_t)0.f;
        Mh[M16_PLANE + row * M16_RS + i] = (half_t)0.f;
    }
}

// st (tuning build): per-wave counters [0] active references, [1] lost references, [2] plane-tiles
template <bool EXACT>
__device__ __forceinline__ ClaimState scatter_claim(const PlaneRefs& r, float* M, int* tg, int* dummy, int i,
                                                    unsigned long long* st = nullptr) {
    ClaimState s;
    const bool a0 = r.c0 != 0.f, a1 = r.c1 != 0.f;
    const int old0 = atomicCAS(a0 ? tg + r.h0 : dummy, -1, r.o0);
    const int old1 = atomicCAS(a1 ? tg + r.h1 : dummy, -1, r.o1);
    s.m0 = old0 == -1;
    s.m1 = old1 == -1;
    s.w0 = tt_eq_either(old0, -1, r.o0);  // won the slot, or it already holds this texel (one compare: tt_device.h)
    s.w1 = tt_eq_either(old1, -1, r.o1);
    // not written to M.  (An inactive reference CASes the dummy tag -2, so it is "not written" too; what makes a
    // reference LOST is a non-zero coefficient on top -- scatter_lost tests the coefficient it selects with this flag,
    // instead of combining two lane masks here.)
    s.l0 = !s.w0;
    s.l1 = !s.w1;
    m_store<EXACT>(M, s.w0 ? r.h0 : 64, i, r.c0);
    m_store<EXACT>(M, s.w1 ? r.h1 : 64, i, r.c1);
#ifdef TT_TUNING
    if (st) {  // wave-uniform values, flushed once per wave with the phase timers
        st[0] += __popcll(__ballot(a0)) + __popcll(__ballot(a1));
        st[1] += __popcll(__ballot((s.l0 ? r.c0 : 0.f) != 0.f)) + __popcll(__ballot((s.l1 ? r.c1 : 0.f) != 0.f));
        st[2] += 1;
    }
#endif
    return s;
}

// references that lost their slot (tile footprint wider than the 8x8 window: sparse rays) go straight to global
// memory, one half-wave per reference (lanes <-> channels: a coalesced 128-byte atomic each)
__device__ __forceinline__ void scatter_lost(const PlaneRefs& r, const ClaimState& s, const float* Qs, float* Ls,
                                             __amdgpu_buffer_rsrc_t grsrc, int i, int hi) {
    const float lc0 = s.l0 ? r.c0 : 0.f, lc1 = s.l1 ? r.c1 : 0.f;  // coefficient of a lost corner, else 0
    const unsigned long long bal = __ballot(__builtin_fabsf(lc0) + __builtin_fabsf(lc1) != 0.f);
    if (bal == 0) return;
    float* Lc = Ls;                                 // [sample][4] coefficient of a lost corner, else 0
    int* Lo = reinterpret_cast<int*>(Ls + 32 * 4);  // [sample][4] absolute texel index
    Lc[4 * i + 2 * hi] = lc0;
    Lc[4 * i + 2 * hi + 1] = lc1;
    Lo[4 * i + 2 * hi] = r.o0;
    Lo[4 * i + 2 * hi + 1] = r.o1;
    // walk only the samples that lost something, two per step (one per half-wave)
    unsigned todo = (unsigned)(bal & 0xffffffffull) | (unsigned)(bal >> 32);
    while (todo) {
        const int s0 = __builtin_ctz(todo);
        todo &= todo - 1;
        int s1 = -1;
        if (todo) {
            s1 = __builtin_ctz(todo);
            todo &= todo - 1;
        }
        const int sidx2 = hi ? s1 : s0;
        if (sidx2 >= 0) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(Lc + 4 * sidx2);
            const i32x4 o4 = *reinterpret_cast<const i32x4*>(Lo + 4 * sidx2);
            const float v = Qs[sidx2 * 33 + i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned tex = c4[q] != 0.f ? (unsigned)o4[q] : ~0u;  // ~0: beyond num_records, dropped
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v * c4[q], grsrc, (int)((tex << 7) | (4u * (unsigned)i)),
                                                                0, 0);
            }
        }
    }
}

// prep(pl, refs): corner set-up of plane pl for this lane's sample (and, where Q differs per plane, its staging into
// Qs[j*33 + ch] -- the previous plane's B operand is in registers by then).  M: all-zero on entry and on exit.
// G = M Q for the two 32-slot tiles is either 32 fp32 MFMAs (EXACT, and the texture kernel -- see there; 2048
// matrix-pipe cycles) or, in the geometry kernel (3.51 -> 3.33 ms), the split-fp16 scheme of tt_mfma16.h: M is already a
// (hi, lo) fp16 image normalised per sample (plane_refs<true>, m_store), the B operand (16 samples of this lane's
// channel per half-wave) is normalised per channel and split, and 12 fp16 MFMAs (384 cycles) do the work.
// prep(pl, refs): corner set-up of plane pl for this lane's sample AND the staging of its row of Q, scaled by refs.qs,
// into Qs[j*33 + ch] (the previous plane's B operand is in registers by then).  M: all-zero on entry and on exit.
template <bool EXACT, class Prep>
__device__ __forceinline__ void scatter_planes(float* __restrict__ grad, unsigned grad_bytes, const float* Qs, float* M,
                                               int* tags, float* Ls, int i, int hi, Prep&& prep,
                                               unsigned long long* st = nullptr) {
    // BUFFER atomics with a 32-bit BYTE offset (texel << 7 | channel * 4) from the gradient copy: an empty slot's tag
    // is -1, its offset 0xFFFFFF80 + 4 ch lies beyond num_records (the host refuses gradient buffers of 4 GB - 256 B
    // and more) and the hardware range check drops the atomic -- no compare, no exec-mask branch per slot (the
    // predicated global atomics this replaced cost ~100 cycles per slot pair, 12 % of the kernel).
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(grad, 0, (int)grad_bytes, 0x00020000);
    const unsigned lane_b = 4u * (unsigned)i;
    int* const dummy = tags + 128 + i;
    PlaneRefs rc, rn;
    ClaimState sc, sn;
    prep(0, rc);
    sc = scatter_claim<EXACT>(rc, M, tags, dummy, i, st);
    scatter_lost(rc, sc, Qs, Ls, grsrc, i, hi);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        int* const tg = tags + 64 * (pl & 1);
        // ---- G = M Q; once its operands are in registers: M back to all-zero, next plane's set-up and Q row; the
        // next plane's claims fill the matrix-pipe time ----
        f32x16 acc0 = ZERO16, acc1 = ZERO16;
        if (EXACT) {
            f32x4 a4[2][4];
            float bq[16];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
                    a4[m][t4] = *reinterpret_cast<const f32x4*>(M + (32 * m + i) * MS + 16 * hi + 4 * t4);
#pragma unroll
            for (int t = 0; t < 16; ++t) bq[t] = Qs[(t + 16 * hi) * 33 + i];
            M[(sc.w0 ? rc.h0 : 64) * MS + i] = 0.f;
            M[(sc.w1 ? rc.h1 : 64) * MS + i] = 0.f;
            if (pl < 2) prep(pl + 1, rn);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0][t >> 2][t & 3], bq[t], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1][t >> 2][t & 3], bq[t], acc1, 0, 0, 0);
            }
        } else {
            const half_t* Mh = reinterpret_cast<const half_t*>(M);
            h8_t ah[2][2], al[2][2];  // [slot tile][k-step]: 8 samples 16 ks + 8 hi .. + 7 of slot row 32 m + i
            float bs[2][8];           // the same samples of channel i
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const half_t* a = Mh + (32 * m + i) * M16_RS + 16 * ks + 8 * hi;
                    ah[m][ks] = *reinterpret_cast<const h8_t*>(a);
                    al[m][ks] = *reinterpret_cast<const h8_t*>(a + M16_PLANE);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[ks][j] = Qs[(16 * ks + 8 * hi + j) * 33 + i];
            m_zero<false>(M, sc.w0 ? rc.h0 : 64, i);
            m_zero<false>(M, sc.w1 ? rc.h1 : 64, i);
            if (pl < 2) prep(pl + 1, rn);
            // per-channel normalisation of the B operand to the top of the fp16 range (column = this lane and lane ^ 32)
            float mx = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, __builtin_fabsf(bs[ks][j]));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            int E = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
            E = E < 16 ? 16 : (E > 240 ? 240 : E);
            const float bsc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);
            const float bun = __builtin_bit_cast(float, (unsigned)(E - 14) << 23);
            h8_t bh[2], bl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = bs[ks][2 * j] * bsc, x1 = bs[ks][2 * j + 1] * bsc;
                    const h2_t ph = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(x0, x1));
                    const h2_t pq =
                        __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(x0 - (float)ph.x, x1 - (float)ph.y));
                    bh[ks][2 * j] = ph.x;
                    bh[ks][2 * j + 1] = ph.y;
                    bl[ks][2 * j] = pq.x;
                    bl[ks][2 * j + 1] = pq.y;
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][ks], bh[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][ks], bh[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][ks], bl[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][ks], bl[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0][ks], bh[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1][ks], bh[ks], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc0[r] *= bun;
                acc1[r] *= bun;
            }
        }
        if (pl < 2) sn = scatter_claim<EXACT>(rn, M, tags + 64 * ((pl + 1) & 1), dummy, i, st);
        // ---- flush: one 128-byte atomic per slot pair, straight from the accumulators (slot of reg 4g+e = LIDX) ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const i32x4 k0 = *reinterpret_cast<const i32x4*>(tg + 8 * g + 4 * hi);
            const i32x4 k1 = *reinterpret_cast<const i32x4*>(tg + 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc0[4 * g + e2], grsrc,
                                                                (int)(((unsigned)k0[e2] << 7) | lane_b), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc1[4 * g + e2], grsrc,
                                                                (int)(((unsigned)k1[e2] << 7) | lane_b), 0, 0);
            }
        }
        // ---- tags of this plane back to empty ----
        *(sc.m0 ? tg + rc.h0 : dummy) = sc.m0 ? -1 : -2;
        *(sc.m1 ? tg + rc.h1 : dummy) = sc.m1 ? -1 : -2;
        if (pl < 2) {
            scatter_lost(rn, sn, Qs, Ls, grsrc, i, hi);
            rc = rn;
            sc = sn;
        }
    }
}

struct MlpGradPtrs {
    float* w1;
    float* w2;
    float* w3;
    float* v1;
    float* v2;
    float* v3;
};

// =====================================================================================================
// geometry half
// =====================================================================================================
struct BwdGeoParams {
    const float* packed;
    MlpPtrs w;
    const float* rays_o;
    const float* rays_d;
    const float* t_starts;
    const float* t_ends;
    tt_render_cfg cfg;
    TileGeom geom;
    long long n_items;
    int* queue;  // per-XCD item counters (tt_queue_counters)
    const float* ws;  // (n_rays*S, 4): d/d sdf, d/d sdf_grad xyz  (from k_march_bwd)
    int n_copies;     // privatised copies of grad_packed
    float* grad_packed;
    MlpGradPtrs grads;
    unsigned long long* phase_cycles;  // tuning build only (TT_PHASE), else null
};

#define GEO_SCRATCH_FLOATS (2 * 64 * XS)
// split-fp16 weight images (tt_mfma16.h): W1, W2 at their fp32 offsets (same bytes), transposes appended
#define GOFF_W1T LDS_GEO_FLOATS
#define GOFF_W2T (GOFF_W1T + IMG16_FLOATS(32, 64))
#define LDS_GEO16_FLOATS (GOFF_W2T + IMG16_FLOATS(64, 64))

template <bool EXACT, bool WG16>
__global__ __launch_bounds__(256, 1) void k_decode_bwd_geo(BwdGeoParams p) {
    __shared__ __attribute__((aligned(16))) float L[LDS_GEO16_FLOATS + 4 * (GEO_SCRATCH_FLOATS + SCATTER_TAG_INTS)];
    {
        MlpPtrs w = p.w;
        stage_weights<EXACT, 64, 32>(L + OFF_W1, w.w1);
        stage_weights<EXACT, 64, 64>(L + OFF_W2, w.w2);
        lds_load_matrix(L + OFF_W3, w.w3, 1, 64, 64);
        stage_weights_t<EXACT, 64, 32>(L + GOFF_W1T, w.w1);
        stage_weights_t<EXACT, 64, 64>(L + GOFF_W2T, w.w2);
    }
    const tt_render_cfg& cfg = p.cfg;
    // ---- per-launch operand scales of the fp16 outer products dW1 += a1 u^T, dW2 += a2 v^T (wgrad16 above) ----
    // rigorous magnitude bounds from the weights and the launch's maxima (planes, upstream: reduced on the stream in front
    // of this kernel into the queue slot, tt_host.h):   |a2| <= max |w3|,   |a1_j| <= sum_i |W2[i][j]| |w3_i|,
    //   |f| <= 3 P,  |h1| <= max_i ||W1_i||_1 3 P,  |u| = |sum_corners coef texel| <= 3 P (Sb + 2 (ju + jv) Gb)  (the four
    //   bilinear weights of a plane sum to <= 1, their derivatives to <= 2 per axis),  |qbar| = |u - sbar f| <= 3 P 2 (ju +
    //   jv) Gb,  |b1bar| <= max_i ||W1_i||_1 |qbar|,  |v| = |sbar h1 + b1bar|.
    float sA1 = 1.f, sU = 1.f, sA2 = 1.f, sV = 1.f;
    if (WG16) {
        const unsigned* bnd = reinterpret_cast<const unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        const float Pm = __builtin_bit_cast(float, bnd[TT_BOUND_PLANES]), Sb = __builtin_bit_cast(float, bnd[TT_BOUND_UP0]),
                    Gb = __builtin_bit_cast(float, bnd[TT_BOUND_UP1]);
        unsigned* word = reinterpret_cast<unsigned*>(L + LDS_GEO16_FLOATS);  // scratch is free until the main loop
        const int t = threadIdx.x;
        float w1row = 0.f, a1col = 0.f, w3abs = 0.f;
        if (t < 64) {
            for (int c = 0; c < 32; ++c) w1row += __builtin_fabsf(p.w.w1[t * 32 + c]);
            for (int r = 0; r < 64; ++r) a1col += __builtin_fabsf(p.w.w2[r * 64 + t]) * __builtin_fabsf(p.w.w3[r]);
            w3abs = __builtin_fabsf(p.w.w3[t]);
        }
        const float W1max = block_max(w1row, word), A1max = block_max(a1col, word), A2max = block_max(w3abs, word);
        const float jsum = (0.5f * cfg.plane_w + 0.5f * cfg.plane_h) / cfg.radius;
        const float Fmax = 3.f * Pm, H1max = W1max * Fmax;
        const float Umax = Fmax * (Sb + 2.f * jsum * Gb), QBmax = Fmax * 2.f * jsum * Gb;
        const float Vmax = Sb * H1max + W1max * QBmax;
        sA1 = wg16_scale(A1max);
        sU = wg16_scale(Umax);
        sA2 = wg16_scale(A2max);
        sV = wg16_scale(Vmax);
    }
    const TileGeom& tg = p.geom;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5, wave_in_blk = threadIdx.x >> 6;
    float* Xs = L + LDS_GEO16_FLOATS + wave_in_blk * (GEO_SCRATCH_FLOATS + SCATTER_TAG_INTS);
    float* Ys = Xs + 64 * XS;
    int* tags = reinterpret_cast<int*>(Ys + 64 * XS);
    scatter_init_tags(tags, lane);
    __syncthreads();
    const int S = cfg.n_samples;
    ItemQueue iq = item_queue(p.queue, tg.n_blocks, tg.n_chunks, tg.unit);
    const int H = cfg.plane_h, W = cfg.plane_w;
    const size_t HW = (size_t)H * W;
    const size_t plane_stride = 6 * HW * TT_C;
    const float ju = 0.5f * W / cfg.radius, jv = 0.5f * H / cfg.radius;
    // privatised gradient planes: this workgroup scatters into copy (blockIdx % n_copies); the copies are summed by
    // tt_planes_unpack_grad.  Spreads same-texel atomics (which serialise at the memory side) over n_copies addresses.
    float* const grad_out =
        p.grad_packed + (size_t)(blockIdx.x % (unsigned)p.n_copies) * cfg.n_prompts * plane_stride;
    const unsigned grad_bytes = (unsigned)(cfg.n_prompts * plane_stride * sizeof(float));  // one copy, < 4 GB - 256

    f32x16 accW1[2][1] = {{ZERO16}, {ZERO16}};
    f32x16 accW2[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};
    float accw3 = 0.f;
#ifdef TT_TUNING
    unsigned long long ph_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif

#pragma nounroll
    for (;;) {
        long long b;
        int ck;
        if (!item_pop(iq, tg.order, tg.n_chunks, b, ck)) break;
        if (b >= tg.n_blocks) continue;  // padding of the ragged last deal round
        bool ray_ok;
        const long long ray = tile_ray(tg, b, i, ray_ok);
        const float ray_okf = tt_opaque(ray_ok ? 1.f : 0.f);  // 0/1 factor the compiler cannot fold back into a mask
        const int ks = i % tg.sb;  // this lane's sample offset inside a tile step
        const int view = (int)(ray / cfg.rays_per_view);
        const size_t pofs = (size_t)(view / cfg.views_per_prompt) * plane_stride;
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        // rays_d == null: explicit points (tt_points_bwd_*), x = rays_o exactly
        const float dx = p.rays_d ? p.rays_d[ray * 3 + 0] : 0.f, dy = p.rays_d ? p.rays_d[ray * 3 + 1] : 0.f,
                    dz = p.rays_d ? p.rays_d[ray * 3 + 2] : 0.f;
        const int s_end = (ck + 1) * tg.chunk < S ? (ck + 1) * tg.chunk : S;
        // per-step inputs are prefetched one tile step ahead (see k_decode_bwd_tex); a step past the chunk reads a
        // clamped, valid address
        struct StepIn {
            f32x4 up;  // upstream (from the march backward): d/d sdf and d/d sdf_grad of this sample
            float ts, te;
        };
        auto load_step = [&](int sb0) {
            StepIn r;
            const int si = sb0 + ks;
            const long long sidx = ray * S + (si < S ? si : S - 1);
            r.up = *reinterpret_cast<const f32x4*>(p.ws + sidx * 4);
            r.ts = p.rays_d ? p.t_starts[sidx] : 0.f;
            r.te = p.rays_d ? p.t_ends[sidx] : 0.f;
            return r;
        };
        StepIn in = load_step(ck * tg.chunk), in_next;
#pragma nounroll
        for (int sb0 = ck * tg.chunk; sb0 < s_end; sb0 += tg.sb, in = in_next) {
            in_next = load_step(sb0 + tg.sb);
            const int si = sb0 + ks;
            const bool rvalid = ray_ok && si < s_end;
            // validity as a 0/1 FACTOR built from single compares (x * 1 = x, finite * 0 = 0): no select on a lane mask
            // that the scalar ALU has just combined (see corners_setup in tt_device.h)
            const float vf = ray_okf * (si < s_end ? 1.f : 0.f);
            const float sbar = in.up[0] * vf, gbx = in.up[1] * vf, gby = in.up[2] * vf, gbz = in.up[3] * vf;
            TT_PHASE(0);
            // exact with skip_eps_geo = 0 (the default); > 0: the opt-in approximation of tt_abi.h.  (!(x <= eps): a NaN
            // upstream is never skipped)
            if (!__any(!((__builtin_fabsf(sbar) + __builtin_fabsf(gbx)) + (__builtin_fabsf(gby) + __builtin_fabsf(gbz)) <=
                         cfg.skip_eps_geo)))
                continue;
            float tm, px, py, pz;
            sample_position(ox, oy, oz, dx, dy, dz, in.ts, in.te, tm, px, py, pz);
            const float X = scale_coord(px, cfg.radius), Y = scale_coord(py, cfg.radius),
                        Z = scale_coord(pz, cfg.radius);
            // ---- recompute the geometry decode ----
            float f[16], u[16];  // u = sbar f + J gbar
            bool anyp[3];
            const bool any = __any(gather_geo_bwd_c(p.packed, (unsigned)(pofs / TT_C), H, W, X, Y, Z, rvalid, sbar, gbx,
                                                    gby, gbz, ju, jv, lane, Xs, f, u, anyp));
            TT_PHASE(1);
            if (!any) continue;  // exact: no in-bounds texel => f = J = 0, every mask false
            float h1[32], h2[32], a2[32], a1[32], q[16];
            mvx<EXACT, 64, 32>(L + OFF_W1, f, h1, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
            mvx<EXACT, 64, 64>(L + OFF_W2, h1, h2, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h2[r] = fmaxf(h2[r], 0.f);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f32x4 w3 = *reinterpret_cast<const f32x4*>(L + OFF_W3 + 8 * g + 4 * hi);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) a2[4 * g + e2] = h2[4 * g + e2] > 0.f ? w3[e2] : 0.f;
            }
            mvtx<EXACT, 64, 64>(L + GOFF_W2T, L + OFF_W2, a2, a1, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) a1[r] = h1[r] > 0.f ? a1[r] : 0.f;
            mvtx<EXACT, 32, 64>(L + GOFF_W1T, L + OFF_W1, a1, q, i, hi);
            TT_PHASE(3);
            // ---- network + plane gradients ----
            {
                float qb[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) qb[r] = fmaf(-sbar, f[r], u[r]);  // qbar = J gbar = u - sbar f
                // (no opaque scheduling region here, unlike the texture kernel: with the outer products on the fp16 pipe
                // this kernel has register slack and one scheduling region is faster: 3.22 -> 3.14 ms)
                const bool region = WG16 ? true : cfg.flags >= 0;  // (always true; opaque to the compiler unless WG16)
#ifdef TT_X_GEO_REGION_SCATTER
                const bool region_s = cfg.flags >= 0;
#else
                const bool region_s = region;
#endif
                const bool do_wgrad = region && !TT_DBG(cfg.flags, TT_DBG_NO_WGRAD);
                // dW1 += a1 (sbar f + qbar)^T
                if (do_wgrad) {
                    if (WG16) {
                        stage_rows16<64>(Xs, a1, i, hi, sA1);
                        stage_rows16<32>(Ys, u, i, hi, sU);
                        wgrad16<64, 32>(accW1, Xs, Ys, i, hi);
                    } else {
                        stage_rows<64>(Xs, a1, i, hi);
                        stage_rows<32>(Ys, u, i, hi);
                        wgrad<64, 32>(accW1, Xs, Ys, i, hi);
                    }
                }
                TT_PHASE(7);
                // a1bar = W1 qbar ; b1bar = m1 . a1bar ; v = sbar h1 + b1bar
                float t1[32];
                mvx<EXACT, 64, 32>(L + OFF_W1, qb, t1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) t1[r] = h1[r] > 0.f ? t1[r] : 0.f;  // b1bar
                float v[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) v[r] = fmaf(sbar, h1[r], t1[r]);
                TT_PHASE(4);
                // dW2 += a2 v^T
                if (do_wgrad) {
                    if (WG16) {
                        stage_rows16<64>(Xs, a2, i, hi, sA2);
                        stage_rows16<64>(Ys, v, i, hi, sV);
                        wgrad16<64, 64>(accW2, Xs, Ys, i, hi);
                    } else {
                        stage_rows<64>(Xs, a2, i, hi);
                        stage_rows<64>(Ys, v, i, hi);
                        wgrad<64, 64>(accW2, Xs, Ys, i, hi);
                    }
                }
                TT_PHASE(8);
                // a2bar = W2 b1bar ; dw3 += sbar h2 + m2 . a2bar
                float t2[32];
                mvx<EXACT, 64, 64>(L + OFF_W2, t1, t2, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) t2[r] = fmaf(sbar, h2[r], h2[r] > 0.f ? t2[r] : 0.f);
                stage_rows<64>(Xs, t2, i, hi);
                accw3 += rowsum32(Xs, lane);
                TT_PHASE(5);
                // ---- scatter d/d geometry planes: texel(p,c)[ch] += q[ch] * coef(p,c) ----
                if (region_s && !TT_DBG(cfg.flags, TT_DBG_NO_SCATTER)) {
                    scatter_clear<EXACT>(Xs, lane);  // M = 0 (Xs held wgrad staging; SCATTER_M_FLOATS reach into Ys)
                    float* Qs = Xs + SCATTER_M_FLOATS;  // the sample's row of Q: q scaled per plane, stride 33
                    TT_PHASE(9);
                    const int tex0 = (int)(pofs / TT_C);
                    scatter_planes<EXACT>(grad_out, grad_bytes, Qs, Xs, tags, Qs + 32 * 33, i, hi,
                                          [&](int pl, PlaneRefs& refs) {
                        Corners c;
                        float coef[4];  // per corner: w sbar + dw/dx . gbar -- gather AND scatter coefficient
                        geo_corner_coefs(pl, H, W, X, Y, Z, rvalid, sbar, gbx, gby, gbz, ju, jv, c, coef);
                        int aoff[4];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) aoff[q4] = tex0 + (int)(pl * HW) + c.off[q4];
                        refs = plane_refs<!EXACT>(coef, aoff, c.hs, hi);
#pragma unroll
                        for (int r = 0; r < 16; ++r) Qs[i * 33 + LIDX(r, hi)] = q[r] * refs.qs;
                    }
#ifdef TT_TUNING
                    , ph_acc + 14
#endif
                    );
                    TT_PHASE(10);
                }
            }
        }
    }
#ifdef TT_TUNING
    TT_PHASE(11);
    if (p.phase_cycles && lane == 0)
        for (int k = 0; k < 20; ++k) atomicAdd(p.phase_cycles + 20 + k, ph_acc[k]);
#endif
    // ---- flush the persistent weight-gradient accumulators ----
    flush_wgrad<64, 32>(accW1, p.grads.w1, i, hi, 1.f / sA1, 1.f / sU);
    flush_wgrad<64, 64>(accW2, p.grads.w2, i, hi, 1.f / sA2, 1.f / sV);
    atomicAdd(p.grads.w3 + lane, accw3);
}

// =====================================================================================================
// texture half
// =====================================================================================================
struct BwdTexParams {
    const float* packed;
    MlpPtrs w;
    const float* rays_o;
    const float* rays_d;
    const float* t_starts;
    const float* t_ends;
    tt_render_cfg cfg;
    const float* weights;
    const float* features;
    const float* g_rgb;
    const float* g_features;
    TileGeom geom;
    long long n_items;
    int* queue;  // per-XCD item counters (tt_queue_counters)
    int n_copies;
    float* grad_packed;
    MlpGradPtrs grads;
    unsigned long long* phase_cycles;  // tuning build only (TT_PHASE), else null
};

#define TEX_W_FLOATS (LDS_W_FLOATS - OFF_V1)
#define TV1 0
#define TV2 (OFF_V2 - OFF_V1)
#define TV3 (OFF_V3 - OFF_V1)
// V1, V2 and their transposes as split-fp16 images (tt_mfma16.h): every mat-vec product of the kernel runs on the fp16
// pipe.  To make room for the V2^T image the per-wave scratch is 128 rows (was 160): the parked e (96 rows) shares it
// with a 32-row window through which k2 (for dV3) and k1bar (for dV1) are transposed in two halves.
#define TV1T TEX_W_FLOATS
#define TV2T (TV1T + IMG16_FLOATS(96, 64))
#define TEX_W16_FLOATS (TV2T + IMG16_FLOATS(64, 64))
#define TEX_SCRATCH_FLOATS (128 * XS)

template <bool EXACT, bool WG16>
__global__ __launch_bounds__(256, 1) void k_decode_bwd_tex(BwdTexParams p) {
    __shared__ __attribute__((aligned(16))) float Lt[TEX_W16_FLOATS + 4 * (TEX_SCRATCH_FLOATS + SCATTER_TAG_INTS)];
    {
        MlpPtrs w = p.w;
        stage_weights<EXACT, 64, 96>(Lt + TV1, w.v1);
        stage_weights<EXACT, 64, 64>(Lt + TV2, w.v2);
        lds_load_matrix(Lt + TV3, w.v3, 3, 64, 64);
        stage_weights_t<EXACT, 64, 96>(Lt + TV1T, w.v1);
        stage_weights_t<EXACT, 64, 64>(Lt + TV2T, w.v2);
    }
    const tt_render_cfg& cfg = p.cfg;
    // ---- per-launch operand scales of the fp16 outer products dV1 += k1bar e^T, dV2 += k2bar k1^T (wgrad16) ----
    // bounds:  |e| <= P (bilinear weights are a convex combination),  |k1_i| <= ||V1_i||_1 P,
    //   |cbar| <= |shrink| 1.002 / 4 Gr + Gf  (weights <= 1, sigmoid' <= 1/4; Gr / Gf = max |g_rgb| / |g_features|),
    //   |k2bar_i| <= sum_o |V3[o][i]| |cbar|,   |k1bar_j| <= sum_i |V2[i][j]| (bound of k2bar_i).
    float sKB1 = 1.f, sE = 1.f, sK2B = 1.f, sK1 = 1.f;
    if (WG16) {
        const unsigned* bnd = reinterpret_cast<const unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        const float Pm = __builtin_bit_cast(float, bnd[TT_BOUND_PLANES]), Gr = __builtin_bit_cast(float, bnd[TT_BOUND_UP0]),
                    Gf = __builtin_bit_cast(float, bnd[TT_BOUND_UP1]);
        unsigned* word = reinterpret_cast<unsigned*>(Lt + TEX_W16_FLOATS);  // scratch is free until the main loop
        const int t = threadIdx.x;
        const float CBmax = __builtin_fabsf(cfg.rgb_grad_shrink) * (1.002f * 0.25f) * Gr + Gf;
        float v1row = 0.f, k2b = 0.f, kb1 = 0.f;
        if (t < 64) {
            for (int c = 0; c < 96; ++c) v1row += __builtin_fabsf(p.w.v1[t * 96 + c]);
            for (int o = 0; o < 3; ++o) k2b += __builtin_fabsf(p.w.v3[o * 64 + t]);
            for (int r = 0; r < 64; ++r) {
                float c3 = 0.f;
                for (int o = 0; o < 3; ++o) c3 += __builtin_fabsf(p.w.v3[o * 64 + r]);
                kb1 += __builtin_fabsf(p.w.v2[r * 64 + t]) * c3;
            }
        }
        const float V1max = block_max(v1row, word), K2Bw = block_max(k2b, word), KB1w = block_max(kb1, word);
        sE = wg16_scale(Pm);
        sK1 = wg16_scale(V1max * Pm);
        sK2B = wg16_scale(K2Bw * CBmax);
        sKB1 = wg16_scale(KB1w * CBmax);
    }
    const TileGeom& tg = p.geom;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5, wave_in_blk = threadIdx.x >> 6;
    // per-wave scratch: rows 0..31 = Xs (transposition window / first half of bigger operands), rows 32..127 = Ys
    float* Xs = Lt + TEX_W16_FLOATS + wave_in_blk * (TEX_SCRATCH_FLOATS + SCATTER_TAG_INTS);
    float* Ys = Xs + 32 * XS;  // 96 rows: the parked e
    int* tags = reinterpret_cast<int*>(Xs + 128 * XS);
    // cbar of the tile, [3][32]: in the 4 pad columns of Xs rows 0..23 (row r holds floats 4r..4r+3 of the 96) --
    // stage_rows / the scatter matrix only touch columns 0..31 of a row
    float* Cb = Xs + 32;
#define CB_AT(idx) Cb[((idx) >> 2) * XS + ((idx)&3)]
    scatter_init_tags(tags, lane);
    __syncthreads();
    const int S = cfg.n_samples;
    const int H = cfg.plane_h, W = cfg.plane_w;
    const size_t HW = (size_t)H * W;
    const size_t plane_stride = 6 * HW * TT_C;
    ItemQueue iq = item_queue(p.queue, tg.n_blocks, tg.n_chunks, tg.unit);
    const float shrink = cfg.rgb_grad_shrink;
    float* const grad_out =  // private copy of the gradient planes of this workgroup (see k_decode_bwd_geo)
        p.grad_packed + (size_t)(blockIdx.x % (unsigned)p.n_copies) * cfg.n_prompts * plane_stride;
    const unsigned grad_bytes = (unsigned)(cfg.n_prompts * plane_stride * sizeof(float));  // one copy, < 4 GB - 256

    f32x16 accV1a[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};  // dV1[:, 0:64]
    f32x16 accV1b[2][1] = {{ZERO16}, {ZERO16}};                  // dV1[:, 64:96]
    f32x16 accV2[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};
    float accV3[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // [half of the 64 indices][output]; this lane: 16 samples
#ifdef TT_TUNING
    unsigned long long ph_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long scat_st[3] = {0, 0, 0};
    unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif

#pragma nounroll
    for (;;) {
      long long b;
      int ck;
      TT_PHASE(17);
      if (!item_pop(iq, tg.order, tg.n_chunks, b, ck)) break;
      TT_PHASE(18);
        if (b >= tg.n_blocks) continue;  // padding of the ragged last deal round
      bool ray_ok;
      const long long ray = tile_ray(tg, b, i, ray_ok);
      const float ray_okf = tt_opaque(ray_ok ? 1.f : 0.f);  // 0/1 factor the compiler cannot fold back into a mask
      const int ks = i % tg.sb;  // this lane's sample offset inside a tile step
      const int view = (int)(ray / cfg.rays_per_view);
      const size_t pofs = (size_t)(view / cfg.views_per_prompt) * plane_stride;
      const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
      const float dx = p.rays_d ? p.rays_d[ray * 3 + 0] : 0.f, dy = p.rays_d ? p.rays_d[ray * 3 + 1] : 0.f,
                  dz = p.rays_d ? p.rays_d[ray * 3 + 2] : 0.f;
      float grgb[3];
#pragma unroll
      for (int o = 0; o < 3; ++o) grgb[o] = p.g_rgb ? p.g_rgb[ray * 3 + o] : 0.f;
      const int s_end = (ck + 1) * tg.chunk < S ? (ck + 1) * tg.chunk : S;
      TT_PHASE(19);
      // Per-step inputs (weight, features, interval, upstream) are PREFETCHED one tile step ahead: their loads are
      // issued at the top of the previous step and have landed long before they are needed (they used to cost two
      // exposed memory round trips per step, 8 % of the kernel; the old "weights first" early-out bought nothing on a
      // scene where 95 % of the tile steps are live).  A step past the chunk reads a clamped, valid address.
      struct StepIn {
          float wgt, f[3], gf[3], ts, te;
      };
      auto load_step = [&](int sb0) {
          StepIn r;
          const int si = sb0 + ks;
          const long long sidx = ray * S + (si < S ? si : S - 1);
          r.wgt = p.weights ? p.weights[sidx] : 0.f;  // null: no march above (points)
#pragma unroll
          for (int o = 0; o < 3; ++o) {
              r.f[o] = p.weights ? p.features[sidx * 3 + o] : 0.f;  // (features only enter through the weights)
              r.gf[o] = p.g_features ? p.g_features[sidx * 3 + o] : 0.f;
          }
          r.ts = p.rays_d ? p.t_starts[sidx] : 0.f;
          r.te = p.rays_d ? p.t_ends[sidx] : 0.f;
          return r;
      };
      StepIn in = load_step(ck * tg.chunk), in_next;
#pragma nounroll
      for (int sb0 = ck * tg.chunk; sb0 < s_end; sb0 += tg.sb, in = in_next) {
        in_next = load_step(sb0 + tg.sb);
        const int si = sb0 + ks;
        const bool valid = ray_ok && si < s_end;
        const float vf = ray_okf * (si < s_end ? 1.f : 0.f);
        // ---- upstream: cbar_o = shrink * w_i * g_rgb[ray,o] * 1.002 * s(1-s) + g_features ----
        float cb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float s = sigmoid_(in.f[o]);
            const float c = shrink * in.wgt * grgb[o] * 1.002f * s * (1.f - s) + in.gf[o];
            cb[o] = c * vf;  // 0/1 factor, not a select on a freshly combined lane mask (see k_decode_bwd_geo)
        }
        TT_PHASE(0);
        // exact with skip_eps_tex = 0 (the default: nothing flows back); > 0: the opt-in approximation of tt_abi.h
        if (!__any(!((__builtin_fabsf(cb[0]) + __builtin_fabsf(cb[1])) + __builtin_fabsf(cb[2]) <= cfg.skip_eps_tex)))
            continue;
#ifdef TT_TUNING
        {  // live-lane statistics (tools/phase_cycles.py): slots 12 / 13 are unused by the timers
            const unsigned long long live = __ballot(cb[0] != 0.f || cb[1] != 0.f || cb[2] != 0.f) & 0xffffffffull;
            const unsigned long long big = __ballot(fabsf(cb[0]) + fabsf(cb[1]) + fabsf(cb[2]) > 1e-12f) & 0xffffffffull;
            ph_acc[12] += 1;
            ph_acc[13] += __popcll(live);
            ph_acc[14] += __popcll(big);
        }
#endif
        float tm, px, py, pz;
        sample_position(ox, oy, oz, dx, dy, dz, in.ts, in.te, tm, px, py, pz);
        const float X = scale_coord(px, cfg.radius), Y = scale_coord(py, cfg.radius), Z = scale_coord(pz, cfg.radius);
        float e[48];
        const bool any = __any(gather_tex_c(p.packed, (unsigned)(pofs / TT_C), H, W, X, Y, Z, valid, lane, Xs, e));
        TT_PHASE(1);
        if (!any) continue;  // exact: e == 0 => k1 = k2 = 0 and every mask is false
        // e is needed again only as the Y operand of the dV1 outer product: park it in LDS now ([idx][sample]
        // layout, 96 rows) so its 48 registers are free during the MLP chain.
        // (`region`: always true -- tt_validate_cfg rejects negative flags -- but opaque to the compiler.  The two
        // conditional regions below split this ~9000-instruction loop body into separate scheduling / allocation
        // regions: 97 -> 45 spilled registers, 5.46 -> 4.84 ms.  __builtin_amdgcn_sched_barrier does not have that
        // effect; found by noticing that the -DTT_TUNING build, whose ablation branches are live, was FASTER.)
        // With the outer products on the fp16 pipe (WG16) only the scatter keeps its own region: 27 -> 0 spilled
        // registers, 3.37 -> 3.18 ms (no region at all: 67 spills, 3.99 ms; region around the outer products only: 3.57).
        const bool region = cfg.flags >= 0;
        const bool region_w = WG16 ? true : region;
        const bool do_wgrad = region_w && !TT_DBG(cfg.flags, TT_DBG_NO_WGRAD);
        if (do_wgrad) {
            if (WG16)
                stage_rows16<96>(Ys, e, i, hi, sE);
            else
                stage_rows<96>(Ys, e, i, hi);
        }
        TT_PHASE(2);
        float k1[32], k2[32];
        mvx<EXACT, 64, 96>(Lt + TV1, e, k1, i, hi);
#pragma unroll
        for (int r = 0; r < 32; ++r) k1[r] = fmaxf(k1[r], 0.f);
        TT_PHASE(3);
        mvx<EXACT, 64, 64>(Lt + TV2, k1, k2, i, hi);
#pragma unroll
        for (int r = 0; r < 32; ++r) k2[r] = fmaxf(k2[r], 0.f);
        TT_PHASE(4);
        // ---- dV3[o][idx] += sum_s cbar_o[s] k2[idx][s]: k2 goes through the 32-row window in two halves (registers
        // 0..15 hold indices 0..31, registers 16..31 indices 32..63); lane (i, hi) sums samples 16 hi .. 16 hi + 15 of
        // row i against cbar ----
        if (hi == 0) {
            CB_AT(0 * 32 + i) = cb[0];
            CB_AT(1 * 32 + i) = cb[1];
            CB_AT(2 * 32 + i) = cb[2];
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            if (h2 == 0)
                stage_rows_sub<32, 0, 32>(Xs, k2, i, hi);
            else
                stage_rows_sub<32, 16, 32>(Xs, k2, i, hi);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kk = *reinterpret_cast<const f32x4*>(Xs + i * XS + 16 * hi + 4 * g);
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const f32x4 cc = *reinterpret_cast<const f32x4*>(&CB_AT(o * 32 + 16 * hi + 4 * g));
                    accV3[h2][o] += (kk[0] * cc[0] + kk[1] * cc[1]) + (kk[2] * cc[2] + kk[3] * cc[3]);
                }
            }
        }
        TT_PHASE(5);
        // ---- k2bar = n2 . (V3^T cbar) ----
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 0 * 64 + 8 * g + 4 * hi);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 1 * 64 + 8 * g + 4 * hi);
            f32x4 v2 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 2 * 64 + 8 * g + 4 * hi);
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const float t = fmaf(v0[e2], cb[0], fmaf(v1[e2], cb[1], v2[e2] * cb[2]));
                k2[4 * g + e2] = k2[4 * g + e2] > 0.f ? t : 0.f;
            }
        }
        // ---- k1bar = n1 . (V2^T k2bar) ----
        float kb1[32];
        mvtx<EXACT, 64, 64>(Lt + TV2T, Lt + TV2, k2, kb1, i, hi);
#pragma unroll
        for (int r = 0; r < 32; ++r) kb1[r] = k1[r] > 0.f ? kb1[r] : 0.f;
        TT_PHASE(6);
        if (do_wgrad) {
            // ---- dV1 += k1bar e^T  (e parked in Ys rows 0..95; k1bar through the 32-row window, half by half) ----
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                if (WG16) {
                    if (h2 == 0)
                        stage_rows16_sub<32, 0, 32>(Xs, kb1, i, hi, sKB1);
                    else
                        stage_rows16_sub<32, 16, 32>(Xs, kb1, i, hi, sKB1);
                    wgrad16_row<64>(accV1a[h2], Xs, Ys, i, hi);
                    wgrad16_row<32>(accV1b[h2], Xs, Ys + 64 * XS, i, hi);
                } else {
                    if (h2 == 0)
                        stage_rows_sub<32, 0, 32>(Xs, kb1, i, hi);
                    else
                        stage_rows_sub<32, 16, 32>(Xs, kb1, i, hi);
                    wgrad_row<64>(accV1a[h2], Xs, Ys, i, hi);
                    wgrad_row<32>(accV1b[h2], Xs, Ys + 64 * XS, i, hi);
                }
            }
            TT_PHASE(7);
            // ---- dV2 += k2bar k1^T  (e is dead: k2bar in rows 0..63, k1 in rows 64..127) ----
            if (WG16) {
                stage_rows16<64>(Xs, k2, i, hi, sK2B);
                stage_rows16<64>(Xs + 64 * XS, k1, i, hi, sK1);
                wgrad16<64, 64>(accV2, Xs, Xs + 64 * XS, i, hi);
            } else {
                stage_rows<64>(Xs, k2, i, hi);
                stage_rows<64>(Xs + 64 * XS, k1, i, hi);
                wgrad<64, 64>(accV2, Xs, Xs + 64 * XS, i, hi);
            }
            TT_PHASE(8);
        }
        // ---- ebar = V1^T k1bar (one plane at a time) ; scatter texel(3+p, c)[ch] += w_c * ebar[32p + ch] ----
        if (region && !TT_DBG(cfg.flags, TT_DBG_NO_SCATTER)) {
            // the combine GEMM on the fp16 pipe as in the geometry kernel (round 2 measured +26 spilled registers and
            // 3.88 -> 4.16 ms for this; with the outer products on the fp16 pipe it fits: 3.65 -> 3.37 ms)
            constexpr bool SC_EXACT = EXACT || !WG16;  // (the TT_R_WGRAD_F32 A/B variant = the round-2 kernel)
            float* M = Xs;              // rows 0..63: the slot x sample coefficient matrix (fp32), row 64: dump row
            float* Es = Xs + (SC_EXACT ? 65 * XS : SCATTER_M_FLOATS);  // ebar rows [sample][32], stride 33; fallback lists
            scatter_clear<SC_EXACT>(M, lane);
            // ebar = V1^T k1bar for the three planes in ONE product (96 rows: k1bar is split into fp16 terms once)
            float eb[48];
            mvtx<EXACT, 96, 64, V1S>(Lt + TV1T, Lt + TV1, kb1, eb, i, hi);
            TT_PHASE(9);
            const int tex0 = (int)(pofs / TT_C);
            scatter_planes<SC_EXACT>(grad_out, grad_bytes, Es, M, tags, Es + 32 * 33, i, hi, [&](int pl, PlaneRefs& refs) {
                Corners c;
                corners_setup(PLANE_U(pl, X, Y, Z), PLANE_V(pl, X, Y, Z), H, W, valid, c);
                int aoff[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)  // absolute texel index, prompt included
                    aoff[q4] = tex0 + (int)((3 + pl) * HW) + c.off[q4];
                refs = plane_refs<!SC_EXACT>(c.w, aoff, c.hs, hi);
#pragma unroll
                for (int r = 0; r < 16; ++r) Es[i * 33 + LIDX(r, hi)] = eb[16 * pl + r] * refs.qs;
            }
#ifdef TT_TUNING
            , scat_st
#endif
            );
            TT_PHASE(10);
        }
      }
    }
#ifdef TT_TUNING
    TT_PHASE(11);
    ph_acc[15] = scat_st[0];  // active references
    ph_acc[16] = scat_st[1];  // lost references (plane-tiles = 3 per live tile step, slot 12)
    if (p.phase_cycles && lane == 0)
        for (int k = 0; k < 20; ++k) atomicAdd(p.phase_cycles + k, ph_acc[k]);
#endif
    // dV1 is (64, 96) row-major: columns 0..63 from accV1a, 64..95 from accV1b
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * m + LIDX(r, hi);
            const float u1 = 1.f / sKB1, u2 = 1.f / sE;  // inverse operand scales (1 for the fp32 outer products)
            atomicAdd(p.grads.v1 + row * 96 + i, (accV1a[m][0][r] * u1) * u2);
            atomicAdd(p.grads.v1 + row * 96 + 32 + i, (accV1a[m][1][r] * u1) * u2);
            atomicAdd(p.grads.v1 + row * 96 + 64 + i, (accV1b[m][0][r] * u1) * u2);
        }
    flush_wgrad<64, 64>(accV2, p.grads.v2, i, hi, 1.f / sK2B, 1.f / sK1);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int o = 0; o < 3; ++o) atomicAdd(p.grads.v3 + o * 64 + 32 * h2 + i, accV3[h2][o]);
}

// =====================================================================================================
// host side
// =====================================================================================================
static MlpPtrs to_ptrs(const tt_mlp_weights* w) {
    MlpPtrs m;
    m.w1 = w->w1;
    m.w2 = w->w2;
    m.w3 = w->w3;
    m.v1 = w->v1;
    m.v2 = w->v2;
    m.v3 = w->v3;
    return m;
}
static MlpGradPtrs to_gptrs(const tt_mlp_grads* g) {
    MlpGradPtrs m;
    m.w1 = g->w1;
    m.w2 = g->w2;
    m.w3 = g->w3;
    m.v1 = g->v1;
    m.v2 = g->v2;
    m.v3 = g->v3;
    return m;
}

static int debug_flags() {
#ifdef TT_TUNING
    const char* e = getenv("TT_DEBUG_FLAGS");  // profiling ablations, tuning build only
    return e ? (int)strtol(e, nullptr, 0) : 0;
#else
    return 0;
#endif
}

// the scatter addresses texels with 32-bit byte offsets from the (copy of the) gradient buffer
static bool grad_buffer_too_large(const tt_render_cfg* cfg) {
    return (long long)cfg->n_prompts * 6 * cfg->plane_h * cfg->plane_w * TT_C * 4 >= (1LL << 32) - 256;
}

// one 4-wave workgroup per CU (register- and LDS-limited), grid a multiple of 8 (XCD chunking)
static long long persistent_blocks(long long n_items, int cus) {
    long long blocks = cus;
    long long need = (n_items + 3) / 4;
    if (blocks > need) blocks = need;
    return (blocks + 7) / 8 * 8;
}

#ifdef TT_TUNING
static unsigned long long* g_phase_cycles = nullptr;
// tuning build only: cycles per phase of k_decode_bwd_tex summed over waves since the last call (host copy), then reset
extern "C" int tt_tuning_phase_cycles(unsigned long long* out40) {
    if (!g_phase_cycles) {
        if (hipMalloc((void**)&g_phase_cycles, 40 * sizeof(unsigned long long)) != hipSuccess) return -4;
        if (hipMemset(g_phase_cycles, 0, 40 * sizeof(unsigned long long)) != hipSuccess) return -4;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (out40 && hipMemcpy(out40, g_phase_cycles, 40 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return -4;
    return hipMemset(g_phase_cycles, 0, 40 * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
#endif
// ---- per-launch magnitude bounds for the fp16 outer products (tt_host.h: TT_SLOT_BOUNDS) --------------------------
// max |x| over flat float4 data, raised into *out with one atomicMax per workgroup (non-negative floats order like their
// bit patterns; NaN / Inf patterns order above every finite value and end up clamped by wg16_scale).
__global__ __launch_bounds__(256) void k_absmax4(const f32x4* __restrict__ x, long long n4, long long seg4,
                                                 long long seg_stride4, unsigned* __restrict__ out0,
                                                 unsigned* __restrict__ out1) {
    // n4 float4 elements in segments of seg4 elements that start seg_stride4 apart (seg4 == seg_stride4: contiguous).
    // out0 <- max |.x| (and, if out1 is null, of the other three components as well); out1 <- max |.y|, |.z|, |.w|
    float m0 = 0.f, m1 = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
        const long long sgm = e / seg4;
        const f32x4 v = x[sgm * seg_stride4 + (e - sgm * seg4)];
        m0 = fmaxf(m0, __builtin_fabsf(v[0]));
        m1 = fmaxf(m1, fmaxf(__builtin_fabsf(v[1]), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
    }
    if (!out1) m0 = fmaxf(m0, m1);
    __shared__ unsigned w[2];
    if (threadIdx.x < 2) w[threadIdx.x] = 0u;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, o));
        m1 = fmaxf(m1, __shfl_xor(m1, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&w[0], __builtin_bit_cast(unsigned, m0));
        atomicMax(&w[1], __builtin_bit_cast(unsigned, m1));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(out0, w[0]);
        if (out1) atomicMax(out1, w[1]);
    }
}
// flat float data of any length (g_rgb: n_rays x 3, g_features: n x 3): scalar loads
__global__ __launch_bounds__(256) void k_absmax1(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, __builtin_fabsf(x[e]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ unsigned w;
    if (threadIdx.x == 0) w = 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicMax(&w, __builtin_bit_cast(unsigned, m));
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, w);
}
static unsigned absmax_blocks(long long n) {
    long long b = (n + 256 * 8 - 1) / (256 * 8);
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
// max |texel| of planes first_plane .. first_plane + 2 of every prompt of the packed buffer
static void launch_planes_bound(const float* packed, const tt_render_cfg& cfg, int first_plane, unsigned* out,
                                hipStream_t s) {
    const long long hw4 = (long long)cfg.plane_h * cfg.plane_w * TT_C / 4;  // float4s per plane
    const long long n4 = 3 * hw4 * cfg.n_prompts;
    hipLaunchKernelGGL(k_absmax4, dim3(absmax_blocks(n4)), dim3(256), 0, s,
                       reinterpret_cast<const f32x4*>(packed) + first_plane * hw4, n4, 3 * hw4, 6 * hw4, out,
                       (unsigned*)nullptr);
}

static bool use_wg16(const tt_render_cfg& cfg) { return !(cfg.flags & (TT_R_EXACT_F32 | TT_R_WGRAD_F32)); }

static void launch_bwd_geo(const BwdGeoParams& p0, long long blocks, hipStream_t s) {
    BwdGeoParams p = p0;
#ifdef TT_TUNING
    p.phase_cycles = g_phase_cycles;
#else
    p.phase_cycles = nullptr;
#endif
    if (use_wg16(p.cfg)) {
        unsigned* bnd = reinterpret_cast<unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        launch_planes_bound(p.packed, p.cfg, 0, bnd + TT_BOUND_PLANES, s);
        const long long n = p.cfg.n_rays * p.cfg.n_samples;  // upstream float4 (d sdf, d sdf_grad) per sample
        hipLaunchKernelGGL(k_absmax4, dim3(absmax_blocks(n)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(p.ws), n, n,
                           n, bnd + TT_BOUND_UP0, bnd + TT_BOUND_UP1);
        hipLaunchKernelGGL((k_decode_bwd_geo<false, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    } else if (p.cfg.flags & TT_R_EXACT_F32) {
        hipLaunchKernelGGL((k_decode_bwd_geo<true, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((k_decode_bwd_geo<false, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    }
}
static void launch_bwd_tex(const BwdTexParams& p0, long long blocks, hipStream_t s) {
    BwdTexParams p = p0;
#ifdef TT_TUNING
    p.phase_cycles = g_phase_cycles;
#else
    p.phase_cycles = nullptr;
#endif
    if (use_wg16(p.cfg)) {
        unsigned* bnd = reinterpret_cast<unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        launch_planes_bound(p.packed, p.cfg, 3, bnd + TT_BOUND_PLANES, s);  // (p.packed may be slid by 3 planes: points)
        if (p.g_rgb)
            hipLaunchKernelGGL(k_absmax1, dim3(absmax_blocks(p.cfg.n_rays * 3)), dim3(256), 0, s, p.g_rgb,
                               (long long)p.cfg.n_rays * 3, bnd + TT_BOUND_UP0);
        if (p.g_features) {
            const long long n = p.cfg.n_rays * p.cfg.n_samples * 3;
            hipLaunchKernelGGL(k_absmax1, dim3(absmax_blocks(n)), dim3(256), 0, s, p.g_features, n, bnd + TT_BOUND_UP1);
        }
        hipLaunchKernelGGL((k_decode_bwd_tex<false, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    } else if (p.cfg.flags & TT_R_EXACT_F32) {
        hipLaunchKernelGGL((k_decode_bwd_tex<true, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((k_decode_bwd_tex<false, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    }
}

int tt_launch_march_bwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                        const float* sdf, const float* sdf_grad, const float* features, const float* trans,
                        const float* opacity, const float* depth, const float* g_opacity, const float* g_depth,
                        const float* g_rgb_fg, const float* g_z_variance, const float* g_normal_acc,
                        const float* g_weights, const float* g_sdf, const float* g_sdf_grad, float* ws,
                        hipStream_t stream);

extern "C" int tt_render_bwd_geo(const float* packed, const tt_mlp_weights* w, const float* rays_o,
                                 const float* rays_d, const float* t_starts, const float* t_ends,
                                 const tt_render_cfg* cfg, const float* opacity, const float* depth,
                                 const float* trans, const float* sdf, const float* sdf_grad, const float* features,
                                 const float* g_opacity, const float* g_depth, const float* g_rgb_fg,
                                 const float* g_z_variance, const float* g_normal_acc, const float* g_weights,
                                 const float* g_sdf, const float* g_sdf_grad, float* workspace, float* grad_packed,
                                 const tt_mlp_grads* grads, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !opacity || !depth || !trans || !sdf ||
        !sdf_grad || !features || !workspace || !grad_packed || !grads)
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !grads->w1 || !grads->w2 || !grads->w3) return TT_ERR_BAD_ARG;
    if (grad_buffer_too_large(cfg)) return TT_ERR_UNSUPPORTED;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    st = tt_launch_march_bwd(rays_d, t_starts, t_ends, cfg, sdf, sdf_grad, features, trans, opacity, depth, g_opacity,
                             g_depth, g_rgb_fg, g_z_variance, g_normal_acc, g_weights, g_sdf, g_sdf_grad, workspace,
                             s);
    if (st != TT_OK) return st;
    BwdGeoParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.cfg.flags |= debug_flags();
    p.ws = workspace;
    p.grad_packed = grad_packed;
    p.n_copies = cfg->grad_copies > 0 ? cfg->grad_copies : 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(cfg, 4LL * cus, &p.geom, 1);
    long long blocks = persistent_blocks(p.n_items, cus);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    p.queue = tt_queue_counters(s);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_geo(p, blocks, s);
    return tt_check_launch();
}

extern "C" int tt_render_bwd_tex(const float* packed, const tt_mlp_weights* w, const float* rays_o,
                                 const float* rays_d, const float* t_starts, const float* t_ends,
                                 const tt_render_cfg* cfg, const float* weights, const float* features,
                                 const float* g_rgb_fg, const float* g_features, float* grad_packed,
                                 const tt_mlp_grads* grads, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !weights || !features || !grad_packed ||
        !grads)
        return TT_ERR_BAD_ARG;
    if (!w->v1 || !w->v2 || !w->v3 || !grads->v1 || !grads->v2 || !grads->v3) return TT_ERR_BAD_ARG;
    if (grad_buffer_too_large(cfg)) return TT_ERR_UNSUPPORTED;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    BwdTexParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.cfg.flags |= debug_flags();
    p.weights = weights;
    p.features = features;
    p.g_rgb = g_rgb_fg;
    p.g_features = g_features;
    p.grad_packed = grad_packed;
    p.n_copies = cfg->grad_copies > 0 ? cfg->grad_copies : 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(cfg, 4LL * cus, &p.geom, 1);
    long long blocks = persistent_blocks(p.n_items, cus);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    p.queue = tt_queue_counters((hipStream_t)stream);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_tex(p, blocks, (hipStream_t)stream);
    return tt_check_launch();
}

// ---- backward of the per-point queries (tt_query_points / tt_query_field): the same decode-backward kernels, with
// "rays" of one sample whose origin is the point (direction null => x = o exactly) ----
__global__ void k_interleave_ws(const float* __restrict__ g_sdf, const float* __restrict__ g_sdf_grad,
                                float* __restrict__ ws, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f32x4 o = {g_sdf ? g_sdf[i] : 0.f, g_sdf_grad ? g_sdf_grad[i * 3 + 0] : 0.f,
               g_sdf_grad ? g_sdf_grad[i * 3 + 1] : 0.f, g_sdf_grad ? g_sdf_grad[i * 3 + 2] : 0.f};
    *reinterpret_cast<f32x4*>(ws + i * 4) = o;
}

static int points_cfg(tt_render_cfg* c, int32_t n_batch, int64_t n_points, int32_t n_prompts,
                      int32_t views_per_prompt, int32_t plane_h, int32_t plane_w, float radius, float sdf_bias_radius,
                      int32_t grad_copies, int32_t q_flags) {
    if (n_batch <= 0 || n_points <= 0 || n_prompts <= 0 || views_per_prompt <= 0) return TT_ERR_BAD_ARG;
    if ((int64_t)n_prompts * views_per_prompt != n_batch || n_points > 0x7fffffffLL) return TT_ERR_BAD_ARG;
    c->n_prompts = n_prompts;
    c->views_per_prompt = views_per_prompt;
    c->plane_h = plane_h;
    c->plane_w = plane_w;
    c->rays_per_view = (int32_t)n_points;
    c->n_samples = 1;
    c->n_rays = (int64_t)n_batch * n_points;
    c->radius = radius;
    c->sdf_bias_radius = sdf_bias_radius;
    c->inv_std = 1.f;  // unused by the decode kernels
    c->cos_anneal_ratio = 1.f;
    c->rgb_grad_shrink = 1.f;
    c->flags = (q_flags & TT_Q_EXACT_F32) ? TT_R_EXACT_F32 : 0;
    c->image_w = 0;
    c->tile_sb = 1;
    c->tile_chunk = 0;
    c->grad_copies = grad_copies;
    c->skip_eps_tex = 0.f;
    c->skip_eps_geo = 0.f;
    return tt_validate_cfg(c);
}

extern "C" int tt_points_bwd_geo(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                                 int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                                 int32_t plane_w, float radius, float sdf_bias_radius, int32_t flags,
                                 const float* g_sdf, const float* g_sdf_grad, float* workspace, float* grad_packed,
                                 const tt_mlp_grads* grads, void* stream) {
    tt_render_cfg cfg;
    int st = points_cfg(&cfg, n_batch, n_points, n_prompts, views_per_prompt, plane_h, plane_w, radius,
                        sdf_bias_radius, 1, flags);
    if (st != TT_OK) return st;
    if (grad_buffer_too_large(&cfg)) return TT_ERR_UNSUPPORTED;
    if (!packed || !w || !points || !workspace || !grad_packed || !grads || (!g_sdf && !g_sdf_grad))
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !grads->w1 || !grads->w2 || !grads->w3) return TT_ERR_BAD_ARG;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const long long n = cfg.n_rays;
    hipLaunchKernelGGL(k_interleave_ws, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g_sdf, g_sdf_grad,
                       workspace, n);
    st = tt_check_launch();
    if (st != TT_OK) return st;
    BwdGeoParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = points;
    p.rays_d = nullptr;
    p.t_starts = nullptr;
    p.t_ends = nullptr;
    p.cfg = cfg;
    p.ws = workspace;
    p.grad_packed = grad_packed;
    p.n_copies = 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(&cfg, 4LL * cus, &p.geom, 1);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    long long blocks = persistent_blocks(p.n_items, cus);
    p.queue = tt_queue_counters(s);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_geo(p, blocks, s);
    return tt_check_launch();
}

extern "C" int tt_points_bwd_tex(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                                 int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                                 int32_t plane_w, float radius, int32_t plane_base, int32_t flags,
                                 const float* g_features, float* grad_packed, const tt_mlp_grads* grads,
                                 void* stream) {
    tt_render_cfg cfg;
    int st = points_cfg(&cfg, n_batch, n_points, n_prompts, views_per_prompt, plane_h, plane_w, radius, 0.5f, 1,
                        flags);
    if (st != TT_OK) return st;
    if (grad_buffer_too_large(&cfg)) return TT_ERR_UNSUPPORTED;
    if (!packed || !w || !points || !g_features || !grad_packed || !grads) return TT_ERR_BAD_ARG;
    if (!w->v1 || !w->v2 || !w->v3 || !grads->v1 || !grads->v2 || !grads->v3) return TT_ERR_BAD_ARG;
    if (plane_base != 0 && plane_base != 3) return TT_ERR_BAD_ARG;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    // the kernel addresses planes 3..5 of each prompt; plane_base = 0 slides that window onto planes 0..2
    const ptrdiff_t shift = (ptrdiff_t)(plane_base - 3) * plane_h * plane_w * TT_C;
    BwdTexParams p;
    p.packed = packed + shift;
    p.w = to_ptrs(w);
    p.rays_o = points;
    p.rays_d = nullptr;
    p.t_starts = nullptr;
    p.t_ends = nullptr;
    p.cfg = cfg;
    p.weights = nullptr;
    p.features = nullptr;
    p.g_rgb = nullptr;
    p.g_features = g_features;
    p.grad_packed = grad_packed + shift;
    p.n_copies = 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(&cfg, 4LL * cus, &p.geom, 1);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    long long blocks = persistent_blocks(p.n_items, cus);
    p.queue = tt_queue_counters((hipStream_t)stream);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_tex(p, blocks, (hipStream_t)stream);
    return tt_check_launch();
}
