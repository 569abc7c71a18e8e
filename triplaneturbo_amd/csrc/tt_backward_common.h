// tt_backward_common.h -- what the two translation units of the fused render backward share: the LDS transposition /
// outer-product / matrix-core scatter building blocks, the per-launch bound reductions and small host helpers.
// tt_backward.hip holds the geometry half, tt_backward_tex.hip the texture half (separate units: each kernel is
// ~9000 instructions per loop body, and the texture unit is compiled with its own scheduler strategy, _lib.py).
#pragma once
#include "tt_device.h"
#include "tt_mfma16.h"
#include "tt_alpha.h"
#include "tt_host.h"
#include <stdlib.h>
#include <type_traits>

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
__device__ __forceinline__ unsigned wg16_pack(float x) {  // (hi | lo << 16), both round-to-nearest-even: hi + lo ~ x
#if TT_SPLIT_PAIR_ASM
    // three instructions in one block (see split_pair, tt_mfma16.h): hi = f16(x); r = x * 1.0 - hi (exact); pack(x, r)
    unsigned out;
    float r;
    asm("v_cvt_pk_f16_f32 %0, %2, 0\n\t"
        "v_fma_mix_f32 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %1"
        : "=&v"(out), "=&v"(r)
        : "v"(x));
    return out;
#else
    const float hf = (float)cvt_pk16(x, 0.f).x;
#if TT_SPLIT_MODE != 2
    return cvt_pk16u(x, x - hf);  // (both halves with the same rounding: one packed convert)
#else
    const unsigned h = cvt_pk16u(x, 0.f), l = cvt_pk16u_lo(x - hf, 0.f);
    return (h & 0xffffu) | (l << 16);
#endif
#endif
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
// the same rows from a vector already split by split16_vec (tt_mfma16.h): one v_perm per staged dword
template <int N, int PAIR, int NT>
__device__ __forceinline__ void stage_rows16_pre(float* S, const Split16<N, PAIR, NT>& v, int j, int hi) {
    unsigned* U = reinterpret_cast<unsigned*>(S);
#pragma unroll
    for (int t = 0; t < N / 4; ++t) {  // dword t of the split holds registers pair_reg<PAIR>(t, 0 / 1)
        U[LIDX(pair_reg<PAIR>(t, 0), hi) * XS + j] = __builtin_amdgcn_perm(v.l[t], v.h[t], 0x05040100u);  // (hi | lo << 16)
        U[LIDX(pair_reg<PAIR>(t, 1), hi) * XS + j] = __builtin_amdgcn_perm(v.l[t], v.h[t], 0x07060302u);
    }
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

// Workgroup pre-reduction of one 32 x 32 accumulator tile at kernel end: the four waves' copies are summed through LDS
// (R: 2 x 4096 floats, alternating by `parity` so that one barrier per tile is enough) and wave w adds registers
// 4 w .. 4 w + 3 of the sum to memory -- a quarter of the same-address float atomics, which serialise at the memory side
// (~1 000 waves add into the same 66 KB).  All 256 threads call it, in the same order, after the main loop.
template <class Addr>
__device__ __forceinline__ void flush_tile_reduced(float* R, int parity, const f32x16& acc, int wave, int lane,
                                                   float ux, float uy, Addr addr) {
    float* B = R + parity * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) B[(wave * 16 + r) * 64 + lane] = (acc[r] * ux) * uy;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        const float v = (B[r * 64 + lane] + B[(16 + r) * 64 + lane]) + (B[(32 + r) * 64 + lane] + B[(48 + r) * 64 + lane]);
        atomicAdd(addr(r), v);
    }
}
template <int NX, int NY>
__device__ __forceinline__ void flush_wgrad_reduced(float* R, int& parity, const f32x16 (&acc)[NX / 32][NY / 32],
                                                    float* __restrict__ dst, int wave, int lane, float ux, float uy) {
    const int i = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int m = 0; m < NX / 32; ++m)
#pragma unroll
        for (int n = 0; n < NY / 32; ++n) {
            flush_tile_reduced(R, parity, acc[m][n], wave, lane, ux, uy,
                               [&](int r) { return dst + (32 * m + LIDX(r, hi)) * NY + 32 * n + i; });
            parity ^= 1;
        }
}

#define ZERO16 \
    { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }

// ---- plane-gradient scatter, combined on the matrix cores --------------------------------------------------
// (The text below describes the idea and its ROUNDS 2-5 form, kept as -DTT_SCATTER_V1 for A/B builds; the product build
// uses the round-6 form further down -- 16 x 16 texel table, dense ranks, short flush -- under "#else  // !TT_SCATTER_V1".)
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

// (TT_SCATTER_V1, rounds 2-5; kept for A/B builds) The three planes of one tile step, software-pipelined.  Everything except the rare lost-reference path is
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
    int h0, h1;     // entry of the texel table: 16x16 torus hash of the texel coordinates (V1: 8x8 slot)
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
#ifdef TT_SCATTER_V1  // the 8 x 8 window of rounds 2-5 out of the 16 x 16 hash
    r.h0 = ((r.h0 >> 1) & 0x38) | (r.h0 & 7);
    r.h1 = ((r.h1 >> 1) & 0x38) | (r.h1 & 7);
#endif
    return r;
}
__device__ __forceinline__ void scatter_init_tags(int* tags, int lane) {
#ifdef TT_SCATTER_V1
    tags[lane] = -1;
    tags[64 + lane] = -1;
    if (lane < 32) tags[128 + lane] = -2;  // dummies: never empty, never equal to a texel index
#else
    if (lane < 32) tags[lane] = -2;  // the dummy words inactive references CAS: never empty, never equal to a texel index
#endif
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

// references that lost their slot (tile footprint wider than the slot window / texel table: sparse rays) go straight to global
// memory, one half-wave per reference (lanes <-> channels: a coalesced 128-byte atomic each)
__device__ __forceinline__ void scatter_lost(const PlaneRefs& r, bool l0, bool l1, const float* Qs, float* Ls,
                                             __amdgpu_buffer_rsrc_t grsrc, int i, int hi) {
    const float lc0 = l0 ? r.c0 : 0.f, lc1 = l1 ? r.c1 : 0.f;  // coefficient of a lost corner, else 0
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

#ifdef TT_SCATTER_V1
struct ClaimState {
    bool w0, w1;  // wrote M (slot won or shared with the same texel)
    bool m0, m1;  // won the slot: this lane resets the tag
    bool l0, l1;  // lost the slot to a different texel: direct atomics
};

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
#ifdef TT_TUNING  // sub-phase cycles of the epilogue (geometry kernel: st[3..5] = operands+prep / GEMM+claim / flush+reset+lost)
    unsigned long long sp_t = __builtin_amdgcn_s_memtime();
#define TT_SUBPHASE(k)                                                    \
    do {                                                                  \
        if (st) {                                                         \
            __builtin_amdgcn_sched_barrier(0);                            \
            const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
            st[k] += t_now - sp_t;                                        \
            sp_t = t_now;                                                 \
            __builtin_amdgcn_sched_barrier(0);                            \
        }                                                                 \
    } while (0)
#else
#define TT_SUBPHASE(k) \
    do {               \
    } while (0)
#endif
#ifdef TT_SCATTER_DIRECT  // dev A/B (tools/build_variants.py): every reference straight to global memory, no slots
    for (int pl = 0; pl < 3; ++pl) {
        prep(pl, rc);
        sc.w0 = sc.w1 = sc.m0 = sc.m1 = false;
        sc.l0 = sc.l1 = true;
        scatter_lost(rc, sc.l0, sc.l1, Qs, Ls, grsrc, i, hi);
    }
    return;
#endif
    prep(0, rc);
    sc = scatter_claim<EXACT>(rc, M, tags, dummy, i, st);
    scatter_lost(rc, sc.l0, sc.l1, Qs, Ls, grsrc, i, hi);
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
                    h2_t ph, pq;
                    split_pair(x0, x1, ph, pq);
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
        TT_SUBPHASE(3);
        if (pl < 2) sn = scatter_claim<EXACT>(rn, M, tags + 64 * ((pl + 1) & 1), dummy, i, st);
        TT_SUBPHASE(4);
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
            scatter_lost(rn, sn.l0, sn.l1, Qs, Ls, grsrc, i, hi);
            rc = rn;
            sc = sn;
        }
        TT_SUBPHASE(5);
    }
}

#else  // !TT_SCATTER_V1

// =====================================================================================================================
// Round 6: texel table -> DENSE ranks -> short flush, up to 128 distinct texels per plane-tile (two 64-row passes).
//
// What rounds 2-5 did (TT_SCATTER_V1 above): slot = 8 x 8 torus hash of the texel, 64 slots, every plane flushes all 64
// slots (32 atomic instructions, the empty ones dropped by the buffer range check), references that lose their slot go
// out one by one.  Measured / simulated in round 6 (tools/scatter_sim.py, profiles/r06_scatter_*.txt): a plane-tile of the
// headline scene holds 25-34 DISTINCT texels (half of the 32 flush instructions carry nothing), one of the reference's
// sparse training renders 63-73 (up to 128) -- a hash window of 64 slots keeps ~44 of them and sends 28 % of the references
// down the one-by-one path.  Now:
//   * claims go into a 16 x 16 torus table (256 entries; 4 % of the training shape's references collide there, 0.1 % of
//     the headline's) -- one LDS CAS per reference as before;
//   * the distinct texels are RANKED: the winner of an entry (exactly one reference per distinct texel) takes the next rank
//     r < n (two ballots + mbcnt over the winner flags), copies the tag to ctag[r] and leaves r in the entry for the
//     references that share the texel;
//   * M row = rank.  Ranks 0..63 are combined and flushed by pass A, ranks 64..127 (n > 64: wave-uniform branch) by a
//     second pass over the SAME 64-row M buffer and accumulators (32 samples x 4 corners = 128 references: n <= 128 always);
//   * the flush issues 4 ceil(n / 8) atomic instructions instead of 32 (wave-uniform branches per group of 8 rows;
//     ctag[n .. n+7] = -1 keeps the tail of the last group out of range).
// Tags need no double buffer any more: a plane's table entries go back to empty right after they were ranked (DS
// operations of a wave execute in order), only the compacted tags (read by the flush after the next plane's claims) alternate.
// LDS per wave (ints, behind the lost-reference lists at Ls + 256; all inside the kernels' existing scratch):
//   htab[256] | ctag[2][136] | cdump[64]          dummies: tags[0..31] (scatter_init_tags)
#ifndef TT_SC2_NO_PASSB
#define TT_SC2_NO_PASSB 0
#endif
#ifndef TT_SC2_FLUSH_ALL
#define TT_SC2_FLUSH_ALL 0
#endif
#define SC2_HT 256
#define SC2_CT 136
#define SC2_INTS (SC2_HT + 2 * SC2_CT + 64)

struct Sc2State {
    int rx0, rx1;  // dense rank of the lane's two references; 255 = not in the table (inactive, or lost to another texel)
    bool l0, l1;   // not in the table (LOST if the coefficient is non-zero: scatter_lost tests that)
    int n;         // wave-uniform: distinct texels of this plane-tile in the table
};

// claims of one plane, ranking, compacted tags into ct[0 .. n-1] (+ 8 x -1), table back to empty.
// The WINNER of an entry (its CAS found it empty: exactly one reference per distinct texel) takes the next rank -- two
// ballots + mbcnt over the winner flags, no pass over the table --, leaves it in the entry (as -2 - rank: never "empty",
// never a texel index) for the references that share the texel to read, copies the tag to ct[rank], and puts the entry
// back to empty once everybody has read (DS operations of a wave execute in order).
// st (tuning build): per-wave counters [0] active references, [1] lost references, [2] plane-tiles
__device__ __forceinline__ Sc2State sc2_claim_rank(const PlaneRefs& r, int* htab, int* ct, int* cdump, int* dummy,
                                                   int lane, unsigned long long* st = nullptr) {
    Sc2State s;
    const bool a0 = r.c0 != 0.f, a1 = r.c1 != 0.f;
    int* const e0 = a0 ? htab + r.h0 : dummy;
    int* const e1 = a1 ? htab + r.h1 : dummy;
    const int old0 = atomicCAS(e0, -1, r.o0);
    const int old1 = atomicCAS(e1, -1, r.o1);
    const bool w0 = tt_eq_either(old0, -1, r.o0), w1 = tt_eq_either(old1, -1, r.o1);  // the entry holds this texel
    s.l0 = !w0;
    s.l1 = !w1;
    const bool m0 = old0 == -1, m1 = old1 == -1;
    const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1);
    const int n0 = __popcll(b0);
    const int r0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u));
    const int r1 = n0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u));
    s.n = n0 + __popcll(b1);
    int* const x0 = m0 ? e0 : dummy;  // (a value <= -2 in a dummy word is as good as the -2 it starts with)
    int* const x1 = m1 ? e1 : dummy;
    *x0 = -2 - r0;
    *x1 = -2 - r1;
    *(m0 ? ct + r0 : cdump + lane) = r.o0;
    *(m1 ? ct + r1 : cdump + lane) = r.o1;
    *(lane < 8 ? ct + s.n + lane : cdump + lane) = -1;  // the tail of the last flush group: out of range, dropped
    const int q0 = -2 - *e0, q1 = -2 - *e1;
    *x0 = m0 ? -1 : -2;  // the table back to empty
    *x1 = m1 ? -1 : -2;
    s.rx0 = w0 ? q0 : 255;
    s.rx1 = w1 ? q1 : 255;
#ifdef TT_TUNING
    if (st) {  // wave-uniform values, flushed once per wave with the phase timers
        st[0] += __popcll(__ballot(a0)) + __popcll(__ballot(a1));
        st[1] += __popcll(__ballot((s.l0 ? r.c0 : 0.f) != 0.f)) + __popcll(__ballot((s.l1 ? r.c1 : 0.f) != 0.f));
        st[2] += 1;
    }
#endif
    return s;
}
// M rows of a lane's two references in pass A (ranks 0..63) / pass B (ranks 64..127); 64 = the dump row
__device__ __forceinline__ int sc2_row_a(int rx) { return rx < 64 ? rx : 64; }
__device__ __forceinline__ int sc2_row_b(int rx) {
    const unsigned d = (unsigned)(rx - 64);
    return (int)(d < 64u ? d : 64u);
}

// operands of the combine GEMM G[64 rows x 32 ch] = M[64 x 32 samples] Q[32 x 32]: A from the rows of M, B from Qs
struct ScA16 {
    h8_t ah[2][2], al[2][2];  // [row tile][k-step]: 8 samples 16 ks + 8 hi .. + 7 of row 32 m + i
};
struct ScB16 {
    float bs[2][8];  // raw (dead after sc2_split_b)
    h8_t bh[2], bl[2];
    float bun;
};
struct ScA32 {
    f32x4 a4[2][4];
};
struct ScB32 {
    float bq[16];
};
// (n: rows in use, wave-uniform -- the second row tile is read / multiplied only when n > 32)
__device__ __forceinline__ void sc2_load_a(const float* M, int i, int hi, int n, ScA16& A) {
    const half_t* Mh = reinterpret_cast<const half_t*>(M);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m == 1 && n <= 32) break;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half_t* a = Mh + (32 * m + i) * M16_RS + 16 * ks + 8 * hi;
            A.ah[m][ks] = *reinterpret_cast<const h8_t*>(a);
            A.al[m][ks] = *reinterpret_cast<const h8_t*>(a + M16_PLANE);
        }
    }
}
__device__ __forceinline__ void sc2_load_a(const float* M, int i, int hi, int n, ScA32& A) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m == 1 && n <= 32) break;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
            A.a4[m][t4] = *reinterpret_cast<const f32x4*>(M + (32 * m + i) * MS + 16 * hi + 4 * t4);
    }
}
__device__ __forceinline__ void sc2_load_b(const float* Qs, int i, int hi, ScB16& B) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) B.bs[ks][j] = Qs[(16 * ks + 8 * hi + j) * 33 + i];
}
__device__ __forceinline__ void sc2_load_b(const float* Qs, int i, int hi, ScB32& B) {
#pragma unroll
    for (int t = 0; t < 16; ++t) B.bq[t] = Qs[(t + 16 * hi) * 33 + i];
}
__device__ __forceinline__ void sc2_split_b(ScB32&) {}
// per-channel normalisation of the B operand to the top of the fp16 range (column = this lane and lane ^ 32) + split
__device__ __forceinline__ void sc2_split_b(ScB16& B) {
    const float (&bs)[2][8] = B.bs;
    float mx = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, __builtin_fabsf(bs[ks][j]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    int E = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
    E = E < 16 ? 16 : (E > 240 ? 240 : E);
    const float bsc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);
    B.bun = __builtin_bit_cast(float, (unsigned)(E - 14) << 23);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = bs[ks][2 * j] * bsc, x1 = bs[ks][2 * j + 1] * bsc;
            h2_t ph, pq;
            split_pair(x0, x1, ph, pq);
            B.bh[ks][2 * j] = ph.x;
            B.bh[ks][2 * j + 1] = ph.y;
            B.bl[ks][2 * j] = pq.x;
            B.bl[ks][2 * j + 1] = pq.y;
        }
}
__device__ __forceinline__ void sc2_gemm(const ScA16& A, const ScB16& B, int n, f32x16& acc0, f32x16& acc1) {
    acc0 = f32x16 ZERO16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.ah[0][ks], B.bh[ks], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.ah[0][ks], B.bl[ks], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.al[0][ks], B.bh[ks], acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] *= B.bun;
    if (n > 32) {
        acc1 = f32x16 ZERO16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.ah[1][ks], B.bh[ks], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.ah[1][ks], B.bl[ks], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.al[1][ks], B.bh[ks], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] *= B.bun;
    }
}
__device__ __forceinline__ void sc2_gemm(const ScA32& A, const ScB32& B, int n, f32x16& acc0, f32x16& acc1) {
    acc0 = f32x16 ZERO16;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a4[0][t >> 2][t & 3], B.bq[t], acc0, 0, 0, 0);
    if (n > 32) {
        acc1 = f32x16 ZERO16;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a4[1][t >> 2][t & 3], B.bq[t], acc1, 0, 0, 0);
    }
}
// flush rows 0 .. n-1 (rounded up to a group of 8): one 128-byte atomic per row pair, straight from the accumulators
// (row of register 4 g + e = LIDX: e + 8 g + 4 hi); ct: the compacted tags of these rows
__device__ __forceinline__ void sc2_flush(const f32x16& acc0, const f32x16& acc1, const int* ct, int n,
                                          __amdgpu_buffer_rsrc_t grsrc, unsigned lane_b, int hi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (TT_SC2_FLUSH_ALL || n > 8 * g) {
            const i32x4 k0 = *reinterpret_cast<const i32x4*>(ct + 8 * g + 4 * hi);
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc0[4 * g + e2], grsrc,
                                                                (int)(((unsigned)k0[e2] << 7) | lane_b), 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (TT_SC2_FLUSH_ALL || n > 32 + 8 * g) {
            const i32x4 k1 = *reinterpret_cast<const i32x4*>(ct + 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc1[4 * g + e2], grsrc,
                                                                (int)(((unsigned)k1[e2] << 7) | lane_b), 0, 0);
        }
    }
}

// prep(pl, refs): corner set-up of plane pl for this lane's sample AND the staging of its row of Q, scaled by refs.qs,
// into Qs[j*33 + ch] (the previous plane's B operand is in registers by then).  M: all-zero on entry and on exit.
// tags: the 32 dummy words (scatter_init_tags).  Ls: 256 floats of lost-reference lists, followed by SC2_INTS ints.
// Pipeline per plane p:  operands A(p) -> registers, M rows back to zero, [n > 64: rows of pass B -> M]; prep(p+1);
//   GEMM A(p); [pass B: flush A(p), operands B(p) -> registers, M back to zero]; claims + ranks(p+1), lost references of
//   p+1 (under the MFMAs); [pass B: GEMM B(p), flush B(p) | else: flush A(p)]; rows of pass A(p+1) -> M.
template <bool EXACT, class Prep>
__device__ __forceinline__ void scatter_planes(float* __restrict__ grad, unsigned grad_bytes, const float* Qs, float* M,
                                               int* tags, float* Ls, int i, int hi, Prep&& prep,
                                               unsigned long long* st = nullptr) {
    // BUFFER atomics with a 32-bit BYTE offset (texel << 7 | channel * 4) from the gradient copy: a tag of -1 gives the
    // offset 0xFFFFFF80 + 4 ch, beyond num_records (the host refuses gradient buffers of 4 GB - 256 B and more), and the
    // hardware range check drops the atomic -- no compare, no exec-mask branch per row.
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(grad, 0, (int)grad_bytes, 0x00020000);
    const unsigned lane_b = 4u * (unsigned)i;
    const int lane = i + 32 * hi;
    int* const dummy = tags + i;
    int* const htab = reinterpret_cast<int*>(Ls + 256);
    int* const ctag = htab + SC2_HT;
    int* const cdump = ctag + 2 * SC2_CT;
    typedef typename std::conditional<EXACT, ScA32, ScA16>::type AOp;
    typedef typename std::conditional<EXACT, ScB32, ScB16>::type BOp;
#ifdef TT_TUNING  // sub-phase cycles of the epilogue (st[3..5] = operands+prep / GEMM+claim / flush+reset+lost)
    unsigned long long sp_t = __builtin_amdgcn_s_memtime();
#define TT_SUBPHASE(k)                                                    \
    do {                                                                  \
        if (st) {                                                         \
            __builtin_amdgcn_sched_barrier(0);                            \
            const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
            st[k] += t_now - sp_t;                                        \
            sp_t = t_now;                                                 \
            __builtin_amdgcn_sched_barrier(0);                            \
        }                                                                 \
    } while (0)
#elif defined(TT_SC2_FENCE)
#define TT_SUBPHASE(k) __builtin_amdgcn_sched_barrier(0)
#else
#define TT_SUBPHASE(k) \
    do {               \
    } while (0)
#endif
    {  // the table starts empty (its LDS is shared with the other phases of the tile step)
        const i32x4 e4 = {-1, -1, -1, -1};
        *reinterpret_cast<i32x4*>(htab + 4 * lane) = e4;
    }
    PlaneRefs rc, rn;
    Sc2State sc, sn;
    prep(0, rc);
    sc = sc2_claim_rank(rc, htab, ctag, cdump, dummy, lane, st);
    scatter_lost(rc, sc.l0, sc.l1, Qs, Ls, grsrc, i, hi);
    m_store<EXACT>(M, sc2_row_a(sc.rx0), i, rc.c0);
    m_store<EXACT>(M, sc2_row_a(sc.rx1), i, rc.c1);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const int* const ct = ctag + SC2_CT * (pl & 1);
        int* const ct_next = ctag + SC2_CT * ((pl + 1) & 1);
        const bool pass_b = TT_SC2_NO_PASSB ? false : sc.n > 64;  // wave-uniform
        f32x16 acc0, acc1;
        AOp A;
        BOp B;
        sc2_load_a(M, i, hi, sc.n, A);
        m_zero<EXACT>(M, sc2_row_a(sc.rx0), i);
        m_zero<EXACT>(M, sc2_row_a(sc.rx1), i);
        if (pass_b) {
            m_store<EXACT>(M, sc2_row_b(sc.rx0), i, rc.c0);
            m_store<EXACT>(M, sc2_row_b(sc.rx1), i, rc.c1);
        }
        sc2_load_b(Qs, i, hi, B);
        if (pl < 2) prep(pl + 1, rn);
        sc2_split_b(B);
        sc2_gemm(A, B, sc.n, acc0, acc1);
        TT_SUBPHASE(3);
        if (pass_b) {  // (rare on dense rays: the first pass is flushed before the next plane's claims)
            sc2_flush(acc0, acc1, ct, 64, grsrc, lane_b, hi);
            sc2_load_a(M, i, hi, sc.n - 64, A);
            m_zero<EXACT>(M, sc2_row_b(sc.rx0), i);
            m_zero<EXACT>(M, sc2_row_b(sc.rx1), i);
        }
        if (pl < 2) sn = sc2_claim_rank(rn, htab, ct_next, cdump, dummy, lane, st);
        TT_SUBPHASE(4);
        if (pass_b) {
            sc2_gemm(A, B, sc.n - 64, acc0, acc1);
            sc2_flush(acc0, acc1, ct + 64, sc.n - 64, grsrc, lane_b, hi);
        } else {
            sc2_flush(acc0, acc1, ct, sc.n, grsrc, lane_b, hi);
        }
        if (pl < 2) {
            scatter_lost(rn, sn.l0, sn.l1, Qs, Ls, grsrc, i, hi);
            m_store<EXACT>(M, sc2_row_a(sn.rx0), i, rn.c0);
            m_store<EXACT>(M, sc2_row_a(sn.rx1), i, rn.c1);
            rc = rn;
            sc = sn;
        }
        TT_SUBPHASE(5);
    }
}
#endif  // TT_SCATTER_V1

struct MlpGradPtrs {
    float* w1;
    float* w2;
    float* w3;
    float* v1;
    float* v2;
    float* v3;
};

// ---- texture half: parameters and weight-image map of tt_backward_tex.hip ----
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
// V1, V2 as split-fp16 images (tt_mfma16.h): every mat-vec product of the kernel runs on the fp16 pipe.  The V2^T / V1^T
// products either read the same images through ds_read_b64_tr_b16 (mv16t: the wave-pair kernel, where LDS is what limits
// the pairs per CU) or use transposed COPIES (TT_BWD_WT_COPIES, the one-wave kernels' default: 43 KB more LDS that nothing
// else wants at one wave per SIMD, and plain ds_read_b128 fragments: texture backward 2.94 -> 2.85 ms, bit-identical).
// The per-wave scratch is 128 rows: the parked e (96 rows) shares it with a 32-row window through which k2 (for dV3)
// and k1bar (for dV1) are transposed in two halves.
#ifndef TT_BWD_WT_COPIES
#define TT_BWD_WT_COPIES 1
#endif
#define TV1T TEX_W_FLOATS
#define TV2T (TV1T + IMG16_FLOATS(96, 64))
#define TEX_W16_FLOATS (TT_BWD_WT_COPIES ? TV2T + IMG16_FLOATS(64, 64) : TEX_W_FLOATS)
// PREC_S3 (three-piece products): the images of the third terms follow V3, and there are NO transposed copies -- with them
// the kernel would need 204 KB of LDS -- so the V2^T / V1^T products read the forward images through ds_read_b64_tr_b16
// (mv16t, as the forward kernels do): 66 KB of images + 76 KB of per-wave scratch = 143 KB.
#define TLO_V1 TEX_W_FLOATS
#define TLO_V2 (TLO_V1 + LO16_FLOATS(64, 96))
#define TEX_W3P_FLOATS (TLO_V2 + LO16_FLOATS(64, 64))
template <int PREC>
struct TexWFloats {
    static constexpr int value = PREC == PREC_S3 ? TEX_W3P_FLOATS : TEX_W16_FLOATS;
};

// =====================================================================================================
// host side
// =====================================================================================================
static inline MlpPtrs to_ptrs(const tt_mlp_weights* w) {
    MlpPtrs m;
    m.w1 = w->w1;
    m.w2 = w->w2;
    m.w3 = w->w3;
    m.v1 = w->v1;
    m.v2 = w->v2;
    m.v3 = w->v3;
    return m;
}
static inline MlpGradPtrs to_gptrs(const tt_mlp_grads* g) {
    MlpGradPtrs m;
    m.w1 = g->w1;
    m.w2 = g->w2;
    m.w3 = g->w3;
    m.v1 = g->v1;
    m.v2 = g->v2;
    m.v3 = g->v3;
    return m;
}

static inline int debug_flags() {
#ifdef TT_TUNING
    const char* e = getenv("TT_DEBUG_FLAGS");  // profiling ablations, tuning build only
    return e ? (int)strtol(e, nullptr, 0) : 0;
#else
    return 0;
#endif
}

// the scatter addresses texels with 32-bit byte offsets from the (copy of the) gradient buffer
static inline bool grad_buffer_too_large(const tt_render_cfg* cfg) {
    return (long long)cfg->n_prompts * 6 * cfg->plane_h * cfg->plane_w * TT_C * 4 >= (1LL << 32) - 256;
}

// one 4-wave workgroup per CU (register- and LDS-limited), grid a multiple of 8 (XCD chunking)
static inline long long persistent_blocks(long long n_items, int cus) {
    long long blocks = cus;
    long long need = (n_items + 3) / 4;
    if (blocks > need) blocks = need;
    return (blocks + 7) / 8 * 8;
}

#ifdef TT_TUNING
extern unsigned long long* g_phase_cycles;  // tt_backward.hip (tuning build only)
#endif
// ---- per-launch magnitude bounds for the fp16 outer products (tt_host.h: TT_SLOT_BOUNDS) --------------------------
// max |x| over flat float4 data, raised into *out with one atomicMax per workgroup (non-negative floats order like their
// bit patterns; NaN / Inf patterns order above every finite value and end up clamped by wg16_scale).
static __global__ __launch_bounds__(256) void k_absmax4(const f32x4* __restrict__ x, long long seg4, long long seg_stride4,
                                                 unsigned* __restrict__ out0, unsigned* __restrict__ out1) {
    // gridDim.y segments of seg4 float4 elements that start seg_stride4 apart; four independent loads in flight per lane.
    // out0 <- max |.x| (and, if out1 is null, of the other three components as well); out1 <- max |.y|, |.z|, |.w|
    const f32x4* seg = x + (long long)blockIdx.y * seg_stride4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float m0 = 0.f, m1 = 0.f;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; e + 3 * stride < seg4; e += 4 * stride) {
        const f32x4 a = seg[e], b = seg[e + stride], c = seg[e + 2 * stride], d = seg[e + 3 * stride];
        m0 = fmaxf(fmaxf(m0, __builtin_fabsf(a[0])), fmaxf(__builtin_fabsf(b[0]), fmaxf(__builtin_fabsf(c[0]), __builtin_fabsf(d[0]))));
#pragma unroll
        for (int k = 1; k < 4; ++k)
            m1 = fmaxf(fmaxf(m1, __builtin_fabsf(a[k])), fmaxf(__builtin_fabsf(b[k]), fmaxf(__builtin_fabsf(c[k]), __builtin_fabsf(d[k]))));
    }
    for (; e < seg4; e += stride) {
        const f32x4 v = seg[e];
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
    // same-address atomics serialise at ~11 ns each (4096 workgroups x 2 words: 90 us, measured): the words only grow, so
    // a workgroup first LOOKS (L2-coherent load) and raises a word only if that changes it
    if (threadIdx.x == 0) {
        if (w[0] > __hip_atomic_load(out0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out0, w[0]);
        if (out1 && w[1] > __hip_atomic_load(out1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out1, w[1]);
    }
}
// flat float data of any length (g_rgb: n_rays x 3, g_features: n x 3): scalar loads
static __global__ __launch_bounds__(256) void k_absmax1(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, __builtin_fabsf(x[e]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ unsigned w;
    if (threadIdx.x == 0) w = 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicMax(&w, __builtin_bit_cast(unsigned, m));
    __syncthreads();
    if (threadIdx.x == 0 && w > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, w);
}
static inline unsigned absmax_blocks(long long n) {  // >= 8 elements per thread, at most 1024 workgroups
    long long b = (n + 256 * 8 - 1) / (256 * 8);
    return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
// max |texel| of planes first_plane .. first_plane + 2 of every prompt of the packed buffer
static inline void launch_planes_bound(const float* packed, const tt_render_cfg& cfg, int first_plane, unsigned* out,
                                hipStream_t s) {
    const long long hw4 = (long long)cfg.plane_h * cfg.plane_w * TT_C / 4;  // float4s per plane
    hipLaunchKernelGGL(k_absmax4, dim3(absmax_blocks(3 * hw4), (unsigned)cfg.n_prompts), dim3(256), 0, s,
                       reinterpret_cast<const f32x4*>(packed) + first_plane * hw4, 3 * hw4, 6 * hw4, out,
                       (unsigned*)nullptr);
}

// outer products on the fp16 pipe: both split modes (the fp32 MFMA mode and the tuning build's TT_R_WGRAD_F32 A/B kernel
// use fp32 outer products)
static inline bool use_wg16(const tt_render_cfg& cfg) { return !(cfg.flags & (TT_R_EXACT_F32 | TT_R_WGRAD_F32)); }

static inline int points_cfg(tt_render_cfg* c, int32_t n_batch, int64_t n_points, int32_t n_prompts,
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
    c->inv_std_dev = nullptr;
    c->stats = nullptr;
    c->cos_anneal_ratio = 1.f;
    c->rgb_grad_shrink = 1.f;
    if (!tt_qflags_ok(q_flags)) return TT_ERR_BAD_ARG;
    c->flags = (q_flags & TT_Q_EXACT_F32) ? TT_R_EXACT_F32 : ((q_flags & TT_Q_SPLIT2) ? TT_R_SPLIT2 : 0);
    c->image_w = 0;
    c->tile_sb = 1;
    c->tile_chunk = 0;
    c->grad_copies = grad_copies;
    c->skip_eps_tex = 0.f;
    c->skip_eps_geo = 0.f;
    return tt_validate_cfg(c);
}

