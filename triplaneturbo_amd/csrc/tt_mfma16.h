// tt_mfma16.h -- fp32-accurate mat-vec products on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, 16x the rate of
// v_mfma_f32_32x32x2_f32 on gfx950).
//
// Every fp32 operand is split in two fp16 terms, v = hi + lo with hi = f16(v) and lo = f16(v - hi)  (v - hi is exact
// in fp32).  Operands are first normalised by a power of two so that the largest |v| of the matrix / of the sample's
// activation vector sits in [2^14, 2^15) -- the TOP of the fp16 range -- so that lo (< 2^-10 |v|) is a NORMAL fp16
// number, i.e. hi + lo carries 22 significand bits, for every entry within 2^-19 of the largest (below that lo is an
// fp16 subnormal, which the gfx950 MFMA honours -- tools/mfma16_probe.hip -- at an absolute 2^-24, i.e. 2^-39 of the
// largest entry).  A product keeps the three terms above 2^-20:
//      a b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi
// = 3 MFMAs per 16-deep k-step into ONE fp32 accumulator (fp16 x fp16 products are exact, accumulation is fp32 inside
// the MFMA) instead of 8 fp32 MFMAs of twice the issue time: 5.3x less matrix-pipe time.  Error per dot product
// <= ~2^-23 |a|max |b|max per term, i.e. fp32-grade norm-wise (fp32 accumulation itself is ~sqrt(K) 2^-24).
// (Round 1 normalised to [0.5, 1) and scaled lo by 2^11 to keep it normal, which needed a second accumulator set per
// row tile -- 32 more registers in a 64-row product -- and a combine pass over the outputs: forward 2.15 -> 1.94 ms,
// backward 4.13 / 5.84 -> 3.97 / 5.68 ms with the single accumulator.)
//
// Layout.  Activations stay in the LIDX register layout of tt_device.h (reg r of lane (j, hi) <-> element
// (r&3) + 8(r>>2) + 4hi): the 8 registers 8s..8s+7 of a lane are the 8 k-slots that lane feeds to k-step s of a
// 32x32x16 MFMA (B operand); which element sits in which slot does not matter as long as the A operand uses the same
// assignment, so the weight image is stored pre-permuted: row R of a [ROWS][K] matrix holds, for k-step s, term t in
// {hi, lo} and half-wave h, the 8 halfs  W[R][16s + (j&3) + 8(j>>2) + 4h], j = 0..7  at
//      R * (2K + 8) + ((2s + t) * 2 + h) * 8      (in halfs)
// i.e. exactly the bytes and the 16-byte-per-lane, conflict-free read pattern of the fp32 image (row stride K+4
// floats).  `W^T x` products use a second image built from the transposed matrix.
#pragma once
#include "tt_device.h"

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half_t;
typedef float f2_t __attribute__((ext_vector_type(2)));

#define IMG16_FLOATS(ROWS, K) ((ROWS) * ((K) + 4)) /* LDS floats one (hi, mid) image occupies */
#define LO16_FLOATS(ROWS, K) ((ROWS) * ((K) / 2 + 4)) /* ... and the image of the THIRD terms (three-piece mode) */

// ---- THREE-PIECE mode (NT = 3; TT_R_SPLIT3, "fp32-grade products on the fp16 pipe", round 5) -----------------------------
// The two-piece product above carries ~2^-21.5 per product (tools/mfma16_probe.hip): each operand is represented to 2^-23
// and the lo x lo term is dropped.  The reference multiplies in fp32 (threestudio/models/networks.py:91-97, autocast off),
// and in ill-conditioned scenes (NeuS alpha = a ratio of nearly equal sigmoids, inv_std = 100) those two bits decide whether
// a gradient lands within 1e-4 of the fp32 oracle.  With NT = 3 every operand is split EXACTLY,
//      v = hi + mid + lo,   hi = f16(v), mid = f16(v - hi), lo = f16(v - hi - mid)      (all residuals exact in fp32;
//      11 + 11 + 11 significand bits + signs >= the 24 of fp32, for every entry within 2^-15 of the largest of its
//      column / matrix; below that the absolute error is 2^-25, i.e. 2^-39 of the largest),
// and a product keeps the SIX terms above 2^-33:  hh ; hm, mh ; hl, lh, mm  -- 6 MFMAs per 16-deep k-step into the same
// fp32 accumulator, issued small-to-large inside a k-step.  What is left is the fp32 accumulation of the MFMA itself, i.e.
// the same class of error as the k-ordered fmaf chain of the fp32 MFMA / of the reference's GEMM.
// Image: the (hi, mid) pair is stored EXACTLY like the two-piece (hi, lo) image (same bytes, same reads); the third terms
// live in a second image in the same two-term format over K / 2 columns -- "term 0" = lo of columns [0, K/2), "term 1" =
// lo of columns [K/2, K) -- so that both the row-wise ds_read_b128 fragments and the transposed ds_read_b64_tr_b16 fragments
// keep the conflict-free addressing of the format (row stride K/2 + 4 dwords = 4 mod 16).

// Two fp32 -> one dword of two fp16, ROUND-TO-NEAREST-EVEN: v_cvt_pk_f16_f32 (gfx950; one instruction per pair, like the
// round-toward-zero v_cvt_pkrtz_f16_f32 it replaces since round 4).  With RNE |v - hi| <= 2^-11 |v| and lo = f16(v - hi)
// is again correctly rounded, so hi + lo carries ~24 significand bits instead of the ~22 of the truncating split
// (tools/mfma16_probe.hip measures both).  Nothing can overflow: operands are normalised below 2^15 < 65504.
#ifndef TT_SPLIT_MODE
// dev A/B (tools/build_variants.py), measured on one box, ms per bench step (forward / geometry / texture backward):
//   1 RTZ (rounds 1-3) 7.77 (1.70 / 2.88 / 2.86)   0 RNE through __builtin_convertvector 8.01 (1.85 / 2.90 / 2.92): hipcc
//   re-schedules around the fptrunc (+6 spilled registers in k_decode_rays)   4 RNE, the SAME instruction as inline asm
//   7.81 (1.70 / 2.88 / 2.89), bit-identical results to 0   2 hi RNE / lo RTZ 7.89   3 bias + RTZ 8.23   5 lo only asm 7.85
#define TT_SPLIT_MODE 4
#endif
__device__ __forceinline__ h2_t cvt_pk16_rne(float a, float b) {
    const f2_t v = {a, b};
    return __builtin_convertvector(v, h2_t);
}
__device__ __forceinline__ h2_t cvt_pk16_rtz(float a, float b) {
    return __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// round to nearest (ties away from zero) on the truncating instruction: half an fp16 ulp (bit 12 of the fp32 pattern)
// added to the magnitude first
__device__ __forceinline__ h2_t cvt_pk16_bias(float a, float b) {
    return cvt_pk16_rtz(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) + 0x1000u),
                        __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) + 0x1000u));
}
__device__ __forceinline__ h2_t cvt_pk16_asm(float a, float b) {  // the same instruction, opaque to the optimiser
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return __builtin_bit_cast(h2_t, r);
}
// hi term of a split
__device__ __forceinline__ h2_t cvt_pk16(float a, float b) {
#if TT_SPLIT_MODE == 4
    return cvt_pk16_asm(a, b);
#elif TT_SPLIT_MODE == 1
    return cvt_pk16_rtz(a, b);
#elif TT_SPLIT_MODE == 3
    return cvt_pk16_bias(a, b);
#else
    return cvt_pk16_rne(a, b);
#endif
}
// lo term (the residual) of a split
__device__ __forceinline__ h2_t cvt_pk16_lo(float a, float b) {
#if TT_SPLIT_MODE == 4 || TT_SPLIT_MODE == 5
    return cvt_pk16_asm(a, b);
#elif TT_SPLIT_MODE == 1 || TT_SPLIT_MODE == 2
    return cvt_pk16_rtz(a, b);
#elif TT_SPLIT_MODE == 3
    return cvt_pk16_bias(a, b);
#else
    return cvt_pk16_rne(a, b);
#endif
}
__device__ __forceinline__ unsigned cvt_pk16u(float a, float b) { return __builtin_bit_cast(unsigned, cvt_pk16(a, b)); }
__device__ __forceinline__ unsigned cvt_pk16u_lo(float a, float b) { return __builtin_bit_cast(unsigned, cvt_pk16_lo(a, b)); }

// The split of a PAIR of values: hi = pk_f16(a, b), lo = pk_f16(a - hi.x, b - hi.y) (both residuals exact).
// TT_SPLIT_PAIR_ASM: one asm block of FOUR instructions -- convert, two v_fma_mix_f32 that subtract the f16 halves of hi
// straight from the fp32 inputs (a * 1.0 - hi: exact), convert -- where hipcc emits six (convert, v_cvt_f32_f16, subtract,
// v_cvt_f32_f16_sdwa, subtract, convert).  It also removes the wait states hipcc puts around a lone inline-asm convert
// (it assumes a dst_sel forwarding hazard for every asm result: +95 s_nop per tile step of the texture backward, which is
// what the single-instruction asm form of round 4 cost against the truncating builtin).  Bit-identical to the generic form.
#ifndef TT_SPLIT_PAIR_ASM
#define TT_SPLIT_PAIR_ASM (TT_SPLIT_MODE == 4)
#endif
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
#if TT_SPLIT_PAIR_ASM
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mix_f32 %2, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %3, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %1, %2, %3"
        : "=&v"(hi), "=v"(lo), "+v"(a), "+v"(b));
#else
    const h2_t p = cvt_pk16(a, b);
    hi = __builtin_bit_cast(unsigned, p);
    lo = cvt_pk16u_lo(a - (float)p.x, b - (float)p.y);
#endif
}
__device__ __forceinline__ void split_pair(float a, float b, h2_t& hi, h2_t& lo) {
    unsigned h, l;
    split_pair(a, b, h, l);
    hi = __builtin_bit_cast(h2_t, h);
    lo = __builtin_bit_cast(h2_t, l);
}

// The EXACT three-piece split of a pair (NT = 3): hi, mid as above, lo = pk_f16(a - hi - mid) -- seven instructions in one
// block (convert, two exact fma_mix residuals, convert, two more residuals, convert).
__device__ __forceinline__ void split_pair3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
#if TT_SPLIT_PAIR_ASM
    asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
        "v_fma_mix_f32 %3, %3, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %4, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %1, %3, %4\n\t"
        "v_fma_mix_f32 %3, %3, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %4, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %2, %3, %4"
        : "=&v"(hi), "=&v"(mid), "=&v"(lo), "+v"(a), "+v"(b));
#else
    const h2_t p = cvt_pk16(a, b);
    hi = __builtin_bit_cast(unsigned, p);
    const float ra = a - (float)p.x, rb = b - (float)p.y;
    const h2_t q = cvt_pk16_lo(ra, rb);
    mid = __builtin_bit_cast(unsigned, q);
    lo = cvt_pk16u_lo(ra - (float)q.x, rb - (float)q.y);
#endif
}
__device__ __forceinline__ void split_pair3(float a, float b, h2_t& hi, h2_t& mid, h2_t& lo) {
    unsigned h, m, l;
    split_pair3(a, b, h, m, l);
    hi = __builtin_bit_cast(h2_t, h);
    mid = __builtin_bit_cast(h2_t, m);
    lo = __builtin_bit_cast(h2_t, l);
}

__device__ __forceinline__ void split16(float v, half_t& hi, half_t& lo) {
    const h2_t p = cvt_pk16(v, 0.f);
    hi = p.x;
    const h2_t q = cvt_pk16_lo(v - (float)hi, 0.f);  // may be subnormal
    lo = q.x;
}
__device__ __forceinline__ void split16_3(float v, half_t& hi, half_t& mid, half_t& lo) {
    split16(v, hi, mid);
    const h2_t q = cvt_pk16_lo((v - (float)hi) - (float)mid, 0.f);  // both residuals exact in fp32
    lo = q.x;
}

// Builds the image of M (ROWS x K).  src is row-major: M[R][c] = src[R * ld + c], or, with TRANSPOSED, the image of
// src^T: M[R][c] = src[c * ld + R].  The matrix is first normalised by the power of two that brings its largest
// |entry| into [2^14, 2^15) (exact; any weight scale, from 1e-30 to 1e30, then splits with full precision and nothing
// can overflow fp16); the inverse factor is stored in the first pad float of EVERY row (so row slices of an image carry
// it) and mv16 folds it into its result scaling.  All threads of the workgroup must call this (it synchronises).
// NT = 3: the third terms go to the image at `lo_f` (LO16_FLOATS(ROWS, K) floats; header comment): row stride K + 8 halfs,
// column c at ((c % (K/2)) / 16 * 4 + h) * 8 + j + 16 * (c / (K/2)).
template <int ROWS, int K, bool TRANSPOSED, int NT = 2>
__device__ __forceinline__ void stage_image16(float* dst_f, const float* __restrict__ src, int ld, float* lo_f = nullptr) {
    half_t* dst = reinterpret_cast<half_t*>(dst_f);
    half_t* dlo = reinterpret_cast<half_t*>(lo_f);
    constexpr int RS = 2 * K + 8, RSL = K + 8;
    unsigned* slot = reinterpret_cast<unsigned*>(dst_f + K);  // pad of row 0: max |entry| as bits, then the factor
    if (threadIdx.x == 0) *slot = 0u;
    __syncthreads();
    // max |entry| on the BIT PATTERNS (non-negative floats order like their bit patterns): integer max only -- the float
    // form made hipcc emit NaN-test masks combined on the scalar ALU in this loop (tools/mask_hazard_lint.py)
    unsigned mb = 0u;
    for (int e = threadIdx.x; e < ROWS * K; e += blockDim.x) {
        const unsigned b = __builtin_bit_cast(unsigned, src[e]) & 0x7fffffffu;
        mb = b > mb ? b : mb;
    }
    atomicMax(slot, mb);
    __syncthreads();
    int E = (int)(*slot >> 23);
    E = E < 16 ? 16 : (E > 240 ? 240 : E);
    const float sc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);  // 2^(141 - E): max |entry| -> [2^14, 2^15)
    const float un = __builtin_bit_cast(float, (unsigned)(E - 14) << 23);   // 1 / sc
    __syncthreads();
    for (int e = threadIdx.x; e < ROWS * K; e += blockDim.x) {
        const int R = e / K, c = e - R * K;
        const float v = (TRANSPOSED ? src[(size_t)c * ld + R] : src[(size_t)R * ld + c]) * sc;
        half_t hi, lo, l3 = (half_t)0.f;
        if (NT == 3)
            split16_3(v, hi, lo, l3);
        else
            split16(v, hi, lo);
        // element c = 16 s + (j&3) + 8 (j>>2) + 4 h
        const int s = c >> 4, w = c & 15;
        const int h = (w >> 2) & 1, j = (w & 3) + 4 * (w >> 3);
        half_t* p = dst + (size_t)R * RS + (size_t)(4 * s + h) * 8 + j;
        p[0] = hi;   // term 0
        p[16] = lo;  // term 1: +2 half-wave blocks of 8 halfs
        if (NT == 3) {
            const int hf = c >= K / 2 ? 1 : 0, s2 = (c - hf * (K / 2)) >> 4;
            dlo[(size_t)R * RSL + (size_t)(4 * s2 + h) * 8 + j + 16 * hf] = l3;
        }
    }
    for (int R = threadIdx.x; R < ROWS; R += blockDim.x) dst_f[(size_t)R * (K + 4) + K] = un;
}

// Compiler scheduling fence in front of the split + MFMA loop of a product (after the per-sample scale search): hipcc may
// not mix the tail of the previous product / the scale search into the loop, nor hoist the loop's LDS reads above it.
// Measured per translation unit: the forward kernels gain 6-7 % (k_decode_rays 1.80 -> 1.68 ms, the fused eval kernel
// 2.38 -> 1.93 ms), so tt_forward.hip defines TT_MV16_FENCE 1 before including this header; the backward kernels do not
// (profiles/experiments/README.md).  __builtin_amdgcn_sched_barrier emits no instruction.  (Found through s_setprio around
// the loop, which helps by the same amount with priority 0: it is the fence, not the wave priority, that matters.)
#ifndef TT_MV16_FENCE
#define TT_MV16_FENCE 0
#endif

// y[NOUT] = M[NOUT][NIN] x[NIN], M given as an image; x, y in the LIDX register layout.
// SCALED: every sample (= MFMA column = this lane and its partner lane ^ 32) is first normalised by the power of two
// that brings its largest |x| into [2^14, 2^15) and the result is scaled back -- exact, and it makes the product
// independent of the magnitude of x (gradient chains carry values of 1e-10 that fp16 cannot hold; columns of a
// mat-vec are independent, so each gets its own exponent).
//
// DEFERRED FACTORS (RAW): every scale in this scheme is a power of two, so "multiply the accumulators back" commutes exactly
// with everything a hidden vector goes through before it is consumed (ReLU, masks by sign, the next product's
// normalisation).  With RAW the product returns the accumulators as they are plus the per-lane factor `*yf` (true value =
// y * yf); the next product takes that factor as `xf`: its per-sample exponent search runs on the raw values (same operand
// bits as on the true ones) and xf goes into ITS result factor.  One multiply per hidden value and layer less, bit-identical
// results.
// ---- the MFMA schedule of one k-step, shared by every product below --------------------------------------------------
// NT = 2: hh, hl, lh (the three terms above 2^-20).  NT = 3: the six terms above 2^-33, small to large (a0 / b0 = hi,
// a1 / b1 = mid, a2 / b2 = lo): h.l, l.h, m.m ; h.m, m.h ; h.h.  Row tiles are interleaved so that consecutive MFMAs never
// depend on each other when MT > 1.
template <int MT, int NT>
__device__ __forceinline__ void mfma_terms(f32x16 (&acc)[MT], const h8_t (&a0)[MT], const h8_t (&a1)[MT],
                                           const h8_t (&a2)[MT], const h8_t& b0, const h8_t& b1, const h8_t& b2) {
    if constexpr (NT == 3) {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[m], b2, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[m], b0, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[m], b1, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[m], b1, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[m], b0, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[m], b0, acc[m], 0, 0, 0);
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[m], b0, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[m], b1, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[m], b0, acc[m], 0, 0, 0);
    }
}
// the 8 activations of k-step s (registers r0 .. of x, paired as PAIR says) -> B fragments (NT pieces)
template <int NT, bool SCALED>
__device__ __forceinline__ void split_bfrag(float x0, float x1, float sc, int d, h8_t& b0, h8_t& b1, h8_t& b2) {
    const f2_t ab = {x0, x1};
    const f2_t as = SCALED ? ab * sc : ab;  // exact (power of two)
    h2_t p, q, r;
    if constexpr (NT == 3)
        split_pair3(as.x, as.y, p, q, r);
    else
        split_pair(as.x, as.y, p, q);
    b0[2 * d] = p.x;
    b0[2 * d + 1] = p.y;
    b1[2 * d] = q.x;
    b1[2 * d + 1] = q.y;
    if constexpr (NT == 3) {
        b2[2 * d] = r.x;
        b2[2 * d + 1] = r.y;
    }
}
// row-wise A fragment of the third terms: k-step s of a K-column matrix sits in the lo image (row stride K + 8 halfs) at
// k-step s % (KS/2) of "term" s / (KS/2)
template <int K>
__device__ __forceinline__ h8_t lo_frag(const half_t* lrow, int m, int s) {
    constexpr int RSL = K + 8, KH = K / 32;
    return *reinterpret_cast<const h8_t*>(lrow + (size_t)(32 * m) * RSL + 32 * (s % KH) + 16 * (s / KH));
}

template <int NOUT, int NIN, bool SCALED = true, bool RAW = false, int NT = 2>
__device__ __forceinline__ void mv16(const float* img_f, const float (&x)[NIN / 2], float (&y)[NOUT / 2], int i,
                                     int hi, float xf = 1.f, float* yf = nullptr, const float* lo_f = nullptr) {
    constexpr int MT = NOUT / 32, KS = NIN / 16, RS = 2 * NIN + 8;
    const half_t* row = reinterpret_cast<const half_t*>(img_f) + (size_t)i * RS + 8 * hi;
    const half_t* lrow = reinterpret_cast<const half_t*>(lo_f) + (size_t)i * (NIN + 8) + 8 * hi;
    const float wun = img_f[(size_t)i * (NIN + 4) + NIN];  // inverse of the matrix normalisation (stage_image16)
    float sc = 1.f, un = wun;
    if (SCALED) {
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < NIN / 2; ++r) m = fmaxf(m, __builtin_fabsf(x[r]));
        m = fmaxf(m, __shfl_xor(m, 32));
        int E = (int)(__builtin_bit_cast(unsigned, m) >> 23);  // biased exponent (m >= 0)
        E = E < 16 ? 16 : (E > 240 ? 240 : E);                 // keeps both factors normal
        sc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);       // 2^(141 - E): max |x| -> [2^14, 2^15)
        un = wun * __builtin_bit_cast(float, (unsigned)(E - 14) << 23);  // 1 / sc
    }
    un *= xf;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        acc[m] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (TT_MV16_FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        // split the 8 activations of this k-step
        h8_t bh, bl, bt;
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bfrag<NT, SCALED>(x[8 * s + 2 * j], x[8 * s + 2 * j + 1], sc, j, bh, bl, bt);
        // all terms of a row tile go to the same accumulator
        h8_t ah[MT], al[MT], at[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const half_t* a = row + (size_t)(32 * m) * RS + 32 * s;
            ah[m] = *reinterpret_cast<const h8_t*>(a);
            al[m] = *reinterpret_cast<const h8_t*>(a + 16);
            if constexpr (NT == 3) at[m] = lo_frag<NIN>(lrow, m, s);
        }
        mfma_terms<MT, NT>(acc, ah, al, at, bh, bl, bt);
    }
    if (RAW) *yf = un;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) y[16 * m + k] = RAW ? acc[m][k] : acc[m][k] * un;
}

// ---- operands split ONCE per vector, with a scale known before the launch --------------------------------------------
// A vector that is both the input of a mat-vec product and one side of a weight-gradient outer product (which needs one
// scale for ALL samples: tt_backward_common.h) is split once under that per-launch scale: hi / lo pairs in the k-slot
// order of mv16, from which the outer product's (hi | lo << 16) dwords are one v_perm each.  Used for vectors whose
// magnitude does not depend on the sample's upstream gradient (activations, masked weight products): their bound is within
// a few binades of every sample's own maximum, so the per-launch scale loses nothing against the per-sample one.
// Which two registers share a dword of a B fragment.  PAIR_SEQ: dword t <-> registers 2 t, 2 t + 1 (products whose A operand
// is read row-wise from the image).  PAIR_TR: inside every group of 8 registers (one k-step) the dwords hold registers
// (0,2) (4,6) (1,3) (5,7): the k-slot order [t0 t2 t4 t6 | t1 t3 t5 t7] in which the TRANSPOSED products receive their A
// fragments from two ds_read_b64_tr_b16 (mv16t below).
#define PAIR_SEQ 0
#define PAIR_TR 1
template <int PAIR>
__device__ __forceinline__ constexpr int pair_reg(int t, int which) {
    return PAIR == PAIR_SEQ ? 2 * t + which : 8 * (t >> 2) + ((t & 3) >> 1) + 4 * (t & 1) + 2 * which;
}
// NT = 3: (h, l) are the hi and MID pieces -- what the outer products stage, exactly as in two-piece mode -- and t the
// third; t is dead once the mat-vec product has consumed it.
template <int N, int PAIR = PAIR_SEQ, int NT = 2>
struct Split16 {
    unsigned h[N / 4], l[N / 4];  // dword t <-> registers pair_reg<PAIR>(t, 0 / 1) of the LIDX layout: (f16 hi | f16 hi' << 16), same for lo
    unsigned t[NT == 3 ? N / 4 : 1];
};
template <int N, int PAIR, int NT>
__device__ __forceinline__ void split16_vec(const float (&x)[N / 2], float sc, Split16<N, PAIR, NT>& o) {
#pragma unroll
    for (int t = 0; t < N / 4; ++t) {
        const f2_t ab = {x[pair_reg<PAIR>(t, 0)], x[pair_reg<PAIR>(t, 1)]};
        const f2_t as = ab * sc;  // exact (power of two)
        if constexpr (NT == 3)
            split_pair3(as.x, as.y, o.h[t], o.l[t], o.t[t]);
        else
            split_pair(as.x, as.y, o.h[t], o.l[t]);
    }
}
// mv16 on a pre-split operand: y = M x with x = (hi + lo) * un_x
template <int NOUT, int NIN, bool RAW = false, int NT = 2>
__device__ __forceinline__ void mv16_pre(const float* img_f, const Split16<NIN, PAIR_SEQ, NT>& x, float un_x,
                                         float (&y)[NOUT / 2], int i, int hi, float* yf = nullptr,
                                         const float* lo_f = nullptr) {
    constexpr int MT = NOUT / 32, KS = NIN / 16, RS = 2 * NIN + 8;
    const half_t* row = reinterpret_cast<const half_t*>(img_f) + (size_t)i * RS + 8 * hi;
    const half_t* lrow = reinterpret_cast<const half_t*>(lo_f) + (size_t)i * (NIN + 8) + 8 * hi;
    const float un = img_f[(size_t)i * (NIN + 4) + NIN] * un_x;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        acc[m] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const h8_t bh = __builtin_bit_cast(h8_t, u4_t{x.h[4 * s], x.h[4 * s + 1], x.h[4 * s + 2], x.h[4 * s + 3]});
        const h8_t bl = __builtin_bit_cast(h8_t, u4_t{x.l[4 * s], x.l[4 * s + 1], x.l[4 * s + 2], x.l[4 * s + 3]});
        h8_t bt;
        if constexpr (NT == 3)
            bt = __builtin_bit_cast(h8_t, u4_t{x.t[4 * s], x.t[4 * s + 1], x.t[4 * s + 2], x.t[4 * s + 3]});
        h8_t ah[MT], al[MT], at[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const half_t* a = row + (size_t)(32 * m) * RS + 32 * s;
            ah[m] = *reinterpret_cast<const h8_t*>(a);
            al[m] = *reinterpret_cast<const h8_t*>(a + 16);
            if constexpr (NT == 3) at[m] = lo_frag<NIN>(lrow, m, s);
        }
        mfma_terms<MT, NT>(acc, ah, al, at, bh, bl, bt);
    }
    if (RAW) *yf = un;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) y[16 * m + k] = RAW ? acc[m][k] : acc[m][k] * un;
}

// ---- TRANSPOSED products from the forward image: y = M^T x without a second (transposed) image ----------------------------
// gfx950's LDS transpose read, ds_read_b64_tr_b16: inside every group of 16 lanes, output lane i, element j = half (i % 4)
// of the 8-byte chunk addressed by lane 4 j + i / 4 (tools/ds_tr_probe.hip).  The forward image of M (rows R = the
// contraction index of M^T x, row length KM) stores every aligned group of four consecutive columns as four contiguous
// halfs, so a lane group pointed at rows {R0, R0 + 2, R0 + 8, R0 + 10} receives, per lane = column c, the four k-slots
// t = 0, 2, 4, 6 of an MFMA A fragment of M^T, and a second read one row further down the slots t = 1, 3, 5, 7.  With that
// choice of rows (and the images' row stride of K + 4 floats = 4 banks mod 64) the 32 lanes of a read cycle touch all 64
// banks exactly once: conflict-free.  The B operand pairs its registers accordingly (PAIR_TR).  Rounds 1-3 kept a second,
// transposed image of W1, W2 (26 KB) and of V1, V2 (43 KB) per workgroup for these products.
typedef short sv4_t __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) sv4_t lds_sv4_t;

// per-lane base of the transposed fragments of the NOUT columns that start at column col0 (a multiple of 16)
template <int KM>
__device__ __forceinline__ const lds_sv4_t* tr_lane_base(const float* img_f, int col0, int lane) {
    constexpr int RS = 2 * KM + 8;
    const int L = lane & 15, g1 = (lane >> 4) & 1, hh = lane >> 5, j = L >> 2, q = L & 3;
    const int r1 = 2 * (j & 1) + 8 * (j >> 1) + 4 * hh;
    const int off = r1 * RS + 32 * ((col0 >> 4) + g1) + 8 * (q & 1) + 4 * (q >> 1);  // halfs; a multiple of 4
    return (const lds_sv4_t*)(reinterpret_cast<const half_t*>(img_f) + off);
}
// A fragment of M^T: k-step ks (16 rows of M), column tile m (32 columns), term 0 (hi) / 1 (lo)
template <int KM>
__device__ __forceinline__ h8_t tr_frag(const lds_sv4_t* base, int ks, int m, int term) {
    constexpr int RS = 2 * KM + 8;
    const int off4 = (16 * ks * RS + 64 * m + 16 * term) / 4;  // in 8-byte units (compile-time constant after unrolling)
    const sv4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_sv4_t*>(base) + off4);
    const sv4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_sv4_t*>(base) + off4 + RS / 4);
    typedef short sv8_t __attribute__((__vector_size__(8 * sizeof(short))));
    const sv8_t ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h8_t, ab);
}

// The same for the image of the third terms (NT = 3): the 16-column block cb of M sits at k-step cb % (KM/32) of "term"
// cb / (KM/32) of the lo image, so the two 16-lane groups of a 32-column tile (blocks 2 m' and 2 m' + 1) are in general NOT
// a fixed distance apart: one per-lane base per column tile (MT <= 3 address registers; the compiler folds them when the
// distance is the same for all tiles, e.g. KM = 64).  Bank conflicts: none for KM = 64; two-way on half of the banks for
// KM = 32 and for the middle tile of KM = 96 (one fragment in three of those products).
template <int KM>
__device__ __forceinline__ const lds_sv4_t* tr_lane_base_lo(const float* lo_f, int col0, int m, int lane) {
    constexpr int RSL = KM + 8, KH = KM / 32;
    const int L = lane & 15, g1 = (lane >> 4) & 1, hh = lane >> 5, j = L >> 2, q = L & 3;
    const int r1 = 2 * (j & 1) + 8 * (j >> 1) + 4 * hh;
    const int cb0 = (col0 >> 4) + 2 * m, cb1 = cb0 + 1;
    const int b0 = 32 * (cb0 % KH) + 16 * (cb0 / KH), b1 = 32 * (cb1 % KH) + 16 * (cb1 / KH);
    const int off = r1 * RSL + (g1 ? b1 : b0) + 8 * (q & 1) + 4 * (q >> 1);  // halfs; a multiple of 4
    return (const lds_sv4_t*)(reinterpret_cast<const half_t*>(lo_f) + off);
}
template <int KM>
__device__ __forceinline__ h8_t tr_frag_lo(const lds_sv4_t* base_m, int ks) {
    constexpr int RSL = KM + 8;
    const int off4 = (16 * ks * RSL) / 4;
    const sv4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_sv4_t*>(base_m) + off4);
    const sv4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_sv4_t*>(base_m) + off4 + RSL / 4);
    typedef short sv8_t __attribute__((__vector_size__(8 * sizeof(short))));
    const sv8_t ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h8_t, ab);
}

// y[NOUT] = M[NIN][col0 .. col0 + NOUT)^T x[NIN]; img_f = forward image of M (NIN rows, KM columns)
template <int NOUT, int NIN, int KM, bool SCALED = true, bool RAW = false, int NT = 2>
__device__ __forceinline__ void mv16t(const float* img_f, int col0, const float (&x)[NIN / 2], float (&y)[NOUT / 2],
                                      int lane, float xf = 1.f, float* yf = nullptr, const float* lo_f = nullptr) {
    constexpr int MT = NOUT / 32, KS = NIN / 16;
    const lds_sv4_t* base = tr_lane_base<KM>(img_f, col0, lane);
    const lds_sv4_t* base_lo[MT];
    if constexpr (NT == 3) {
#pragma unroll
        for (int m = 0; m < MT; ++m) base_lo[m] = tr_lane_base_lo<KM>(lo_f, col0, m, lane);
    }
    const float wun = img_f[KM];  // inverse of the matrix normalisation (the same in the pad of every row)
    float sc = 1.f, un = wun;
    if (SCALED) {
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < NIN / 2; ++r) m = fmaxf(m, __builtin_fabsf(x[r]));
        m = fmaxf(m, __shfl_xor(m, 32));
        int E = (int)(__builtin_bit_cast(unsigned, m) >> 23);
        E = E < 16 ? 16 : (E > 240 ? 240 : E);
        sc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);
        un = wun * __builtin_bit_cast(float, (unsigned)(E - 14) << 23);
    }
    un *= xf;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        acc[m] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (TT_MV16_FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        h8_t bh, bl, bt;  // k-slot order [t0 t2 t4 t6 | t1 t3 t5 t7]
#pragma unroll
        for (int d = 0; d < 4; ++d)
            split_bfrag<NT, SCALED>(x[8 * s + pair_reg<PAIR_TR>(d, 0)], x[8 * s + pair_reg<PAIR_TR>(d, 1)], sc, d, bh, bl, bt);
        h8_t ah[MT], al[MT], at[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[m] = tr_frag<KM>(base, s, m, 0);
            al[m] = tr_frag<KM>(base, s, m, 1);
            if constexpr (NT == 3) at[m] = tr_frag_lo<KM>(base_lo[m], s);
        }
        mfma_terms<MT, NT>(acc, ah, al, at, bh, bl, bt);
    }
    if (RAW) *yf = un;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) y[16 * m + k] = RAW ? acc[m][k] : acc[m][k] * un;
}
// the same on an operand already split in PAIR_TR order (x = (hi + lo) * un_x)
template <int NOUT, int NIN, int KM, int NT = 2>
__device__ __forceinline__ void mv16t_pre(const float* img_f, int col0, const Split16<NIN, PAIR_TR, NT>& x, float un_x,
                                          float (&y)[NOUT / 2], int lane, const float* lo_f = nullptr) {
    constexpr int MT = NOUT / 32, KS = NIN / 16;
    const lds_sv4_t* base = tr_lane_base<KM>(img_f, col0, lane);
    const lds_sv4_t* base_lo[MT];
    if constexpr (NT == 3) {
#pragma unroll
        for (int m = 0; m < MT; ++m) base_lo[m] = tr_lane_base_lo<KM>(lo_f, col0, m, lane);
    }
    const float un = img_f[KM] * un_x;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        acc[m] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const h8_t bh = __builtin_bit_cast(h8_t, u4_t{x.h[4 * s], x.h[4 * s + 1], x.h[4 * s + 2], x.h[4 * s + 3]});
        const h8_t bl = __builtin_bit_cast(h8_t, u4_t{x.l[4 * s], x.l[4 * s + 1], x.l[4 * s + 2], x.l[4 * s + 3]});
        h8_t bt;
        if constexpr (NT == 3)
            bt = __builtin_bit_cast(h8_t, u4_t{x.t[4 * s], x.t[4 * s + 1], x.t[4 * s + 2], x.t[4 * s + 3]});
        h8_t ah[MT], al[MT], at[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[m] = tr_frag<KM>(base, s, m, 0);
            al[m] = tr_frag<KM>(base, s, m, 1);
            if constexpr (NT == 3) at[m] = tr_frag_lo<KM>(base_lo[m], s);
        }
        mfma_terms<MT, NT>(acc, ah, al, at, bh, bl, bt);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) y[16 * m + k] = acc[m][k] * un;
}

// ---- precision switch ----------------------------------------------------------------------------------------------
// PREC_S2 (default until round 5; now the named FAST mode, TT_R_SPLIT2): two-piece operands, 3 MFMAs per k-step, ~2^-21.5 per
//   product.
// PREC_S3 (TT_R_SPLIT3, the default since round 5): three-piece operands, 6 MFMAs per k-step: fp32-grade products (header).
// PREC_F32 (cfg.flags & TT_R_EXACT_F32): every matrix product on v_mfma_f32_32x32x2_f32 (bit-for-bit a k-ordered fmaf
//   chain, 1/16 of the fp16 pipe's rate) from plain fp32 weight images -- the A/B reference of both split modes.
// The fp32 image of a matrix occupies the same LDS floats as its (hi, mid) split-fp16 image (row stride K + 4), so the
// kernels' LDS maps do not depend on F32-vs-split; PREC_S3 appends the images of the third terms (`lo` pointers below,
// ignored by the other two modes).
enum { PREC_S2 = 0, PREC_F32 = 1, PREC_S3 = 2 };
template <int PREC>
struct PrecNT {
    static constexpr int value = PREC == PREC_S3 ? 3 : 2;
};
template <int PREC, int ROWS, int K>
__device__ __forceinline__ void stage_weights(float* dst_f, float* lo_f, const float* __restrict__ src) {
    if constexpr (PREC == PREC_F32) {
        lds_load_matrix(dst_f, src, ROWS, K, K + 4);
    } else {
        stage_image16<ROWS, K, false, PrecNT<PREC>::value>(dst_f, src, K, lo_f);
    }
}

// image of src^T (src is ROWS_SRC x K_SRC row-major) for kernels that keep a transposed COPY for their `M^T x` products
// instead of reading the forward image through ds_read_b64_tr_b16: the one-wave backward kernels, where LDS is not scarce
// (occupancy is register-bound) and ds_read_b128 fragments issue at twice the rate of the transposed reads.  Nothing to
// do under PREC_F32.
template <int PREC, int ROWS_SRC, int K_SRC>
__device__ __forceinline__ void stage_weights_t(float* dst_f, float* lo_f, const float* __restrict__ src) {
    if constexpr (PREC != PREC_F32) stage_image16<K_SRC, ROWS_SRC, true, PrecNT<PREC>::value>(dst_f, src, K_SRC, lo_f);
}

// y[NOUT] = M[NOUT][NIN] x
// RAW (deferred factors, see mv16): y comes back unscaled with its per-lane factor in *yf, x may carry a factor xf.  The
// fp32 path has no factors: it returns true values and *yf = 1 (callers pass 1 on as xf).
template <int PREC, int NOUT, int NIN, bool RAW = false>
__device__ __forceinline__ void mvx(const float* img, const float* lo, const float (&x)[NIN / 2], float (&y)[NOUT / 2],
                                    int i, int hi, float xf = 1.f, float* yf = nullptr) {
    if constexpr (PREC == PREC_F32) {
        mv_fwd<NOUT, NIN>(img, x, y, i, hi);
        if (RAW) *yf = 1.f;
    } else {
        mv16<NOUT, NIN, true, RAW, PrecNT<PREC>::value>(img, x, y, i, hi, xf, yf, lo);
    }
}
// y[NOUT] = M^T x with `img_t` / `lo_t` the split-fp16 images of M^T (stage_weights_t) and `img` the fp32 image of M (F32)
template <int PREC, int NOUT, int NIN, int KM, bool RAW = false>
__device__ __forceinline__ void mvtx_copy(const float* img_t, const float* lo_t, const float* img,
                                          const float (&x)[NIN / 2], float (&y)[NOUT / 2], int i, int hi, float xf = 1.f,
                                          float* yf = nullptr) {
    if constexpr (PREC == PREC_F32) {
        mv_bwd<NOUT, NIN, KM + 4>(img, x, y, i, hi);
        if (RAW) *yf = 1.f;
    } else {
        mv16<NOUT, NIN, true, RAW, PrecNT<PREC>::value>(img_t, x, y, i, hi, xf, yf, lo_t);
    }
}
// y[NOUT] = M[:, col0 .. col0 + NOUT)^T x for M (NIN rows, KM columns) staged by stage_weights at `img` (+ `lo`): from the
// fp32 image by strided column reads (F32), from the split-fp16 images by transposed reads (mv16t)
template <int PREC, int NOUT, int NIN, int KM, bool RAW = false>
__device__ __forceinline__ void mvtx(const float* img, const float* lo, int col0, const float (&x)[NIN / 2],
                                     float (&y)[NOUT / 2], int i, int hi, float xf = 1.f, float* yf = nullptr) {
    if constexpr (PREC == PREC_F32) {
        mv_bwd<NOUT, NIN, KM + 4>(img + col0, x, y, i, hi);
        if (RAW) *yf = 1.f;
    } else {
        mv16t<NOUT, NIN, KM, true, RAW, PrecNT<PREC>::value>(img, col0, x, y, 32 * hi + i, xf, yf, lo);
    }
}

// ---- LDS map of the third-term images (PREC_S3) of the forward-shaped kernels: appended to the fp32-sized map of
// tt_device.h (OFF_W1 ... LDS_W_FLOATS), so the (hi, mid) images keep their offsets in every mode ----
#define LO_W1 LDS_W_FLOATS
#define LO_W2 (LO_W1 + LO16_FLOATS(64, 32))
#define LO_V1 (LO_W2 + LO16_FLOATS(64, 64))
#define LO_V2 (LO_V1 + LO16_FLOATS(64, 96))
#define LDS_W3P_FLOATS (LO_V2 + LO16_FLOATS(64, 64))
// floats of weight images a forward-shaped kernel keeps in LDS
template <int PREC>
struct FwdWFloats {
    static constexpr int value = PREC == PREC_S3 ? LDS_W3P_FLOATS : LDS_W_FLOATS;
};
// host side: precision of a launch from the flag bits of tt_abi.h (EXACT_F32 wins; SPLIT2 = the fast mode; default S3)
#define TT_PREC_OF(exact, split2) ((exact) ? PREC_F32 : ((split2) ? PREC_S2 : PREC_S3))
