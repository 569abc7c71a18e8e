// tt_forward.hip -- plane pack/unpack, per-point decode (tt_query_points) and the fused forward render.
#ifndef TT_FWD_PREFETCH
#define TT_FWD_PREFETCH 1  // next tile step's interval loaded one step ahead: forward 1.676 -> 1.666 ms (A/B/A/B, one box)
#endif
#ifndef TT_MV16_FENCE
#define TT_MV16_FENCE 1  // scheduling fence in front of every product's MFMA loop: see tt_mfma16.h
#endif
#include "tt_device.h"
#include "tt_mfma16.h"
#include "tt_alpha.h"
#include "tt_host.h"

#include <stdlib.h>

// =====================================================================================================
// plane pack: (P,6,32,H,W) NCHW  ->  (P,6,H,W,32) channels-last with rotate_planes "v1" folded in
//   R0[h,w] = P0[w,h]   R1[h,w] = P1[H-1-h, W-1-w]   R2[h,w] = P2[H-1-w, h]   (few_step...:212-225)
// One block per (plane, source row y): the 32 x W slab is read coalesced along x, transposed through
// LDS and written as whole 128-byte texels.
// =====================================================================================================
__device__ __forceinline__ void rot_dst(int k3, int y, int x, int H, int W, int& h, int& w) {
    if (k3 == 0) {
        h = x;
        w = y;
    } else if (k3 == 1) {
        h = H - 1 - y;
        w = W - 1 - x;
    } else {
        h = x;
        w = H - 1 - y;
    }
}

// UNPACK: src holds n_copies privatised copies (stride copy_stride floats) that are summed on the fly.
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_planes_pack(const float* __restrict__ src, float* __restrict__ dst, int H,
                                                     int W, int n_copies, size_t copy_stride) {
    extern __shared__ float tile[];  // [32][W+1]
    const int y = blockIdx.x, plane = blockIdx.y;  // plane = p*6 + k
    const int k3 = plane % 3;
    const size_t HW = (size_t)H * W;
    const float* nchw_c = (UNPACK ? dst : src) + (size_t)plane * TT_C * HW;
    float* nchw = const_cast<float*>(nchw_c);
    const float* nhwc_c = (UNPACK ? src : dst) + (size_t)plane * HW * TT_C;
    float* nhwc = const_cast<float*>(nhwc_c);
    const int ws = W + 1;
    if (!UNPACK) {
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int c = e / W, x = e - c * W;
            tile[c * ws + x] = nchw[(size_t)c * HW + (size_t)y * W + x];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int x = e >> 5, c = e & 31, h, w;
            rot_dst(k3, y, x, H, W, h, w);
            nhwc[((size_t)h * W + w) * TT_C + c] = tile[c * ws + x];
        }
    } else {
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int x = e >> 5, c = e & 31, h, w;
            rot_dst(k3, y, x, H, W, h, w);
            const size_t o = ((size_t)h * W + w) * TT_C + c;
            float acc = nhwc[o];
            for (int k = 1; k < n_copies; ++k) acc += nhwc[o + (size_t)k * copy_stride];
            tile[c * ws + x] = acc;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int c = e / W, x = e - c * W;
            nchw[(size_t)c * HW + (size_t)y * W + x] = tile[c * ws + x];
        }
    }
}

// The same for W % 4 == 0 (every size the reference uses): 16-byte accesses on the NCHW side and on both LDS sides (a
// quarter of the instructions), whole 128-byte texel lines per half-wave on the channels-last side.  tile = [32][RS],
// RS = W + pad with RS = 4 (mod 32): the 16-byte LDS reads of 8 consecutive channel rows then cover all 32 banks.
__host__ __device__ inline int pack4_row_stride(int W) { return W + (36 - W % 32) % 32; }
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_planes_pack4(const float* __restrict__ src, float* __restrict__ dst, int H,
                                                      int W, int n_copies, size_t copy_stride) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [32][RS]
    const int y = blockIdx.x, plane = blockIdx.y;  // plane = p*6 + k
    const int k3 = plane % 3;
    const size_t HW = (size_t)H * W;
    const float* nchw_c = (UNPACK ? dst : src) + (size_t)plane * TT_C * HW + (size_t)y * W;
    float* nchw = const_cast<float*>(nchw_c);
    const float* nhwc_c = (UNPACK ? src : dst) + (size_t)plane * HW * TT_C;
    float* nhwc = const_cast<float*>(nhwc_c);
    const int RS = pack4_row_stride(W), W4 = W >> 2;
    const int c_lane = threadIdx.x & 31, x4_0 = threadIdx.x >> 5;
    if (!UNPACK) {
        for (int e4 = threadIdx.x; e4 < TT_C * W4; e4 += 256) {
            const int c = e4 / W4, x4 = e4 - c * W4;
            *reinterpret_cast<f32x4*>(tile + c * RS + 4 * x4) =
                *reinterpret_cast<const f32x4*>(nchw_c + (size_t)c * HW + 4 * x4);
        }
        __syncthreads();
        for (int x4 = x4_0; x4 < W4; x4 += 8) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile + c_lane * RS + 4 * x4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int h, w;
                rot_dst(k3, y, 4 * x4 + j, H, W, h, w);
                nhwc[((size_t)h * W + w) * TT_C + c_lane] = v[j];
            }
        }
    } else {
        for (int x4 = x4_0; x4 < W4; x4 += 8) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int h, w;
                rot_dst(k3, y, 4 * x4 + j, H, W, h, w);
                const size_t o = ((size_t)h * W + w) * TT_C + c_lane;
                float acc = nhwc_c[o];
                for (int k = 1; k < n_copies; ++k) acc += nhwc_c[o + (size_t)k * copy_stride];
                v[j] = acc;
            }
            *reinterpret_cast<f32x4*>(tile + c_lane * RS + 4 * x4) = v;
        }
        __syncthreads();
        for (int e4 = threadIdx.x; e4 < TT_C * W4; e4 += 256) {
            const int c = e4 / W4, x4 = e4 - c * W4;
            *reinterpret_cast<f32x4*>(nchw + (size_t)c * HW + 4 * x4) =
                *reinterpret_cast<const f32x4*>(tile + c * RS + 4 * x4);
        }
    }
}

// =====================================================================================================
// per-tile decode (forward): gather -> sdf net (+ input-gradient chain) -> feature net
// =====================================================================================================
struct DecodeCfg {
    const float* planes;  // packed planes: n_prompts x 6 x H x W x 32
    unsigned tex0;        // texel index of this lane's prompt
    float* T;             // wave-private LDS scratch of the coalesced gathers (GC_SCRATCH_FLOATS)
    int H, W;
    float radius;
    float ju, jv;  // 0.5*W/radius, 0.5*H/radius
    int dbg;       // profiling-only ablation flags
    TileStats st = {nullptr};  // work accounting of the render kernels (tt_render_cfg.stats), else null
};

// outputs are identical in both half-waves.  gq = J^T q (WITHOUT the sphere term).
// the weight images in L are split-fp16 images (tt_mfma16.h); the W2^T / W1^T products of the normal chain read the SAME
// images through transposed LDS reads (mv16t) -- rounds 1-3 kept 26 KB of transposed copies here

// texture half: e -> feature net -> c (3 raw features).  Lanes with !valid gather nothing (their c is 0).
template <int PREC>
__device__ __forceinline__ void decode_tex_fwd(const float* L, const DecodeCfg& dc, float X, float Y, float Z,
                                               bool valid, int i, int hi, float (&c)[3]) {
    c[0] = c[1] = c[2] = 0.f;
    float e[48];
    bool any = gather_tex_cp(dc.planes, dc.tex0, dc.H, dc.W, X, Y, Z, valid, 32 * hi + i, dc.T, e);
    if (TT_DBG(dc.dbg, TT_DBG_NO_MLP)) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 48; ++r) t += e[r];
        c[0] = c[1] = c[2] = t;
    } else if (__any(any)) {  // exact skip: e == 0 for the whole tile => features == 0 (bias-free MLP)
        // hidden vectors stay in RAW form (accumulators + a per-lane power-of-two factor, tt_mfma16.h): ReLU and the next
        // product's normalisation do not care, the factor is applied once to the three outputs
        float k1[32], k2[32], u1, u2;
        mvx<PREC, 64, 96, true>(L + OFF_V1, L + LO_V1, e, k1, i, hi, 1.f, &u1);
#pragma unroll
        for (int r = 0; r < 32; ++r) k1[r] = fmaxf(k1[r], 0.f);
        mvx<PREC, 64, 64, true>(L + OFF_V2, L + LO_V2, k1, k2, i, hi, u1, &u2);
#pragma unroll
        for (int r = 0; r < 32; ++r) k2[r] = fmaxf(k2[r], 0.f);
#pragma unroll
        for (int o = 0; o < 3; ++o) c[o] = dot_lds<64>(L + OFF_V3 + 64 * o, k2, hi) * u2;
    }
}

// geometry half: f (+ J) -> sdf net -> s0 and, if NEED_N, gq = J^T q (WITHOUT the sphere term)
template <bool NEED_N, int PREC>
__device__ __forceinline__ void decode_geo_fwd(const float* L, const DecodeCfg& dc, float X, float Y, float Z,
                                               bool valid, int i, int hi, float& s0, float (&gq)[3]) {
    s0 = 0.f;
    gq[0] = gq[1] = gq[2] = 0.f;
    float f[16], jx[16], jy[16], jz[16];
    bool any = gather_geo_c<NEED_N>(dc.planes, dc.tex0, dc.H, dc.W, X, Y, Z, valid, dc.ju, dc.jv, 32 * hi + i, dc.T, f, jx,
                                 jy, jz, tile_stat_ptr(dc.st, TT_STAT_INBOUNDS));
    if (TT_DBG(dc.dbg, TT_DBG_NO_MLP)) {
        float t = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            t += f[r];
            tx += jx[r];
            ty += jy[r];
            tz += jz[r];
        }
        s0 = t;
        gq[0] = tx;
        gq[1] = ty;
        gq[2] = tz;
    } else if (__any(any)) {
        tile_stat(dc.st, TT_STAT_EXECUTED);
        float h1[32], h2[32], u1, u2;  // RAW hidden vectors (see decode_tex_fwd): only their signs and the dot product matter
        mvx<PREC, 64, 32, true>(L + OFF_W1, L + LO_W1, f, h1, i, hi, 1.f, &u1);
#pragma unroll
        for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
        mvx<PREC, 64, 64, true>(L + OFF_W2, L + LO_W2, h1, h2, i, hi, u1, &u2);
#pragma unroll
        for (int r = 0; r < 32; ++r) h2[r] = fmaxf(h2[r], 0.f);
        s0 = dot_lds<64>(L + OFF_W3, h2, hi) * u2;
        if (NEED_N) {
            // reverse-mode input gradient: a2 = m2 . w3 ; a1 = m1 . (W2^T a2) ; q = W1^T a1
            float a2[32], a1[32], q[16];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f32x4 w3 = *reinterpret_cast<const f32x4*>(L + OFF_W3 + 8 * g + 4 * hi);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) a2[4 * g + e2] = h2[4 * g + e2] > 0.f ? w3[e2] : 0.f;
            }
            float ua1;
            mvtx<PREC, 64, 64, 64, true>(L + OFF_W2, L + LO_W2, 0, a2, a1, i, hi, 1.f, &ua1);
#pragma unroll
            for (int r = 0; r < 32; ++r) a1[r] = h1[r] > 0.f ? a1[r] : 0.f;
            mvtx<PREC, 32, 64, 32>(L + OFF_W1, L + LO_W1, 0, a1, q, i, hi, ua1);
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sx = fmaf(q[r], jx[r], sx);
                sy = fmaf(q[r], jy[r], sy);
                sz = fmaf(q[r], jz[r], sz);
            }
            gq[0] = sx + __shfl_xor(sx, 32);
            gq[1] = sy + __shfl_xor(sy, 32);
            gq[2] = sz + __shfl_xor(sz, 32);
        }
    }
}

template <bool NEED_N, bool NEED_TEX, int PREC>
__device__ __forceinline__ void decode_fwd(const float* L, const DecodeCfg& dc, float px, float py, float pz,
                                           bool valid, int i, int hi, float& s0, float (&gq)[3], float (&c)[3]) {
    const float X = scale_coord(px, dc.radius), Y = scale_coord(py, dc.radius), Z = scale_coord(pz, dc.radius);
    c[0] = c[1] = c[2] = 0.f;
    if (NEED_TEX) decode_tex_fwd<PREC>(L, dc, X, Y, Z, valid, i, hi, c);
    decode_geo_fwd<NEED_N, PREC>(L, dc, X, Y, Z, valid, i, hi, s0, gq);
}

// =====================================================================================================
// tt_query_points
// =====================================================================================================
struct QueryParams {
    const float* packed;
    MlpPtrs w;
    const float* points;
    int n_batch;
    long long n_points;
    int views_per_prompt;
    int H, W;
    float radius, bias_radius;
    float* out_sdf;
    float* out_grad;
    float* out_feat;
};

// weight images of the forward decode kernels: split-fp16 (tt_mfma16.h)
template <bool NEED_N, bool NEED_TEX, int PREC>
__device__ __forceinline__ void stage_decode_images(float* L, const MlpPtrs& w) {
    stage_weights<PREC, 64, 32>(L + OFF_W1, L + LO_W1, w.w1);
    stage_weights<PREC, 64, 64>(L + OFF_W2, L + LO_W2, w.w2);
    lds_load_matrix(L + OFF_W3, w.w3, 1, 64, 64);
    if (NEED_TEX) {
        stage_weights<PREC, 64, 96>(L + OFF_V1, L + LO_V1, w.v1);
        stage_weights<PREC, 64, 64>(L + OFF_V2, L + LO_V2, w.v2);
        lds_load_matrix(L + OFF_V3, w.v3, 3, 64, 64);
    }
}

#ifndef TT_DR_GEO_SAMPLES
#define TT_DR_GEO_SAMPLES 64  // samples of a ray block per work item of the sdf-only decode (tt_decode_rays)
#endif
#ifndef TT_DR_GEO_MIN_ITEMS
#define TT_DR_GEO_MIN_ITEMS 0  // ... and no minimum number of items per wave slot
#endif
// 8 waves (2 per SIMD) share one set of split-fp16 weight images (93 KB: one workgroup per CU)
#ifndef DECODE_THREADS
#define DECODE_THREADS 512
#endif

// Staggered wave priority (dev A/B, TT_PRIO_MODE; 0 = off): the two waves of a SIMD run the same phases (gather -> products -> ...)
// and, arbitrated evenly, tend to stay in step -- both in their matrix phase, then both in their VALU phase -- so the pipes
// alternate instead of overlapping.  A STATIC higher priority for one wave of each pair lets it run as if alone while the other
// fills the slots it leaves.  Mode 1: waves 4..7 of the 8-wave workgroup (wave w sits on SIMD w % 4); mode 2: odd waves.
#ifndef TT_PRIO_MODE
#define TT_PRIO_MODE 0
#endif
__device__ __forceinline__ void tt_stagger_priority() {
#if TT_PRIO_MODE == 1
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(3);
#elif TT_PRIO_MODE == 2
    if ((threadIdx.x >> 6) & 1) __builtin_amdgcn_s_setprio(3);
#elif TT_PRIO_MODE == 3
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
#endif
}
template <bool NEED_N, bool NEED_TEX, int PREC>
__global__ __launch_bounds__(DECODE_THREADS) void k_query_points(QueryParams p) {
    __shared__ __attribute__((aligned(16))) float L[FwdWFloats<PREC>::value + (DECODE_THREADS / 64) * GC_SCRATCH_FLOATS];
    float* T = L + FwdWFloats<PREC>::value + (threadIdx.x >> 6) * GC_SCRATCH_FLOATS;
    stage_decode_images<NEED_N, NEED_TEX, PREC>(L, p.w);
    __syncthreads();
    tt_stagger_priority();
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const long long tiles_per_batch = (p.n_points + TT_TILE - 1) / TT_TILE;
    const long long n_tiles = tiles_per_batch * p.n_batch;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const size_t plane_stride = (size_t)6 * p.H * p.W * TT_C;
#pragma nounroll
    for (long long tile = wave0; tile < n_tiles; tile += n_waves) {
        const int b = (int)(tile / tiles_per_batch);
        const long long n = (tile - (long long)b * tiles_per_batch) * TT_TILE + i;
        const bool valid = n < p.n_points;
        const long long idx = (long long)b * p.n_points + (valid ? n : 0);
        DecodeCfg dc;
        dc.planes = p.packed;
        dc.tex0 = (unsigned)((size_t)(b / p.views_per_prompt) * (plane_stride / TT_C));
        dc.T = T;
        dc.H = p.H;
        dc.W = p.W;
        dc.radius = p.radius;
        dc.ju = 0.5f * p.W / p.radius;
        dc.jv = 0.5f * p.H / p.radius;
        dc.dbg = 0;
        const float px = p.points[idx * 3 + 0], py = p.points[idx * 3 + 1], pz = p.points[idx * 3 + 2];
        float s0, gq[3], c[3];
        decode_fwd<NEED_N, NEED_TEX, PREC>(L, dc, px, py, pz, valid, i, hi, s0, gq, c);
        float nrm;
        const float sdf = s0 + sphere_bias(px, py, pz, p.bias_radius, nrm);
        if (valid && hi == 0) {
            if (p.out_sdf) p.out_sdf[idx] = sdf;
            if (NEED_N && p.out_grad) {
                p.out_grad[idx * 3 + 0] = gq[0] + px / nrm;
                p.out_grad[idx * 3 + 1] = gq[1] + py / nrm;
                p.out_grad[idx * 3 + 2] = gq[2] + pz / nrm;
            }
            if (NEED_TEX && p.out_feat) {
                p.out_feat[idx * 3 + 0] = c[0];
                p.out_feat[idx * 3 + 1] = c[1];
                p.out_feat[idx * 3 + 2] = c[2];
            }
        }
    }
}

// =====================================================================================================
// tt_query_field: sdf + deformation head on the geometry planes (forward_field, few_step...:375-394)
// =====================================================================================================
#define OFF_D1 LDS_GEO_FLOATS
#define OFF_D2 (OFF_D1 + 64 * W1S)
#define OFF_D3 (OFF_D2 + 64 * W2S)
#define LDS_FIELD_FLOATS (OFF_D3 + 3 * 64)
// third-term images (PREC_S3), appended
#define LO_FW1 LDS_FIELD_FLOATS
#define LO_FW2 (LO_FW1 + LO16_FLOATS(64, 32))
#define LO_D1 (LO_FW2 + LO16_FLOATS(64, 64))
#define LO_D2 (LO_D1 + LO16_FLOATS(64, 32))
#define LDS_FIELD3_FLOATS (LO_D2 + LO16_FLOATS(64, 64))
template <int PREC>
struct FieldWFloats {
    static constexpr int value = PREC == PREC_S3 ? LDS_FIELD3_FLOATS : LDS_FIELD_FLOATS;
};

struct QueryFieldParams {
    const float* packed;
    MlpPtrs w;  // sdf net in w1..w3, deformation net in v1..v3 (32->64->64->3)
    const float* points;
    int n_batch;
    long long n_points;
    int views_per_prompt;
    int H, W;
    float radius, bias_radius;
    float* out_sdf;
    float* out_def;
};

// Two waves per SIMD in every mode: two 4-wave workgroups per CU (81 KB each), or -- PREC_S3, whose third-term images
// bring the weights to 83 KB -- ONE 8-wave workgroup sharing one set of images (136 KB).  (Round 5: the S3 instantiation ran
// one 4-wave workgroup per CU until the kernel-resource table showed its occupancy of 1.)
template <int PREC>
struct QueryFieldWaves {
#ifdef TT_QF_WAVES4  // dev A/B: the one-workgroup-of-four form of PREC_S3
    static constexpr int value = 4;
#else
    static constexpr int value = PREC == PREC_S3 ? 8 : 4;
#endif
};
template <int PREC>
__global__ __launch_bounds__(64 * QueryFieldWaves<PREC>::value, PREC == PREC_S3 ? 1 : 2) void k_query_field(QueryFieldParams p) {
    __shared__ __attribute__((aligned(16))) float L[FieldWFloats<PREC>::value + QueryFieldWaves<PREC>::value * GC_SCRATCH_FLOATS];
    float* T = L + FieldWFloats<PREC>::value + (threadIdx.x >> 6) * GC_SCRATCH_FLOATS;
    {  // split-fp16 images (tt_mfma16.h), same footprint as the fp32 ones
        MlpPtrs w = p.w;
        stage_weights<PREC, 64, 32>(L + OFF_W1, L + LO_FW1, w.w1);
        stage_weights<PREC, 64, 64>(L + OFF_W2, L + LO_FW2, w.w2);
        lds_load_matrix(L + OFF_W3, w.w3, 1, 64, 64);
        stage_weights<PREC, 64, 32>(L + OFF_D1, L + LO_D1, w.v1);
        stage_weights<PREC, 64, 64>(L + OFF_D2, L + LO_D2, w.v2);
        lds_load_matrix(L + OFF_D3, w.v3, 3, 64, 64);
    }
    __syncthreads();
    tt_stagger_priority();
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const long long tiles_per_batch = (p.n_points + TT_TILE - 1) / TT_TILE;
    const long long n_tiles = tiles_per_batch * p.n_batch;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const size_t plane_stride = (size_t)6 * p.H * p.W * TT_C;
#pragma nounroll
    for (long long tile = wave0; tile < n_tiles; tile += n_waves) {
        const int b = (int)(tile / tiles_per_batch);
        const long long n = (tile - (long long)b * tiles_per_batch) * TT_TILE + i;
        const bool valid = n < p.n_points;
        const long long idx = (long long)b * p.n_points + (valid ? n : 0);
        const unsigned tex0 = (unsigned)((size_t)(b / p.views_per_prompt) * (plane_stride / TT_C));
        const float px = p.points[idx * 3 + 0], py = p.points[idx * 3 + 1], pz = p.points[idx * 3 + 2];
        const float X = scale_coord(px, p.radius), Y = scale_coord(py, p.radius), Z = scale_coord(pz, p.radius);
        float f[16], jx[16], jy[16], jz[16];
        const bool any =
            __any(gather_geo_c<false>(p.packed, tex0, p.H, p.W, X, Y, Z, valid, 0.f, 0.f, lane, T, f, jx, jy, jz));
        float s0 = 0.f, d[3] = {0.f, 0.f, 0.f};
        if (any) {  // exact skip otherwise: bias-free MLPs of a zero vector
            float h1[32], h2[32];
            mvx<PREC, 64, 32>(L + OFF_W1, L + LO_FW1, f, h1, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
            mvx<PREC, 64, 64>(L + OFF_W2, L + LO_FW2, h1, h2, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h2[r] = fmaxf(h2[r], 0.f);
            s0 = dot_lds<64>(L + OFF_W3, h2, hi);
            mvx<PREC, 64, 32>(L + OFF_D1, L + LO_D1, f, h1, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
            mvx<PREC, 64, 64>(L + OFF_D2, L + LO_D2, h1, h2, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h2[r] = fmaxf(h2[r], 0.f);
#pragma unroll
            for (int o = 0; o < 3; ++o) d[o] = dot_lds<64>(L + OFF_D3 + 64 * o, h2, hi);
        }
        float nrm;
        const float sdf = s0 + sphere_bias(px, py, pz, p.bias_radius, nrm);
        if (valid && hi == 0) {
            p.out_sdf[idx] = sdf;
            p.out_def[idx * 3 + 0] = d[0];
            p.out_def[idx * 3 + 1] = d[1];
            p.out_def[idx * 3 + 2] = d[2];
        }
    }
}

// =====================================================================================================
#ifndef TT_FWD_REREAD_RAY
#define TT_FWD_REREAD_RAY 0
#endif
// K1: decode every sample of every ray (tiles of 32 adjacent rays x one sample index, chunks of CH indices)
// =====================================================================================================
struct DecodeRaysParams {
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
    float* sdf;
    float* sdf_grad;
    float* features;
};

template <bool NEED_N, bool NEED_TEX, int PREC>
__global__ __launch_bounds__(DECODE_THREADS) void k_decode_rays(DecodeRaysParams p) {
    __shared__ __attribute__((aligned(16))) float L[FwdWFloats<PREC>::value + (DECODE_THREADS / 64) * GC_SCRATCH_FLOATS];
    float* T = L + FwdWFloats<PREC>::value + (threadIdx.x >> 6) * GC_SCRATCH_FLOATS;
    stage_decode_images<NEED_N, NEED_TEX, PREC>(L, p.w);
    __syncthreads();
    tt_stagger_priority();
    const tt_render_cfg& cfg = p.cfg;
    const TileGeom& tg = p.geom;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int S = cfg.n_samples;
    ItemQueue iq = item_queue(p.queue, tg.n_blocks, tg.n_chunks, tg.unit);
    const size_t plane_stride = (size_t)6 * cfg.plane_h * cfg.plane_w * TT_C;
    const TileStats st = tile_stats(cfg.stats);

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
        DecodeCfg dc;
        dc.st = st;
        dc.planes = p.packed;
        dc.tex0 = (unsigned)((size_t)(view / cfg.views_per_prompt) * (plane_stride / TT_C));
        dc.T = T;
        dc.H = cfg.plane_h;
        dc.W = cfg.plane_w;
        dc.radius = cfg.radius;
        dc.ju = 0.5f * cfg.plane_w / cfg.radius;
        dc.jv = 0.5f * cfg.plane_h / cfg.radius;
        dc.dbg = cfg.flags;
#if !TT_FWD_REREAD_RAY
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
#endif
        const int s_end = (ck + 1) * tg.chunk < S ? (ck + 1) * tg.chunk : S;
#if TT_FWD_PREFETCH
        // the sample interval of the NEXT tile step is loaded one step ahead (a step past the chunk reads a clamped, valid
        // address), as the backward kernels do
        auto load_t = [&](int sb0_, float& ts_, float& te_) {
            const int si_ = sb0_ + ks;
            const long long sidx_ = ray * S + (si_ < S ? si_ : S - 1);
            ts_ = p.t_starts[sidx_];
            te_ = p.t_ends[sidx_];
        };
        float ts_n, te_n;
        load_t(ck * tg.chunk, ts_n, te_n);
#endif
#pragma nounroll
        for (int sb0 = ck * tg.chunk; sb0 < s_end; sb0 += tg.sb) {
            const int si = sb0 + ks;
            // validity as a product of 0/1 factors tested ONCE (no select on a lane mask the scalar ALU has just combined:
            // tools/mask_hazard_lint.py)
            const bool rvalid = ray_okf * (si < s_end ? 1.f : 0.f) != 0.f;
            tile_stat(st, TT_STAT_VISITED);
            const long long sidx = ray * S + (si < S ? si : S - 1);
#if TT_FWD_PREFETCH
            const float ts = ts_n, te = te_n;
            load_t(sb0 + tg.sb, ts_n, te_n);
#else
            const float ts = p.t_starts[sidx], te = p.t_ends[sidx];
#endif
            float tm, px, py, pz;
#if TT_FWD_REREAD_RAY
            // (dev A/B, off by default: the ray re-read per tile step and the position rebuilt after the decode)
            {
                long long rr = ray;
                asm volatile("" : "+v"(rr));
                sample_position(p.rays_o[rr * 3 + 0], p.rays_o[rr * 3 + 1], p.rays_o[rr * 3 + 2], p.rays_d[rr * 3 + 0],
                                p.rays_d[rr * 3 + 1], p.rays_d[rr * 3 + 2], ts, te, tm, px, py, pz);
            }
#else
            sample_position(ox, oy, oz, dx, dy, dz, ts, te, tm, px, py, pz);
#endif
            float s0, gq[3], c[3];
            decode_fwd<NEED_N, NEED_TEX, PREC>(L, dc, px, py, pz, rvalid, i, hi, s0, gq, c);
            float nrm;
            const float sdf = s0 + sphere_bias(px, py, pz, cfg.sdf_bias_radius, nrm);
            if (rvalid && hi == 0 && !TT_DBG(cfg.flags, TT_DBG_NO_STORE)) {
                p.sdf[sidx] = sdf;
                if (NEED_N) {
                    p.sdf_grad[sidx * 3 + 0] = gq[0] + px / nrm;
                    p.sdf_grad[sidx * 3 + 1] = gq[1] + py / nrm;
                    p.sdf_grad[sidx * 3 + 2] = gq[2] + pz / nrm;
                }
                if (NEED_TEX) {
                    p.features[sidx * 3 + 0] = c[0];
                    p.features[sidx * 3 + 1] = c[1];
                    p.features[sidx * 3 + 2] = c[2];
                }
            }
        }
    }
    tile_stats_flush(st);
}

// =====================================================================================================
// Fused EVAL render: decode + march per ray tile, front to back, with ballot-based early exit
// =====================================================================================================
// In eval mode the renderer returns per-ray outputs only (renderer :532-545 adds the per-sample extras in training
// only), so nothing forces every sample to be decoded: a wave owns 32 ADJACENT RAYS (an 8x4 pixel block; lane <-> ray,
// both half-waves carry the same ray state) and walks them front to back, one sample index per step:
//   geometry decode (+ normal) -> NeuS alpha -> w = alpha T, T *= 1 - alpha          (no scan: a lane IS a ray)
//   texture decode only if ANY ray of the tile has w > eps_w (wave ballot), and only for those lanes
//   stop when EVERY ray of the tile has T < eps_T                                     (wave ballot)
// Induced error: the dropped weights of a ray sum to < eps_T (opacity, rgb; depth x far), the skipped colours to
// < S eps_w.  eps_T = eps_w = 0 disables both and reproduces tt_render_fwd's per-ray outputs (parity tests); nothing
// per-sample is written, which alone removes the 0.5 GB of per-sample traffic and the separate march kernel.
struct RenderEvalParams {
    const float* packed;
    MlpPtrs w;
    const float* rays_o;
    const float* rays_d;
    const float* t_starts;
    const float* t_ends;
    tt_render_cfg cfg;
    TileGeom geom;
    int* queue;
    float eps_T, eps_w;
    float* opacity;
    float* depth;
    float* rgb_fg;
    float* z_var;
    float* nacc;
    unsigned long long* stats;  // [0] += tile steps decoded, [1] += tile steps with a texture decode (may be null)
};

template <int PREC>
__global__ __launch_bounds__(DECODE_THREADS) void k_render_eval(RenderEvalParams p) {
    __shared__ __attribute__((aligned(16))) float L[FwdWFloats<PREC>::value + (DECODE_THREADS / 64) * GC_SCRATCH_FLOATS];
    float* T = L + FwdWFloats<PREC>::value + (threadIdx.x >> 6) * GC_SCRATCH_FLOATS;
    stage_decode_images<true, true, PREC>(L, p.w);
    __syncthreads();
    tt_stagger_priority();
    const tt_render_cfg& cfg = p.cfg;
    const TileGeom& tg = p.geom;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int S = cfg.n_samples;
    const float kstd = load_inv_std(cfg.inv_std_dev, cfg.inv_std);
    ItemQueue iq = item_queue(p.queue, tg.n_blocks, 1, tg.unit);
    const size_t plane_stride = (size_t)6 * cfg.plane_h * cfg.plane_w * TT_C;
    unsigned long long n_geo = 0, n_tex = 0;
#pragma nounroll
    for (;;) {
        long long b;
        int ck;
        if (!item_pop(iq, 1, 1, b, ck)) break;
        if (b >= tg.n_blocks) continue;
        bool ray_ok;
        const long long ray = tile_ray(tg, b, i, ray_ok);
        const float ray_okf = tt_opaque(ray_ok ? 1.f : 0.f);  // 0/1 factor (lane-mask hygiene, tt_device.h)
        const int view = (int)(ray / cfg.rays_per_view);
        DecodeCfg dc;
        dc.planes = p.packed;
        dc.tex0 = (unsigned)((size_t)(view / cfg.views_per_prompt) * (plane_stride / TT_C));
        dc.T = T;
        dc.H = cfg.plane_h;
        dc.W = cfg.plane_w;
        dc.radius = cfg.radius;
        dc.ju = 0.5f * cfg.plane_w / cfg.radius;
        dc.jv = 0.5f * cfg.plane_h / cfg.radius;
        dc.dbg = 0;
        // Register budget (two waves per SIMD = 256 registers, the decode peaks at ~250): what must survive a decode is kept
        // small.  (i) The ray's origin / direction are re-read from memory every step (24 B per lane out of L1: the loads a
        // spill would issue anyway, minus the scratch).  (ii) Both half-waves carry the same ray (lane i and i + 32), so the
        // nine accumulators are split between them: half 0 owns opacity, depth, sum w t^2, n.x, n.y, half 1 owns r, g, b,
        // n.z -- five registers (acc[k] += w * (hi ? B_k : A_k)) instead of nine; T is needed by both.
        float T = 1.f, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, acc4 = 0.f;
        const float hif = (float)hi, lof = 1.f - hif;
#pragma nounroll
        for (int si = 0; si < S; ++si) {
            float tm, px, py, pz, cosv, ux, uy, uz, sdf, dt;
            {
                long long rr = ray;
                asm volatile("" : "+v"(rr));  // (opaque: the addresses and the loads stay inside the loop)
                const float* ro = p.rays_o + rr * 3;
                const float* rd = p.rays_d + rr * 3;
                const float ts = p.t_starts[rr * S + si], te = p.t_ends[rr * S + si];
                const float ox = ro[0], oy = ro[1], oz = ro[2];
                const float dx = rd[0], dy = rd[1], dz = rd[2];
                sample_position(ox, oy, oz, dx, dy, dz, ts, te, tm, px, py, pz);
            }
            const float X = scale_coord(px, dc.radius), Y = scale_coord(py, dc.radius), Z = scale_coord(pz, dc.radius);
            // live = ray_ok && !(T < eps_T) as a product of 0/1 factors, then ONE compare (eps_T = 0: always live)
            const float livef = ray_okf * (__builtin_fabsf(T) < p.eps_T ? 0.f : 1.f);
            const bool live = livef != 0.f;
            {
                float s0, gq[3];
                decode_geo_fwd<true, PREC>(L, dc, X, Y, Z, live, i, hi, s0, gq);
                ++n_geo;
                float nrm;
                // (the position again from the re-read ray: identical arithmetic, identical bits)
                long long rr = ray;
                asm volatile("" : "+v"(rr));
                const float* ro = p.rays_o + rr * 3;
                const float* rd = p.rays_d + rr * 3;
                const float ts = p.t_starts[rr * S + si], te = p.t_ends[rr * S + si];
                dt = te - ts;
                const float ox = ro[0], oy = ro[1], oz = ro[2];
                const float dx = rd[0], dy = rd[1], dz = rd[2];
                sample_position(ox, oy, oz, dx, dy, dz, ts, te, tm, px, py, pz);
                sdf = s0 + sphere_bias(px, py, pz, cfg.sdf_bias_radius, nrm);
                const float gx = gq[0] + px / nrm, gy = gq[1] + py / nrm, gz = gq[2] + pz / nrm;
                const float gn = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize eps
                const float ign = rcp_(gn);
                ux = gx * ign, uy = gy * ign, uz = gz * ign;
                cosv = dx * ux + dy * uy + dz * uz;
            }
            // (select, not a product: a dead lane's alpha may be NaN -- it decodes nothing)
            float alpha = (cfg.flags & TT_R_VOLSDF) ? volsdf_alpha_terms(sdf, dt, kstd).alpha
                                                    : neus_alpha_terms(sdf, cosv, dt, kstd, cfg.cos_anneal_ratio).alpha;
            if (!live) alpha = 0.f;
            const float wgt = alpha * T;
            T *= 1.f - alpha;
            // half 0: opacity, depth, sum w t^2, n.x, n.y;  half 1: n.z now, the colours after the texture decode
            acc0 = fmaf(wgt, lof, acc0);
            acc1 = fmaf(wgt * lof, tm, acc1);
            acc2 = fmaf(wgt * lof * tm, tm, acc2);
            acc3 = fmaf(wgt, hi ? uz : ux, acc3);
            acc4 = fmaf(wgt * lof, uy, acc4);
            // want_tex = live && wgt > eps_w: wgt = alpha T is exactly 0 on a dead lane (alpha = 0 above) and eps_w >= 0,
            // so the weight test alone decides (eps_w = 0: every sample with a non-zero weight)
            const bool want_tex = __builtin_fabsf(wgt) > p.eps_w;  // (|.|: VolSDF alphas are not clipped, T and the weights can change sign)
            if (__any(want_tex)) {
                float c[3];
                if constexpr (PREC == PREC_F32) {
                    // the fp32-MFMA instantiation needs 11 more registers: the plane coordinates are rebuilt from the re-read
                    // ray as well (same arithmetic, same bits) instead of living across the geometry decode
                    long long rr = ray;
                    asm volatile("" : "+v"(rr));
                    const float* ro = p.rays_o + rr * 3;
                    const float* rd = p.rays_d + rr * 3;
                    float tm2, qx, qy, qz;
                    sample_position(ro[0], ro[1], ro[2], rd[0], rd[1], rd[2], p.t_starts[rr * S + si], p.t_ends[rr * S + si], tm2,
                                    qx, qy, qz);
                    decode_tex_fwd<PREC>(L, dc, scale_coord(qx, dc.radius), scale_coord(qy, dc.radius),
                                         scale_coord(qz, dc.radius), want_tex, i, hi, c);
                } else {
                    decode_tex_fwd<PREC>(L, dc, X, Y, Z, want_tex, i, hi, c);
                }
                ++n_tex;
                if (want_tex) {  // NoMaterial + sigmoid-mipnerf (no_material.py:41-54, ops.py:118-119)
                    const float wh = wgt * hif;
                    acc0 = fmaf(wh, sigmoid_(c[0]) * 1.002f - 0.001f, acc0);
                    acc1 = fmaf(wh, sigmoid_(c[1]) * 1.002f - 0.001f, acc1);
                    acc2 = fmaf(wh, sigmoid_(c[2]) * 1.002f - 0.001f, acc2);
                }
            }
            if (p.eps_T > 0.f && !__any(ray_okf * (__builtin_fabsf(T) < p.eps_T ? 0.f : 1.f) != 0.f)) break;  // every ray is opaque
        }
        if (ray_ok) {
            if (hi == 0) {
                const float op = acc0, dep = acc1, wt2 = acc2;
                p.opacity[ray] = op;
                p.depth[ray] = dep;
                // z_variance = sum w (t - D)^2 with D = sum w t  (renderer :424-431)  =  sum w t^2 - 2 D^2 + D^2 sum w
                p.z_var[ray] = wt2 - 2.f * dep * dep + dep * dep * op;
                p.nacc[ray * 3 + 0] = acc3;
                p.nacc[ray * 3 + 1] = acc4;
            } else {
                p.rgb_fg[ray * 3 + 0] = acc0;
                p.rgb_fg[ray * 3 + 1] = acc1;
                p.rgb_fg[ray * 3 + 2] = acc2;
                p.nacc[ray * 3 + 2] = acc3;
            }
        }
    }
    if (p.stats && lane == 0) {
        atomicAdd(p.stats + 0, n_geo);
        atomicAdd(p.stats + 1, n_tex);
    }
}

// =====================================================================================================
// host side (C ABI)
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

extern "C" int tt_planes_pack(const float* space_cache, float* packed, int32_t n_prompts, int32_t plane_h,
                              int32_t plane_w, void* stream) {
    if (!space_cache || !packed || n_prompts <= 0 || plane_h <= 0 || plane_w <= 0) return TT_ERR_BAD_ARG;
    if (plane_h != plane_w) return TT_ERR_UNSUPPORTED;
    dim3 grid(plane_h, n_prompts * 6);
    if (plane_w % 4 == 0 && ((uintptr_t)space_cache & 15) == 0) {
        size_t lds = (size_t)TT_C * pack4_row_stride(plane_w) * sizeof(float);
        hipLaunchKernelGGL(k_planes_pack4<false>, grid, dim3(256), lds, (hipStream_t)stream, space_cache, packed,
                           plane_h, plane_w, 1, (size_t)0);
    } else {
        size_t lds = (size_t)TT_C * (plane_w + 1) * sizeof(float);
        hipLaunchKernelGGL(k_planes_pack<false>, grid, dim3(256), lds, (hipStream_t)stream, space_cache, packed,
                           plane_h, plane_w, 1, (size_t)0);
    }
    return tt_check_launch();
}

extern "C" int tt_planes_unpack_grad(const float* grad_packed, float* grad_space_cache, int32_t n_prompts,
                                     int32_t plane_h, int32_t plane_w, int32_t n_copies, void* stream) {
    if (!grad_packed || !grad_space_cache || n_prompts <= 0 || plane_h <= 0 || plane_w <= 0 || n_copies <= 0)
        return TT_ERR_BAD_ARG;
    if (plane_h != plane_w) return TT_ERR_UNSUPPORTED;
    dim3 grid(plane_h, n_prompts * 6);
    const size_t copy_stride = (size_t)n_prompts * 6 * plane_h * plane_w * TT_C;
    if (plane_w % 4 == 0 && ((uintptr_t)grad_space_cache & 15) == 0) {
        size_t lds = (size_t)TT_C * pack4_row_stride(plane_w) * sizeof(float);
        hipLaunchKernelGGL(k_planes_pack4<true>, grid, dim3(256), lds, (hipStream_t)stream, grad_packed,
                           grad_space_cache, plane_h, plane_w, n_copies, copy_stride);
    } else {
        size_t lds = (size_t)TT_C * (plane_w + 1) * sizeof(float);
        hipLaunchKernelGGL(k_planes_pack<true>, grid, dim3(256), lds, (hipStream_t)stream, grad_packed,
                           grad_space_cache, plane_h, plane_w, n_copies, copy_stride);
    }
    return tt_check_launch();
}

extern "C" int tt_query_points(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                               int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                               int32_t plane_w, float radius, float sdf_bias_radius, int32_t flags, float* out_sdf,
                               float* out_sdf_grad, float* out_features, void* stream) {
    if (!packed || !w || !points || n_batch <= 0 || n_points <= 0 || n_prompts <= 0 || views_per_prompt <= 0)
        return TT_ERR_BAD_ARG;
    if (n_batch != n_prompts * views_per_prompt || !(radius > 0.f) || !tt_qflags_ok(flags)) return TT_ERR_BAD_ARG;
    if (plane_h != plane_w || plane_h <= 0) return TT_ERR_UNSUPPORTED;
    const bool need_n = (flags & TT_Q_NORMAL) != 0, need_t = (flags & TT_Q_TEX) != 0;
    if (!w->w1 || !w->w2 || !w->w3 || (need_t && (!w->v1 || !w->v2 || !w->v3))) return TT_ERR_BAD_ARG;
    QueryParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.points = points;
    p.n_batch = n_batch;
    p.n_points = n_points;
    p.views_per_prompt = views_per_prompt;
    p.H = plane_h;
    p.W = plane_w;
    p.radius = radius;
    p.bias_radius = sdf_bias_radius;
    p.out_sdf = out_sdf;
    p.out_grad = out_sdf_grad;
    p.out_feat = out_features;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    long long n_tiles = ((n_points + TT_TILE - 1) / TT_TILE) * n_batch;
    long long blocks = (n_tiles + 7) / 8;
    if (blocks > cus) blocks = cus;
    dim3 grid((unsigned)blocks), block(DECODE_THREADS);
    hipStream_t s = (hipStream_t)stream;
    const int prec = tt_prec_of_q(flags);
#define LAUNCH_QP(N, T)                                                                        \
    do {                                                                                       \
        if (prec == PREC_F32)                                                                  \
            hipLaunchKernelGGL((k_query_points<N, T, PREC_F32>), grid, block, 0, s, p);        \
        else if (prec == PREC_S3)                                                              \
            hipLaunchKernelGGL((k_query_points<N, T, PREC_S3>), grid, block, 0, s, p);         \
        else                                                                                   \
            hipLaunchKernelGGL((k_query_points<N, T, PREC_S2>), grid, block, 0, s, p);         \
    } while (0)
    if (need_n && need_t)
        LAUNCH_QP(true, true);
    else if (need_n)
        LAUNCH_QP(true, false);
    else if (need_t)
        LAUNCH_QP(false, true);
    else
        LAUNCH_QP(false, false);
#undef LAUNCH_QP
    return tt_check_launch();
}

// the backward gathers address texels with a 32-bit byte offset from the packed buffer
bool tt_planes_too_large(long long n_prompts, int plane_h, int plane_w) {
    return n_prompts * 6 * plane_h * plane_w * TT_C * 4 >= (1LL << 32);
}

int tt_validate_cfg(const tt_render_cfg* cfg) {
    if (!cfg) return TT_ERR_BAD_ARG;
    if (cfg->n_prompts <= 0 || cfg->views_per_prompt <= 0 || cfg->rays_per_view <= 0 || cfg->n_samples <= 0 ||
        cfg->n_rays <= 0)
        return TT_ERR_BAD_ARG;
    if (cfg->n_rays != (int64_t)cfg->n_prompts * cfg->views_per_prompt * cfg->rays_per_view) return TT_ERR_BAD_ARG;
    if (cfg->plane_h <= 0 || cfg->plane_h != cfg->plane_w) return TT_ERR_UNSUPPORTED;
    if (tt_planes_too_large(cfg->n_prompts, cfg->plane_h, cfg->plane_w)) return TT_ERR_UNSUPPORTED;
    if (!(cfg->radius > 0.f) || (!cfg->inv_std_dev && !(cfg->inv_std > 0.f))) return TT_ERR_BAD_ARG;
    if (cfg->flags < 0) return TT_ERR_BAD_ARG;  // (the kernels use `flags >= 0` as an always-true opaque condition)
    {  // at most one precision mode
        const int pbits = cfg->flags & (TT_R_EXACT_F32 | TT_R_SPLIT2 | TT_R_SPLIT3);
        if (pbits & (pbits - 1)) return TT_ERR_BAD_ARG;
    }
    if (cfg->flags & TT_R_BWD_PAIR) return TT_ERR_UNSUPPORTED;  // reserved (the wave-pair kernel left the tree in round 6)
#ifndef TT_TUNING
    if (cfg->flags & TT_R_WGRAD_F32) return TT_ERR_UNSUPPORTED;  // dev A/B kernel: tuning build only
#endif
    if (!(cfg->skip_eps_tex >= 0.f) || !(cfg->skip_eps_geo >= 0.f)) return TT_ERR_BAD_ARG;
    return TT_OK;
}

// Fills the tile geometry and picks the chunk length: enough items (>= 8 per wave slot) for balance, chunks as
// long as possible so consecutive depths of the same rays reuse L1/L2-resident texels.
long long tt_make_geom(const tt_render_cfg* cfg, long long wave_slots, TileGeom* g, int default_order,
                       int steps_per_item, int min_items_per_slot) {
    g->n_rays = cfg->n_rays;
    g->rays_per_view = cfg->rays_per_view;
    g->n_samples = cfg->n_samples;
    g->image_w = 0;
    g->image_h = 0;
    g->bpr = g->bpv = 0;
    int sb = cfg->tile_sb;
#ifdef TT_TUNING
    if (const char* e = getenv("TT_SB")) sb = atoi(e);
#endif
    // 0 = auto: 4x4 pixels x 2 samples (measured at the bench shapes: backward 10.1 ms vs 10.5 at sb = 1, 11.8 at 4)
    if (!(sb == 1 || sb == 2 || sb == 4 || sb == 8 || sb == 16 || sb == 32)) sb = 2;
    while (sb > 1 && sb > cfg->n_samples) sb /= 2;
    g->sb = sb;
    static const int BW[6] = {8, 4, 4, 2, 2, 1}, BH[6] = {4, 4, 2, 2, 1, 1};
    int l2 = 0;
    while ((1 << l2) < sb) ++l2;
    g->bw = BW[l2];
    g->bh = BH[l2];
    const int rb = 32 / sb;
    long long n_blocks;
    if (cfg->image_w > 0 && cfg->rays_per_view % cfg->image_w == 0) {
        g->image_w = cfg->image_w;
        g->image_h = cfg->rays_per_view / cfg->image_w;
        g->bpr = (g->image_w + g->bw - 1) / g->bw;
        g->bpv = g->bpr * ((g->image_h + g->bh - 1) / g->bh);
        n_blocks = (cfg->n_rays / cfg->rays_per_view) * g->bpv;
    } else {
        n_blocks = (cfg->n_rays + rb - 1) / rb;
    }
    const int n_steps = (cfg->n_samples + sb - 1) / sb;
    // ~6 (geometry backward) / ~12 (forward, texture backward) tile steps per item -- the measured sweet spots of the
    // dynamic queue: per-item ray setup + one pop amortised, items fine enough that the last ones finish together -- but
    // at least 8 items per wave slot
    int n_chunks = (n_steps + steps_per_item - 1) / steps_per_item;
    const int min_chunks = (int)(((long long)min_items_per_slot * wave_slots + n_blocks - 1) / n_blocks);
    if (n_chunks < min_chunks) n_chunks = min_chunks;
    if (n_chunks < 1) n_chunks = 1;
    if (n_chunks > n_steps) n_chunks = n_steps;
    if (cfg->tile_chunk > 0) {  // explicit samples per work item
        n_chunks = (cfg->n_samples + cfg->tile_chunk - 1) / cfg->tile_chunk;
        if (n_chunks > n_steps) n_chunks = n_steps;
    }
#ifdef TT_TUNING
    if (const char* e = getenv("TT_CHUNK")) {
        int c = atoi(e);
        if (c > 0) n_chunks = (cfg->n_samples + c - 1) / c;
        if (n_chunks > n_steps) n_chunks = n_steps;
    }
#endif
    const int steps_per_chunk = (n_steps + n_chunks - 1) / n_chunks;
    g->chunk = steps_per_chunk * sb;  // always a multiple of sb
    g->n_chunks = (cfg->n_samples + g->chunk - 1) / g->chunk;
    g->n_blocks = n_blocks;
    g->unit = n_blocks / 256;  // >= 32 deal rounds: ragged-round imbalance <= 3 %
    if (g->unit > 32) g->unit = 32;
#ifdef TT_TUNING
    if (const char* e = getenv("TT_UNIT")) {
        long long u = atoll(e);
        if (u > 0) g->unit = u;
    }
#endif
    if (g->unit < 1) g->unit = 1;
    g->order = default_order;
#ifdef TT_TUNING
    if (const char* e = getenv("TT_ORDER")) g->order = atoi(e) ? 1 : 0;
#endif
    return n_blocks * g->n_chunks;
}

int tt_launch_march_fwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                        const float* sdf, const float* sdf_grad, const float* features, float* opacity, float* depth,
                        float* rgb_fg, float* z_variance, float* normal_acc, float* weights, float* trans,
                        hipStream_t stream);

extern "C" int tt_render_fwd(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                             const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, float* opacity,
                             float* depth, float* rgb_fg, float* z_variance, float* normal_acc, float* weights,
                             float* trans, float* sdf, float* sdf_grad, float* features, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !opacity || !depth || !rgb_fg || !z_variance ||
        !normal_acc || !weights || !trans || !sdf || !sdf_grad || !features)
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !w->v1 || !w->v2 || !w->v3) return TT_ERR_BAD_ARG;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    DecodeRaysParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
#ifdef TT_TUNING
    if (const char* e = getenv("TT_DEBUG_FLAGS")) p.cfg.flags |= (int)strtol(e, nullptr, 0);
#endif
    p.sdf = sdf;
    p.sdf_grad = sdf_grad;
    p.features = features;
    const long long slots = (long long)cus * (DECODE_THREADS / 64);  // one 8-wave workgroup per CU (LDS 93 KB)
    p.n_items = tt_make_geom(cfg, slots, &p.geom, 1, 12);
    long long blocks = cus;
    long long need = (p.n_items + 7) / 8;
    if (blocks > need) blocks = need;
    blocks = (blocks + 7) / 8 * 8;
    hipStream_t s = (hipStream_t)stream;
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    p.queue = tt_queue_counters(s);
    if (!p.queue) return TT_ERR_DEVICE;
    const int prec = tt_prec_of_r(cfg->flags);
    if (prec == PREC_F32)
        hipLaunchKernelGGL((k_decode_rays<true, true, PREC_F32>), dim3((unsigned)blocks), dim3(DECODE_THREADS), 0, s, p);
    else if (prec == PREC_S3)
        hipLaunchKernelGGL((k_decode_rays<true, true, PREC_S3>), dim3((unsigned)blocks), dim3(DECODE_THREADS), 0, s, p);
    else
        hipLaunchKernelGGL((k_decode_rays<true, true, PREC_S2>), dim3((unsigned)blocks), dim3(DECODE_THREADS), 0, s, p);
    st = tt_check_launch();
    if (st != TT_OK) return st;
    return tt_launch_march_fwd(rays_d, t_starts, t_ends, cfg, sdf, sdf_grad, features, opacity, depth, rgb_fg,
                               z_variance, normal_acc, weights, trans, s);
}

// Decode only (no march): sdf [+ sdf_grad] [+ features] at the mid-points of the given intervals.  Used by the
// importance sampler's proposal pass (sdf head only: the reference evaluates and discards the texture path there,
// few_step...:291,299-306) and by eval-style queries along rays.
extern "C" int tt_decode_rays(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                              const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, int32_t flags,
                              float* sdf, float* sdf_grad, float* features, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    const bool need_n = (flags & TT_Q_NORMAL) != 0, need_t = (flags & TT_Q_TEX) != 0;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !sdf || (need_n && !sdf_grad) ||
        (need_t && !features) || !tt_qflags_ok(flags))
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || (need_t && (!w->v1 || !w->v2 || !w->v3))) return TT_ERR_BAD_ARG;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    DecodeRaysParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.sdf = sdf;
    p.sdf_grad = sdf_grad;
    p.features = features;
    const long long slots = (long long)cus * (DECODE_THREADS / 64);
    // The sdf-only decode (the sampler's proposal pass) is so cheap per tile step that an item's ray set-up, its queue pop and
    // its first-touch texel misses show: items of TT_DR_GEO_SAMPLES (64) samples and no minimum number of items per wave slot
    // (round 5 sweep, tools/time_proposal.py / time_training_shapes.py: 256 x 256 x 128 0.837 -> 0.815 ms, the two launches of
    // a PatchRenderer step at the training shape 0.80 -> 0.63 ms; whole-ray items are better still for large launches and
    // worse for the 800-block patch render).
    if (!need_n && !need_t) {
        const int sb = cfg->tile_sb > 0 ? cfg->tile_sb : 2;
        const int spi = (TT_DR_GEO_SAMPLES + sb - 1) / sb;
        p.n_items = tt_make_geom(cfg, slots, &p.geom, 1, spi > 1 ? spi : 1, TT_DR_GEO_MIN_ITEMS);
    } else {
        p.n_items = tt_make_geom(cfg, slots, &p.geom, 1, 12);
    }
    long long blocks = cus;
    long long need = (p.n_items + 7) / 8;
    if (blocks > need) blocks = need;
    blocks = (blocks + 7) / 8 * 8;
    dim3 grid((unsigned)blocks), blk(DECODE_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    p.queue = tt_queue_counters(s);
    if (!p.queue) return TT_ERR_DEVICE;
    // precision: the query flags if they name one, else the render configuration's
    const int prec = (flags & (TT_Q_EXACT_F32 | TT_Q_SPLIT2 | TT_Q_SPLIT3)) ? tt_prec_of_q(flags) : tt_prec_of_r(cfg->flags);
#define LAUNCH_DR(N, T)                                                                  \
    do {                                                                                 \
        if (prec == PREC_F32)                                                            \
            hipLaunchKernelGGL((k_decode_rays<N, T, PREC_F32>), grid, blk, 0, s, p);     \
        else if (prec == PREC_S3)                                                        \
            hipLaunchKernelGGL((k_decode_rays<N, T, PREC_S3>), grid, blk, 0, s, p);      \
        else                                                                             \
            hipLaunchKernelGGL((k_decode_rays<N, T, PREC_S2>), grid, blk, 0, s, p);      \
    } while (0)
    if (need_n && need_t)
        LAUNCH_DR(true, true);
    else if (need_n)
        LAUNCH_DR(true, false);
    else if (need_t)
        LAUNCH_DR(false, true);
    else
        LAUNCH_DR(false, false);
#undef LAUNCH_DR
    return tt_check_launch();
}

// sdf + deformation on the geometry planes.  `w`: sdf net in w1..w3 and the DEFORMATION net (32->64->64->3,
// few_step...:113-122) in v1..v3.
extern "C" int tt_query_field(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                              int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                              int32_t plane_w, float radius, float sdf_bias_radius, int32_t flags, float* out_sdf,
                              float* out_deformation, void* stream) {
    if (!packed || !w || !points || !out_sdf || !out_deformation || n_batch <= 0 || n_points <= 0 || n_prompts <= 0 ||
        views_per_prompt <= 0)
        return TT_ERR_BAD_ARG;
    if (n_batch != n_prompts * views_per_prompt || !(radius > 0.f) || !tt_qflags_ok(flags)) return TT_ERR_BAD_ARG;
    if (plane_h != plane_w || plane_h <= 0) return TT_ERR_UNSUPPORTED;
    if (!w->w1 || !w->w2 || !w->w3 || !w->v1 || !w->v2 || !w->v3) return TT_ERR_BAD_ARG;
    QueryFieldParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.points = points;
    p.n_batch = n_batch;
    p.n_points = n_points;
    p.views_per_prompt = views_per_prompt;
    p.H = plane_h;
    p.W = plane_w;
    p.radius = radius;
    p.bias_radius = sdf_bias_radius;
    p.out_sdf = out_sdf;
    p.out_def = out_deformation;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    long long n_tiles = ((n_points + TT_TILE - 1) / TT_TILE) * n_batch;
    const int prec = tt_prec_of_q(flags);
    const int waves = prec == PREC_S3 ? QueryFieldWaves<PREC_S3>::value : 4;  // per workgroup; 8 waves per CU either way
    long long blocks = (n_tiles + waves - 1) / waves;
    if (blocks > (8LL / waves) * cus) blocks = (8LL / waves) * cus;
    if (prec == PREC_F32)
        hipLaunchKernelGGL(k_query_field<PREC_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else if (prec == PREC_S3)
        hipLaunchKernelGGL(k_query_field<PREC_S3>, dim3((unsigned)blocks), dim3(64 * waves), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_query_field<PREC_S2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return tt_check_launch();
}


// Eval-mode render: per-ray outputs only, decode and march fused per ray tile, optional early termination.
extern "C" int tt_render_eval(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                              const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                              float transmittance_eps, float weight_eps, float* opacity, float* depth, float* rgb_fg,
                              float* z_variance, float* normal_acc, uint64_t* stats, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !opacity || !depth || !rgb_fg || !z_variance ||
        !normal_acc)
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !w->v1 || !w->v2 || !w->v3) return TT_ERR_BAD_ARG;
    if (!(transmittance_eps >= 0.f) || !(weight_eps >= 0.f)) return TT_ERR_BAD_ARG;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    RenderEvalParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.cfg.tile_sb = 1;                     // a lane is a ray: 8x4 pixel blocks, one sample index per step
    p.cfg.tile_chunk = cfg->n_samples;     // one work item per ray block (the march is sequential in depth)
    p.eps_T = transmittance_eps;
    p.eps_w = weight_eps;
    p.opacity = opacity;
    p.depth = depth;
    p.rgb_fg = rgb_fg;
    p.z_var = z_variance;
    p.nacc = normal_acc;
    p.stats = (unsigned long long*)stats;
    const long long slots = (long long)cus * (DECODE_THREADS / 64);
    const long long n_items = tt_make_geom(&p.cfg, slots, &p.geom, 1);
    if (p.geom.n_chunks != 1 || n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    long long blocks = cus;
    const long long need = (n_items + 7) / 8;
    if (blocks > need) blocks = need;
    blocks = (blocks + 7) / 8 * 8;
    hipStream_t s = (hipStream_t)stream;
    p.queue = tt_queue_counters(s);
    if (!p.queue) return TT_ERR_DEVICE;
    const int prec = tt_prec_of_r(cfg->flags);
    if (prec == PREC_F32)
        hipLaunchKernelGGL(k_render_eval<PREC_F32>, dim3((unsigned)blocks), dim3(DECODE_THREADS), 0, s, p);
    else if (prec == PREC_S3)
        hipLaunchKernelGGL(k_render_eval<PREC_S3>, dim3((unsigned)blocks), dim3(DECODE_THREADS), 0, s, p);
    else
        hipLaunchKernelGGL(k_render_eval<PREC_S2>, dim3((unsigned)blocks), dim3(DECODE_THREADS), 0, s, p);
    return tt_check_launch();
}
