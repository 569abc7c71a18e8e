// tt_points.hip -- gradient of the per-point decode w.r.t. the QUERY POINTS (tt_points_bwd_x).
//
// The reference keeps `points` in the autograd graph of geometry.forward (few_step...:283-286, 329-335): the raster
// renderer decodes positions interpolated from differentiable mesh vertices
// (generative_space_mesh_rasterize_renderer.py:307-331), so d loss / d points flows back through
//   * sdf            : d sdf / dx = J^T q + x / |x|                       (first order: aten grid_sampler_2d_backward's grad_grid)
//   * features       : d feat / dx = Jtex^T ebar                          (same, on the texture planes)
//   * sdf_grad/normal: d (gbar . sdf_grad) / dx                           (second order: K1's `grad_grid` output,
//                      gridsample_cuda.cu:196-208 -- bilinear interpolation has a cross derivative d2f/du dv -- plus the
//                      Hessian of the sphere bias |x|)
// with q = W1^T (m1 . W2^T (m2 . w3)) and ebar = V1^T (n1 . V2^T (n2 . V3^T gfeat)) (masks are constants: ReLU'' = 0).
// Per-point workloads are small (27 k pixels per training call, 300 k vertices at export), so this is ONE simple kernel
// off the hot path: pass 1 gathers f, e and runs the chains on the matrix cores; pass 2 re-gathers the (L1-hot) texels
// and reduces them against q / ebar, so no Jacobian is ever materialised.  No atomics: a point is owned by one lane pair.
#include "tt_device.h"
#include "tt_mfma16.h"
#include "tt_host.h"

#define PX_W1 0
#define PX_W2 (PX_W1 + IMG16_FLOATS(64, 32))
#define PX_W3 (PX_W2 + IMG16_FLOATS(64, 64))
#define PX_V1 (PX_W3 + 64)
#define PX_V2 (PX_V1 + IMG16_FLOATS(64, 96))
#define PX_V3 (PX_V2 + IMG16_FLOATS(64, 64))
#define PX_FLOATS (PX_V3 + 3 * 64) /* (the transposed products read the same images: mv16t) */
// PREC_S3: images of the third terms, appended
#define PXLO_W1 PX_FLOATS
#define PXLO_W2 (PXLO_W1 + LO16_FLOATS(64, 32))
#define PXLO_V1 (PXLO_W2 + LO16_FLOATS(64, 64))
#define PXLO_V2 (PXLO_V1 + LO16_FLOATS(64, 96))
#define PX3_FLOATS (PXLO_V2 + LO16_FLOATS(64, 64))

struct PointsBwdXParams {
    const float* packed;
    MlpPtrs w;
    const float* points;
    int n_batch;
    long long n_points;
    int views_per_prompt;
    int H, W;
    float radius;
    const float* g_sdf;       // (n) or null
    const float* g_sdf_grad;  // (n,3) or null
    const float* g_feat;      // (n,3) or null
    float* grad_points;       // (n,3), overwritten
};

// sum over the 32 channels of this lane pair: this lane holds 16 of them (LIDX(r, hi)), the partner lane ^ 32 the rest
__device__ __forceinline__ float dot16_pair(const f32x4 (&t)[4], const float* v) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(t[q][e], v[4 * q + e], s);
    return s + __shfl_xor(s, 32);
}

// One workgroup per CU holds the weight images; eight waves share them in the split modes (two waves per SIMD: <= 256
// registers each), four in the fp32-MFMA mode (282 registers).
template <int PREC>
struct PointsBwdXWaves {
#ifdef TT_PX_WAVES4  // dev A/B: four waves in every mode (rounds 3-4)
    static constexpr int value = 4;
#else
    static constexpr int value = PREC == PREC_F32 ? 4 : 8;
#endif
};
// The masked vectors a2 / k2 are built from f32x4 LDS loads, so hipcc carries them as <4 x float> values; the transposed
// products then take the register PAIRS (0,2), (1,3) of each group (PAIR_TR, tt_mfma16.h) and the compiler lowers that
// shuffle through a 12-byte stack slot per group -- 48 B of scratch per lane in the split modes (rounds 4-5).  An empty asm
// per element (no instruction) makes them plain scalars again.
template <int N>
__device__ __forceinline__ void scalarize(float (&v)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) asm("" : "+v"(v[r]));
}
struct Axis3 {  // three per-axis accumulators addressed by a compile-time axis
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    template <int A>
    __device__ __forceinline__ float& at() {
        if constexpr (A == 0) return a0;
        else if constexpr (A == 1) return a1;
        else return a2;
    }
    template <int A>
    __device__ __forceinline__ float get() const {
        if constexpr (A == 0) return a0;
        else if constexpr (A == 1) return a1;
        else return a2;
    }
};
template <int PREC>
__global__ __launch_bounds__(64 * PointsBwdXWaves<PREC>::value, 1) void k_points_bwd_x(PointsBwdXParams p) {
    __shared__ __attribute__((aligned(16))) float L[PREC == PREC_S3 ? PX3_FLOATS : PX_FLOATS];
    {
        const MlpPtrs w = p.w;
        stage_weights<PREC, 64, 32>(L + PX_W1, L + PXLO_W1, w.w1);
        stage_weights<PREC, 64, 64>(L + PX_W2, L + PXLO_W2, w.w2);
        lds_load_matrix(L + PX_W3, w.w3, 1, 64, 64);
        if (p.g_feat) {  // (block-uniform: the staging helpers synchronise)
            stage_weights<PREC, 64, 96>(L + PX_V1, L + PXLO_V1, w.v1);
            stage_weights<PREC, 64, 64>(L + PX_V2, L + PXLO_V2, w.v2);
            lds_load_matrix(L + PX_V3, w.v3, 3, 64, 64);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const long long tiles_per_batch = (p.n_points + TT_TILE - 1) / TT_TILE;
    const long long n_tiles = tiles_per_batch * p.n_batch;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int H = p.H, W = p.W;
    const size_t HW = (size_t)H * W, plane_stride = 6 * HW * TT_C;
    const float ju = 0.5f * W / p.radius, jv = 0.5f * H / p.radius;
#pragma nounroll
    for (long long tile = wave0; tile < n_tiles; tile += n_waves) {
        const int b = (int)(tile / tiles_per_batch);
        const long long n = (tile - (long long)b * tiles_per_batch) * TT_TILE + i;
        const bool valid = n < p.n_points;
        const long long idx = (long long)b * p.n_points + (valid ? n : 0);
        const float* pbase = p.packed + (size_t)(b / p.views_per_prompt) * plane_stride;
        const float px = p.points[idx * 3 + 0], py = p.points[idx * 3 + 1], pz = p.points[idx * 3 + 2];
        const float X = scale_coord(px, p.radius), Y = scale_coord(py, p.radius), Z = scale_coord(pz, p.radius);
        const float vf = valid ? 1.f : 0.f;  // (idx is a valid address for every lane; validity as a factor)
        const float gs = p.g_sdf ? p.g_sdf[idx] * vf : 0.f;
        float gg[3], gf[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            gg[o] = p.g_sdf_grad ? p.g_sdf_grad[idx * 3 + o] * vf : 0.f;
            gf[o] = p.g_feat ? p.g_feat[idx * 3 + o] * vf : 0.f;
        }
        const bool need_geo = __any(tt_any_nonzero4(gs, gg[0], gg[1], gg[2]));  // one compare each (tt_device.h)
        const bool need_tex = __any(tt_any_nonzero3(gf[0], gf[1], gf[2]));
        float q[16], eb[48];
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 48; ++r) eb[r] = 0.f;
        // ---- pass 1: decode, reverse chains ----
        if (need_geo) {
            float f[16], jx[16], jy[16], jz[16];
            const bool any = __any(gather_geo<false>(pbase, H, W, X, Y, Z, valid, 0.f, 0.f, hi, f, jx, jy, jz));
            if (any) {  // else f = 0: every mask false, q = 0
                float h1[32], h2[32], a2[32], a1[32];
                mvx<PREC, 64, 32>(L + PX_W1, L + PXLO_W1, f, h1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
                mvx<PREC, 64, 64>(L + PX_W2, L + PXLO_W2, h1, h2, i, hi);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 w3 = *reinterpret_cast<const f32x4*>(L + PX_W3 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) a2[4 * g + e2] = h2[4 * g + e2] > 0.f ? w3[e2] : 0.f;
                }
                scalarize(a2);
                mvtx<PREC, 64, 64, 64>(L + PX_W2, L + PXLO_W2, 0, a2, a1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) a1[r] = h1[r] > 0.f ? a1[r] : 0.f;
                scalarize(a1);
                mvtx<PREC, 32, 64, 32>(L + PX_W1, L + PXLO_W1, 0, a1, q, i, hi);
            }
        }
        if (need_tex) {
            float e[48];
            const bool any = __any(gather_tex(pbase, H, W, X, Y, Z, valid, hi, e));
            if (any) {
                float k1[32], k2[32], kb1[32];
                mvx<PREC, 64, 96>(L + PX_V1, L + PXLO_V1, e, k1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) k1[r] = fmaxf(k1[r], 0.f);
                mvx<PREC, 64, 64>(L + PX_V2, L + PXLO_V2, k1, k2, i, hi);
#pragma unroll
                for (int g = 0; g < 8; ++g) {  // k2bar = n2 . (V3^T gfeat)
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(L + PX_V3 + 0 * 64 + 8 * g + 4 * hi);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(L + PX_V3 + 1 * 64 + 8 * g + 4 * hi);
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(L + PX_V3 + 2 * 64 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const float t = fmaf(v0[e2], gf[0], fmaf(v1[e2], gf[1], v2[e2] * gf[2]));
                        k2[4 * g + e2] = k2[4 * g + e2] > 0.f ? t : 0.f;
                    }
                }
                scalarize(k2);
                mvtx<PREC, 64, 64, 64>(L + PX_V2, L + PXLO_V2, 0, k2, kb1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) kb1[r] = k1[r] > 0.f ? kb1[r] : 0.f;
                scalarize(kb1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {  // ebar_p = (V1[:, 32p : 32p+32])^T k1bar
                    float ebp[16];
                    mvtx<PREC, 32, 64, 96>(L + PX_V1, L + PXLO_V1, 32 * pl, kb1, ebp, i, hi);
#pragma unroll
                    for (int r = 0; r < 16; ++r) eb[16 * pl + r] = ebp[r];
                }
            }
        }
        // ---- pass 2: re-gather the texels and reduce them against q / ebar ----
        // (three named scalars each, not float[3]: hipcc's SLP pass pairs neighbouring elements of a 3-array into <2 x float>
        // operations and then keeps the array in scratch memory)
        Axis3 gx3, hx3, ex3;  // J^T q (world x, y, z);  d (gbar . J^T q) / dx;  Jtex^T ebar
        const Axis3 gg3 = {gg[0], gg[1], gg[2]};
        // (one body per plane with the plane index a compile-time constant: the early exit of an empty plane would otherwise
        // keep the loop rolled and the axis-indexed accumulators gx3[au] ... in scratch memory -- 48 B per lane until round 6)
        auto plane_pass = [&](auto PLc) {
            constexpr int pl = decltype(PLc)::value;
            Corners c;
            corners_setup(PLANE_U(pl, X, Y, Z), PLANE_V(pl, X, Y, Z), H, W, valid, c);
            if (!__any(c.any)) return;
            constexpr int au = pl == 2 ? 2 : 0, av = pl == 1 ? 2 : 1;  // world axis of the plane's u / v coordinate
            float Gu = 0.f, Gv = 0.f, Sx = 0.f, Eu = 0.f, Ev = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float sign = (k == 0 || k == 3) ? 1.f : -1.f;
                const float ck = ((c.inmask >> k) & 1) ? sign : 0.f;
                if (need_geo) {
                    const f32x4* t = reinterpret_cast<const f32x4*>(pbase + (pl * HW + (size_t)c.off[k]) * TT_C) + hi;
                    const f32x4 v[4] = {t[0], t[2], t[4], t[6]};
                    const float d = dot16_pair(v, q);
                    Gu = fmaf(c.du[k], d, Gu);
                    Gv = fmaf(c.dv[k], d, Gv);
                    Sx = fmaf(ck, d, Sx);
                }
                if (need_tex) {
                    const f32x4* t =
                        reinterpret_cast<const f32x4*>(pbase + ((3 + pl) * HW + (size_t)c.off[k]) * TT_C) + hi;
                    const f32x4 v[4] = {t[0], t[2], t[4], t[6]};
                    const float d = dot16_pair(v, eb + 16 * pl);
                    Eu = fmaf(c.du[k], d, Eu);
                    Ev = fmaf(c.dv[k], d, Ev);
                }
            }
            gx3.at<au>() = fmaf(ju, Gu, gx3.at<au>());
            gx3.at<av>() = fmaf(jv, Gv, gx3.at<av>());
            hx3.at<au>() = fmaf(gg3.get<av>() * (ju * jv), Sx, hx3.at<au>());
            hx3.at<av>() = fmaf(gg3.get<au>() * (ju * jv), Sx, hx3.at<av>());
            ex3.at<au>() = fmaf(ju, Eu, ex3.at<au>());
            ex3.at<av>() = fmaf(jv, Ev, ex3.at<av>());
        };
        plane_pass(std::integral_constant<int, 0>{});
        plane_pass(std::integral_constant<int, 1>{});
        plane_pass(std::integral_constant<int, 2>{});
        // sphere bias |x| - r: gradient x / |x|, Hessian (I - xhat xhat^T) / |x|
        const float nrm = sqrtf((px * px + py * py) + pz * pz);
        const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
        const float xh0 = px * inv, xh1 = py * inv, xh2 = pz * inv;
        const float xg = xh0 * gg3.a0 + xh1 * gg3.a1 + xh2 * gg3.a2;
        if (valid && hi == 0) {
            p.grad_points[idx * 3 + 0] = gs * (gx3.a0 + xh0) + hx3.a0 + (gg3.a0 - xh0 * xg) * inv + ex3.a0;
            p.grad_points[idx * 3 + 1] = gs * (gx3.a1 + xh1) + hx3.a1 + (gg3.a1 - xh1 * xg) * inv + ex3.a1;
            p.grad_points[idx * 3 + 2] = gs * (gx3.a2 + xh2) + hx3.a2 + (gg3.a2 - xh2 * xg) * inv + ex3.a2;
        }
    }
}

extern "C" int tt_points_bwd_x(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                               int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                               int32_t plane_w, float radius, int32_t flags, const float* g_sdf,
                               const float* g_sdf_grad, const float* g_features, float* grad_points, void* stream) {
    if (!packed || !w || !points || !grad_points || n_batch <= 0 || n_points <= 0 || n_prompts <= 0 ||
        views_per_prompt <= 0)
        return TT_ERR_BAD_ARG;
    if (n_batch != n_prompts * views_per_prompt || !(radius > 0.f)) return TT_ERR_BAD_ARG;
    if (plane_h != plane_w || plane_h <= 0) return TT_ERR_UNSUPPORTED;
    if (!w->w1 || !w->w2 || !w->w3 || (g_features && (!w->v1 || !w->v2 || !w->v3))) return TT_ERR_BAD_ARG;
    PointsBwdXParams p;
    p.packed = packed;
    p.w.w1 = w->w1;
    p.w.w2 = w->w2;
    p.w.w3 = w->w3;
    p.w.v1 = w->v1;  // read only when g_features is given
    p.w.v2 = w->v2;
    p.w.v3 = w->v3;
    p.points = points;
    p.n_batch = n_batch;
    p.n_points = n_points;
    p.views_per_prompt = views_per_prompt;
    p.H = plane_h;
    p.W = plane_w;
    p.radius = radius;
    p.g_sdf = g_sdf;
    p.g_sdf_grad = g_sdf_grad;
    p.g_feat = g_features;
    p.grad_points = grad_points;
    const int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    const long long n_tiles = ((n_points + TT_TILE - 1) / TT_TILE) * n_batch;
    if (!tt_qflags_ok(flags)) return TT_ERR_BAD_ARG;
    const int prec = tt_prec_of_q(flags);
    const int waves = prec == PREC_F32 ? PointsBwdXWaves<PREC_F32>::value : PointsBwdXWaves<PREC_S3>::value;
    long long blocks = (n_tiles + waves - 1) / waves;
    if (blocks > cus) blocks = cus;
    if (prec == PREC_F32)
        hipLaunchKernelGGL(k_points_bwd_x<PREC_F32>, dim3((unsigned)blocks), dim3(64 * waves), 0, (hipStream_t)stream, p);
    else if (prec == PREC_S3)
        hipLaunchKernelGGL(k_points_bwd_x<PREC_S3>, dim3((unsigned)blocks), dim3(64 * waves), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_points_bwd_x<PREC_S2>, dim3((unsigned)blocks), dim3(64 * waves), 0, (hipStream_t)stream, p);
    return tt_check_launch();
}
