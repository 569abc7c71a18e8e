// tt_backward.hip -- backward of the fused render, split by network so each kernel's persistent
// weight-gradient accumulators fit the register file beside its working set:
//
//   k_render_bwd_geo : one wave per ray (needs the reverse scan along the ray).  Recomputes the geometry
//       decode per 32-sample tile, turns the per-ray upstream grads into d/d alpha with a division-free
//       reverse affine scan, back-propagates through NeuS alpha, the normalisation and BOTH the value chain
//       and the input-gradient chain of the sdf net (the reference's second-order path:
//       gridsample_cuda.cu:27-210 + aten grid_sampler_2d_backward + transposed GEMMs), accumulates
//       dW1/dW2 with MFMA outer products (K = the 32 samples of the tile) and scatters d/d planes 0..2.
//   k_render_bwd_tex : purely per-sample (no ray structure): feature net backward, dV1/dV2/dV3,
//       scatter of d/d planes 3..5.
//
// Everything per-sample is RECOMPUTED from the planes; the forward saves only trans / weights / features.
#include "tt_device.h"
#include "tt_host.h"
#include <stdlib.h>

#define TT_DBG_NO_SCATTER 0x100  // profiling-only ablations (set through TT_DEBUG_FLAGS; results are then wrong)
#define TT_DBG_NO_WGRAD 0x200
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define XS 36  // row stride (floats) of the [index][sample] transposition scratch

// ---- LDS transposition helpers (wave-private scratch; DS ops of one wave execute in order) ----------
template <int N>
__device__ __forceinline__ void stage_rows(float* S, const float (&v)[N / 2], int j, int hi) {
#pragma unroll
    for (int r = 0; r < N / 2; ++r) S[LIDX(r, hi) * XS + j] = v[r];
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

template <int NX, int NY>
__device__ __forceinline__ void flush_wgrad(const f32x16 (&acc)[NX / 32][NY / 32], float* __restrict__ dst, int i,
                                            int hi) {
#pragma unroll
    for (int m = 0; m < NX / 32; ++m)
#pragma unroll
        for (int n = 0; n < NY / 32; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(dst + (32 * m + LIDX(r, hi)) * NY + 32 * n + i, acc[m][n][r]);
}

#define ZERO16 \
    { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }

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
    const float* opacity;
    const float* depth;
    const float* trans;
    const float* features;
    const float* g_opacity;
    const float* g_depth;
    const float* g_rgb;
    const float* g_zvar;
    const float* g_nacc;
    const float* g_weights;
    const float* g_sdf;
    const float* g_sdf_grad;
    float* grad_packed;
    MlpGradPtrs grads;
};

#define GEO_SCRATCH_FLOATS (2 * 64 * XS)

__global__ __launch_bounds__(256, 1) void k_render_bwd_geo(BwdGeoParams p) {
    __shared__ __attribute__((aligned(16))) float L[LDS_GEO_FLOATS + 4 * GEO_SCRATCH_FLOATS];
    {
        MlpPtrs w = p.w;
        lds_load_geo_weights(L, w);
    }
    __syncthreads();
    const tt_render_cfg& cfg = p.cfg;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5, wave_in_blk = threadIdx.x >> 6;
    float* Xs = L + LDS_GEO_FLOATS + wave_in_blk * GEO_SCRATCH_FLOATS;
    float* Ys = Xs + 64 * XS;
    const int S = cfg.n_samples;
    const int n_tiles = (S + TT_TILE - 1) / TT_TILE;
    const long long n_rays = cfg.n_rays;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long chunk = (n_rays + 7) / 8;
    const long long lo = xcd * chunk, hiR = (lo + chunk < n_rays) ? lo + chunk : n_rays;
    const int waves_per_xcd = (gridDim.x >> 3) * (blockDim.x >> 6);
    const int wv = slot * (blockDim.x >> 6) + wave_in_blk;
    const int H = cfg.plane_h, W = cfg.plane_w;
    const size_t HW = (size_t)H * W;
    const size_t plane_stride = 6 * HW * TT_C;
    const float ju = 0.5f * W / cfg.radius, jv = 0.5f * H / cfg.radius;
    const float kstd = cfg.inv_std, ratio = cfg.cos_anneal_ratio;

    f32x16 accW1[2][1] = {{ZERO16}, {ZERO16}};
    f32x16 accW2[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};
    float accw3 = 0.f;

#pragma nounroll
    for (long long ray = lo + wv; ray < hiR; ray += waves_per_xcd) {
        const int view = (int)(ray / cfg.rays_per_view);
        const size_t pofs = (size_t)(view / cfg.views_per_prompt) * plane_stride;
        const float* pbase = p.packed + pofs;
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
        const float op = p.opacity[ray], D = p.depth[ray];
        const float b_op = p.g_opacity ? p.g_opacity[ray] : 0.f;
        const float b_d = p.g_depth ? p.g_depth[ray] : 0.f;
        const float b_z = p.g_zvar ? p.g_zvar[ray] : 0.f;
        const float b_r = p.g_rgb ? p.g_rgb[ray * 3 + 0] : 0.f, b_g = p.g_rgb ? p.g_rgb[ray * 3 + 1] : 0.f,
                    b_b = p.g_rgb ? p.g_rgb[ray * 3 + 2] : 0.f;
        const float b_nx = p.g_nacc ? p.g_nacc[ray * 3 + 0] : 0.f, b_ny = p.g_nacc ? p.g_nacc[ray * 3 + 1] : 0.f,
                    b_nz = p.g_nacc ? p.g_nacc[ray * 3 + 2] : 0.f;
        float Rcarry = 0.f;  // R_{i+1} entering from the tiles behind
#pragma nounroll
        for (int tile = n_tiles - 1; tile >= 0; --tile) {
            const int si = tile * TT_TILE + i;
            const bool valid = si < S;
            const long long sidx = ray * S + (valid ? si : 0);
            const float ts = valid ? p.t_starts[sidx] : 0.f, te = valid ? p.t_ends[sidx] : 0.f;
            float tm, px, py, pz;
            sample_position(ox, oy, oz, dx, dy, dz, ts, te, tm, px, py, pz);
            const float X = scale_coord(px, cfg.radius), Y = scale_coord(py, cfg.radius),
                        Z = scale_coord(pz, cfg.radius);
            // ---- recompute the geometry decode ----
            float f[16], jx[16], jy[16], jz[16];
            const bool any = __any(gather_geo<true>(pbase, H, W, X, Y, Z, valid, ju, jv, hi, f, jx, jy, jz));
            float h1[32], h2[32], a2[32], a1[32], q[16];
            float s0 = 0.f, gqx = 0.f, gqy = 0.f, gqz = 0.f;
            if (any) {
                mv_fwd<64, 32>(L + OFF_W1, f, h1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
                mv_fwd<64, 64>(L + OFF_W2, h1, h2, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) h2[r] = fmaxf(h2[r], 0.f);
                s0 = dot_lds<64>(L + OFF_W3, h2, hi);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    f32x4 w3 = *reinterpret_cast<const f32x4*>(L + OFF_W3 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) a2[4 * g + e2] = h2[4 * g + e2] > 0.f ? w3[e2] : 0.f;
                }
                mv_bwd<64, 64>(L + OFF_W2, a2, a1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) a1[r] = h1[r] > 0.f ? a1[r] : 0.f;
                mv_bwd<32, 64>(L + OFF_W1, a1, q, i, hi);
                float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sx = fmaf(q[r], jx[r], sx);
                    sy = fmaf(q[r], jy[r], sy);
                    sz = fmaf(q[r], jz[r], sz);
                }
                gqx = sx + __shfl_xor(sx, 32);
                gqy = sy + __shfl_xor(sy, 32);
                gqz = sz + __shfl_xor(sz, 32);
            }
            // ---- per-sample scalars (same arithmetic as the forward) ----
            float nrm;
            const float sdf = s0 + sphere_bias(px, py, pz, cfg.sdf_bias_radius, nrm);
            const float gx = gqx + px / nrm, gy = gqy + py / nrm, gz = gqz + pz / nrm;
            const float gn_raw = sqrtf(gx * gx + gy * gy + gz * gz);
            const float gn = fmaxf(gn_raw, 1e-12f);
            const float nx = gx / gn, ny = gy / gn, nz = gz / gn;
            const float cosv = dx * nx + dy * ny + dz * nz;
            const float c1 = -cosv * 0.5f + 0.5f, c2 = -cosv;
            const float ic = -(fmaxf(c1, 0.f) * (1.f - ratio) + fmaxf(c2, 0.f) * ratio);
            const float dic_dcos = (c1 > 0.f ? 0.5f * (1.f - ratio) : 0.f) + (c2 > 0.f ? ratio : 0.f);
            const float half = (te - ts) * 0.5f;
            const float sA = sigmoidf_((sdf - ic * half) * kstd), sB = sigmoidf_((sdf + ic * half) * kstd);
            const float den = sA + 1e-5f;
            const float rat = (sA - sB + 1e-5f) / den;
            float alpha = fminf(fmaxf(rat, 0.f), 1.f);
            if (!valid) alpha = 0.f;
            const bool pass = valid && rat >= 0.f && rat <= 1.f;
            const float Ti = valid ? p.trans[sidx] : 0.f;
            const float wgt = alpha * Ti;
            float cr = 0.f, cg = 0.f, cb = 0.f;
            if (valid && p.features) {
                cr = p.features[sidx * 3 + 0];
                cg = p.features[sidx * 3 + 1];
                cb = p.features[sidx * 3 + 2];
            }
            const float rr = sigmoidf_(cr) * 1.002f - 0.001f, rg = sigmoidf_(cg) * 1.002f - 0.001f,
                        rb = sigmoidf_(cb) * 1.002f - 0.001f;
            // ---- dL/dw_i ----
            const float dd = tm - D;
            float V = b_op + b_d * tm + b_z * (dd * dd - 2.f * tm * D * (1.f - op)) + (b_r * rr + b_g * rg + b_b * rb) +
                      (b_nx * nx + b_ny * ny + b_nz * nz);
            if (p.g_weights && valid) V += p.g_weights[sidx];
            if (!valid) V = 0.f;
            // ---- reverse affine scan: R_i = V_i a_i + (1 - a_i) R_{i+1};  dL/d alpha_i = T_i (V_i - R_{i+1}) ----
            float A_ = 1.f - alpha, B_ = V * alpha;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const float Ao = __shfl_down(A_, d, 32), Bo = __shfl_down(B_, d, 32);
                if (i + d < 32) {
                    B_ = fmaf(A_, Bo, B_);
                    A_ *= Ao;
                }
            }
            const float Ri = fmaf(A_, Rcarry, B_);
            float Rnext = __shfl_down(Ri, 1, 32);
            if (i == 31) Rnext = Rcarry;
            Rcarry = __shfl(Ri, 0, 32);
            const float dalpha = Ti * (V - Rnext);
            // ---- alpha -> (sdf, iter_cos) ----
            const float drat = pass ? dalpha : 0.f;
            const float dnum = drat / den, dden = -drat * rat / den;
            const float dA = (dnum + dden) * sA * (1.f - sA) * kstd, dB = (-dnum) * sB * (1.f - sB) * kstd;
            float sbar = dA + dB;
            const float dcos = half * (dB - dA) * dic_dcos;
            // ---- normal -> sdf_grad ----
            const float nbx = wgt * b_nx + dcos * dx, nby = wgt * b_ny + dcos * dy, nbz = wgt * b_nz + dcos * dz;
            float gbx, gby, gbz;
            if (gn_raw > 1e-12f) {
                const float nd = nx * nbx + ny * nby + nz * nbz;
                gbx = (nbx - nx * nd) / gn;
                gby = (nby - ny * nd) / gn;
                gbz = (nbz - nz * nd) / gn;
            } else {
                gbx = nbx / 1e-12f;
                gby = nby / 1e-12f;
                gbz = nbz / 1e-12f;
            }
            if (valid) {
                if (p.g_sdf) sbar += p.g_sdf[sidx];
                if (p.g_sdf_grad) {
                    gbx += p.g_sdf_grad[sidx * 3 + 0];
                    gby += p.g_sdf_grad[sidx * 3 + 1];
                    gbz += p.g_sdf_grad[sidx * 3 + 2];
                }
            } else {
                sbar = gbx = gby = gbz = 0.f;
            }
            // ---- network + plane gradients (exactly zero when the tile has no in-bounds texel) ----
            if (any && __any(sbar != 0.f || gbx != 0.f || gby != 0.f || gbz != 0.f)) {
                float u[16], qb[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    qb[r] = fmaf(jx[r], gbx, fmaf(jy[r], gby, jz[r] * gbz));  // qbar = J gbar
                    u[r] = fmaf(sbar, f[r], qb[r]);
                }
                // dW1 += a1 (sbar f + qbar)^T
                const bool do_wgrad = !(cfg.flags & TT_DBG_NO_WGRAD);
                if (do_wgrad) {
                    stage_rows<64>(Xs, a1, i, hi);
                    stage_rows<32>(Ys, u, i, hi);
                    wgrad<64, 32>(accW1, Xs, Ys, i, hi);
                }
                // a1bar = W1 qbar ; b1bar = m1 . a1bar ; v = sbar h1 + b1bar
                float t1[32];
                mv_fwd<64, 32>(L + OFF_W1, qb, t1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) t1[r] = h1[r] > 0.f ? t1[r] : 0.f;  // b1bar
                float v[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) v[r] = fmaf(sbar, h1[r], t1[r]);
                // dW2 += a2 v^T
                if (do_wgrad) {
                    stage_rows<64>(Xs, a2, i, hi);
                    stage_rows<64>(Ys, v, i, hi);
                    wgrad<64, 64>(accW2, Xs, Ys, i, hi);
                }
                // a2bar = W2 b1bar ; dw3 += sbar h2 + m2 . a2bar
                float t2[32];
                mv_fwd<64, 64>(L + OFF_W2, t1, t2, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) t2[r] = fmaf(sbar, h2[r], h2[r] > 0.f ? t2[r] : 0.f);
                stage_rows<64>(Xs, t2, i, hi);
                accw3 += rowsum32(Xs, lane);
                // ---- scatter d/d geometry planes: texel(p,c)[ch] += q[ch] * coef(p,c) ----
                // stage q as [sample][32] (stride 33), coef / texel offsets as [sample][12]
                float* Qs = Ys;              // 32*33 floats
                float* Cs = Xs;              // 32*12 floats
                int* Os = reinterpret_cast<int*>(Xs + 32 * 12);
#pragma unroll
                for (int r = 0; r < 16; ++r) Qs[i * 33 + LIDX(r, hi)] = q[r];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    Corners c;
                    corners_setup(PLANE_U(pl, X, Y, Z), PLANE_V(pl, X, Y, Z), H, W, valid, c);
                    const float gu = (pl == 2 ? gbz : gbx) * ju, gv = (pl == 1 ? gbz : gby) * jv;
                    if (hi == 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            Cs[i * 12 + pl * 4 + k] = fmaf(c.w[k], sbar, fmaf(c.du[k], gu, c.dv[k] * gv));
                            Os[i * 12 + pl * 4 + k] = (int)(pl * HW) + c.off[k];
                        }
                    }
                }
                float* gp = p.grad_packed + pofs;
                if (!(cfg.flags & TT_DBG_NO_SCATTER)) {
#pragma unroll 2
                    for (int it = 0; it < 16; ++it) {
                        const int s = 2 * it + hi;  // each half-wave scatters one sample per iteration
                        const float qv = Qs[s * 33 + i];
                        float cf[12];
                        int of[12];
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const f32x4 c4 = *reinterpret_cast<const f32x4*>(Cs + s * 12 + 4 * g);
                            const i32x4 o4 = *reinterpret_cast<const i32x4*>(Os + s * 12 + 4 * g);
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2) {
                                cf[4 * g + e2] = c4[e2];
                                of[4 * g + e2] = o4[e2];
                            }
                        }
#pragma unroll
                        for (int cc = 0; cc < 12; ++cc)
                            if (cf[cc] != 0.f) atomicAdd(gp + (size_t)of[cc] * TT_C + i, qv * cf[cc]);
                    }
                }
            }
        }
    }
    // ---- flush the persistent weight-gradient accumulators ----
    flush_wgrad<64, 32>(accW1, p.grads.w1, i, hi);
    flush_wgrad<64, 64>(accW2, p.grads.w2, i, hi);
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
    float* grad_packed;
    MlpGradPtrs grads;
};

#define TEX_W_FLOATS (LDS_W_FLOATS - OFF_V1)
#define TV1 0
#define TV2 (OFF_V2 - OFF_V1)
#define TV3 (OFF_V3 - OFF_V1)
#define TEX_SCRATCH_FLOATS ((64 + 96) * XS)

__global__ __launch_bounds__(256, 1) void k_render_bwd_tex(BwdTexParams p) {
    __shared__ __attribute__((aligned(16))) float Lt[TEX_W_FLOATS + 4 * TEX_SCRATCH_FLOATS];
    {
        MlpPtrs w = p.w;
        lds_load_matrix(Lt + TV1, w.v1, 64, 96, V1S);
        lds_load_matrix(Lt + TV2, w.v2, 64, 64, V2S);
        lds_load_matrix(Lt + TV3, w.v3, 3, 64, 64);
    }
    __syncthreads();
    const tt_render_cfg& cfg = p.cfg;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5, wave_in_blk = threadIdx.x >> 6;
    float* Xs = Lt + TEX_W_FLOATS + wave_in_blk * TEX_SCRATCH_FLOATS;
    float* Ys = Xs + 64 * XS;
    const int S = cfg.n_samples;
    const int H = cfg.plane_h, W = cfg.plane_w;
    const size_t HW = (size_t)H * W;
    const size_t plane_stride = 6 * HW * TT_C;
    const long long N = cfg.n_rays * S;
    const long long n_tiles = (N + TT_TILE - 1) / TT_TILE;
    // XCD-aware: contiguous chunk of tiles per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long chunk = (n_tiles + 7) / 8;
    const long long lo = xcd * chunk, hiT = (lo + chunk < n_tiles) ? lo + chunk : n_tiles;
    const int waves_per_xcd = (gridDim.x >> 3) * (blockDim.x >> 6);
    const int wv = slot * (blockDim.x >> 6) + wave_in_blk;
    const float shrink = cfg.rgb_grad_shrink;

    f32x16 accV1[2][3] = {{ZERO16, ZERO16, ZERO16}, {ZERO16, ZERO16, ZERO16}};
    f32x16 accV2[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};
    float accV3[3] = {0.f, 0.f, 0.f};

#pragma nounroll
    for (long long tile = lo + wv; tile < hiT; tile += waves_per_xcd) {
        const long long n = tile * TT_TILE + i;
        const bool valid = n < N;
        const long long sidx = valid ? n : 0;
        const long long ray = sidx / S;
        const int view = (int)(ray / cfg.rays_per_view);
        const size_t pofs = (size_t)(view / cfg.views_per_prompt) * plane_stride;
        // ---- upstream: cbar_o = shrink * w_i * g_rgb[ray,o] * 1.002 * s(1-s) + g_features ----
        float cb[3];
        {
            const float wgt = valid ? p.weights[sidx] : 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float s = sigmoidf_(p.features[sidx * 3 + o]);
                const float gr = p.g_rgb ? p.g_rgb[ray * 3 + o] : 0.f;
                float v = shrink * wgt * gr * 1.002f * s * (1.f - s);
                if (p.g_features) v += p.g_features[sidx * 3 + o];
                cb[o] = valid ? v : 0.f;
            }
        }
        if (!__any(cb[0] != 0.f || cb[1] != 0.f || cb[2] != 0.f)) continue;  // exact: nothing flows back
        const float ts = p.t_starts[sidx], te = p.t_ends[sidx];
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
        float tm, px, py, pz;
        sample_position(ox, oy, oz, dx, dy, dz, ts, te, tm, px, py, pz);
        const float X = scale_coord(px, cfg.radius), Y = scale_coord(py, cfg.radius), Z = scale_coord(pz, cfg.radius);
        float e[48];
        const bool any = __any(gather_tex(p.packed + pofs, H, W, X, Y, Z, valid, hi, e));
        if (!any) continue;  // exact: e == 0 => k1 = k2 = 0 and every mask is false
        float k1[32], k2[32];
        mv_fwd<64, 96>(Lt + TV1, e, k1, i, hi);
#pragma unroll
        for (int r = 0; r < 32; ++r) k1[r] = fmaxf(k1[r], 0.f);
        mv_fwd<64, 64>(Lt + TV2, k1, k2, i, hi);
#pragma unroll
        for (int r = 0; r < 32; ++r) k2[r] = fmaxf(k2[r], 0.f);
        // ---- dV3[o][idx] += sum_s cbar_o[s] k2[idx][s]  (lane <-> idx through the transposition scratch) ----
        stage_rows<64>(Xs, k2, i, hi);
        if (hi == 0) {
            Ys[0 * 32 + i] = cb[0];
            Ys[1 * 32 + i] = cb[1];
            Ys[2 * 32 + i] = cb[2];
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 kk = *reinterpret_cast<const f32x4*>(Xs + lane * XS + 4 * g);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                f32x4 cc = *reinterpret_cast<const f32x4*>(Ys + o * 32 + 4 * g);
                accV3[o] += (kk[0] * cc[0] + kk[1] * cc[1]) + (kk[2] * cc[2] + kk[3] * cc[3]);
            }
        }
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
        // ---- dV2 += k2bar k1^T ----
        const bool do_wgrad = !(cfg.flags & TT_DBG_NO_WGRAD);
        if (do_wgrad) {
            stage_rows<64>(Xs, k2, i, hi);
            stage_rows<64>(Ys, k1, i, hi);
            wgrad<64, 64>(accV2, Xs, Ys, i, hi);
        }
        // ---- k1bar = n1 . (V2^T k2bar) ----
        float kb1[32];
        mv_bwd<64, 64>(Lt + TV2, k2, kb1, i, hi);
#pragma unroll
        for (int r = 0; r < 32; ++r) kb1[r] = k1[r] > 0.f ? kb1[r] : 0.f;
        // ---- dV1 += k1bar e^T ----
        if (do_wgrad) {
            stage_rows<64>(Xs, kb1, i, hi);
            stage_rows<96>(Ys, e, i, hi);
            wgrad<64, 96>(accV1, Xs, Ys, i, hi);
        }
        // ---- ebar = V1^T k1bar ; scatter texel(3+p, c)[ch] += w_c * ebar[32p + ch] ----
        float eb[48];
        mv_bwd<96, 64>(Lt + TV1, kb1, eb, i, hi);
        float* Es = Ys;  // [sample][96] stride 97
        float* Cs = Xs;  // [sample][12]
        int* Os = reinterpret_cast<int*>(Xs + 32 * 12);
#pragma unroll
        for (int r = 0; r < 48; ++r) Es[i * 97 + LIDX(r, hi)] = eb[r];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            Corners c;
            corners_setup(PLANE_U(pl, X, Y, Z), PLANE_V(pl, X, Y, Z), H, W, valid, c);
            if (hi == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    Cs[i * 12 + pl * 4 + k] = c.w[k];
                    // texel offset relative to grad_packed, prompt included (a tile may straddle two views)
                    Os[i * 12 + pl * 4 + k] = (int)(pofs / TT_C) + (int)((3 + pl) * HW) + c.off[k];
                }
            }
        }
        if (!(cfg.flags & TT_DBG_NO_SCATTER)) {
#pragma unroll 2
            for (int it = 0; it < 16; ++it) {
                const int s = 2 * it + hi;
                float ev[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ev[pl] = Es[s * 97 + pl * 32 + i];
                float cf[12];
                int of[12];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(Cs + s * 12 + 4 * g);
                    const i32x4 o4 = *reinterpret_cast<const i32x4*>(Os + s * 12 + 4 * g);
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        cf[4 * g + e2] = c4[e2];
                        of[4 * g + e2] = o4[e2];
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 12; ++cc)
                    if (cf[cc] != 0.f) atomicAdd(p.grad_packed + (size_t)of[cc] * TT_C + i, ev[cc >> 2] * cf[cc]);
            }
        }
    }
    flush_wgrad<64, 96>(accV1, p.grads.v1, i, hi);
    flush_wgrad<64, 64>(accV2, p.grads.v2, i, hi);
#pragma unroll
    for (int o = 0; o < 3; ++o) atomicAdd(p.grads.v3 + o * 64 + lane, accV3[o]);
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
    const char* e = getenv("TT_DEBUG_FLAGS");  // profiling ablations only
    return e ? (int)strtol(e, nullptr, 0) : 0;
}

static long long persistent_blocks(long long work_items_per_wave_granule) {
    int cus = tt_num_cus();
    if (cus <= 0) return -1;
    long long blocks = cus;  // one 4-wave workgroup per CU (register- and LDS-limited)
    long long need = (work_items_per_wave_granule + 3) / 4;
    if (blocks > need) blocks = need;
    return (blocks + 7) / 8 * 8;
}

extern "C" int tt_render_bwd_geo(const float* packed, const tt_mlp_weights* w, const float* rays_o,
                                 const float* rays_d, const float* t_starts, const float* t_ends,
                                 const tt_render_cfg* cfg, const float* opacity, const float* depth,
                                 const float* trans, const float* features, const float* g_opacity,
                                 const float* g_depth, const float* g_rgb_fg, const float* g_z_variance,
                                 const float* g_normal_acc, const float* g_weights, const float* g_sdf,
                                 const float* g_sdf_grad, float* grad_packed, const tt_mlp_grads* grads,
                                 void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !opacity || !depth || !trans || !features ||
        !grad_packed || !grads)
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !grads->w1 || !grads->w2 || !grads->w3) return TT_ERR_BAD_ARG;
    BwdGeoParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.cfg.flags |= debug_flags();
    p.opacity = opacity;
    p.depth = depth;
    p.trans = trans;
    p.features = features;
    p.g_opacity = g_opacity;
    p.g_depth = g_depth;
    p.g_rgb = g_rgb_fg;
    p.g_zvar = g_z_variance;
    p.g_nacc = g_normal_acc;
    p.g_weights = g_weights;
    p.g_sdf = g_sdf;
    p.g_sdf_grad = g_sdf_grad;
    p.grad_packed = grad_packed;
    p.grads = to_gptrs(grads);
    long long blocks = persistent_blocks(cfg->n_rays);
    if (blocks <= 0) return TT_ERR_DEVICE;
    hipLaunchKernelGGL(k_render_bwd_geo, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
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
    p.grads = to_gptrs(grads);
    const long long n_tiles = (cfg->n_rays * cfg->n_samples + TT_TILE - 1) / TT_TILE;
    long long blocks = persistent_blocks(n_tiles);
    if (blocks <= 0) return TT_ERR_DEVICE;
    hipLaunchKernelGGL(k_render_bwd_tex, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return tt_check_launch();
}
