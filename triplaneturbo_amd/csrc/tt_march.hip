// tt_march.hip -- the ray march itself (NeuS alpha -> transmittance scan -> per-ray accumulation) and its
// backward.  Both kernels touch ~10 floats per sample and are bandwidth-bound; the expensive per-sample
// decode (plane gather + MLPs) lives in tt_decode.hip / tt_backward.hip and is decoupled from ray order.
//
// One wave per ray, lane <-> sample (64 samples per pass, coalesced [ray][sample] rows); scans and reductions on DPP
// modifiers, divisions as reciprocal-multiplies, the logistic on the hardware exp2: with library expf, IEEE
// divisions and ds_bpermute shuffles these kernels were instruction-bound (42.7 M VALU instructions per forward launch
// = 80 us of issue) rather than bandwidth-bound.
//   forward : alpha_i (neus_volume_renderer.py:93-117), T_i = prod_{j<i}(1-alpha_j), w_i = alpha_i T_i
//             (nerfacc.render_weight_from_alpha), opacity/depth/rgb/normal sums and z_variance
//             (nerfacc.accumulate_along_rays x5, renderer :414-431,467-472).
//   backward: dL/dw_i from the per-ray upstream grads, division-free reverse affine scan for dL/d alpha_i,
//             then through alpha, the cosine and F.normalize down to (d/d sdf_i, d/d sdf_grad_i), written as
//             one float4 per sample for the decode backward to consume.
#include "tt_device.h"
#include "tt_alpha.h"
#include "tt_host.h"

struct MarchFwdParams {
    const float* rays_d;
    const float* t_starts;
    const float* t_ends;
    const float* sdf;
    const float* sdf_grad;
    const float* features;
    long long n_rays;
    int S;
    float inv_std, ratio;
    int volsdf;                // TT_R_VOLSDF: alpha = |dt| x VolSDF density (tt_alpha.h)
    const float* inv_std_dev;  // device scalar overriding inv_std (trainable variance), or null
    float* opacity;
    float* depth;
    float* rgb_fg;
    float* z_var;
    float* nacc;
    float* weights;
    float* trans;
};

// ---- cross-lane primitives on DPP (VALU data-parallel-primitive modifiers: no LDS traffic, ~1 instruction per step;
// the ds_bpermute shuffles they replace made these bandwidth-bound kernels instruction-bound) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity),
                                                                 __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf,
                                                                 false));
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_BCAST15 0x142  // lane 15 of every row -> the next row (enable rows 1, 3: mask 0xa)
#define DPP_BCAST31 0x143  // lane 31 -> rows 2, 3 (mask 0xc)
#define DPP_WAVE_SHR1 0x138

// sum over the 64 lanes, wave-uniform
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<DPP_ROW_SHR(1), 0xf>(0.f, v);
    v += dpp_f<DPP_ROW_SHR(2), 0xf>(0.f, v);
    v += dpp_f<DPP_ROW_SHR(4), 0xf>(0.f, v);
    v += dpp_f<DPP_ROW_SHR(8), 0xf>(0.f, v);
    v += dpp_f<DPP_BCAST15, 0xa>(0.f, v);
    v += dpp_f<DPP_BCAST31, 0xc>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// exclusive prefix product over the 64 lanes (lane 0 gets 1); total = product of all lanes (wave-uniform)
__device__ __forceinline__ float wave_excl_prod(float v, float& total) {
    v *= dpp_f<DPP_ROW_SHR(1), 0xf>(1.f, v);
    v *= dpp_f<DPP_ROW_SHR(2), 0xf>(1.f, v);
    v *= dpp_f<DPP_ROW_SHR(4), 0xf>(1.f, v);
    v *= dpp_f<DPP_ROW_SHR(8), 0xf>(1.f, v);  // inclusive within each row of 16
    v *= dpp_f<DPP_BCAST15, 0xa>(1.f, v);
    v *= dpp_f<DPP_BCAST31, 0xc>(1.f, v);  // inclusive over the wave
    total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return dpp_f<DPP_WAVE_SHR1, 0xf>(1.f, v);  // shift right by one lane: exclusive
}

// Affine recurrence R_l = A_l R_{l-1} + B_l over the lanes, R_{-1} = carry (the backward maps lane l to the sample
// 63 - l of a pass, so that the suffix recurrence over samples is this prefix recurrence over lanes).  Returns
// R_{l-1} for this lane and updates carry to R_63.  Inclusive scan of the maps by composition
// (A, B)_l o (A, B)_seg = (A_l A_seg, A_l B_seg + B_l), then one lane shift.
__device__ __forceinline__ float wave_affine_prev(float A, float B, float& carry) {
#define AFF_STEP(CTRL, MASK)                                   \
    {                                                          \
        const float As = dpp_f<CTRL, MASK>(1.f, A), Bs = dpp_f<CTRL, MASK>(0.f, B); \
        B = fmaf(A, Bs, B);                                    \
        A *= As;                                               \
    }
    AFF_STEP(DPP_ROW_SHR(1), 0xf)
    AFF_STEP(DPP_ROW_SHR(2), 0xf)
    AFF_STEP(DPP_ROW_SHR(4), 0xf)
    AFF_STEP(DPP_ROW_SHR(8), 0xf)
    AFF_STEP(DPP_BCAST15, 0xa)
    AFF_STEP(DPP_BCAST31, 0xc)
#undef AFF_STEP
    const float R = fmaf(A, carry, B);                       // R_l
    const float prev = dpp_f<DPP_WAVE_SHR1, 0xf>(carry, R);  // R_{l-1}; lane 0 gets the carry
    carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, R), 63));
    return prev;
}


struct FwdIn {
    float ts, te, sdf, gx, gy, gz, f0, f1, f2;
};
__device__ __forceinline__ FwdIn march_load_fwd(const MarchFwdParams& p, long long sidx) {
    FwdIn v;
    v.ts = p.t_starts[sidx];
    v.te = p.t_ends[sidx];
    v.sdf = p.sdf[sidx];
    v.gx = p.sdf_grad[sidx * 3 + 0];
    v.gy = p.sdf_grad[sidx * 3 + 1];
    v.gz = p.sdf_grad[sidx * 3 + 2];
    v.f0 = p.features[sidx * 3 + 0];
    v.f1 = p.features[sidx * 3 + 1];
    v.f2 = p.features[sidx * 3 + 2];
    return v;
}
struct FwdAcc {
    float T, op, d, r, g, b, nx, ny, nz;
};
// one 64-sample pass of a ray: returns this lane's (weight, transmittance, t mid-point)
__device__ __forceinline__ void march_pass_fwd(const MarchFwdParams& p, float kstd, const FwdIn& v, bool valid, int lane,
                                               float dx, float dy, float dz, FwdAcc& a, float& wgt, float& Ti,
                                               float& tm) {
    tm = (v.ts + v.te) / 2.f;
    const float gn = fmaxf(sqrtf(v.gx * v.gx + v.gy * v.gy + v.gz * v.gz), 1e-12f);  // F.normalize eps
    const float ign = rcp_(gn);
    const float nx = v.gx * ign, ny = v.gy * ign, nz = v.gz * ign;
    const float cosv = dx * nx + dy * ny + dz * nz;
    float alpha = p.volsdf ? volsdf_alpha_terms(v.sdf, v.te - v.ts, kstd).alpha
                           : neus_alpha_terms(v.sdf, cosv, v.te - v.ts, kstd, p.ratio).alpha;
    if (!valid) alpha = 0.f;
    float total;
    Ti = a.T * wave_excl_prod(1.f - alpha, total);
    a.T *= total;
    wgt = alpha * Ti;
    // NoMaterial + sigmoid-mipnerf (no_material.py:41-54, ops.py:118-119)
    const float r = sigmoid_(v.f0) * 1.002f - 0.001f, g = sigmoid_(v.f1) * 1.002f - 0.001f,
                b = sigmoid_(v.f2) * 1.002f - 0.001f;
    a.op += wgt;
    a.d = fmaf(wgt, tm, a.d);
    a.r = fmaf(wgt, r, a.r);
    a.g = fmaf(wgt, g, a.g);
    a.b = fmaf(wgt, b, a.b);
    a.nx = fmaf(wgt, nx, a.nx);
    a.ny = fmaf(wgt, ny, a.ny);
    a.nz = fmaf(wgt, nz, a.nz);
}
__device__ __forceinline__ void march_reduce_fwd(FwdAcc& a) {
    a.op = wave_sum(a.op);
    a.d = wave_sum(a.d);
    a.r = wave_sum(a.r);
    a.g = wave_sum(a.g);
    a.b = wave_sum(a.b);
    a.nx = wave_sum(a.nx);
    a.ny = wave_sum(a.ny);
    a.nz = wave_sum(a.nz);
}
__device__ __forceinline__ void march_store_ray(const MarchFwdParams& p, long long ray, const FwdAcc& a, float zv) {
    p.opacity[ray] = a.op;
    p.depth[ray] = a.d;
    p.rgb_fg[ray * 3 + 0] = a.r;
    p.rgb_fg[ray * 3 + 1] = a.g;
    p.rgb_fg[ray * 3 + 2] = a.b;
    p.z_var[ray] = zv;
    p.nacc[ray * 3 + 0] = a.nx;
    p.nacc[ray * 3 + 1] = a.ny;
    p.nacc[ray * 3 + 2] = a.nz;
}

// NP > 0: S <= 64 NP; all passes of the ray are loaded up front (every load of the ray in flight at once, no
// dependent second round trip) and the lane's weights / mid-points stay in registers for z_variance.
// NP == 0: any S, streaming (one pass in flight, z_variance re-reads the weights it just wrote).
template <int NP>
__global__ __launch_bounds__(256) void k_march_fwd(MarchFwdParams p) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int S = p.S;
    const float kstd = load_inv_std(p.inv_std_dev, p.inv_std);
    for (long long ray = wave; ray < p.n_rays; ray += n_waves) {
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
        FwdAcc a = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float zv = 0.f;
        if constexpr (NP > 0) {
            FwdIn in[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int si = 64 * k + lane;
                in[k] = march_load_fwd(p, ray * S + (si < S ? si : 0));
            }
            float w[NP], tmid[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int si = 64 * k + lane;
                float Ti;
                march_pass_fwd(p, kstd, in[k], si < S, lane, dx, dy, dz, a, w[k], Ti, tmid[k]);
                if (si < S) {
                    p.weights[ray * S + si] = w[k];
                    p.trans[ray * S + si] = Ti;
                }
            }
            march_reduce_fwd(a);
            // z_variance = sum w (t - depth)^2 (renderer :424-431)
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const float dd = tmid[k] - a.d;
                zv = fmaf(w[k], dd * dd, zv);  // w == 0 on padding lanes
            }
        } else {
            for (int base = 0; base < S; base += 64) {
                const int si = base + lane;
                const bool valid = si < S;
                const long long sidx = ray * S + (valid ? si : 0);
                const FwdIn v = march_load_fwd(p, sidx);
                float wgt, Ti, tm;
                march_pass_fwd(p, kstd, v, valid, lane, dx, dy, dz, a, wgt, Ti, tm);
                if (valid) {
                    p.weights[sidx] = wgt;
                    p.trans[sidx] = Ti;
                }
            }
            march_reduce_fwd(a);
            for (int si = lane; si < S; si += 64) {  // second pass over this lane's own weights
                const long long sidx = ray * S + si;
                const float tm = (p.t_starts[sidx] + p.t_ends[sidx]) / 2.f;
                const float dd = tm - a.d;
                zv = fmaf(p.weights[sidx], dd * dd, zv);
            }
        }
        zv = wave_sum(zv);
        if (lane == 0) march_store_ray(p, ray, a, zv);
    }
}

struct MarchBwdParams {
    const float* rays_d;
    const float* t_starts;
    const float* t_ends;
    const float* sdf;
    const float* sdf_grad;
    const float* features;
    const float* trans;
    const float* opacity;
    const float* depth;
    const float* g_opacity;
    const float* g_depth;
    const float* g_rgb;
    const float* g_zvar;
    const float* g_nacc;
    const float* g_weights;
    const float* g_sdf;
    const float* g_sdf_grad;
    long long n_rays;
    int S;
    float inv_std, ratio;
    int volsdf;                // TT_R_VOLSDF: alpha = |dt| x VolSDF density (tt_alpha.h)
    const float* inv_std_dev;  // device scalar overriding inv_std (trainable variance), or null
    float* g_inv_std_rays;     // (n_rays) d loss / d inv_std per ray, or null
    float* ws;  // (n_rays*S, 4): d/d sdf, d/d sdf_grad xyz
};

struct BwdIn {
    float ts, te, sdf, gx, gy, gz, f0, f1, f2, trans, gw, gs, ggx, ggy, ggz;
};
__device__ __forceinline__ BwdIn march_load_bwd(const MarchBwdParams& p, long long sidx) {
    BwdIn v;
    v.ts = p.t_starts[sidx];
    v.te = p.t_ends[sidx];
    v.sdf = p.sdf[sidx];
    v.gx = p.sdf_grad[sidx * 3 + 0];
    v.gy = p.sdf_grad[sidx * 3 + 1];
    v.gz = p.sdf_grad[sidx * 3 + 2];
    v.f0 = p.features[sidx * 3 + 0];
    v.f1 = p.features[sidx * 3 + 1];
    v.f2 = p.features[sidx * 3 + 2];
    v.trans = p.trans[sidx];
    v.gw = p.g_weights ? p.g_weights[sidx] : 0.f;
    v.gs = p.g_sdf ? p.g_sdf[sidx] : 0.f;
    v.ggx = p.g_sdf_grad ? p.g_sdf_grad[sidx * 3 + 0] : 0.f;
    v.ggy = p.g_sdf_grad ? p.g_sdf_grad[sidx * 3 + 1] : 0.f;
    v.ggz = p.g_sdf_grad ? p.g_sdf_grad[sidx * 3 + 2] : 0.f;
    return v;
}
struct RayBar {  // per-ray upstream gradients and forward results
    float dx, dy, dz, op, D, b_op, b_d, b_z, b_r, b_g, b_b, b_nx, b_ny, b_nz;
};
// one 64-sample pass (passes run from the far end of the ray to the near end; Rcarry links them)
__device__ __forceinline__ f32x4 march_pass_bwd(const MarchBwdParams& p, float kstd, const BwdIn& v, bool valid, int lane,
                                                const RayBar& rb, float& Rcarry, float& dk) {
    const float tm = (v.ts + v.te) / 2.f;
    const float gn_raw = sqrtf(v.gx * v.gx + v.gy * v.gy + v.gz * v.gz);
    const float gn = fmaxf(gn_raw, 1e-12f);
    const float ign = rcp_(gn);
    const float nx = v.gx * ign, ny = v.gy * ign, nz = v.gz * ign;
    const float cosv = rb.dx * nx + rb.dy * ny + rb.dz * nz;
    const AlphaTerms a = p.volsdf ? volsdf_alpha_terms(v.sdf, v.te - v.ts, kstd)
                                  : neus_alpha_terms(v.sdf, cosv, v.te - v.ts, kstd, p.ratio);
    const float alpha = valid ? a.alpha : 0.f;
    const float Ti = valid ? v.trans : 0.f;
    const float wgt = alpha * Ti;
    const float rr = sigmoid_(v.f0) * 1.002f - 0.001f, rg = sigmoid_(v.f1) * 1.002f - 0.001f,
                rbl = sigmoid_(v.f2) * 1.002f - 0.001f;
    // dL/dw_i (z_variance = sum w (t-D)^2 with D = sum w t)
    const float dd = tm - rb.D;
    float V = rb.b_op + rb.b_d * tm + rb.b_z * (dd * dd - 2.f * tm * rb.D * (1.f - rb.op)) +
              (rb.b_r * rr + rb.b_g * rg + rb.b_b * rbl) + (rb.b_nx * nx + rb.b_ny * ny + rb.b_nz * nz);
    V += v.gw;
    if (!valid) V = 0.f;
    // R_i = V_i a_i + (1 - a_i) R_{i+1};  dL/d alpha_i = T_i (V_i - R_{i+1})
    // (lanes hold the samples of a pass in REVERSE order: the sample after this one sits in lane - 1)
    const float Rnext = wave_affine_prev(1.f - alpha, V * alpha, Rcarry);
    const float dalpha = Ti * (V - Rnext);
    const float drat = dalpha * a.pass * (valid ? 1.f : 0.f);
    float sbar, dcos;
    if (p.volsdf) {  // (wave-uniform) alpha depends on the sdf and inv_std only
        sbar = drat * a.d_sdf;
        dcos = 0.f;
        dk += drat * a.d_k;
    } else {
        const float iden = rcp_(a.den);
        const float dnum = drat * iden, dden = -drat * a.rat * iden;
        const float dargA = (dnum + dden) * a.sA * (1.f - a.sA), dargB = (-dnum) * a.sB * (1.f - a.sB);
        const float dA = dargA * kstd, dB = dargB * kstd;
        // d loss / d inv_std of this sample: both logistic arguments are (estimated sdf) * inv_std  (neus...:108-109)
        dk += dargA * a.prev_sdf + dargB * a.next_sdf;
        sbar = dA + dB;
        dcos = a.half * (dB - dA) * a.dic_dcos;
    }
    const float nbx = wgt * rb.b_nx + dcos * rb.dx, nby = wgt * rb.b_ny + dcos * rb.dy,
                nbz = wgt * rb.b_nz + dcos * rb.dz;
    float gbx, gby, gbz;
    if (gn_raw > 1e-12f) {
        const float nd = nx * nbx + ny * nby + nz * nbz;
        gbx = (nbx - nx * nd) * ign;
        gby = (nby - ny * nd) * ign;
        gbz = (nbz - nz * nd) * ign;
    } else {
        gbx = nbx * 1e12f;
        gby = nby * 1e12f;
        gbz = nbz * 1e12f;
    }
    f32x4 o = {sbar + v.gs, gbx + v.ggx, gby + v.ggy, gbz + v.ggz};
    return o;
}

// NP as in k_march_fwd: NP > 0 loads all passes of the ray before the (reverse) scans, NP == 0 streams.
template <int NP>
__global__ __launch_bounds__(256) void k_march_bwd(MarchBwdParams p) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int S = p.S;
    const float kstd = load_inv_std(p.inv_std_dev, p.inv_std);
    for (long long ray = wave; ray < p.n_rays; ray += n_waves) {
        RayBar rb;
        rb.dx = p.rays_d[ray * 3 + 0];
        rb.dy = p.rays_d[ray * 3 + 1];
        rb.dz = p.rays_d[ray * 3 + 2];
        rb.op = p.opacity[ray];
        rb.D = p.depth[ray];
        rb.b_op = p.g_opacity ? p.g_opacity[ray] : 0.f;
        rb.b_d = p.g_depth ? p.g_depth[ray] : 0.f;
        rb.b_z = p.g_zvar ? p.g_zvar[ray] : 0.f;
        rb.b_r = p.g_rgb ? p.g_rgb[ray * 3 + 0] : 0.f;
        rb.b_g = p.g_rgb ? p.g_rgb[ray * 3 + 1] : 0.f;
        rb.b_b = p.g_rgb ? p.g_rgb[ray * 3 + 2] : 0.f;
        rb.b_nx = p.g_nacc ? p.g_nacc[ray * 3 + 0] : 0.f;
        rb.b_ny = p.g_nacc ? p.g_nacc[ray * 3 + 1] : 0.f;
        rb.b_nz = p.g_nacc ? p.g_nacc[ray * 3 + 2] : 0.f;
        float Rcarry = 0.f, dk = 0.f;
        if constexpr (NP > 0) {
            BwdIn in[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int si = 64 * k + 63 - lane;  // reverse order inside a pass (see march_pass_bwd)
                in[k] = march_load_bwd(p, ray * S + (si < S ? si : 0));
            }
#pragma unroll
            for (int k = NP - 1; k >= 0; --k) {
                const int si = 64 * k + 63 - lane;
                const f32x4 o = march_pass_bwd(p, kstd, in[k], si < S, lane, rb, Rcarry, dk);
                if (si < S) *reinterpret_cast<f32x4*>(p.ws + (ray * S + si) * 4) = o;
            }
        } else {
            for (int base = ((S - 1) / 64) * 64; base >= 0; base -= 64) {
                const int si = base + 63 - lane;
                const bool valid = si < S;
                const long long sidx = ray * S + (valid ? si : 0);
                const BwdIn v = march_load_bwd(p, sidx);
                const f32x4 o = march_pass_bwd(p, kstd, v, valid, lane, rb, Rcarry, dk);
                if (valid) *reinterpret_cast<f32x4*>(p.ws + sidx * 4) = o;
            }
        }
        if (p.g_inv_std_rays) {  // trainable variance: one partial per ray, summed by the caller (deterministic)
            dk = wave_sum(dk);
            if (lane == 0) p.g_inv_std_rays[ray] = dk;
        }
    }
}

// one ray per wave (the dispatcher overlaps the load phase of fresh waves with the scans of resident ones)
static unsigned march_blocks(long long n_rays) {
    long long blocks = (n_rays + 3) / 4;
    const long long cap = 1LL << 22;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

int tt_launch_march_fwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                        const float* sdf, const float* sdf_grad, const float* features, float* opacity, float* depth,
                        float* rgb_fg, float* z_variance, float* normal_acc, float* weights, float* trans,
                        hipStream_t stream) {
    MarchFwdParams p;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.sdf = sdf;
    p.sdf_grad = sdf_grad;
    p.features = features;
    p.n_rays = cfg->n_rays;
    p.S = cfg->n_samples;
    p.inv_std = cfg->inv_std;
    p.inv_std_dev = cfg->inv_std_dev;
    p.volsdf = (cfg->flags & TT_R_VOLSDF) ? 1 : 0;
    p.ratio = cfg->cos_anneal_ratio;
    p.opacity = opacity;
    p.depth = depth;
    p.rgb_fg = rgb_fg;
    p.z_var = z_variance;
    p.nacc = normal_acc;
    p.weights = weights;
    p.trans = trans;
    const dim3 grid(march_blocks(cfg->n_rays)), blk(256);
    if (p.S <= 64)
        hipLaunchKernelGGL(k_march_fwd<1>, grid, blk, 0, stream, p);
    else if (p.S <= 128)
        hipLaunchKernelGGL(k_march_fwd<2>, grid, blk, 0, stream, p);
    else if (p.S <= 256)
        hipLaunchKernelGGL(k_march_fwd<4>, grid, blk, 0, stream, p);
    else
        hipLaunchKernelGGL(k_march_fwd<0>, grid, blk, 0, stream, p);
    return tt_check_launch();
}

int tt_launch_march_bwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                        const float* sdf, const float* sdf_grad, const float* features, const float* trans,
                        const float* opacity, const float* depth, const float* g_opacity, const float* g_depth,
                        const float* g_rgb_fg, const float* g_z_variance, const float* g_normal_acc,
                        const float* g_weights, const float* g_sdf, const float* g_sdf_grad, float* g_inv_std_rays,
                        float* ws, hipStream_t stream) {
    MarchBwdParams p;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.sdf = sdf;
    p.sdf_grad = sdf_grad;
    p.features = features;
    p.trans = trans;
    p.opacity = opacity;
    p.depth = depth;
    p.g_opacity = g_opacity;
    p.g_depth = g_depth;
    p.g_rgb = g_rgb_fg;
    p.g_zvar = g_z_variance;
    p.g_nacc = g_normal_acc;
    p.g_weights = g_weights;
    p.g_sdf = g_sdf;
    p.g_sdf_grad = g_sdf_grad;
    p.n_rays = cfg->n_rays;
    p.S = cfg->n_samples;
    p.inv_std = cfg->inv_std;
    p.inv_std_dev = cfg->inv_std_dev;
    p.volsdf = (cfg->flags & TT_R_VOLSDF) ? 1 : 0;
    p.g_inv_std_rays = g_inv_std_rays;
    p.ratio = cfg->cos_anneal_ratio;
    p.ws = ws;
    const dim3 grid(march_blocks(cfg->n_rays)), blk(256);
    if (p.S <= 64)
        hipLaunchKernelGGL(k_march_bwd<1>, grid, blk, 0, stream, p);
    else if (p.S <= 128)
        hipLaunchKernelGGL(k_march_bwd<2>, grid, blk, 0, stream, p);
    else if (p.S <= 256)
        hipLaunchKernelGGL(k_march_bwd<4>, grid, blk, 0, stream, p);
    else
        hipLaunchKernelGGL(k_march_bwd<0>, grid, blk, 0, stream, p);
    return tt_check_launch();
}

extern "C" int tt_march_fwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                            const float* sdf, const float* sdf_grad, const float* features, float* opacity,
                            float* depth, float* rgb_fg, float* z_variance, float* normal_acc, float* weights,
                            float* trans, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!rays_d || !t_starts || !t_ends || !sdf || !sdf_grad || !features || !opacity || !depth || !rgb_fg ||
        !z_variance || !normal_acc || !weights || !trans)
        return TT_ERR_BAD_ARG;
    return tt_launch_march_fwd(rays_d, t_starts, t_ends, cfg, sdf, sdf_grad, features, opacity, depth, rgb_fg,
                               z_variance, normal_acc, weights, trans, (hipStream_t)stream);
}

extern "C" int tt_march_bwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                            const float* opacity, const float* depth, const float* trans, const float* sdf,
                            const float* sdf_grad, const float* features, const float* g_opacity,
                            const float* g_depth, const float* g_rgb_fg, const float* g_z_variance,
                            const float* g_normal_acc, const float* g_weights, const float* g_sdf,
                            const float* g_sdf_grad, float* g_inv_std_rays, float* out_grad, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!rays_d || !t_starts || !t_ends || !opacity || !depth || !trans || !sdf || !sdf_grad || !features || !out_grad)
        return TT_ERR_BAD_ARG;
    return tt_launch_march_bwd(rays_d, t_starts, t_ends, cfg, sdf, sdf_grad, features, trans, opacity, depth,
                               g_opacity, g_depth, g_rgb_fg, g_z_variance, g_normal_acc, g_weights, g_sdf, g_sdf_grad,
                               g_inv_std_rays, out_grad, (hipStream_t)stream);
}
