// tt_march.hip -- the ray march itself (NeuS alpha -> transmittance scan -> per-ray accumulation) and its
// backward.  Both kernels touch ~10 floats per sample and are bandwidth-bound; the expensive per-sample
// decode (plane gather + MLPs) lives in tt_decode.hip / tt_backward.hip and is decoupled from ray order.
//
// One wave per ray, lane <-> sample (64 samples per pass, coalesced [ray][sample] rows).
//   forward : alpha_i (neus_volume_renderer.py:93-117), T_i = prod_{j<i}(1-alpha_j), w_i = alpha_i T_i
//             (nerfacc.render_weight_from_alpha), opacity/depth/rgb/normal sums and z_variance
//             (nerfacc.accumulate_along_rays x5, renderer :414-431,467-472).
//   backward: dL/dw_i from the per-ray upstream grads, division-free reverse affine scan for dL/d alpha_i,
//             then through alpha, the cosine and F.normalize down to (d/d sdf_i, d/d sdf_grad_i), written as
//             one float4 per sample for the decode backward to consume.
#include "tt_device.h"
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
    float* opacity;
    float* depth;
    float* rgb_fg;
    float* z_var;
    float* nacc;
    float* weights;
    float* trans;
};

struct AlphaTerms {
    float alpha, rat, den, sA, sB, half, dic_dcos;
    bool pass;
};

// neus_volume_renderer.py:98-116 (use_volsdf = False)
__device__ __forceinline__ AlphaTerms neus_alpha_terms(float sdf, float cosv, float dt, float kstd, float ratio) {
    AlphaTerms a;
    const float c1 = -cosv * 0.5f + 0.5f, c2 = -cosv;
    const float ic = -(fmaxf(c1, 0.f) * (1.f - ratio) + fmaxf(c2, 0.f) * ratio);
    a.dic_dcos = (c1 > 0.f ? 0.5f * (1.f - ratio) : 0.f) + (c2 > 0.f ? ratio : 0.f);
    a.half = dt * 0.5f;
    const float next_sdf = sdf + ic * a.half, prev_sdf = sdf - ic * a.half;
    a.sA = sigmoidf_(prev_sdf * kstd);
    a.sB = sigmoidf_(next_sdf * kstd);
    a.den = a.sA + 1e-5f;
    a.rat = ((a.sA - a.sB) + 1e-5f) / a.den;
    a.alpha = fminf(fmaxf(a.rat, 0.f), 1.f);
    a.pass = a.rat >= 0.f && a.rat <= 1.f;
    return a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(256) void k_march_fwd(MarchFwdParams p) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int S = p.S;
    for (long long ray = wave; ray < p.n_rays; ray += n_waves) {
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
        float T = 1.f;
        float a_op = 0.f, a_d = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f, a_nx = 0.f, a_ny = 0.f, a_nz = 0.f;
        for (int base = 0; base < S; base += 64) {
            const int si = base + lane;
            const bool valid = si < S;
            const long long sidx = ray * S + (valid ? si : 0);
            const float ts = p.t_starts[sidx], te = p.t_ends[sidx];
            const float tm = (ts + te) / 2.f;
            const float sdf = p.sdf[sidx];
            const float gx = p.sdf_grad[sidx * 3 + 0], gy = p.sdf_grad[sidx * 3 + 1], gz = p.sdf_grad[sidx * 3 + 2];
            const float gn = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize eps
            const float nx = gx / gn, ny = gy / gn, nz = gz / gn;
            const float cosv = dx * nx + dy * ny + dz * nz;
            float alpha = neus_alpha_terms(sdf, cosv, te - ts, p.inv_std, p.ratio).alpha;
            if (!valid) alpha = 0.f;
            float total;
            const float Ti = T * seg_excl_prod<64>(1.f - alpha, lane, total);
            T *= total;
            const float wgt = alpha * Ti;
            // NoMaterial + sigmoid-mipnerf (no_material.py:41-54, ops.py:118-119)
            const float r = sigmoidf_(p.features[sidx * 3 + 0]) * 1.002f - 0.001f,
                        g = sigmoidf_(p.features[sidx * 3 + 1]) * 1.002f - 0.001f,
                        b = sigmoidf_(p.features[sidx * 3 + 2]) * 1.002f - 0.001f;
            a_op += wgt;
            a_d = fmaf(wgt, tm, a_d);
            a_r = fmaf(wgt, r, a_r);
            a_g = fmaf(wgt, g, a_g);
            a_b = fmaf(wgt, b, a_b);
            a_nx = fmaf(wgt, nx, a_nx);
            a_ny = fmaf(wgt, ny, a_ny);
            a_nz = fmaf(wgt, nz, a_nz);
            if (valid) {
                p.weights[sidx] = wgt;
                p.trans[sidx] = Ti;
            }
        }
        a_op = wave_sum(a_op);
        a_d = wave_sum(a_d);
        a_r = wave_sum(a_r);
        a_g = wave_sum(a_g);
        a_b = wave_sum(a_b);
        a_nx = wave_sum(a_nx);
        a_ny = wave_sum(a_ny);
        a_nz = wave_sum(a_nz);
        // z_variance = sum w (t - depth)^2 (renderer :424-431): second pass over this lane's own weights
        float zv = 0.f;
        for (int si = lane; si < S; si += 64) {
            const long long sidx = ray * S + si;
            const float tm = (p.t_starts[sidx] + p.t_ends[sidx]) / 2.f;
            const float dd = tm - a_d;
            zv = fmaf(p.weights[sidx], dd * dd, zv);
        }
        zv = wave_sum(zv);
        if (lane == 0) {
            p.opacity[ray] = a_op;
            p.depth[ray] = a_d;
            p.rgb_fg[ray * 3 + 0] = a_r;
            p.rgb_fg[ray * 3 + 1] = a_g;
            p.rgb_fg[ray * 3 + 2] = a_b;
            p.z_var[ray] = zv;
            p.nacc[ray * 3 + 0] = a_nx;
            p.nacc[ray * 3 + 1] = a_ny;
            p.nacc[ray * 3 + 2] = a_nz;
        }
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
    float* ws;  // (n_rays*S, 4): d/d sdf, d/d sdf_grad xyz
};

__global__ __launch_bounds__(256) void k_march_bwd(MarchBwdParams p) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int S = p.S;
    const float kstd = p.inv_std;
    for (long long ray = wave; ray < p.n_rays; ray += n_waves) {
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
        const float op = p.opacity[ray], D = p.depth[ray];
        const float b_op = p.g_opacity ? p.g_opacity[ray] : 0.f;
        const float b_d = p.g_depth ? p.g_depth[ray] : 0.f;
        const float b_z = p.g_zvar ? p.g_zvar[ray] : 0.f;
        const float b_r = p.g_rgb ? p.g_rgb[ray * 3 + 0] : 0.f, b_g = p.g_rgb ? p.g_rgb[ray * 3 + 1] : 0.f,
                    b_b = p.g_rgb ? p.g_rgb[ray * 3 + 2] : 0.f;
        const float b_nx = p.g_nacc ? p.g_nacc[ray * 3 + 0] : 0.f, b_ny = p.g_nacc ? p.g_nacc[ray * 3 + 1] : 0.f,
                    b_nz = p.g_nacc ? p.g_nacc[ray * 3 + 2] : 0.f;
        float Rcarry = 0.f;
        for (int base = ((S - 1) / 64) * 64; base >= 0; base -= 64) {
            const int si = base + lane;
            const bool valid = si < S;
            const long long sidx = ray * S + (valid ? si : 0);
            const float ts = p.t_starts[sidx], te = p.t_ends[sidx];
            const float tm = (ts + te) / 2.f;
            const float sdf = p.sdf[sidx];
            const float gx = p.sdf_grad[sidx * 3 + 0], gy = p.sdf_grad[sidx * 3 + 1], gz = p.sdf_grad[sidx * 3 + 2];
            const float gn_raw = sqrtf(gx * gx + gy * gy + gz * gz);
            const float gn = fmaxf(gn_raw, 1e-12f);
            const float nx = gx / gn, ny = gy / gn, nz = gz / gn;
            const float cosv = dx * nx + dy * ny + dz * nz;
            const AlphaTerms a = neus_alpha_terms(sdf, cosv, te - ts, kstd, p.ratio);
            const float alpha = valid ? a.alpha : 0.f;
            const float Ti = valid ? p.trans[sidx] : 0.f;
            const float wgt = alpha * Ti;
            const float rr = sigmoidf_(p.features[sidx * 3 + 0]) * 1.002f - 0.001f,
                        rg = sigmoidf_(p.features[sidx * 3 + 1]) * 1.002f - 0.001f,
                        rb = sigmoidf_(p.features[sidx * 3 + 2]) * 1.002f - 0.001f;
            // dL/dw_i (z_variance = sum w (t-D)^2 with D = sum w t)
            const float dd = tm - D;
            float V = b_op + b_d * tm + b_z * (dd * dd - 2.f * tm * D * (1.f - op)) + (b_r * rr + b_g * rg + b_b * rb) +
                      (b_nx * nx + b_ny * ny + b_nz * nz);
            if (p.g_weights) V += p.g_weights[sidx];
            if (!valid) V = 0.f;
            // R_i = V_i a_i + (1 - a_i) R_{i+1};  dL/d alpha_i = T_i (V_i - R_{i+1})
            const float Rnext = seg_rev_affine<64>(1.f - alpha, V * alpha, lane, Rcarry);
            const float dalpha = Ti * (V - Rnext);
            const float drat = (valid && a.pass) ? dalpha : 0.f;
            const float dnum = drat / a.den, dden = -drat * a.rat / a.den;
            const float dA = (dnum + dden) * a.sA * (1.f - a.sA) * kstd, dB = (-dnum) * a.sB * (1.f - a.sB) * kstd;
            float sbar = dA + dB;
            const float dcos = a.half * (dB - dA) * a.dic_dcos;
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
                f32x4 o = {sbar, gbx, gby, gbz};
                *reinterpret_cast<f32x4*>(p.ws + sidx * 4) = o;
            }
        }
    }
}

static unsigned march_blocks(long long n_rays) {
    int cus = tt_num_cus();
    long long blocks = (n_rays + 3) / 4;
    long long cap = (long long)(cus > 0 ? cus : 256) * 8;
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
    p.ratio = cfg->cos_anneal_ratio;
    p.opacity = opacity;
    p.depth = depth;
    p.rgb_fg = rgb_fg;
    p.z_var = z_variance;
    p.nacc = normal_acc;
    p.weights = weights;
    p.trans = trans;
    hipLaunchKernelGGL(k_march_fwd, dim3(march_blocks(cfg->n_rays)), dim3(256), 0, stream, p);
    return tt_check_launch();
}

int tt_launch_march_bwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                        const float* sdf, const float* sdf_grad, const float* features, const float* trans,
                        const float* opacity, const float* depth, const float* g_opacity, const float* g_depth,
                        const float* g_rgb_fg, const float* g_z_variance, const float* g_normal_acc,
                        const float* g_weights, const float* g_sdf, const float* g_sdf_grad, float* ws,
                        hipStream_t stream) {
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
    p.ratio = cfg->cos_anneal_ratio;
    p.ws = ws;
    hipLaunchKernelGGL(k_march_bwd, dim3(march_blocks(cfg->n_rays)), dim3(256), 0, stream, p);
    return tt_check_launch();
}
