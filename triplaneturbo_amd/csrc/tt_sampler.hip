// tt_sampler.hip -- sample placement along rays (no grad): the level-0 uniform / stratified intervals and the
// one-level importance estimator of the reference
//   threestudio/models/estimators.py:22-118 (ImportanceEstimator.sampling, _transform_stot "uniform")
//   generative_space_sdf_volume_renderer.py:243-316 (prop_sigma_fn: fixed-step NeuS density, :288-297)
//   nerfacc v0.5.2 render_transmittance_from_density / importance_sampling (un-vendored; contract restated in
//   oracle/cpu_ref.py::importance_sampling, "parity unpinned")
// as two bandwidth-trivial kernels.  The per-point sdf of the proposal pass comes from tt_decode_rays; random
// numbers come from the caller (a U[0,1) tensor), so the placement is a pure function of its inputs.
//
// k_sample_importance: one wave per ray, lane <-> proposal interval.
//   sigma_k dt_k -> exclusive wave scan -> T_k = exp(-sum) -> cdf_k = 1 - T_k (cdf_K = 1), edges + cdf parked in
//   wave-private LDS; each fine edge u_j binary-searches the cdf (searchsorted right) and interpolates; the two
//   sorted edge lists are merged by RANK (position = own index + number of smaller elements of the other list, one
//   binary search each) instead of a sort of K + F + 2 values.
#include "tt_device.h"
#include "tt_host.h"
#include "tt_alpha.h"

#pragma clang fp contract(off)  // replay the torch op order of the contract (s*far + (1-s)*near etc.)

// torch.linspace(0, 1, n + 1)[k] as the ROCm/CUDA kernel computes it (symmetric about the middle)
__device__ __forceinline__ float linspace01(int k, int n) {
    const float step = 1.f / (float)n;
    return (k < (n + 1) / 2) ? step * (float)k : 1.f - step * (float)(n - k);
}

// u of edge j of n + 1 under the two placement conventions (tt_abi.h, tt_sample_placement):
//   TT     : u_j = j / n, end points pinned (jitter handled by the callers: interior edges -+ half a cell at level 0,
//            + U / n clamped at the fine level)
//   CENTER : u_j = (j + 0.5) / (n + 1), or (j + U_j) / (n + 1) when jittered: n + 1 equal cells of [0,1], one edge per
//            cell, nothing pinned to 0 or 1
__device__ __forceinline__ float center_u(int j, int n, const float* jit) {
    return ((float)j + (jit ? *jit : 0.5f)) / (float)(n + 1);
}

__global__ __launch_bounds__(256) void k_sample_uniform(long long n_rays, int n, float near, float far,
                                                        const float* __restrict__ jitter, int placement,
                                                        float* __restrict__ ts, float* __restrict__ te) {
    const long long total = n_rays * (long long)(n + 1);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long ray = idx / (n + 1);
        const int k = (int)(idx - ray * (n + 1));
        float s;
        if (placement == TT_PLACE_CENTER) {
            s = center_u(k, n, jitter ? jitter + idx : nullptr);
        } else {
            s = linspace01(k, n);
            if (jitter && k > 0 && k < n) s = s + (jitter[idx] - 0.5f) / (float)n;  // interior edges only
        }
        const float t = s * far + (1.f - s) * near;
        if (k < n) ts[ray * n + k] = t;
        if (k > 0) te[ray * n + k - 1] = t;
    }
}

struct ImportanceParams {
    const float* ts;   // (n_rays, K) proposal intervals
    const float* te;
    const float* sdf;  // (n_rays, K) at the interval mid-points
    const float* u;    // (n_rays, F + 1) U[0,1) or null (deterministic u_j = j / F)
    long long n_rays;
    int K, F;
    int placement;
    float inv_std, step;
    const float* inv_std_dev;  // device scalar overriding inv_std (clamped to [1e-6, 1e6] like LearnedVariance), or null
    float* out_ts;  // (n_rays, K + F + 1)
    float* out_te;
};

__global__ __launch_bounds__(256) void k_sample_importance(ImportanceParams p) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int K = p.K, F = p.F;
    float* tv = lds + (size_t)wave * (2 * (K + 1) + (F + 1));  // K + 1 proposal edges
    float* cdf = tv + (K + 1);                                 // K + 1
    float* tf = cdf + (K + 1);                                 // F + 1 fine edges
    const long long ray = (long long)blockIdx.x * 4 + wave;
    if (ray >= p.n_rays) return;  // wave-private LDS, no block barrier below

    // ---- proposal density -> transmittance -> cdf ----
    const float kstd = p.inv_std_dev ? fminf(fmaxf(p.inv_std_dev[0], 1.0e-6f), 1.0e6f) : p.inv_std;
    float carry = 0.f;
    for (int base = 0; base < K; base += 64) {
        const int k = base + lane;
        const bool valid = k < K;
        const long long i = ray * K + (valid ? k : 0);
        const float ts = p.ts[i], te = p.te[i], sdf = p.sdf[i];
        float sigma;
        if (p.placement & TT_PLACE_VOLSDF) {  // renderer :286-287
            float E, kc;
            sigma = volsdf_density(sdf, kstd, E, kc);
        } else {  // :288-297
            const float prev = sigmoidf_((sdf + p.step * 0.5f) * kstd);
            const float next = sigmoidf_((sdf - p.step * 0.5f) * kstd);
            sigma = fminf(fmaxf((prev - next + 1e-5f) / (prev + 1e-5f), 0.f), 1.f) / p.step;
        }
        const float sd = valid ? sigma * (te - ts) : 0.f;
        float inc = sd;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        const float before = __shfl_up(inc, 1);
        const float excl = carry + (lane == 0 ? 0.f : before);
        if (valid) {
            tv[k] = ts;
            cdf[k] = 1.f - expf(-excl);
        }
        carry += __shfl(inc, 63);
    }
    if (lane == 0) {
        tv[K] = p.te[ray * K + K - 1];
        cdf[K] = 1.f;  // 1 - [T, 0]
    }

    // ---- fine edges: inverse CDF at u_j ----
    for (int j = lane; j <= F; j += 64) {
        float u;
        if ((p.placement & ~TT_PLACE_VOLSDF) == TT_PLACE_CENTER) {
            u = center_u(j, F, p.u ? p.u + ray * (F + 1) + j : nullptr);
        } else {
            u = linspace01(j, F);
            if (p.u) u = fminf(fmaxf(u + p.u[ray * (F + 1) + j] / (float)F, 0.f), 1.f);
        }
        int lo = 0, hi = K + 1;  // searchsorted(cdf, u, right=True): first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int l = min(max(lo - 1, 0), K), h = min(lo, K);
        const float c_lo = cdf[l], c_hi = cdf[h], t_lo = tv[l], t_hi = tv[h];
        const float denom = c_hi - c_lo;
        float frac = denom > 0.f ? (u - c_lo) / denom : 0.f;
        frac = fminf(fmaxf(frac, 0.f), 1.f);
        tf[j] = t_lo + frac * (t_hi - t_lo);
    }

    // ---- merge by rank: proposal edges before equal fine edges ----
    const int M = K + F + 1;  // intervals out; K + F + 2 edges
    float* o_ts = p.out_ts + ray * (long long)M;
    float* o_te = p.out_te + ray * (long long)M;
    for (int i = lane; i <= K; i += 64) {
        const float a = tv[i];
        int lo = 0, hi = F + 1;  // number of fine edges < a
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tf[mid] < a)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int pos = i + lo;
        if (pos < M) o_ts[pos] = a;
        if (pos > 0) o_te[pos - 1] = a;
    }
    for (int j = lane; j <= F; j += 64) {
        const float b = tf[j];
        int lo = 0, hi = K + 1;  // number of proposal edges <= b
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tv[mid] <= b)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int pos = j + lo;
        if (pos < M) o_ts[pos] = b;
        if (pos > 0) o_te[pos - 1] = b;
    }
}

extern "C" int tt_sample_uniform(int64_t n_rays, int32_t n_samples, float near_plane, float far_plane,
                                 const float* jitter, int32_t placement, float* t_starts, float* t_ends,
                                 void* stream) {
    if (n_rays <= 0 || n_samples <= 0 || !t_starts || !t_ends || !(far_plane > near_plane)) return TT_ERR_BAD_ARG;
    if (placement != TT_PLACE_TT && placement != TT_PLACE_CENTER) return TT_ERR_BAD_ARG;
    const long long total = n_rays * (long long)(n_samples + 1);
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_sample_uniform, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)n_rays,
                       n_samples, near_plane, far_plane, jitter, (int)placement, t_starts, t_ends);
    return tt_check_launch();
}

extern "C" int tt_sample_importance(const float* t_starts, const float* t_ends, const float* sdf, int64_t n_rays,
                                    int32_t n_proposal, int32_t n_fine, float inv_std, const float* inv_std_dev,
                                    float render_step_size, const float* u_jitter, int32_t placement,
                                    float* out_t_starts, float* out_t_ends, void* stream) {
    if (!t_starts || !t_ends || !sdf || !out_t_starts || !out_t_ends || n_rays <= 0 || n_proposal <= 0 || n_fine <= 0)
        return TT_ERR_BAD_ARG;
    const int32_t place = placement & ~TT_PLACE_VOLSDF;
    if (place != TT_PLACE_TT && place != TT_PLACE_CENTER) return TT_ERR_BAD_ARG;
    if ((!inv_std_dev && !(inv_std > 0.f)) || !(render_step_size > 0.f)) return TT_ERR_BAD_ARG;
    const size_t lds = 4u * (2u * (n_proposal + 1) + (n_fine + 1)) * sizeof(float);
    if (lds > 64u * 1024u) return TT_ERR_UNSUPPORTED;
    if ((n_rays + 3) / 4 > 0x7fffffffLL) return TT_ERR_UNSUPPORTED;
    ImportanceParams p;
    p.ts = t_starts;
    p.te = t_ends;
    p.sdf = sdf;
    p.u = u_jitter;
    p.n_rays = n_rays;
    p.K = n_proposal;
    p.F = n_fine;
    p.placement = placement;
    p.inv_std = inv_std;
    p.inv_std_dev = inv_std_dev;
    p.step = render_step_size;
    p.out_ts = out_t_starts;
    p.out_te = out_t_ends;
    hipLaunchKernelGGL(k_sample_importance, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), lds, (hipStream_t)stream, p);
    return tt_check_launch();
}
