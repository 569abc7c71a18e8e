// tt_alpha.h -- NeuS alpha of one sample (threestudio/models/renderers/neus_volume_renderer.py:93-117) and the fast
// logistic it is built on; shared by the ray-march kernels (tt_march.hip) and the fused eval render (tt_forward.hip).
#pragma once
#include "tt_device.h"

struct AlphaTerms {
    float alpha, rat, den, sA, sB, half, dic_dcos;
    float pass;
    float prev_sdf, next_sdf;  // the estimated sdf at the ends of the interval (d/d inv_std needs them)
};

__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }  // 1 ulp
// logistic function on the hardware exp2 / rcp (2 + 2 instructions instead of ~25 for 1 / (1 + expf(-x)) with the
// IEEE division): relative error ~|x| 2^-24 from the rounded x log2(e), i.e. <= 1e-6 wherever the result is not
// saturated; overflow of exp2 for x < -88 gives rcp(inf) = 0 like the reference
__device__ __forceinline__ float sigmoid_(float x) { return rcp_(1.f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }

// inv_std of the launch: a device scalar when the caller gave one (tt_render_cfg.inv_std_dev: trainable variance, no host
// read-back), with LearnedVariance.forward's clamp (renderer :34-35); else the host value of the config
__device__ __forceinline__ float load_inv_std(const float* dev, float host) {
    return dev ? fminf(fmaxf(dev[0], 1.0e-6f), 1.0e6f) : host;
}

// neus_volume_renderer.py:98-116 (use_volsdf = False)
__device__ __forceinline__ AlphaTerms neus_alpha_terms(float sdf, float cosv, float dt, float kstd, float ratio) {
    AlphaTerms a;
    const float c1 = -cosv * 0.5f + 0.5f, c2 = -cosv;
    const float ic = -(fmaxf(c1, 0.f) * (1.f - ratio) + fmaxf(c2, 0.f) * ratio);
    a.dic_dcos = (c1 > 0.f ? 0.5f * (1.f - ratio) : 0.f) + (c2 > 0.f ? ratio : 0.f);
    a.half = dt * 0.5f;
    const float next_sdf = sdf + ic * a.half, prev_sdf = sdf - ic * a.half;
    a.prev_sdf = prev_sdf;
    a.next_sdf = next_sdf;
    a.sA = sigmoid_(prev_sdf * kstd);
    a.sB = sigmoid_(next_sdf * kstd);
    a.den = a.sA + 1e-5f;
    a.rat = ((a.sA - a.sB) + 1e-5f) * rcp_(a.den);
    a.alpha = fminf(fmaxf(a.rat, 0.f), 1.f);
    // 1 inside the clip, 0 outside -- a FACTOR made of single compares, not a combined lane mask (tt_device.h, corners_setup)
    a.pass = (a.rat >= 0.f ? 1.f : 0.f) * (a.rat <= 1.f ? 1.f : 0.f);
    return a;
}

