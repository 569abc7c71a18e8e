// tt_alpha.h -- NeuS alpha of one sample (threestudio/models/renderers/neus_volume_renderer.py:93-117) and the fast
// logistic it is built on; shared by the ray-march kernels (tt_march.hip) and the fused eval render (tt_forward.hip).
#pragma once
#include "tt_device.h"

struct AlphaTerms {
    float alpha, rat, den, sA, sB, half, dic_dcos;
    float pass;
    float prev_sdf, next_sdf;  // the estimated sdf at the ends of the interval (d/d inv_std needs them)
    float d_sdf, d_k;          // use_volsdf only: d alpha / d sdf, d alpha / d inv_std
};

// 1 / x for finite, non-zero x: the hardware reciprocal (1 ulp) + one Newton step = within ~0.5 ulp of the IEEE quotient
// (round 5: with fp32-grade MLP products the march's own approximations became the largest difference to the reference's
// arithmetic -- NeuS alpha is a DIFFERENCE of nearly equal logistics, so every ulp of them is amplified; tools/fuzz_seeds.py,
// seeds 5 / 198 / 229 / 391: all three precision modes equally far from the fp32 oracle).  The march kernels are HBM-bound:
// the extra FMAs are free there.
__device__ __forceinline__ float rcp_(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
#ifdef TT_FAST_RCP  // (dev A/B: rounds 1-4)
    return r;
#else
    // (x = 0, +-inf or a denormal: r is inf / 0 and the refinement would be inf * 0 = NaN -- keep the hardware's r, as the
    // reference's IEEE quotient gives inf / 0 there; one NaN test, no lane-mask combination)
    const float nr = fmaf(fmaf(-x, r, 1.f), r, r);
    return nr == nr ? nr : r;
#endif
}
// logistic function 1 / (1 + exp(-x)) on the hardware exp2 + reciprocal, to ~2 ulp: the argument t = -x log2(e) is carried
// as a pair (t_hi = rn(x c_hi), t_lo = the rounding error of that product + x c_lo: ~2^-48 |t|), exp2(t_hi) corrected by
// (1 + ln2 t_lo) -- rounds 1-4 used exp2(rn(x c)) alone, whose relative error grows as |x| 2^-24 (1e-6 at |x| = 20).
// t is clamped at 126 so that 1 + exp2(t) stays finite (x < -87: the result is ~1e-38 instead of the reference's 0 / denormal).
__device__ __forceinline__ float sigmoid_(float x) {
#ifdef TT_FAST_LOGISTIC  // (dev A/B: rounds 1-4)
    return rcp_(1.f + __builtin_amdgcn_exp2f(x * -1.44269504f));
#endif
    // |x| > 128 saturates like torch.sigmoid (an overflowing sdf * inv_std, +-inf: t_lo would be inf - inf); a NaN stays a NaN
    x = x < -128.f ? -128.f : x;
    x = x > 128.f ? 128.f : x;
    const float c_hi = -1.44269504f, c_lo = -1.92596303e-8f;  // -log2(e) = c_hi + c_lo
    const float t_hi = fminf(x * c_hi, 126.f);
    const float t_lo = fmaf(x, c_hi, -(x * c_hi)) + x * c_lo;
    const float e = __builtin_amdgcn_exp2f(t_hi);
    return rcp_(1.f + fmaf(e, t_lo * 0.693147181f, e));
}

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
    a.pass = tt_opaque(a.rat >= 0.f ? 1.f : 0.f) * (a.rat <= 1.f ? 1.f : 0.f);  // (opaque: or hipcc re-fuses the two compares)
    return a;
}


// exp(-x) for x >= 0 on the hardware exp2 with the argument carried as a pair, as in sigmoid_ (~2 ulp)
__device__ __forceinline__ float exp_neg_(float x) {
    const float c_hi = -1.44269504f, c_lo = -1.92596303e-8f;  // -log2(e) = c_hi + c_lo
    const float t_hi = x * c_hi;
    const float t_lo = fmaf(x, c_hi, -t_hi) + x * c_lo;
    const float e = __builtin_amdgcn_exp2f(t_hi);
    return fmaf(e, t_lo * 0.693147181f, e);
}

// VolSDF density (neus_volume_renderer.py:19-23): k (0.5 + 0.5 sign(sdf) expm1(-|sdf| k)), k = clamp(inv_std, 0, 80)
//   sdf > 0: k E / 2,  sdf < 0: k (1 - E / 2),  sdf = 0: k / 2      with E = exp(-|sdf| k)
// `E` / `k` return exp(-|sdf| k) and the clamped k for the derivatives.
__device__ __forceinline__ float volsdf_density(float sdf, float kstd, float& E, float& k) {
    k = fminf(fmaxf(kstd, 0.f), 80.f);
    E = exp_neg_(fabsf(sdf) * k);
    return k * (sdf > 0.f ? 0.5f * E : 1.f - 0.5f * E);  // (sdf = 0: E = 1, both arms give k / 2)
}

// use_volsdf = True (neus_volume_renderer.py:95-96): alpha = |dists| x density, no clip, no normal / cosine term.
//   d alpha / d sdf     = -|dt| k^2 E / 2         (autograd's sign(0) = 0, |.|'(0) = 0: zero at sdf = 0)
//   d alpha / d inv_std = |dt| (density / k - sign(sdf) |sdf| k E / 2) inside the clamp [0, 80], 0 outside
__device__ __forceinline__ AlphaTerms volsdf_alpha_terms(float sdf, float dt, float kstd) {
    AlphaTerms a = {};
    float E, k;
    const float dens = volsdf_density(sdf, kstd, E, k);
    const float adt = fabsf(dt);
    a.alpha = adt * dens;
    a.d_sdf = sdf != 0.f ? -0.5f * adt * k * k * E : 0.f;
    const float shape = sdf > 0.f ? 0.5f * E : 1.f - 0.5f * E;  // density / k
    const float dshape = (sdf > 0.f ? -0.5f : sdf < 0.f ? 0.5f : 0.f) * fabsf(sdf) * k * E;
    a.d_k = (kstd >= 0.f && kstd <= 80.f) ? adt * (shape + dshape) : 0.f;
    a.pass = 1.f;
    return a;
}
