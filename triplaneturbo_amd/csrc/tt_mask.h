// tt_mask.h -- lane-mask hygiene helpers shared by every kernel file (see corners_setup in tt_device.h and
// tools/mask_hazard_lint.py).
#pragma once
#include <hip/hip_runtime.h>

// A v_cndmask that reads an SGPR pair the scalar ALU has just produced by combining freshly written VALU compare masks
// (v_cmp -> s_and_b64 / s_or_b64 -> v_cndmask) was observed to see stale bits for the upper lanes on MI355X.  The
// kernels therefore never let a SELECT depend on `a && b` / `a || b` of two per-lane compares:
//   * a flag that is combined with others lives as a 0/1 FLOAT made by one compare + select and is combined by
//     multiplication; tt_opaque() hides the fact that it is 0/1 from the compiler, which otherwise folds
//     (a ? 1 : 0) * (b ? 1 : 0) back into s_and_b64 + one select;
//   * "any of these is non-zero" is ONE compare of a sum of magnitudes (tt_any_nonzero*), not an or of compares;
//   * "x == a || x == b" on integers is ONE compare of a min of differences (tt_eq_either).
// tests/test_host_logic.py runs the lint on the built library.
__device__ __forceinline__ float tt_opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ bool tt_any_nonzero3(float a, float b, float c) {
    return (__builtin_fabsf(a) + __builtin_fabsf(b)) + __builtin_fabsf(c) != 0.f;  // NaN / Inf count as non-zero
}
__device__ __forceinline__ bool tt_any_nonzero4(float a, float b, float c, float d) {
    return (__builtin_fabsf(a) + __builtin_fabsf(b)) + (__builtin_fabsf(c) + __builtin_fabsf(d)) != 0.f;
}
// x == a || x == b
__device__ __forceinline__ bool tt_eq_either(int x, int a, int b) {
    const unsigned da = (unsigned)(x ^ a), db = (unsigned)(x ^ b);
    return (da < db ? da : db) == 0u;
}

// the same for any arithmetic type of the typed kernels (float / double)
template <typename A>
__device__ __forceinline__ A tt_opaque_t(A x) {
    asm volatile("" : "+v"(x));
    return x;
}
