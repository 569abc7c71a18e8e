// Dev probe: issue rate of v_cvt_pk_f16_f32 (RNE) vs v_cvt_pkrtz_f16_f32 on gfx950, for inputs whose fp16 results are
// normal vs SUBNORMAL (the `lo` terms of the operand splits are often fp16 subnormals).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float scale, unsigned long long* out, unsigned* sink) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x + 1.37f * i + 1.f) * scale;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            unsigned r;
            if (MODE == 0) r = __builtin_bit_cast(unsigned, __builtin_convertvector((f2_t){x[i], x[i + 1]}, h2_t));
            else r = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]));
            acc ^= r;
            x[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[i]) ^ (acc & 1u));  // keep the loop honest
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main() {
    unsigned long long* d; unsigned* s;
    hipMalloc(&d, 8); hipMalloc(&s, 256);
    const float scales[3] = {1.0f, 1e-6f, 1e-9f};  // results: normal, subnormal, zero
    const char* names[3] = {"normal results", "subnormal results", "results underflow to 0"};
    for (int m = 0; m < 2; ++m)
        for (int c = 0; c < 3; ++c) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, scales[c], d, s);
            else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, scales[c], d, s);
            unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("%s, %-24s: %.2f cycles per convert (+ 3 filler VALU)\n", m == 0 ? "v_cvt_pk_f16_f32 (RNE)" : "v_cvt_pkrtz_f16_f32   ",
                   names[c], (double)h / (256 * 8));
        }
    return 0;
}
