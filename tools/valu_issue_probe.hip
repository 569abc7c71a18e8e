// valu_issue_probe.hip -- how fast can ONE wave issue VALU instructions on a gfx950 SIMD, as a function of the number of
// independent dependency chains and of the waves per SIMD?  (Round 5: the backward decode kernels run one wave per SIMD.)
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o valu_issue_probe tools/valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int C, int OP>  // C independent chains; OP 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_cvt_pk_f16_f32 style mix, 3: v_add_u32
__global__ __launch_bounds__(1024) void k_valu(float* out, int iters, float seed) {
    float v[C];
    float2 w[C];
    for (int j = 0; j < C; ++j) {
        v[j] = seed + j + threadIdx.x;
        w[j] = make_float2(v[j], v[j] + 1.f);
    }
    const float m = 1.0001f + seed, c = 0.5f;
    const float2 m2 = make_float2(m, m), c2 = make_float2(c, c);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r % C]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[r % C]) : "v"(m2), "v"(c2));
            if (OP == 2) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(v[r % C]) : "v"(m), "v"(c));
            if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[r % C]) : "v"(m));
        }
    }
    float s = 0.f;
    for (int j = 0; j < C; ++j) s += v[j] + w[j].x + w[j].y;
    if (s == 123.456f) out[0] = s;
}

template <int C, int OP>
static void run(int waves_per_simd) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((k_valu<C, OP>), dim3(256), dim3(threads), 0, 0, out, 2000, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_valu<C, OP>), dim3(256), dim3(threads), 0, 0, out, iters, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 64.0 * iters * waves_per_simd;
    static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_mix_f32", "v_add_u32"};
    printf("%-14s chains %2d  waves/SIMD %d:  %7.3f ms  %6.2f ns per instruction per SIMD (x 2.1 GHz = %5.2f cycles)\n",
           names[OP], C, waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.1);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<4, 0>(w);
        run<8, 0>(w);
        run<16, 0>(w);
        run<32, 0>(w);
        run<16, 1>(w);
        run<16, 2>(w);
        run<16, 3>(w);
    }
    return 0;
}
