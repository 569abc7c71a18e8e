// pipe_overlap_probe.hip -- does a gfx950 SIMD overlap one wave's MFMAs with VALU work (its own or another wave's)?
// Times three loops per configuration: MFMA only, VALU only, both interleaved (1 MFMA : R VALU in program order).
//   hipcc --offload-arch=gfx950 -O3 -o pipe_overlap_probe tools/pipe_overlap_probe.hip && ./pipe_overlap_probe
// Round 5: the decode kernels' matrix-pipe time adds to their other work whatever the instruction order
// (profiles/experiments/README.md); this probe asks the hardware the same question without the kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, int R, bool DEP>  // MODE 0: mfma, 1: valu, 2: both; R valu per mfma; DEP: all MFMAs on one accumulator
__global__ __launch_bounds__(512) void k_probe(float* out, int iters, float seed) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(seed + threadIdx.x * 0.001f + j);
        b[j] = (_Float16)(seed - j);
    }
    f16v acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x;
    const float m = 1.0001f + seed, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE != 1) {
                if (DEP) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                else if (u == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                else if (u == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                else if (u == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
                else acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE != 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r & 7]) : "v"(m), "v"(c));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j] + acc2[j] + acc3[j];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int R, bool DEP>
static float run(int blocks, int threads, int iters) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<MODE, R, DEP>), dim3(blocks), dim3(threads), 0, 0, out, 10, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<MODE, R, DEP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

template <int R, bool DEP>
static void row(int waves_per_simd) {
    const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;  // one block per CU
    const float tm = run<0, R, DEP>(blocks, threads, iters), tv = run<1, R, DEP>(blocks, threads, iters),
                tb = run<2, R, DEP>(blocks, threads, iters);
    printf("waves/SIMD %d  VALU per MFMA %2d  %s accumulators:  mfma %7.3f ms  valu %7.3f ms  both %7.3f ms   "
           "sum %7.3f  max %7.3f  -> overlap %.2f\n",
           waves_per_simd, R, DEP ? "one " : "four", tm, tv, tb, tm + tv, tm > tv ? tm : tv,
           (tm + tv - tb) / (tm < tv ? tm : tv));
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        row<4, false>(w);
        row<8, false>(w);
        row<8, true>(w);
        row<16, false>(w);
    }
    return 0;
}
