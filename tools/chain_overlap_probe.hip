// chain_overlap_probe.hip -- can ONE wave hide the VALU work of an MLP-style chain behind its own MFMAs if it carries two
// independent half-width tiles instead of one full-width tile?
//
// The decode kernels are chains: product (a block of MFMAs) -> VALU on its result (ReLU, scale search, operand split) ->
// next product ...  Each step depends on the one before, so a wave with ONE tile alternates between the matrix pipe and
// the VALU and their times add up (profiles/experiments/README.md, rounds 3-6; pipe_overlap_probe.hip shows the hardware
// overlaps INDEPENDENT work).  Two 16-sample tiles per wave (v_mfma_f32_16x16x32_f16: the same MFMA count per product, half
// the cycles each, half the VALU instructions per vector) give a wave two independent chains of the same total work and
// the same register footprint; software-pipelined by half a layer -- MFMAs of tile X next to the VALU block of tile Y --
// they could overlap.  This probe times exactly that dependency structure, nothing else (operands in registers, no LDS,
// no memory):
//   serial     per layer: NM x mfma 32x32x16 (two accumulators) on a B operand made from v[], then NV x v_fma that READ the
//              accumulators and update v[]  (the next layer's B operand depends on them)
//   pipelined  tiles X and Y, per layer and tile NM x mfma 16x16x32 (four accumulators) and NV / 2 x v_fma, program order
//              "1 MFMA of X : R VALU of Y" then "1 MFMA of Y : R VALU of X", pinned with scheduling barriers
//   hipcc --offload-arch=gfx950 -O3 -o tools/chain_overlap_probe tools/chain_overlap_probe.hip && ./tools/chain_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ h8 pack8(const float* v) {
    h8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)v[j];
    return b;
}

// one full-width chain
template <int NM, int NV>
__global__ __launch_bounds__(512) void k_serial(float* out, int layers, float seed) {
    h8 a, a2;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(seed + 0.001f * (threadIdx.x & 31) + 0.01f * j);
        a2[j] = (_Float16)(seed + 0.001f * (threadIdx.x & 31) + 0.01f * j + 0.1f);
    }
    float v[32];
    for (int j = 0; j < 32; ++j) v[j] = seed + 0.01f * j + 0.001f * threadIdx.x;
    const float m = 0.5f + seed;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < layers; ++l) {
        f32x16 acc0 = {}, acc1 = {};
        const h8 b0 = pack8(v), b1 = pack8(v + 8);
        FENCE();
#pragma unroll
        for (int k = 0; k < NM / 2; ++k) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, (k & 1) ? b1 : b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, (k & 1) ? b0 : b1, acc1, 0, 0, 0);
        }
        FENCE();
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int j = r & 31;
            const float x = j < 16 ? acc0[j] : acc1[j - 16];
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j]) : "v"(x), "v"(m));
        }
        FENCE();
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (float)(t_end - t_begin) / (float)layers;
    float s = 0.f;
    for (int j = 0; j < 32; ++j) s += v[j];
    if (s == 123.456f) out[0] = s;
}

// two half-width chains, software-pipelined by half a layer; R = VALU instructions placed behind every MFMA
template <int NM, int NV, int R>
__global__ __launch_bounds__(512) void k_pipelined(float* out, int layers, float seed) {
    h8 a4[4];  // one A operand (weight row tile) per accumulator: four distinct products, nothing for the compiler to merge
    for (int t = 0; t < 4; ++t)
        for (int j = 0; j < 8; ++j) a4[t][j] = (_Float16)(seed + 0.001f * (threadIdx.x & 15) + 0.01f * j + 0.1f * t);
    float vx[16], vy[16];
    for (int j = 0; j < 16; ++j) {
        vx[j] = seed + 0.01f * j + 0.001f * threadIdx.x;
        vy[j] = seed - 0.01f * j + 0.002f * threadIdx.x;
    }
    const float m = 0.5f + seed;
    f32x4 ax[4] = {}, ay[4] = {};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    // prologue: the MFMAs of tile X, layer 0
    {
        const h8 b0 = pack8(vx), b1 = pack8(vx + 8);
#pragma unroll
        for (int k = 0; k < NM; ++k) ax[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[k & 3], (k & 4) ? b1 : b0, ax[k & 3], 0, 0, 0);
    }
    for (int l = 0; l < layers; ++l) {
        // ---- slot 1: MFMAs of Y(l)  ||  VALU of X(l) (reads ax) ----
        {
            const h8 b0 = pack8(vy), b1 = pack8(vy + 8);
            f32x4 n0 = {}, n1 = {}, n2 = {}, n3 = {};
            FENCE();
            int r = 0;
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                const h8 bb = (k & 4) ? b1 : b0;
                if ((k & 3) == 0) n0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[0], bb, n0, 0, 0, 0);
                else if ((k & 3) == 1) n1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[1], bb, n1, 0, 0, 0);
                else if ((k & 3) == 2) n2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[2], bb, n2, 0, 0, 0);
                else n3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[3], bb, n3, 0, 0, 0);
                FENCE();
#pragma unroll
                for (int q = 0; q < R; ++q, ++r) {
                    if (r < NV / 2) {
                        const int j = r & 15;
                        const float x = ax[j >> 2][j & 3];
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vx[j]) : "v"(x), "v"(m));
                    }
                }
                FENCE();
            }
#pragma unroll
            for (; r < NV / 2; ++r) {
                const int j = r & 15;
                const float x = ax[j >> 2][j & 3];
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vx[j]) : "v"(x), "v"(m));
            }
            ay[0] = n0, ay[1] = n1, ay[2] = n2, ay[3] = n3;
        }
        // ---- slot 2: MFMAs of X(l + 1)  ||  VALU of Y(l) (reads ay) ----
        {
            const h8 b0 = pack8(vx), b1 = pack8(vx + 8);
            f32x4 n0 = {}, n1 = {}, n2 = {}, n3 = {};
            FENCE();
            int r = 0;
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                const h8 bb = (k & 4) ? b1 : b0;
                if ((k & 3) == 0) n0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[0], bb, n0, 0, 0, 0);
                else if ((k & 3) == 1) n1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[1], bb, n1, 0, 0, 0);
                else if ((k & 3) == 2) n2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[2], bb, n2, 0, 0, 0);
                else n3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[3], bb, n3, 0, 0, 0);
                FENCE();
#pragma unroll
                for (int q = 0; q < R; ++q, ++r) {
                    if (r < NV / 2) {
                        const int j = r & 15;
                        const float x = ay[j >> 2][j & 3];
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vy[j]) : "v"(x), "v"(m));
                    }
                }
                FENCE();
            }
#pragma unroll
            for (; r < NV / 2; ++r) {
                const int j = r & 15;
                const float x = ay[j >> 2][j & 3];
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vy[j]) : "v"(x), "v"(m));
            }
            ax[0] = n0, ax[1] = n1, ax[2] = n2, ax[3] = n3;
        }
    }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (float)(t_end - t_begin) / (float)layers;
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += vx[j] + vy[j] + ax[j >> 2][j & 3];
    if (s == 123.456f) out[0] = s;
}

template <class F>
static float timed(F launch, float* out, float* cycles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(4000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(cycles, out + 1, 4, hipMemcpyDeviceToHost);  // s_memtime cycles per layer of wave 0 (one wave's view)
    return ms;
}

template <int NM, int NV, int R>
static void row(int waves_per_simd, float* out) {
    const int blocks = 256, threads = 256 * waves_per_simd;  // one workgroup per CU
    float cs, cp;
    const float ts = timed([&](int layers) { hipLaunchKernelGGL((k_serial<NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, layers, 0.f); }, out, &cs);
    const float tp = timed([&](int layers) { hipLaunchKernelGGL((k_pipelined<NM, NV, R>), dim3(blocks), dim3(threads), 0, 0, out, layers, 0.f); }, out, &cp);
    // per layer of 32 samples and wave: NM x 32 matrix-pipe cycles (both forms), NV VALU instructions
    printf("waves/SIMD %d  per layer and wave: %2d MFMA = %4d pipe cycles, %3d VALU:  one tile %6.0f cycles   two pipelined half tiles %6.0f   "
           "(%.3f / %.3f ms)  -> x%.2f\n",
           waves_per_simd, NM, NM * 32, NV, cs, cp, ts, tp, ts / tp);
}

int main() {
    float* out;
    hipMalloc(&out, 8);
    printf("s_memtime cycles one wave spends per layer (4000 layers); waves/SIMD 2: the two waves of a SIMD share its pipes\n");
    for (int w = 1; w <= 2; ++w) {
        row<48, 288, 3>(w, out);   // a 64 x 64 three-piece product + ~290 VALU (scale search, split, ReLU): the kernels' ratio
        row<48, 192, 2>(w, out);
        row<24, 288, 6>(w, out);   // a two-piece product with the same VALU
        row<48, 96, 1>(w, out);
    }
    return 0;
}
