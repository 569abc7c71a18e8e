// Dev microbenchmark: fp32 global atomic add throughput by memory scope (agent vs workgroup vs wavefront).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SCOPE>
__global__ void k(float* buf, unsigned n_texels, int iters, int per_xcd) {
    const int lane = threadIdx.x & 63, ch = lane & 31, hi = lane >> 5;
    unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned xcc = 0;
    if (per_xcd) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
    }
    float* base = buf + (size_t)xcc * n_texels * 32;
    unsigned state = wave * 2654435761u + 12345u + hi * 40503u;
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        unsigned t = (state >> 8) % n_texels;
        __hip_atomic_fetch_add(base + (size_t)t * 32 + ch, 1.0f + ch, __ATOMIC_RELAXED, SCOPE);
    }
}
int main() {
    const unsigned n_texels = 6 * 256 * 256;
    float* buf;
    hipMalloc(&buf, (size_t)8 * n_texels * 32 * 4);
    hipMemset(buf, 0, (size_t)8 * n_texels * 32 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 1024;
    for (int per_xcd = 0; per_xcd < 2; ++per_xcd)
    for (int sc = 0; sc < 3; ++sc) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (sc == 0) hipLaunchKernelGGL((k<__HIP_MEMORY_SCOPE_AGENT>), dim3(blocks), dim3(256), 0, 0, buf, n_texels, iters, per_xcd);
            if (sc == 1) hipLaunchKernelGGL((k<__HIP_MEMORY_SCOPE_WORKGROUP>), dim3(blocks), dim3(256), 0, 0, buf, n_texels, iters, per_xcd);
            if (sc == 2) hipLaunchKernelGGL((k<__HIP_MEMORY_SCOPE_WAVEFRONT>), dim3(blocks), dim3(256), 0, 0, buf, n_texels, iters, per_xcd);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        double n = (double)blocks * 256 * iters;
        const char* names[] = {"agent", "workgroup", "wavefront"};
        printf("per_xcd_buffer=%d scope=%-10s %8.3f ms  %8.1f G float-atomics/s\n", per_xcd, names[sc], ms, n / ms / 1e6);
    }
    // correctness of the per-XCD scheme: sum of the 8 copies must equal the number of adds
    return 0;
}
