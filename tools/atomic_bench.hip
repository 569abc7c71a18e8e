// Dev microbenchmark: fp32 global atomicAdd throughput in the scatter pattern of the backward kernels
// (each half-wave adds 32 contiguous floats = one 128-byte texel; texels pseudo-random in a 50 MB buffer).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_atomic(float* buf, unsigned n_texels, int iters, int mode, unsigned hot) {
    const int lane = threadIdx.x & 63, ch = lane & 31, hi = lane >> 5;
    unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned state = wave * 2654435761u + 12345u + hi * 40503u;
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        unsigned t;
        if (mode == 0) t = (state >> 8) % n_texels;                 // uniform random texel
        else if (mode == 1) t = (state >> 8) % hot;                  // small hot set (contention)
        else t = (wave * 64u + (it & 63) * 2u + hi) % n_texels;      // streaming, distinct lines
        float v = 1.0f + ch;
        if (mode == 3) { buf[(size_t)t * 32 + ch] = v; }             // plain stores for comparison
        else atomicAdd(buf + (size_t)t * 32 + ch, v);
    }
}

int main() {
    const unsigned n_texels = 6 * 256 * 256;
    float* buf;
    hipMalloc(&buf, (size_t)n_texels * 32 * 4);
    hipMemset(buf, 0, (size_t)n_texels * 32 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 2000;
    const char* names[] = {"random texels", "hot set of 4096 texels", "streaming distinct", "plain stores random(ish)"};
    for (int blocks : {256, 1024, 2048}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, buf, n_texels, 10, mode, 4096u);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, buf, n_texels, iters, mode == 3 ? 3 : mode, 4096u);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            double n = (double)blocks * 256 * iters;
            printf("blocks=%4d %-28s %8.3f ms  %8.1f G float-ops/s  %7.2f TB/s\n", blocks, names[mode], ms,
                   n / ms / 1e6, n * 4 / ms / 1e9);
        }
    }
    return 0;
}
