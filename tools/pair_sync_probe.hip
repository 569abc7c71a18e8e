// Dev probe: latency of a wave-pair hand-off through LDS flags inside one workgroup (no s_barrier): wave A writes 4 KB of
// payload + a sequence flag, wave B spins on the flag, reads the payload, answers the same way.  Cycles per round trip
// (s_memtime) for pairs (0,1), (0,2), (0,4) of an 8-wave workgroup, with the other waves idle or spinning on VALU work.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int wa, int wb, int busy, int iters, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) float box[2][64 * 16];
    __shared__ volatile int flag[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 2) flag[threadIdx.x] = 0;
    __syncthreads();
    float acc = lane;
    if (wave != wa && wave != wb) {
        if (busy) for (int i = 0; i < iters * 400; ++i) acc = fmaf(acc, 1.0001f, 0.5f);
        sink[threadIdx.x] = acc;
        return;
    }
    const int me = wave == wa ? 0 : 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 1; it <= iters; ++it) {
        if ((it & 1) == (me ^ 1)) {  // my turn to send: odd iterations wave A, even wave B
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc, acc + 1, acc + 2, acc + 3};
                *reinterpret_cast<f32x4*>(&box[me][(q * 64 + lane) * 4]) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): DS ops of a wave execute in order, but be explicit
            if (lane == 0) flag[me] = it;
        } else {
            while (flag[me ^ 1] < it) __builtin_amdgcn_s_sleep(1);
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&box[me ^ 1][(q * 64 + lane) * 4]);
                acc += v[0] + v[1] + v[2] + v[3];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[me] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main() {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 16);
    hipMalloc(&sink, 512 * 4);
    const int iters = 2000;
    for (int busy = 0; busy < 2; ++busy)
        for (int wb : {1, 2, 4, 7}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, 0, wb, busy, iters, d, sink);
            unsigned long long h[2];
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("pair (0,%d) others %s: %.1f cycles per one-way hand-off (4 KB payload + flag)\n", wb,
                   busy ? "busy" : "idle", (double)h[0] / iters);
        }
    return 0;
}
