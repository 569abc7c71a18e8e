// Dev microbenchmark: LDS atomic throughput per wave-instruction (cycles), float vs int, by address sharing.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int mode, int share, int iters, float* out, long long* cyc) {
    __shared__ float win[64 * 33 * 4];
    __shared__ int iw[64 * 33 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 64 * 33 * 4; e += blockDim.x) { win[e] = 0.f; iw[e] = 0; }
    __syncthreads();
    float* w = win + wave * 64 * 33;
    int* wi = iw + wave * 64 * 33;
    // address: lanes grouped by `share` map to the same slot; bank = slot + ch (stride 33)
    const int slot = (lane & 31) / share, ch = (lane >> 5) * 4;
    long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int a = ((slot + it) & 63) * 33 + ch + (r & 3) + 8 * (r >> 2);
            if (mode == 0) atomicAdd(&w[a], 1.0f + r);
            else if (mode == 1) atomicAdd(&wi[a], 1 + r);
            else if (mode == 2) acc += __int_as_float(atomicAdd(&wi[a], 1 + r));
            else if (mode == 3) { w[a] = w[a] + 1.0f; }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (acc == 123.f) out[0] = acc;
    out[1 + threadIdx.x % 4] = w[lane] + wi[lane];
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 8);
    const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_u32", "plain rmw (racy)"};
    for (int waves : {1, 4}) for (int mode = 0; mode < 4; ++mode) for (int share : {1, 2, 4, 8, 32}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, mode, share, 200, out, cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("waves/CU=%d %-18s share=%2d : %7.1f cycles per wave-instruction\n", waves, names[mode], share, (double)c / (200 * 16));
    }
    return 0;
}
