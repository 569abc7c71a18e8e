// Dev probe: what does ds_read_b64_tr_b16 (gfx950 LDS transpose read) return?  LDS is filled with halfs whose value is
// their own index; every lane passes an address and prints the 4 halfs it receives.  Patterns:
//   A: lane l -> byte address 8 l                      (64 consecutive 8-byte groups)
//   B: lane l -> row (l & 15), 8-byte column (l >> 4) of a [16][RS] half matrix (RS = 32 halfs)
//   C: the (l&15) + j*16 + (l>>4)*64 layout of the guide
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half_t;
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(int pattern, int rs, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int e = threadIdx.x; e < 4096; e += 64) lds[e] = (short)e;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = 8u * l;
    else if (pattern == 1) addr = 2u * ((l & 15) * rs + 4 * (l >> 4));
    else addr = 2u * ((l & 15) * 4 + (l >> 4) * 64);
    addr += (unsigned)(size_t)lds;  // LDS base of the array (static allocation starts at 0, but be explicit)
    s4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d;
    hipMalloc(&d, 256 * 2);
    short h[256];
    for (int pattern = 0; pattern < 3; ++pattern) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, pattern, 32, d);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("pattern %d: lane -> 4 half indices received\n", pattern);
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if ((l & 3) == 3) printf("\n");
        }
    }
    return 0;
}
