// Probe: two v_cndmask_b32_e64 that read the same SGPR-pair lane mask shortly after a SALU wrote it must select the same
// lanes.  (Observed in k_query_points: the FIRST consumer sometimes sees stale bits for lanes 48..63 when the SIMD runs a
// single wave -- bilinear weight w[2] wrong while the texel offset selected by the same mask is right.)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/sgpr_hazard_probe tools/sgpr_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VARIANT>
__global__ void probe(int iters, unsigned long long* bad, unsigned long long* badmask) {
    const int lane = threadIdx.x & 63;
    unsigned long long nbad = 0, bm = 0;
    int x = lane * 7919 + 13;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525 + 1013904223 + lane;
        int a = (x >> 8) & 1023, b = ((x >> 3) ^ it) & 1023, c = 512;
        int r0, r1;
        const int one = 0x11111111, two = 0x22222222;
        // six compares in a row (as the bilinear in-bounds tests), SALU combination of the LAST one right behind
#define CMPS "v_cmp_lt_i32_e32 vcc, -2, %2\n v_cmp_lt_i32_e64 s[30:31], -1, %2\n v_cmp_gt_i32_e64 s[32:33], %4, %2\n" \
             "v_cmp_gt_i32_e64 s[34:35], %4, %3\n v_cmp_lt_i32_e64 s[20:21], -1, %3\n v_cmp_gt_i32_e64 s[24:25], %4, %3\n"
#define TAIL "v_cndmask_b32_e64 %0, 0, %5, s[22:23]\n v_cndmask_b32_e64 %1, 0, %6, s[22:23]\n"
#define OPS : "=&v"(r0), "=&v"(r1) : "v"(a), "v"(b), "v"(c), "v"(one), "v"(two) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s30", "s31", "s32", "s33", "s34", "s35", "vcc"
        if (VARIANT == 0)
            asm volatile(CMPS "s_and_b64 s[26:27], s[30:31], s[32:33]\n s_and_b64 s[32:33], vcc, s[34:35]\n"
                              "s_and_b64 s[22:23], s[20:21], s[24:25]\n" TAIL OPS);
        else if (VARIANT == 1)
            asm volatile(CMPS "s_and_b64 s[22:23], s[20:21], s[24:25]\n" TAIL OPS);
        else if (VARIANT == 2)
            asm volatile(CMPS "s_and_b64 s[22:23], s[20:21], s[24:25]\n s_nop 4\n" TAIL OPS);
        else if (VARIANT == 3)
            asm volatile(CMPS "s_nop 4\n s_and_b64 s[22:23], s[20:21], s[24:25]\n" TAIL OPS);
        else
            asm volatile(CMPS "s_nop 4\n s_and_b64 s[22:23], s[20:21], s[24:25]\n s_nop 4\n" TAIL OPS);
        const bool m = b > -1 && c > b;
        if (r0 != (m ? one : 0) || r1 != (m ? two : 0)) {
            ++nbad;
            bm |= 1ull << lane;
        }
    }
    if (nbad) {
        atomicAdd(bad, nbad);
        atomicOr(badmask, bm);
    }
}

template <int V>
void run(unsigned long long* bad, unsigned long long* bm) {
    for (int waves = 1; waves <= 2; ++waves) {
        (void)hipMemset(bad, 0, 8);
        (void)hipMemset(bm, 0, 8);
        probe<V><<<256, 256 * waves>>>(400000, bad, bm);
        (void)hipDeviceSynchronize();
        unsigned long long hb, hm;
        (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hm, bm, 8, hipMemcpyDeviceToHost);
        printf("variant %d, %d wave(s)/SIMD: %llu mismatches, lane mask %016llx\n", V, waves, hb, hm);
    }
}

int main() {
    unsigned long long *bad, *bm;
    (void)hipMalloc(&bad, 8);
    (void)hipMalloc(&bm, 8);
    run<0>(bad, bm); run<1>(bad, bm); run<2>(bad, bm); run<3>(bad, bm); run<4>(bad, bm);
    return 0;
}
