// Dev probe: does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs on gfx950 (no flush)?  And how accurate is the
// single-accumulator split  a b ~= ah bh + ah bl + al bh  with UNSCALED lo = f16(a - ah) (may be subnormal)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D, float* D3) {
    // A: 32 x 16 row-major, B: 16 x 32 row-major; lane (i, h): A[i][8h+e], B[8h+e][i]
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    h8_t ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
        float a = A[i * 16 + 8 * h + e], b = B[(8 * h + e) * 32 + i];
        _Float16 x = (_Float16)a; ah[e] = x; al[e] = (_Float16)(a - (float)x);
        _Float16 y = (_Float16)b; bh[e] = y; bl[e] = (_Float16)(b - (float)y);
    }
    f32x16 z = {0};
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, z, 0, 0, 0);
    f32x16 d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, d, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, d3, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        D[row * 32 + i] = d[r];
        D3[row * 32 + i] = d3[r];
    }
}
// Round 4: round-toward-zero (v_cvt_pkrtz_f16_f32, rounds 1-3) against round-to-nearest-even (v_cvt_pk_f16_f32) operand
// splits, operands normalised to the top of the fp16 range as the kernels do: 3-term product error of each.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
template <bool RNE>
__device__ void split2(float a, float b, _Float16& ha, _Float16& hb, _Float16& la, _Float16& lb) {
    h2_t p, q;
    if (RNE) {
        p = __builtin_convertvector((f2_t){a, b}, h2_t);
        q = __builtin_convertvector((f2_t){a - (float)p.x, b - (float)p.y}, h2_t);
    } else {
        p = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(a, b));
        q = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(a - (float)p.x, b - (float)p.y));
    }
    ha = p.x; hb = p.y; la = q.x; lb = q.y;
}
template <bool RNE>
__global__ void k_round(const float* A, const float* B, float* D3) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    h8_t ah, al, bh, bl;
    for (int e = 0; e < 8; e += 2) {
        _Float16 x0, x1, y0, y1;
        split2<RNE>(A[i * 16 + 8 * h + e], A[i * 16 + 8 * h + e + 1], x0, x1, y0, y1);
        ah[e] = x0; ah[e + 1] = x1; al[e] = y0; al[e + 1] = y1;
        split2<RNE>(B[(8 * h + e) * 32 + i], B[(8 * h + e + 1) * 32 + i], x0, x1, y0, y1);
        bh[e] = x0; bh[e + 1] = x1; bl[e] = y0; bl[e + 1] = y1;
    }
    f32x16 z = {0};
    f32x16 d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, z, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, d3, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, d3, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D3[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = d3[r];
}
static void rounding_probe() {
    float hA[512], hB[512], hD[1024];
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    for (int mode = 0; mode < 2; ++mode) {
        double worst = 0, sum2 = 0, ref2 = 0;
        for (int trial = 0; trial < 64; ++trial) {
            srand(100 + trial);
            for (int e = 0; e < 512; ++e) {  // top of the fp16 range: |x| < 2^15, a wide spread of magnitudes below
                hA[e] = (rand() / (float)RAND_MAX - 0.5f) * 65000.f * powf(2.f, -(float)(rand() % 6));
                hB[e] = (rand() / (float)RAND_MAX - 0.5f) * 65000.f * powf(2.f, -(float)(rand() % 6));
            }
            hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
            if (mode) hipLaunchKernelGGL(k_round<true>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
            else hipLaunchKernelGGL(k_round<false>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
            hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
            for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
                double ref = 0, mag = 0;
                for (int kk = 0; kk < 16; ++kk) { ref += (double)hA[r * 16 + kk] * hB[kk * 32 + c]; mag += fabs((double)hA[r * 16 + kk] * hB[kk * 32 + c]); }
                const double err = fabs(hD[r * 32 + c] - ref);
                worst = fmax(worst, err / mag); sum2 += err * err; ref2 += ref * ref;
            }
        }
        printf("split rounding %s: worst |err| / sum|a b| = %.3e (2^%.1f), norm-wise %.3e\n", mode ? "RNE (v_cvt_pk_f16_f32)" : "RTZ (v_cvt_pkrtz)  ",
               worst, log2(worst), sqrt(sum2 / ref2));
    }
}
int main() {
    rounding_probe();
    float hA[512], hB[512], hD[1024], hD3[1024];
    srand(1);
    for (int t = 0; t < 2; ++t) {
        for (int e = 0; e < 512; ++e) {
            hA[e] = (rand() / (float)RAND_MAX - 0.5f) * (t == 0 ? 1.f : 3e-6f);  // t=1: all of A in fp16-subnormal range
            hB[e] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
        }
        float *dA, *dB, *dD, *dD3;
        hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096); hipMalloc(&dD3, 4096);
        hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD3);
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(hD3, dD3, 4096, hipMemcpyDeviceToHost);
        double e1 = 0, e3 = 0, nrm = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double ref = 0; for (int kk = 0; kk < 16; ++kk) ref += (double)hA[r * 16 + kk] * hB[kk * 32 + c];
            e1 = fmax(e1, fabs(hD[r * 32 + c] - ref)); e3 = fmax(e3, fabs(hD3[r * 32 + c] - ref)); nrm = fmax(nrm, fabs(ref));
        }
        printf("case %d (|A| ~ %g): max|ref| %.3e  err 1-term %.3e (rel %.2e)  err 3-term %.3e (rel %.2e)\n", t,
               t == 0 ? 0.5 : 1.5e-6, nrm, e1, e1 / nrm, e3, e3 / nrm);
    }
    return 0;
}
