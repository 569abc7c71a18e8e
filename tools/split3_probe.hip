// Dev probe (round 5): the REAL product code of csrc/tt_mfma16.h -- stage_image16 / mv16 / mv16t / mv16_pre / mv16t_pre,
// two-piece (NT = 2) and three-piece (NT = 3) -- against fp64 on random matrices and activation tiles.  Reports per
// product the worst |err| / sum |w||x| (the "product error" of DESIGN.md) and the norm-wise error, next to what a plain fp32
// fmaf chain (the reference's arithmetic, and the fp32-MFMA mode's) gives on the same data.  Also the layout check of the
// third-term images: a wrong lo-image address shows up as an error of 2^-22 instead of 2^-24 or as garbage.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I triplaneturbo_amd/csrc -I include tools/split3_probe.hip -o tools/split3_probe
#include "tt_device.h"
#include "tt_mfma16.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

// y = M x (ROWS x K) for one 32-sample tile; X: [K][32] row-major, Y: [ROWS][32]
template <int ROWS, int K, int NT, bool PRE>
__global__ void k_fwd(const float* M, const float* X, float* Y) {
    __shared__ __attribute__((aligned(16))) float L[IMG16_FLOATS(ROWS, K) + LO16_FLOATS(ROWS, K)];
    float* lo = L + IMG16_FLOATS(ROWS, K);
    stage_image16<ROWS, K, false, NT>(L, M, K, lo);
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    float x[K / 2], y[ROWS / 2];
    for (int r = 0; r < K / 2; ++r) x[r] = X[LIDX(r, hi) * 32 + i];
    if (PRE) {
        float m = 0.f;  // one scale for the tile (the kernels use a per-launch bound)
        for (int e = 0; e < K * 32; ++e) m = fmaxf(m, fabsf(X[e]));
        int E = (int)(__builtin_bit_cast(unsigned, m * 1.0001f) >> 23);
        const float sc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);
        Split16<K, PAIR_SEQ, NT> xs;
        split16_vec<K, PAIR_SEQ, NT>(x, sc, xs);
        mv16_pre<ROWS, K, false, NT>(L, xs, 1.f / sc, y, i, hi, nullptr, lo);
    } else {
        mv16<ROWS, K, true, false, NT>(L, x, y, i, hi, 1.f, nullptr, lo);
    }
    for (int r = 0; r < ROWS / 2; ++r) Y[LIDX(r, hi) * 32 + i] = y[r];
}
// y = M[:, col0 .. col0 + NOUT)^T x for M (NIN rows x KM columns); X: [NIN][32], Y: [NOUT][32]
template <int NOUT, int NIN, int KM, int NT, bool PRE>
__global__ void k_tr(const float* M, const float* X, float* Y, int col0) {
    __shared__ __attribute__((aligned(16))) float L[IMG16_FLOATS(NIN, KM) + LO16_FLOATS(NIN, KM)];
    float* lo = L + IMG16_FLOATS(NIN, KM);
    stage_image16<NIN, KM, false, NT>(L, M, KM, lo);
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    float x[NIN / 2], y[NOUT / 2];
    for (int r = 0; r < NIN / 2; ++r) x[r] = X[LIDX(r, hi) * 32 + i];
    if (PRE) {
        float m = 0.f;
        for (int e = 0; e < NIN * 32; ++e) m = fmaxf(m, fabsf(X[e]));
        int E = (int)(__builtin_bit_cast(unsigned, m * 1.0001f) >> 23);
        const float sc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);
        Split16<NIN, PAIR_TR, NT> xs;
        split16_vec<NIN, PAIR_TR, NT>(x, sc, xs);
        mv16t_pre<NOUT, NIN, KM, NT>(L, col0, xs, 1.f / sc, y, lane, lo);
    } else {
        mv16t<NOUT, NIN, KM, true, false, NT>(L, col0, x, y, lane, 1.f, nullptr, lo);
    }
    for (int r = 0; r < NOUT / 2; ++r) Y[LIDX(r, hi) * 32 + i] = y[r];
}

struct Acc {
    double worst = 0, e2 = 0, r2 = 0, worst32 = 0, e2_32 = 0;
};
static float frand() { return rand() / (float)RAND_MAX - 0.5f; }

template <class Launch>
static void run(const char* name, int rows_out, int n_in, int mrows, int mcols, bool transposed, int col0, bool one_scale, Launch launch) {
    Acc a;
    std::vector<float> M(mrows * mcols), X(n_in * 32), Y(rows_out * 32);
    float *dM, *dX, *dY;
    hipMalloc(&dM, M.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4);
    for (int trial = 0; trial < 24; ++trial) {
        srand(7 + trial);
        const float ms = powf(10.f, (float)(trial % 5) - 2.f);
        for (auto& v : M) v = frand() * ms * powf(2.f, -(float)(rand() % 6));
        for (int s = 0; s < 32; ++s) {
            // every sample its own magnitude -- except for operands split under ONE scale (the kernels use those for vectors
            // whose magnitude does not depend on the sample: activations, masked weight products)
            const float xs = powf(10.f, (float)((trial + (one_scale ? 0 : s)) % 7) - 3.f);
            for (int c = 0; c < n_in; ++c) X[c * 32 + s] = frand() * xs * powf(2.f, -(float)(rand() % 6));
        }
        hipMemcpy(dM, M.data(), M.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
        launch(dM, dX, dY);
        if (hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: HIP error\n", name); return; }
        for (int r = 0; r < rows_out; ++r)
            for (int s = 0; s < 32; ++s) {
                double ref = 0, mag = 0;
                float f32 = 0.f;
                for (int c = 0; c < n_in; ++c) {
                    const float w = transposed ? M[c * mcols + col0 + r] : M[r * mcols + c];
                    ref += (double)w * X[c * 32 + s];
                    mag += fabs((double)w * X[c * 32 + s]);
                    f32 = fmaf(w, X[c * 32 + s], f32);
                }
                const double err = fabs(Y[r * 32 + s] - ref), e32 = fabs(f32 - ref);
                a.worst = fmax(a.worst, err / mag); a.worst32 = fmax(a.worst32, e32 / mag);
                // norm-wise per sample column would hide nothing here: accumulate relative to mag as well
                a.e2 += (err / mag) * (err / mag); a.e2_32 += (e32 / mag) * (e32 / mag); a.r2 += 1.0;
            }
    }
    printf("%-44s worst |err|/sum|wx| = %.3e (2^%6.2f)  rms %.3e (2^%6.2f) | fp32 fmaf chain: worst 2^%6.2f rms 2^%6.2f\n", name, a.worst,
           log2(a.worst), sqrt(a.e2 / a.r2), log2(sqrt(a.e2 / a.r2)), log2(a.worst32), log2(sqrt(a.e2_32 / a.r2)));
    hipFree(dM); hipFree(dX); hipFree(dY);
}

#define FWD(ROWS, K, NT, PRE)                                                                                           \
    run("mv16" #PRE "<" #ROWS "," #K "> NT=" #NT, ROWS, K, ROWS, K, false, 0, PRE, [](float* m, float* x, float* y) {    \
        hipLaunchKernelGGL((k_fwd<ROWS, K, NT, PRE>), dim3(1), dim3(64), 0, 0, m, x, y);                                \
    })
#define TR(NOUT, NIN, KM, NT, PRE, COL0)                                                                                 \
    run("mv16t" #PRE "<" #NOUT "," #NIN "," #KM "> col0=" #COL0 " NT=" #NT, NOUT, NIN, NIN, KM, true, COL0, PRE,         \
        [](float* m, float* x, float* y) {                                                                              \
            hipLaunchKernelGGL((k_tr<NOUT, NIN, KM, NT, PRE>), dim3(1), dim3(64), 0, 0, m, x, y, COL0);                 \
        })

// ---- ONE 16-deep k-step: the error of the PRODUCT scheme itself (no accumulation over k-steps) ----
template <int NT>
__global__ void k_step(const float* A, const float* B, float* D) {
    // A: 32 x 16 row-major, B: 16 x 32 row-major; lane (i, h): A[i][8h+e], B[8h+e][i]
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    h8_t a0, a1, a2, b0, b1, b2;
    for (int e = 0; e < 8; e += 2) {
        h2_t p, q, r;
        if (NT == 3) split_pair3(A[i * 16 + 8 * h + e], A[i * 16 + 8 * h + e + 1], p, q, r);
        else split_pair(A[i * 16 + 8 * h + e], A[i * 16 + 8 * h + e + 1], p, q);
        a0[e] = p.x; a0[e + 1] = p.y; a1[e] = q.x; a1[e + 1] = q.y;
        if (NT == 3) { a2[e] = r.x; a2[e + 1] = r.y; }
        if (NT == 3) split_pair3(B[(8 * h + e) * 32 + i], B[(8 * h + e + 1) * 32 + i], p, q, r);
        else split_pair(B[(8 * h + e) * 32 + i], B[(8 * h + e + 1) * 32 + i], p, q);
        b0[e] = p.x; b0[e + 1] = p.y; b1[e] = q.x; b1[e + 1] = q.y;
        if (NT == 3) { b2[e] = r.x; b2[e + 1] = r.y; }
    }
    f32x16 acc[1] = {{0}};
    const h8_t A0[1] = {a0}, A1[1] = {a1}, A2[1] = {a2};
    mfma_terms<1, NT>(acc, A0, A1, A2, b0, b1, b2);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[0][r];
}
template <int NT>
static void step_probe() {
    float hA[512], hB[512], hD[1024];
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    double worst = 0, e2 = 0, n = 0, worst32 = 0, e2_32 = 0;
    for (int trial = 0; trial < 64; ++trial) {
        srand(100 + trial);
        for (int e = 0; e < 512; ++e) {  // top of the fp16 range as the kernels normalise, a spread of magnitudes below
            hA[e] = frand() * 65000.f * powf(2.f, -(float)(rand() % 6));
            hB[e] = frand() * 65000.f * powf(2.f, -(float)(rand() % 6));
        }
        hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_step<NT>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double ref = 0, mag = 0; float f32 = 0.f;
            for (int k = 0; k < 16; ++k) { ref += (double)hA[r * 16 + k] * hB[k * 32 + c]; mag += fabs((double)hA[r * 16 + k] * hB[k * 32 + c]); f32 = fmaf(hA[r * 16 + k], hB[k * 32 + c], f32); }
            const double err = fabs(hD[r * 32 + c] - ref) / mag, e32 = fabs(f32 - ref) / mag;
            worst = fmax(worst, err); e2 += err * err; worst32 = fmax(worst32, e32); e2_32 += e32 * e32; n += 1;
        }
    }
    printf("one 32x32x16 k-step, %d-piece operands (%d MFMA terms): worst |err|/sum|ab| = 2^%6.2f  rms 2^%6.2f | fp32 fmaf chain of the same 16 products: worst 2^%6.2f rms 2^%6.2f\n",
           NT, NT == 3 ? 6 : 3, log2(worst), log2(sqrt(e2 / n)), log2(worst32), log2(sqrt(e2_32 / n)));
}

int main() {
    step_probe<2>();
    step_probe<3>();
    printf("(name: 0 = per-sample scale inside the product, 1 = pre-split operand under one scale)\n");
    FWD(64, 32, 2, 0); FWD(64, 32, 3, 0);
    FWD(64, 64, 2, 0); FWD(64, 64, 3, 0);
    FWD(64, 96, 2, 0); FWD(64, 96, 3, 0);
    FWD(32, 64, 3, 0);
    FWD(64, 64, 2, 1); FWD(64, 64, 3, 1); FWD(64, 96, 3, 1); FWD(32, 64, 3, 1);
    TR(64, 64, 64, 2, 0, 0); TR(64, 64, 64, 3, 0, 0);
    TR(32, 64, 32, 2, 0, 0); TR(32, 64, 32, 3, 0, 0);
    TR(96, 64, 96, 2, 0, 0); TR(96, 64, 96, 3, 0, 0);
    TR(32, 64, 96, 3, 0, 0); TR(32, 64, 96, 3, 0, 32); TR(32, 64, 96, 3, 0, 64);
    TR(64, 64, 64, 3, 1, 0); TR(32, 64, 32, 3, 1, 0); TR(96, 64, 96, 3, 1, 0);
    return 0;
}
