// tt_grad2.hip -- operator-level drop-in for the reference's only native op:
//   gridsample_grad2.grad2_2d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners)
//   -> [grad_grad_output, grad_input, grad_grid]       (gridsample_cuda.cpp:26-37,53-54; kernel gridsample_cuda.cu:27-210)
// i.e. the backward of aten::grid_sampler_2d_backward (bilinear), needed because stock PyTorch has no double backward
// for grid_sample.  The fused render path of this library does NOT use it (its second-order terms are folded into
// k_decode_bwd_geo); it is exported so that reference code that still calls `grad2_2d` runs on MI355X.
//
// Layout as the reference: contiguous NCHW input / grad2_grad_input / grad_input, grid (N,Ho,Wo,2),
// grad_output / grad_grad_output (N,C,Ho,Wo).  Supported: bilinear, padding zeros, align_corners = False (what the
// reference's call site uses, cuda_gridsample.py:39-40); anything else returns TT_ERR_UNSUPPORTED.
// One lane per output point (coalesced grid / grad_output traffic), loop over channels, atomics into grad_input.
#include "tt_device.h"
#include "tt_host.h"

struct Grad2Params {
    const float* g2_inp;
    const float* g2_grid;
    const float* g_out;
    const float* inp;
    const float* grid;
    int N, C, H, W;
    long long M;  // Ho*Wo
    float* gg_out;
    float* g_inp;
    float* g_grid;
};

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_grid_sample_2d_grad2(Grad2Params p) {
    const long long total = (long long)p.N * p.M;
    const size_t HW = (size_t)p.H * p.W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / p.M, m = idx - n * p.M;
        const float x = p.grid[idx * 2 + 0], y = p.grid[idx * 2 + 1];
        const float gix_mult = 0.5f * p.W, giy_mult = 0.5f * p.H;  // d ix / d x for align_corners = False
        const float ix = ((x + 1.f) * p.W - 1.f) / 2.f, iy = ((y + 1.f) * p.H - 1.f) / 2.f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)p.W + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)p.H + 1.f);
        const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
        const bool bx0 = x0 >= 0 && x0 < p.W, bx1 = x0 + 1 >= 0 && x0 + 1 < p.W;
        const bool by0 = y0 >= 0 && y0 < p.H, by1 = y0 + 1 >= 0 && y0 + 1 < p.H;
        const bool in_nw = bx0 && by0, in_ne = bx1 && by0, in_sw = bx0 && by1, in_se = bx1 && by1;
        const size_t o_nw = (size_t)y0 * p.W + x0, o_ne = o_nw + 1, o_sw = o_nw + p.W, o_se = o_sw + 1;
        const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
        const float dx = p.g2_grid[idx * 2 + 0] * gix_mult, dy = p.g2_grid[idx * 2 + 1] * giy_mult;
        const float nw_tmp = -dx * wy0 - dy * wx0, ne_tmp = dx * wy0 - dy * wx1;
        const float sw_tmp = -dx * wy1 + dy * wx0, se_tmp = dx * wy1 + dy * wx1;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < p.C; ++c) {
            const size_t pl = ((size_t)n * p.C + c) * HW;
            const float v_nw = in_nw ? p.inp[pl + o_nw] : 0.f, v_ne = in_ne ? p.inp[pl + o_ne] : 0.f;
            const float v_sw = in_sw ? p.inp[pl + o_sw] : 0.f, v_se = in_se ? p.inp[pl + o_se] : 0.f;
            const float g_nw = in_nw ? p.g2_inp[pl + o_nw] : 0.f, g_ne = in_ne ? p.g2_inp[pl + o_ne] : 0.f;
            const float g_sw = in_sw ? p.g2_inp[pl + o_sw] : 0.f, g_se = in_se ? p.g2_inp[pl + o_se] : 0.f;
            const size_t oo = ((size_t)n * p.C + c) * p.M + m;
            float ggo = g_nw * nw + g_ne * ne + g_sw * sw + g_se * se;
            ggo += v_nw * nw_tmp + ne_tmp * v_ne + sw_tmp * v_sw + se_tmp * v_se;
            p.gg_out[oo] = ggo;
            const float go = p.g_out[oo];
            if (in_nw) atomicAdd(p.g_inp + pl + o_nw, nw_tmp * go);
            if (in_ne) atomicAdd(p.g_inp + pl + o_ne, ne_tmp * go);
            if (in_sw) atomicAdd(p.g_inp + pl + o_sw, sw_tmp * go);
            if (in_se) atomicAdd(p.g_inp + pl + o_se, se_tmp * go);
            const float dxy = v_nw - v_ne - v_sw + v_se;
            gix += go * (-g_nw * wy0 + g_ne * wy0 - g_sw * wy1 + g_se * wy1);
            gix += go * dy * dxy;
            giy += go * (-g_nw * wx0 - g_ne * wx1 + g_sw * wx0 + g_se * wx1);
            giy += go * dx * dxy;
        }
        p.g_grid[idx * 2 + 0] = gix * gix_mult;
        p.g_grid[idx * 2 + 1] = giy * giy_mult;
    }
}
#pragma clang fp contract(fast)

extern "C" int tt_grid_sample_2d_grad2(const float* grad2_grad_input, const float* grad2_grad_grid,
                                       const float* grad_output, const float* input, const float* grid, int32_t n,
                                       int32_t c, int32_t h, int32_t w, int64_t n_points_per_batch,
                                       int32_t padding_mode, int32_t align_corners, float* grad_grad_output,
                                       float* grad_input, float* grad_grid, void* stream) {
    if (!grad2_grad_input || !grad2_grad_grid || !grad_output || !input || !grid || !grad_grad_output || !grad_input ||
        !grad_grid || n <= 0 || c <= 0 || h <= 0 || w <= 0 || n_points_per_batch <= 0)
        return TT_ERR_BAD_ARG;
    if (padding_mode != 0 || align_corners != 0) return TT_ERR_UNSUPPORTED;
    Grad2Params p;
    p.g2_inp = grad2_grad_input;
    p.g2_grid = grad2_grad_grid;
    p.g_out = grad_output;
    p.inp = input;
    p.grid = grid;
    p.N = n;
    p.C = c;
    p.H = h;
    p.W = w;
    p.M = n_points_per_batch;
    p.gg_out = grad_grad_output;
    p.g_inp = grad_input;
    p.g_grid = grad_grid;
    hipStream_t s = (hipStream_t)stream;
    // grad_input accumulates with atomics: start from zero like the reference's zeros_like (gridsample_cuda.cu:553-555)
    if (hipMemsetAsync(grad_input, 0, (size_t)n * c * h * w * sizeof(float), s) != hipSuccess) return TT_ERR_LAUNCH;
    long long total = (long long)n * n_points_per_batch;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_grid_sample_2d_grad2, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return tt_check_launch();
}
