// tt_composite.hip -- image-space composites around the renderer (small, bandwidth-trivial kernels that replace
// chains of 5-10 PyTorch ops each).
//
//   tt_patch_composite_fwd/_bwd : PatchRenderer.forward's per-key composite (threestudio/models/renderers/
//       patch_renderer.py:74-88): bilinear upsample (F.interpolate, align_corners=False) of the low-resolution global
//       render to (H, W) with the high-resolution patch pasted at (py, px).  The backward is the exact adjoint, in
//       GATHER form (each low-resolution pixel sums the few output pixels it fed): deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tt_host.h"

struct PatchParams {
    const float* low;    // (B, h, w, C)
    const float* patch;  // (B, PS, PS, C)
    float* out;          // (B, H, W, C)
    int B, h, w, H, W, C, PS, py, px;
};

// ATen area_pixel_compute_source_index (align_corners = False, not cubic): scale * (dst + 0.5) - 0.5, clamped at 0
__device__ __forceinline__ void up_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void k_patch_composite_fwd(PatchParams p) {
    const long long n = (long long)p.B * p.H * p.W * p.C;
    const float sh = (float)p.h / (float)p.H, sw = (float)p.w / (float)p.W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % p.C);
        long long t = e / p.C;
        const int x = (int)(t % p.W);
        t /= p.W;
        const int y = (int)(t % p.H), b = (int)(t / p.H);
        float v;
        if (y >= p.py && y < p.py + p.PS && x >= p.px && x < p.px + p.PS) {
            v = p.patch[(((long long)b * p.PS + (y - p.py)) * p.PS + (x - p.px)) * p.C + c];
        } else {
            int y0, y1, x0, x1;
            float ly0, ly1, lx0, lx1;
            up_index(y, sh, p.h, y0, y1, ly0, ly1);
            up_index(x, sw, p.w, x0, x1, lx0, lx1);
            const float* L = p.low + (long long)b * p.h * p.w * p.C + c;
            const float v00 = L[((long long)y0 * p.w + x0) * p.C], v01 = L[((long long)y0 * p.w + x1) * p.C];
            const float v10 = L[((long long)y1 * p.w + x0) * p.C], v11 = L[((long long)y1 * p.w + x1) * p.C];
            v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);  // ATen upsample_bilinear2d op order
        }
        p.out[e] = v;
    }
}

struct PatchBwdParams {
    const float* g_out;  // (B, H, W, C)
    float* g_low;        // (B, h, w, C) or null (global_detach)
    float* g_patch;      // (B, PS, PS, C)
    int B, h, w, H, W, C, PS, py, px;
};

__global__ __launch_bounds__(256) void k_patch_composite_bwd(PatchBwdParams p) {
    const long long n_patch = (long long)p.B * p.PS * p.PS * p.C;
    const long long n_low = p.g_low ? (long long)p.B * p.h * p.w * p.C : 0;
    const float sh = (float)p.h / (float)p.H, sw = (float)p.w / (float)p.W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_patch + n_low;
         e += (long long)gridDim.x * blockDim.x) {
        if (e < n_patch) {  // the pasted region passes its gradient to the patch render
            const int c = (int)(e % p.C);
            long long t = e / p.C;
            const int x = (int)(t % p.PS);
            t /= p.PS;
            const int y = (int)(t % p.PS), b = (int)(t / p.PS);
            p.g_patch[e] = p.g_out[(((long long)b * p.H + (p.py + y)) * p.W + (p.px + x)) * p.C + c];
            continue;
        }
        const long long el = e - n_patch;
        const int c = (int)(el % p.C);
        long long t = el / p.C;
        const int lx = (int)(t % p.w);
        t /= p.w;
        const int ly = (int)(t % p.h), b = (int)(t / p.h);
        // output rows y whose source index s(y) = sh (y + 0.5) - 0.5 lies in (ly - 1, ly + 1): a conservative window,
        // the exact membership (i0 == ly or i1 == ly) is re-derived per pixel with the forward's own index function
        int ya = (int)floorf(((float)ly - 1.f + 0.5f) / sh - 0.5f) - 1, yb = (int)ceilf(((float)ly + 1.f + 0.5f) / sh - 0.5f) + 1;
        int xa = (int)floorf(((float)lx - 1.f + 0.5f) / sw - 0.5f) - 1, xb = (int)ceilf(((float)lx + 1.f + 0.5f) / sw - 0.5f) + 1;
        ya = ya < 0 ? 0 : ya;
        xa = xa < 0 ? 0 : xa;
        yb = yb > p.H - 1 ? p.H - 1 : yb;
        xb = xb > p.W - 1 ? p.W - 1 : xb;
        float acc = 0.f;
        for (int y = ya; y <= yb; ++y) {
            int y0, y1;
            float ly0, ly1;
            up_index(y, sh, p.h, y0, y1, ly0, ly1);
            const float wy = (y0 == ly ? ly0 : 0.f) + (y1 == ly ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int x = xa; x <= xb; ++x) {
                if (y >= p.py && y < p.py + p.PS && x >= p.px && x < p.px + p.PS) continue;  // overwritten by the patch
                int x0, x1;
                float lx0, lx1;
                up_index(x, sw, p.w, x0, x1, lx0, lx1);
                const float wx = (x0 == lx ? lx0 : 0.f) + (x1 == lx ? lx1 : 0.f);
                if (wx != 0.f) acc = fmaf(wy * wx, p.g_out[(((long long)b * p.H + y) * p.W + x) * p.C + c], acc);
            }
        }
        p.g_low[el] = acc;
    }
}

static unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

static int check_patch(int B, int h, int w, int H, int W, int C, int PS, int py, int px) {
    if (B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || PS <= 0) return TT_ERR_BAD_ARG;
    if (py < 0 || px < 0 || py + PS > H || px + PS > W) return TT_ERR_BAD_ARG;
    return TT_OK;
}

extern "C" int tt_patch_composite_fwd(const float* low, const float* patch, float* out, int32_t B, int32_t h, int32_t w,
                                      int32_t H, int32_t W, int32_t C, int32_t PS, int32_t py, int32_t px,
                                      void* stream) {
    if (!low || !patch || !out) return TT_ERR_BAD_ARG;
    const int st = check_patch(B, h, w, H, W, C, PS, py, px);
    if (st != TT_OK) return st;
    PatchParams p = {low, patch, out, B, h, w, H, W, C, PS, py, px};
    hipLaunchKernelGGL(k_patch_composite_fwd, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0,
                       (hipStream_t)stream, p);
    return tt_check_launch();
}

extern "C" int tt_patch_composite_bwd(const float* g_out, float* g_low, float* g_patch, int32_t B, int32_t h, int32_t w,
                                      int32_t H, int32_t W, int32_t C, int32_t PS, int32_t py, int32_t px,
                                      void* stream) {
    if (!g_out || !g_patch) return TT_ERR_BAD_ARG;
    const int st = check_patch(B, h, w, H, W, C, PS, py, px);
    if (st != TT_OK) return st;
    PatchBwdParams p = {g_out, g_low, g_patch, B, h, w, H, W, C, PS, py, px};
    const long long n = (long long)B * PS * PS * C + (g_low ? (long long)B * h * w * C : 0);
    hipLaunchKernelGGL(k_patch_composite_bwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p);
    return tt_check_launch();
}
