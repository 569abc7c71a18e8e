// tt_grad2.hip -- operator-level drop-in for the reference's only native op:
//   gridsample_grad2.grad2_2d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners)
//   -> [grad_grad_output, grad_input, grad_grid]       (gridsample_cuda.cpp:26-37,53-54; kernel gridsample_cuda.cu:27-210,
//   dispatch :560-594: half / float / double, zeros or border padding, either align_corners)
// i.e. the backward of aten::grid_sampler_2d_backward (bilinear), needed because stock PyTorch has no double backward
// for grid_sample.  The fused render path of this library does NOT use it (its second-order terms are folded into
// k_decode_bwd_geo / k_points_bwd_x); it is exported so that reference code that still calls `grad2_2d` runs on MI355X.
//
// Formulation (the one k_points_bwd_x uses, not the reference's corner-by-corner temporaries): with the sample position
// p = (x, y) in normalised grid units, the bilinear weights w_k(p) of the four corners k and
//     a_k = d w_k / dx,   b_k = d w_k / dy,   c_k = d^2 w_k / dx dy           (all including the unnormalisation and, for
// border padding, the clip factor; 0 for an out-of-bounds corner), per channel with v_k / g_k the corner values of
// input / grad2_grad_input, go = grad_output and (tx, ty) = grad2_grad_grid:
//     grad_grad_output = sum_k w_k g_k + (a_k tx + b_k ty) v_k
//     grad_input[k]   += (a_k tx + b_k ty) go
//     grad_grid.x     += go (sum_k a_k g_k + ty sum_k c_k v_k),   grad_grid.y += go (sum_k b_k g_k + tx sum_k c_k v_k)
// Tensors are NCHW (that is the op's contract), so a lane owns one output point and walks the channels: per channel
// the (n,c,Ho,Wo) operands are read / written coalesced over the points of a wave; the corner accesses into (n,c,H,W)
// are scattered 4-byte accesses by nature of the layout (the fused path packs planes channels-last for that reason).
// Every in-bounds test is a float 0/1 flag folded into the coefficients (no lane-mask logic, see corners_setup in
// tt_device.h); out-of-bounds corners read a clamped, valid texel with coefficient 0 and add 0 to it.
#include <hip/hip_fp16.h>

#include "tt_device.h"
#include "tt_host.h"

template <typename T>
struct Grad2Params {
    const T* g2_inp;
    const T* g2_grid;
    const T* g_out;
    const T* inp;
    const T* grid;
    int N, C, H, W;
    long long M;  // Ho*Wo
    int border, align;
    T* gg_out;
    T* g_inp;
    T* g_grid;
};

template <typename T>
struct Compute {
    typedef float type;
};
template <>
struct Compute<double> {
    typedef double type;
};

template <typename A>
__device__ __forceinline__ A ld(const float* p) { return (A)*p; }
template <typename A>
__device__ __forceinline__ A ld(const double* p) { return (A)*p; }
template <typename A>
__device__ __forceinline__ A ld(const __half* p) { return (A)__half2float(*p); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(double* p, double v) { *p = v; }
__device__ __forceinline__ void st(__half* p, float v) { *p = __float2half(v); }
__device__ __forceinline__ void add(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void add(double* p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void add(__half* p, float v) {
    // packed-half atomic on the aligned pair that holds the element (the other half gets +0)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const bool odd = (reinterpret_cast<uintptr_t>(p) & 2) != 0;
    h2 x;
    x[0] = odd ? (_Float16)0.f : (_Float16)v;
    x[1] = odd ? (_Float16)v : (_Float16)0.f;
    h2* q = reinterpret_cast<h2*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3);
    __builtin_amdgcn_global_atomic_fadd_v2f16(q, x);
}

// one axis: unnormalise (ATen GridSampler.h grid_sampler_unnormalize_set_grad), optionally clip to [0, size-1]
// (clip_coordinates_set_grad: gradient 0 outside), split into the two taps.  All in-bounds tests are 0/1 factors.
template <typename A>
struct Axis {
    A w0, w1;  // tap weights (0 when the tap is out of bounds)
    A d0, d1;  // d w / d (normalised coordinate)
    A f0, f1;  // in-bounds flags (0/1)
    A mult;    // d (pixel coordinate) / d (normalised coordinate), 0 where the clip is active
    int i0, i1;  // clamped tap indices (always valid)
};
template <typename A>
__device__ __forceinline__ Axis<A> axis_setup(A g, int size, int border, int align) {
    Axis<A> r;
    A mult, ip;
    if (align) {
        mult = (A)(size - 1) / 2;
        ip = ((g + 1) / 2) * (A)(size - 1);
    } else {
        mult = (A)size / 2;
        ip = ((g + 1) * (A)size - 1) / 2;
    }
    if (border) {
        const A hi = (A)(size - 1);
        // (tt_opaque_t: otherwise the compiler folds the product of two 0/1 selects back into s_and_b64 + one select)
        const A inside = tt_opaque_t<A>(ip > 0 ? (A)1 : (A)0) * (ip < hi ? (A)1 : (A)0);
        mult = mult * inside;
        ip = ip < 0 ? (A)0 : (ip > hi ? hi : ip);
    }
    const A fl = floor(ip);
    const A t1 = ip - fl, t0 = (fl + 1) - ip;
    const A lo = (A)-2, up = (A)size + 1;
    const int i = (int)(fl < lo ? lo : (fl > up ? up : fl));
    r.f0 = (unsigned)i < (unsigned)size ? (A)1 : (A)0;
    r.f1 = (unsigned)(i + 1) < (unsigned)size ? (A)1 : (A)0;
    r.w0 = t0 * r.f0;
    r.w1 = t1 * r.f1;
    r.d0 = -(mult * r.f0);
    r.d1 = mult * r.f1;
    r.mult = mult;
    r.i0 = min(max(i, 0), size - 1);
    r.i1 = min(max(i + 1, 0), size - 1);
    return r;
}

#pragma clang fp contract(off)
template <typename T>
__global__ __launch_bounds__(256) void k_grid_sample_2d_grad2(Grad2Params<T> p) {
    typedef typename Compute<T>::type A;
    const long long total = (long long)p.N * p.M;
    const size_t HW = (size_t)p.H * p.W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / p.M, m = idx - n * p.M;
        const Axis<A> ax = axis_setup<A>(ld<A>(p.grid + idx * 2 + 0), p.W, p.border, p.align);
        const Axis<A> ay = axis_setup<A>(ld<A>(p.grid + idx * 2 + 1), p.H, p.border, p.align);
        const A tx = ld<A>(p.g2_grid + idx * 2 + 0), ty = ld<A>(p.g2_grid + idx * 2 + 1);
        // corners k = (x tap, y tap): 00 nw, 10 ne, 01 sw, 11 se
        const A w[4] = {ax.w0 * ay.w0, ax.w1 * ay.w0, ax.w0 * ay.w1, ax.w1 * ay.w1};
        const A a[4] = {ax.d0 * ay.w0, ax.d1 * ay.w0, ax.d0 * ay.w1, ax.d1 * ay.w1};
        const A b[4] = {ax.w0 * ay.d0, ax.w1 * ay.d0, ax.w0 * ay.d1, ax.w1 * ay.d1};
        const A c[4] = {ax.d0 * ay.d0, ax.d1 * ay.d0, ax.d0 * ay.d1, ax.d1 * ay.d1};
        const size_t off[4] = {(size_t)ay.i0 * p.W + ax.i0, (size_t)ay.i0 * p.W + ax.i1, (size_t)ay.i1 * p.W + ax.i0,
                               (size_t)ay.i1 * p.W + ax.i1};
        A t[4];  // coefficient of v_k in grad_grad_output = coefficient of go in grad_input[k]
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = a[k] * tx + b[k] * ty;
        A gx = 0, gy = 0;
        for (int ch = 0; ch < p.C; ++ch) {
            const size_t plane = ((size_t)n * p.C + ch) * HW;
            const size_t o = ((size_t)n * p.C + ch) * p.M + m;
            const A go = ld<A>(p.g_out + o);
            A ggo = 0, sa = 0, sb = 0, sc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const A v = ld<A>(p.inp + plane + off[k]), g = ld<A>(p.g2_inp + plane + off[k]);
                ggo += w[k] * g + t[k] * v;
                sa += a[k] * g;
                sb += b[k] * g;
                sc += c[k] * v;
                add(p.g_inp + plane + off[k], t[k] * go);
            }
            st(p.gg_out + o, ggo);
            gx += go * (sa + ty * sc);
            gy += go * (sb + tx * sc);
        }
        st(p.g_grid + idx * 2 + 0, gx);
        st(p.g_grid + idx * 2 + 1, gy);
    }
}
#pragma clang fp contract(fast)

template <typename T>
static int launch_grad2(const void* g2i, const void* g2g, const void* go, const void* inp, const void* grid, int n, int c,
                        int h, int w, long long m, int padding_mode, int align_corners, void* ggo, void* gi, void* gg,
                        hipStream_t s) {
    Grad2Params<T> p;
    p.g2_inp = (const T*)g2i;
    p.g2_grid = (const T*)g2g;
    p.g_out = (const T*)go;
    p.inp = (const T*)inp;
    p.grid = (const T*)grid;
    p.N = n;
    p.C = c;
    p.H = h;
    p.W = w;
    p.M = m;
    p.border = padding_mode;
    p.align = align_corners;
    p.gg_out = (T*)ggo;
    p.g_inp = (T*)gi;
    p.g_grid = (T*)gg;
    // grad_input accumulates with atomics: start from zero like the reference's zeros_like (gridsample_cuda.cu:553-555)
    if (hipMemsetAsync(gi, 0, (size_t)n * c * h * w * sizeof(T), s) != hipSuccess) return TT_ERR_LAUNCH;
    long long total = (long long)n * m;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_grid_sample_2d_grad2<T>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return tt_check_launch();
}

extern "C" int tt_grid_sample_2d_grad2_typed(int32_t dtype, const void* grad2_grad_input, const void* grad2_grad_grid,
                                             const void* grad_output, const void* input, const void* grid, int32_t n,
                                             int32_t c, int32_t h, int32_t w, int64_t n_points_per_batch,
                                             int32_t padding_mode, int32_t align_corners, void* grad_grad_output,
                                             void* grad_input, void* grad_grid, void* stream) {
    if (!grad2_grad_input || !grad2_grad_grid || !grad_output || !input || !grid || !grad_grad_output || !grad_input ||
        !grad_grid || n <= 0 || c <= 0 || h <= 0 || w <= 0 || n_points_per_batch <= 0)
        return TT_ERR_BAD_ARG;
    // the reference's op takes padding_mode as a bool (gridsample_cuda.cu:542): zeros or border; reflection and the 3-D
    // variant (grad2_3d) are not built -- the reference never calls them
    if (padding_mode < 0 || padding_mode > 1 || align_corners < 0 || align_corners > 1) return TT_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define GO(T)                                                                                                         \
    launch_grad2<T>(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, n, c, h, w, n_points_per_batch,     \
                    padding_mode, align_corners, grad_grad_output, grad_input, grad_grid, s)
    switch (dtype) {
        case TT_DTYPE_F32: return GO(float);
        case TT_DTYPE_F16: return GO(__half);
        case TT_DTYPE_F64: return GO(double);
        default: return TT_ERR_UNSUPPORTED;
    }
#undef GO
}

extern "C" int tt_grid_sample_2d_grad2(const float* grad2_grad_input, const float* grad2_grad_grid,
                                       const float* grad_output, const float* input, const float* grid, int32_t n,
                                       int32_t c, int32_t h, int32_t w, int64_t n_points_per_batch,
                                       int32_t padding_mode, int32_t align_corners, float* grad_grad_output,
                                       float* grad_input, float* grad_grid, void* stream) {
    return tt_grid_sample_2d_grad2_typed(TT_DTYPE_F32, grad2_grad_input, grad2_grad_grid, grad_output, input, grid, n, c,
                                         h, w, n_points_per_batch, padding_mode, align_corners, grad_grad_output,
                                         grad_input, grad_grid, stream);
}
