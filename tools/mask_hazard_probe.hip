// Negative control of tools/mask_hazard_lint.py: the TEXTBOOK in-bounds logic that corners_setup (csrc/tt_device.h)
// replaced -- `in = bx && by; w = in ? wx * wy : 0` -- which hipcc lowers to v_cmp -> s_and_b64 -> v_cndmask, the shape
// that delivered stale lane-mask bits on MI355X.  tests/test_host_logic.py compiles this file and
// checks that the lint FLAGS it (and that it does not flag the product library).
#include <hip/hip_runtime.h>

extern "C" __global__ void k_textbook_corners(const float* __restrict__ gx, const float* __restrict__ gy, int H, int W,
                                              float* __restrict__ w_out, int* __restrict__ off_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const float ix = ((gx[t] + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy[t] + 1.f) * (float)H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
    const bool in0 = bx0 && by0, in1 = bx1 && by0, in2 = bx0 && by1, in3 = bx1 && by1;
    w_out[4 * t + 0] = in0 ? wx0 * wy0 : 0.f;
    w_out[4 * t + 1] = in1 ? wx1 * wy0 : 0.f;
    w_out[4 * t + 2] = in2 ? wx0 * wy1 : 0.f;
    w_out[4 * t + 3] = in3 ? wx1 * wy1 : 0.f;
    off_out[4 * t + 0] = in0 ? y0 * W + x0 : 0;
    off_out[4 * t + 1] = in1 ? y0 * W + x0 + 1 : 0;
    off_out[4 * t + 2] = in2 ? (y0 + 1) * W + x0 : 0;
    off_out[4 * t + 3] = in3 ? (y0 + 1) * W + x0 + 1 : 0;
}
