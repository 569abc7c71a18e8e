"""Drop-in for the reference's `custom/triplaneturbo/extern/grid_sample_gradfix/cuda_gridsample.py`:
`grid_sample_2d(input, grid, padding_mode, align_corners)` with a working double backward on MI355X.
Forward and first backward are stock aten (they exist on ROCm); the second-order step calls the HIP kernel
tt_grid_sample_2d_grad2 instead of the CUDA extension.  Same class structure as the reference (:31-79)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .ops import _chk, _ptr, _stream


def grid_sample_2d(input, grid, padding_mode="zeros", align_corners=True):
    # (defaults as the reference's cuda_gridsample.py:24: align_corners=True; its one call site, geometry/utils.py:23,
    # passes False explicitly)
    assert padding_mode in ["zeros", "border"]
    return _GridSample2dForward.apply(input, grid, padding_mode, align_corners)


_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}  # TT_DTYPE_* (include/tt_abi.h)


def grad2_2d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, padding_mode, align_corners):
    """gridsample_cuda.cpp:26-37 signature; returns [grad_grad_output, grad_input, grad_grid].  half / float / double,
    padding_mode 0 (zeros) or 1 (border) -- the reference passes it as a bool --, either align_corners."""
    dt = input.dtype
    if dt not in _DTYPES:
        raise TypeError(f"grad2_2d: unsupported dtype {dt}")
    input = _chk(input, "input", dtype=dt)
    grid = _chk(grid, "grid", dtype=dt)
    grad_output = _chk(grad_output, "grad_output", dtype=dt)
    g2i = _chk(grad2_grad_input, "grad2_grad_input", input.shape, dtype=dt)
    g2g = _chk(grad2_grad_grid, "grad2_grad_grid", grid.shape, dtype=dt)
    N, C, H, W = input.shape
    M = grid.shape[1] * grid.shape[2]
    ggo = torch.empty_like(grad_output)
    gi = torch.empty_like(input)
    gg = torch.empty_like(grid)
    st = _lib.load().tt_grid_sample_2d_grad2_typed(_DTYPES[dt], _ptr(g2i), _ptr(g2g), _ptr(grad_output), _ptr(input),
                                                   _ptr(grid), N, C, H, W, M, int(padding_mode),
                                                   int(bool(align_corners)), _ptr(ggo), _ptr(gi), _ptr(gg), _stream())
    _lib.check(st, "tt_grid_sample_2d_grad2_typed")
    return [ggo, gi, gg]


class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, padding_mode="zeros", align_corners=True):
        assert input.ndim == 4 and grid.ndim == 4 and input.shape[0] == grid.shape[0] and grid.shape[3] == 2
        output = torch.nn.functional.grid_sample(input=input, grid=grid, mode="bilinear", padding_mode=padding_mode,
                                                 align_corners=align_corners)
        ctx.save_for_backward(input, grid)
        ctx.padding_mode = ["zeros", "border"].index(padding_mode)
        ctx.align_corners = align_corners
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        grad_input, grad_grid = _GridSample2dBackward.apply(grad_output, input, grid, ctx.padding_mode,
                                                            ctx.align_corners)
        return grad_input, grad_grid, None, None


class _GridSample2dBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid, padding_mode=0, align_corners=True):
        mask = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        grad_input, grad_grid = torch.ops.aten.grid_sampler_2d_backward(grad_output, input, grid, 0, padding_mode,
                                                                        align_corners, mask)
        ctx.save_for_backward(grad_output, input, grid)
        ctx.padding_mode = padding_mode
        ctx.align_corners = align_corners
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grad_output, input, grid = ctx.saved_tensors
        if grad2_grad_input is None:
            grad2_grad_input = torch.zeros_like(input)
        if grad2_grad_grid is None:
            grad2_grad_grid = torch.zeros_like(grid)
        ggo, gi, gg = grad2_2d(grad2_grad_input.contiguous(), grad2_grad_grid.contiguous(),
                               grad_output.contiguous(), input.contiguous(), grid.contiguous(), ctx.padding_mode,
                               ctx.align_corners)
        return ggo, gi, gg, None, None
