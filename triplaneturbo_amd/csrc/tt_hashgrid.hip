// tt_hashgrid.hip -- multiresolution hash encoding of 3-D points in [0,1]^3 (forward + gradient w.r.t. the table):
// the `HashGrid` encoding the reference's background obtains from tiny-cuda-nn
//   custom/triplaneturbo/models/background/multi_prompt_neural_environment_hashgrid_map_background.py:25-34,54,104-105
//   threestudio/models/networks.py:17-26,54-64 (TCNNEncoding / get_encoding)
// tiny-cuda-nn is an un-vendored CUDA-only dependency; this follows its published algorithm (Mueller et al. 2022,
// tcnn grid.h: grid_scale / grid_resolution / pos_fract / grid_index / coherent prime hash, Linear interpolation),
// in fp32 (tcnn computes it in fp16).  One thread per (point, level); the 8 corners of a cell are float2/float4
// table reads; the backward is 8 x F atomics per (point, level).  Per-ray work (n_rays points per render), far off
// the per-sample hot path.
#include <math.h>

#include "tt_device.h"
#include "tt_host.h"

#pragma clang fp contract(off)  // pos = x*scale + 0.5 and the corner weights exactly as written

#define TT_HG_MAX_LEVELS 16

struct HashLevels {
    int n_levels, n_features;
    unsigned offset[TT_HG_MAX_LEVELS];  // in table entries
    unsigned size[TT_HG_MAX_LEVELS];
    unsigned res[TT_HG_MAX_LEVELS];
    unsigned dense[TT_HG_MAX_LEVELS];
    float scale[TT_HG_MAX_LEVELS];
    long long total;  // table entries over all levels
};

static int make_levels(const tt_hashgrid_cfg* c, HashLevels* h) {
    if (!c || c->n_levels <= 0 || c->n_levels > TT_HG_MAX_LEVELS) return TT_ERR_BAD_ARG;
    if (!(c->n_features_per_level == 1 || c->n_features_per_level == 2 || c->n_features_per_level == 4 ||
          c->n_features_per_level == 8))
        return TT_ERR_UNSUPPORTED;
    if (c->log2_hashmap_size < 3 || c->log2_hashmap_size > 28 || c->base_resolution <= 0 ||
        !(c->per_level_scale >= 1.f))
        return TT_ERR_BAD_ARG;
    h->n_levels = c->n_levels;
    h->n_features = c->n_features_per_level;
    // tcnn: scale_l = exp2f(l * log2f(per_level_scale)) * base - 1; evaluated here in double and rounded once, so
    // that the value does not depend on a libm's fp32 last-ulp behaviour (1 ulp of scale moves every sample of the
    // 256-cell level by 3e-5 cells)
    const double l2 = log2((double)c->per_level_scale);
    unsigned long long offset = 0;
    for (int l = 0; l < c->n_levels; ++l) {
        const float scale = (float)(exp2((double)l * l2) * (double)c->base_resolution - 1.0);
        const unsigned long long res = (unsigned long long)ceilf(scale) + 1ull;
        const unsigned long long cube = res * res * res;
        const unsigned long long cap = 1ull << c->log2_hashmap_size;
        unsigned long long size = (cube + 7ull) / 8ull * 8ull;
        if (res > 2048ull || size > cap) size = cap;
        h->offset[l] = (unsigned)offset;
        h->size[l] = (unsigned)size;
        h->res[l] = (unsigned)res;
        h->dense[l] = cube <= size ? 1u : 0u;
        h->scale[l] = scale;
        offset += size;
        if (offset > 0x7fffffffull) return TT_ERR_UNSUPPORTED;
    }
    h->total = (long long)offset;
    return TT_OK;
}

__device__ __forceinline__ unsigned hg_index(const HashLevels& h, int l, unsigned x, unsigned y, unsigned z) {
    unsigned idx;
    if (h.dense[l])
        idx = x + y * h.res[l] + z * h.res[l] * h.res[l];
    else
        idx = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return idx % h.size[l];
}

template <int F, bool BWD>
__global__ __launch_bounds__(256) void k_hashgrid(HashLevels h, const float* __restrict__ x, long long n,
                                                  const float* __restrict__ table, const float* __restrict__ g_out,
                                                  float* __restrict__ out_or_grad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= n) return;
    const float scale = h.scale[l];
    float frac[3];
    unsigned cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float pos = x[i * 3 + d] * scale + 0.5f;
        const float fl = floorf(pos);
        cell[d] = (unsigned)(int)fl;
        frac[d] = pos - fl;
    }
    const int width = h.n_levels * F;
    float acc[F];
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        acc[f] = 0.f;
        g[f] = BWD ? g_out[i * width + l * F + f] : 0.f;
    }
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        float w = 1.f;
        unsigned c3[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (corner & (1 << d)) {
                w = w * frac[d];
                c3[d] = cell[d] + 1u;
            } else {
                w = w * (1.f - frac[d]);
                c3[d] = cell[d];
            }
        }
        const size_t e = ((size_t)h.offset[l] + hg_index(h, l, c3[0], c3[1], c3[2])) * F;
        if (BWD) {
#pragma unroll
            for (int f = 0; f < F; ++f) atomicAdd(out_or_grad + e + f, w * g[f]);
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = acc[f] + w * table[e + f];
        }
    }
    if (!BWD) {
#pragma unroll
        for (int f = 0; f < F; ++f) out_or_grad[i * width + l * F + f] = acc[f];
    }
}

template <bool BWD>
static int launch(const HashLevels& h, const float* x, long long n, const float* table, const float* g_out,
                  float* dst, hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)h.n_levels), blk(256);
    switch (h.n_features) {
        case 1: hipLaunchKernelGGL((k_hashgrid<1, BWD>), grid, blk, 0, s, h, x, n, table, g_out, dst); break;
        case 2: hipLaunchKernelGGL((k_hashgrid<2, BWD>), grid, blk, 0, s, h, x, n, table, g_out, dst); break;
        case 4: hipLaunchKernelGGL((k_hashgrid<4, BWD>), grid, blk, 0, s, h, x, n, table, g_out, dst); break;
        default: hipLaunchKernelGGL((k_hashgrid<8, BWD>), grid, blk, 0, s, h, x, n, table, g_out, dst); break;
    }
    return tt_check_launch();
}

extern "C" int64_t tt_hashgrid_n_params(const tt_hashgrid_cfg* cfg) {
    HashLevels h;
    const int st = make_levels(cfg, &h);
    if (st != TT_OK) return st;
    return (int64_t)h.total * h.n_features;
}

extern "C" int tt_hashgrid_fwd(const float* x, int64_t n, const float* params, const tt_hashgrid_cfg* cfg, float* out,
                               void* stream) {
    HashLevels h;
    const int st = make_levels(cfg, &h);
    if (st != TT_OK) return st;
    if (!x || !params || !out || n <= 0 || (n + 255) / 256 > 0x7fffffffLL) return TT_ERR_BAD_ARG;
    return launch<false>(h, x, n, params, nullptr, out, (hipStream_t)stream);
}

extern "C" int tt_hashgrid_bwd(const float* x, int64_t n, const float* g_out, const tt_hashgrid_cfg* cfg,
                               float* grad_params, void* stream) {
    HashLevels h;
    const int st = make_levels(cfg, &h);
    if (st != TT_OK) return st;
    if (!x || !g_out || !grad_params || n <= 0 || (n + 255) / 256 > 0x7fffffffLL) return TT_ERR_BAD_ARG;
    return launch<true>(h, x, n, nullptr, g_out, grad_params, (hipStream_t)stream);
}
