// tt_forward.hip -- plane pack/unpack, per-point decode (tt_query_points) and the fused forward render.
#include "tt_device.h"
#include "tt_host.h"

// =====================================================================================================
// plane pack: (P,6,32,H,W) NCHW  ->  (P,6,H,W,32) channels-last with rotate_planes "v1" folded in
//   R0[h,w] = P0[w,h]   R1[h,w] = P1[H-1-h, W-1-w]   R2[h,w] = P2[H-1-w, h]   (few_step...:212-225)
// One block per (plane, source row y): the 32 x W slab is read coalesced along x, transposed through
// LDS and written as whole 128-byte texels.
// =====================================================================================================
__device__ __forceinline__ void rot_dst(int k3, int y, int x, int H, int W, int& h, int& w) {
    if (k3 == 0) {
        h = x;
        w = y;
    } else if (k3 == 1) {
        h = H - 1 - y;
        w = W - 1 - x;
    } else {
        h = x;
        w = H - 1 - y;
    }
}

template <bool UNPACK>
__global__ __launch_bounds__(256) void k_planes_pack(const float* __restrict__ src, float* __restrict__ dst, int H,
                                                     int W) {
    extern __shared__ float tile[];  // [32][W+1]
    const int y = blockIdx.x, plane = blockIdx.y;  // plane = p*6 + k
    const int k3 = plane % 3;
    const size_t HW = (size_t)H * W;
    const float* nchw_c = (UNPACK ? dst : src) + (size_t)plane * TT_C * HW;
    float* nchw = const_cast<float*>(nchw_c);
    const float* nhwc_c = (UNPACK ? src : dst) + (size_t)plane * HW * TT_C;
    float* nhwc = const_cast<float*>(nhwc_c);
    const int ws = W + 1;
    if (!UNPACK) {
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int c = e / W, x = e - c * W;
            tile[c * ws + x] = nchw[(size_t)c * HW + (size_t)y * W + x];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int x = e >> 5, c = e & 31, h, w;
            rot_dst(k3, y, x, H, W, h, w);
            nhwc[((size_t)h * W + w) * TT_C + c] = tile[c * ws + x];
        }
    } else {
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int x = e >> 5, c = e & 31, h, w;
            rot_dst(k3, y, x, H, W, h, w);
            tile[c * ws + x] = nhwc[((size_t)h * W + w) * TT_C + c];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TT_C * W; e += blockDim.x) {
            int c = e / W, x = e - c * W;
            nchw[(size_t)c * HW + (size_t)y * W + x] = tile[c * ws + x];
        }
    }
}

// =====================================================================================================
// per-tile decode (forward): gather -> sdf net (+ input-gradient chain) -> feature net
// =====================================================================================================
struct DecodeCfg {
    const float* pbase;  // packed planes of this prompt: 6 x H x W x 32
    int H, W;
    float radius;
    float ju, jv;  // 0.5*W/radius, 0.5*H/radius
};

// outputs are identical in both half-waves.  gq = J^T q (WITHOUT the sphere term).
template <bool NEED_N, bool NEED_TEX>
__device__ __forceinline__ void decode_fwd(const float* L, const DecodeCfg& dc, float px, float py, float pz,
                                           bool valid, int i, int hi, float& s0, float (&gq)[3], float (&c)[3]) {
    const float X = scale_coord(px, dc.radius), Y = scale_coord(py, dc.radius), Z = scale_coord(pz, dc.radius);
    s0 = 0.f;
    gq[0] = gq[1] = gq[2] = 0.f;
    c[0] = c[1] = c[2] = 0.f;
    if (NEED_TEX) {
        float e[48];
        bool any = gather_tex(dc.pbase, dc.H, dc.W, X, Y, Z, valid, hi, e);
        if (__any(any)) {  // exact skip: e == 0 for the whole tile => features == 0 (bias-free MLP)
            float k1[32], k2[32];
            mv_fwd<64, 96>(L + OFF_V1, e, k1, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) k1[r] = fmaxf(k1[r], 0.f);
            mv_fwd<64, 64>(L + OFF_V2, k1, k2, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) k2[r] = fmaxf(k2[r], 0.f);
#pragma unroll
            for (int o = 0; o < 3; ++o) c[o] = dot_lds<64>(L + OFF_V3 + 64 * o, k2, hi);
        }
    }
    {
        float f[16], jx[16], jy[16], jz[16];
        bool any = gather_geo<NEED_N>(dc.pbase, dc.H, dc.W, X, Y, Z, valid, dc.ju, dc.jv, hi, f, jx, jy, jz);
        if (__any(any)) {
            float h1[32], h2[32];
            mv_fwd<64, 32>(L + OFF_W1, f, h1, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
            mv_fwd<64, 64>(L + OFF_W2, h1, h2, i, hi);
#pragma unroll
            for (int r = 0; r < 32; ++r) h2[r] = fmaxf(h2[r], 0.f);
            s0 = dot_lds<64>(L + OFF_W3, h2, hi);
            if (NEED_N) {
                // reverse-mode input gradient: a2 = m2 . w3 ; a1 = m1 . (W2^T a2) ; q = W1^T a1
                float a2[32], a1[32], q[16];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    f32x4 w3 = *reinterpret_cast<const f32x4*>(L + OFF_W3 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) a2[4 * g + e2] = h2[4 * g + e2] > 0.f ? w3[e2] : 0.f;
                }
                mv_bwd<64, 64>(L + OFF_W2, a2, a1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) a1[r] = h1[r] > 0.f ? a1[r] : 0.f;
                mv_bwd<32, 64>(L + OFF_W1, a1, q, i, hi);
                float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sx = fmaf(q[r], jx[r], sx);
                    sy = fmaf(q[r], jy[r], sy);
                    sz = fmaf(q[r], jz[r], sz);
                }
                gq[0] = sx + __shfl_xor(sx, 32);
                gq[1] = sy + __shfl_xor(sy, 32);
                gq[2] = sz + __shfl_xor(sz, 32);
            }
        }
    }
}

// =====================================================================================================
// tt_query_points
// =====================================================================================================
struct QueryParams {
    const float* packed;
    MlpPtrs w;
    const float* points;
    int n_batch;
    long long n_points;
    int views_per_prompt;
    int H, W;
    float radius, bias_radius;
    float* out_sdf;
    float* out_grad;
    float* out_feat;
};

template <bool NEED_N, bool NEED_TEX>
__global__ __launch_bounds__(256, 2) void k_query_points(QueryParams p) {
    __shared__ __attribute__((aligned(16))) float L[LDS_W_FLOATS];
    MlpPtrs w = p.w;
    lds_load_geo_weights(L, w);
    if (NEED_TEX) lds_load_tex_weights(L, w);
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const long long tiles_per_batch = (p.n_points + TT_TILE - 1) / TT_TILE;
    const long long n_tiles = tiles_per_batch * p.n_batch;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const size_t plane_stride = (size_t)6 * p.H * p.W * TT_C;
#pragma nounroll
    for (long long tile = wave0; tile < n_tiles; tile += n_waves) {
        const int b = (int)(tile / tiles_per_batch);
        const long long n = (tile - (long long)b * tiles_per_batch) * TT_TILE + i;
        const bool valid = n < p.n_points;
        const long long idx = (long long)b * p.n_points + (valid ? n : 0);
        DecodeCfg dc;
        dc.pbase = p.packed + (size_t)(b / p.views_per_prompt) * plane_stride;
        dc.H = p.H;
        dc.W = p.W;
        dc.radius = p.radius;
        dc.ju = 0.5f * p.W / p.radius;
        dc.jv = 0.5f * p.H / p.radius;
        const float px = p.points[idx * 3 + 0], py = p.points[idx * 3 + 1], pz = p.points[idx * 3 + 2];
        float s0, gq[3], c[3];
        decode_fwd<NEED_N, NEED_TEX>(L, dc, px, py, pz, valid, i, hi, s0, gq, c);
        float nrm;
        const float sdf = s0 + sphere_bias(px, py, pz, p.bias_radius, nrm);
        if (valid && hi == 0) {
            if (p.out_sdf) p.out_sdf[idx] = sdf;
            if (NEED_N && p.out_grad) {
                p.out_grad[idx * 3 + 0] = gq[0] + px / nrm;
                p.out_grad[idx * 3 + 1] = gq[1] + py / nrm;
                p.out_grad[idx * 3 + 2] = gq[2] + pz / nrm;
            }
            if (NEED_TEX && p.out_feat) {
                p.out_feat[idx * 3 + 0] = c[0];
                p.out_feat[idx * 3 + 1] = c[1];
                p.out_feat[idx * 3 + 2] = c[2];
            }
        }
    }
}

// =====================================================================================================
// fused forward render: one wave per ray, 32 samples per tile
// =====================================================================================================
struct RenderFwdParams {
    const float* packed;
    MlpPtrs w;
    const float* rays_o;
    const float* rays_d;
    const float* t_starts;
    const float* t_ends;
    tt_render_cfg cfg;
    float* opacity;
    float* depth;
    float* rgb_fg;
    float* z_var;
    float* nacc;
    float* weights;
    float* trans;
    float* sdf;
    float* sdf_grad;
    float* features;
};

__device__ __forceinline__ float neus_alpha(float sdf, float cosv, float dt, float inv_std, float ratio) {
    // neus_volume_renderer.py:98-116
    const float iter_cos = -(fmaxf(-cosv * 0.5f + 0.5f, 0.f) * (1.f - ratio) + fmaxf(-cosv, 0.f) * ratio);
    const float next_sdf = sdf + iter_cos * dt * 0.5f;
    const float prev_sdf = sdf - iter_cos * dt * 0.5f;
    const float prev_cdf = sigmoidf_(prev_sdf * inv_std);
    const float next_cdf = sigmoidf_(next_sdf * inv_std);
    const float pp = prev_cdf - next_cdf;
    const float a = (pp + 1e-5f) / (prev_cdf + 1e-5f);
    return fminf(fmaxf(a, 0.f), 1.f);
}

__global__ __launch_bounds__(256, 2) void k_render_fwd(RenderFwdParams p) {
    __shared__ __attribute__((aligned(16))) float L[LDS_W_FLOATS];
    {
        MlpPtrs w = p.w;
        lds_load_geo_weights(L, w);
        lds_load_tex_weights(L, w);
    }
    __syncthreads();
    const tt_render_cfg& cfg = p.cfg;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int S = cfg.n_samples;
    const int n_tiles = (S + TT_TILE - 1) / TT_TILE;
    // XCD-aware ray assignment: block b runs on XCD b % 8; give each XCD one contiguous chunk of rays so the
    // texels its waves touch stay in that XCD's L2.
    const long long n_rays = cfg.n_rays;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long chunk = (n_rays + 7) / 8;
    const long long lo = xcd * chunk, hiR = (lo + chunk < n_rays) ? lo + chunk : n_rays;
    const int waves_per_xcd = (gridDim.x >> 3) * (blockDim.x >> 6);
    const int wv = slot * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t plane_stride = (size_t)6 * cfg.plane_h * cfg.plane_w * TT_C;
    const bool per_sample = (cfg.flags & TT_R_PER_SAMPLE) != 0;

#pragma nounroll
    for (long long ray = lo + wv; ray < hiR; ray += waves_per_xcd) {
        const int view = (int)(ray / cfg.rays_per_view);
        DecodeCfg dc;
        dc.pbase = p.packed + (size_t)(view / cfg.views_per_prompt) * plane_stride;
        dc.H = cfg.plane_h;
        dc.W = cfg.plane_w;
        dc.radius = cfg.radius;
        dc.ju = 0.5f * cfg.plane_w / cfg.radius;
        dc.jv = 0.5f * cfg.plane_h / cfg.radius;
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
        float T = 1.f;
        float a_op = 0.f, a_d = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f, a_nx = 0.f, a_ny = 0.f, a_nz = 0.f;
#pragma nounroll
        for (int tile = 0; tile < n_tiles; ++tile) {
            const int si = tile * TT_TILE + i;
            const bool valid = si < S;
            const long long sidx = ray * S + (valid ? si : 0);
            const float ts = valid ? p.t_starts[sidx] : 0.f, te = valid ? p.t_ends[sidx] : 0.f;
            float tm, px, py, pz;
            sample_position(ox, oy, oz, dx, dy, dz, ts, te, tm, px, py, pz);
            float s0, gq[3], c[3];
            decode_fwd<true, true>(L, dc, px, py, pz, valid, i, hi, s0, gq, c);
            float nrm;
            const float sdf = s0 + sphere_bias(px, py, pz, cfg.sdf_bias_radius, nrm);
            const float gx = gq[0] + px / nrm, gy = gq[1] + py / nrm, gz = gq[2] + pz / nrm;
            const float gn = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize eps
            const float nx = gx / gn, ny = gy / gn, nz = gz / gn;
            const float cosv = dx * nx + dy * ny + dz * nz;
            float alpha = neus_alpha(sdf, cosv, te - ts, cfg.inv_std, cfg.cos_anneal_ratio);
            if (!valid) alpha = 0.f;
            float total;
            const float Ti = T * excl_prod32(1.f - alpha, i, total);
            T *= total;
            const float wgt = alpha * Ti;
            // NoMaterial + sigmoid-mipnerf (no_material.py:41-54, ops.py:118-119)
            const float r = sigmoidf_(c[0]) * 1.002f - 0.001f, g = sigmoidf_(c[1]) * 1.002f - 0.001f,
                        b = sigmoidf_(c[2]) * 1.002f - 0.001f;
            a_op += wgt;
            a_d = fmaf(wgt, tm, a_d);
            a_r = fmaf(wgt, r, a_r);
            a_g = fmaf(wgt, g, a_g);
            a_b = fmaf(wgt, b, a_b);
            a_nx = fmaf(wgt, nx, a_nx);
            a_ny = fmaf(wgt, ny, a_ny);
            a_nz = fmaf(wgt, nz, a_nz);
            if (valid && hi == 0) {
                p.weights[sidx] = wgt;
                p.trans[sidx] = Ti;
                if (per_sample) {
                    if (p.sdf) p.sdf[sidx] = sdf;
                    if (p.sdf_grad) {
                        p.sdf_grad[sidx * 3 + 0] = gx;
                        p.sdf_grad[sidx * 3 + 1] = gy;
                        p.sdf_grad[sidx * 3 + 2] = gz;
                    }
                    if (p.features) {
                        p.features[sidx * 3 + 0] = c[0];
                        p.features[sidx * 3 + 1] = c[1];
                        p.features[sidx * 3 + 2] = c[2];
                    }
                }
            }
        }
        a_op = half_sum(a_op);
        a_d = half_sum(a_d);
        a_r = half_sum(a_r);
        a_g = half_sum(a_g);
        a_b = half_sum(a_b);
        a_nx = half_sum(a_nx);
        a_ny = half_sum(a_ny);
        a_nz = half_sum(a_nz);
        // z_variance = sum w (t - depth)^2 (renderer :424-431): second pass over this lane's own weights
        float zv = 0.f;
        for (int tile = 0; tile < n_tiles; ++tile) {
            const int si = tile * TT_TILE + i;
            if (si < S && hi == 0) {
                const long long sidx = ray * S + si;
                const float tm = (p.t_starts[sidx] + p.t_ends[sidx]) / 2.f;
                const float dd = tm - a_d;
                zv = fmaf(p.weights[sidx], dd * dd, zv);
            }
        }
        zv = half_sum(zv);
        if (lane == 0) {
            p.opacity[ray] = a_op;
            p.depth[ray] = a_d;
            p.rgb_fg[ray * 3 + 0] = a_r;
            p.rgb_fg[ray * 3 + 1] = a_g;
            p.rgb_fg[ray * 3 + 2] = a_b;
            p.z_var[ray] = zv;
            p.nacc[ray * 3 + 0] = a_nx;
            p.nacc[ray * 3 + 1] = a_ny;
            p.nacc[ray * 3 + 2] = a_nz;
        }
    }
}

// =====================================================================================================
// host side (C ABI)
// =====================================================================================================
static MlpPtrs to_ptrs(const tt_mlp_weights* w) {
    MlpPtrs m;
    m.w1 = w->w1;
    m.w2 = w->w2;
    m.w3 = w->w3;
    m.v1 = w->v1;
    m.v2 = w->v2;
    m.v3 = w->v3;
    return m;
}

extern "C" int tt_planes_pack(const float* space_cache, float* packed, int32_t n_prompts, int32_t plane_h,
                              int32_t plane_w, void* stream) {
    if (!space_cache || !packed || n_prompts <= 0 || plane_h <= 0 || plane_w <= 0) return TT_ERR_BAD_ARG;
    if (plane_h != plane_w) return TT_ERR_UNSUPPORTED;
    dim3 grid(plane_h, n_prompts * 6);
    size_t lds = (size_t)TT_C * (plane_w + 1) * sizeof(float);
    hipLaunchKernelGGL(k_planes_pack<false>, grid, dim3(256), lds, (hipStream_t)stream, space_cache, packed, plane_h,
                       plane_w);
    return tt_check_launch();
}

extern "C" int tt_planes_unpack_grad(const float* grad_packed, float* grad_space_cache, int32_t n_prompts,
                                     int32_t plane_h, int32_t plane_w, void* stream) {
    if (!grad_packed || !grad_space_cache || n_prompts <= 0 || plane_h <= 0 || plane_w <= 0) return TT_ERR_BAD_ARG;
    if (plane_h != plane_w) return TT_ERR_UNSUPPORTED;
    dim3 grid(plane_h, n_prompts * 6);
    size_t lds = (size_t)TT_C * (plane_w + 1) * sizeof(float);
    hipLaunchKernelGGL(k_planes_pack<true>, grid, dim3(256), lds, (hipStream_t)stream, grad_packed, grad_space_cache,
                       plane_h, plane_w);
    return tt_check_launch();
}

extern "C" int tt_query_points(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                               int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                               int32_t plane_w, float radius, float sdf_bias_radius, int32_t flags, float* out_sdf,
                               float* out_sdf_grad, float* out_features, void* stream) {
    if (!packed || !w || !points || n_batch <= 0 || n_points <= 0 || n_prompts <= 0 || views_per_prompt <= 0)
        return TT_ERR_BAD_ARG;
    if (n_batch != n_prompts * views_per_prompt || !(radius > 0.f)) return TT_ERR_BAD_ARG;
    if (plane_h != plane_w || plane_h <= 0) return TT_ERR_UNSUPPORTED;
    const bool need_n = (flags & TT_Q_NORMAL) != 0, need_t = (flags & TT_Q_TEX) != 0;
    if (!w->w1 || !w->w2 || !w->w3 || (need_t && (!w->v1 || !w->v2 || !w->v3))) return TT_ERR_BAD_ARG;
    QueryParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.points = points;
    p.n_batch = n_batch;
    p.n_points = n_points;
    p.views_per_prompt = views_per_prompt;
    p.H = plane_h;
    p.W = plane_w;
    p.radius = radius;
    p.bias_radius = sdf_bias_radius;
    p.out_sdf = out_sdf;
    p.out_grad = out_sdf_grad;
    p.out_feat = out_features;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    long long n_tiles = ((n_points + TT_TILE - 1) / TT_TILE) * n_batch;
    long long blocks = (n_tiles + 3) / 4;
    if (blocks > 2LL * cus) blocks = 2LL * cus;
    dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (need_n && need_t)
        hipLaunchKernelGGL((k_query_points<true, true>), grid, block, 0, s, p);
    else if (need_n)
        hipLaunchKernelGGL((k_query_points<true, false>), grid, block, 0, s, p);
    else if (need_t)
        hipLaunchKernelGGL((k_query_points<false, true>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((k_query_points<false, false>), grid, block, 0, s, p);
    return tt_check_launch();
}

int tt_validate_cfg(const tt_render_cfg* cfg) {
    if (!cfg) return TT_ERR_BAD_ARG;
    if (cfg->n_prompts <= 0 || cfg->views_per_prompt <= 0 || cfg->rays_per_view <= 0 || cfg->n_samples <= 0 ||
        cfg->n_rays <= 0)
        return TT_ERR_BAD_ARG;
    if (cfg->n_rays != (int64_t)cfg->n_prompts * cfg->views_per_prompt * cfg->rays_per_view) return TT_ERR_BAD_ARG;
    if (cfg->plane_h <= 0 || cfg->plane_h != cfg->plane_w) return TT_ERR_UNSUPPORTED;
    if (!(cfg->radius > 0.f) || !(cfg->inv_std > 0.f)) return TT_ERR_BAD_ARG;
    return TT_OK;
}

extern "C" int tt_render_fwd(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                             const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, float* opacity,
                             float* depth, float* rgb_fg, float* z_variance, float* normal_acc, float* weights,
                             float* trans, float* sdf, float* sdf_grad, float* features, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !opacity || !depth || !rgb_fg || !z_variance ||
        !normal_acc || !weights || !trans)
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !w->v1 || !w->v2 || !w->v3) return TT_ERR_BAD_ARG;
    RenderFwdParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.opacity = opacity;
    p.depth = depth;
    p.rgb_fg = rgb_fg;
    p.z_var = z_variance;
    p.nacc = normal_acc;
    p.weights = weights;
    p.trans = trans;
    p.sdf = sdf;
    p.sdf_grad = sdf_grad;
    p.features = features;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    long long blocks = 2LL * cus;  // 2 workgroups of 4 waves per CU (LDS 69 KB each)
    long long need = (cfg->n_rays + 3) / 4;
    if (blocks > need) blocks = need;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL(k_render_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return tt_check_launch();
}
