// tt_backward.hip -- backward of the fused render, split by network so each kernel's persistent
// weight-gradient accumulators fit the register file beside its working set:
//
//   k_decode_bwd_geo : consumes (d/d sdf, d/d sdf_grad) per sample from k_march_bwd (tt_march.hip), recomputes
//       the geometry decode per 32-sample tile and back-propagates through BOTH the value chain and the
//       input-gradient chain of the sdf net (the reference's second-order path: gridsample_cuda.cu:27-210 +
//       aten grid_sampler_2d_backward + transposed GEMMs), accumulates dW1/dW2 with MFMA outer products
//       (K = the 32 samples of the tile) and scatters d/d planes 0..2.
//   k_decode_bwd_tex : feature net backward, dV1/dV2/dV3, scatter of d/d planes 3..5.
// Both are purely per-sample: tiles are 32 adjacent rays at one sample index (see tt_device.h).
//
// Everything per-sample is RECOMPUTED from the planes; the forward saves only trans / weights / features.
#include "tt_backward_common.h"
#ifndef TT_GEO_REREAD_RAY
#define TT_GEO_REREAD_RAY 0
#endif

// =====================================================================================================
// geometry half
// =====================================================================================================
struct BwdGeoParams {
    const float* packed;
    MlpPtrs w;
    const float* rays_o;
    const float* rays_d;
    const float* t_starts;
    const float* t_ends;
    tt_render_cfg cfg;
    TileGeom geom;
    long long n_items;
    int* queue;  // per-XCD item counters (tt_queue_counters)
    const float* ws;  // (n_rays*S, 4): d/d sdf, d/d sdf_grad xyz  (from k_march_bwd)
    int n_copies;     // privatised copies of grad_packed
    float* grad_packed;
    MlpGradPtrs grads;
    unsigned long long* phase_cycles;  // tuning build only (TT_PHASE), else null
};

#define GEO_SCRATCH_FLOATS (2 * 64 * XS)
// split-fp16 weight images (tt_mfma16.h): W1, W2 at their fp32 offsets (same bytes); the transposed products use
// transposed COPIES appended to them (TT_BWD_WT_COPIES, tt_backward_common.h: 26 KB of LDS nothing else wants at one wave
// per SIMD) or, without, read the forward images through ds_read_b64_tr_b16 (mv16t)
#define GOFF_W1T LDS_GEO_FLOATS
#define GOFF_W2T (GOFF_W1T + IMG16_FLOATS(32, 64))
#define LDS_GEO16_FLOATS (TT_BWD_WT_COPIES ? GOFF_W2T + IMG16_FLOATS(64, 64) : LDS_GEO_FLOATS)
#define GEO_PAIR (TT_BWD_WT_COPIES ? PAIR_SEQ : PAIR_TR)
// PREC_S3: the images of the third terms, appended (28 KB with the transposed copies: 157 KB per workgroup in all)
#define GLO_W1 LDS_GEO16_FLOATS
#define GLO_W2 (GLO_W1 + LO16_FLOATS(64, 32))
#define GLO_W1T (GLO_W2 + LO16_FLOATS(64, 64))
#define GLO_W2T (GLO_W1T + (TT_BWD_WT_COPIES ? LO16_FLOATS(32, 64) : 0))
#define LDS_GEO3_FLOATS (GLO_W2T + (TT_BWD_WT_COPIES ? LO16_FLOATS(64, 64) : 0))
template <int PREC>
struct GeoWFloats {
    static constexpr int value = PREC == PREC_S3 ? LDS_GEO3_FLOATS : LDS_GEO16_FLOATS;
};

// STATS: the work accounting (tt_render_cfg.stats) compiled in.  In this kernel even a never-taken scalar branch per
// counting site costs 2-3 % (2.95 vs 2.85 ms: the branches cut hipcc's scheduling regions), so production launches
// (stats == null) run the instantiation without it.
template <int PREC, bool WG16, bool STATS = false>
__global__ __launch_bounds__(256, 1) void k_decode_bwd_geo(BwdGeoParams p) {
    constexpr bool EXACT = PREC == PREC_F32;
    constexpr int NT = PrecNT<PREC>::value, WF = GeoWFloats<PREC>::value;
    __shared__ __attribute__((aligned(16))) float L[WF + 4 * (GEO_SCRATCH_FLOATS + SCATTER_TAG_INTS)];
    {
        MlpPtrs w = p.w;
        stage_weights<PREC, 64, 32>(L + OFF_W1, L + GLO_W1, w.w1);
        stage_weights<PREC, 64, 64>(L + OFF_W2, L + GLO_W2, w.w2);
        lds_load_matrix(L + OFF_W3, w.w3, 1, 64, 64);
        if (TT_BWD_WT_COPIES) {
            stage_weights_t<PREC, 64, 32>(L + GOFF_W1T, L + GLO_W1T, w.w1);
            stage_weights_t<PREC, 64, 64>(L + GOFF_W2T, L + GLO_W2T, w.w2);
        }
    }
    const tt_render_cfg& cfg = p.cfg;
    // ---- per-launch operand scales of the fp16 outer products dW1 += a1 u^T, dW2 += a2 v^T (wgrad16 above) ----
    // rigorous magnitude bounds from the weights and the launch's maxima (planes, upstream: reduced on the stream in front
    // of this kernel into the queue slot, tt_host.h):   |a2| <= max |w3|,   |a1_j| <= sum_i |W2[i][j]| |w3_i|,
    //   |f| <= 3 P,  |h1| <= max_i ||W1_i||_1 3 P,  |u| = |sum_corners coef texel| <= 3 P (Sb + 2 (ju + jv) Gb)  (the four
    //   bilinear weights of a plane sum to <= 1, their derivatives to <= 2 per axis),  |qbar| = |u - sbar f| <= 3 P 2 (ju +
    //   jv) Gb,  |b1bar| <= max_i ||W1_i||_1 |qbar|,  |v| = |sbar h1 + b1bar|.
    float sA1 = 1.f, sU = 1.f, sA2 = 1.f, sV = 1.f;
    if (WG16) {
        const unsigned* bnd = reinterpret_cast<const unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        const float Pm = __builtin_bit_cast(float, bnd[TT_BOUND_PLANES]), Sb = __builtin_bit_cast(float, bnd[TT_BOUND_UP0]),
                    Gb = __builtin_bit_cast(float, bnd[TT_BOUND_UP1]);
        unsigned* word = reinterpret_cast<unsigned*>(L + WF);  // scratch is free until the main loop
        const int t = threadIdx.x;
        float w1row = 0.f, a1col = 0.f, w3abs = 0.f;
        if (t < 64) {
            for (int c = 0; c < 32; ++c) w1row += __builtin_fabsf(p.w.w1[t * 32 + c]);
            for (int r = 0; r < 64; ++r) a1col += __builtin_fabsf(p.w.w2[r * 64 + t]) * __builtin_fabsf(p.w.w3[r]);
            w3abs = __builtin_fabsf(p.w.w3[t]);
        }
        const float W1max = block_max(w1row, word), A1max = block_max(a1col, word), A2max = block_max(w3abs, word);
        const float jsum = (0.5f * cfg.plane_w + 0.5f * cfg.plane_h) / cfg.radius;
        const float Fmax = 3.f * Pm, H1max = W1max * Fmax;
        const float Umax = Fmax * (Sb + 2.f * jsum * Gb), QBmax = Fmax * 2.f * jsum * Gb;
        const float Vmax = Sb * H1max + W1max * QBmax;
        sA1 = wg16_scale(A1max);
        sU = wg16_scale(Umax);
        sA2 = wg16_scale(A2max);
        sV = wg16_scale(Vmax);
    }
    const TileGeom& tg = p.geom;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5, wave_in_blk = threadIdx.x >> 6;
    float* Xs = L + WF + wave_in_blk * (GEO_SCRATCH_FLOATS + SCATTER_TAG_INTS);
    float* Ys = Xs + 64 * XS;
    int* tags = reinterpret_cast<int*>(Ys + 64 * XS);
    scatter_init_tags(tags, lane);
    __syncthreads();
    const int S = cfg.n_samples;
    ItemQueue iq = item_queue(p.queue, tg.n_blocks, tg.n_chunks, tg.unit);
    const int H = cfg.plane_h, W = cfg.plane_w;
    const size_t HW = (size_t)H * W;
    const size_t plane_stride = 6 * HW * TT_C;
    const float ju = 0.5f * W / cfg.radius, jv = 0.5f * H / cfg.radius;
    // privatised gradient planes: this workgroup scatters into copy (blockIdx % n_copies); the copies are summed by
    // tt_planes_unpack_grad.  Spreads same-texel atomics (which serialise at the memory side) over n_copies addresses.
    float* const grad_out =
        p.grad_packed + (size_t)(blockIdx.x % (unsigned)p.n_copies) * cfg.n_prompts * plane_stride;
    // (tuning build, TT_DBG_NO_ATOMICS: num_records = 0 -- every flush atomic is still issued and then dropped by the range check)
    const unsigned grad_bytes =
        TT_DBG(cfg.flags, TT_DBG_NO_ATOMICS) ? 0u : (unsigned)(cfg.n_prompts * plane_stride * sizeof(float));  // one copy, < 4 GB - 256

    const TileStats st = STATS ? tile_stats(cfg.stats) : TileStats{nullptr};
    f32x16 accW1[2][1] = {{ZERO16}, {ZERO16}};
    f32x16 accW2[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};
    float accw3 = 0.f;
#ifdef TT_TUNING
    unsigned long long ph_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif

#pragma nounroll
    for (;;) {
        long long b;
        int ck;
        if (!item_pop(iq, tg.order, tg.n_chunks, b, ck)) break;
        if (b >= tg.n_blocks) continue;  // padding of the ragged last deal round
        bool ray_ok;
        const long long ray = tile_ray(tg, b, i, ray_ok);
        const float ray_okf = tt_opaque(ray_ok ? 1.f : 0.f);  // 0/1 factor the compiler cannot fold back into a mask
        const int ks = i % tg.sb;  // this lane's sample offset inside a tile step
        const int view = (int)(ray / cfg.rays_per_view);
        const size_t pofs = (size_t)(view / cfg.views_per_prompt) * plane_stride;
#if !TT_GEO_REREAD_RAY
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        // rays_d == null: explicit points (tt_points_bwd_*), x = rays_o exactly
        const float dx = p.rays_d ? p.rays_d[ray * 3 + 0] : 0.f, dy = p.rays_d ? p.rays_d[ray * 3 + 1] : 0.f,
                    dz = p.rays_d ? p.rays_d[ray * 3 + 2] : 0.f;
#endif
        const int s_end = (ck + 1) * tg.chunk < S ? (ck + 1) * tg.chunk : S;
        // per-step inputs are prefetched one tile step ahead (see k_decode_bwd_tex); a step past the chunk reads a
        // clamped, valid address
        struct StepIn {
            f32x4 up;  // upstream (from the march backward): d/d sdf and d/d sdf_grad of this sample
            float ts, te;
        };
        auto load_step = [&](int sb0) {
            StepIn r;
            const int si = sb0 + ks;
            const long long sidx = ray * S + (si < S ? si : S - 1);
            r.up = *reinterpret_cast<const f32x4*>(p.ws + sidx * 4);
            r.ts = p.rays_d ? p.t_starts[sidx] : 0.f;
            r.te = p.rays_d ? p.t_ends[sidx] : 0.f;
            return r;
        };
        StepIn in = load_step(ck * tg.chunk), in_next;
#pragma nounroll
        for (int sb0 = ck * tg.chunk; sb0 < s_end; sb0 += tg.sb, in = in_next) {
            in_next = load_step(sb0 + tg.sb);
            const int si = sb0 + ks;
            const bool rvalid = ray_ok && si < s_end;
            // validity as a 0/1 FACTOR built from single compares (x * 1 = x, finite * 0 = 0): no select on a lane mask
            // that the scalar ALU has just combined (see corners_setup in tt_device.h)
            const float vf = ray_okf * (si < s_end ? 1.f : 0.f);
            const float sbar = in.up[0] * vf, gbx = in.up[1] * vf, gby = in.up[2] * vf, gbz = in.up[3] * vf;
            tile_stat(st, TT_STAT_VISITED);
            TT_PHASE(0);
            // exact with skip_eps_geo = 0 (the default); > 0: the opt-in approximation of tt_abi.h.  (!(x <= eps): a NaN
            // upstream is never skipped)
            if (!__any(!((__builtin_fabsf(sbar) + __builtin_fabsf(gbx)) + (__builtin_fabsf(gby) + __builtin_fabsf(gbz)) <=
                         cfg.skip_eps_geo)))
                continue;
            float tm, px, py, pz;
#if TT_GEO_REREAD_RAY
            // (per-ray constants re-read per tile step, as in k_decode_bwd_tex: dev A/B, off by default)
            long long rr = ray;
            asm volatile("" : "+v"(rr));
            const float ox = p.rays_o[rr * 3 + 0], oy = p.rays_o[rr * 3 + 1], oz = p.rays_o[rr * 3 + 2];
            const float dx = p.rays_d ? p.rays_d[rr * 3 + 0] : 0.f, dy = p.rays_d ? p.rays_d[rr * 3 + 1] : 0.f,
                        dz = p.rays_d ? p.rays_d[rr * 3 + 2] : 0.f;
#endif
            sample_position(ox, oy, oz, dx, dy, dz, in.ts, in.te, tm, px, py, pz);
            const float X = scale_coord(px, cfg.radius), Y = scale_coord(py, cfg.radius),
                        Z = scale_coord(pz, cfg.radius);
            // ---- recompute the geometry decode ----
            float f[16], u[16];  // u = sbar f + J gbar
            bool anyp[3];
            const bool any = __any(gather_geo_bwd_c(p.packed, (unsigned)(pofs / TT_C), H, W, X, Y, Z, rvalid, sbar, gbx,
                                                    gby, gbz, ju, jv, lane, Xs, f, u, anyp,
                                                    tile_stat_ptr(st, TT_STAT_INBOUNDS)));
            TT_PHASE(1);
            if (!any) continue;  // exact: no in-bounds texel => f = J = 0, every mask false
            tile_stat(st, TT_STAT_EXECUTED);
            // h1, h2 (and a1 under WG16) stay in RAW form: accumulators + a per-lane power-of-two factor (tt_mfma16.h,
            // "deferred factors"); their consumers are signs, the next product, and fmas that take the factor on the scalar
            float h1[32], h2[32], a2[32], a1[32], q[16], u1, u2;
            mvx<PREC, 64, 32, true>(L + OFF_W1, L + GLO_W1, f, h1, i, hi, 1.f, &u1);
#pragma unroll
            for (int r = 0; r < 32; ++r) h1[r] = fmaxf(h1[r], 0.f);
            mvx<PREC, 64, 64, true>(L + OFF_W2, L + GLO_W2, h1, h2, i, hi, u1, &u2);
#pragma unroll
            for (int r = 0; r < 32; ++r) h2[r] = fmaxf(h2[r], 0.f);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f32x4 w3 = *reinterpret_cast<const f32x4*>(L + OFF_W3 + 8 * g + 4 * hi);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) a2[4 * g + e2] = h2[4 * g + e2] > 0.f ? w3[e2] : 0.f;
            }
            // a2 and a1 feed a product AND an outer product (dW2, dW1): split once under the per-launch scales
            Split16<64, GEO_PAIR, NT> a2s, a1s;  // (consumed by the transposed products and the outer-product staging)
            float ua1 = 1.f;                 // factor of a1 where it is RAW
            if (WG16) {
                split16_vec<64, GEO_PAIR, NT>(a2, sA2, a2s);
#if TT_BWD_WT_COPIES
                mv16_pre<64, 64, true, NT>(L + GOFF_W2T, a2s, 1.f / sA2, a1, i, hi, &ua1, L + GLO_W2T);
#else
                mv16t_pre<64, 64, 64, NT>(L + OFF_W2, 0, a2s, 1.f / sA2, a1, lane, L + GLO_W2);
#endif
            } else if constexpr (TT_BWD_WT_COPIES) {
                mvtx_copy<PREC, 64, 64, 64>(L + GOFF_W2T, L + GLO_W2T, L + OFF_W2, a2, a1, i, hi);
            } else {
                mvtx<PREC, 64, 64, 64>(L + OFF_W2, L + GLO_W2, 0, a2, a1, i, hi);
            }
#pragma unroll
            for (int r = 0; r < 32; ++r) a1[r] = h1[r] > 0.f ? a1[r] : 0.f;
            if (WG16) {
                split16_vec<64, GEO_PAIR, NT>(a1, sA1 * ua1, a1s);
#if TT_BWD_WT_COPIES
                mv16_pre<32, 64, false, NT>(L + GOFF_W1T, a1s, 1.f / sA1, q, i, hi, nullptr, L + GLO_W1T);
#else
                mv16t_pre<32, 64, 32, NT>(L + OFF_W1, 0, a1s, 1.f / sA1, q, lane, L + GLO_W1);
#endif
            } else if constexpr (TT_BWD_WT_COPIES) {
                mvtx_copy<PREC, 32, 64, 32>(L + GOFF_W1T, L + GLO_W1T, L + OFF_W1, a1, q, i, hi);
            } else {
                mvtx<PREC, 32, 64, 32>(L + OFF_W1, L + GLO_W1, 0, a1, q, i, hi);
            }
            TT_PHASE(3);
            // ---- network + plane gradients ----
            {
                float qb[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) qb[r] = fmaf(-sbar, f[r], u[r]);  // qbar = J gbar = u - sbar f
                // (no opaque scheduling region here, unlike the texture kernel: with the outer products on the fp16 pipe
                // this kernel has register slack and one scheduling region is faster: 3.22 -> 3.14 ms)
                const bool region = WG16 ? true : cfg.flags >= 0;  // (always true; opaque to the compiler unless WG16)
                const bool do_wgrad = region && !TT_DBG(cfg.flags, TT_DBG_NO_WGRAD);
                // dW1 += a1 (sbar f + qbar)^T
                if (do_wgrad) {
                    if (WG16) {
                        stage_rows16_pre<64>(Xs, a1s, i, hi);
                        stage_rows16<32>(Ys, u, i, hi, sU);
                        wgrad16<64, 32>(accW1, Xs, Ys, i, hi);
                    } else {
                        stage_rows<64>(Xs, a1, i, hi);
                        stage_rows<32>(Ys, u, i, hi);
                        wgrad<64, 32>(accW1, Xs, Ys, i, hi);
                    }
                }
                TT_PHASE(7);
                // a1bar = W1 qbar ; b1bar = m1 . a1bar ; v = sbar h1 + b1bar
                float t1[32];
                mvx<PREC, 64, 32>(L + OFF_W1, L + GLO_W1, qb, t1, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) t1[r] = h1[r] > 0.f ? t1[r] : 0.f;  // b1bar
                float v[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) v[r] = fmaf(sbar * u1, h1[r], t1[r]);
                TT_PHASE(4);
                // dW2 += a2 v^T
                if (do_wgrad) {
                    if (WG16) {
                        stage_rows16_pre<64>(Xs, a2s, i, hi);
                        stage_rows16<64>(Ys, v, i, hi, sV);
                        wgrad16<64, 64>(accW2, Xs, Ys, i, hi);
                    } else {
                        stage_rows<64>(Xs, a2, i, hi);
                        stage_rows<64>(Ys, v, i, hi);
                        wgrad<64, 64>(accW2, Xs, Ys, i, hi);
                    }
                }
                TT_PHASE(8);
                // a2bar = W2 b1bar ; dw3 += sbar h2 + m2 . a2bar
                float t2[32];
                mvx<PREC, 64, 64>(L + OFF_W2, L + GLO_W2, t1, t2, i, hi);
#pragma unroll
                for (int r = 0; r < 32; ++r) t2[r] = fmaf(sbar * u2, h2[r], h2[r] > 0.f ? t2[r] : 0.f);
                stage_rows<64>(Xs, t2, i, hi);
                accw3 += rowsum32(Xs, lane);
                TT_PHASE(5);
                // ---- scatter d/d geometry planes: texel(p,c)[ch] += q[ch] * coef(p,c) ----
                if (region && !TT_DBG(cfg.flags, TT_DBG_NO_SCATTER)) {
                    scatter_clear<EXACT>(Xs, lane);  // M = 0 (Xs held wgrad staging; SCATTER_M_FLOATS reach into Ys)
                    float* Qs = Xs + SCATTER_M_FLOATS;  // the sample's row of Q: q scaled per plane, stride 33
                    TT_PHASE(9);
                    const int tex0 = (int)(pofs / TT_C);
                    scatter_planes<EXACT>(grad_out, grad_bytes, Qs, Xs, tags, Qs + 32 * 33, i, hi,
                                          [&](int pl, PlaneRefs& refs) {
                        Corners c;
                        float coef[4];  // per corner: w sbar + dw/dx . gbar -- gather AND scatter coefficient
                        geo_corner_coefs(pl, H, W, X, Y, Z, rvalid, sbar, gbx, gby, gbz, ju, jv, c, coef);
                        int aoff[4];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) aoff[q4] = tex0 + (int)(pl * HW) + c.off[q4];
                        refs = plane_refs<!EXACT>(coef, aoff, c.hs, hi);
#pragma unroll
                        for (int r = 0; r < 16; ++r) Qs[i * 33 + LIDX(r, hi)] = q[r] * refs.qs;
                    }
#ifdef TT_TUNING
                    , ph_acc + 14
#endif
                    );
                    TT_PHASE(10);
                }
            }
        }
    }
#ifdef TT_TUNING
    TT_PHASE(11);
    if (p.phase_cycles && lane == 0)
        for (int k = 0; k < 20; ++k) atomicAdd(p.phase_cycles + 20 + k, ph_acc[k]);
#endif
    // ---- flush the persistent weight-gradient accumulators ----
    // (summed over the workgroup's four waves first: the weight images in LDS are dead by now)
    __syncthreads();
    int parity = 0;
    flush_wgrad_reduced<64, 32>(L, parity, accW1, p.grads.w1, wave_in_blk, lane, 1.f / sA1, 1.f / sU);
    flush_wgrad_reduced<64, 64>(L, parity, accW2, p.grads.w2, wave_in_blk, lane, 1.f / sA2, 1.f / sV);
    atomicAdd(p.grads.w3 + lane, accw3);
    tile_stats_flush(st);
}

#ifdef TT_TUNING
unsigned long long* g_phase_cycles = nullptr;
// tuning build only: cycles per phase of k_decode_bwd_tex summed over waves since the last call (host copy), then reset
extern "C" int tt_tuning_phase_cycles(unsigned long long* out40) {
    if (!g_phase_cycles) {
        if (hipMalloc((void**)&g_phase_cycles, 40 * sizeof(unsigned long long)) != hipSuccess) return -4;
        if (hipMemset(g_phase_cycles, 0, 40 * sizeof(unsigned long long)) != hipSuccess) return -4;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (out40 && hipMemcpy(out40, g_phase_cycles, 40 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return -4;
    return hipMemset(g_phase_cycles, 0, 40 * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
#endif
static void launch_bwd_geo(const BwdGeoParams& p0, long long blocks, hipStream_t s) {
    BwdGeoParams p = p0;
#ifdef TT_TUNING
    p.phase_cycles = g_phase_cycles;
#else
    p.phase_cycles = nullptr;
#endif
    const int prec = tt_prec_of_r(p.cfg.flags);
#define LAUNCH_GEO(PREC_, WG_)                                                                                        \
    do {                                                                                                              \
        if (p.cfg.stats)                                                                                              \
            hipLaunchKernelGGL((k_decode_bwd_geo<PREC_, WG_, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);     \
        else                                                                                                          \
            hipLaunchKernelGGL((k_decode_bwd_geo<PREC_, WG_>), dim3((unsigned)blocks), dim3(256), 0, s, p);           \
    } while (0)
    if (use_wg16(p.cfg)) {
        unsigned* bnd = reinterpret_cast<unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        launch_planes_bound(p.packed, p.cfg, 0, bnd + TT_BOUND_PLANES, s);
        const long long n = p.cfg.n_rays * p.cfg.n_samples;  // upstream float4 (d sdf, d sdf_grad) per sample
        hipLaunchKernelGGL(k_absmax4, dim3(absmax_blocks(n), 1), dim3(256), 0, s, reinterpret_cast<const f32x4*>(p.ws), n, n,
                           bnd + TT_BOUND_UP0, bnd + TT_BOUND_UP1);
        if (prec == PREC_S3)
            LAUNCH_GEO(PREC_S3, true);
        else
            LAUNCH_GEO(PREC_S2, true);
    } else if (prec == PREC_F32) {
        LAUNCH_GEO(PREC_F32, false);
    }
#ifdef TT_TUNING
    else {  // TT_R_WGRAD_F32: the round-2 A/B kernel (two-piece products, fp32 outer products)
        LAUNCH_GEO(PREC_S2, false);
    }
#endif
#undef LAUNCH_GEO
}
int tt_launch_march_bwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                        const float* sdf, const float* sdf_grad, const float* features, const float* trans,
                        const float* opacity, const float* depth, const float* g_opacity, const float* g_depth,
                        const float* g_rgb_fg, const float* g_z_variance, const float* g_normal_acc,
                        const float* g_weights, const float* g_sdf, const float* g_sdf_grad, float* g_inv_std_rays,
                        float* ws, hipStream_t stream);

extern "C" int tt_render_bwd_geo(const float* packed, const tt_mlp_weights* w, const float* rays_o,
                                 const float* rays_d, const float* t_starts, const float* t_ends,
                                 const tt_render_cfg* cfg, const float* opacity, const float* depth,
                                 const float* trans, const float* sdf, const float* sdf_grad, const float* features,
                                 const float* g_opacity, const float* g_depth, const float* g_rgb_fg,
                                 const float* g_z_variance, const float* g_normal_acc, const float* g_weights,
                                 const float* g_sdf, const float* g_sdf_grad, float* g_inv_std_rays, float* workspace,
                                 float* grad_packed, const tt_mlp_grads* grads, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !opacity || !depth || !trans || !sdf ||
        !sdf_grad || !features || !workspace || !grad_packed || !grads)
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !grads->w1 || !grads->w2 || !grads->w3) return TT_ERR_BAD_ARG;
    if (grad_buffer_too_large(cfg)) return TT_ERR_UNSUPPORTED;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    st = tt_launch_march_bwd(rays_d, t_starts, t_ends, cfg, sdf, sdf_grad, features, trans, opacity, depth, g_opacity,
                             g_depth, g_rgb_fg, g_z_variance, g_normal_acc, g_weights, g_sdf, g_sdf_grad, g_inv_std_rays,
                             workspace, s);
    if (st != TT_OK) return st;
    BwdGeoParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.cfg.flags |= debug_flags();
    p.ws = workspace;
    p.grad_packed = grad_packed;
    p.n_copies = cfg->grad_copies > 0 ? cfg->grad_copies : 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(cfg, 4LL * cus, &p.geom, 1);
    long long blocks = persistent_blocks(p.n_items, cus);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    p.queue = tt_queue_counters(s);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_geo(p, blocks, s);
    return tt_check_launch();
}

// ---- backward of the per-point queries (tt_query_points / tt_query_field): the same decode-backward kernels, with
// "rays" of one sample whose origin is the point (direction null => x = o exactly) ----
__global__ void k_interleave_ws(const float* __restrict__ g_sdf, const float* __restrict__ g_sdf_grad,
                                float* __restrict__ ws, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f32x4 o = {g_sdf ? g_sdf[i] : 0.f, g_sdf_grad ? g_sdf_grad[i * 3 + 0] : 0.f,
               g_sdf_grad ? g_sdf_grad[i * 3 + 1] : 0.f, g_sdf_grad ? g_sdf_grad[i * 3 + 2] : 0.f};
    *reinterpret_cast<f32x4*>(ws + i * 4) = o;
}

extern "C" int tt_points_bwd_geo(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                                 int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                                 int32_t plane_w, float radius, float sdf_bias_radius, int32_t flags,
                                 const float* g_sdf, const float* g_sdf_grad, float* workspace, float* grad_packed,
                                 const tt_mlp_grads* grads, void* stream) {
    tt_render_cfg cfg;
    int st = points_cfg(&cfg, n_batch, n_points, n_prompts, views_per_prompt, plane_h, plane_w, radius,
                        sdf_bias_radius, 1, flags);
    if (st != TT_OK) return st;
    if (grad_buffer_too_large(&cfg)) return TT_ERR_UNSUPPORTED;
    if (!packed || !w || !points || !workspace || !grad_packed || !grads || (!g_sdf && !g_sdf_grad))
        return TT_ERR_BAD_ARG;
    if (!w->w1 || !w->w2 || !w->w3 || !grads->w1 || !grads->w2 || !grads->w3) return TT_ERR_BAD_ARG;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    const long long n = cfg.n_rays;
    hipLaunchKernelGGL(k_interleave_ws, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g_sdf, g_sdf_grad,
                       workspace, n);
    st = tt_check_launch();
    if (st != TT_OK) return st;
    BwdGeoParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = points;
    p.rays_d = nullptr;
    p.t_starts = nullptr;
    p.t_ends = nullptr;
    p.cfg = cfg;
    p.ws = workspace;
    p.grad_packed = grad_packed;
    p.n_copies = 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(&cfg, 4LL * cus, &p.geom, 1);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    long long blocks = persistent_blocks(p.n_items, cus);
    p.queue = tt_queue_counters(s);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_geo(p, blocks, s);
    return tt_check_launch();
}

