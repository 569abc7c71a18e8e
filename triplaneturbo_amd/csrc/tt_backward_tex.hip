// tt_backward_tex.hip -- texture half of the fused render backward: feature net backward, dV1 / dV2 / dV3, scatter of
// d/d planes 3..5 (k_decode_bwd_tex), and the per-point variant tt_points_bwd_tex.  See tt_backward.hip for the overview.
#include "tt_backward_common.h"
#ifndef TT_TEX_REREAD_RAY
#define TT_TEX_REREAD_RAY 1
#endif

// =====================================================================================================
// texture half
// =====================================================================================================
#define TEX_SCRATCH_FLOATS (128 * XS)

// STATS: work accounting compiled in (see k_decode_bwd_geo): production launches run the instantiation without it
template <int PREC, bool WG16, bool STATS = false>
__global__ __launch_bounds__(256, 1) void k_decode_bwd_tex(BwdTexParams p) {
    constexpr bool EXACT = PREC == PREC_F32;
    constexpr bool COPIES = TT_BWD_WT_COPIES && PREC != PREC_S3;  // transposed weight copies (tt_backward_common.h)
    constexpr int NT = PrecNT<PREC>::value, WF = TexWFloats<PREC>::value;
    __shared__ __attribute__((aligned(16))) float Lt[WF + 4 * (TEX_SCRATCH_FLOATS + SCATTER_TAG_INTS)];
    {
        MlpPtrs w = p.w;
        stage_weights<PREC, 64, 96>(Lt + TV1, Lt + TLO_V1, w.v1);
        stage_weights<PREC, 64, 64>(Lt + TV2, Lt + TLO_V2, w.v2);
        lds_load_matrix(Lt + TV3, w.v3, 3, 64, 64);
        if constexpr (COPIES) {
            stage_weights_t<PREC, 64, 96>(Lt + TV1T, nullptr, w.v1);
            stage_weights_t<PREC, 64, 64>(Lt + TV2T, nullptr, w.v2);
        }
    }
    const tt_render_cfg& cfg = p.cfg;
    // ---- per-launch operand scales of the fp16 outer products dV1 += k1bar e^T, dV2 += k2bar k1^T (wgrad16) ----
    // bounds:  |e| <= P (bilinear weights are a convex combination),  |k1_i| <= ||V1_i||_1 P,
    //   |cbar| <= |shrink| 1.002 / 4 Gr + Gf  (weights <= 1, sigmoid' <= 1/4; Gr / Gf = max |g_rgb| / |g_features|),
    //   |k2bar_i| <= sum_o |V3[o][i]| |cbar|,   |k1bar_j| <= sum_i |V2[i][j]| (bound of k2bar_i).
    float sKB1 = 1.f, sE = 1.f, sK2B = 1.f, sK1 = 1.f;
    if (WG16) {
        const unsigned* bnd = reinterpret_cast<const unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        const float Pm = __builtin_bit_cast(float, bnd[TT_BOUND_PLANES]), Gr = __builtin_bit_cast(float, bnd[TT_BOUND_UP0]),
                    Gf = __builtin_bit_cast(float, bnd[TT_BOUND_UP1]);
        unsigned* word = reinterpret_cast<unsigned*>(Lt + WF);  // scratch is free until the main loop
        const int t = threadIdx.x;
        const float CBmax = __builtin_fabsf(cfg.rgb_grad_shrink) * (1.002f * 0.25f) * Gr + Gf;
        float v1row = 0.f, k2b = 0.f, kb1 = 0.f;
        if (t < 64) {
            for (int c = 0; c < 96; ++c) v1row += __builtin_fabsf(p.w.v1[t * 96 + c]);
            for (int o = 0; o < 3; ++o) k2b += __builtin_fabsf(p.w.v3[o * 64 + t]);
            for (int r = 0; r < 64; ++r) {
                float c3 = 0.f;
                for (int o = 0; o < 3; ++o) c3 += __builtin_fabsf(p.w.v3[o * 64 + r]);
                kb1 += __builtin_fabsf(p.w.v2[r * 64 + t]) * c3;
            }
        }
        const float V1max = block_max(v1row, word), K2Bw = block_max(k2b, word), KB1w = block_max(kb1, word);
        sE = wg16_scale(Pm);
        sK1 = wg16_scale(V1max * Pm);
        sK2B = wg16_scale(K2Bw * CBmax);
        sKB1 = wg16_scale(KB1w * CBmax);
    }
    const TileGeom& tg = p.geom;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5, wave_in_blk = threadIdx.x >> 6;
    // per-wave scratch: rows 0..31 = Xs (transposition window / first half of bigger operands), rows 32..127 = Ys
    float* Xs = Lt + WF + wave_in_blk * (TEX_SCRATCH_FLOATS + SCATTER_TAG_INTS);
    float* Ys = Xs + 32 * XS;  // 96 rows: the parked e
    int* tags = reinterpret_cast<int*>(Xs + 128 * XS);
    // cbar of the tile, [3][32]: in the 4 pad columns of Xs rows 0..23 (row r holds floats 4r..4r+3 of the 96) --
    // stage_rows / the scatter matrix only touch columns 0..31 of a row
    float* Cb = Xs + 32;
#define CB_AT(idx) Cb[((idx) >> 2) * XS + ((idx)&3)]
    scatter_init_tags(tags, lane);
    __syncthreads();
    const int S = cfg.n_samples;
    const int H = cfg.plane_h, W = cfg.plane_w;
    const size_t HW = (size_t)H * W;
    const size_t plane_stride = 6 * HW * TT_C;
    ItemQueue iq = item_queue(p.queue, tg.n_blocks, tg.n_chunks, tg.unit);
    const float shrink = cfg.rgb_grad_shrink;
    float* const grad_out =  // private copy of the gradient planes of this workgroup (see k_decode_bwd_geo)
        p.grad_packed + (size_t)(blockIdx.x % (unsigned)p.n_copies) * cfg.n_prompts * plane_stride;
    // (tuning build, TT_DBG_NO_ATOMICS: num_records = 0 -- every flush atomic is still issued and then dropped by the range check)
    const unsigned grad_bytes =
        TT_DBG(cfg.flags, TT_DBG_NO_ATOMICS) ? 0u : (unsigned)(cfg.n_prompts * plane_stride * sizeof(float));  // one copy, < 4 GB - 256

    const TileStats st = STATS ? tile_stats(cfg.stats) : TileStats{nullptr};
    f32x16 accV1a[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};  // dV1[:, 0:64]
    f32x16 accV1b[2][1] = {{ZERO16}, {ZERO16}};                  // dV1[:, 64:96]
    f32x16 accV2[2][2] = {{ZERO16, ZERO16}, {ZERO16, ZERO16}};
    float accV3[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // [half of the 64 indices][output]; this lane: 16 samples
#ifdef TT_TUNING
    unsigned long long ph_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long scat_st[3] = {0, 0, 0};
    unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif

#pragma nounroll
    for (;;) {
      long long b;
      int ck;
      TT_PHASE(17);
      if (!item_pop(iq, tg.order, tg.n_chunks, b, ck)) break;
      TT_PHASE(18);
        if (b >= tg.n_blocks) continue;  // padding of the ragged last deal round
      bool ray_ok;
      const long long ray = tile_ray(tg, b, i, ray_ok);
      const float ray_okf = tt_opaque(ray_ok ? 1.f : 0.f);  // 0/1 factor the compiler cannot fold back into a mask
      const int ks = i % tg.sb;  // this lane's sample offset inside a tile step
      const int view = (int)(ray / cfg.rays_per_view);
      const size_t pofs = (size_t)(view / cfg.views_per_prompt) * plane_stride;
#if TT_TEX_REREAD_RAY
      // Per-ray constants (origin, direction, d loss / d rgb) are RE-READ at the top of every tile step instead of living in
      // nine registers across the item: the same addresses for every step of an item (L1 hits, ~0.5 % of a step), and the
      // registers the allocator would otherwise spill to scratch memory (4 in the default mode, 68 in the fp32-MFMA mode).
      // The ray index is laundered through an empty asm so that hipcc cannot hoist the loads back out of the loop.
#else
      const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
      const float dx = p.rays_d ? p.rays_d[ray * 3 + 0] : 0.f, dy = p.rays_d ? p.rays_d[ray * 3 + 1] : 0.f,
                  dz = p.rays_d ? p.rays_d[ray * 3 + 2] : 0.f;
      float grgb[3];
#pragma unroll
      for (int o = 0; o < 3; ++o) grgb[o] = p.g_rgb ? p.g_rgb[ray * 3 + o] : 0.f;
#endif
      const int s_end = (ck + 1) * tg.chunk < S ? (ck + 1) * tg.chunk : S;
      TT_PHASE(19);
      // Per-step inputs (weight, features, interval, upstream) are PREFETCHED one tile step ahead: their loads are
      // issued at the top of the previous step and have landed long before they are needed (they used to cost two
      // exposed memory round trips per step, 8 % of the kernel; the old "weights first" early-out bought nothing on a
      // scene where 95 % of the tile steps are live).  A step past the chunk reads a clamped, valid address.
      struct StepIn {
          float wgt, f[3], gf[3], ts, te;
      };
      auto load_step = [&](int sb0) {
          StepIn r;
          const int si = sb0 + ks;
          const long long sidx = ray * S + (si < S ? si : S - 1);
          r.wgt = p.weights ? p.weights[sidx] : 0.f;  // null: no march above (points)
#pragma unroll
          for (int o = 0; o < 3; ++o) {
              r.f[o] = p.weights ? p.features[sidx * 3 + o] : 0.f;  // (features only enter through the weights)
              r.gf[o] = p.g_features ? p.g_features[sidx * 3 + o] : 0.f;
          }
          r.ts = p.rays_d ? p.t_starts[sidx] : 0.f;
          r.te = p.rays_d ? p.t_ends[sidx] : 0.f;
          return r;
      };
      StepIn in = load_step(ck * tg.chunk), in_next;
#pragma nounroll
      for (int sb0 = ck * tg.chunk; sb0 < s_end; sb0 += tg.sb, in = in_next) {
        in_next = load_step(sb0 + tg.sb);
        const int si = sb0 + ks;
        const bool valid = ray_ok && si < s_end;
        const float vf = ray_okf * (si < s_end ? 1.f : 0.f);
#if TT_TEX_REREAD_RAY
        long long rr = ray;
        asm volatile("" : "+v"(rr));
        const float ox = p.rays_o[rr * 3 + 0], oy = p.rays_o[rr * 3 + 1], oz = p.rays_o[rr * 3 + 2];
        const float dx = p.rays_d ? p.rays_d[rr * 3 + 0] : 0.f, dy = p.rays_d ? p.rays_d[rr * 3 + 1] : 0.f,
                    dz = p.rays_d ? p.rays_d[rr * 3 + 2] : 0.f;
        float grgb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) grgb[o] = p.g_rgb ? p.g_rgb[rr * 3 + o] : 0.f;
#endif
        // ---- upstream: cbar_o = shrink * w_i * g_rgb[ray,o] * 1.002 * s(1-s) + g_features ----
        float cb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float s = sigmoid_(in.f[o]);
            const float c = shrink * in.wgt * grgb[o] * 1.002f * s * (1.f - s) + in.gf[o];
            cb[o] = c * vf;  // 0/1 factor, not a select on a freshly combined lane mask (see k_decode_bwd_geo)
        }
        tile_stat(st, TT_STAT_VISITED);
        TT_PHASE(0);
        // exact with skip_eps_tex = 0 (the default: nothing flows back); > 0: the opt-in approximation of tt_abi.h
        if (!__any(!((__builtin_fabsf(cb[0]) + __builtin_fabsf(cb[1])) + __builtin_fabsf(cb[2]) <= cfg.skip_eps_tex)))
            continue;
#ifdef TT_TUNING
        {  // live-lane statistics (tools/phase_cycles.py): slots 12 / 13 are unused by the timers
            const unsigned long long live = __ballot(cb[0] != 0.f || cb[1] != 0.f || cb[2] != 0.f) & 0xffffffffull;
            const unsigned long long big = __ballot(fabsf(cb[0]) + fabsf(cb[1]) + fabsf(cb[2]) > 1e-12f) & 0xffffffffull;
            ph_acc[12] += 1;
            ph_acc[13] += __popcll(live);
            ph_acc[14] += __popcll(big);
        }
#endif
        float tm, px, py, pz;
        sample_position(ox, oy, oz, dx, dy, dz, in.ts, in.te, tm, px, py, pz);
        const float X = scale_coord(px, cfg.radius), Y = scale_coord(py, cfg.radius), Z = scale_coord(pz, cfg.radius);
        float e[48];
        const bool any =
            __any(gather_tex_c(p.packed, (unsigned)(pofs / TT_C), H, W, X, Y, Z, valid, lane, Xs, e,
                               tile_stat_ptr(st, TT_STAT_INBOUNDS)));
        TT_PHASE(1);
        if (!any) continue;  // exact: e == 0 => k1 = k2 = 0 and every mask is false
        tile_stat(st, TT_STAT_EXECUTED);
        // e is needed again only as the Y operand of the dV1 outer product: park it in LDS now ([idx][sample]
        // layout, 96 rows) so its 48 registers are free during the MLP chain.
        // (`region`: always true -- tt_validate_cfg rejects negative flags -- but opaque to the compiler.  The two
        // conditional regions below split this ~9000-instruction loop body into separate scheduling / allocation
        // regions: 97 -> 45 spilled registers, 5.46 -> 4.84 ms.  __builtin_amdgcn_sched_barrier does not have that
        // effect; found by noticing that the -DTT_TUNING build, whose ablation branches are live, was FASTER.)
        // With the outer products on the fp16 pipe (WG16) only the scatter keeps its own region: 27 -> 0 spilled
        // registers, 3.37 -> 3.18 ms (no region at all: 67 spills, 3.99 ms; region around the outer products only: 3.57).
        const bool region = cfg.flags >= 0;
        const bool region_w = WG16 ? true : region;
        const bool do_wgrad = region_w && !TT_DBG(cfg.flags, TT_DBG_NO_WGRAD);
        float k1[32], k2[32];
        if (WG16) {  // e is split once, under its per-launch scale, for the outer product and for V1 e
            Split16<96, PAIR_SEQ, NT> es;
            split16_vec<96, PAIR_SEQ, NT>(e, sE, es);
            if (do_wgrad) stage_rows16_pre<96>(Ys, es, i, hi);
            TT_PHASE(2);
            mv16_pre<64, 96, false, NT>(Lt + TV1, es, 1.f / sE, k1, i, hi, nullptr, Lt + TLO_V1);
        } else {
            if (do_wgrad) stage_rows<96>(Ys, e, i, hi);
            TT_PHASE(2);
            mvx<PREC, 64, 96>(Lt + TV1, Lt + TLO_V1, e, k1, i, hi);
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) k1[r] = fmaxf(k1[r], 0.f);
        TT_PHASE(3);
        Split16<64, PAIR_SEQ, NT> k1s;  // k1 likewise: V2 k1 now, the dV2 outer product later
        if (WG16) {
            split16_vec<64, PAIR_SEQ, NT>(k1, sK1, k1s);
            mv16_pre<64, 64, false, NT>(Lt + TV2, k1s, 1.f / sK1, k2, i, hi, nullptr, Lt + TLO_V2);
        } else {
            mvx<PREC, 64, 64>(Lt + TV2, Lt + TLO_V2, k1, k2, i, hi);
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) k2[r] = fmaxf(k2[r], 0.f);
        TT_PHASE(4);
        // ---- dV3[o][idx] += sum_s cbar_o[s] k2[idx][s]: k2 goes through the 32-row window in two halves (registers
        // 0..15 hold indices 0..31, registers 16..31 indices 32..63); lane (i, hi) sums samples 16 hi .. 16 hi + 15 of
        // row i against cbar ----
        if (hi == 0) {
            CB_AT(0 * 32 + i) = cb[0];
            CB_AT(1 * 32 + i) = cb[1];
            CB_AT(2 * 32 + i) = cb[2];
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            if (h2 == 0)
                stage_rows_sub<32, 0, 32>(Xs, k2, i, hi);
            else
                stage_rows_sub<32, 16, 32>(Xs, k2, i, hi);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kk = *reinterpret_cast<const f32x4*>(Xs + i * XS + 16 * hi + 4 * g);
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const f32x4 cc = *reinterpret_cast<const f32x4*>(&CB_AT(o * 32 + 16 * hi + 4 * g));
                    accV3[h2][o] += (kk[0] * cc[0] + kk[1] * cc[1]) + (kk[2] * cc[2] + kk[3] * cc[3]);
                }
            }
        }
        TT_PHASE(5);
        // ---- k2bar = n2 . (V3^T cbar) ----
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 0 * 64 + 8 * g + 4 * hi);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 1 * 64 + 8 * g + 4 * hi);
            f32x4 v2 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 2 * 64 + 8 * g + 4 * hi);
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const float t = fmaf(v0[e2], cb[0], fmaf(v1[e2], cb[1], v2[e2] * cb[2]));
                k2[4 * g + e2] = k2[4 * g + e2] > 0.f ? t : 0.f;
            }
        }
        // ---- k1bar = n1 . (V2^T k2bar) ----
        float kb1[32];
        if constexpr (COPIES)
            mvtx_copy<PREC, 64, 64, 64>(Lt + TV2T, nullptr, Lt + TV2, k2, kb1, i, hi);
        else
            mvtx<PREC, 64, 64, 64>(Lt + TV2, Lt + TLO_V2, 0, k2, kb1, i, hi);
#pragma unroll
        for (int r = 0; r < 32; ++r) kb1[r] = k1[r] > 0.f ? kb1[r] : 0.f;
        TT_PHASE(6);
        if (do_wgrad) {
            // (compiler scheduling fence in front of the outer products: 2.88 -> 2.85 ms; the same fence in front of the
            // dV2 products, or both, gains nothing -- profiles/experiments/README.md)
            __builtin_amdgcn_sched_barrier(0);
            // ---- dV1 += k1bar e^T  (e parked in Ys rows 0..95; k1bar through the 32-row window, half by half) ----
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                if (WG16) {
                    if (h2 == 0)
                        stage_rows16_sub<32, 0, 32>(Xs, kb1, i, hi, sKB1);
                    else
                        stage_rows16_sub<32, 16, 32>(Xs, kb1, i, hi, sKB1);
                    wgrad16_row<64>(accV1a[h2], Xs, Ys, i, hi);
                    wgrad16_row<32>(accV1b[h2], Xs, Ys + 64 * XS, i, hi);
                } else {
                    if (h2 == 0)
                        stage_rows_sub<32, 0, 32>(Xs, kb1, i, hi);
                    else
                        stage_rows_sub<32, 16, 32>(Xs, kb1, i, hi);
                    wgrad_row<64>(accV1a[h2], Xs, Ys, i, hi);
                    wgrad_row<32>(accV1b[h2], Xs, Ys + 64 * XS, i, hi);
                }
            }
            TT_PHASE(7);
            // ---- dV2 += k2bar k1^T  (e is dead: k2bar in rows 0..63, k1 in rows 64..127) ----
            if (WG16) {
                stage_rows16<64>(Xs, k2, i, hi, sK2B);
                stage_rows16_pre<64>(Xs + 64 * XS, k1s, i, hi);
                wgrad16<64, 64>(accV2, Xs, Xs + 64 * XS, i, hi);
            } else {
                stage_rows<64>(Xs, k2, i, hi);
                stage_rows<64>(Xs + 64 * XS, k1, i, hi);
                wgrad<64, 64>(accV2, Xs, Xs + 64 * XS, i, hi);
            }
            TT_PHASE(8);
        }
        // ---- ebar = V1^T k1bar (one plane at a time) ; scatter texel(3+p, c)[ch] += w_c * ebar[32p + ch] ----
        if (region && !TT_DBG(cfg.flags, TT_DBG_NO_SCATTER)) {
            // the combine GEMM on the fp16 pipe as in the geometry kernel (round 2 measured +26 spilled registers and
            // 3.88 -> 4.16 ms for this; with the outer products on the fp16 pipe it fits: 3.65 -> 3.37 ms)
            constexpr bool SC_EXACT = EXACT || !WG16;  // (the TT_R_WGRAD_F32 A/B variant = the round-2 kernel)
            float* M = Xs;              // rows 0..63: the slot x sample coefficient matrix (fp32), row 64: dump row
            float* Es = Xs + (SC_EXACT ? 65 * XS : SCATTER_M_FLOATS);  // ebar rows [sample][32], stride 33; fallback lists
            scatter_clear<SC_EXACT>(M, lane);
            // ebar = V1^T k1bar for the three planes in ONE product (96 rows: k1bar is split into fp16 terms once)
            float eb[48];
            if constexpr (COPIES)
                mvtx_copy<PREC, 96, 64, 96>(Lt + TV1T, nullptr, Lt + TV1, kb1, eb, i, hi);
            else
                mvtx<PREC, 96, 64, 96>(Lt + TV1, Lt + TLO_V1, 0, kb1, eb, i, hi);
            TT_PHASE(9);
            const int tex0 = (int)(pofs / TT_C);
            scatter_planes<SC_EXACT>(grad_out, grad_bytes, Es, M, tags, Es + 32 * 33, i, hi, [&](int pl, PlaneRefs& refs) {
                Corners c;
                corners_setup(PLANE_U(pl, X, Y, Z), PLANE_V(pl, X, Y, Z), H, W, valid, c);
                int aoff[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)  // absolute texel index, prompt included
                    aoff[q4] = tex0 + (int)((3 + pl) * HW) + c.off[q4];
                refs = plane_refs<!SC_EXACT>(c.w, aoff, c.hs, hi);
#pragma unroll
                for (int r = 0; r < 16; ++r) Es[i * 33 + LIDX(r, hi)] = eb[16 * pl + r] * refs.qs;
            }
#ifdef TT_TUNING
            , scat_st
#endif
            );
            TT_PHASE(10);
        }
      }
    }
#ifdef TT_TUNING
    TT_PHASE(11);
    ph_acc[15] = scat_st[0];  // active references
    ph_acc[16] = scat_st[1];  // lost references (plane-tiles = 3 per live tile step, slot 12)
    if (p.phase_cycles && lane == 0)
        for (int k = 0; k < 20; ++k) atomicAdd(p.phase_cycles + k, ph_acc[k]);
#endif
    // dV1 is (64, 96) row-major: columns 0..63 from accV1a, 64..95 from accV1b.  Every matrix is summed over the
    // workgroup's four waves first (the weight images in LDS are dead by now)
    __syncthreads();
    int parity = 0;
    {
        const float u1 = 1.f / sKB1, u2 = 1.f / sE;  // inverse operand scales (1 for the fp32 outer products)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float* const base = p.grads.v1 + (32 * m) * 96 + i;
            flush_tile_reduced(Lt, parity, accV1a[m][0], wave_in_blk, lane, u1, u2,
                               [&](int r) { return base + LIDX(r, hi) * 96; });
            parity ^= 1;
            flush_tile_reduced(Lt, parity, accV1a[m][1], wave_in_blk, lane, u1, u2,
                               [&](int r) { return base + LIDX(r, hi) * 96 + 32; });
            parity ^= 1;
            flush_tile_reduced(Lt, parity, accV1b[m][0], wave_in_blk, lane, u1, u2,
                               [&](int r) { return base + LIDX(r, hi) * 96 + 64; });
            parity ^= 1;
        }
    }
    flush_wgrad_reduced<64, 64>(Lt, parity, accV2, p.grads.v2, wave_in_blk, lane, 1.f / sK2B, 1.f / sK1);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int o = 0; o < 3; ++o) atomicAdd(p.grads.v3 + o * 64 + 32 * h2 + i, accV3[h2][o]);
    tile_stats_flush(st);
}

static void launch_bwd_tex(const BwdTexParams& p0, long long blocks, hipStream_t s) {
    BwdTexParams p = p0;
#ifdef TT_TUNING
    p.phase_cycles = g_phase_cycles;
#else
    p.phase_cycles = nullptr;
#endif
    const int prec = tt_prec_of_r(p.cfg.flags);
#define LAUNCH_TEX(PREC_, WG_)                                                                                        \
    do {                                                                                                              \
        if (p.cfg.stats)                                                                                              \
            hipLaunchKernelGGL((k_decode_bwd_tex<PREC_, WG_, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);     \
        else                                                                                                          \
            hipLaunchKernelGGL((k_decode_bwd_tex<PREC_, WG_>), dim3((unsigned)blocks), dim3(256), 0, s, p);           \
    } while (0)
    if (use_wg16(p.cfg)) {
        unsigned* bnd = reinterpret_cast<unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        launch_planes_bound(p.packed, p.cfg, 3, bnd + TT_BOUND_PLANES, s);  // (p.packed may be slid by 3 planes: points)
        if (p.g_rgb)
            hipLaunchKernelGGL(k_absmax1, dim3(absmax_blocks(p.cfg.n_rays * 3)), dim3(256), 0, s, p.g_rgb,
                               (long long)p.cfg.n_rays * 3, bnd + TT_BOUND_UP0);
        if (p.g_features) {
            const long long n = p.cfg.n_rays * p.cfg.n_samples * 3;
            hipLaunchKernelGGL(k_absmax1, dim3(absmax_blocks(n)), dim3(256), 0, s, p.g_features, n, bnd + TT_BOUND_UP1);
        }
        if (prec == PREC_S3)
            LAUNCH_TEX(PREC_S3, true);
        else
            LAUNCH_TEX(PREC_S2, true);
    } else if (prec == PREC_F32) {
        LAUNCH_TEX(PREC_F32, false);
    }
#ifdef TT_TUNING
    else {  // TT_R_WGRAD_F32: the round-2 A/B kernel
        LAUNCH_TEX(PREC_S2, false);
    }
#endif
#undef LAUNCH_TEX
}

extern "C" int tt_render_bwd_tex(const float* packed, const tt_mlp_weights* w, const float* rays_o,
                                 const float* rays_d, const float* t_starts, const float* t_ends,
                                 const tt_render_cfg* cfg, const float* weights, const float* features,
                                 const float* g_rgb_fg, const float* g_features, float* grad_packed,
                                 const tt_mlp_grads* grads, void* stream) {
    int st = tt_validate_cfg(cfg);
    if (st != TT_OK) return st;
    if (!packed || !w || !rays_o || !rays_d || !t_starts || !t_ends || !weights || !features || !grad_packed ||
        !grads)
        return TT_ERR_BAD_ARG;
    if (!w->v1 || !w->v2 || !w->v3 || !grads->v1 || !grads->v2 || !grads->v3) return TT_ERR_BAD_ARG;
    if (grad_buffer_too_large(cfg)) return TT_ERR_UNSUPPORTED;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    BwdTexParams p;
    p.packed = packed;
    p.w = to_ptrs(w);
    p.rays_o = rays_o;
    p.rays_d = rays_d;
    p.t_starts = t_starts;
    p.t_ends = t_ends;
    p.cfg = *cfg;
    p.cfg.flags |= debug_flags();
    p.weights = weights;
    p.features = features;
    p.g_rgb = g_rgb_fg;
    p.g_features = g_features;
    p.grad_packed = grad_packed;
    p.n_copies = cfg->grad_copies > 0 ? cfg->grad_copies : 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(cfg, 4LL * cus, &p.geom, 1, 12);
    long long blocks = persistent_blocks(p.n_items, cus);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    p.queue = tt_queue_counters((hipStream_t)stream);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_tex(p, blocks, (hipStream_t)stream);
    return tt_check_launch();
}

extern "C" int tt_points_bwd_tex(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                                 int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h,
                                 int32_t plane_w, float radius, int32_t plane_base, int32_t flags,
                                 const float* g_features, float* grad_packed, const tt_mlp_grads* grads,
                                 void* stream) {
    tt_render_cfg cfg;
    int st = points_cfg(&cfg, n_batch, n_points, n_prompts, views_per_prompt, plane_h, plane_w, radius, 0.5f, 1,
                        flags);
    if (st != TT_OK) return st;
    if (grad_buffer_too_large(&cfg)) return TT_ERR_UNSUPPORTED;
    if (!packed || !w || !points || !g_features || !grad_packed || !grads) return TT_ERR_BAD_ARG;
    if (!w->v1 || !w->v2 || !w->v3 || !grads->v1 || !grads->v2 || !grads->v3) return TT_ERR_BAD_ARG;
    if (plane_base != 0 && plane_base != 3) return TT_ERR_BAD_ARG;
    int cus = tt_num_cus();
    if (cus <= 0) return TT_ERR_DEVICE;
    // the kernel addresses planes 3..5 of each prompt; plane_base = 0 slides that window onto planes 0..2
    const ptrdiff_t shift = (ptrdiff_t)(plane_base - 3) * plane_h * plane_w * TT_C;
    BwdTexParams p;
    p.packed = packed + shift;
    p.w = to_ptrs(w);
    p.rays_o = points;
    p.rays_d = nullptr;
    p.t_starts = nullptr;
    p.t_ends = nullptr;
    p.cfg = cfg;
    p.weights = nullptr;
    p.features = nullptr;
    p.g_rgb = nullptr;
    p.g_features = g_features;
    p.grad_packed = grad_packed + shift;
    p.n_copies = 1;
    p.grads = to_gptrs(grads);
    p.n_items = tt_make_geom(&cfg, 4LL * cus, &p.geom, 1);
    if (p.n_items > (1LL << 30)) return TT_ERR_UNSUPPORTED;
    long long blocks = persistent_blocks(p.n_items, cus);
    p.queue = tt_queue_counters((hipStream_t)stream);
    if (!p.queue) return TT_ERR_DEVICE;
    launch_bwd_tex(p, blocks, (hipStream_t)stream);
    return tt_check_launch();
}

