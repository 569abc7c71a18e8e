// tt_backward_tex2.hip -- texture half of the fused render backward as WAVE PAIRS (round 4): two waves per SIMD.
//
// k_decode_bwd_tex (tt_backward_tex.hip, rounds 1-3) gives a whole 32-sample tile to ONE wave: 160 persistent
// weight-gradient accumulator registers + the activation vectors need the full 512-register budget, i.e. one wave per SIMD,
// and that lone wave serialises its ~2 400 VALU instructions, ~240 MFMAs and ~430 LDS instructions per tile step with
// nothing to hide latencies behind (VALU 39 %, MFMA 22 % of the cycles; profiles/r03_extra_counters.md).  Here a PAIR of
// waves (on two different SIMDs) shares the 32 samples of a tile:
//   * every hidden vector is split by ROWS: wave H (0 / 1) computes elements 32 H .. 32 H + 31 of k1, k2, k2bar, k1bar
//     for all 32 samples (a single 32 x 32 MFMA row tile per product, half the operand splits, half the post-processing),
//     writes its half as ready-made split-fp16 B FRAGMENTS into a pair-shared LDS buffer and reads the partner's half from
//     there -- 4 ds_write_b128 + 4 ds_read_b128 per wave and exchange;
//   * the accumulators are split too: wave H owns dV1[32 H .. , :] (48 registers), dV2[:, 32 H ..] (32) and dV3[:, 32 H ..];
//   * the outer products take their [index][sample] operands NOT through an LDS transposition scratch but from the same
//     fragment images, read TRANSPOSED by ds_read_b64_tr_b16 (tr_operand<> in tt_backward_common.h; the images are stored
//     in the PAIR_TR register pairing and their rows are un-permuted at the flush, tr_elem()): an image written once
//     serves the next product (B operand), the partner (exchange) and both outer products it takes part in;
//   * the gather is split by samples (16 each), the plane-gradient scatter by planes (plane 0 / 1, plane 2 alternating).
// => 80 accumulator registers + half-size vectors; three pairs per CU (384 threads, LDS-bound), 168 registers per wave;
// the pair synchronises six times per tile step through sequence flags in LDS (no s_barrier: the pairs of a workgroup
// run independently).
// All split operands use PER-LAUNCH scales (rigorous bounds, as the outer products always did): one split serves the
// product, the exchange and the outer product.  Default precision only; TT_R_EXACT_F32 / TT_R_WGRAD_F32 and the per-point
// variant stay on k_decode_bwd_tex.
//
// STATUS (round 4, profiles/experiments/README.md "Round 4"): CORRECT -- every gradient within 1e-6 of the one-wave kernel
// (tests/test_gpu_pair.py) -- but SLOWER: 4.07 ms against 2.87 ms on configs[1] (4.31 / 2.93 when first built), so it is opt-in (TT_R_BWD_PAIR) and the
// default stays k_decode_bwd_tex.  Measured causes: the code around the accumulators needs ~245 registers (gather
// coefficients, scatter lists, fragment addressing), so at the 168-register cap of 3 waves / SIMD ~70 values live in
// scratch and the accumulators are reloaded around the scatter; and a pair hand-off costs ~740 cycles of flag ping-pong
// (tools/pair_sync_probe.hip), six times per tile step.  What a faster version needs is listed in DESIGN.md ("wave
// pairs"): a <= 150-register non-accumulator body and fewer, coarser hand-offs.
// TUNING BUILD ONLY since round 5 (-DTT_TUNING, libtt_hip_tuning.so): the product library carries ONE texture-backward kernel.
#ifdef TT_TUNING
#include "tt_backward_common.h"

#ifndef P2_PAIRS
#define P2_PAIRS 3
#endif
#define P2_THREADS (128 * P2_PAIRS)

// ---- LDS map of a pair (floats) ----
// A fragment image = blocks [k-step][half-wave] of 32 samples x 8 halfs (the B fragment of a sample = one 16-byte row),
// each block padded by 64 bytes: the transposed reads of tr_operand then spread over all 64 banks (block stride = 16
// dwords mod 64, k-step stride = 32).  Every vector has a hi image and a lo image.
#define P2_BLK FR_BLK                     /* halfs per block (tt_backward_common.h) */
#define P2_EF_HALFS (6 * 2 * P2_BLK)      /* e: 6 k-steps */
#define P2_XF_HALFS (4 * 2 * P2_BLK)      /* a 64-vector: 4 k-steps */
#define P2_EF 0                           /* floats: hi image, lo image */
#define P2_XF0 (P2_EF + P2_EF_HALFS)
#define P2_XF1 (P2_XF0 + P2_XF_HALFS)
#define P2_FRAGS (P2_XF1 + P2_XF_HALFS)
#define P2_TAB P2_XF1                     /* gather tables (XF1 is free until the k2bar exchange) */
#define P2_SCAT_FLOATS 4096               /* per wave, over the (by then dead) fragment images: M, Es, lists, tags */
#define P2_REGION (P2_FRAGS > 2 * P2_SCAT_FLOATS ? P2_FRAGS : 2 * P2_SCAT_FLOATS)
#define P2_CTRL P2_REGION                 /* 16 ints: flags[2], any[2], item b (2), ck, ok; then cbar[2 waves][3][32] */
#define P2_ES2 (P2_CTRL + 16 + 2 * 96)    /* Q rows of plane 2 (the wave whose turn it is): [32 samples][33] */
#define P2_PAIR_FLOATS (P2_ES2 + 32 * 33 + 8)
static_assert(SCATTER_M_FLOATS + 32 * 33 + 2 * 32 * 4 + SCATTER_TAG_INTS <= P2_SCAT_FLOATS, "scatter region too small");
static_assert(3 * 32 * 8 <= P2_XF_HALFS, "gather tables must fit XF1");
static_assert(16 * XS <= 4 * P2_BLK / 2, "the dV3 window (16 rows) must fit my own blocks of XF1");

// (P2_DBG_NO_SCATTER / P2_DBG_NO_OUTER / P2_DBG_NO_ATOMICS: compile-time ablations behind the numbers of
// profiles/experiments/README.md, tools/build_variants.py; never defined in the product build)
// tuning build only: cycles per phase summed over waves into p.phase_cycles[0..19] (tools/phase_cycles_pair.py)
#ifdef TT_TUNING
#define P2_PHASE(k)                                                    \
    do {                                                               \
        __builtin_amdgcn_sched_barrier(0);                             \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
        ph_acc[k] += t_now - ph_t;                                     \
        ph_t = t_now;                                                  \
        __builtin_amdgcn_sched_barrier(0);                             \
    } while (0)
#else
#define P2_PHASE(k) \
    do {            \
    } while (0)
#endif

struct PairCtx {
    int* ctrl;  // flags[0], flags[1], any[0], any[1], b_lo, b_hi, ck, ok
    int H;
    int seq;
};
// Both waves of the pair: everything I wrote to LDS before is visible to the partner after, and the partner has finished
// every LDS access it issued before ITS call.  LDS only: the DS instructions of a wave are processed in order, so a flag
// written after the data is seen after the data, and reads issued after the flag has been seen come after it -- no
// s_waitcnt at all on the way (a workgroup-scope release FENCE would also wait for vmcnt(0), i.e. for every outstanding
// scatter atomic of the wave to reach the L2: measured, 65 % of the wave cycles parked).  The compiler is kept from
// moving LDS accesses across by signal fences.  Bounded spin (a protocol error must not hang the GPU).
__device__ __forceinline__ void pair_sync(PairCtx& pc, int lane) {
    __builtin_amdgcn_sched_barrier(0);
    ++pc.seq;
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (lane == 0) __hip_atomic_store(pc.ctrl + pc.H, pc.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(
               __hip_atomic_load(pc.ctrl + (pc.H ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < pc.seq) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) __builtin_trap();  // a lost partner is a bug: fault the launch, never continue on half-written fragments
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_sched_barrier(0);
}

#define Z16 f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}

// my 16 registers (one 32-element half of a vector) -> the two k-steps' fragments under a per-launch scale
template <int PAIR>
__device__ __forceinline__ void split_half(const float (&x)[16], float sc, Frag (&f)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u4_t h, l;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float a = x[8 * s + pair_reg<PAIR>(d, 0)] * sc, b = x[8 * s + pair_reg<PAIR>(d, 1)] * sc;
            unsigned ph, pl;
            split_pair(a, b, ph, pl);
            h[d] = ph;
            l[d] = pl;
        }
        f[s].h = __builtin_bit_cast(h8_t, h);
        f[s].l = __builtin_bit_cast(h8_t, l);
    }
}
// three-term split product step into one accumulator
__device__ __forceinline__ void mfma3(f32x16& acc, const h8_t ah, const h8_t al, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.h, acc, 0, 0, 0);
}

// ---- scatter of up to two planes by one wave (the straight-line, software-pipelined scheme of scatter_planes, with
// the two 32-slot tiles of the combine GEMM done one after the other: half the operand / accumulator registers) ----
template <int NP, class Prep>
__device__ __forceinline__ void scatter_planes_n(float* __restrict__ grad, unsigned grad_bytes, const float* Qs0,
                                                 const float* Qs1, float* M, int* tags, float* Ls, int i, int hi,
                                                 Prep&& prep) {
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(grad, 0, (int)grad_bytes, 0x00020000);
    const unsigned lane_b = 4u * (unsigned)i;
    int* const dummy = tags + 128 + i;
    PlaneRefs rc, rn;
    ClaimState sc, sn;
    prep(0, rc);
    sc = scatter_claim<false>(rc, M, tags, dummy, i);
    scatter_lost(rc, sc, Qs0, Ls, grsrc, i, hi);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const float* Qs = pl == 0 ? Qs0 : Qs1;
        int* const tg = tags + 64 * (pl & 1);
        const half_t* Mh = reinterpret_cast<const half_t*>(M);
        // B operand: 16 samples of this lane's channel, normalised per channel and split
        h8_t bh[2], bl[2];
        float bun;
        {
            float bs[2][8];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[ks][j] = Qs[(16 * ks + 8 * hi + j) * 33 + i];
            float mx = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, __builtin_fabsf(bs[ks][j]));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            int E = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
            E = E < 16 ? 16 : (E > 240 ? 240 : E);
            const float bsc = __builtin_bit_cast(float, (unsigned)(268 - E) << 23);
            bun = __builtin_bit_cast(float, (unsigned)(E - 14) << 23);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = bs[ks][2 * j] * bsc, x1 = bs[ks][2 * j + 1] * bsc;
                    h2_t ph, pq;
                    split_pair(x0, x1, ph, pq);
                    bh[ks][2 * j] = ph.x;
                    bh[ks][2 * j + 1] = ph.y;
                    bl[ks][2 * j] = pq.x;
                    bl[ks][2 * j + 1] = pq.y;
                }
        }
        // the two 32-slot tiles one after the other (with two waves per SIMD the other wave fills the gaps; the plane-to-
        // plane software pipelining of scatter_planes would cost 40 more live registers here)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            h8_t ah[2], al[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half_t* a = Mh + (32 * m + i) * M16_RS + 16 * ks + 8 * hi;
                ah[ks] = *reinterpret_cast<const h8_t*>(a);
                al[ks] = *reinterpret_cast<const h8_t*>(a + M16_PLANE);
            }
            f32x16 acc = ZERO16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acc, 0, 0, 0);
            }
            // flush: one 128-byte atomic per slot, straight from the accumulator (slot of register 4 g + e = LIDX)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const i32x4 k0 = *reinterpret_cast<const i32x4*>(tg + 32 * m + 8 * g + 4 * hi);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
#ifdef P2_DBG_NO_ATOMICS
                    asm volatile("" ::"v"(acc[4 * g + e2] * bun), "v"(k0[e2]));
#else
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[4 * g + e2] * bun, grsrc,
                                                                    (int)(((unsigned)k0[e2] << 7) | lane_b), 0, 0);
#endif
                }
            }
        }
        m_zero<false>(M, sc.w0 ? rc.h0 : 64, i);
        m_zero<false>(M, sc.w1 ? rc.h1 : 64, i);
        *(sc.m0 ? tg + rc.h0 : dummy) = sc.m0 ? -1 : -2;
        *(sc.m1 ? tg + rc.h1 : dummy) = sc.m1 ? -1 : -2;
        if (pl + 1 < NP) {
            prep(pl + 1, rn);
            sn = scatter_claim<false>(rn, M, tags + 64 * ((pl + 1) & 1), dummy, i);
            scatter_lost(rn, sn, Qs1, Ls, grsrc, i, hi);
            rc = rn;
            sc = sn;
        }
    }
}

__global__ __launch_bounds__(P2_THREADS, 2) void k_decode_bwd_tex2(BwdTexParams p) {
    __shared__ __attribute__((aligned(16))) float Lt[TEX_W_FLOATS + P2_PAIRS * P2_PAIR_FLOATS];
    {
        MlpPtrs w = p.w;
        stage_weights<PREC_S2, 64, 96>(Lt + TV1, nullptr, w.v1);
        stage_weights<PREC_S2, 64, 64>(Lt + TV2, nullptr, w.v2);
        lds_load_matrix(Lt + TV3, w.v3, 3, 64, 64);
    }
    const tt_render_cfg& cfg = p.cfg;
    // ---- per-launch operand scales (bounds as in k_decode_bwd_tex) ----
    float sKB1, sE, sK2B, sK1;
    {
        const unsigned* bnd = reinterpret_cast<const unsigned*>(p.queue) + TT_SLOT_BOUNDS;
        const float Pm = __builtin_bit_cast(float, bnd[TT_BOUND_PLANES]), Gr = __builtin_bit_cast(float, bnd[TT_BOUND_UP0]),
                    Gf = __builtin_bit_cast(float, bnd[TT_BOUND_UP1]);
        unsigned* word = reinterpret_cast<unsigned*>(Lt + TEX_W_FLOATS);  // pair regions are free until the main loop
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
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    // (the wave index stays a vector value for the compiler: made scalar with readfirstlane, the kernel spills 236 registers
    // instead of 70 -- hipcc then turns the H-dependent selects into branches and keeps more state live across them)
    const int wave = threadIdx.x >> 6;
    const int H = wave & 1;
    float* PR = Lt + TEX_W_FLOATS + (wave >> 1) * P2_PAIR_FLOATS;
    half_t* EFh = reinterpret_cast<half_t*>(PR + P2_EF);
    half_t* EFl = EFh + P2_EF_HALFS;
    half_t* XF0h = reinterpret_cast<half_t*>(PR + P2_XF0);
    half_t* XF0l = XF0h + P2_XF_HALFS;
    half_t* XF1h = reinterpret_cast<half_t*>(PR + P2_XF1);
    half_t* XF1l = XF1h + P2_XF_HALFS;
    // my dV3 transposition window (16 rows at a time) = my own blocks of XF1, before my k2bar fragments go there
    float* Win = PR + P2_XF1 + (4 * H) * (P2_BLK / 2);
    float* Cb = PR + P2_CTRL + 16 + 96 * H;  // cbar of the tile, [3][32]
    float* Tab = PR + P2_TAB + H * (3 * 16 * 8);  // my gather tables
    float* Scat = PR + H * P2_SCAT_FLOATS;
    float* M = Scat;
    float* Es = Scat + SCATTER_M_FLOATS;
    float* Ls = Es + 32 * 33;
    int* tags = reinterpret_cast<int*>(Ls + 2 * 32 * 4);
    PairCtx pc;
    pc.ctrl = reinterpret_cast<int*>(PR + P2_CTRL);
    pc.H = H;
    pc.seq = 0;
    __syncthreads();  // (block_max's last barrier already passed; this one orders the ctrl init below)
    if (lane < 8 && H == 0) pc.ctrl[lane] = 0;
    __syncthreads();
    const int S = cfg.n_samples;
    const int Hp = cfg.plane_h, Wp = cfg.plane_w;
    const size_t HW = (size_t)Hp * Wp;
    const size_t plane_stride = 6 * HW * TT_C;
    ItemQueue iq = item_queue(p.queue, tg.n_blocks, tg.n_chunks, tg.unit);
    const float shrink = cfg.rgb_grad_shrink;
    float* const grad_out = p.grad_packed + (size_t)(blockIdx.x % (unsigned)p.n_copies) * cfg.n_prompts * plane_stride;
    const unsigned grad_bytes = (unsigned)(cfg.n_prompts * plane_stride * sizeof(float));

    // weight-image rows of my half (A operands of the forward products): row 32 H + i, 8 halfs at 8 hi of every term block
    const half_t* v1row = reinterpret_cast<const half_t*>(Lt + TV1) + (size_t)(32 * H + i) * (2 * 96 + 8) + 8 * hi;
    const half_t* v2row = reinterpret_cast<const half_t*>(Lt + TV2) + (size_t)(32 * H + i) * (2 * 64 + 8) + 8 * hi;
    const float un_v1 = Lt[TV1 + 96], un_v2 = Lt[TV2 + 64];  // inverse matrix normalisations (pad of row 0)
    const lds_sv4_t* v2t = tr_lane_base<64>(Lt + TV2, 32 * H, lane);  // V2^T fragments of my 32 columns

    f32x16 accV1[3] = {Z16, Z16, Z16};  // dV1[32 H + row][32 p + col]
    f32x16 accV2[2] = {Z16, Z16};       // dV2[32 m + row][32 H + col]
    float accV3[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // dV3[o][32 H + 16 a + (lane & 15)], this lane: 8 samples
    const TileStats st = tile_stats(cfg.stats);
#ifdef TT_TUNING
    unsigned long long ph_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif
    unsigned parity = 0;  // which wave takes plane 2 of the scatter / the V1^T product (alternates per live tile step)

#pragma nounroll
    for (;;) {
        // ---- one item for the pair: wave 0 pops and posts it ----
        if (H == 0) {
            long long b0;
            int ck0;
            const bool ok = item_pop(iq, tg.order, tg.n_chunks, b0, ck0);
            if (lane == 0) {
                pc.ctrl[4] = (int)(b0 & 0xffffffffll);
                pc.ctrl[5] = (int)(b0 >> 32);
                pc.ctrl[6] = ck0;
                pc.ctrl[7] = ok ? 1 : 0;
            }
        }
        pair_sync(pc, lane);
        const long long b = (long long)(unsigned)pc.ctrl[4] | ((long long)pc.ctrl[5] << 32);
        const int ck = pc.ctrl[6];
        const bool ok = pc.ctrl[7] != 0;
        pair_sync(pc, lane);  // (both have read the mailbox before wave 0 posts the next item)
        P2_PHASE(15);
        if (!ok) break;
        if (b >= tg.n_blocks) continue;
        bool ray_ok;
        const long long ray = tile_ray(tg, b, i, ray_ok);
        const float ray_okf = tt_opaque(ray_ok ? 1.f : 0.f);
        const int ks_l = i % tg.sb;
        const int view = (int)(ray / cfg.rays_per_view);
        const size_t pofs = (size_t)(view / cfg.views_per_prompt) * plane_stride;
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
        float grgb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) grgb[o] = p.g_rgb ? p.g_rgb[ray * 3 + o] : 0.f;
        const int s_end = (ck + 1) * tg.chunk < S ? (ck + 1) * tg.chunk : S;
        // per-step inputs prefetched one tile step ahead (as in k_decode_bwd_tex): a step past the chunk reads a clamped address
        struct StepIn {
            float wgt, f[3], gf[3], ts, te;
        };
        auto load_step = [&](int sb0) {
            StepIn r;
            const int si = sb0 + ks_l;
            const long long sidx = ray * S + (si < S ? si : S - 1);
            r.wgt = p.weights[sidx];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                r.f[o] = p.features[sidx * 3 + o];
                r.gf[o] = p.g_features ? p.g_features[sidx * 3 + o] : 0.f;
            }
            r.ts = p.t_starts[sidx];
            r.te = p.t_ends[sidx];
            return r;
        };
        StepIn in = load_step(ck * tg.chunk), in_next;
#pragma nounroll
        for (int sb0 = ck * tg.chunk; sb0 < s_end; sb0 += tg.sb, in = in_next) {
            in_next = load_step(sb0 + tg.sb);
            const int si = sb0 + ks_l;
            const bool valid = ray_ok && si < s_end;
            const float vf = ray_okf * (si < s_end ? 1.f : 0.f);
            float cb[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float s = sigmoid_(in.f[o]);
                const float c = shrink * in.wgt * grgb[o] * 1.002f * s * (1.f - s) + in.gf[o];
                cb[o] = c * vf;
            }
            if (H == 0) tile_stat(st, TT_STAT_VISITED);
            P2_PHASE(0);
            if (!__any(!((__builtin_fabsf(cb[0]) + __builtin_fabsf(cb[1])) + __builtin_fabsf(cb[2]) <= cfg.skip_eps_tex)))
                continue;  // (the same decision in both waves: same data)
            float tm, px, py, pz;
            sample_position(ox, oy, oz, dx, dy, dz, in.ts, in.te, tm, px, py, pz);
            const float X = scale_coord(px, cfg.radius), Y = scale_coord(py, cfg.radius), Z = scale_coord(pz, cfg.radius);
            const int tex0 = (int)(pofs / TT_C);

            // ================= gather: my 16 samples (16 H ..), all three planes -> e fragments =================
            {
                // corner set-up: lane L < 48 <-> (sample 16 H + (L & 15), plane L >> 4)
                const int src = 16 * H + (lane & 15), pl = lane >> 4;
                const float Xs_ = __shfl(X, src), Ys_ = __shfl(Y, src), Zs_ = __shfl(Z, src);
                const float vfs = __shfl(vf, src) * (pl < 3 ? 1.f : 0.f);
                const int tex0s = __shfl(tex0, src);
                Corners cn;
                const float gu = pl == 2 ? Zs_ : Xs_, gv = pl == 1 ? Zs_ : Ys_;
                corners_setup(gu, gv, Hp, Wp, vfs != 0.f, cn);
                const unsigned long long inm = __ballot(cn.any);
                tile_stat(st, TT_STAT_INBOUNDS, (unsigned)__popcll(inm & 0x0000ffffffffffffull));
                if (lane < 48) {
                    const unsigned bb = (unsigned)tex0s + (unsigned)((3 + pl) * HW);
                    const ti32x4 o4 = {(int)(bb + cn.off[0]), (int)(bb + cn.off[1]), (int)(bb + cn.off[2]),
                                       (int)(bb + cn.off[3])};
                    *reinterpret_cast<ti32x4*>(Tab + (pl * 16 + (lane & 15)) * 8) = o4;
                    const f32x4 w4 = {cn.w[0], cn.w[1], cn.w[2], cn.w[3]};
                    *reinterpret_cast<f32x4*>(Tab + (pl * 16 + (lane & 15)) * 8 + 4) = w4;
                }
                if (lane == 0) pc.ctrl[2 + H] = inm != 0 ? 1 : 0;
                const int js = lane >> 3, c = lane & 7;
#pragma unroll
                for (int pl2 = 0; pl2 < 3; ++pl2) {
                    const bool live = ((inm >> (16 * pl2)) & 0xffffull) != 0;  // wave-uniform
                    f32x4 acc[2];
                    if (live) {
                        f32x4 t[2][4];
                        f32x4 w4[2];
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const float* T = Tab + (pl2 * 16 + 8 * n + js) * 8;
                            const ti32x4 o4 = *reinterpret_cast<const ti32x4*>(T);
                            w4[n] = *reinterpret_cast<const f32x4*>(T + 4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) t[n][k] = *gc_addr(p.packed, o4[k], 16u * (unsigned)c);
                        }
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int ee = 0; ee < 4; ++ee) a[ee] = fmaf(w4[n][k], t[n][k][ee], a[ee]);
                            acc[n] = a;
                        }
                    } else {
                        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[1] = acc[0];
                    }
                    // channels 4 c .. 4 c + 3 of plane pl2 = k-step 2 pl2 + (c >> 2), half-wave c & 1, slots 4 ((c >> 1) & 1) ..
                    const int ks = 2 * pl2 + (c >> 2), hh = c & 1, piece = (c >> 1) & 1;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int smp = 16 * H + 8 * n + js;
                        const float a0 = acc[n][0] * sE, a1 = acc[n][1] * sE, a2 = acc[n][2] * sE, a3 = acc[n][3] * sE;
                        unsigned h0, l0, h1, l1;
                        split_pair(a0, a1, h0, l0);
                        split_pair(a2, a3, h1, l1);
                        const u2_t hh2 = {h0, h1}, ll2 = {l0, l1};
                        const int off = (ks * 2 + hh) * P2_BLK + smp * 8 + 4 * piece;
                        *reinterpret_cast<u2_t*>(EFh + off) = hh2;
                        *reinterpret_cast<u2_t*>(EFl + off) = ll2;
                    }
                }
            }
            P2_PHASE(1);
            pair_sync(pc, lane);  // (1) e fragments + in-bounds flags of both halves
            P2_PHASE(2);
            if ((pc.ctrl[2] | pc.ctrl[3]) == 0) {
                pair_sync(pc, lane);  // both have read the flags before the next gather rewrites them
                continue;             // exact: e == 0 for the whole tile
            }
            if (H == 0) tile_stat(st, TT_STAT_EXECUTED);

            // ================= k1 half = relu(V1[32 H .., :] e) =================
            unsigned n1 = 0;  // bit r: k1[r] > 0 (the ReLU mask k1bar needs; the values live on as fragments only)
            {
                float k1[16];
                f32x16 a0 = Z16;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const h8_t ah = *reinterpret_cast<const h8_t*>(v1row + 32 * s);
                    const h8_t al = *reinterpret_cast<const h8_t*>(v1row + 32 * s + 16);
                    const Frag bf = frag_lds(EFh, EFl, s, hi, i);
                    mfma3(a0, ah, al, bf);
                }
                const float un = un_v1 / sE;
#pragma unroll
                for (int r = 0; r < 16; ++r) k1[r] = fmaxf(a0[r] * un, 0.f);
#pragma unroll
                for (int r = 0; r < 16; ++r) n1 |= (k1[r] > 0.f ? 1u : 0u) << r;
                Frag k1f[2];
                split_half<PAIR_SEQ>(k1, sK1, k1f);
                frag_store(XF0h, XF0l, 2 * H + 0, hi, i, k1f[0]);
                frag_store(XF0h, XF0l, 2 * H + 1, hi, i, k1f[1]);
            }
            P2_PHASE(3);
            pair_sync(pc, lane);  // (2) k1 fragments
            P2_PHASE(4);

            // ================= k2 half = relu(V2[32 H .., :] k1); dV3; k2bar half =================
            {
                float k2[16];
                f32x16 a0 = Z16;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const h8_t ah = *reinterpret_cast<const h8_t*>(v2row + 32 * s);
                    const h8_t al = *reinterpret_cast<const h8_t*>(v2row + 32 * s + 16);
                    mfma3(a0, ah, al, frag_lds(XF0h, XF0l, s, hi, i));
                }
                const float un = un_v2 / sK1;
#pragma unroll
                for (int r = 0; r < 16; ++r) k2[r] = fmaxf(a0[r] * un, 0.f);
                // dV3[o][32 H + idx] += sum_s cbar_o[s] k2[idx][s]: my 32 rows through the window, 16 at a time; lane
                // (row = lane & 15, quarter = lane >> 4) sums the samples 8 quarter .. 8 quarter + 7 of its row
                if (hi == 0) {
                    Cb[0 * 32 + i] = cb[0];
                    Cb[1 * 32 + i] = cb[1];
                    Cb[2 * 32 + i] = cb[2];
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) Win[(LIDX(8 * a + r, hi) - 16 * a) * XS + i] = k2[8 * a + r];
                    const int row = lane & 15, qt = lane >> 4;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const f32x4 kk = *reinterpret_cast<const f32x4*>(Win + row * XS + 8 * qt + 4 * g);
#pragma unroll
                        for (int o = 0; o < 3; ++o) {
                            const f32x4 cc = *reinterpret_cast<const f32x4*>(Cb + o * 32 + 8 * qt + 4 * g);
                            accV3[a][o] += (kk[0] * cc[0] + kk[1] * cc[1]) + (kk[2] * cc[2] + kk[3] * cc[3]);
                        }
                    }
                }
                // k2bar = n2 . (V3^T cbar), my 32 elements
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 0 * 64 + 32 * H + 8 * g + 4 * hi);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 1 * 64 + 32 * H + 8 * g + 4 * hi);
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(Lt + TV3 + 2 * 64 + 32 * H + 8 * g + 4 * hi);
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const float t = fmaf(v0[e2], cb[0], fmaf(v1[e2], cb[1], v2[e2] * cb[2]));
                        k2[4 * g + e2] = k2[4 * g + e2] > 0.f ? t : 0.f;
                    }
                }
                Frag k2bf[2];
                split_half<PAIR_TR>(k2, sK2B, k2bf);  // (consumed by the transposed product V2^T k2bar and the outer products)
                frag_store(XF1h, XF1l, 2 * H + 0, hi, i, k2bf[0]);  // (over my window: its reads are done -- same wave, in order)
                frag_store(XF1h, XF1l, 2 * H + 1, hi, i, k2bf[1]);
            }
            P2_PHASE(5);
            pair_sync(pc, lane);  // (3) k2bar fragments
            P2_PHASE(6);

            // ================= dV2[32 m + tr_elem(row)][32 H + col] += k2bar k1^T =================
            // (operands read transposed from the fragment images: my k1 fragments in XF0, all of k2bar in XF1)
#ifndef P2_DBG_NO_OUTER
            {
                unsigned XT[16], YT[16];
                tr_operand(XF0h, XF0l, H, lane, YT);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    tr_operand(XF1h, XF1l, m, lane, XT);
                    outer16(accV2[m], XT, YT);
                }
            }
#endif

            P2_PHASE(7);
            // ================= k1bar half = n1 . (V2[:, 32 H ..]^T k2bar) =================
            {
                f32x16 a0 = Z16;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const h8_t ah = tr_frag<64>(v2t, s, 0, 0), al = tr_frag<64>(v2t, s, 0, 1);
                    mfma3(a0, ah, al, frag_lds(XF1h, XF1l, s, hi, i));
                }
                const float un = un_v2 / sK2B;
                float kb1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) kb1[r] = ((n1 >> r) & 1u) ? a0[r] * un : 0.f;
                Frag kb1f[2];
                split_half<PAIR_TR>(kb1, sKB1, kb1f);
                frag_store(XF0h, XF0l, 2 * H + 0, hi, i, kb1f[0]);  // (over MY k1 fragments: only I read those after (3))
                frag_store(XF0h, XF0l, 2 * H + 1, hi, i, kb1f[1]);
            }
            P2_PHASE(8);
            pair_sync(pc, lane);  // (4) k1bar fragments
            P2_PHASE(9);

            // ================= dV1[32 H + tr_elem(row)][32 pl + col] += k1bar e^T =================
#ifndef P2_DBG_NO_OUTER
            {
                unsigned XT[16], YT[16];
                tr_operand(XF0h, XF0l, H, lane, XT);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    tr_operand(EFh, EFl, pl, lane, YT);
                    outer16(accV1[pl], XT, YT);
                }
            }
#endif

            P2_PHASE(10);
            // ================= ebar of my planes = V1[:, 32 pl ..]^T k1bar =================
            const bool two = ((parity & 1u) == (unsigned)H);  // this step I take plane 2 as well
            ++parity;
            auto refs_of = [&](int pl, PlaneRefs& refs) {
                Corners c;
                corners_setup(PLANE_U(pl, X, Y, Z), PLANE_V(pl, X, Y, Z), Hp, Wp, valid, c);
                int aoff[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) aoff[q4] = tex0 + (int)((3 + pl) * HW) + c.off[q4];
                refs = plane_refs<true>(c.w, aoff, c.hs, hi);
            };
            auto ebar = [&](int pl, float (&out)[16]) {
                const lds_sv4_t* v1t = tr_lane_base<96>(Lt + TV1, 32 * pl, lane);
                f32x16 a0 = Z16;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const h8_t ah = tr_frag<96>(v1t, s, 0, 0), al = tr_frag<96>(v1t, s, 0, 1);
                    mfma3(a0, ah, al, frag_lds(XF0h, XF0l, s, hi, i));
                }
                const float un = un_v1 / sKB1;
#pragma unroll
                for (int r = 0; r < 16; ++r) out[r] = a0[r] * un;
            };
            float* Es2 = PR + P2_ES2;
            if (two) {  // plane 2's Q rows go straight to LDS (scaled per sample like its coefficients): nothing of them stays live
                float e2[16];
                ebar(2, e2);
                PlaneRefs r2;
                refs_of(2, r2);
#pragma unroll
                for (int r = 0; r < 16; ++r) Es2[i * 33 + LIDX(r, hi)] = e2[r] * r2.qs;
            }
            float eb0[16];
            ebar(H, eb0);
            P2_PHASE(11);
            pair_sync(pc, lane);  // (5) both are done with the fragment buffers: the scatter regions may overwrite them

            P2_PHASE(12);
            // ================= scatter of my planes =================
#ifndef P2_DBG_NO_SCATTER
            {
                scatter_clear<false>(M, lane);
                scatter_init_tags(tags, lane);
                auto prep = [&](int q, PlaneRefs& refs) {
                    refs_of(q == 0 ? H : 2, refs);
                    if (q == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Es[i * 33 + LIDX(r, hi)] = eb0[r] * refs.qs;
                    }
                };
                if (two)
                    scatter_planes_n<2>(grad_out, grad_bytes, Es, Es2, M, tags, Ls, i, hi, prep);
                else
                    scatter_planes_n<1>(grad_out, grad_bytes, Es, Es2, M, tags, Ls, i, hi, prep);
            }
#else
            accV3[0][0] += eb0[0];
#endif
            P2_PHASE(13);
            pair_sync(pc, lane);  // (6) scatter regions are free again: the next gather may write the fragment buffers
#ifdef TT_TUNING
            if (two) P2_PHASE(17); else P2_PHASE(14);
#endif
        }
    }

    // ---- flush my accumulators: dV1 rows 32 H + tr_elem(.), dV2 rows 32 m + tr_elem(.) / columns 32 H .., dV3 columns 32 H ..
    {
        const float u1 = 1.f / sKB1, u2 = 1.f / sE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                atomicAdd(p.grads.v1 + (32 * H + tr_elem(LIDX(r, hi))) * 96 + 32 * pl + i, (accV1[pl][r] * u1) * u2);
        const float w1 = 1.f / sK2B, w2 = 1.f / sK1;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                atomicAdd(p.grads.v2 + (32 * m + tr_elem(LIDX(r, hi))) * 64 + 32 * H + i, (accV2[m][r] * w1) * w2);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int o = 0; o < 3; ++o) atomicAdd(p.grads.v3 + o * 64 + 32 * H + 16 * a + (lane & 15), accV3[a][o]);
    }
    tile_stats_flush(st);
#ifdef TT_TUNING
    P2_PHASE(16);
    if (p.phase_cycles && lane == 0)
        for (int k = 0; k < 20; ++k) atomicAdd(p.phase_cycles + k, ph_acc[k]);
#endif
}

// launch hook used by tt_backward_tex.hip (the bounds reductions have been enqueued by the caller)
void tt_launch_bwd_tex2(const BwdTexParams& p, int cus, hipStream_t s) {
    long long blocks = cus;  // one workgroup of P2_PAIRS pairs per CU
    const long long need = (p.n_items + P2_PAIRS - 1) / P2_PAIRS;
    if (blocks > need) blocks = need;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL(k_decode_bwd_tex2, dim3((unsigned)blocks), dim3(P2_THREADS), 0, s, p);
}

#endif  // TT_TUNING
