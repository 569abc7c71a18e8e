/*
 * tt_abi.h -- C ABI of libtt_hip.so: the MI355X (gfx950) triplane volume-render hot path.
 *
 * This is the drop-in boundary of triplaneturbo_amd.  Every entry point is plain C:
 * raw DEVICE pointers (fp32 unless noted), explicit sizes, a config struct, a hipStream_t
 * passed as void*.  The caller (PyTorch-ROCm host code, or any FFI) allocates every output
 * and workspace; the library itself keeps one 266 KB device scratch per GPU (work-queue counters of the per-sample
 * kernels: one 64-byte slot per stream, zeroed on the stream (by a one-wave kernel) in front of every launch, and a never-reused slot per
 * launch recorded under stream capture, so a launch can be captured in a hipGraph; allocated at the first launch,
 * which therefore must not be inside a capture); no environment variable is read; re-entrant; safe from one
 * host thread per device.  Return 0 on success, a negative tt_status on error (no C++
 * exceptions cross the boundary).  tt_strerror() maps codes to text.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *
 *   tt_planes_pack          few_step_triplane_dual_stable_diffusion.py:212-239 (rotate_planes "v1" copy)
 *                           + geometry/utils.py:131 (view as N*n_planes,C,H,W); produces the channels-last,
 *                           rotation-folded plane image the kernels gather from.
 *   tt_planes_unpack_grad   the autograd transpose of the above (d loss / d space_cache, NCHW).
 *   tt_query_points         StableDiffusionTriplaneDualAttention.forward   few_step...:273-351
 *                           (= sample_from_planes geometry/utils.py:127-145 -> aten grid_sampler_2d,
 *                            VanillaMLP networks.py:67-104, get_shifted_sdf :131-154, analytic normal :329-335
 *                            -> aten grid_sampler_2d_backward via cuda_gridsample.py:55-58)
 *                           and forward_sdf :353-373 (flags without TT_Q_TEX / TT_Q_NORMAL).
 *   tt_query_field          forward_field :375-394 (sdf + deformation head; mesh renderer / exporter grid query)
 *   tt_decode_rays          the geometry call of prop_sigma_fn (renderer :243-299 -> few_step...:273-306, sdf only)
 *   tt_sample_uniform       ImportanceEstimator.sampling level 0 (threestudio/models/estimators.py:61-79 with
 *                           _transform_stot "uniform" :104-118): n equal / stratified intervals on [near, far].
 *   tt_sample_importance    the rest of ImportanceEstimator.sampling (estimators.py:72-101) with prop_sigma_fn's
 *                           fixed-step NeuS density (renderer :288-297): nerfacc render_transmittance_from_density,
 *                           importance_sampling (inverse-CDF resample) and the merge + sort of the edges (:317-324).
 *   tt_render_fwd           GenerativeSpaceSDFVolumeRenderer._forward
 *                           generative_space_sdf_volume_renderer.py:326-431,467-472 (positions, geometry,
 *                           NoMaterial no_material.py:41-54, get_alpha neus_volume_renderer.py:93-117,
 *                           nerfacc.render_weight_from_alpha, nerfacc.accumulate_along_rays x5).
 *   tt_render_eval          the same in eval mode (per-ray outputs only): decode + march fused per ray tile with
 *                           wave-ballot early termination / texture-decode skipping
 *   tt_march_fwd / _bwd     the ray march alone (second half of tt_render_fwd / first half of tt_render_bwd_geo):
 *                           get_alpha neus_volume_renderer.py:93-117 + nerfacc.render_weight_from_alpha +
 *                           nerfacc.accumulate_along_rays x5 (renderer :407-431,467-472) on given per-sample
 *                           sdf / sdf_grad / features, and its backward.  Bandwidth-bound.
 *   tt_render_bwd_geo /     the autograd backward of the same, incl. the second-order terms the reference
 *   tt_render_bwd_tex       obtains from gridsample_cuda.cu:27-210 (grad2_2d, cuda_gridsample.py:68-79),
 *                           aten grid_sampler_2d_backward and the transposed cuBLAS GEMMs.
 *   tt_points_bwd_x         the same backward w.r.t. the query points themselves (grad_grid of both grid_sample
 *                           backwards + K1's grad_grid, gridsample_cuda.cu:196-208)
 *   tt_points_bwd_geo /     the autograd backward of tt_query_points / tt_query_field w.r.t. planes and MLP weights
 *   tt_points_bwd_tex       (training-time callers: generative_space_mesh_rasterize_renderer.py:428-452 field
 *                           query, :321-376 per-pixel geometry decode); same kernels as tt_render_bwd_*.
 *   tt_composite_fwd/_bwd   the renderer's per-ray composite: comp_rgb, disparity, comp_normal, camera-space normal
 *                           maps (generative_space_sdf_volume_renderer.py:433-530)
 *   tt_patch_composite_*    PatchRenderer.forward's upsample + paste per output key (patch_renderer.py:74-88)
 *   tt_hashgrid_fwd / _bwd  tiny-cuda-nn's `HashGrid` encoding as used by the background
 *                           (multi_prompt_neural_environment_hashgrid_map_background.py:25-34,54,104-105 via
 *                           threestudio/models/networks.py:17-26,54-64); tcnn is CUDA-only and un-vendored.
 *   tt_grid_sample_2d_grad2 gridsample_cuda.cpp:26-37 `grad2_2d` itself (operator-level drop-in; _typed: half / float /
 *   (_typed)                double, zeros / border padding, either align_corners, as gridsample_cuda.cu:560-594).
 */
#ifndef TT_ABI_H
#define TT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_ABI_VERSION 17
#define TT_CHANNELS 32 /* feature channels per plane (space_generator output_dim/2, yaml :95) */
#define TT_HIDDEN 64   /* mlp_network_config.n_neurons */

typedef enum {
    TT_OK = 0,
    TT_ERR_BAD_ARG = -1,     /* null pointer / non-positive size / unsupported shape */
    TT_ERR_UNSUPPORTED = -2, /* e.g. non-square planes */
    TT_ERR_LAUNCH = -3,      /* hipLaunch / hipGetLastError failure */
    TT_ERR_DEVICE = -4       /* not a gfx950 device / attribute query failed */
} tt_status;

/* MLP weights exactly as torch stores nn.Linear(bias=False).weight: row-major (out, in).
 * sdf net  : w1 (64,32)  w2 (64,64)  w3 (1,64)     few_step...:101-105
 * feat net : v1 (64,96)  v2 (64,64)  v3 (3,64)     few_step...:106-112  (tex_interpolate v2 => 96 inputs) */
typedef struct {
    const float* w1;
    const float* w2;
    const float* w3;
    const float* v1;
    const float* v2;
    const float* v3;
} tt_mlp_weights;

typedef struct {
    float* w1;
    float* w2;
    float* w3;
    float* v1;
    float* v2;
    float* v3;
} tt_mlp_grads; /* accumulated into (+=) with fp32 atomics; caller zero-initialises */

typedef struct {
    int32_t n_prompts;         /* P: planes are (P,6,H,W,32) packed */
    int32_t views_per_prompt;  /* view b samples prompt b / views_per_prompt  (renderer :120-143) */
    int32_t plane_h, plane_w;  /* must be equal (rotation v1 transposes) */
    int32_t rays_per_view;     /* Hh*Ww; ray r belongs to view r / rays_per_view */
    int32_t n_samples;         /* S samples per ray (dense layout, renderer :317-324) */
    int64_t n_rays;            /* total rays = views * rays_per_view */
    float radius;              /* geometry.radius: bbox = [-radius, radius]^3 */
    float sdf_bias_radius;     /* sdf_bias "sphere", sdf_bias_params (yaml :78-79) */
    float inv_std;             /* LearnedVariance.inv_std, clamped to [1e-6,1e6] (renderer :29-35) */
    float cos_anneal_ratio;    /* neus_volume_renderer.py:101-104 */
    float rgb_grad_shrink;     /* renderer :397-400 (backward only) */
    int32_t flags;             /* TT_R_* */
    int32_t image_w;           /* rays of a view form an image_w x (rays_per_view/image_w) image (pixel-block tiles);
                                  0 = unknown: tiles are runs of consecutive rays */
    int32_t tile_sb;           /* consecutive samples of one ray per 32-sample tile: 1, 2, 4, ... 32; 0 = default (2).
                                  1-2 suit evenly spaced samples; 8 suits importance sampling, where consecutive samples
                                  of a ray share texels and are then combined inside the tile (performance only:
                                  results do not depend on it beyond fp32 summation order) */
    int32_t grad_copies;       /* backward: grad_packed holds this many privatised copies (copies,P,6,H,W,32), each
                                  workgroup scatters into one of them and tt_planes_unpack_grad sums them; spreads
                                  same-texel atomics.  0/1 = a single copy */
    int32_t tile_chunk;        /* samples of a ray block per work item of the dynamic queue; 0 = automatic (performance
                                  only, like tile_sb) */
    float skip_eps_tex;        /* backward, OPT-IN approximation (0 = exact, the default): a 32-sample tile whose upstream
                                  colour gradients all satisfy |cbar|_1 <= skip_eps_tex is skipped by tt_render_bwd_tex
                                  (cbar = d loss / d raw feature = shrink w g_rgb 1.002 s (1 - s) + g_features: samples in
                                  empty space carry weights ~1e-5 and almost no gradient).  Induced error of d/d texture
                                  planes and d/d feature net: at most skip_eps_tex x (skipped samples) x the local
                                  sensitivity; measured at the training shapes in tests/test_gpu_skip.py. */
    float skip_eps_geo;        /* the same for tt_render_bwd_geo on |d loss/d sdf| + |d loss/d sdf_grad|_1 per sample (dense
                                  under an eikonal loss, so usually nothing to skip there) */
    const float* inv_std_dev;  /* null, or a DEVICE pointer to one float that replaces `inv_std` (clamped to [1e-6,1e6] in the
                                  kernels like LearnedVariance.forward, renderer :34-35): trainable_variance=True
                                  (renderer :53,82; neus_volume_renderer.py:26-37) without a host read-back per step.  Read
                                  by tt_render_fwd / _bwd_geo, tt_march_fwd / _bwd and tt_render_eval */
    uint64_t* stats;           /* null, or a DEVICE pointer to 4 x uint64 the caller zero-fills: work accounting of the decode
                                  kernel of tt_render_fwd / tt_decode_rays / tt_render_bwd_geo / tt_render_bwd_tex, added to
                                  with one atomic per wave: [0] 32-sample tile steps visited, [1] tile steps EXECUTED (those
                                  that pass the exact skip tests: some texel in bounds -- a tile without one decodes to exact
                                  zeros --, and in the backward some non-zero upstream gradient), [2] (plane, sample) pairs
                                  with an in-bounds texel over the gathers that ran (3 planes per sample; the samples a launch
                                  visits are exactly n_rays * n_samples), [3] reserved.  Measurement only (bench.py:
                                  live_tile_frac, inbounds_plane_frac) */
} tt_render_cfg;

#define TT_R_PER_SAMPLE 1 /* also write per-sample sdf / sdf_grad / features (training extras, renderer :532-545) */
/* ---- precision of the matrix products (the per-point MLPs; everything else is plain fp32 in every mode) ----
 * The reference multiplies in fp32 (threestudio/models/networks.py:91-97: nn.Linear with autocast disabled; trainer
 * precision 32, configs/TriplaneTurbo_v1.yaml:254).  Three modes; at most one of the three bits may be set, none = the
 * default = TT_R_SPLIT3:
 *   TT_R_SPLIT3     fp32-GRADE products on the fp16 matrix pipe: every operand is split EXACTLY in three fp16 pieces
 *                   (hi + mid + lo = v), the six product terms above 2^-33 are accumulated in fp32 (6 x
 *                   v_mfma_f32_32x32x16_f16 per k-step).  Product error <= 2^-24 of sum |a b| (tools/mfma16_probe.hip):
 *                   the reference's precision at ~1/3 of the fp32 MFMA's matrix-pipe time.  SCOPE: the mat-vec chains of
 *                   every kernel (forward, recompute, both gradient chains, per-point queries, eval, d/d points) are
 *                   three-piece; the REDUCTIONS OVER SAMPLES of the backward -- the weight-gradient outer products and
 *                   the scatter's combine GEMM -- use TWO-piece operands (hi + lo, all four cross terms, per-launch /
 *                   per-sample power-of-two scales): each operand is represented to 2^-23 and the round-to-nearest
 *                   errors average over the thousands to millions of samples such a sum runs over (measured: weight
 *                   and plane gradients 4e-7 ... 1e-6 from the fp32 oracle in all three modes).
 *   TT_R_EXACT_F32  every product on the fp32-input MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain).  The A/B
 *                   reference of the split modes.
 *   TT_R_SPLIT2     the FAST mode (the default of rounds 2-4): two fp16 pieces per operand, three product terms, ~2^-21.5
 *                   per product -- a tolerance-bounded approximation (gradients within 1e-4 of the fp32 math on
 *                   well-conditioned scenes, not on every scene: DESIGN.md section 6). */
#define TT_R_EXACT_F32 2
#define TT_R_SPLIT2 32
#define TT_R_SPLIT3 64

/* use_volsdf = True of the reference (threestudio/models/renderers/neus_volume_renderer.py:19-23,:95-96 and
 * generative_space_sdf_volume_renderer.py:286-287): alpha = |t_end - t_start| x density(sdf), density = k (0.5 + 0.5
 * sign(sdf) expm1(-|sdf| k)) with k = inv_std clamped to [0, 80]; NOT clipped to [0,1] and independent of the normal and of
 * cos_anneal_ratio.  Honoured by tt_render_forward / _backward, the march entry points and the fused eval render. */
#define TT_R_VOLSDF 128

#define TT_R_WGRAD_F32 4  /* TUNING BUILD ONLY (-DTT_TUNING; the product library returns TT_ERR_UNSUPPORTED): backward
                             weight-gradient outer products on the fp32-input MFMA instead of split-fp16 products with
                             per-launch operand scales; the round-2 A/B switch, TT_R_SPLIT2 only */
#define TT_R_BWD_SOLO 8   /* backward: force the one-wave-per-tile decode kernels (the default) */
#define TT_R_BWD_PAIR 16  /* RESERVED, always TT_ERR_UNSUPPORTED: selected the experimental wave-pair texture kernel of round 4
                             (two waves per SIMD; correct and 1.4x slower; removed from the tree in round 6) */

/* tt_query_points / tt_query_field / tt_decode_rays / tt_points_bwd_* flags */
#define TT_Q_NORMAL 1    /* output sdf_grad (analytic normal path) */
#define TT_Q_TEX 2       /* output features (texture planes + feature net) */
#define TT_Q_EXACT_F32 4 /* as TT_R_EXACT_F32 */
#define TT_Q_SPLIT2 8    /* as TT_R_SPLIT2 */
#define TT_Q_SPLIT3 16   /* as TT_R_SPLIT3 (the default) */

const char* tt_strerror(int status);
int tt_abi_version(void);
/* sha256 (hex) of the sources and build flags the library was built from ("unknown" for a hand-made build): the host
 * side rebuilds when it differs from the tree (triplaneturbo_amd/_lib.py), whatever the file times say. */
const char* tt_source_hash(void);
/* Test hook: leaves the work-queue counters of `stream` dirty, as a faulted kernel would; the next launch on that
 * stream must be unaffected (the counters are zeroed on the stream in front of every launch). */
int tt_debug_poison_queue(void* stream);

/* space_cache (P,6,32,H,W) NCHW  ->  packed (P,6,H,W,32), planes re-oriented per rotate_planes "v1". */
int tt_planes_pack(const float* space_cache, float* packed, int32_t n_prompts, int32_t plane_h, int32_t plane_w,
                   void* stream);
/* grad wrt packed, n_copies privatised copies (n_copies,P,6,H,W,32) -> their sum as grad wrt space_cache
 * (P,6,32,H,W); overwrites dst. */
int tt_planes_unpack_grad(const float* grad_packed, float* grad_space_cache, int32_t n_prompts, int32_t plane_h,
                          int32_t plane_w, int32_t n_copies, void* stream);

/* Per-point decode.  points (n_batch, n_points, 3) world units; batch b reads prompt b / views_per_prompt.
 * out_sdf (n_batch*n_points), out_sdf_grad (n_batch*n_points,3) [if TT_Q_NORMAL], out_features (.,3) [if TT_Q_TEX].
 * Null outputs are skipped. */
int tt_query_points(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                    int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h, int32_t plane_w,
                    float radius, float sdf_bias_radius, int32_t flags, float* out_sdf, float* out_sdf_grad,
                    float* out_features, void* stream);

/* Implicit-field query for isosurface extraction (forward_field, few_step...:375-394; callers
 * generative_space_mesh_rasterize_renderer.py:428-452 and triplaneturbo_executable/utils/mesh_exporter.py:78-105):
 * sdf (n_batch*n_points) and deformation (n_batch*n_points,3) from the geometry planes only.
 * `w`: sdf net in w1..w3, DEFORMATION net (32->64->64->3, few_step...:113-122) in v1..v3.  flags: TT_Q_EXACT_F32 or 0. */
int tt_query_field(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                   int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h, int32_t plane_w,
                   float radius, float sdf_bias_radius, int32_t flags, float* out_sdf, float* out_deformation,
                   void* stream);

/* Decode only, along rays: sdf [+ sdf_grad if TT_Q_NORMAL] [+ features if TT_Q_TEX] at the mid-points of the
 * intervals (n_rays,S).  The importance sampler's proposal pass (prop_sigma_fn, renderer :243-299) uses flags = 0. */
int tt_decode_rays(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                   const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, int32_t flags, float* sdf,
                   float* sdf_grad, float* features, void* stream);

/* Where the n + 1 edges of a level are placed in cdf space (u in [0,1]) before going through the inverse cdf.  The
 * reference draws both levels with nerfacc v0.5.2 `importance_sampling(intervals, cdfs, n, stratified)`
 * (threestudio/models/estimators.py:72-90); nerfacc is un-vendored and its pdf.cu is not available here, so its exact
 * convention is UNVERIFIABLE in this build and the choice is an explicit, switchable contract:
 *   TT_PLACE_TT     (default) u_j = j / n, j = 0..n: first / last edge pinned to near / far.  Stratified: level-0
 *                   interior edges -+ half a cell ((jitter - 0.5) / n), fine level u_j + jitter_j / n clamped to [0,1].
 *   TT_PLACE_CENTER u_j = (j + 0.5) / (n + 1): the centres of n + 1 equal cells, nothing pinned to 0 or 1.
 *                   Stratified: u_j = (j + jitter_j) / (n + 1), one uniform draw per cell.
 * Under either one the empirical distribution of the resampled edges follows the proposal cdf within one cell
 * (tests/test_gpu_sampler.py checks that against the cdf itself, not against a restatement of the kernel). */
enum tt_sample_placement { TT_PLACE_TT = 0, TT_PLACE_CENTER = 1 };
/* OR-ed into tt_sample_importance's `placement`: the proposal density is the VolSDF density of TT_R_VOLSDF (renderer
 * :286-287) instead of the fixed-step NeuS density (:288-297) */
#define TT_PLACE_VOLSDF 0x100

/* Level-0 sample intervals from the uniform cdf: edges s_k = u_k of `placement` (jitter (n_rays, n+1), U[0,1), or
 * null = deterministic), t = s*far + (1-s)*near (_transform_stot "uniform", estimators.py:104-118);
 * t_starts/t_ends (n_rays, n). */
int tt_sample_uniform(int64_t n_rays, int32_t n_samples, float near_plane, float far_plane, const float* jitter,
                      int32_t placement, float* t_starts, float* t_ends, void* stream);

/* Importance resampling of one proposal level.  In: proposal intervals t_starts/t_ends (n_rays, K) and the sdf
 * (n_rays, K) at their mid-points (tt_decode_rays, flags = 0).  sigma = NeuS alpha over a fixed step / step (or the
 * VolSDF density: placement | TT_PLACE_VOLSDF),
 * T = exp(-exclusive_cumsum(sigma dt)), cdf = 1 - [T, 0]; F + 1 fine edges at the u_j of `placement` (u_jitter
 * (n_rays, F+1), U[0,1), or null = deterministic) through the piecewise-linear inverse cdf; out = the K + F + 2
 * edges merged in increasing order as out_t_starts/out_t_ends (n_rays, K + F + 1).  inv_std_dev: null, or a device scalar
 * that replaces inv_std (as tt_render_cfg.inv_std_dev). */
int tt_sample_importance(const float* t_starts, const float* t_ends, const float* sdf, int64_t n_rays,
                         int32_t n_proposal, int32_t n_fine, float inv_std, const float* inv_std_dev,
                         float render_step_size, const float* u_jitter, int32_t placement, float* out_t_starts,
                         float* out_t_ends, void* stream);

/* Forward render for explicit sample intervals.
 * rays_o, rays_d (n_rays,3); t_starts, t_ends (n_rays,S).
 * Per-ray outputs: opacity (n_rays), depth (n_rays), rgb_fg (n_rays,3), z_variance (n_rays),
 *                  normal_acc (n_rays,3) = sum_i w_i n_i (NOT normalised).
 * Per-sample outputs (n_rays*S), all required: weights, trans, sdf, sdf_grad (.,3), features (.,3).  The decode
 *   kernel writes sdf/sdf_grad/features, the march kernel reads them back (they are also the renderer's
 *   training extras, renderer :532-545, and the saved state of the backward). */
int tt_render_fwd(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                  const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, float* opacity, float* depth,
                  float* rgb_fg, float* z_variance, float* normal_acc, float* weights, float* trans, float* sdf,
                  float* sdf_grad, float* features, void* stream);

/* Eval-mode render (the renderer returns no per-sample tensors outside training, renderer :532-545): the per-ray outputs
 * of tt_render_fwd from ONE kernel that decodes and marches each 8x4-pixel ray tile front to back.
 * transmittance_eps > 0: a ray stops contributing once its transmittance is below it and a tile stops when all its rays
 * have (wave ballot); weight_eps > 0: the texture decode of a tile step is skipped unless some ray has a larger weight,
 * and runs for those rays only.  Induced error: opacity / rgb < transmittance_eps + S * weight_eps per ray (depth: x far).
 * Both 0: no approximation (the skips that remain are exact).  stats (device, 2 x uint64, may be null; caller zero-fills):
 * += wave tile steps with a geometry decode / with a texture decode.  No gradients: eval only. */
int tt_render_eval(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                   const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, float transmittance_eps,
                   float weight_eps, float* opacity, float* depth, float* rgb_fg, float* z_variance, float* normal_acc,
                   uint64_t* stats, void* stream);

/* The ray march alone, on per-sample sdf (n_rays*S), sdf_grad (.,3), features (.,3) that the caller already has
 * (tt_decode_rays / tt_query_points): per-ray and per-sample outputs as in tt_render_fwd. */
int tt_march_fwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                 const float* sdf, const float* sdf_grad, const float* features, float* opacity, float* depth,
                 float* rgb_fg, float* z_variance, float* normal_acc, float* weights, float* trans, void* stream);

/* Backward of tt_march_fwd down to the per-sample quantities: out_grad (n_rays*S,4) = (d/d sdf, d/d sdf_grad xyz),
 * upstream grads as in tt_render_bwd_geo (null = 0).  (d/d features = weights * g_rgb_fg * d sigmoid is formed inside
 * tt_render_bwd_tex.)  This is the `workspace` tt_render_bwd_geo fills for its decode backward.
 * g_inv_std_rays (n_rays, may be null; overwritten): d loss / d inv_std PER RAY (both logistic arguments of get_alpha are
 * sdf estimates times inv_std, neus_volume_renderer.py:108-109); the caller sums the rays (fixed order: reproducible) and
 * chains through its own parametrisation (LearnedVariance: inv_std = exp(10 p) clamped).  trainable_variance=True. */
int tt_march_bwd(const float* rays_d, const float* t_starts, const float* t_ends, const tt_render_cfg* cfg,
                 const float* opacity, const float* depth, const float* trans, const float* sdf,
                 const float* sdf_grad, const float* features, const float* g_opacity, const float* g_depth,
                 const float* g_rgb_fg, const float* g_z_variance, const float* g_normal_acc, const float* g_weights,
                 const float* g_sdf, const float* g_sdf_grad, float* g_inv_std_rays, float* out_grad, void* stream);

/* Backward, geometry half: d/d(geometry planes 0..2) and d/d(sdf net).
 * Per-ray upstream grads (any may be null = 0): g_opacity, g_depth, g_rgb_fg(3), g_z_variance, g_normal_acc(3).
 * Per-sample upstream grads (null = 0): g_weights, g_sdf, g_sdf_grad(3).
 * Saved forward state: opacity, depth (per ray), trans, sdf, sdf_grad, features (per sample).
 * g_inv_std_rays: as in tt_march_bwd (null unless the variance is trained).
 * workspace: n_rays*S*4 floats (written by the march backward, read by the decode backward).
 * grad_packed (P,6,H,W,32) and mlp grads are accumulated into (caller zero-fills).  The packed planes (= one copy of
 * grad_packed) must be smaller than 4 GB - 256 B (85 prompts of 256^2 planes), else TT_ERR_UNSUPPORTED -- the limit
 * applies to every entry point that takes a tt_render_cfg or a packed-planes pointer (32-bit texel byte offsets). */
int tt_render_bwd_geo(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                      const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, const float* opacity,
                      const float* depth, const float* trans, const float* sdf, const float* sdf_grad,
                      const float* features, const float* g_opacity, const float* g_depth, const float* g_rgb_fg,
                      const float* g_z_variance, const float* g_normal_acc, const float* g_weights,
                      const float* g_sdf, const float* g_sdf_grad, float* g_inv_std_rays, float* workspace,
                      float* grad_packed, const tt_mlp_grads* grads, void* stream);

/* Backward, texture half: d/d(texture planes 3..5) and d/d(feature net).
 * Needs saved weights (per sample) and features; g_rgb_fg (per ray), g_features (per sample; null = 0). */
int tt_render_bwd_tex(const float* packed, const tt_mlp_weights* w, const float* rays_o, const float* rays_d,
                      const float* t_starts, const float* t_ends, const tt_render_cfg* cfg, const float* weights,
                      const float* features, const float* g_rgb_fg, const float* g_features, float* grad_packed,
                      const tt_mlp_grads* grads, void* stream);

/* Backward of the per-point queries.  points (n_batch, n_points, 3) as in tt_query_points
 * (constants here: the gradient w.r.t. the points is tt_points_bwd_x)
 * _geo: upstream g_sdf (n) and/or g_sdf_grad (n,3) (one may be null) -> d/d geometry planes 0..2 (accumulated into
 *       grad_packed, caller zero-fills) and d/d sdf net (grads->w1..w3).  workspace: n*4 floats.
 * _tex: upstream g_features (n,3) of a 96->64->64->3 net in w->v1..v3 reading planes plane_base..plane_base+2 of
 *       each prompt: plane_base = 3 is the feature net on the texture planes (tt_query_points with TT_Q_TEX);
 *       plane_base = 0 with v1 = [U1 U1 U1] is a 32->64->64->3 net on the SUM of the geometry planes, i.e. the
 *       deformation head of tt_query_field (d/d U1 = the sum of the three 64x32 column blocks of grads->v1). */
int tt_points_bwd_geo(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                      int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h, int32_t plane_w,
                      float radius, float sdf_bias_radius, int32_t flags, const float* g_sdf, const float* g_sdf_grad,
                      float* workspace, float* grad_packed, const tt_mlp_grads* grads, void* stream);
int tt_points_bwd_tex(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                      int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h, int32_t plane_w,
                      float radius, int32_t plane_base, int32_t flags, const float* g_features, float* grad_packed,
                      const tt_mlp_grads* grads, void* stream);

/* Gradient of the per-point decode w.r.t. the QUERY POINTS (the reference keeps `points` in the autograd graph:
 * few_step...:283-286,329-335; caller generative_space_mesh_rasterize_renderer.py:307-331): for upstream g_sdf (n),
 * g_sdf_grad (n,3), g_features (n,3) (any may be null = 0)
 *   grad_points (n,3) = g_sdf d sdf/dx + d (g_sdf_grad . sdf_grad)/dx + (d features/dx)^T g_features,
 * i.e. aten grid_sampler_2d_backward's grad_grid for sdf / features and K1's `grad_grid` output
 * (gridsample_cuda.cu:196-208: the cross derivative of the bilinear interpolation) plus the sphere bias's Hessian for the
 * second-order term.  Overwrites grad_points.  flags: TT_Q_EXACT_F32 or 0.  (d/d planes and d/d weights of the same
 * upstream come from tt_points_bwd_geo / _tex.) */
int tt_points_bwd_x(const float* packed, const tt_mlp_weights* w, const float* points, int32_t n_batch,
                    int64_t n_points, int32_t n_prompts, int32_t views_per_prompt, int32_t plane_h, int32_t plane_w,
                    float radius, int32_t flags, const float* g_sdf, const float* g_sdf_grad, const float* g_features,
                    float* grad_points, void* stream);

/* Multiresolution hash encoding of 3-D points in [0,1]^3 (tcnn "HashGrid", Linear interpolation, fp32).
 * params: flat table, level-major, entry-major, feature-minor (tcnn's `params` layout), tt_hashgrid_n_params floats
 * (negative tt_status on a bad config); x (n,3); out / g_out (n, n_levels*n_features_per_level) row-major.
 * _bwd accumulates d/d params into grad_params (caller zero-fills); no gradient w.r.t. x. */
typedef struct {
    int32_t n_levels;             /* <= 16 */
    int32_t n_features_per_level; /* 1, 2, 4 or 8 */
    int32_t log2_hashmap_size;
    int32_t base_resolution;
    float per_level_scale;
} tt_hashgrid_cfg;
int64_t tt_hashgrid_n_params(const tt_hashgrid_cfg* cfg);
int tt_hashgrid_fwd(const float* x, int64_t n, const float* params, const tt_hashgrid_cfg* cfg, float* out,
                    void* stream);
int tt_hashgrid_bwd(const float* x, int64_t n, const float* g_out, const tt_hashgrid_cfg* cfg, float* grad_params,
                    void* stream);

/* PatchRenderer's per-key composite (threestudio/models/renderers/patch_renderer.py:74-88): out (B,H,W,C) = bilinear
 * upsample (F.interpolate, align_corners=False) of low (B,h,w,C) with patch (B,PS,PS,C) pasted at rows py.., columns
 * px..  _bwd: g_patch = the pasted region of g_out; g_low = the adjoint of the upsample over the pixels the patch did
 * not overwrite (null = global_detach: skipped).  Both overwrite their outputs. */
int tt_patch_composite_fwd(const float* low, const float* patch, float* out, int32_t B, int32_t h, int32_t w, int32_t H,
                           int32_t W, int32_t C, int32_t PS, int32_t py, int32_t px, void* stream);
int tt_patch_composite_bwd(const float* g_out, float* g_low, float* g_patch, int32_t B, int32_t h, int32_t w, int32_t H,
                           int32_t W, int32_t C, int32_t PS, int32_t py, int32_t px, void* stream);

/* Eikonal regulariser of the training loop on the renderer's per-sample `sdf_grad` output (n,3):
 *   loss[0] = mean((||sdf_grad||_2 - 1)^2)      (multiprompt_dual_renderer_multistep_generator.py:696-699)
 * one pass each way instead of ~10 torch kernels over the per-sample tensor.  _fwd overwrites loss[0] (device);
 * _bwd: g_sdf_grad (n,3) = g_loss[0] * 2 (||g|| - 1) / (n ||g||) * g (0 where ||g|| = 0), g_loss a DEVICE scalar. */
int tt_eikonal_fwd(const float* sdf_grad, int64_t n, float* loss, void* stream);
int tt_eikonal_bwd(const float* sdf_grad, const float* g_loss, int64_t n, float* g_sdf_grad, void* stream);

/* The renderer's per-ray composite (generative_space_sdf_volume_renderer.py:433-530) as one kernel each way:
 *   comp_rgb = rgb_fg + bg (1 - opacity)                                   bg: (3) with bg_stride 0, or (n,3) with 3
 *   disparity = clamp((far - (depth opacity + (1 - opacity) far)) / (far - near), 0, 1), far/near = d_cam +- sqrt(3)
 *   comp_normal = normalize(normal_acc)
 *   mode 1 ("camera"): n_cam = comp_normal @ inverse(c2w)[:3,:3]^T @ diag(-1,1,1);
 *                      normal_cam_vis = (n_cam+1)/2 opacity + (1-opacity) (0.5,0.5,1), _white with (1,1,1)
 *   mode 2 ("front") : the camera of view (v / view_group) * view_group, no flip, _white only;  mode 0 ("world"): neither.
 * _bwd: gradients w.r.t. opacity, depth, rgb_fg, normal_acc and, if g_bg is given, the PER-RAY background colour (n,3)
 * (a constant colour's gradient is its sum over rays), all overwritten, from the upstream gradients (null = 0). */
int tt_composite_fwd(const float* opacity, const float* depth, const float* rgb_fg, const float* normal_acc,
                     const float* bg, int32_t bg_stride, const float* camera_distances, const float* c2w, int64_t n_rays,
                     int32_t rays_per_view, int32_t mode, int32_t view_group, float* comp_rgb, float* disparity,
                     float* comp_normal, float* normal_cam_vis, float* normal_cam_vis_white, void* stream);
int tt_composite_bwd(const float* opacity, const float* depth, const float* rgb_fg, const float* normal_acc,
                     const float* bg, int32_t bg_stride, const float* camera_distances, const float* c2w, int64_t n_rays,
                     int32_t rays_per_view, int32_t mode, int32_t view_group, const float* g_comp_rgb,
                     const float* g_disparity, const float* g_comp_normal, const float* g_normal_cam_vis,
                     const float* g_normal_cam_vis_white, float* g_opacity, float* g_depth, float* g_rgb_fg,
                     float* g_normal_acc, float* g_bg, void* stream);

/* Operator-level drop-in for the reference's pybind op `gridsample_grad2.grad2_2d`
 * (gridsample_cuda.cpp:26-37; dispatch gridsample_cuda.cu:560-594): backward of aten::grid_sampler_2d_backward,
 * bilinear.  Contiguous tensors of one dtype: input / grad2_grad_input / grad_input (n,c,h,w); grid / grad2_grad_grid /
 * grad_grid (n,Ho,Wo,2); grad_output / grad_grad_output (n,c,Ho,Wo); n_points_per_batch = Ho*Wo.
 * padding_mode 0 = zeros, 1 = border (the reference passes it as a bool); align_corners 0 / 1; dtype TT_DTYPE_*
 * (half computes in fp32).  Reflection padding and the 3-D variant are TT_ERR_UNSUPPORTED / not exported (the
 * reference never calls them).  grad_input is zero-filled inside (like the reference).
 * tt_grid_sample_2d_grad2 is the fp32 entry point. */
#define TT_DTYPE_F32 0
#define TT_DTYPE_F16 1
#define TT_DTYPE_F64 2
int tt_grid_sample_2d_grad2_typed(int32_t dtype, const void* grad2_grad_input, const void* grad2_grad_grid,
                                  const void* grad_output, const void* input, const void* grid, int32_t n, int32_t c,
                                  int32_t h, int32_t w, int64_t n_points_per_batch, int32_t padding_mode,
                                  int32_t align_corners, void* grad_grad_output, void* grad_input, void* grad_grid,
                                  void* stream);
int tt_grid_sample_2d_grad2(const float* grad2_grad_input, const float* grad2_grad_grid, const float* grad_output,
                            const float* input, const float* grid, int32_t n, int32_t c, int32_t h, int32_t w,
                            int64_t n_points_per_batch, int32_t padding_mode, int32_t align_corners,
                            float* grad_grad_output, float* grad_input, float* grad_grid, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TT_ABI_H */
