// tt_host.h -- host-side helpers shared by the translation units of libtt_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tt_abi.h"

int tt_check_launch();
int tt_num_cus();
int tt_validate_cfg(const tt_render_cfg* cfg);
bool tt_planes_too_large(long long n_prompts, int plane_h, int plane_w);  // packed planes >= 4 GB: unsupported

// precision mode of a launch from its flag bits (tt_abi.h): 0 = two-piece split (fast), 1 = fp32 MFMA, 2 = three-piece
// split (default) -- the PREC_* values of tt_mfma16.h.  More than one precision bit is rejected by tt_validate_cfg /
// tt_validate_qflags.
static inline int tt_prec_of_r(int flags) {
    return (flags & TT_R_EXACT_F32) ? 1 : ((flags & TT_R_SPLIT2) ? 0 : 2);
}
static inline int tt_prec_of_q(int flags) {
    return (flags & TT_Q_EXACT_F32) ? 1 : ((flags & TT_Q_SPLIT2) ? 0 : 2);
}
static inline bool tt_qflags_ok(int flags) {
    const int pbits = flags & (TT_Q_EXACT_F32 | TT_Q_SPLIT2 | TT_Q_SPLIT3);
    return (pbits & (pbits - 1)) == 0;
}

struct TileGeom;
// fills the tile geometry / chunking for a render config; returns the number of work items.
// default_order: order of an XCD's item queue, 0 = chunk-major, 1 = block-major (measured slightly faster in all
// three kernels once the queue is dynamic: concurrent waves of an XCD then walk ONE pixel block's depth chunks and
// its neighbours rather than one depth slab of the whole image share).
long long tt_make_geom(const tt_render_cfg* cfg, long long wave_slots, TileGeom* g, int default_order,
                       int steps_per_item = 6, int min_items_per_slot = 8);

// Work-queue counters for one kernel launch: 8 int32 heads (one per XCD) in a library-owned device scratch, zeroed on
// `stream` by a one-wave kernel enqueued here (so the caller must launch the kernel on the same stream, next).  A
// ring of slots per (device, stream) for eager launches, a round-robin pool for launches recorded under stream capture
// (see tt_host.cpp).  Returns nullptr on a HIP error.  (The only state the library keeps: a 328 KB allocation per
// device, never freed; its first use must not happen inside a stream capture.)
int* tt_queue_counters(hipStream_t stream);
// layout of a slot (ints): [0..8] queue heads; [TT_SLOT_BOUNDS + k] = bit pattern of a non-negative float, raised with
// atomicMax by the reduction kernels launched in front of a backward kernel (tt_backward.hip: magnitude bounds of the
// operands of the split-fp16 weight-gradient outer products); everything is zeroed together with the queue heads.
#define TT_SLOT_INTS 32
#define TT_SLOT_BOUNDS 16
#define TT_BOUND_PLANES 0   /* max |texel| of the three planes the kernel reads */
#define TT_BOUND_UP0 1      /* geometry: max |d/d sdf|;            texture: max |g_rgb| */
#define TT_BOUND_UP1 2      /* geometry: max |d/d sdf_grad| comp.; texture: max |g_features| */
#define TT_SLOT_EIKONAL 24  /* tt_eikonal_fwd: 4 ints (8-byte aligned): fixed-point sum, overflow float, arrival counter */
