// tt_host.h -- host-side helpers shared by the translation units of libtt_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tt_abi.h"

int tt_check_launch();
int tt_num_cus();
int tt_validate_cfg(const tt_render_cfg* cfg);

struct TileGeom;
// fills the tile geometry / chunking for a render config; returns the number of work items.
// default_order: 0 = chunk-major within an XCD (forward: best L2 reuse), 1 = block-major (backward: neighbouring
// pixel blocks at the same depth would hammer the same texels with atomics at the same time).
long long tt_make_geom(const tt_render_cfg* cfg, long long wave_slots, TileGeom* g, int default_order);
