/* websplat_env.h -- NOT part of the library: a header-only helper for harnesses (the tools/ mains, C test drivers) that want the
 * historical WS_* environment switches.  It translates the process environment into a ws_context_config; the library itself
 * reads no environment variable (websplat.h, ws_context_config).  The Python harness has the same table in
 * web-splat_amd/websplat/api.py (config_from_env). */
#ifndef WEBSPLAT_ENV_H
#define WEBSPLAT_ENV_H
#include <stdlib.h>
#include <string.h>

#include "websplat.h"

static inline int ws_env_int_(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
static inline int ws_env_is_(const char* name, const char* value) {
    const char* v = getenv(name);
    return v && strcmp(v, value) == 0;
}
static inline void ws_context_config_from_env(ws_context_config* c) {
    ws_context_config_init(c);
    c->use_graph = ws_env_int_("WS_GRAPH", c->use_graph);
    c->depth_skip_top = ws_env_int_("WS_DEPTH_SKIP_TOP", c->depth_skip_top);
    c->blend_order = ws_env_int_("WS_BLEND_ORDER", c->blend_order);
    c->blend_split = ws_env_int_("WS_BLEND_SPLIT", c->blend_split);
    if (getenv("WS_BIN_SHIFT")) c->bin_request = ws_env_is_("WS_BIN_SHIFT", "1") ? 2 : (ws_env_is_("WS_BIN_SHIFT", "0") ? 0 : 1);
    c->batch_threads = ws_env_int_("WS_BATCH_THREADS", c->batch_threads);
    c->batch_queue_depth = ws_env_int_("WS_BATCH_QUEUE_DEPTH", c->batch_queue_depth);
    c->blend_tpw_log2 = ws_env_int_("WS_BLEND_TPW_LOG2", c->blend_tpw_log2);
    c->blend_lds_pad_kb = ws_env_int_("WS_BLEND_LDS_PAD_KB", c->blend_lds_pad_kb);
    if (ws_env_is_("WS_TILE_SHAPE", "2x2")) c->tile_qw = c->tile_qh = 2;
    else if (ws_env_is_("WS_TILE_SHAPE", "4x2")) { c->tile_qw = 4; c->tile_qh = 2; }
    else if (getenv("WS_TILE_SHAPE") && !ws_env_is_("WS_TILE_SHAPE", "4x4")) c->tile_qw = c->tile_qh = -1; /* refused by the library */
    c->debug_cut = ws_env_int_("WS_DEBUG_CUT", c->debug_cut);
    c->capture = ws_env_int_("WS_CAPTURE", c->capture);
    c->render_views_fast_blend = ws_env_is_("WS_RENDER_VIEWS_BLEND", "fast");
    c->ply_decode_host = ws_env_is_("WS_PLY_DECODE", "host");
    c->depth_digit_bits = ws_env_int_("WS_DEPTH_DIGIT_BITS", c->depth_digit_bits);
    c->depth_tile_kpt = ws_env_int_("WS_DEPTH_TILE_KPT", c->depth_tile_kpt);
    c->blend_async = ws_env_int_("WS_BLEND_ASYNC", c->blend_async);
    if (ws_env_is_("WS_DEPTH_SORT", "onesweep")) c->exp_depth_sort = 1;
    else if (ws_env_is_("WS_DEPTH_SORT", "coop")) c->exp_depth_sort = 2;
    c->exp_dsort_fat_grid = ws_env_int_("WS_DSORT_FAT_GRID", c->exp_dsort_fat_grid);
    c->exp_blend_variant = ws_env_int_("WS_BLEND_VARIANT", c->exp_blend_variant);
    c->exp_blend_dma = ws_env_int_("WS_BLEND_DMA", c->exp_blend_dma) ? 1 : 0;
    c->exp_batch_k1 = ws_env_int_("WS_BATCH_K1", c->exp_batch_k1);
    c->exp_footprint_ellipse = ws_env_is_("WS_FOOTPRINT", "ellipse");
    c->exp_tile_sort_wide = ws_env_is_("WS_TILE_SORT", "wide");
}
#endif
