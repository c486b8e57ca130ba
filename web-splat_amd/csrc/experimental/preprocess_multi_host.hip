// preprocess_multi_host.hip -- EXPERIMENTAL BUILD ONLY (make -C web-splat_amd experimental, -DWS_EXPERIMENTAL -> lib_exp/libwebsplat_hip.so).
// Measured-and-lost variant(s), kept bit-exact and tested against lib_exp (DESIGN_LOG.md); textually included by preprocess.hip at the
// place the code used to stand, inside namespace ws.  The product library (lib/libwebsplat_hip.so) never compiles this file.
int launch_preprocess_multi(const K1Params* p, const K1Buffers* b, uint32_t nv, bool compressed, int footprint_mode,
                            hipStream_t stream) {
    if (nv == 0 || nv > (uint32_t)K1_MAX_VIEWS) return fail(WS_ERR_INVALID, "preprocess: 1..4 views per launch");
    const uint32_t blocks = preprocess_blocks(p[0].num_points);
    if (blocks == 0) return WS_OK;
    K1MultiArgs a;
    for (uint32_t v = 0; v < (uint32_t)K1_MAX_VIEWS; ++v) {
        a.p[v] = p[v < nv ? v : 0];
        a.b[v] = b[v < nv ? v : 0];
    }
    a.nv = nv;
#define WS_K1M(C, M) hipLaunchKernelGGL((k_preprocess_multi<C, M>), dim3(blocks), dim3(K1_THREADS), 0, stream, a)
    if (compressed) {
        if (footprint_mode == FP_ELLIPSE) WS_K1M(true, FP_ELLIPSE);
        else if (footprint_mode == FP_RECT_COUNT) WS_K1M(true, FP_RECT_COUNT);
        else WS_K1M(true, FP_RECT_PACKED);
    } else {
        if (footprint_mode == FP_ELLIPSE) WS_K1M(false, FP_ELLIPSE);
        else if (footprint_mode == FP_RECT_COUNT) WS_K1M(false, FP_RECT_COUNT);
        else WS_K1M(false, FP_RECT_PACKED);
    }
#undef WS_K1M
    WS_HIP(hipGetLastError());
    return WS_OK;
}
