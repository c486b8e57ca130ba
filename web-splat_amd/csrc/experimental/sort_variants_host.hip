// sort_variants_host.hip -- EXPERIMENTAL BUILD ONLY (make -C web-splat_amd experimental, -DWS_EXPERIMENTAL -> lib_exp/libwebsplat_hip.so).
// Measured-and-lost variant(s), kept bit-exact and tested against lib_exp (DESIGN_LOG.md); textually included by sort.hip at the
// place the code used to stand, inside namespace ws.  The product library (lib/libwebsplat_hip.so) never compiles this file.
size_t fat_sort_status_words() { return (size_t)4 * FAT_MAX_GRID * RADIX; }

uint32_t fat_sort_grid(uint32_t n, int num_cus, int grid_request) {
    uint32_t most = num_cus > 0 && (uint32_t)num_cus < FAT_MAX_GRID ? (uint32_t)num_cus : FAT_MAX_GRID;
    most &= ~7u;
    if (most == 0u) return 0u;
    uint32_t g = ((n + FAT_CHUNK_MAX - 1u) / FAT_CHUNK_MAX + 7u) & ~7u;  // fewest chunks that hold n
    if (g > most) return 0u;
    // at least half the CUs, so that a frame's ranking is spread over 512 SIMDs; two round trips of predecessor sums
    uint32_t want = grid_request > 0 ? ((uint32_t)grid_request + 7u) & ~7u : 128u;
    if (want > most) want = most;
    return g > want ? g : want;
}

int launch_depth_sort_fat(const FatSortScratch& sc, uint32_t* keys, uint32_t* vals, uint32_t* aux, const uint32_t* d_count,
                          uint32_t n, bool implicit_iota, bool coop, uint32_t epoch, int num_cus, hipStream_t stream,
                          KernelMarks* km) {
    if (n == 0) return WS_OK;
    if (n > sc.cap) return fail(WS_ERR_INVALID, "depth sort: n exceeds the scratch capacity");
    if (aux && !sc.aux_alt) return fail(WS_ERR_INVALID, "depth sort: companion values without a companion scratch buffer");
    if ((reinterpret_cast<uintptr_t>(keys) & 15u) != 0) return fail(WS_ERR_INVALID, "sort: keys must be 16-byte aligned");
    const uint32_t grid = fat_sort_grid(n, num_cus, sc.grid_request);
    if (grid == 0u) return fail(WS_ERR_INVALID, "depth sort: input beyond the fat-tile form's capacity");
    FatSortArgs a;
    a.keys[0] = keys;  a.keys[1] = sc.keys_alt;
    a.vals[0] = vals;  a.vals[1] = sc.vals_alt;
    a.aux[0] = aux;    a.aux[1] = sc.aux_alt;
    a.d_count = d_count;
    a.n = n;
    a.hist = sc.hist;
    a.status = sc.status;
    a.tickets = sc.tickets;
    a.barrier = sc.barrier;
    a.error = sc.error;
    a.epoch = epoch;
    a.d_epoch = sc.d_epoch;
    a.iota = implicit_iota ? 1 : 0;
    const bool small = (uint64_t)n <= (uint64_t)grid * FAT_THREADS * 4u;  // at most four pairs per thread: the 70-KB build
#define WS_FAT(KPT_, CARRY_, COOP_, PB_, PE_)                                                                            \
    hipLaunchKernelGGL((k_dsort_fat<KPT_, CARRY_, COOP_>), dim3(grid), dim3(FAT_THREADS), 0, stream, a, PB_, PE_)
#define WS_FAT_ANY(COOP_, PB_, PE_)                                                                                      \
    do {                                                                                                                 \
        if (aux) { if (small) WS_FAT(4, true, COOP_, PB_, PE_); else WS_FAT(8, true, COOP_, PB_, PE_); }                  \
        else { if (small) WS_FAT(4, false, COOP_, PB_, PE_); else WS_FAT(8, false, COOP_, PB_, PE_); }                    \
    } while (0)
    if (coop) {
        if (!sc.barrier) return fail(WS_ERR_INVALID, "depth sort: the single-launch form needs barrier state");
        WS_FAT_ANY(true, 0, 4);
        km_mark(km, "depth:k_dsort_fat_coop");
    } else {
        uint32_t hist_blocks = (n / 4 + SORT_THREADS * 8 - 1) / (SORT_THREADS * 8);
        if (hist_blocks < 1) hist_blocks = 1;
        if (hist_blocks > 1024) hist_blocks = 1024;
        hipLaunchKernelGGL(k_sort_hist, dim3(hist_blocks), dim3(SORT_THREADS), 0, stream, keys, d_count, n, 0, 4, sc.hist);
        km_mark(km, "depth:k_sort_hist");
        for (int p = 0; p < 4; ++p) {
            WS_FAT_ANY(false, p, p + 1);
            km_mark(km, "depth:k_dsort_fat");
        }
    }
#undef WS_FAT_ANY
#undef WS_FAT
    WS_HIP(hipGetLastError());
    return WS_OK;
}

int launch_tile_sort_wide(const SortScratch& sc, const uint32_t* keys16, const uint32_t* vals, const uint32_t* d_count,
                          uint32_t n, int bits, hipStream_t stream, KernelMarks* km, uint2* ranges, uint32_t nranges) {
    if (n == 0) return WS_OK;
    if (bits < 7) bits = 7;
    const uint32_t bins = 1u << bits;
    if (bits > TILE_SORT_WIDE_MAX_BITS || sc.wide_bins < bins || !sc.wide_hist || n > sc.cap || nranges > bins)
        return fail(WS_ERR_INVALID, "tile sort: the single-pass form needs at most 2048 ids and scratch sized for them");
    if (sort_tile_size(n) != (uint32_t)SORT_TILE) return fail(WS_ERR_INVALID, "tile sort: the single-pass form uses the large sort tile");
    const uint32_t tiles = sort_grid((n + SORT_TILE - 1) / SORT_TILE);
    hipLaunchKernelGGL(k_tile_col_scan_wide, dim3(bins / 16u), dim3(WIDE_SCAN_THREADS), 0, stream, d_count, n,
                       (uint32_t)SORT_TILE, sc.tile_sums, bins, sc.wide_hist);
    km_mark(km, "tiles:k_sort_col_scan");
    const uint16_t* k16 = reinterpret_cast<const uint16_t*>(keys16);
#define WS_WIDE(BITS_)                                                                                                 \
    hipLaunchKernelGGL((k_tile_scatter_wide<SORT_KPT, BITS_>), dim3(tiles), dim3(SORT_THREADS), 0, stream, k16, vals,  \
                       sc.vals_alt, d_count, n, sc.wide_hist, sc.tile_sums, ranges, nranges)
    switch (bits) {
        case 7: WS_WIDE(7); break;
        case 8: WS_WIDE(8); break;
        case 9: WS_WIDE(9); break;
        case 10: WS_WIDE(10); break;
        default: WS_WIDE(11); break;
    }
#undef WS_WIDE
    km_mark(km, "tiles:k_sort_scatter");
    WS_HIP(hipGetLastError());
    return WS_OK;
}
