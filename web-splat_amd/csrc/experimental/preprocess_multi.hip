// preprocess_multi.hip -- EXPERIMENTAL BUILD ONLY (make -C web-splat_amd experimental, -DWS_EXPERIMENTAL -> lib_exp/libwebsplat_hip.so).
// Measured-and-lost variant(s), kept bit-exact and tested against lib_exp (DESIGN_LOG.md); textually included by preprocess.hip at the
// place the code used to stand, inside namespace ws.  The product library (lib/libwebsplat_hip.so) never compiles this file.
// ---- K1 for several views of one batch in ONE launch -----------------------------------------------------------------
// The views of a batch are independent but read the SAME scene: one launch per view fetches the 124 B of every Gaussian
// once per view (148 MB per frame on the 1 M scene: the largest single consumer of HBM bytes, and what K1 is bound by with
// frames in flight).  Here a workgroup loads its 1024 Gaussians once and runs front end, ordered compaction and back end
// for up to K1_MAX_VIEWS cameras, each into its own renderer's scratch (Splat records, keys, footprint words, counters,
// look-back words).  Per view the arithmetic, the visible set and the store order are those of k_preprocess -- the same
// functions on the same inputs -- so every later stage, and the image, is unchanged.
//   * one ticket (view 0's dispenser) gives the workgroup its block for all views: the start order a look-back needs is
//     shared;
//   * wave v runs the look-back of view v (up to four views, four waves), all at once;
//   * covariance and SH planes are fetched when ANY view keeps the Gaussian, for the highest SH degree any view asks for.
template <bool COMPRESSED, int FPMODE>
__global__ __launch_bounds__(K1_THREADS, WS_K1_MINWAVES) void k_preprocess_multi(const K1MultiArgs a) {
    static_assert(K1_MAX_VIEWS <= K1_THREADS / 64, "one look-back wave per view");
    static_assert(K1_MAX_VIEWS * K1_ITEMS <= 32, "visibility bits of a thread fit one word");
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_cnt[K1_MAX_VIEWS][K1_ITEMS][K1_THREADS / 64];  // visible per (view, item, wave); then exclusive offsets
    __shared__ uint32_t s_tot[K1_MAX_VIEWS];
    __shared__ uint32_t s_base[K1_MAX_VIEWS];
    __shared__ uint32_t s_t32[K1_MAX_VIEWS], s_t64[K1_MAX_VIEWS];  // footprint totals per view (bin_shift_decide)
    const uint32_t nv = a.nv;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (tid == 0) s_bid = atomicAdd(&a.b[0].counters->k1_ticket, 1u);  // START order, shared by the views (k_preprocess)
    __syncthreads();
    const uint32_t bid = s_bid;
    const uint32_t n = a.p[0].num_points;
    const uint32_t block_base = bid * (K1_THREADS * K1_ITEMS);

    // ---- front end: positions once, cull per view ------------------------------------------------------------
    Front fr[K1_ITEMS];
#pragma unroll
    for (int it = 0; it < K1_ITEMS; ++it) {
        const uint32_t idx = block_base + it * K1_THREADS + tid;
        k1_load_front<COMPRESSED>(a.b[0], idx < n ? idx : n - 1u, &fr[it]);
    }
    uint32_t vis_bits = 0u;              // bit (v * K1_ITEMS + it)
    uint32_t rank_pack[K1_MAX_VIEWS];    // 8 bits per item: rank of this lane among its wave's survivors
#pragma unroll
    for (int v = 0; v < K1_MAX_VIEWS; ++v) {
        rank_pack[v] = 0u;
        if ((uint32_t)v < nv) {  // uniform
#pragma unroll
            for (int it = 0; it < K1_ITEMS; ++it) {
                const uint32_t idx = block_base + it * K1_THREADS + tid;
                float camspace[4], pos2d[4];
                const bool vv = (idx < n) && k1_project<COMPRESSED>(a.p[v], fr[it].xyz, camspace, pos2d);
                const unsigned long long vmask = __ballot(vv);
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(vmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vmask, 0u));
                rank_pack[v] |= rk << (8 * it);
                if (vv) vis_bits |= 1u << (v * K1_ITEMS + it);
                if (lane == 0) s_cnt[v][it][wave] = (uint32_t)__popcll(vmask);
            }
        }
    }
    __syncthreads();
    // counts -> exclusive offsets inside the block, (item, wave) order = Gaussian index order; one thread per view
    if ((uint32_t)tid < nv) {
        uint32_t run = 0u;
#pragma unroll
        for (int it = 0; it < K1_ITEMS; ++it)
#pragma unroll
            for (int w = 0; w < K1_THREADS / 64; ++w) {
                const uint32_t c = s_cnt[tid][it][w];
                s_cnt[tid][it][w] = run;
                run += c;
            }
        s_tot[tid] = run;
    }
    __syncthreads();
    // ---- ordered compaction: wave v publishes and looks back for view v ---------------------------------------
    if ((uint32_t)wave < nv) {
        const int v = wave;
        const uint32_t block_cnt = s_tot[v];
        const uint32_t epoch = a.p[v].epoch;
        if (lane == 0) {
            lb::st(a.b[v].block_status + bid, lb::pack(epoch, bid == 0 ? lb::FLAG_INCL : lb::FLAG_AGG, block_cnt));
            if (bid == 0) {  // for the later kernels of that view's frame
                a.b[v].counters->epoch = epoch;
                a.b[v].counters->bin_request = FPMODE == FP_RECT_PACKED ? a.p[v].bin_request : (uint32_t)BIN_NEVER;
            }
            s_t32[v] = 0u;
            s_t64[v] = 0u;
        }
        const uint32_t excl = lb::wave_lookback(a.b[v].block_status, bid, epoch, lane, &a.b[v].counters->overflow, 2u);
        if (lane == 0) {
            s_base[v] = excl;
            if (bid != 0) lb::st(a.b[v].block_status + bid, lb::pack(epoch, lb::FLAG_INCL, excl + block_cnt));
            if (bid == gridDim.x - 1) a.b[v].counters->num_visible = excl + block_cnt;
        }
    }
    __syncthreads();

    // ---- back end: planes once per Gaussian that any view keeps, maths and stores per view ---------------------
    uint32_t deg_max = 0u;
#pragma unroll
    for (int v = 0; v < K1_MAX_VIEWS; ++v)
        if ((uint32_t)v < nv) deg_max = max(deg_max, a.p[v].rs.max_sh_deg);
    constexpr uint32_t ANY = (1u << 0) | (1u << K1_ITEMS) | (1u << (2 * K1_ITEMS)) | (1u << (3 * K1_ITEMS));
    // The loop over the views is NOT unrolled around the back-end maths: four inlined copies per item made 18 k
    // instructions (146 KB of code against a 64-KB instruction cache) and the launch slower than four single-view ones.
    // `v` is uniform, so the per-view values come from kernel arguments / LDS by index and the rank word by a select.
    auto rank_of = [&](uint32_t v, int it) -> uint32_t {
        const uint32_t w = v == 0u ? rank_pack[0] : (v == 1u ? rank_pack[1] : (v == 2u ? rank_pack[2] : rank_pack[3]));
        return (w >> (8 * it)) & 0xFFu;
    };
    auto store = [&](uint32_t v, int it, const SplatOut& so) {
        const uint32_t slot = s_base[v] + s_cnt[v][it][wave] + rank_of(v, it);
        uint32_t* sp = reinterpret_cast<uint32_t*>(a.b[v].splats + (size_t)slot * SPLAT_STRIDE);
        sp[0] = so.w[0];
        sp[1] = so.w[1];
        sp[2] = so.w[2];
        sp[3] = so.w[3];
        sp[4] = so.w[4];
        a.b[v].keys[slot] = so.key;
        a.b[v].footprints[slot] = so.fp;
        if (FPMODE == FP_RECT_PACKED) {  // (LDS atomics: a few per thread and view)
            const uint32_t t32 = rect_tiles(so.fp);
            if (t32) {
                atomicAdd(&s_t32[v], t32);
                atomicAdd(&s_t64[v], rect_tiles64(so.fp));
            }
        }
    };
    if (!COMPRESSED) {
        const uint32_t safe_idx = block_base < n ? block_base : 0u;
        K1Params pl = a.p[0];   // (only num_points and the SH degree are read by the plane loads)
        pl.rs.max_sh_deg = deg_max;
#pragma unroll
        for (int grp = 0; grp < K1_ITEMS; grp += K1_BACK_GROUP) {
            RawBack rb[K1_BACK_GROUP];
#pragma unroll
            for (int u = 0; u < K1_BACK_GROUP; ++u) {
                const int it = grp + u;
                const bool any = (vis_bits & (ANY << it)) != 0u;
                k1_back_load(pl, a.b[0], any ? block_base + it * K1_THREADS + tid : safe_idx, &rb[u]);
            }
#pragma unroll
            for (int u = 0; u < K1_BACK_GROUP; ++u) {
                const int it = grp + u;
#pragma unroll 1
                for (uint32_t v = 0; v < nv; ++v) {
                    if (vis_bits & (1u << (v * K1_ITEMS + it))) {
                        SplatOut so;
                        k1_back_math<FPMODE>(a.p[v], fr[it], rb[u], &so);
                        store(v, it, so);
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < K1_ITEMS; ++it)
#pragma unroll 1
            for (uint32_t v = 0; v < nv; ++v)
                if (vis_bits & (1u << (v * K1_ITEMS + it))) {
                    SplatOut so;
                    k1_back_compressed<FPMODE>(a.p[v], a.b[v], fr[it], &so);
                    store(v, it, so);
                }
    }
    if (FPMODE == FP_RECT_PACKED) {
        __syncthreads();
        if ((uint32_t)tid < nv && s_t32[tid]) {
            uint32_t* ts = a.b[tid].counters->tile_sums + (blockIdx.x & (TILE_SUM_SLOTS - 1)) * TILE_SUM_STRIDE;
            atomicAdd(ts, s_t32[tid]);
            atomicAdd(ts + 1, s_t64[tid]);
        }
    }
}
