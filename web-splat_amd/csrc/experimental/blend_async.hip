// blend_async.hip -- EXPERIMENTAL BUILD ONLY (make -C web-splat_amd experimental, -DWS_EXPERIMENTAL -> lib_exp/libwebsplat_hip.so).
// Measured-and-lost variant (profiles/r06/blend_async_staging_ab.txt; DESIGN_LOG.md R6), bit-identical to k_blend and tested against lib_exp;
// textually included by raster.hip inside namespace ws.  The product library never compiles this file.
// ---- k_blend2: the same tile, the same arithmetic, NO per-batch barriers (round 6; verdict r05 item 3) ---------------------
// k_blend's waves meet twice per staged batch: at the staging barrier and at the end-of-batch vote -- 26 % (hd1m) / 34 % (c3) of
// the kernel's span is waves waiting there for the slowest of sixteen.  Here the staged batch is double-buffered
// and the barriers are LDS arrival counters:
//   staged[b & 1]   stager waves that have written batch b (8 per batch; counts run on, never reset: batch b is there at
//                   8 * (b / 2 + 1)); a wave walks batch b once it is -- whatever the other waves are doing;
//   walked[b & 1]   waves that are done with batch b (16 per batch): the buffer of batch b may be overwritten with batch b + 2
//                   once it is complete;
//   dead            waves whose 64 pixels are all saturated (T only falls: counted once per wave): the tile ends when all
//                   sixteen are -- no vote.
// A stager wave stages batch b + 1 EARLY (before it walks batch b) if batch b - 1's buffer is already free -- true for the
// slowest wave, which is what keeps everybody supplied -- else LATE (behind its walk of batch b): a fast wave runs up to one
// batch ahead of the slowest instead of waiting for it twice per batch.  A tile with ONE batch (the median hd1m tile) never waits
// at all behind its staging: every wave stores its pixels when its own walk ends.  Per wave the compaction, the walk and the
// saturation test are k_blend's, statement for statement: the image is bit-identical (test_async_blend_is_bit_identical).
// One tile per workgroup at the 32x32 tile (4 x 4 quadrants); every other shape / MULTI / capture / timing stays with k_blend.
// LDS: 2 x 16.4 KB records + 2 x 1 KB masks + 33.8 KB lists = 68.7 KB: two workgroups per CU, as k_blend.
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* w) {  // wave-uniform read of an LDS counter other waves bump
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)w) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void lds_bump(uint32_t* w) {  // this wave's earlier LDS writes land first (LDS is in order per wave)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
#ifndef WS_B2_SLEEP
#define WS_B2_SLEEP 4
#endif
#ifndef WS_B2_FIRST_BARRIER
#define WS_B2_FIRST_BARRIER 0
#endif
template <int FORMAT>
__global__ __launch_bounds__(1024, 8) void k_blend2(const BlendParams p) {
    constexpr int QW = 4, QH = 4, NW = 16, STAGE = 512, SLOTS = STAGE + 1, LCAP = 512, TW = 32, TH = 32;
    constexpr uint32_t NSTAGERS = STAGE / 64;
    __shared__ float4 s_rec[2][2 * SLOTS];
    __shared__ __attribute__((aligned(16))) uint16_t s_m[2][STAGE];
    __shared__ __attribute__((aligned(16))) uint32_t s_list[NW][LCAP + 16];
    __shared__ uint32_t s_ctr[8];  // [0..1] staged per parity, [2..3] walked per parity, [4] dead waves

    if (blockIdx.x == 0 && threadIdx.x == 0 && p.sticky) {
        const uint32_t bits = p.counters->overflow;
        if (bits) fold_frame_errors(p, bits);
        post_frame_progress(p);
    }
    const BlendShape shape = blend_shape(QW, QH);
    const BlendBlock blk = blend_block_of(blockIdx.x, p.tiles_x, p.tiles_y, shape, 0u);
    const bool ordered = p.order != nullptr;
    if (!ordered && !blk.valid) return;  // block-uniform
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int qx = wave % QW, qy = wave / QW;
    const float lx = (float)(qx * 8 + (lane & 7)) + 0.5f;
    const float ly = (float)(qy * 8 + (lane >> 3)) + 0.5f;
    const uint32_t qbit = 1u << wave;
    const bool stager = wave < (int)NSTAGERS;  // waves 0..7 (decided on the SGPR copy of the wave index: every `if (stager)` is a scalar branch)
    uint2 range = make_uint2(0u, 0u);
    uint32_t code = 0xFFFFFFFFu;
    if (ordered) {
        const unsigned long long* op = reinterpret_cast<const unsigned long long*>(p.order + blockIdx.x);
        const unsigned long long o0 = __hip_atomic_load(op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long o1 = __hip_atomic_load(op + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        code = __builtin_amdgcn_readfirstlane((uint32_t)o0);
        range = make_uint2(__builtin_amdgcn_readfirstlane((uint32_t)(o0 >> 32)), __builtin_amdgcn_readfirstlane((uint32_t)o1));
    } else {
        const uint32_t slot = blk.w;
        const uint32_t tx = (blk.bx << shape.tbx_log2) + (slot & ((1u << shape.tbx_log2) - 1u));
        const uint32_t ty = (blk.by << shape.tby_log2) + (slot >> shape.tbx_log2);
        if (tx < p.tiles_x && ty < p.tiles_y) {
            code = tx | (ty << 16);
            range = p.tile_ranges[tile_list_index(p, tx, ty)];
            range.x = range.y ? 0xFFFFFFFFu - range.x : 0u;
        }
    }
    if (code == 0xFFFFFFFFu) return;  // block-uniform
    const uint32_t tx = code & 0xFFFFu, ty = code >> 16;
    const uint32_t len = range.y - range.x;
    const uint32_t nbatch = (len + (uint32_t)STAGE - 1u) / (uint32_t)STAGE;
    // batch k (0 = nearest) ends at hi_of(k) and holds min(STAGE, hi_of(k) - range.x) entries
    auto hi_of = [&](uint32_t k) -> uint32_t { return range.y - k * (uint32_t)STAGE; };
    // the first batch's dependent chain (entry index -> Splat record) leaves before anything else
    RawSplat raw = {{0u, 0u, 0u, 0u}, 0u};
    uint32_t idx_next = 0u;
    if (stager && nbatch) {
        raw = blend_fetch_raw<STAGE>(p, range, range.y, tid);
        idx_next = blend_entry_idx<STAGE>(p, range, nbatch > 1u ? hi_of(1u) : range.x, tid);
    }
    if (tid < 8) s_ctr[tid] = 0u;
    if (tid < 2) {  // the null record of both buffers (a' = 1e18: never inside the cut-off), pads the lists to multiples of four
        s_rec[tid][STAGE] = make_float4(0.0f, 0.0f, 1.0e9f, 0.0f);
        s_rec[tid][SLOTS + STAGE] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const uint32_t px = tx * TW + qx * 8 + (lane & 7);
    const uint32_t py = ty * TH + qy * 8 + (lane >> 3);
    const bool inside = px < p.width && py < p.height;
    float T = inside ? 1.0f : 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    const float W = (float)p.width, H = (float)p.height;
    const float tile_x0 = (float)(tx * TW), tile_y0 = (float)(ty * TH);
    uint32_t* my_list = s_list[wave];
    __syncthreads();  // the ONLY workgroup barrier: counters and null records are set up (the first gather is in flight across it)

    if (nbatch) {
        bool wave_dead = __ballot(T >= T_MIN) == 0ull;  // (a quadrant outside the image)
        if (wave_dead) lds_bump(&s_ctr[4]);
        uint32_t sb = 0u;  // stager waves: the next batch this wave stages
        bool walked_since_stage = true;  // (the gather of batch sb left when batch sb - 1 was staged: it has landed once a walk lies between)
        // stage batch sb into buffer sb & 1 (this thread's slot) and send the loads of the batches behind it on their way
        auto stage_batch = [&]() {
            const uint32_t hi = hi_of(sb);
            const uint32_t nb = (hi - range.x) < (uint32_t)STAGE ? (hi - range.x) : (uint32_t)STAGE;
            float4* rec = s_rec[sb & 1u];
            uint32_t mask = 0u;
            if ((uint32_t)tid < nb) {
                const stage::Staged st = stage::decode<QW, QH>(raw.a.x, raw.a.y, raw.a.z, raw.a.w, raw.w4, W, H, tile_x0, tile_y0, CUT_A2);
                mask = st.mask;
                rec[tid] = make_float4(st.i00, st.i01, st.c0, st.i10);
                rec[SLOTS + tid] = make_float4(st.i11, st.c1, __uint_as_float(raw.a.w), __uint_as_float(raw.w4));
            }
            s_m[sb & 1u][((uint32_t)tid & 63u) * (LCAP / 64) + ((uint32_t)tid >> 6)] = (uint16_t)mask;
            // (unconditional, addresses clamped into the tile's range -- see k_blend: a conditional prefetch waits for its own data)
            raw = blend_gather(p, idx_next);
            const uint32_t h2 = sb + 2u < nbatch ? hi_of(sb + 2u) : range.x;
            idx_next = blend_entry_idx<STAGE>(p, range, h2, tid);
            lds_bump(&s_ctr[sb & 1u]);
            ++sb;
            walked_since_stage = false;
        };
        // is the buffer of batch k free, i.e. has every wave left batch k - 2 ?
        auto buffer_free = [&](uint32_t k) -> bool {
            if (k < 2u) return true;
            return lds_peek(&s_ctr[2u + (k & 1u)]) >= (uint32_t)NW * ((k - 2u) / 2u + 1u);
        };
        // One loop, three wave-uniform actions: STAGE the next batch if it is this wave's turn and the buffer is free (early: before
        // the walk of the batch in front of it; late: behind it -- the same code, whichever comes first), WALK batch b if it is
        // there, else WAIT (a short sleep; the tile may end while we do).
        uint32_t b = 0u, polls = 0u;
#if WS_B2_FIRST_BARRIER
        // the first batch the classic way: staged behind ONE workgroup barrier (every wave sleeps in hardware instead of polling
        // through the start-up chain index -> record -> decode, which a tile with a single batch -- the median tile -- is mostly made of)
        if (stager) stage_batch();
        __syncthreads();
#endif
        for (;;) {
            if (wave_dead && lds_peek(&s_ctr[4]) >= (uint32_t)NW) break;  // every wave is saturated: nothing left to stage or walk
            // (early staging only with a walk behind the previous staging: else the decode would sit waiting for the gather that has
            //  just left -- the latency the walk is there to hide; batch b itself, sb == b, is never early)
            if (stager && sb < nbatch && (sb == b || (sb == b + 1u && walked_since_stage)) && buffer_free(sb)) {
                stage_batch();
                continue;
            }
            if (lds_peek(&s_ctr[b & 1u]) < NSTAGERS * (b / 2u + 1u)) {  // batch b is not there yet
                if (lds_peek(&s_ctr[4]) >= (uint32_t)NW) break;         // ... and never will be: every wave is saturated
                // (bounded, like every spin of this library: ~1 s without progress sets error bit 4 instead of hanging the queue)
                if (++polls > (1u << 24)) {
                    if (lane == 0 && p.sticky) atomicOr(p.sticky, 16u);
                    break;
                }
                __builtin_amdgcn_s_sleep(WS_B2_SLEEP);
                continue;
            }
            polls = 0u;
            if (!wave_dead) {
                // (the buffer's byte offset is folded into the list entries: the walk's LDS addresses stay list entry + constant)
                const uint32_t bufoff = (b & 1u) * (uint32_t)(2 * SLOTS * 16);
                const float4* rec = s_rec[0];
                // wave-private compaction: records whose kept ellipse reaches this quadrant, near -> far (k_blend's, one sub-round)
                const uint2* mp = reinterpret_cast<const uint2*>(s_m[b & 1u] + (uint32_t)lane * (LCAP / 64));
                uint32_t n = 0;
                uint32_t slot16 = (uint32_t)lane * 16u + bufoff;
                asm volatile("" : "+v"(slot16));
#pragma unroll
                for (int h = 0; h < LCAP / 256; ++h) {
                    const uint2 mm = mp[h];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = h * 4 + q;
                        const uint32_t word = (q & 2) ? mm.y : mm.x;
                        const bool t = (word & (qbit << ((q & 1) * 16))) != 0u;  // (slots past the batch's end hold mask 0)
                        const unsigned long long bal = __ballot(t);
                        const uint32_t pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        if (t) my_list[pos] = slot16 + (uint32_t)r * 1024u;
                        n += (uint32_t)__popcll(bal);
                    }
                }
                if (n > 0u) {
                    if (lane < 4 && ((n + (uint32_t)lane) >> 2) == (n >> 2) && (n & 3u)) my_list[n + lane] = (uint32_t)STAGE * 16u + bufoff;  // pad to x4
                    const uint32_t n4 = (n + 3u) >> 2;
                    const uint4* lp = reinterpret_cast<const uint4*>(my_list);
                    uint4 o = lp[0];
                    uint4 on = lp[n4 > 1u ? 1u : 0u];
                    BlendRec cur = blend_load_rec<SLOTS>(rec, o.x);
                    for (uint32_t g = 0; g < n4; ++g) {
                        const BlendRec r1 = blend_load_rec<SLOTS>(rec, o.y);
                        blend_composite(cur, lx, ly, T, cr, cg, cb);
                        const BlendRec r2 = blend_load_rec<SLOTS>(rec, o.z);
                        blend_composite(r1, lx, ly, T, cr, cg, cb);
                        const BlendRec r3 = blend_load_rec<SLOTS>(rec, o.w);
                        blend_composite(r2, lx, ly, T, cr, cg, cb);
                        cur = blend_load_rec<SLOTS>(rec, on.x);
                        blend_composite(r3, lx, ly, T, cr, cg, cb);
                        if (__ballot(T >= T_MIN) == 0ull) break;
                        o = on;
                        on = lp[g + 2u < n4 ? g + 2u : n4 - 1u];
                    }
                }
                if (__ballot(T >= T_MIN) == 0ull) {
                    wave_dead = true;
                    lds_bump(&s_ctr[4]);
                }
            }
            if (b + 1u == nbatch) break;  // the last batch: nobody stages behind it, nobody needs to know we left it
            lds_bump(&s_ctr[2u + (b & 1u)]);
            ++b;
            walked_since_stage = true;
        }
    }
    {
        const uint32_t sx = tx * TW + (uint32_t)lx, sy = ty * TH + (uint32_t)ly;
        if (sx < p.width && sy < p.height)
            store_pixel<FORMAT>(p, sx, sy, cr + p.background[0] * T, cg + p.background[1] * T, cb + p.background[2] * T,
                                (1.0f - T) + p.background[3] * T);
    }
}

