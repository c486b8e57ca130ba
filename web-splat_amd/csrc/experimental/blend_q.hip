// blend_q.hip -- EXPERIMENTAL BUILD ONLY (make -C web-splat_amd experimental, -DWS_EXPERIMENTAL -> lib_exp/libwebsplat_hip.so).
// Measured-and-lost variant(s), kept bit-exact and tested against lib_exp (DESIGN_LOG.md); textually included by raster.hip at the
// place the code used to stand, inside namespace ws.  The product library (lib/libwebsplat_hip.so) never compiles this file.
template <int FORMAT>
__global__ __launch_bounds__(64) void k_blend_q(const BlendParams p) {
    // blockIdx -> (tile, quadrant): workgroup b runs on XCD b % 8 (observed; used for locality only)
    const uint32_t b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0 && p.sticky) {  // as in k_blend
        const uint32_t bits = p.counters->overflow;
        if (bits) fold_frame_errors(p, bits);
        post_frame_progress(p);
    }
    const uint32_t xcd = b & 7u, j = b >> 3;
    const uint32_t nq = p.qw * p.qh;
    const uint32_t q = j % nq;
    const uint32_t tile = (j / nq) * 8u + xcd;
    const uint32_t ntiles = p.tiles_x * p.tiles_y;
    if (tile >= ntiles) return;
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int lane = threadIdx.x;
    const uint32_t qx0 = tx * p.qw * 8u + (q % p.qw) * 8u, qy0 = ty * p.qh * 8u + (q / p.qw) * 8u;
    const uint32_t px = qx0 + (lane & 7);
    const uint32_t py = qy0 + (lane >> 3);
    const bool inside = px < p.width && py < p.height;
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float qx_lo = (float)qx0 + 0.5f;
    const float qy_lo = (float)qy0 + 0.5f;
    const float W = (float)p.width, H = (float)p.height;

    uint2 range = p.tile_ranges[tile_list_index(p, tx, ty)];
    range.x = range.y ? 0xFFFFFFFFu - range.x : 0u;
    float T = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    bool done = !inside;

    // Two-deep software pipeline over 64-entry chunks, walked from the END of the range (near) to its start (far):
    // while chunk c is composited, the Splat gather of chunk c+1 and the entry-index load of chunk c+2 are in
    // flight.  All loads are UNCONDITIONAL (addresses clamped into the range, validity tracked separately) so the
    // loop body stays one basic block and the compiler can use counted s_waitcnt instead of draining vmcnt.
    if (range.y > range.x) {
    // entry index of this lane in the chunk that ends at hi_ (lane 0 = nearest); hi_ is clamped so the address
    // is always inside [range.x, range.y)
    auto entry_at = [&](uint32_t hi_) -> uint32_t {
        const uint32_t h = hi_ > range.x ? hi_ : range.x + 1u;
        const uint32_t nbb = (h - range.x) < 64u ? (h - range.x) : 64u;
        const uint32_t off = (uint32_t)lane < nbb ? (uint32_t)lane : nbb - 1u;
        return p.entry_vals[h - 1u - off];
    };
    auto chunk_len = [&](uint32_t hi_) -> uint32_t {
        return hi_ > range.x ? ((hi_ - range.x) < 64u ? (hi_ - range.x) : 64u) : 0u;
    };
    uint32_t hi = range.y;                        // chunk being composited ends here
    uint32_t hi1 = hi - chunk_len(hi);            // next chunk
    uint32_t idx_cur = entry_at(hi);
    uint32_t idx_next = entry_at(hi1);
    uint32_t w0, w1, w2, w3, w4;
    {
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(p.splats + (size_t)idx_cur * SPLAT_STRIDE);
        w0 = sp[0];
        w1 = sp[1];
        w2 = sp[2];
        w3 = sp[3];
        w4 = sp[4];
    }
    while (true) {
        const uint32_t nb = chunk_len(hi);
        const bool cur_valid = (uint32_t)lane < nb;
        // issue: Splat gather of the next chunk, entry indices of the one after
        const uint32_t* spn = reinterpret_cast<const uint32_t*>(p.splats + (size_t)idx_next * SPLAT_STRIDE);
        const uint32_t n0 = spn[0], n1 = spn[1], n2 = spn[2], n3 = spn[3], n4 = spn[4];
        const uint32_t hi2 = hi1 - chunk_len(hi1);
        const uint32_t idx_nn = entry_at(hi2);

        const StagedSplat s = decode_splat(w0, w1, w2, w3, w4, W, H, qx_lo, qy_lo, cur_valid);
        unsigned long long rel = __ballot(s.touch);
        while (rel) {
            const int k = __ffsll((long long)rel) - 1;
            rel &= rel - 1ull;
            const float dx = fx - bcast(s.cx, k), dy = fy - bcast(s.cy, k);
            const float p0 = bcast(s.i00, k) * dx + bcast(s.i01, k) * dy;
            const float p1 = bcast(s.i10, k) * dx + bcast(s.i11, k) * dy;
            const float a = p0 * p0 + p1 * p1;
            if (a <= CUT_A && !done) {
                const float bb = fminf(0.99f, __expf(-a) * bcast(s.alpha, k));
                const float wgt = bb * T;
                cr += wgt * bcast(s.r, k);
                cg += wgt * bcast(s.g, k);
                cb += wgt * bcast(s.b, k);
                T *= (1.0f - bb);
                if (T < T_MIN) done = true;
            }
        }
        if (__ballot(!done) == 0ull || hi1 <= range.x) {  // quadrant saturated, or that was the last chunk
            if (p.debug_consumed && lane == 0) atomicMax(p.debug_consumed + tile, range.y - hi1);
            break;
        }
        hi = hi1;
        hi1 = hi2;
        idx_next = idx_nn;
        w0 = n0;
        w1 = n1;
        w2 = n2;
        w3 = n3;
        w4 = n4;
    }
    }  // non-empty tile

    if (inside) {
        const float r = cr + p.background[0] * T;
        const float g = cg + p.background[1] * T;
        const float bch = cb + p.background[2] * T;
        const float al = (1.0f - T) + p.background[3] * T;
        char* row = reinterpret_cast<char*>(p.out) + (size_t)py * p.pitch;
        if (FORMAT == WS_FORMAT_RGBA32_FLOAT) {
            reinterpret_cast<float4*>(row)[px] = make_float4(r, g, bch, al);
        } else if (FORMAT == WS_FORMAT_RGBA16_FLOAT) {
            const uint32_t lo = (uint32_t)__half_as_ushort(__float2half_rn(r)) | ((uint32_t)__half_as_ushort(__float2half_rn(g)) << 16);
            const uint32_t hi2 = (uint32_t)__half_as_ushort(__float2half_rn(bch)) | ((uint32_t)__half_as_ushort(__float2half_rn(al)) << 16);
            reinterpret_cast<uint2*>(row)[px] = make_uint2(lo, hi2);
        } else {
            auto q8 = [](float v) -> uint32_t {
                v = fminf(fmaxf(v, 0.0f), 1.0f);
                return (uint32_t)__float2int_rn(v * 255.0f);
            };
            reinterpret_cast<uint32_t*>(row)[px] = q8(r) | (q8(g) << 8) | (q8(bch) << 16) | (q8(al) << 24);
        }
    }
}
