"""Where do k_blend's waves spend their time?  (GPU; round-5 verdict item 3.)

Renders frames of a workload with the time-stamped build of the compositing kernel (ws_renderer_enable_blend_timing: the
production form -- 32x32 tiles, f32 target, the frame's own binning -- with s_memtime stamps around every phase of a tile)
and prices the phases:

  start-up chain  tile-range load + the first batch's dependent gathers (entry index -> Splat record)
  gather wait     later batches: what the two-deep staging prefetch did not hide
  decode          Splat record -> affine map + quadrant mask -> LDS
  staging barrier waiting for the slowest stager (waves 8..15 of a tile do not stage: for them this IS the gather chain)
  vote barrier    the end-of-batch vote: waiting for the wave with the longest walk
  compaction      per-wave list of the records that reach the wave's quadrant
  walk            the (pixel, record) loop: VALU + LDS reads
  store           pixel store
  other           wave start skew inside a workgroup, stamps, loop overhead (tile duration minus the mean wave's phases)

Per tile the phase time is the MEAN over its 16 waves; a tile's duration is first wave start -> last wave end on the
100-MHz clock.  The kernel's time is priced as  sum over tiles(duration) / resident slots  (+ the tail where slots idle):
a tile holds one of 2 x CUs workgroup slots for its duration, so a phase's share of the slot time is its share of the
kernel.  Writes gpurun_out/blend_wait_breakdown_<workload>.json (and prints it).

  python scripts/blend_wait_breakdown.py hd1m [view] [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np  # noqa: E402

PHASES = ["range_load", "first_gather_chain", "gather_wait", "decode", "stage_barrier", "compaction", "walk", "vote_barrier",
          "store"]
GROUPS = {"start-up chain": ["range_load", "first_gather_chain"], "gather wait": ["gather_wait"], "decode": ["decode"],
          "staging barrier": ["stage_barrier"], "vote barrier": ["vote_barrier"], "compaction": ["compaction"], "walk": ["walk"],
          "store": ["store"]}


def analyse(tm, kernel_us, timing_kernel_us, num_cus, list_len=None):
    """tm: [tiles, 16, 16] uint32 (websplat.h layout).  -> dict"""
    tm = tm.astype(np.int64)
    live = tm[:, 0, 12] != tm[:, 0, 11]                      # tiles that ran (inside the image)
    t = tm[live]
    nt = t.shape[0]
    ph = t[:, :, :9].astype(np.float64)                     # cycles per phase, [tile, wave, phase]
    d_clk = (t[:, :, 12] - t[:, :, 11]) & 0xFFFFFFFF        # shader-clock cycles start -> end, per wave
    d_real = (t[:, :, 14] - t[:, :, 13]) & 0xFFFFFFFF       # 100-MHz ticks
    ok = d_real > 200                                       # (>= 2 us: enough ticks to calibrate)
    mhz = float(np.median(d_clk[ok] / d_real[ok]) * 100.0) if ok.any() else 2400.0
    us = 1.0 / mhz                                          # one cycle in us
    # the 100-MHz clock is chip-wide: tile timeline on it (wrap-safe relative to the earliest start)
    r0 = t[:, :, 13]
    base = int(r0.min())
    start = ((r0 - base) & 0xFFFFFFFF).min(axis=1) / 100.0   # us
    end = ((t[:, :, 14] - base) & 0xFFFFFFFF).max(axis=1) / 100.0
    dur = end - start
    span = float(end.max() - start.min())
    slots = 2 * num_cus
    mean_phase_us = ph.mean(axis=1) * us                    # [tile, phase]
    crit_phase_us = ph.max(axis=1) * us
    other = dur - mean_phase_us.sum(axis=1)
    slot_time = float(dur.sum() / slots)

    def grp(arr):  # [.., 9] -> dict of groups
        return {g: float(sum(arr[..., PHASES.index(p)] for p in ps)) for g, ps in GROUPS.items()}

    share = {g: float(sum(mean_phase_us[:, PHASES.index(p)].sum() for p in ps) / slots) for g, ps in GROUPS.items()}
    share["other (wave skew, stamps, loop overhead)"] = float(other.sum() / slots)
    # concurrency: tiles resident over the kernel's span
    ev = np.concatenate([np.stack([start, np.ones(nt)], 1), np.stack([end, -np.ones(nt)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0], append=ev[-1, 0])
    mean_conc = float((conc * dt).sum() / max(span, 1e-9))
    t_full = float(dt[conc >= 0.9 * slots].sum())
    order = np.argsort(dur)

    def tile_row(i):
        row = {"tile_duration_us": float(dur[i]), "start_us": float(start[i]), "batches": int(t[i, 0, 9]),
               "records_walked_mean_wave": float(t[i, :, 10].mean()), "records_walked_max_wave": int(t[i, :, 10].max()),
               "mean_wave_us": grp(mean_phase_us[i]), "slowest_wave_per_phase_us": grp(crit_phase_us[i]),
               "other_us": float(other[i])}
        if list_len is not None:
            row["list_len"] = int(list_len[i])
        return row
    walked = t[:, :, 10].astype(np.float64)
    walk_us = ph[:, :, 6] * us
    # stager waves (0..7 stage the 512-entry batches) against the rest, chip-wide means per tile
    role = {}
    for name, sl in (("stager waves 0-7", slice(0, 8)), ("other waves 8-15", slice(8, 16))):
        role[name] = {g: float(sum((ph[:, sl, PHASES.index(p)].mean(axis=1) * us).sum() for p in ps) / nt) for g, ps in GROUPS.items()}
    out = {
        "tiles": int(nt), "shader_clock_mhz": mhz, "resident_slots": slots,
        "kernel_us_production_build": kernel_us, "kernel_us_timing_build": timing_kernel_us,
        "span_us_first_start_to_last_end": span,
        "slot_time_us (sum of tile durations / slots)": slot_time,
        "tail_and_idle_us (span - slot time)": span - slot_time,
        "mean_tiles_resident": mean_conc, "us_with_at_least_90pct_slots_busy": t_full,
        "kernel_share_us": share,
        "kernel_share_sum_us": float(sum(share.values())),
        "kernel_share_frac_of_span": {k: v / span for k, v in share.items()},
        "tile_duration_us_pct": {str(p): float(np.percentile(dur, p)) for p in (10, 50, 90, 99, 100)},
        "mean_tile_us_by_wave_role": role,
        "median_tile": tile_row(int(order[nt // 2])),
        "p99_tile": tile_row(int(order[min(nt - 1, int(nt * 0.99))])),
        "longest_tile": tile_row(int(order[-1])),
        "walk": {"records_per_wave_mean": float(walked.mean()), "us_per_wave_mean": float(walk_us.mean()),
                 "cycles_per_record_per_wave": float((ph[:, :, 6].sum()) / max(walked.sum(), 1.0)),
                 "note": "wall cycles a wave spends per record it walks, with the CU's other waves competing for the same "
                         "SIMD: ~41 issue cycles of its own (14 VALU at 2.4 + v_exp_f32 at 8) times the waves walking at the same time"},
        "start_time_us_pct (when tiles begin: dispatch rounds)": {str(p): float(np.percentile(start, p)) for p in (1, 25, 50, 75, 99)},
    }
    return out


def main():
    import websplat as ws
    import bench
    name = sys.argv[1] if len(sys.argv) > 1 else "hd1m"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", f"blend_wait_breakdown_{name}.json")
    ctx = ws.Context(0)
    gpc, views, (w, h), _ = bench.build_workload(ws, name, 8)
    pc = ws.PointCloud(ctx, gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", gpc.sh_deg, gpc.compressed)
    r.enable_timers(2)

    def blend_us():
        return [ms * 1e3 for n, ms in r.kernel_times() if n == "k_blend"][-1]
    for _ in range(4):
        r.prepare(pc, views[view])
        r.render(pc)
    prod = []
    for _ in range(5):
        r.prepare(pc, views[view])
        r.render(pc)
        prod.append(blend_us())
    img_prod = r.download_target().copy()
    r.enable_blend_timing(True)
    timed = []
    for _ in range(3):
        r.prepare(pc, views[view])
        r.render(pc)
        timed.append(blend_us())
    tm = r.blend_timing()
    img_timed = r.download_target()
    st = r.frame_stats()
    ll = r.tile_stats()["list_len"]
    bw, bh = r.binning_tile()
    tw, th = ctx.tile_size()
    tiles_x = -(-w // tw)
    tiles_y = -(-h // th)
    # list length of every BLEND tile (the frame may have binned 2 x 2 blend tiles per list)
    s = 1 if bw > tw else 0
    lx = (tiles_x + s) >> s
    lidx = ((np.arange(tiles_y)[:, None] >> s) * lx + (np.arange(tiles_x)[None, :] >> s)).reshape(-1)
    list_len_tile = ll[lidx]
    live = tm[:, 0, 12] != tm[:, 0, 11]
    ncu = ctx.device_info()["cus"]
    out = analyse(tm, float(np.median(prod)), float(np.median(timed)), int(ncu), list_len_tile[live])
    out.update({"workload": name, "view": view, "viewport": [w, h], "binning_tile": [bw, bh], "num_visible": st["num_visible"],
                "num_tile_entries": st["num_tile_entries"],
                "image_identical_to_production_build": bool(np.array_equal(img_prod, img_timed)),
                "note": "event-interval kernel times (one frame in flight, includes ~3.5 us dispatch overhead); phases are "
                        "means over a tile's 16 waves; kernel_share_us = sum over tiles / resident slots"})
    os.makedirs(os.path.dirname(path), exist_ok=True)
    out["blend_order"] = os.environ.get("WS_BLEND_ORDER", "1") != "0"
    np.savez_compressed(path.replace(".json", "_raw.npz"), tm=tm, list_len=list_len_tile)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))
    r.close()
    pc.close()
    ctx.close()


if __name__ == "__main__":
    main()
