"""Study (GPU, capture mode): could a per-tile DEPTH CUT-OFF taken from an earlier view stop the binning from emitting -- and the
tile-id sort from sorting, twice -- the entries nobody reads?  (Round-4 verdict item 5: "do the study first".)

On the headline class of scenes 86 % of the binned, sorted (tile, splat) entries are never read: every tile's walk stops when
its pixels are saturated.  The idea priced here: remember, per tile, the view depth at which the walk of an EARLIER view
stopped, and for the next view of the same slot emit only entries nearer than that depth x (1 + margin); a tile whose walk
reaches the end of its truncated list unsaturated has to be drawn again from the full list (a redraw).

For every one of c4's 64 orbit views (BASELINE config 4: the c2 scene at 1920x1080) the capture build gives, per 32x32 tile, the
list, how many of its entries the walk consumed, and the depth keys; the key is bits(zfar - p.z) with p.z linear in the view
depth (camera.rs:216-234), so the view depth of every entry and of every tile's stopping point follows.  Then, for view v + D
with the cut-offs of view v (D = 1: the neighbouring view, 5.6 degrees on; D = 4: the previous frame of the same slot of a
four-slot batch, 22.5 degrees on) and a margin m:
   removable   entries of view v + D deeper than its tile's cut-off (never emitted)
   redraw      tiles of view v + D whose own stopping point lies deeper than the cut-off (they would end unsaturated), and the
               share of the view's entries they hold (re-emitted in full)
A tile that did NOT saturate in view v has no cut-off (nothing removed, nothing to redraw).

  python scripts/cutoff_study.py [out.json]        -> gpurun_out/cutoff_study_c4.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np  # noqa: E402


def main():
    import websplat as ws
    import bench
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "cutoff_study_c4.json")
    nviews = int(os.environ.get("CUTOFF_VIEWS", "64"))
    ctx = ws.Context(0)
    gpc, views, (w, h), _ = bench.build_workload(ws, "c4", 64)
    pc = ws.PointCloud(ctx, gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", gpc.sh_deg, False)
    r.enable_capture(True)
    per_view = []
    for v in range(nviews):
        a = views[v]
        r.prepare(pc, a)
        r.render(pc)
        ts = r.tile_stats(with_consumed=True)
        begin, end, entries = r.tile_lists()
        fr = r.download_frame()
        zn, zf = float(a.camera.znear), float(a.camera.zfar)
        kf = fr["keys"].view(np.float32).astype(np.float64)          # zfar' - p.z, zfar' = the shader's own zfar (= zf up to rounding)
        depth = zn + (zf - kf) * (zf - zn) / zf                      # view depth of every visible splat (store order)
        ll = ts["list_len"].astype(np.int64)
        co = ts["consumed"].astype(np.int64)
        nt = ll.size
        z_entry = depth[entries]                                     # per entry, far -> near inside a tile
        tile_of = np.repeat(np.arange(nt), ll)
        # stopping point: the FARTHEST consumed entry = entries[end - consumed]; unsaturated tiles (consumed == len) have none
        sat = (co < ll) & (co > 0)
        z_stop = np.full(nt, np.inf)
        idx = (end.astype(np.int64) - co)[sat]
        z_stop[sat] = z_entry[idx]
        # an unsaturated tile needs its whole list: its own "stopping depth" is its farthest entry
        z_need = z_stop.copy()
        uns = ~sat & (ll > 0)
        z_need[uns] = z_entry[begin.astype(np.int64)[uns]]
        per_view.append({"z_entry": z_entry, "tile_of": tile_of, "z_stop": z_stop, "z_need": z_need, "len": ll, "consumed": co,
                         "saturated": sat})
    r.close()
    pc.close()
    ctx.close()

    rows = []
    for delta in (1, 4):
        for margin in (0.0, 0.05, 0.10, 0.25, 0.50, 1.00):
            rem, red_t, red_e, dtot, tiles = 0, 0, 0, 0, 0
            for v in range(nviews):
                a, b = per_view[v], per_view[(v + delta) % nviews]
                cut = a["z_stop"] * (1.0 + margin)                   # inf where view v did not saturate: no cut-off
                deeper = b["z_entry"] > cut[b["tile_of"]]
                redraw = b["z_need"] > cut                           # the truncated list would end unsaturated
                redraw &= b["len"] > 0
                # entries of redrawn tiles are emitted in full after all: they are not removed
                removable = deeper & ~redraw[b["tile_of"]]
                rem += int(removable.sum())
                red_t += int(redraw.sum())
                red_e += int(b["len"][redraw].sum())
                dtot += int(b["len"].sum())
                tiles += int((b["len"] > 0).sum())
            rows.append({"delta_views": delta, "degrees": delta * 360.0 / 64, "margin": margin,
                         "removable_frac_of_D": rem / dtot, "redrawn_tiles_frac": red_t / tiles,
                         "entries_in_redrawn_tiles_frac_of_D": red_e / dtot})
    consumed = sum(int(p["consumed"].sum()) for p in per_view)
    total = sum(int(p["len"].sum()) for p in per_view)
    out = {"workload": "c4: the c2 scene (1.2 M Gaussians) at 1920x1080, the 64 orbit views; lists at the 32x32 compositing tile "
                       "(capture mode)", "views": nviews, "entries_per_view": total / nviews,
           "consumed_frac_of_D": consumed / total,
           "saturated_tiles_frac": float(np.mean([p["saturated"].mean() for p in per_view])),
           "table": rows,
           "reading": "worth building only if >= 50 % of D is removable at <= 1 % redrawn tiles (round-4 verdict item 5)"}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
