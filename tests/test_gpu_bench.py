"""bench.py on the GPU box: the one-rank RCCL communicator (the collective code path of the N-GPU run), the JSON
contract, the real-scene hook (WEBSPLAT_BONSAI_PLY / WEBSPLAT_BONSAI_CAMERAS), and the error-bit check that
invalidates a run whose frames dropped tile entries."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from websplat import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=e, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_one_rank_rccl_and_contract():
    """`python bench.py --gpus 1` initialises a one-rank nccl (= RCCL) process group and sends the timing through the
    same barrier + MAX all-reduce the 8-GPU run uses."""
    p, out = _bench(["--gpus", "1", "--steps", "40", "--warmup", "10", "--workload", "c1", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-3000:]
    assert out is not None, p.stdout[-2000:]
    assert out["config"]["collective"] == "nccl communicator, world size 1", out["config"]["collective"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 40 and out["warmup"] == 10 and out["value"] > 0
    assert out["config"]["workload"].startswith("c1:") and out["config"]["error_bits"] == 0
    # the timed region is bracketed by host barriers (a gloo group beside the RCCL communicator) + device synchronize; the
    # binning tile is what the device chose for the analysed frames (the compositing tile or 2 x 2 of them)
    assert out["config"]["timing_barrier"].startswith(("host barrier (gloo group)", "nccl barrier"))
    assert out["config"]["compositing_tile"] == "32x32" and out["config"]["binning_tile"] in ("32x32", "64x64")
    rf = out["roofline"]
    # `bound` names the resource the dominant kernel runs out of (round 6): "valu" for k_blend when the PMC pass of the workload
    # is on file (frac = wave64 VALU issue slots used), "hbm" otherwise; the HBM figure is always there as hbm_frac; the whole
    # frame is priced against HBM at the driver-timed frame period
    assert rf["bound"] in ("hbm", "valu") and rf["hbm_frac"] is not None and 0 < rf["hbm_frac"] < 1
    if rf["bound"] == "hbm":
        assert rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["frac"] == rf["hbm_frac"]
    else:
        assert rf["kernel"] == "k_blend" and rf["unit"] == "G wave64-inst/s" and rf["valu_cycles_per_inst"] == 2.0
        assert 0 < rf["frac"] < rf["frac_at_measured_issue_rate"] < 1.2 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-9
    fr = rf["frame"]
    assert fr["bound"] == "hbm" and fr["peak"] == 8000.0 and fr["unit"] == "GB/s" and fr["ms_per_step"] == out["ms_per_step"]
    assert abs(fr["frac"] - fr["alg_bytes_per_frame"] / (out["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9 and 0 < fr["frac"] < 1
    assert abs(out["value"] - 40 / (out["ms_per_step"] * 40 / 1e3)) < 1e-6 * out["value"]
    # the host side of the timed region is always on the record (round-3 verdict: an 8-GPU run that comes back sub-linear
    # must say whether the enqueue threads kept up)
    cfg = out["config"]
    assert cfg["host_enqueue_ms_per_frame"] > 0 and isinstance(cfg["host_bound"], bool) and cfg["host_cpus"] >= 1
    # (40 frames of c1 last a few milliseconds: CPU time comes in 10-ms ticks, so the busy-cores ratio is null below 0.1 s)
    assert (cfg["host_cores_busy_per_rank"] is None or 0 < cfg["host_cores_busy_per_rank"] < 64) and "host_cpu_quota" in cfg
    assert "secondary" not in out      # (only the default workload carries the c3 block)
    # round 6: what runs beside what with frames in flight, measured on the device behind the timed region
    fl = out["inflight"]
    assert "error" not in fl, fl
    assert fl["us_per_frame"] > 0 and fl["k1_in_flight_us"] > 0 and fl["blend_in_flight_us"] > 0
    assert abs(sum(fl["us_per_frame_with"].values()) - fl["us_per_frame"]) < 0.05 * fl["us_per_frame"] + 1.0


def _cpu_groups():
    sys.path.insert(0, ROOT)
    import bench
    return bench._physical_cores(sorted(os.sched_getaffinity(0)))


def test_bench_secondary_c3_block():
    """The default invocation (hd1m, one GPU) also times c3 -- the largest single-GPU configuration -- and reports it
    under "secondary" with the roofline of its dominant kernel and the HBM fractions of the depth sort and K1."""
    p, out = _bench(["--steps", "200", "--warmup", "20", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-3000:]
    c3 = out["secondary"]["c3"]
    assert c3["workload"].startswith("c3: 5000000 Gaussians, 1920x1080") and c3["frames"] == 200
    assert c3["value"] > 100 and c3["single_stream_fps"] > 100 and c3["avg_visible"] > 4_000_000
    assert c3["roofline"]["kernel"] and 0 < c3["roofline"]["frac"] < 1 and c3["roofline"]["bound"] in ("hbm", "valu")
    assert 0 < c3["roofline"]["hbm_frac"] < 1 and 0 < c3["roofline"]["frame"]["frac"] < 1
    for name in ("depth sort", "K1"):
        assert 0 < c3["kernels"][name]["frac"] < 1, (name, c3["kernels"][name])
    assert out["config"]["workload"].startswith("hd1m:") and out["value"] > 1000


def test_bench_holds_its_rate_on_a_sliver_of_the_host(tmp_path):
    """Eight ranks share one node's host.  What one rank needs: measured on the one-GPU box by pinning the whole bench
    process (`taskset`) to FOUR logical CPUs -- two physical cores of the first eighth of the host -- while stand-ins for
    the other seven ranks (two busy threads each, scripts/ubench/cpu_burn.c) spin on CPUs of the other seven eighths.  The
    hd1m line must hold the rate of the same line with the whole host to itself (within 7 %: run-to-run spread is +-2 %)
    and say that the host was not the limit.  The burners respect the container's CPU QUOTA (cgroup cpu.max: this build's
    boxes show 256 CPUs and grant 16 CPUs' worth of run time; 224 busy threads throttle the whole cgroup, bench included,
    to a sixth of its rate -- profiles/r04/host_share_probe.txt -- which says nothing about bench.py)."""
    sys.path.insert(0, ROOT)
    import bench
    cores = _cpu_groups()
    if len(cores) < 16:
        pytest.skip("fewer than 16 physical cores: an eighth of the host is not a meaningful share")
    quota = bench.cpu_quota() or float(len(cores))
    if quota < 8:
        pytest.skip("the container grants fewer than 8 CPUs' worth of run time: no room for the other ranks' stand-ins")
    eighth = len(cores) // 8
    mine = sorted(c for g in cores[:2] for c in g)                 # two physical cores (four logical CPUs with SMT)
    nburn = int(min(14, max(0, quota - 6)))                        # two busy threads per other rank, inside the quota
    burn_cpus = [cores[(1 + i % 7) * eighth + i // 7][0] for i in range(nburn)]
    burn = str(tmp_path / "cpu_burn")
    subprocess.run(["gcc", "-O2", "-pthread", os.path.join(ROOT, "scripts", "ubench", "cpu_burn.c"), "-o", burn], check=True)
    args = ["--steps", "1500", "--warmup", "50", "--no-cpu-baseline", "--no-secondary"]
    p0, free = _bench(args)
    assert p0.returncode == 0, p0.stderr[-3000:]
    burner = subprocess.Popen([burn, "600", *map(str, burn_cpus)], stdout=subprocess.PIPE, text=True) if nburn else None
    try:
        if burner:
            burner.stdout.readline()   # "burning N cpus": the threads run
        p = subprocess.run(["taskset", "-c", ",".join(map(str, mine)), sys.executable, os.path.join(ROOT, "bench.py"), *args],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ), cwd=ROOT)
    finally:
        if burner:
            burner.kill()
            burner.wait()
    assert p.returncode == 0, p.stderr[-3000:]
    shared = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    keys = ("host_enqueue_ms_per_frame", "host_cores_busy_per_rank", "host_cpus", "host_cpu_quota", "host_bound")
    rep = {"whole_host": dict({"fps": free["value"]}, **{k: free["config"][k] for k in keys}),
           "four_cpus_beside_burners": dict({"fps": shared["value"], "burner_threads": nburn}, **{k: shared["config"][k] for k in keys})}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "host_contention.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert shared["config"]["host_cpus"] == len(mine)
    assert shared["config"]["host_bound"] is False, rep
    assert shared["value"] >= 0.93 * free["value"], rep
    # eight such ranks fit the host with room to spare: cores busy per rank x 8 stays below three quarters of what the
    # CONTAINER may use -- the cgroup quota where there is one (16 CPUs on this build's boxes, whatever the 256 in the affinity
    # mask say: round-4 verdict item 2) -- and one rank stays below one core (round 4: 1.9; round 5: the host's run-ahead is
    # bounded by sleeping on a mailbox word, ws_view_batch)
    assert shared["config"]["host_cores_busy_per_rank"] <= 1.0, rep
    assert 8 * shared["config"]["host_cores_busy_per_rank"] <= 0.75 * min(len(cores), quota), rep


def test_eight_ranks_fit_the_hosts_quota():
    """Round-4 verdict item 2(d): eight processes -- launched by bench.py itself from a plain shell, each with its own scene,
    view batch and submission thread, all drawing on the one GPU of this box (--single-device; the collectives on host
    tensors) -- must together stay inside what the container may use of the host: the sum of the cores the ranks keep busy
    is below three quarters of the cgroup quota.  The GPU is shared eight ways here, so the frame rate says nothing; frames
    per CPU-second is what the record keeps (gpurun_out/eight_rank_stand_in.json)."""
    sys.path.insert(0, ROOT)
    import bench
    cores = _cpu_groups()
    quota = bench.cpu_quota() or float(len(cores))
    if quota < 8 or len(cores) < 16:
        pytest.skip("needs a host share of at least 8 CPUs / 16 physical cores")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "OMP_NUM_THREADS")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--single-device", "--dist-backend", "gloo",
                        "--workload", "c2", "--steps", "400", "--warmup", "20", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["error_bits"] == 0
    busy = cfg["host_cores_busy_per_rank"]                    # MAX over the ranks
    rep = {"ranks": 8, "workload": cfg["workload"], "frames_per_s_all_ranks_one_gpu": out["value"], "host_cpu_quota": quota,
           "host_cores_busy_per_rank_max": busy, "sum_upper_bound": 8 * busy,
           "frames_per_cpu_second": out["value"] / max(8 * busy, 1e-9), "host_affinity_rank0": cfg["host_affinity"]}
    with open(os.path.join(ROOT, "gpurun_out", "eight_rank_stand_in.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert 8 * busy <= 0.75 * quota, rep


def test_bench_real_scene_hook(tmp_path):
    """--workload bonsai loads the scene named by WEBSPLAT_BONSAI_PLY through the library's PLY reader and the cameras of
    WEBSPLAT_BONSAI_CAMERAS through its cameras.json reader (scene.rs), at 1200x799 (README.md:55)."""
    rows = synth.scene_c2(n=60_000, seed=11)
    ply = str(tmp_path / "point_cloud.ply")
    synth.write_ply(ply, rows, 3)
    cams = synth.orbit_cameras(9, 1559, 1039, 1160.0, 1160.0)
    cj = str(tmp_path / "cameras.json")
    synth.write_cameras_json(cj, cams)
    p, out = _bench(["--steps", "30", "--warmup", "5", "--workload", "bonsai", "--no-cpu-baseline", "--no-dist"],
                    env={"WEBSPLAT_BONSAI_PLY": ply, "WEBSPLAT_BONSAI_CAMERAS": cj})
    assert p.returncode == 0, p.stderr[-3000:]
    cfg = out["config"]
    assert cfg["workload"].startswith("bonsai: 60000 Gaussians") and cfg["width"] == 1200 and cfg["height"] == 799
    assert cfg["workload_note"].startswith("real scene point_cloud.ply") and "cameras.json" in cfg["workload_note"]
    assert out["data"].startswith("real scene file")
    assert cfg["views"] == 7                      # 7 of 8 cameras are training views (scene.rs:143-151): 9 -> 7 train
    assert cfg["collective"].startswith("disabled")
    # without the asset: a clear message and the stand-in (not run here: 1.2 M Gaussians); the selection logic itself:
    sys.path.insert(0, ROOT)
    import bench
    import websplat as ws
    os.environ.pop("WEBSPLAT_BONSAI_PLY", None)
    assert "bonsai" in bench.WORKLOADS and ws.read_ply(ply).num_points == 60_000


def test_sticky_error_bits_and_driver_retry(ws, ctx, oracle, tmp_path):
    """A frame whose (tile, splat) list overflows its capacity sets bit 0 of the renderer's sticky error word; the word
    survives later (good) frames until reset, a view batch ORs its slots, and ws_render_views grows the list and renders
    again instead of writing an image that lost its nearest splats."""
    import scenes
    sc = scenes.c2(ws, oracle, n=150_000, viewport=(640, 480))
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        r.prepare(pc, sc.args)
        r.render(pc)
        good = r.download_target()
        st = r.frame_stats()
        need = st["num_tile_entries"]
        assert r.errors() == (0, need)
        r.set_tile_entry_capacity(max(4096, need // 3))
        r.prepare(pc, sc.args)
        r.render(pc)
        st2 = r.frame_stats()
        assert st2["overflow"] & 1 and st2["num_tile_entries"] <= max(4096, need // 3)
        bits, needed = r.errors()
        assert bits & 1 and needed == need                     # the unclamped demand is reported
        r.set_tile_entry_capacity(0)                           # automatic again
        r.prepare(pc, sc.args)
        r.render(pc)
        assert r.frame_stats()["overflow"] == 0
        assert r.errors()[0] & 1                               # sticky across the good frame ...
        assert r.errors(reset=True)[0] & 1 and r.errors()[0] == 0   # ... until reset
        assert np.array_equal(r.download_target(), good)
        # the sticky word also survives a prepare() that FAILS (it used to read back as 0 until the next
        # successful prepare, hiding the bits from ws_measure / bench.py)
        r.set_tile_entry_capacity(max(4096, need // 3))
        r.prepare(pc, sc.args)
        r.render(pc)
        assert r.errors()[0] & 1
        import dataclasses
        bad = dataclasses.replace(sc.args, viewport=(0, 0)) if dataclasses.is_dataclass(sc.args) else None
        if bad is not None:
            with pytest.raises(ws.WebSplatError):
                r.prepare(pc, bad)
            assert r.errors()[0] & 1
        r.set_tile_entry_capacity(0)
        r.errors(reset=True)
    finally:
        r.close()
        pc.close()


def test_automatic_entry_capacity_grows_to_an_overflowed_frames_demand(ws, ctx, oracle):
    """The automatic (tile, splat) capacity is sized for twice the BASELINE scenes' demand (4 entries per Gaussian per
    Mpixel, at least 8 M -- round-3 verdict: 24 per Gaussian were 3-4 GB per renderer on c3).  A frame that needs more is
    flagged, leaves its demand in the renderer's sticky words, and once a read-back has seen it the NEXT prepare() allocates
    1.25 x that: the same view then draws completely, and equals the image of a renderer given the capacity up front."""
    import scenes
    rows = synth.scene_c1(n=150_000, seed=31)
    rows[:, 55:58] = np.log(0.6)            # every splat covers most of the 640x480 viewport: ~150 binning tiles each
    rows[:, 54] = -3.0                      # faint, so that the tiles do not saturate after a handful
    sc = scenes.c1(ws, oracle, n=150_000, viewport=(640, 480), seed=31)
    sc.gpc = ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    big = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        r.prepare(pc, sc.args)
        r.render(pc)
        st = r.frame_stats()
        assert st["overflow"] & 1 and st["tile_entries_capacity"] == 8 << 20, st
        bits, need = r.errors(reset=True)
        assert bits & 1 and need > st["tile_entries_capacity"]
        r.prepare(pc, sc.args)               # grows: the demand has been seen
        r.render(pc)
        st2 = r.frame_stats()
        assert st2["overflow"] == 0 and st2["tile_entries_capacity"] >= need and st2["num_tile_entries"] == need
        assert r.errors()[0] == 0
        big.set_tile_entry_capacity(2 * need)
        big.prepare(pc, sc.args)
        big.render(pc)
        assert np.array_equal(r.download_target(), big.download_target())
    finally:
        big.close()
        r.close()
        pc.close()


def test_entry_capacity_grows_without_anyone_polling(ws, ctx, oracle):
    """no render loop polls ws_renderer_errors on its own, so a view heavier than the automatic capacity kept
    dropping its NEAREST splats frame after frame.  The blend now posts an overflowed frame's demand to a host-visible
    mailbox word and prepare() reads it (no synchronisation): a plain prepare / render loop that never looks at the error
    words or the frame statistics draws completely from the second or third frame on; the overflowed frames stay flagged."""
    import scenes
    rows = synth.scene_c1(n=150_000, seed=31)
    rows[:, 55:58] = np.log(0.6)
    rows[:, 54] = -3.0
    sc = scenes.c1(ws, oracle, n=150_000, viewport=(640, 480), seed=31)
    sc.gpc = ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
    pc = ws.PointCloud(ctx, sc.gpc)
    r = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    big = ws.GaussianRenderer(ctx, "rgba32float", 3, False)
    try:
        for _ in range(4):                   # a caller that only draws
            r.prepare(pc, sc.args)
            r.render(pc)
            ctx.sync()
        st = r.frame_stats()
        assert st["overflow"] == 0 and st["tile_entries_capacity"] > 8 << 20, st
        assert st["tile_entries_capacity"] >= st["num_tile_entries"]
        assert r.errors(reset=True)[0] & 1   # the first frame(s) did overflow, and say so
        big.set_tile_entry_capacity(2 * st["num_tile_entries"])
        big.prepare(pc, sc.args)
        big.render(pc)
        assert np.array_equal(r.download_target(), big.download_target())
        # the demand belongs to this scene and viewport: another viewport starts from the formula again
        small = scenes.c1(ws, oracle, n=150_000, viewport=(320, 240), seed=31)
        r.prepare(pc, small.args)
        r.render(pc)
        assert r.frame_stats()["tile_entries_capacity"] == 8 << 20
    finally:
        big.close()
        r.close()
        pc.close()


def test_two_ranks_share_one_gpu_and_draw_identical_views():
    """The N-rank path of bench.py with REAL frames on a one-GPU box: two processes (one per rank, started by bench.py
    itself through torch.distributed.run, as the driver's launcher would), both rendering their shard of the views on
    cuda:0 (--single-device), the collectives on host tensors (--dist-backend gloo: RCCL refuses two ranks on one device).
    Rank 1 draws the odd views, rank 0 the even ones; afterwards every rank draws the first view of every shard and rank 0
    compares the digests: a view's image does not depend on the rank (view_shard_determinism).  The 1 -> 8 GPU
    curve itself stays unmeasured on this box."""
    # from a plain shell, as the driver starts the N = 1 line: bench.py launches its own two ranks (round-4 verdict item 1;
    # the torch.distributed.run form of the same job is covered on CPU by tests/test_shard.py)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c2", "--steps", "30",
           "--warmup", "6", "--views", "8", "--no-cpu-baseline", "--dist-backend", "gloo", "--single-device",
           "--check-shard-determinism"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "4"
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and p.stdout.count("\n") == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 30 and out["scaling"] == "weak" and out["value"] > 0
    cfg = out["config"]
    assert "gloo communicator, world size 2" in cfg["collective"] and cfg["error_bits"] == 0
    sd = cfg["shard_determinism"]
    assert sd["views"] == [0, 1] and sd["ranks"] == 2 and sd["identical"] is True and sd["distinct_images"] == 2
    # whole-job value: both ranks' frames over the MAX elapsed
    assert abs(out["value"] - 2 * 30 / (out["ms_per_step"] * 30 / 1e3)) < 1e-6 * out["value"]


def test_frames_in_flight_stay_bit_identical_over_a_long_run():
    """scripts/soak.py in small: the 64 orbit views of c2 drawn 41 times back to back with four frames in flight (2624 frames:
    the host's run-ahead window, the slots' scratch and mailboxes and the per-frame zero arena cycle hundreds of times), a lone
    renderer of the same context drawing in between.  The targets of the last pass equal those of the first pass bit for bit, the
    device's counts are unchanged, no error bit is set.  (The long form -- 96 000 frames of hd1m -- is a script run:
    profiles/r05/soak_hd1m.json.)"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import soak
    res = soak.soak("c2", rounds=40)
    assert res["pass"], res
    assert res["frames"] == 40 * 64 and res["distinct_images"] == 64 and res["views_that_differ"] == []


def test_two_rccl_ranks_on_one_device_end_with_a_named_error_not_a_hang():
    """Round 6 (verdict r05 item 8b): until now only gloo had seen N > 1 ranks.  `python bench.py --gpus 2` on a box with ONE
    device must not reach RCCL's rendezvous with two ranks on one GPU (ncclInvalidUsage at best, a hang in the bootstrap at
    worst): every rank stops before it, by name, and the launcher returns non-zero within the timeout."""
    import ctypes as C
    import time
    ndev = C.c_int(0)   # (not through torch: its bundled HIP runtime must not be initialised inside the pytest process)
    assert C.CDLL("libamdhip64.so").hipGetDeviceCount(C.byref(ndev)) == 0
    if ndev.value != 1:
        pytest.skip("needs a box with exactly one device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert "RCCL needs one device per rank: 2 local rank(s), 1 visible device(s)" in p.stderr, p.stderr[-2000:]
    assert time.time() - t0 < 240
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]      # no result line from a run that did not happen
    # --single-device with the nccl backend is the same mistake, said the same way
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--workload", "c1", "--steps", "8",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "RCCL needs one device per rank" in p.stderr and "--single-device" in p.stderr
