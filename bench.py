#!/usr/bin/env python3
"""bench.py -- frames/s of the render hot path (prepare + render) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W      (N > 1 from a plain shell: bench.py starts its own N ranks,
                                                      one per GPU, through torch.distributed.run on 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
                                                     (the driver's form: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env)

A "step" is one frame = one pass of the hot path (K1 preprocess -> depth radix sort -> tile binning ->
tile blend) over one camera view of the synthetic scene, inputs resident in HBM, output left in HBM.
Procedure = the reference's bin/measure.rs:98-153: frames are enqueued back-to-back with a single sync at the
end (throughput, not latency).  The views of a batch are independent, so `--streams` frames (default 4) are in
flight per GPU, each on its own HIP stream with its own renderer scratch and target, sharing the resident scene;
the one-frame-at-a-time rate is reported beside it (config.single_stream_fps).  The scene is replicated on every
GPU and views are sharded view i -> rank i mod N (no data-path collective; "scaling": "weak": every rank
renders K frames).  The only collective is the MAX all-reduce of the elapsed time after the timed region; it runs
through RCCL also at N = 1 (a one-rank communicator), so the code path the 8-GPU run takes is exercised by every run.

Workloads (BASELINE.json configs):
  hd1m (default) the north-star headline: bonsai-like synthetic, 1 M Gaussians, 1920x1080
  c2            bonsai-like synthetic, 1.2 M Gaussians, 1200x799 -- configs[1]; the real bonsai .ply is not on disk
  bonsai        the REAL bonsai scene when WEBSPLAT_BONSAI_PLY (and optionally WEBSPLAT_BONSAI_CAMERAS) point at it,
                1200x799 (README.md:55, the one number the reference publishes); without the asset: a clear message
                and the c2 stand-in
  c3            5 M Gaussians, 1920x1080 (sort stress) -- configs[2], the largest single-GPU configuration
  c4            the c2 scene at 1920x1080, 64-view batch -- configs[3]
  c5            compressed c3dgs .npz (native loader), 1 M Gaussians, 3840x2160 -- configs[4]
  c1            10 k Gaussians, 800x600 -- configs[0]
  realistic1m   1 M Gaussians with the SIZE DISTRIBUTION of a trained indoor scene (background-sized splats, needles, discs,
                bimodal opacity), 1920x1080 -- not a BASELINE configuration: what the entry capacity, the binning decision
                and the rectangle packing meet on real data

`--dry-run` (CPU, backend gloo; used by tests/test_shard.py with world size 2) runs the same sharding, planning,
barrier and reduction code without a GPU: frames are not rendered and the line says so ("dry_run": true).
"""
import argparse
import json
import os
import socket
import sys
import time

# HIP multiplexes user streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with the null stream): with
# four frames in flight two of them can end up serialised on one queue (measured: 4800 instead of 5600 frames/s).
# The variable is read when the HIP runtime initialises, i.e. before torch / the library make their first call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "web-splat_amd"), os.path.join(ROOT, "tests"), ROOT]

import numpy as np  # noqa: E402

NUM_SIMDS = 1024                      # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_INST = 2.0            # a wave64 VALU instruction on a SIMD (MI355X_MICROARCH.md); the roofline's `peak`
VALU_CYCLES_PER_INST_MEASURED = 2.4   # scripts/ubench/valu_rate.hip on this part (DESIGN_LOG.md): reported beside it
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = ("hd1m", "c2", "bonsai", "c3", "c4", "c5", "c1", "realistic1m")


def build_workload(ws, name, n_views):
    """-> (host point cloud, [SplattingArgs per view], (w, h), note)."""
    from websplat import synth
    note = None
    cams = None
    rows = gpc = None
    if name == "bonsai":
        ply = os.environ.get("WEBSPLAT_BONSAI_PLY")
        cams_json = os.environ.get("WEBSPLAT_BONSAI_CAMERAS")
        if ply and os.path.exists(ply):
            # the reference's own loading path: io/ply.rs (or io/npz.rs by magic bytes) + scene.rs cameras.json
            w, h = 1200, 799
            with open(ply, "rb") as f:
                magic = f.read(4)
            gpc = ws.read_npz(ply) if magic == b"PK\x03\x04" else ws.read_ply(ply)
            if cams_json and os.path.exists(cams_json):
                scene = ws.Scene.from_json(cams_json)
                cams = scene.cameras("train")[:n_views] or scene.cameras(None)[:n_views]
                scene.close()
                note = f"real scene {os.path.basename(ply)}, {len(cams)} cameras of {os.path.basename(cams_json)}"
            else:
                r = float(np.linalg.norm(np.asarray(gpc.aabb.max) - np.asarray(gpc.aabb.min))) * 0.35
                cams = synth.orbit_cameras(n_views, w, h, 1200.0, 1200.0, radius=max(r, 1.0), height_off=0.25 * max(r, 1.0))
                note = f"real scene {os.path.basename(ply)}, synthetic orbit cameras (WEBSPLAT_BONSAI_CAMERAS not set)"
        else:
            print("[bench] workload 'bonsai': WEBSPLAT_BONSAI_PLY is not set or does not exist -- the asset is not on "
                  "disk and cannot be downloaded here; falling back to the seeded bonsai-like stand-in (c2).", file=sys.stderr)
            note = "WEBSPLAT_BONSAI_PLY not supplied: synthetic stand-in (same as c2)"
            name = "c2"
    if name == "c2":
        rows, (w, h), f = synth.scene_c2(n=1_200_000, seed=1), (1200, 799), 1200.0
        cams = synth.orbit_cameras(n_views, w, h, f, f)
    elif name == "c4":
        # BASELINE config 4: the c2 scene at 1920x1080, 64-view batch (view i -> rank i mod N)
        rows, (w, h), f = synth.scene_c2(n=1_200_000, seed=1), (1920, 1080), 1920.0
        cams = synth.orbit_cameras(n_views, w, h, f, f)
    elif name == "hd1m":
        rows, (w, h), f = synth.scene_c2(n=1_000_000, seed=1), (1920, 1080), 1920.0
        cams = synth.orbit_cameras(n_views, w, h, f, f)
    elif name == "realistic1m":
        # the size distribution of a trained indoor scene (heavy tail of background-sized splats, needles, discs, bimodal
        # opacity: synth.scene_realistic), 1 M Gaussians at 1920x1080 on the hd1m orbit
        rows, (w, h), f = synth.scene_realistic(n=1_000_000, seed=5), (1920, 1080), 1920.0
        cams = synth.orbit_cameras(n_views, w, h, f, f)
    elif name == "c3":
        rows, (w, h) = synth.scene_c3(n=5_000_000, seed=2), (1920, 1080)
        cams = [synth.camera_c3(w, h)] * n_views
    elif name == "c1":
        rows, (w, h) = synth.scene_c1(n=10_000, seed=0), (800, 600)
        cams = [synth.camera_c1(w, h)] * n_views
    elif name == "c5":
        # BASELINE config 5: compressed c3dgs scene at 3840x2160, written as a real .npz and read back by the
        # library's native loader (io/npz.rs path)
        import tempfile
        w, h = 3840, 2160
        a = synth.c3dgs_arrays(n=1_000_000, n_geometry=4096, n_sh=4096, seed=3, sh_deg=3, extent=1.0)
        a["scaling_factor_zero_point"] = np.array(390, dtype=np.int32)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "c5.npz")
            synth.write_npz(path, a)
            gpc = ws.read_npz(path)
        cams = synth.orbit_cameras(n_views, w, h, 3000.0, 3000.0, radius=3.2, height_off=0.6)
    elif name != "bonsai":
        raise SystemExit(f"unknown workload {name} (one of {', '.join(WORKLOADS)})")
    if rows is not None:
        if os.environ.get("WS_BENCH_SCENE_ORDER") == "morton":
            # A/B only (never the default, never the reported configuration): the synthetic scenes are generated in
            # random order; this stores them along a 3-D Morton curve instead, as a trained scene roughly is
            p = rows[:, :3].astype(np.float64)
            q = ((p - p.min(0)) / np.maximum(p.max(0) - p.min(0), 1e-12) * 1023.0).astype(np.uint64)
            code = np.zeros(len(rows), dtype=np.uint64)
            for bit in range(10):
                for axis in range(3):
                    code |= ((q[:, axis] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + axis)
            rows = rows[np.argsort(code, kind="stable")]
            note = (note + "; " if note else "") + "scene stored in Morton order (A/B)"
        gpc = ws.GenericGaussianPointCloud.from_ply_rows(rows, 3)
    args = []
    for cj in cams:
        cam = ws.PerspectiveCamera.from_scene_camera(cj.position, cj.rotation, cj.fx, cj.fy, cj.width, cj.height)
        cam.fit_near_far(gpc.aabb)
        args.append(ws.SplattingArgs(camera=cam, viewport=(w, h), max_sh_deg=min(3, gpc.sh_deg)))
    return gpc, args, (w, h), note


def cpu_baseline(gpc, arg, viewport, budget_s=20.0):
    """The oracle (CPU restatement of the reference path) timed on this host's cores: a bounded sample of
    whole frames of the same workload.  Reported, never used by the product path."""
    os.environ["WS_ORACLE_NATIVE"] = "1"  # -O3 -march=native copy built ON THIS HOST (never shipped between machines)
    import oracle_lib as oracle
    flags = oracle.BUILD_FLAGS
    cam = oracle.make_camera(arg.camera.position, arg.camera.rotation, arg.camera.fovx, arg.camera.fovy,
                             arg.camera.znear, arg.camera.zfar, arg.camera.fov2view_ratio)
    w, h = viewport
    cu = oracle.camera_uniform(cam, w, h)
    rs = oracle.settings_uniform(oracle.make_aabb(gpc.aabb.min, gpc.aabb.max), gpc.center)
    if gpc.compressed:
        q = gpc.quantization
        oq = oracle.make_quantization({n: (getattr(q, n).zero_point, getattr(q, n).scale)
                                       for n in ("color_dc", "color_rest", "opacity", "scaling_factor")})

        def one_frame():
            splats, keys, _ = oracle.preprocess_compressed(gpc.gaussians, gpc.sh_coefs, gpc.covars, oq, gpc.sh_deg, cu, rs)
            _, order = oracle.sort_pairs(keys, np.arange(len(keys), dtype=np.uint32))
            oracle.render(splats, order, w, h, (0, 0, 0, 0), 0)
    else:
        def one_frame():
            oracle.render_frame(gpc.gaussians, gpc.sh_coefs, cu, rs, w, h)
    one_frame()  # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    frames = 0
    while True:
        one_frame()
        frames += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or frames >= 10:
            break
    return {"value": frames / dt, "unit": "frames/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"{frames} whole frames (view 0) of the same workload, 1 warm-up, OpenMP over "
                      f"{oracle.num_threads()} threads, oracle built {flags}"}


def _physical_cores(cpus):
    """The logical CPUs of `cpus` grouped by physical core, cores ordered by (package, core id): SMT siblings stay together,
    so that two ranks never share a core.  Falls back to one group per logical CPU when sysfs does not say."""
    groups = {}
    for c in sorted(cpus):
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            key = (int(open(base + "physical_package_id").read()), int(open(base + "core_id").read()))
        except (OSError, ValueError):
            key = (0, c)
        groups.setdefault(key, []).append(c)
    return [groups[k] for k in sorted(groups)]


def cpu_quota():
    """CPUs' worth of run time the container's cgroup grants per period (cgroup v2 cpu.max, v1 cfs quota), or None: a box
    may show 256 CPUs in its affinity mask and still be throttled to 16 (the one-GPU boxes of this build are)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 else None
    except (OSError, ValueError):
        return None


def _visible_device_ordinals():
    """HIP device index -> position among the node's GPUs (KFD order), honouring HIP/ROCR/CUDA_VISIBLE_DEVICES lists of
    plain integers; None when a variable holds something else (UUIDs)."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                return [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return None
    return None


def gpu_numa_nodes(topology_root="/sys/class/kfd/kfd/topology/nodes", pci_root="/sys/bus/pci/devices"):
    """NUMA node of every GPU of this host, in HIP's enumeration order (the KFD topology nodes that have SIMDs, by node id;
    a GPU's PCI address is `domain` + `location_id` of its properties file, its NUMA node /sys/bus/pci/devices/<bdf>/numa_node).
    -> list of ints (-1 = the platform does not say), [] when there is no KFD topology (no GPU / not Linux)."""
    out = []
    try:
        ids = sorted(int(d) for d in os.listdir(topology_root) if d.isdigit())
    except OSError:
        return out
    for i in ids:
        try:
            props = dict(ln.split()[:2] for ln in open(f"{topology_root}/{i}/properties") if len(ln.split()) >= 2)
        except OSError:   # (a node of another container's GPU: not readable, not ours)
            continue
        if int(props.get("simd_count", "0")) == 0:
            continue  # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        try:
            out.append(int(open(f"{pci_root}/{bdf}/numa_node").read()))
        except (OSError, ValueError):
            out.append(-1)
    return out


def _cpus_of_numa_node(node):
    try:
        txt = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
    except OSError:
        return None
    cpus = set()
    for part in txt.split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


def numa_share(local_rank, local_world, allowed, gpu_nodes=None, cpus_of_node=_cpus_of_numa_node, physical=None):
    """The logical CPUs rank `local_rank` of `local_world` takes: the ranks whose GPUs hang off the same NUMA node split THAT
    node's physical cores (of the allowed set) among themselves, in rank order; when the platform does not say where the GPUs
    are (or a node has fewer cores than ranks) the ranks split all allowed cores as before.  Pure function of sysfs, so every
    rank computes the same partition without talking to the others.  -> (sorted cpus, note)"""
    physical = physical or _physical_cores
    if gpu_nodes is None:
        gpu_nodes = gpu_numa_nodes()
        vis = _visible_device_ordinals()
        if vis is not None and gpu_nodes and all(0 <= v < len(gpu_nodes) for v in vis):
            gpu_nodes = [gpu_nodes[v] for v in vis]
    node_of = [gpu_nodes[r] if r < len(gpu_nodes) else -1 for r in range(local_world)]
    my_node = node_of[local_rank]
    if my_node >= 0 and all(n >= 0 for n in node_of):
        node_cpus = cpus_of_node(my_node)
        peers = [r for r in range(local_world) if node_of[r] == my_node]   # ranks that share this node, in rank order
        if node_cpus:
            cores = physical(sorted(set(allowed) & node_cpus))
            if len(cores) >= len(peers):
                j = peers.index(local_rank)
                lo, hi = j * len(cores) // len(peers), (j + 1) * len(cores) // len(peers)
                return (sorted(c for g in cores[lo:hi] for c in g),
                        f"NUMA node {my_node} (this rank's GPU), physical cores {lo}..{hi - 1} of the node's {len(cores)} "
                        f"shared by ranks {peers}")
    cores = physical(allowed)
    if len(cores) < local_world:
        return None, f"{len(cores)} cores for {local_world} ranks: not pinned"
    lo, hi = local_rank * len(cores) // local_world, (local_rank + 1) * len(cores) // local_world
    return (sorted(c for g in cores[lo:hi] for c in g),
            f"physical cores {lo}..{hi - 1} of {len(cores)} (GPU NUMA placement unknown: equal slices of all allowed cores)")


def pin_host_share(local_rank, local_world):
    """Eight ranks on one node share its host cores: each rank builds the scene with an OpenMP team, then runs ONE enqueue
    thread beside the HIP runtime's helper threads.  Left alone, every rank's team spans every core and the enqueue threads
    migrate between them.  Rank r takes its share of the PHYSICAL cores of the NUMA node its GPU hangs off (numa_share;
    taskset / cgroup respected, all SMT siblings included) and sizes its OpenMP team to the share; must run before torch /
    the library load (OMP_NUM_THREADS is read when the OpenMP runtime starts, the HIP runtime's helper threads inherit the
    affinity of the thread that initialises it).
    -> (cpus of the share, note)"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None, "affinity not available on this platform"
    if os.environ.get("WS_BENCH_PIN", "1") == "0":
        return allowed, "WS_BENCH_PIN=0: not pinned"
    quota = cpu_quota()
    team_cap = max(1, int(quota / max(local_world, 1))) if quota else None   # a team beyond the quota is only throttled
    if local_world <= 1:
        os.environ.setdefault("OMP_NUM_THREADS", str(min(len(allowed), team_cap) if team_cap else len(allowed)))
        return allowed, f"one rank: all {len(allowed)} allowed CPUs" + (f", cgroup quota {quota:g} CPUs" if quota else "")
    try:
        share, where = numa_share(local_rank, local_world, allowed)
    except Exception as e:  # noqa: BLE001  (an unexpected sysfs layout must never keep a rank from starting)
        share, where = None, f"placement lookup failed ({e!r}): not pinned"
    if not share:
        return allowed, where
    os.sched_setaffinity(0, share)
    try:  # (a smaller team asked for by the caller stays)
        want = min(int(os.environ.get("OMP_NUM_THREADS", len(share))), len(share))
    except ValueError:
        want = len(share)
    if team_cap:
        want = min(want, team_cap)
    os.environ["OMP_NUM_THREADS"] = str(max(want, 1))
    return share, (f"rank-local share: {where} ({len(share)} logical CPUs), OMP_NUM_THREADS={os.environ['OMP_NUM_THREADS']}"
                   + (f", cgroup quota {quota:g} CPUs" if quota else ""))


def thread_cpu_times():
    """{tid: (comm, cpu seconds)} of every thread of this process (/proc/self/task/*/stat: utime + stime)."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        tids = os.listdir("/proc/self/task")
    except OSError:
        return out
    for t in tids:
        try:
            st = open(f"/proc/self/task/{t}/stat").read()
        except OSError:
            continue
        try:
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(t)] = (comm, (int(f[11]) + int(f[12])) / tick)   # fields 14, 15 of proc(5): utime, stime
        except (ValueError, IndexError):
            continue
    return out


def thread_busy(before, after, elapsed, floor=0.02):
    """Threads that used more than `floor` cores between two thread_cpu_times() snapshots: [{"thread", "cores"}], busiest
    first (clock ticks are 10 ms: meaningful over regions of >= 0.1 s)."""
    rows = []
    for tid, (comm, t1) in after.items():
        d = t1 - before.get(tid, (comm, 0.0))[1]
        if elapsed > 0 and d / elapsed >= floor:
            rows.append({"thread": comm + (" (main)" if tid == os.getpid() else ""), "cores": round(d / elapsed, 3)})
    return sorted(rows, key=lambda r: -r["cores"])


def self_launch(argv, n):
    """`python bench.py --gpus N` from a plain shell (no RANK / WORLD_SIZE in the environment): become the launcher of N
    ranks, one per GPU -- exec torch.distributed.run with the same arguments (the process is REPLACED: same pid, same
    stdout, the caller's timeout still applies).  Rank 0 prints the one result line; the other ranks and the launcher write
    to stderr only."""
    env = dict(os.environ)
    if "OMP_NUM_THREADS" not in env:
        # torch.distributed.run would set it to 1 for every rank; pin_host_share() caps it to the rank's share anyway
        q = cpu_quota()
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        env["OMP_NUM_THREADS"] = str(max(1, int(min(q or ncpu, ncpu) / n)))
    env["WS_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks through torch.distributed.run on 127.0.0.1", file=sys.stderr)
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def init_distributed(a, torch):
    """One process per GPU.  world > 1: torch.distributed.run supplies RANK / WORLD_SIZE / MASTER_*.  world == 1: a
    one-rank communicator on 127.0.0.1 so that the RCCL initialisation, the barrier and the all-reduce below are the
    same calls the N-GPU run makes.  Returns (dist or None, rank, local_rank, world, note)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world == 1 and a.gpus > 1:   # (main() self-launches before it gets here)
        raise SystemExit("--gpus N > 1 needs N ranks: run `python bench.py --gpus N` from a shell without RANK / WORLD_SIZE set")
    backend = "gloo" if (a.dry_run or a.dist_backend == "gloo") else "nccl"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if backend == "nccl" and world > 1:
        # RCCL wants one device per rank: two ranks of a communicator on one device end in ncclInvalidUsage ("Duplicate GPU
        # detected") at best and in a hang inside the bootstrap at worst.  Say so BEFORE the rendezvous, by name, on every rank.
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        have = torch.cuda.device_count()
        if a.single_device or local_world > have:
            raise SystemExit(f"[bench] RCCL needs one device per rank: {local_world} local rank(s), {have} visible device(s)"
                             + (", --single-device" if a.single_device else "") + ".  To stand several ranks on one device "
                             "(shard determinism, host budget) use --single-device --dist-backend gloo; the collective of the "
                             "real run needs --gpus <= devices")
    if world == 1:
        if a.no_dist:
            return None, rank, local_rank, world, "disabled (--no-dist)"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    try:
        kw = {} if backend == "gloo" else {"device_id": torch.device("cuda", local_rank)}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    except Exception as e:  # noqa: BLE001
        if world > 1:
            raise
        return None, rank, local_rank, world, f"one-rank {backend} communicator failed to initialise: {e!r}"
    return dist, rank, local_rank, world, f"{backend} communicator, world size {world}"


class DryBatch:
    """--dry-run: stands where the ViewBatch stands; frames cost a fixed sleep instead of GPU time."""
    texel_bytes = 16

    def __init__(self, per_frame_s):
        self.per_frame_s = per_frame_s
        self.frames = 0

    def render(self, pc, views, ptrs, pitch):
        self.frames += len(views)
        time.sleep(self.per_frame_s * len(views))

    def errors(self, reset=False):
        return 0

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="hd1m", choices=WORKLOADS)
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--format", default="rgba32float")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("WS_BENCH_STREAMS", "4")),
                    help="frames in flight per GPU: one renderer (private scratch) + one HIP stream each "
                         "(the views of a batch are independent; 1 = strictly one frame at a time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="the default run (hd1m, one GPU) also times c3 -- the largest single-GPU configuration, where the "
                         "HBM fractions of the sort and K1 mean something -- for ~200 frames and reports it under "
                         "\"secondary\"; this switch leaves it out")
    ap.add_argument("--no-dist", action="store_true", help="skip the one-rank RCCL communicator at N = 1")
    ap.add_argument("--no-inflight", action="store_true", help="skip the device-side in-flight trace behind the timed region (the 'inflight' block)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo: the collectives run on host tensors (tests: several ranks sharing ONE GPU, where RCCL refuses "
                         "two ranks on a device); the rendering path is unchanged")
    ap.add_argument("--single-device", action="store_true",
                    help="every rank renders on cuda:0 (tests of the N-rank path on a one-GPU box)")
    ap.add_argument("--check-shard-determinism", action="store_true",
                    help="after the timed region every rank also renders the first view of EVERY rank's shard; rank 0 "
                         "compares the bytes (a view's image must not depend on the rank that draws it)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU only (backend gloo): sharding, planning, barrier and reduction without rendering")
    ap.add_argument("--host-wait", default=os.environ.get("WS_BENCH_HOST_WAIT", "block"), choices=("block", "spin"),
                    help="how the host waits for the device at the end of the timed region: block (interrupt-driven, the "
                         "analogue of the reference's device.poll(Wait), bin/measure.rs:147; default) or spin (the HIP "
                         "runtime's default: a core busy for as long as the wait lasts)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(sys.argv[1:], a.gpus)   # does not return

    # stdout carries exactly ONE line, the result: everything else that writes to file descriptor 1 -- RCCL prints a
    # version banner through C stdio, flushed at exit -- is sent to stderr for the life of the process.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    host_cpus, host_note = pin_host_share(int(os.environ.get("LOCAL_RANK", "0")), local_world)
    # Small scenes are enqueued by one thread PER SLOT (ws_view_batch: 2.3-2.7x the frames/s on c1 for ~4 host cores).  Eight
    # ranks doing that on a 16-CPU quota throttle each other: a rank whose share of the quota is below one core per slot + one
    # keeps the single submission thread (the library reads the switch when the context is created).
    share = (cpu_quota() or float(len(host_cpus or [1]))) / max(local_world, 1)
    if local_world > 1 and share < a.streams + 1:
        os.environ.setdefault("WS_BATCH_THREADS", "0")
    import torch
    dist, rank, local_rank, world, dist_note = init_distributed(a, torch)
    if a.single_device:
        local_rank = 0
    dev = "cpu" if (a.dry_run or a.dist_backend == "gloo") else "cuda"   # where the reduced tensors live
    if not a.dry_run:
        torch.cuda.set_device(local_rank)

    import websplat as ws  # raises if the HIP library is not built: there is no fallback
    gpc, views, viewport, workload_note = build_workload(ws, a.workload, a.views)
    w, h = viewport
    nstreams = max(1, a.streams)
    from websplat.shard import views_for_rank
    view_ids = views_for_rank(len(views), rank, world)
    my_views = [views[i] for i in view_ids] or views[:1]
    packed_all = ws.ViewBatch.pack_views(my_views)
    if a.dry_run:
        ctx = pc = r = None
        batch = DryBatch(2e-4)
        target_ptrs = [0x1000 * (k + 1) for k in range(nstreams)]
        pitch = w * 16
    else:
        ctx = ws.Context(local_rank)
        ctx.set_host_wait(a.host_wait)
        pc = ws.PointCloud(ctx, gpc)
        tdtype = {"rgba32float": torch.float32, "rgba16float": torch.float16, "rgba8unorm": torch.uint8}[a.format]
        # A view batch (ws_view_batch_*): one renderer (private scratch) + one HIP stream per frame in flight, the
        # scene is shared; one output image per slot (frame g of the batch's life runs on slot g % frames_in_flight).
        batch = ws.ViewBatch(ctx, a.format, gpc.sh_deg, gpc.compressed, nstreams)
        targets = [torch.empty((h, w, 4), dtype=tdtype, device="cuda") for _ in range(nstreams)]
        target_ptrs = [t.data_ptr() for t in targets]
        pitch = w * batch.texel_bytes
    planned = [0]  # frames planned so far: frame g of the batch's life lands on slot g % frames_in_flight

    def plan(first_view, count):
        """Argument arrays for frames first_view .. first_view + count - 1 (views cycle through this rank's shard),
        built OUTSIDE any timed region; submit() enqueues them with one call into the library."""
        import ctypes as C
        arr = (type(packed_all[0]) * count)()
        ptrs = (C.c_void_p * count)()
        for j in range(count):
            arr[j] = packed_all[(first_view + j) % len(my_views)]
            ptrs[j] = target_ptrs[(planned[0] + j) % nstreams]
        planned[0] += count
        return arr, ptrs

    def submit(p):
        batch.render(pc, p[0], p[1], pitch)  # enqueues every frame of the plan and returns

    def device_sync():
        if not a.dry_run:
            torch.cuda.synchronize()

    # The barrier that brackets the timed region is a HOST barrier (a gloo group next to the RCCL communicator): an RCCL
    # barrier is a kernel launch plus a proxy round trip -- measured 0.3-0.5 ms on one rank, 10-15 % of the 3 ms a 20-step
    # timed region lasts -- and it is measurement overhead, not frames.  RCCL still carries the collective of the run
    # (the MAX all-reduce below) and one barrier right here, outside the timed region.
    host_pg = None
    if dist is not None and dist.get_backend() == "nccl":
        dist.barrier()
        if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: no hostname lookup (it may not resolve in a container)
        try:
            host_pg = dist.new_group(backend="gloo")
        except Exception as e:  # noqa: BLE001  (no usable interface for gloo: keep the RCCL barrier)
            print(f"[bench] gloo group for the host barrier failed ({e!r}); using the RCCL barrier", file=sys.stderr)
        # every rank must bracket the timed region with the SAME collective: the host group is used only if every rank has it
        ok = torch.tensor([1.0 if host_pg is not None else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() < 1.0:
            host_pg = None

    bar_t = {}

    def barrier():
        ta = time.perf_counter()
        device_sync()
        tb = time.perf_counter()
        if dist is not None:
            if host_pg is not None:
                dist.barrier(group=host_pg)
            else:
                dist.barrier()
            tc = time.perf_counter()
            device_sync()
            bar_t.update(sync1=tb - ta, host_barrier=tc - tb, sync2=time.perf_counter() - tc)
        else:
            bar_t.update(sync1=tb - ta, host_barrier=0.0, sync2=0.0)

    if not a.dry_run:
        # one more renderer on the default stream: the one-frame-at-a-time rate and the per-kernel times (rank 0)
        r = ws.GaussianRenderer(ctx, a.format, gpc.sh_deg, gpc.compressed)
        target1 = targets[0]

        def frame(i):
            r.prepare(pc, my_views[i % len(my_views)])
            r.render(pc, target_ptr=target1.data_ptr())

        # Building the synthetic scene keeps the host busy for seconds while the GPU idles at low clocks: bring it to
        # its steady-state clock with ~0.3 s of (untimed, uncounted) frames before the W warm-up steps.
        t_pre = time.perf_counter()
        i_pre = 0
        while time.perf_counter() - t_pre < 0.3:
            submit(plan(i_pre, 16))
            i_pre += 16
            torch.cuda.synchronize()
        # A scene that needs more (tile, splat) entries than the automatic capacity: the overflowed warm-up frames left their
        # demand in the renderers' sticky words; reading them here makes the next prepare() allocate it (never in the timed
        # region, whose frames are checked below and invalidate the run if they drop anything).
        for _ in range(3):
            if not (batch.errors(reset=True) & 1):
                break
            submit(plan(i_pre, 2 * nstreams))
            i_pre += 2 * nstreams
            torch.cuda.synchronize()
        frame(0)
        if r.errors(reset=True)[0] & 1:
            frame(0)
    if a.warmup:
        submit(plan(0, a.warmup))
    timed = plan(a.warmup, a.steps)
    # The timed region.  Opening bracket: every rank's device idle, then the host barrier, then device synchronize: all ranks
    # start together.  Closing bracket: THIS rank's device synchronize, the clock, then the host barrier.  A rank's interval
    # ends when its own K frames are done; the job's time is the MAX of the intervals over the ranks (the all-reduce below),
    # which is what a closing barrier in front of the clock would measure too -- plus the barrier itself: a gloo barrier is a
    # TCP round trip, measured at 0.15-0.33 ms on ONE rank (profiles/r04/short_run_probe.txt), 5-10 % of the 3-ms region of
    # `--steps 20`, and it is not frames.
    barrier()
    if os.environ.get("WS_BENCH_MARK_FILE"):   # scripts/host_thread_probe.sh: "the timed region starts now" (pid inside)
        with open(os.environ["WS_BENCH_MARK_FILE"], "w") as f:
            f.write(str(os.getpid()))
    thr0 = thread_cpu_times()
    cpu0 = time.process_time()   # CPU time of ALL threads of this process (enqueue thread + the HIP runtime's helpers)
    tcpu0 = time.thread_time()
    t0 = time.perf_counter()
    submit(timed)   # exactly K frames, enqueued back to back, one sync at the end
    # what the submitting thread SPENT (its CPU time): the call itself also waits -- asleep -- whenever a slot already has its
    # three frames queued (the view batch bounds the host's run-ahead), so its wall time says nothing about the host's load
    t_enq = time.thread_time() - tcpu0
    device_sync()
    elapsed = time.perf_counter() - t0
    cpu_busy = (time.process_time() - cpu0) / max(elapsed, 1e-9)   # host cores this rank kept busy during the timed region
    host_threads = thread_busy(thr0, thread_cpu_times(), elapsed)   # ... and which threads they were (rank 0's; >= 0.1 s regions)
    barrier()
    if os.environ.get("WS_BENCH_DEBUG"):
        print(f"[bench debug] enqueue {t_enq * 1e3:.3f} ms, total {elapsed * 1e3:.3f} ms for {a.steps} frames; closing bracket: "
              f"(outside the interval) host barrier {bar_t['host_barrier'] * 1e3:.3f} ms, second sync "
              f"{bar_t['sync2'] * 1e3:.3f} ms", file=sys.stderr)
    # every frame of the timed region (and of the warm-up) must have been drawn completely: the slots' sticky error
    # words collect tile-entry overflow and look-back time-outs of ALL frames since the batch was created
    err_bits = batch.errors()
    if dist is not None:
        t = torch.tensor([elapsed, float(err_bits), t_enq, cpu_busy], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # RCCL: the only collective, off the data path
        elapsed, err_bits, t_enq, cpu_busy = float(t[0].item()), int(t[1].item()), float(t[2].item()), float(t[3].item())
    if err_bits:
        raise SystemExit(f"[bench] INVALID RUN: device-side error bits 0x{err_bits:x} in the timed frames (bit 0 = tile-entry "
                         "list overflow: entries were dropped; bits 1-3 = look-back time-out) -- no result line is printed")

    out = None
    if rank == 0:
        n = gpc.num_points
        fps = world * a.steps / elapsed
        out = {
            "metric": "frames_per_sec", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {n} Gaussians (sh_deg {gpc.sh_deg}), {w}x{h}, {a.format} target, "
                                   f"{len(views)} views sharded view i -> rank i mod N, {nstreams} frame(s) in flight per GPU",
                       "gaussians": n, "width": w, "height": h, "views": len(views), "frames_in_flight": nstreams,
                       "error_bits": err_bits, "collective": dist_note,
                       # the host side of the timed region (MAX over ranks): CPU time the submitting thread spent on the K frames;
                       # host_bound = some host thread was busy > 80 % of the region: the host, not the GPU, may have set the pace
                       "host_enqueue_ms_per_frame": t_enq / a.steps * 1e3,
                       "host_bound": bool(max([t_enq / elapsed] + [t["cores"] for t in (host_threads if elapsed >= 0.1 else [])]) > 0.8),
                       "host_cpus": len(host_cpus) if host_cpus else None, "host_affinity": host_note,
                       # cores one rank keeps busy while it renders (MAX over ranks) and what the container grants in total:
                       # world x busy above the quota means the ranks throttle each other, whatever the core count says
                       # (CPU time comes in 10-ms clock ticks: below 0.1 s of timed region the ratio is noise -- null, like host_threads)
                       "host_cores_busy_per_rank": cpu_busy if elapsed >= 0.1 else None, "host_cpu_quota": cpu_quota(),
                       "host_threads": host_threads if elapsed >= 0.1 else None, "host_wait": a.host_wait,
                       "frame_submission": os.environ.get("WS_GRAPH", "0") not in ("", "0") and "graph replay (WS_GRAPH)" or "launch by launch",
                       "timing_barrier": ("none (one process, --no-dist): device synchronize on both sides" if dist is None else
                                          ("host barrier (gloo group)" if host_pg is not None else f"{dist.get_backend()} barrier")
                                          + " + device synchronize in front of the region; behind it device synchronize, the "
                                            "clock, then the barrier; MAX over ranks of the per-rank interval")},
        }
        if workload_note:
            out["config"]["workload_note"] = workload_note
            if workload_note.startswith("real scene"):
                out["data"] = "real scene file (" + workload_note + ")"
        if a.dry_run:
            out["dry_run"] = True
            out["config"]["rank0_views"] = view_ids
    if a.check_shard_determinism and not a.dry_run:
        # SURVEY 7 view_shard_determinism: the first view of every rank's shard, drawn by EVERY rank on its own renderer
        # scratch; the digests meet on rank 0 (off the timed path)
        import hashlib
        probe = [views_for_rank(len(views), q, world)[0] for q in range(world) if views_for_rank(len(views), q, world)]
        digests = []
        for vi in probe:
            r.prepare(pc, views[vi])
            r.render(pc, target_ptr=targets[0].data_ptr())
            torch.cuda.synchronize()
            digests.append(hashlib.sha256(targets[0].cpu().numpy().tobytes()).hexdigest())
        got = [None] * world
        if dist is not None:
            dist.all_gather_object(got, digests)
        else:
            got = [digests]
        if out is not None:
            out["config"]["shard_determinism"] = {"views": probe, "ranks": world,
                                                  "identical": all(g == got[0] for g in got),
                                                  "distinct_images": len(set(got[0]))}
    if rank == 0 and not a.dry_run:
        analyse(a, ws, ctx, pc, gpc, r, frame, my_views, views, viewport, out, world)
        if nstreams > 1 and not a.no_inflight:
            try:
                out["inflight"] = inflight_trace(ws, batch, pc, plan, submit, device_sync, nstreams, min(max(a.steps, 8 * nstreams), 400))
            except Exception as e:  # noqa: BLE001  (analysis only: never costs the run its result line)
                out["inflight"] = {"error": repr(e)}
        if world == 1 and a.workload == "hd1m" and not a.no_secondary:
            out["secondary"] = {"c3": secondary(a, ws, ctx, torch, "c3", nstreams)}
    barrier()
    if dist is not None and a.dry_run:
        got = [None] * world
        dist.all_gather_object(got, view_ids)   # verification only (tests): every rank's shard
        # ... and every rank's share of the host as pin_host_share() really set it on THIS machine's sysfs (verdict r05 item 8a):
        # the ranks of a node must not overlap, and their OpenMP teams together must fit the cgroup's CPU quota
        try:
            mine = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            mine = None
        shares = [None] * world
        dist.all_gather_object(shares, {"rank": rank, "cpus": mine, "omp_num_threads": int(os.environ.get("OMP_NUM_THREADS", "0") or 0),
                                        "pinned": bool(host_cpus) and "not pinned" not in host_note, "note": host_note})
        if out is not None:
            out["config"]["rank_views"] = got
            pinned = [sh for sh in shares if sh["pinned"] and sh["cpus"]]
            overlap = [(x["rank"], y["rank"]) for i, x in enumerate(pinned) for y in pinned[i + 1:] if set(x["cpus"]) & set(y["cpus"])]
            quota = cpu_quota()
            omp_sum = sum(sh["omp_num_threads"] for sh in shares)
            # (a team is never smaller than one thread: with more ranks than quota'd CPUs the floor is one thread per rank)
            budget = max(quota, float(world)) if quota else None
            out["config"]["host_partition"] = {
                "ranks": [{"rank": sh["rank"], "logical_cpus": len(sh["cpus"]) if sh["cpus"] else None, "first_cpu": sh["cpus"][0] if sh["cpus"] else None,
                           "omp_num_threads": sh["omp_num_threads"], "pinned": sh["pinned"], "note": sh["note"]} for sh in shares],
                "pinned_ranks": len(pinned), "disjoint": not overlap, "overlapping_pairs": overlap,
                "omp_threads_sum": omp_sum, "cpu_quota": quota, "fits_quota": (omp_sum <= budget) if budget else None}
            if overlap or (budget and omp_sum > budget):
                raise SystemExit(f"[bench] host partition INVALID: overlapping rank shares {overlap}, sum of OMP teams {omp_sum} "
                                 f"against a quota of {quota}")
    if not a.dry_run:
        r.close()
        batch.close()
        pc.close()
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if out is not None:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)


def inflight_trace(ws, batch, pc, plan, submit, device_sync, nstreams, frames):
    """What runs beside what with frames in flight, measured ON THE DEVICE behind the timed region (rocprofv3 serialises the hardware
    queues: DESIGN 3.5): K1 and the compositing kernel of `frames` more frames of the same view batch stamp {first workgroup start, last
    workgroup end} on the 100-MHz device clock (ws_renderer_enable_frame_trace; costs ~5 % while it is on, which is why it is not inside
    the timed region).  -> us per frame with >= 2 blends running / one blend beside K1 / ... and the in-flight durations."""
    import numpy as np
    per_slot = (frames + nstreams - 1) // nstreams
    rs = [batch.renderer(s_) for s_ in range(nstreams)]
    device_sync()
    for r_ in rs:
        r_.enable_frame_trace(per_slot)
    t0 = time.perf_counter()
    submit(plan(0, frames))
    device_sync()
    elapsed = time.perf_counter() - t0
    tr = [r_.frame_trace().astype(np.float64) / 100.0 for r_ in rs]   # us
    for r_ in rs:
        r_.enable_frame_trace(0)
    k1, bl = [], []
    for t in tr:
        n = len(t)
        lo, hi = n // 10, n - n // 10            # steady state: without the fill and the drain
        k1 += [(x[0], x[1]) for x in t[lo:hi]]
        bl += [(x[2], x[3]) for x in t[lo:hi]]
    a0 = max(min(s_ for s_, _ in k1), min(s_ for s_, _ in bl))
    b0 = min(max(e for _, e in k1), max(e for _, e in bl))
    pts = [(s_, 0, 1) for s_, _ in k1] + [(e, 0, -1) for _, e in k1] + [(s_, 1, 1) for s_, _ in bl] + [(e, 1, -1) for _, e in bl]
    pts.sort()
    n_run, last, acc = [0, 0], a0, {}
    for t, which, d in pts:
        tt = min(max(t, a0), b0)
        if tt > last:
            key = (min(n_run[0], 2), min(n_run[1], 2))
            acc[key] = acc.get(key, 0.0) + (tt - last)
            last = tt
        n_run[which] += d
    nframes = max(sum(1 for s_, e in k1 if a0 <= s_ and e <= b0), 1)

    def share(pred):
        return sum(v for k, v in acc.items() if pred(*k)) / nframes
    return {"frames": frames, "frames_per_s_while_traced": frames / elapsed, "us_per_frame": (b0 - a0) / nframes,
            "k1_in_flight_us": float(np.mean([e - s_ for s_, e in k1])), "blend_in_flight_us": float(np.mean([e - s_ for s_, e in bl])),
            "us_per_frame_with": {"no K1, no blend": share(lambda k, b: k == 0 and b == 0), "K1 only": share(lambda k, b: k >= 1 and b == 0),
                                  "one blend, no K1": share(lambda k, b: k == 0 and b == 1), "one blend beside K1": share(lambda k, b: k >= 1 and b == 1),
                                  "two or more blends": share(lambda k, b: b >= 2)},
            "note": "device-side stamps of K1 and the compositing kernel (100-MHz clock, first workgroup start / last workgroup end), "
                    "steady-state frames of a traced batch behind the timed region; with frames in flight the frame period is the sum of the "
                    "LONE durations of the chip-filling kernels (DESIGN 3.5)"}


def secondary(a, ws, ctx, torch, workload, nstreams, frames=200):
    """The same measurement on a second workload, rank 0 of a one-GPU run only, after the headline's timed region: `frames`
    frames in flight between two device synchronisations, then analyse() -- one frame at a time, per-kernel event times.
    -> the compact block of the result line's "secondary" (the full kernel table of that workload: --workload c3)."""
    import argparse
    gpc, views, viewport, _ = build_workload(ws, workload, 8)
    w, h = viewport
    pc = ws.PointCloud(ctx, gpc)
    batch = ws.ViewBatch(ctx, a.format, gpc.sh_deg, gpc.compressed, nstreams)
    tdtype = {"rgba32float": torch.float32, "rgba16float": torch.float16, "rgba8unorm": torch.uint8}[a.format]
    targets = [torch.empty((h, w, 4), dtype=tdtype, device="cuda") for _ in range(nstreams)]
    packed = ws.ViewBatch.pack_views(views)
    import ctypes as C

    def plan(first, count):
        arr = (type(packed[0]) * count)()
        ptrs = (C.c_void_p * count)()
        for j in range(count):
            arr[j] = packed[(first + j) % len(views)]
            ptrs[j] = targets[(first + j) % nstreams].data_ptr()
        return arr, ptrs
    pitch = w * batch.texel_bytes
    warm, timed = plan(0, 4 * nstreams), plan(4 * nstreams, frames)
    batch.render(pc, warm[0], warm[1], pitch)
    torch.cuda.synchronize()
    tcpu0 = time.thread_time()
    t0 = time.perf_counter()
    batch.render(pc, timed[0], timed[1], pitch)
    t_enq = time.thread_time() - tcpu0
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    err = batch.errors()
    r = ws.GaussianRenderer(ctx, a.format, gpc.sh_deg, gpc.compressed)

    def frame(i):
        r.prepare(pc, views[i % len(views)])
        r.render(pc, target_ptr=targets[0].data_ptr())
    sub = {"config": {}}
    a2 = argparse.Namespace(**vars(a))
    a2.workload, a2.steps, a2.no_cpu_baseline = workload, frames, True
    analyse(a2, ws, ctx, pc, gpc, r, frame, views, views, viewport, sub, 1)
    r.close()
    batch.close()
    pc.close()
    if err:
        return {"error_bits": err}
    k = sub["kernels"]
    V = sub["config"]["avg_visible"]
    # the depth sort and K1 as STAGES (one event pair around the stage's launches, analyse()): twelve launches at the
    # resolution limit of per-launch events do not add up to the stage
    depth_ms = sub["stages"]["sorting"]["ms"]
    k1_ms = sub["stages"]["preprocess"]["ms"]
    k1 = k.get("k_preprocess") or k.get("k_preprocess<compressed>") or {}
    rf = sub["roofline"]
    return {"workload": f"{workload}: {gpc.num_points} Gaussians, {w}x{h}, {nstreams} frame(s) in flight", "frames": frames,
            "value": frames / elapsed, "unit": "frames/s", "ms_per_step": elapsed / frames * 1e3,
            "single_stream_fps": sub["config"]["single_stream_fps"], "host_enqueue_ms_per_frame": t_enq / frames * 1e3,
            "binning_tile": sub["config"]["binning_tile"], "avg_visible": V, "avg_tile_entries": sub["config"]["avg_tile_entries"],
            "roofline": {"kernel": rf["kernel"], "bound": rf["bound"], "frac": rf["frac"], "achieved": rf["achieved"], "peak": rf["peak"],
                         "unit": rf["unit"], "hbm_frac": rf["hbm_frac"], "hbm_achieved": rf["hbm_achieved"],
                         "frac_at_measured_issue_rate": rf.get("frac_at_measured_issue_rate"),
                         "alg_bytes": rf["alg_bytes_per_launch"], "avg_launch_ms": rf["avg_launch_ms"], "traffic": rf["traffic"],
                         # the whole frame against the HBM roofline, at THIS block's frame period (frames in flight)
                         "frame": {"alg_bytes_per_frame": rf["frame"]["alg_bytes_per_frame"], "ms_per_step": elapsed / frames * 1e3,
                                   "achieved": rf["frame"]["alg_bytes_per_frame"] / (elapsed / frames) / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": rf["frame"]["alg_bytes_per_frame"] / (elapsed / frames) / 1e9 / HBM_PEAK_GBS}},
            "kernels": {
                # 68 V: four passes x (8 B in + 8 B out) + one more key read (the reference's sorter shape)
                "depth sort": {"launches_per_frame": sum(v["launches_per_frame"] for lbl, v in k.items() if lbl.startswith("depth:")),
                               "ms_per_frame": depth_ms, "alg_bytes": 68 * V,
                               "GBps": (68 * V / (depth_ms * 1e-3) / 1e9) if depth_ms else None,
                               "frac": (68 * V / (depth_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if depth_ms else None},
                "K1": {"ms_per_frame": k1_ms, "alg_bytes": k1.get("alg_bytes_per_launch"),
                       "GBps": (k1["alg_bytes_per_launch"] / (k1_ms * 1e-3) / 1e9) if (k1.get("alg_bytes_per_launch") and k1_ms) else None,
                       "frac": (k1["alg_bytes_per_launch"] / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                               if (k1.get("alg_bytes_per_launch") and k1_ms) else None}},
            "note": "roofline: the dominant kernel, per-launch event interval minus the empty-launch interval (as the headline's "
                    "roofline block); `kernels`: whole STAGES between one event pair each, one frame in flight, nothing "
                    "subtracted (a lower bound on the rate); the rocprofv3 durations of the same workload are under profiles/"}


def analyse(a, ws, ctx, pc, gpc, r, frame, my_views, views, viewport, out, world):
    """rank 0, after the timed region: single-stream rate, then per-kernel time (HIP events on the launch stream)
    for the roofline, frame statistics and the CPU baseline."""
    import torch
    w, h = viewport
    torch.cuda.synchronize()
    ks = max(20, a.steps // 2)  # from here on: one frame at a time on renderer `r`, default stream
    for i in range(5):
        frame(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(ks):
        frame(i)
    torch.cuda.synchronize()
    single_stream_fps = ks / (time.perf_counter() - t1)

    reps = min(len(my_views), 16)
    # Stage times first, with STAGE-level events only (the reference's GPUStopwatch labels, utils.rs:26-134: one event pair
    # per stage, the launches inside a stage back to back): a stage of twelve small launches is measured as it runs, not
    # as twelve event-bracketed launches.
    r.enable_timers(1)
    stage_acc = {"preprocess": 0.0, "sorting": 0.0, "binning": 0.0, "rasterization": 0.0}
    frame(0)
    for i in range(reps):
        frame(i)
        st = r.stage_times()
        for k in stage_acc:
            stage_acc[k] += st[k] / reps
    r.enable_timers(2)  # HIP event pairs around every kernel launch, on the launch stream
    per_kernel = {}      # label -> [total ms over reps, launches over reps]
    stat_acc = {"num_visible": 0, "num_tile_entries": 0}
    frame(0)
    r.kernel_times()
    for i in range(reps):
        frame(i)
        for label, ms in r.kernel_times():
            e = per_kernel.setdefault(label, [0.0, 0])
            e[0] += ms
            e[1] += 1
        fs = r.frame_stats()
        for k in stat_acc:
            stat_acc[k] += fs[k] / reps
    single_err, _ = r.errors()
    r.enable_timers(0)
    if single_err:
        raise SystemExit(f"[bench] INVALID RUN: device-side error bits 0x{single_err:x} on the analysis renderer")
    n = gpc.num_points
    V, D = stat_acc["num_visible"], stat_acc["num_tile_entries"]
    tile_w, tile_h = ctx.tile_size()
    tile_bits = 8
    while (1 << tile_bits) < -(-w // tile_w) * -(-h // tile_h):
        tile_bits += 8
    tile_passes = tile_bits // 8
    # ALGORITHMIC bytes per launch (DESIGN.md "Roofline accounting"): N Gaussians, V visible,
    # D (tile, splat) entries.  A radix pass over M pairs reads and writes 8 B per pair: 16*M; the histogram
    # kernels read 4*M; the last tile-id pass does not write the keys (12*D).
    alg = {
        # K1c: 24 B record for all, 12 B covariance + 3*(deg+1)^2 B SH gathers and 28 B out for survivors
        "k_preprocess": n * 124 + V * 28,
        "k_preprocess<compressed>": n * 24 + V * (12 + 3 * (gpc.sh_deg + 1) ** 2) + V * 28,
        "depth:k_sort_tile_hist": 4 * V, "depth:k_sort_col_scan": None, "depth:k_sort_scatter": 16 * V,
        "depth:k_sort_hist": 4 * V,
        # fat-tile one-sweep (round 4): a pass reads and writes key + index + rectangle once; the single-launch form does
        # the histogram read and all four passes in one launch
        "depth:k_dsort_fat": 24 * V, "depth:k_dsort_fat_coop": 4 * 24 * V + 4 * V,
        "k_bin_prefix": V * (4 + 4),
        "k_bin_emit": V * 12 + D * 8,
        "tiles:k_sort_tile_hist": 4 * D, "tiles:k_sort_col_scan": None, "tiles:k_sort_hist": 4 * D,
        "tiles:k_sort_scatter": (16 * D * (tile_passes - 1) + 12 * D) / tile_passes,   # (re-priced below by launch count)
        "k_blend": D * 24 + w * h * 16,
    }
    # An event interval = event + dispatch overhead of a dependent launch + the kernel; rocprofv3 reports the kernel
    # alone.  The library records one EMPTY launch per frame in the same way (after K1): its interval is subtracted.
    # What an interval carries on top of the kernel is mostly the write-back of the PREVIOUS kernel's dirty L2 lines,
    # so it varies (6 us after a memset, 12 us after K1's 30 MB of stores -- about what precedes the blend, for
    # which the calibrated time matches rocprofv3 to 1 %); GBps_interval (nothing subtracted) is the lower bound.
    # Launches shorter than twice the empty interval are below what events can resolve: no calibrated duration
    # for them (their rocprofv3 durations are in profiles/).
    # the tile-id sort is ONE counting pass up to 2048 binning tiles (no histogram launch, one scatter that reads
    # (id, value) and writes the value), digit passes beyond: price the launches that actually ran
    sc = per_kernel.get("tiles:k_sort_scatter")
    if sc:
        tile_passes = max(1, round(sc[1] / reps))
        alg["tiles:k_sort_scatter"] = (16 * D * (tile_passes - 1) + 12 * D) / tile_passes
    empty = per_kernel.pop("_empty_launch", None)
    empty_ms = (empty[0] / empty[1]) if empty else 0.0
    kernels = {}
    for label, (tot, cnt) in per_kernel.items():
        launches = cnt / reps
        interval_ms = tot / cnt
        avg_ms = (interval_ms - empty_ms) if interval_ms >= 2.0 * empty_ms else None
        ab = alg.get(label)
        kernels[label] = {"launches_per_frame": launches, "avg_launch_ms": avg_ms, "event_interval_ms": interval_ms,
                          "ms_per_frame": max(interval_ms - empty_ms, 0.0) * launches, "alg_bytes_per_launch": ab,
                          "GBps": (ab / (avg_ms * 1e-3) / 1e9) if (ab and avg_ms) else None,
                          # lower bound: the whole event interval charged to the kernel
                          "GBps_interval": (ab / (interval_ms * 1e-3) / 1e9) if ab else None}
    # dominant kernel = the launch label with the most GPU time per frame (the same kernel symbols rocprofv3
    # --stats lists: the depth sort and the tile-id sort instantiate k_sort_scatter with different tile sizes)
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_frame"])
    dk = kernels[dom]
    traffic = traffic_detail = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{a.workload}.json")
    if os.path.exists(tpath):  # PMC pass of the same command (scripts/pmc_traffic.py), bytes per launch
        tj = json.load(open(tpath))
        traffic = tj.get(dom)
        # raw FETCH_SIZE / WRITE_SIZE, the kernel's access class and the calibrated read factor applied to it
        traffic_detail = (tj.get("_detail") or {}).get(dom)
    valu = None
    vpath = os.path.join(ROOT, "profiles", f"valu_{a.workload}.json")
    if os.path.exists(vpath):  # PMC pass of the same command (scripts/pmc_valu.py): VALU issue accounting per launch
        valu = json.load(open(vpath)).get(dom)
    hbm_frac = (dk["GBps"] / HBM_PEAK_GBS) if dk["GBps"] else None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": dk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm_frac, "traffic": traffic, "traffic_detail": traffic_detail,
                "hbm_achieved": dk["GBps"], "hbm_frac": hbm_frac,
                "alg_bytes_per_launch": dk["alg_bytes_per_launch"], "avg_launch_ms": dk["avg_launch_ms"],
                "launches_per_frame": dk["launches_per_frame"],
                "event_interval_ms": dk["event_interval_ms"], "empty_launch_interval_ms": empty_ms,
                "valu": valu}
    # The BINDING resource (verdict r05 item 2).  k_blend moves few bytes per (pixel, splat) pair: what it runs out of is the
    # chip's wave64 VALU issue slots.  Then `bound` = "valu", `achieved` = wave64 VALU instructions per second (SQ_INSTS_VALU of
    # the PMC pass of the same command, profiles/valu_<workload>.json, / the launch duration measured HERE), `peak` = what 1024
    # SIMDs issue at VALU_CYCLES_PER_INST cycles per instruction -- 2 by the microarchitecture guide (a wave64 op over a
    # 32-lane-wide SIMD datapath), 2.4 by scripts/ubench/valu_rate.hip on this part: both fractions are on the record, `frac`
    # is the guide's -- and the HBM figure stays beside it as hbm_frac.  Every other kernel is priced against HBM.
    if dom == "k_blend" and valu and valu.get("valu_insts_per_launch") and valu.get("kernel_cycles") and dk["avg_launch_ms"]:
        insts, cyc = float(valu["valu_insts_per_launch"]), float(valu["kernel_cycles"])
        frac_guide = insts * VALU_CYCLES_PER_INST / (NUM_SIMDS * cyc)
        ach = insts / (dk["avg_launch_ms"] * 1e-3) / 1e9
        roofline.update({"bound": "valu", "achieved": ach, "peak": ach / frac_guide, "unit": "G wave64-inst/s", "frac": frac_guide,
                         "valu_cycles_per_inst": VALU_CYCLES_PER_INST,
                         "frac_at_measured_issue_rate": insts * VALU_CYCLES_PER_INST_MEASURED / (NUM_SIMDS * cyc),
                         "valu_cycles_per_inst_measured": VALU_CYCLES_PER_INST_MEASURED})
    # the driver-timed number's own roofline: algorithmic bytes of ALL launches of a frame / the frame period of the timed region
    frame_bytes = sum((k["alg_bytes_per_launch"] or 0.0) * k["launches_per_frame"] for k in kernels.values())
    frame_gbps = frame_bytes / (out["ms_per_step"] * 1e-3) / 1e9 if out.get("ms_per_step") else None
    roofline["frame"] = {"bound": "hbm", "alg_bytes_per_frame": frame_bytes, "ms_per_step": out.get("ms_per_step"),
                         "achieved": frame_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (frame_gbps / HBM_PEAK_GBS) if frame_gbps else None,
                         "note": "sum over the frame's launches of their algorithmic bytes (the kernel table) / ms_per_step of the "
                                 "timed region (frames in flight): how much of the HBM roofline the whole job uses"}
    roofline["note"] = ("dominant = most GPU time per frame summed over its launches; HIP events around every launch on the launch "
                        "stream, one frame in flight; avg_launch_ms = event interval minus the interval of an empty launch recorded the "
                        "same way in every frame (dispatch latency of a dependent launch, which rocprofv3 kernel durations do not "
                        "contain).  No stage is a dense contraction, so MFMA is unused.  `bound` names the resource the dominant kernel "
                        "runs out of: \"valu\" for k_blend (wave64 VALU issue: frac = SQ_INSTS_VALU x cycles per instruction / (1024 SIMDs "
                        "x kernel cycles)), \"hbm\" otherwise; hbm_frac is always the algorithmic bytes against 8 TB/s.  D counts the "
                        "entries of the frame's binned lists: a frame that bins at 64x64 (config.binning_tile) has half the entries of "
                        "the same frame binned at the 32x32 compositing tile")
    # the device picks the binning tile per frame (the compositing tile or 2 x 2 of them): D counts entries of THOSE lists
    bin_w, bin_h = r.binning_tile()
    out["config"].update({"binning_tile": f"{bin_w}x{bin_h}", "compositing_tile": f"{tile_w}x{tile_h}",
                          "avg_visible": V, "avg_tile_entries": D,
                          "single_stream_fps": single_stream_fps})
    out["roofline"] = roofline
    out["kernels"] = kernels
    out["stages"] = {k: {"ms": v} for k, v in stage_acc.items()}
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(gpc, views[0], viewport)
    elif world == 1:
        out["cpu_baseline"] = None


if __name__ == "__main__":
    main()
