"""The N>1 path on CPU: world_size-2 gloo processes exercise the view sharding and the timing reduction that
bench.py uses with RCCL on the GPU box (views shard, nothing else is exchanged)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_views_for_rank_partition():
    sys.path.insert(0, os.path.join(ROOT, "web-splat_amd"))
    from websplat.shard import views_for_rank
    for n_views in (0, 1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            parts = [views_for_rank(n_views, r, world) for r in range(world)]
            flat = sorted(v for p in parts for v in p)
            assert flat == list(range(n_views))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            for r, p in enumerate(parts):
                assert all(v % world == r for v in p)
    with pytest.raises(ValueError):
        views_for_rank(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "web-splat_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from websplat.shard import aggregate_throughput, gather_view_assignment, views_for_rank
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        mine = views_for_rank(9, rank, world)
        everyone = gather_view_assignment(mine, dist)
        frames, elapsed = aggregate_throughput(100, 1.0 + rank, dist, "cpu")  # rank 1 is the slow one
        dist.barrier()
        q.put((rank, mine, everyone, frames, elapsed))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1] == [0, 2, 4, 6, 8] and results[1][1] == [1, 3, 5, 7]
    for _, _, everyone, frames, elapsed in results:
        assert everyone == [[0, 2, 4, 6, 8], [1, 3, 5, 7]]
        assert frames == 200          # whole-job frames
        assert elapsed == 2.0         # MAX over ranks -> fps = 200 / 2.0


def test_bench_dry_run_two_ranks_gloo():
    """bench.py's OWN sharding / planning / barrier / MAX-reduction code, launched exactly as the driver launches the
    N-GPU run (torch.distributed.run, one process per rank), on CPU with backend gloo: frames are not rendered
    (--dry-run), everything else is the code the RCCL run executes."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "c1",
           "--steps", "40", "--warmup", "4", "--views", "9", "--streams", "3"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout            # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 40 and out["warmup"] == 4
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    cfg = out["config"]
    assert cfg["rank_views"] == [[0, 2, 4, 6, 8], [1, 3, 5, 7]]      # view i -> rank i mod N
    assert cfg["frames_in_flight"] == 3 and cfg["workload"].startswith("c1:")
    assert "gloo communicator, world size 2" in cfg["collective"]
    # whole-job value: all ranks' frames over the MAX elapsed (each dry frame sleeps 0.2 ms)
    assert abs(out["value"] - 2 * 40 / (out["ms_per_step"] * 40 / 1e3)) < 1e-6 * out["value"]
    assert 0.15 < out["ms_per_step"] < 5.0


def test_ranks_take_disjoint_shares_of_the_physical_cores():
    """bench.pin_host_share (round 4): rank r of w pins itself to the r-th of w slices of the physical cores it may run on --
    SMT siblings stay together, the shares are disjoint and cover the allowed set, a rank alone is not pinned at all -- and
    sizes its OpenMP team to the share (and to its part of a cgroup CPU quota, if there is one)."""
    import json as _json
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("needs at least four CPUs")
    code = ("import json, os, sys; sys.path.insert(0, %r); import bench; "
            "r, w = int(sys.argv[1]), int(sys.argv[2]); cpus, note = bench.pin_host_share(r, w); "
            "print(json.dumps({'cpus': sorted(cpus), 'aff': sorted(os.sched_getaffinity(0)), 'omp': os.environ.get('OMP_NUM_THREADS'), "
            "'note': note, 'cores': bench._physical_cores(%r), 'quota': bench.cpu_quota()}))") % (ROOT, allowed)
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "WS_BENCH_PIN")}

    def run(r, w, extra=None):
        p = subprocess.run([sys.executable, "-c", code, str(r), str(w)], capture_output=True, text=True, timeout=120,
                           env=dict(env, **(extra or {})))
        assert p.returncode == 0, p.stderr[-2000:]
        return _json.loads(p.stdout.strip().splitlines()[-1])
    alone = run(0, 1)
    assert alone["aff"] == allowed and alone["cpus"] == allowed          # one rank: nothing is pinned
    cores = alone["cores"]
    assert sorted(c for g in cores for c in g) == allowed                # the groups partition the allowed CPUs
    w = 2 if len(cores) >= 2 else 1
    shares = [run(r, w) for r in range(w)]
    seen = []
    for r, s in enumerate(shares):
        assert s["aff"] == s["cpus"] and s["cpus"], s                    # the process IS pinned to what it reports
        assert int(s["omp"]) >= 1 and int(s["omp"]) <= len(s["cpus"])
        if s["quota"]:
            assert int(s["omp"]) <= max(1, int(s["quota"] / w))
        for g in cores:                                                  # a physical core belongs to one rank entirely
            inter = set(g) & set(s["cpus"])
            assert not inter or inter == set(g), (g, s["cpus"])
        seen += s["cpus"]
    assert sorted(seen) == allowed or w == 1                             # disjoint and covering
    assert len(seen) == len(set(seen))
    off = run(1, 2, {"WS_BENCH_PIN": "0"})
    assert off["aff"] == allowed and "not pinned" in off["note"]
