"""The N>1 path on CPU: world_size-2 gloo processes exercise the view sharding and the timing reduction that
bench.py uses with RCCL on the GPU box (views shard, nothing else is exchanged)."""
import os
import socket
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
