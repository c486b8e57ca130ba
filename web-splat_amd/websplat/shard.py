"""View sharding across the GPUs of one node.

A frame is a pure function of (scene, camera): the scene is replicated, view i goes to rank i mod N, and no
collective sits on the data path.  The only communication is the reduction of the timing at the end.
`dist` is torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
from typing import List, Optional, Tuple


def views_for_rank(n_views: int, rank: int, world: int) -> List[int]:
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    return list(range(rank, n_views, world))


def aggregate_throughput(frames_local: int, elapsed_local: float, dist=None, device: str = "cpu") -> Tuple[int, float]:
    """(total frames over all ranks, MAX elapsed over ranks).  Whole-job fps = frames / elapsed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return frames_local, elapsed_local
    import torch
    t = torch.tensor([float(elapsed_local)], dtype=torch.float64, device=device)
    f = torch.tensor([float(frames_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return int(round(f.item())), float(t.item())


def gather_view_assignment(my_views: List[int], dist=None) -> Optional[List[List[int]]]:
    """All ranks' view lists (verification only; never on the timed path)."""
    if dist is None or not dist.is_initialized():
        return [my_views]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, my_views)
    return out
