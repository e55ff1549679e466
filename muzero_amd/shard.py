"""Sharding of independent games over ranks (one process per GPU).

The search needs no collective: rank r owns games [r*G, (r+1)*G) with per-game seed
= global game index.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used
only for the timing barrier / max-over-ranks and, optionally, to gather the small
result arrays on rank 0.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for 1 rank)."""
    rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world


def game_range(rank, games_per_rank):
    """Global game indices owned by `rank` (contiguous blocks, weak scaling)."""
    return np.arange(rank * games_per_rank, (rank + 1) * games_per_rank, dtype=np.int64)


def shard_inputs(rank, games_per_rank, obs_dim, obs_scale=0.05):
    """Synthetic benchmark inputs of one rank: observations from
    RandomState(1000 + rank), per-game seeds = global game index."""
    obs = np.random.RandomState(1000 + rank).uniform(
        -obs_scale, obs_scale, size=(games_per_rank, obs_dim)).astype(np.float32)
    return obs, game_range(rank, games_per_rank)


def max_over_ranks(seconds, device=None):
    """Wall time of the slowest rank (the number the bench reports)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(visits, root_value, device=None):
    """all_gather of the per-rank result arrays (visits i32 [G, A], root_value f32 [G]) in
    global game order; only needed when one process keeps a central replay buffer."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return visits, root_value
    world = dist.get_world_size()
    v = torch.as_tensor(visits, device=device or "cpu").contiguous()
    r = torch.as_tensor(root_value, device=device or "cpu").contiguous()
    vs = [torch.empty_like(v) for _ in range(world)]
    rs = [torch.empty_like(r) for _ in range(world)]
    dist.all_gather(vs, v)
    dist.all_gather(rs, r)
    return torch.cat(vs).cpu().numpy(), torch.cat(rs).cpu().numpy()
