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


def visible_gpus():
    """GPUs this process could use.  torch.cuda.device_count() does not initialise the HIP runtime
    (it reads the driver's device list), so a launcher may call this before spawning ranks."""
    return int(torch.cuda.device_count())


def launch_ranks(script, argv, n_ranks, require_gpus=True, master_port=None, extra_env=None):
    """Start `n_ranks` copies of `script argv...` as one-process-per-GPU ranks on this node and
    return the launcher's exit code (the reference's counterparts: ray.init + one remote task per
    game, self_play.py:212-228; nn.DataParallel, muzero_model.py:333-340).

    The ranks are children of a fresh `python -m torch.distributed.run` subprocess; the calling
    process must not have touched the GPU (it never does here: nothing below initialises HIP), and
    it is never replaced by exec.  Rendezvous on 127.0.0.1.  With require_gpus the call refuses
    to start more ranks than there are visible GPUs instead of letting ranks share a card."""
    import socket
    import subprocess
    import sys
    n_ranks = int(n_ranks)
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    if require_gpus and visible_gpus() < n_ranks:
        print(f"{os.path.basename(script)}: {n_ranks} ranks requested but only {visible_gpus()} GPU(s) "
              "are visible on this node", file=sys.stderr, flush=True)
        return 2
    if master_port is None:
        with socket.socket() as sk:       # a free port; the tiny race with the rendezvous is accepted
            sk.bind(("127.0.0.1", 0))
            master_port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(master_port), script, *argv]
    return subprocess.run(cmd, env=env).returncode


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


def broadcast_blob(blob, device=None, src=0):
    """Rank `src`'s packed weight blob on every rank (one broadcast of ~51 KB for the CartPole MLP;
    SURVEY.md 8e).  The data-parallel training step keeps replicas identical only if they START
    identical, and every process draws its initial weights from its own torch generator — so
    Trainer / SelfPlay call this once at construction.  No-op for one rank."""
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return blob
    t = torch.from_numpy(blob.copy()).to(device or "cpu")
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def assert_replicas_identical(blob, device=None, what="weights"):
    """Raise on every rank if the ranks' blobs differ (all-reduce MIN / MAX of a 64-bit checksum
    and of the blob itself is overkill: the elementwise MAX - MIN of 51 KB is one tiny collective)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32).copy()).to(device or "cpu")
    hi, lo = t.clone(), t.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    if not torch.equal(hi, lo):
        raise RuntimeError(f"data-parallel replicas hold different {what} "
                           f"({int((hi != lo).sum())} of {t.numel()} elements differ)")


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
