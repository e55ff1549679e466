"""World-size-2 gloo test of the multi-rank path (CPU): games shard over ranks with no
data-path collective; the union of the per-rank results equals one single-rank run over
all games.  The per-rank compute here is the CPU oracle (there is no GPU in this container);
on the GPU box the same sharding code drives the HIP engine (bench.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT, fixture_weights

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["MZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MZ_ROOT"], "tests"))
from muzero_amd import rng, shard
from oracle import mz_oracle as mzo
from conftest import fixture_weights
rank, world = shard.init("gloo")
G, S, A = 24, 11, 2
d, blob = fixture_weights("tag123")
obs, seeds = shard.shard_inputs(rank, G, 4)
assert seeds[0] == rank * G and len(seeds) == G
dirichlet, noise = rng.noise_streams(seeds, A, S, 0.25, True)
out = mzo.search(d, blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=dirichlet,
                 noise=noise, train=True)
t = shard.max_over_ranks(1.0 + rank)
assert t == float(world), t
v, r = shard.gather_results(out["visits"], out["root_value"])
if rank == 0:
    np.savez(os.environ["MZ_OUT"], visits=v, root_value=r)
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
'''


def test_two_ranks_equal_one(tmp_path):
    from muzero_amd import rng, shard
    from oracle import mz_oracle as mzo
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MZ_ROOT=ROOT, MZ_OUT=out, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    subprocess.run(cmd, env=env, check=True, timeout=300, capture_output=True)
    got = np.load(out)
    # single-rank reference over the same 48 global games
    d, blob = fixture_weights("tag123")
    obs = np.concatenate([shard.shard_inputs(r, 24, 4)[0] for r in range(2)])
    seeds = np.arange(48)
    dirichlet, noise = rng.noise_streams(seeds, 2, 11, 0.25, True)
    ref = mzo.search(d, blob, 11, 0.997, 0.25, rng.pbc_table(11), obs=obs, dirichlet=dirichlet,
                     noise=noise, train=True)
    assert np.array_equal(got["visits"], ref["visits"])
    assert np.array_equal(got["root_value"], ref["root_value"])


def test_game_range_partitions():
    from muzero_amd import shard
    allg = np.concatenate([shard.game_range(r, 4096) for r in range(8)])
    assert np.array_equal(allg, np.arange(8 * 4096))
