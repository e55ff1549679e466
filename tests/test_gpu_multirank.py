"""Two RCCL ranks on two GPUs (skipped on a one-GPU box): the HIP search sharded by game index and
the data-parallel training step through the launcher `bench.py --gpus N` uses.  The CPU
counterparts (gloo, oracle compute) are tests/test_shard_gloo.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, fixture_weights

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the node")]

WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MZ_ROOT"], "tests"))
from muzero_amd import rng, shard
from muzero_amd.engine import SearchEngine
from muzero_amd.train import Trainer
from conftest import fixture_weights, load_golden
import torch.distributed as dist
rank, world = shard.init("nccl")
local = int(os.environ["LOCAL_RANK"])
dev = torch.device("cuda", local)
G, S, A = 256, 25, 2
d, blob = fixture_weights("tag123")
dims = (d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers)
obs, seeds = shard.shard_inputs(rank, G, 4)
eng = SearchEngine(dims, G, S, discount=0.997, device=local)
# every rank starts from DIFFERENT weights; the trainer's broadcast makes rank 0's the common ones
eng.set_weights(blob if rank == 0 else blob * 0.5)
fx = load_golden("train_tag123_K5_B128.npz")
B = fx["s0_obs"].shape[0]; n = B // world; sl = slice(rank * n, (rank + 1) * n)
tr = Trainer(eng, 5, n, "adam", 0.01, "cosineannealinglr", 10000)
shard.assert_replicas_identical(tr.weights(), dev)
out = eng.run_batch(obs, seeds=seeds, num_simulations=S, train=True)
v, r = shard.gather_results(out["visits"], out["root_value"], dev)
tr.step(fx["s0_obs"][sl], fx["s0_actions"][:, sl], fx["s0_target_value"][:, sl], fx["s0_target_reward"][:, sl],
        fx["s0_target_policy"][:, sl], fx["s0_is_ratio"][sl].astype(np.float32))
torch.cuda.synchronize(dev)
shard.assert_replicas_identical(tr.weights(), dev)
if rank == 0:
    np.savez(os.environ["MZ_OUT"], visits=v, root_value=r, blob=tr.weights(), grad=tr.grad.cpu().numpy())
dist.barrier(); dist.destroy_process_group()
'''


def test_two_gpu_ranks_search_and_train(tmp_path):
    from muzero_amd import rng, shard
    from conftest import load_golden
    from oracle import mz_oracle as mzo
    out = str(tmp_path / "two.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    rc = shard.launch_ranks(str(script), [], 2, extra_env=dict(MZ_ROOT=ROOT, MZ_OUT=out))
    assert rc == 0
    got = np.load(out)
    d, blob = fixture_weights("tag123")
    G, S = 256, 25
    obs = np.concatenate([shard.shard_inputs(r, G, 4)[0] for r in range(2)])
    dirichlet, noise = mzo.streams(np.arange(2 * G), 2, S, 0.25, True)     # the device's streams, bit for bit
    ref = mzo.search(d, blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True)
    same = np.all(got["visits"] == ref["visits"], axis=1)
    assert same.all()
    assert np.array_equal(got["root_value"].view(np.uint32), ref["root_value"].view(np.uint32))
    fx = load_golden("train_tag123_K5_B128.npz")
    g = fx["s0_grad"]
    assert np.abs(got["grad"] - g).max() <= 1e-5 * np.abs(g).max()
    assert np.abs(got["blob"] - fx["s0_blob"]).max() <= 2e-3 * 0.01 + 1e-6


def test_bench_two_ranks_reports_two_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup",
                        "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["world_size"] == 2 and j["config"]["backend"] == "nccl"
    assert j["scaling"] == "weak" and j["value"] > 0


def test_selfplay_bench_two_ranks():
    """tools/bench_selfplay.py --gpus 2: an arena + replay store per rank (disjoint game seeds), data-parallel
    training steps through ONE all-reduce each; the run itself asserts the replicas' weights stayed identical."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_selfplay.py"), "--gpus", "2", "--games", "256",
                        "--sims", "10", "--max-moves", "40", "--rounds", "2", "--games-per-round", "512", "--train-steps",
                        "4", "--batch", "64", "--rolling"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["global_batch"] == 128 and j["games_per_round_whole_job"] == 1024
    assert j["useful_sims_per_s"] > 0 and 0.3 < j["slot_utilisation"] <= 1.0
