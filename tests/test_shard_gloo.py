"""World-size-2 gloo test of the multi-rank path (CPU): games shard over ranks with no
data-path collective; the union of the per-rank results equals one single-rank run over
all games.  The per-rank compute here is the CPU oracle (there is no GPU in this container);
on the GPU box the same sharding code drives the HIP engine (bench.py)."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT, fixture_weights


def run_two_ranks(tmp_path, name, source, out):
    """Start the worker through muzero_amd.shard.launch_ranks — the launcher `bench.py --gpus N`
    and `tools/bench_train.py --gpus N` use (require_gpus=False: gloo ranks on CPU here)."""
    from muzero_amd import shard
    script = tmp_path / name
    script.write_text(source)
    rc = shard.launch_ranks(str(script), [], 2, require_gpus=False,
                            extra_env=dict(MZ_ROOT=ROOT, MZ_OUT=out, OMP_NUM_THREADS="1"))
    assert rc == 0, f"{name}: launcher exit code {rc}"

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
    out = str(tmp_path / "gathered.npz")
    run_two_ranks(tmp_path, "worker.py", WORKER, out)
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


TRAIN_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MZ_ROOT"], "tests"))
from muzero_amd import shard, train
from oracle import mz_oracle as mzo
from conftest import load_golden
rank, world = shard.init("gloo")
fx = load_golden("train_tag123_K5_B128.npz")
d = mzo.dims(*fx["dims"])
B = fx["s0_obs"].shape[0]; n = B // world; sl = slice(rank * n, (rank + 1) * n)
out = mzo.train_grad(d, fx["blob0"], fx["s0_obs"][sl], fx["s0_actions"][:, sl], fx["s0_target_value"][:, sl],
                     fx["s0_target_reward"][:, sl], fx["s0_target_policy"][:, sl],
                     fx["s0_is_ratio"][sl].astype(np.float32), global_batch=B)
bucket, grad, loss = train.make_bucket(out["grad"].size, 5)
grad.copy_(torch.from_numpy(out["grad"])); loss.copy_(torch.from_numpy(np.append(out["loss_terms"], np.float32(out["total_loss"]))))
calls = []
_ar = dist_all_reduce = torch.distributed.all_reduce
torch.distributed.all_reduce = lambda *a, **k: (calls.append(1), _ar(*a, **k))[1]
assert train.allreduce_gradient(bucket) == world
torch.distributed.all_reduce = _ar
assert len(calls) == 1, "a training step is ONE collective"
blob = fx["blob0"].copy(); m = np.zeros_like(blob); v = np.zeros_like(blob)
mzo.adam_step(blob, m, v, grad.numpy(), 1, train.scheduled_lr("cosineannealinglr", 0.01, 10000, 0))
np.savez(os.environ["MZ_OUT"] + f".{rank}.npz", blob=blob, grad=grad.numpy(), loss=loss.numpy())
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
'''


def test_data_parallel_training_step_two_ranks(tmp_path):
    """Each rank: gradients of its half of the batch, ONE all-reduce of the flat bucket, the same
    optimizer step -> both replicas hold the weights of the single-rank step (= the reference's)."""
    from conftest import load_golden
    out = str(tmp_path / "dp")
    run_two_ranks(tmp_path, "train_worker.py", TRAIN_WORKER, out)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["blob"], r1["blob"]) and np.array_equal(r0["grad"], r1["grad"])
    fx = load_golden("train_tag123_K5_B128.npz")
    g = fx["s0_grad"]
    assert np.abs(r0["grad"] - g).max() <= 1e-5 * np.abs(g).max()
    assert abs(r0["loss"][-1] - fx["s0_losses"][0]) <= 1e-5 * fx["s0_losses"][0]
    assert np.abs(r0["blob"] - fx["s0_blob"]).max() <= 2e-3 * 0.01 + 1e-6


SELFPLAY_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MZ_ROOT"], "tests"))
from muzero_amd import shard, selfplay
from oracle import mz_oracle as mzo
import torch.distributed as dist
rank, world = shard.init("gloo")
N, G, T = 40, 16, 30
seeds = (selfplay.game_seed_index(rank, N) * 2654435761 + 5) & 0x7FFFFFFF
ar = mzo.ArenaOracle(G, T, n_rows=N)
ar.begin(seeds, episode=0)
while True:                      # a fixed stand-in "search": visit counts from the observation
    v = np.stack([(ar.obs[:, 2] > 0) * 30 + 10, (ar.obs[:, 2] <= 0) * 30 + 10], 1).astype(np.int32)
    if ar.step(v, ar.obs[:, 0], np.full((G, 2), 0.5), 0.0) == 0:
        break
store = selfplay.ReplayStore(2 * N, 8, 5, 3)
prio = mzo.make_priority(ar.length, ar.traj_reward, ar.traj_root, 3, selfplay.discount_powers(0.997, 3))
store.save_games(np.arange(N), ar.length, priorities=prio)
lens = [torch.zeros(N, dtype=torch.int32) for _ in range(world)]
dist.all_gather(lens, torch.from_numpy(ar.length.copy()))
if rank == 0:
    np.savez(os.environ["MZ_OUT"], lengths=torch.cat(lens).numpy(), big_n=store.big_n_of_importance_sampling_ratio)
dist.barrier(); dist.destroy_process_group()
'''


def test_self_play_shards_by_game_seed(tmp_path):
    """Data-parallel self-play: ranks play disjoint games (seed = global game index), no collective
    in the play path; the union equals one process playing all games (a game's trajectory depends
    on its seed only).  CPU oracle arena stands in for the GPU arena (same arithmetic, bit-exact)."""
    from muzero_amd import selfplay
    from oracle import mz_oracle as mzo
    out = str(tmp_path / "sp.npz")
    run_two_ranks(tmp_path, "sp_worker.py", SELFPLAY_WORKER, out)
    got = np.load(out)
    N, G, T = 80, 7, 30          # one process, all 80 games, a different slot count on purpose
    seeds = (np.arange(N, dtype=np.int64) * 2654435761 + 5) & 0x7FFFFFFF
    ar = mzo.ArenaOracle(G, T, n_rows=N)
    ar.begin(seeds, episode=0)
    while True:
        v = np.stack([(ar.obs[:, 2] > 0) * 30 + 10, (ar.obs[:, 2] <= 0) * 30 + 10], 1).astype(np.int32)
        if ar.step(v, ar.obs[:, 0], np.full((G, 2), 0.5), 0.0) == 0:
            break
    assert np.array_equal(got["lengths"], ar.length)
    assert int(got["big_n"]) == int(ar.length[:40].sum())


LOOP_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MZ_ROOT"], "tests"))
from muzero_amd import shard, selfplay, train
from oracle import mz_oracle as mzo
from conftest import fixture_weights
import torch.distributed as dist
rank, world = shard.init("gloo")
d, blob = fixture_weights("tag123")
blob = blob.copy(); m = np.zeros_like(blob); v = np.zeros_like(blob)
N, G, T, B, K, td = 24, 8, 20, 16, 3, 5
dp = selfplay.discount_powers(0.997, td)
store = selfplay.ReplayStore(4 * N, B, K, td, discount=0.997)
np.random.seed(100 + rank)
calls = []
_ar = torch.distributed.all_reduce
torch.distributed.all_reduce = lambda *a, **k: (calls.append(1), _ar(*a, **k))[1]
ar = mzo.ArenaOracle(G, T, n_rows=4 * N)
for it in range(2):                                   # learning_cycle: play a round, then train (self_play.py:217-262)
    seeds = (selfplay.game_seed_index(rank, N) * 2654435761 + 5) & 0x7FFFFFFF
    rows = np.arange(it * N, (it + 1) * N)
    ar.begin(seeds, episode=it, row0=it * N)
    while True:                  # stand-in search results from the observation + the CURRENT weights (rank-identical)
        bias = float(blob[:8].sum())
        vis = np.stack([(ar.obs[:, 2] + bias > 0) * 30 + 10, (ar.obs[:, 2] + bias <= 0) * 30 + 10], 1).astype(np.int32)
        if ar.step(vis, ar.obs[:, 0], np.full((G, 2), 0.5), 1.0) == 0:
            break
    prio = mzo.make_priority(ar.length[rows], ar.traj_reward[rows], ar.traj_root[rows], td, dp)
    store.save_games(rows, ar.length[rows], priorities=prio)
    for step in range(3):
        row_pos, ratio, game_pos = store.sample_batch()
        obs, actions, tv, tr, tp = mzo.make_batch(ar.length, ar.traj_obs, ar.traj_action, ar.traj_reward, ar.traj_root,
                                                  ar.traj_policy, row_pos, K, td, dp)
        out = mzo.train_grad(d, blob, obs, actions, tv, tr, tp, np.asarray(ratio, np.float32), global_batch=B * world)
        bucket, grad, loss = train.make_bucket(out["grad"].size, K)
        grad.copy_(torch.from_numpy(out["grad"]))
        loss.copy_(torch.from_numpy(np.append(out["loss_terms"], np.float32(out["total_loss"]))))
        train.allreduce_gradient(bucket)
        mzo.adam_step(blob, m, v, grad.numpy(), 3 * it + step + 1, 0.01)
        store.update_value(out["priority"], game_pos)
assert len(calls) == 6, calls                          # ONE collective per training step, none in the play path
torch.distributed.all_reduce = _ar
shard.assert_replicas_identical(blob)
np.savez(os.environ["MZ_OUT"] + f".{rank}.npz", blob=blob, lengths=ar.length, loss=loss.numpy())
dist.barrier(); dist.destroy_process_group()
'''


def test_self_play_training_loop_two_ranks(tmp_path):
    """The multi-rank learning cycle on CPU stand-ins (oracle arena, oracle training step): every rank plays its
    own games and samples its own replay store, a training step is ONE all-reduce of the gradient bucket, and
    the replicas' weights — which steer the next round's play — stay bit-identical.  The GPU counterpart is
    tools/bench_selfplay.py --gpus N (tests/test_gpu_multirank.py)."""
    out = str(tmp_path / "loop")
    run_two_ranks(tmp_path, "loop_worker.py", LOOP_WORKER, out)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["blob"], r1["blob"]) and np.array_equal(r0["loss"], r1["loss"])
    assert not np.array_equal(r0["lengths"], r1["lengths"])          # different games on the two ranks
    d, blob = fixture_weights("tag123")
    assert np.abs(r0["blob"] - blob).max() > 1e-3 and np.isfinite(r0["loss"]).all()


BCAST_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MZ_ROOT"])
from muzero_amd import shard
import torch.distributed as dist
rank, world = shard.init("gloo")
torch.manual_seed(100 + rank)                      # what separate processes do: different initial weights
blob = torch.normal(0.0, 1 / 137.035999, size=(12734,)).numpy()
try:
    shard.assert_replicas_identical(blob)
    differ = False
except RuntimeError:
    differ = True
synced = shard.broadcast_blob(blob)
shard.assert_replicas_identical(synced)            # raises if the broadcast did not make them equal
np.savez(os.environ["MZ_OUT"] + f".{rank}.npz", own=blob, synced=synced, differ=differ)
dist.barrier(); dist.destroy_process_group()
'''


def test_initial_weights_are_broadcast(tmp_path):
    """Every process draws its own initial weights (Muzero.initiate_model_weight uses torch's global
    generator): Trainer.sync_replicas -> shard.broadcast_blob makes rank 0's the common start, and
    assert_replicas_identical catches replicas that differ."""
    out = str(tmp_path / "bc")
    run_two_ranks(tmp_path, "bcast_worker.py", BCAST_WORKER, out)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert bool(r0["differ"]) and bool(r1["differ"])
    assert not np.array_equal(r0["own"], r1["own"])
    assert np.array_equal(r0["synced"], r0["own"]) and np.array_equal(r1["synced"], r0["own"])


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` is how the driver asks for N ranks: with fewer GPUs visible it must
    fail loudly instead of printing an n_gpus: 1 line (there is no GPU in the build container, so
    N = 2 is already too many here; on the GPU box the visible count decides)."""
    import torch
    n = torch.cuda.device_count() + 1
    for script in ("bench.py", os.path.join("tools", "bench_train.py")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", str(max(n, 2))],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0
        assert "GPU(s) are visible" in r.stderr
        assert '"n_gpus"' not in r.stdout


VISION_TRAIN_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MZ_ROOT"], "tests"))
sys.path.insert(0, os.path.join(os.environ["MZ_ROOT"], "oracle"))
from muzero_amd import shard, train
import mz_vision_train_oracle as vo
from test_vision_train_oracle import step_inputs
from conftest import load_golden
import torch.distributed as dist
rank, world = shard.init("gloo")
fx = load_golden("train_visA_K3_B6.npz")
dims, K, B = fx["dims"], int(fx["K"]), int(fx["B"])
n = B // world; sl = slice(rank * n, (rank + 1) * n)
kw = step_inputs(fx, 0)
# this rank's rows of every Dropout2d call's (B, C) mask
masks, p = [], 0
for c in fx["s0_mask_calls"]:
    masks.append(kw["masks"][p:p + B * c].reshape(B, c)[sl].ravel()); p += B * c
out = vo.train_grad(fx["blob0"], dims, K, kw["obs"][sl], kw["actions"][:, sl], kw["tv"][:, sl], kw["tr"][:, sl],
                    kw["tpol"][:, sl], kw["is_ratio"][sl], np.concatenate(masks), inv_global_batch=1.0 / B)
opt = vo.Optimizer("adam", fx["blob0"].size, dims)
# the bucket as mzs_train_grad fills it: gradient, with the running statistics' slots holding this shard's
# share of their change; then ONE all-reduce; then the optimizer adds the summed change to the blob
slots = out["grad"].astype(np.float32)
slots[opt.is_stat] = ((out["running"] - fx["blob0"])[opt.is_stat] * (n / B)).astype(np.float32)
bucket, grad, loss = train.make_bucket(slots.size, K)
grad.copy_(torch.from_numpy(slots)); loss.copy_(torch.from_numpy(out["losses"].astype(np.float32)))
assert train.allreduce_gradient(bucket) == world
g = grad.numpy()
running = fx["blob0"] + np.where(opt.is_stat, g, 0.0)
blob = opt.apply(fx["blob0"], np.where(opt.is_stat, 0.0, g), running, float(fx["lr"]))
np.savez(os.environ["MZ_OUT"] + f".{rank}.npz", blob=blob, grad=g, loss=loss.numpy(), own_running=out["running"])
dist.barrier(); dist.destroy_process_group()
'''


def test_vision_data_parallel_training_step_two_ranks(tmp_path):
    """vision_model on two ranks: BatchNorm statistics are per shard (as torch.nn.DataParallel's replicas compute
    them, muzero_model.py:333-340), the gradient bucket carries the running statistics' change in their own
    slots, ONE all-reduce, and both replicas end with identical weights whose running statistics are the mean
    of the two shards'."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mz_vision_train_oracle as vo
    from conftest import load_golden
    out = str(tmp_path / "vdp")
    run_two_ranks(tmp_path, "vision_train_worker.py", VISION_TRAIN_WORKER, out)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["blob"], r1["blob"]) and np.array_equal(r0["grad"], r1["grad"])
    fx = load_golden("train_visA_K3_B6.npz")
    stat = vo.Optimizer("adam", fx["blob0"].size, fx["dims"]).is_stat
    mean_running = 0.5 * (r0["own_running"] + r1["own_running"])
    assert not np.array_equal(r0["own_running"][stat], r1["own_running"][stat])
    assert np.abs(r0["blob"][stat] - mean_running[stat]).max() <= 1e-6
    assert np.isfinite(r0["blob"]).all() and not np.array_equal(r0["blob"][~stat], fx["blob0"][~stat])
    # loss terms: the two shards' shares of the batch mean add up to a full-batch mean of per-shard-normalised nets
    assert np.isfinite(r0["loss"]).all() and r0["loss"][-1] > 0
