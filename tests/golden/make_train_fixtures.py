#!/usr/bin/env python3
"""Golden vectors of the training step (container-only; never runs on the GPU box).

Imports the reference by path from /root/reference (through make_fixtures.py's recipe),
runs its own ``Muzero.train`` (muzero_model.py:791-811: reshape_batch -> compute_forward ->
evaluate_loss -> backpropagation) on seeded synthetic batches and stores, per step: the
loss terms (``store_loss[-1]``), ``new_priority``, every parameter's ``.grad`` and the
weights after the optimizer step, all in the packed blob order of include/mzs_layout.h.

Batch element format (replay_buffer.py:150-154):
  (obs Tensor(1, obs_dim), [K action vectors (A,)], [[value, last_reward, policy (A,)] x K])
Absorbing positions (replay_buffer.py:120-129, game.py:257): zero action vector, value 0,
zero policy.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_train_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_fixtures as mf  # noqa: E402  (installs the gymnasium stub, imports the reference)

mm = mf.mm


def pack_grads(mz):
    """Gradients in blob order (same walk as make_fixtures.pack_weights)."""
    L = mz.number_of_hidden_layer
    last = 2 * L + 2
    r = dict(mz.representation_function.named_parameters())
    d = dict(mz.dynamics_function.named_parameters())
    p = dict(mz.prediction_function.named_parameters())
    names = [(r, "state_norm.0")]
    if L > 0:
        names.append((r, "state_norm.2"))
    names += [(r, f"state_norm.{last}"), (d, "reward.0")]
    if L > 0:
        names.append((d, "reward.2"))
    names += [(d, f"reward.{last}"), (d, f"next_state_normalized.{last}"), (p, "policy.0")]
    if L > 0:
        names.append((p, "policy.2"))
    names += [(p, f"policy.{last}"), (p, f"value.{last}")]
    parts = []
    for tab, n in names:
        for suffix in (".weight", ".bias"):
            g = tab[n + suffix].grad
            parts.append(np.zeros(tab[n + suffix].shape, np.float32).ravel() if g is None
                         else g.detach().numpy().astype(np.float32).ravel())
    return np.concatenate(parts)


def copy_weights(dst, src):
    with torch.no_grad():
        for a, b in ((dst.representation_function, src.representation_function),
                     (dst.dynamics_function, src.dynamics_function),
                     (dst.prediction_function, src.prediction_function)):
            for pd, ps in zip(a.parameters(), b.parameters()):
                pd.copy_(ps)


def make_model(dims, K, lr, opt, sch, epoch, src=None, seed=0, std=0.3):
    obs, A, Sd, H, L = dims
    torch.manual_seed(seed)
    mz = mm.Muzero(model_structure="mlp_model", observation_space_dimensions=mf.Box((obs,)),
                   action_space_dimensions=mf.Discrete(A), state_space_dimensions=Sd,
                   hidden_layer_dimensions=H, number_of_hidden_layer=L, k_hypothetical_steps=K,
                   learning_rate=lr, optimizer=opt, lr_scheduler=sch, loss_type="general",
                   num_of_epoch=epoch, device="cpu")
    if src is not None:
        copy_weights(mz, src)
    else:
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
                for p in net.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * std)
    return mz


def synth_batch(rs, dims, K, B, absorbing_frac, is_mode):
    obs_dim, A, Sd, H, L = dims
    obs = rs.uniform(-1, 1, size=(B, obs_dim)).astype(np.float32)
    act_idx = rs.randint(0, A, size=(K, B))
    actions = np.zeros((K, B, A), np.float32)
    for k in range(K):
        actions[k, np.arange(B), act_idx[k]] = 1.0
    value = rs.uniform(0.0, 60.0, size=(K, B)).astype(np.float32)
    value[:, : B // 8] = rs.uniform(-20.0, 0.0, size=(K, B // 8))      # negative targets too
    value[0, B // 8: B // 8 + 2] = [3.0, 400.0]                          # integer / clamped targets
    reward = rs.choice([0.0, 1.0, 1.0, -1.0, 2.5], size=(K, B)).astype(np.float32)
    policy = rs.dirichlet([0.7] * A, size=(K, B)).astype(np.float32)
    # absorbing tails: from some unroll position on, zero action / value / policy
    n_abs = int(B * absorbing_frac)
    for b in range(n_abs):
        start = 1 + rs.randint(0, K)
        actions[start - 1:, b] = 0.0 if start - 1 < K else actions[start - 1:, b]
        value[start:, b] = 0.0
        policy[start:, b] = 0.0
    if is_mode == "vector":
        is_ratio = rs.uniform(0.2, 1.5, size=B)
    else:
        is_ratio = np.array([0])          # replay_buffer.py:165: uniform sampling
    return dict(obs=obs, actions=actions, target_value=value, target_reward=reward,
                target_policy=policy, is_ratio=np.asarray(is_ratio, np.float64))


def to_reference_batch(bt):
    K, B, A = bt["actions"].shape
    batch = []
    for b in range(B):
        acts = [bt["actions"][k, b].astype(np.float64) for k in range(K)]
        tg = [[float(bt["target_value"][k, b]), float(bt["target_reward"][k, b]),
               bt["target_policy"][k, b].astype(np.float64)] for k in range(K)]
        batch.append((torch.tensor(bt["obs"][b:b + 1]), acts, tg))
    pos = np.stack([np.arange(B), np.arange(B)], axis=1)
    return (batch, bt["is_ratio"], pos)


def run_case(name, mz, dims, K, B, n_steps, seed, absorbing_frac, is_mode, hyper):
    rs = np.random.RandomState(seed)
    out = dict(dims=np.array(dims, np.int32), K=K, B=B, n_steps=n_steps,
               blob0=mf.pack_weights(mz), **{k: np.asarray(v) for k, v in hyper.items()})
    for s in range(n_steps):
        bt = synth_batch(rs, dims, K, B, absorbing_frac, is_mode)
        prio, pos = mz.train(to_reference_batch(bt))
        for k, v in bt.items():
            out[f"s{s}_{k}"] = v
        out[f"s{s}_losses"] = np.array([float(x) for x in mz.store_loss[-1]], np.float64)
        out[f"s{s}_priority"] = np.stack([p[:, 0] for p in prio]).astype(np.float32)   # [K, B]
        out[f"s{s}_grad"] = pack_grads(mz)
        out[f"s{s}_blob"] = mf.pack_weights(mz)
        out[f"s{s}_lr_next"] = np.float64(mz.optimizer.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "losses[0] =", out["s0_losses"][:4], "|g| =", np.abs(out["s0_grad"]).max())


def main():
    torch.set_num_threads(1)
    # A: BASELINE config C1 shape (tag 123 weights), unroll 5, batch 128, Adam + cosine annealing
    src = mf.load_checkpoint(123)
    dims = (4, 2, 31, 64, 0)
    hyper = dict(lr=0.01, optimizer="adam", scheduler="cosineannealinglr", epoch=10000)
    mz = make_model(dims, 5, 0.01, "adam", "cosineannealinglr", 10000, src=src)
    run_case("train_tag123_K5_B128.npz", mz, dims, 5, 128, 3, 11, 0.1, "vector", hyper)
    # B: tied mid layers (tag 133, L = 4), Adam, no scheduler, many absorbing rows
    src = mf.load_checkpoint(133)
    dims = (4, 2, 31, 64, 4)
    hyper = dict(lr=0.003, optimizer="adam", scheduler="none", epoch=100)
    mz = make_model(dims, 3, 0.003, "adam", None, 100, src=src)
    run_case("train_tag133_K3_B32.npz", mz, dims, 3, 32, 2, 12, 0.4, "vector", hyper)
    # C: LunarLander-like shape, one mid layer, SGD momentum + StepLR, uniform sampling (no IS ratio)
    dims = (8, 4, 31, 64, 1)
    hyper = dict(lr=0.05, optimizer="sgd", scheduler="steplr", epoch=100)
    mz = make_model(dims, 4, 0.05, "sgd", "steplr", 100, seed=5, std=0.25)
    run_case("train_synA4_K4_B16.npz", mz, dims, 4, 16, 3, 13, 0.2, "scalar", hyper)
    # D: reference config 123 unroll (K = 10) on fresh N(0, 1/137) weights (weights_init)
    dims = (4, 2, 31, 64, 0)
    hyper = dict(lr=0.01, optimizer="adam", scheduler="cosineannealinglr", epoch=10000)
    mz = make_model(dims, 10, 0.01, "adam", "cosineannealinglr", 10000, seed=7, std=1 / 137.035999)
    run_case("train_init_K10_B64.npz", mz, dims, 10, 64, 2, 14, 0.1, "vector", hyper)


if __name__ == "__main__":
    main()
