#!/usr/bin/env python3
"""Golden vectors of the 'vision_model' training step (container-only; never runs on the GPU box).

Runs the reference's own ``Muzero.train`` (muzero_model.py:791-811) on its ResNet-v2 networks
(neural_network_vision_model.py) in training mode: BatchNorm2d on batch statistics (and updating its
running statistics), Dropout2d(p = 0.5) live.  torch draws the dropout masks from its global generator,
whose stream nothing outside torch can reproduce; so forward hooks RECORD the mask of every Dropout2d call
in call order (one bit per (sample, channel): which channels survived) and the fixtures store them —
the reference runs unmodified, and whoever restates the step is handed the same masks
(``mzs_train_set_dropout``), the way the injected-output search fixtures hand over model outputs.

Stored per step: the batch (observations regenerated from `obs_seed` by the tests), the masks, the loss
terms (``store_loss[-1]``), ``new_priority``, every parameter's ``.grad`` (zeros for tensors autograd never
reached and for the running statistics) and the blob after the optimizer step (running statistics as the
K + 1 forward passes left them), all in muzero_amd.model.vision_blob_layout order.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_vision_train_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
import make_fixtures as mf  # noqa: E402
import make_vision_fixtures as mv  # noqa: E402

mm = mf.mm


def pack_vision_grads(mz, L):
    """Gradients in blob order: make_vision_fixtures.pack_vision with every parameter's storage swapped for
    its .grad (zeros where autograd left None) and the running statistics for zeros, then restored."""
    nets = (mz.representation_function, mz.dynamics_function, mz.prediction_function)
    saved = []
    with torch.no_grad():
        for net in nets:
            for t in list(net.parameters()) + list(net.buffers()):
                if t.dtype != torch.float32:
                    continue                                   # num_batches_tracked
                saved.append((t, t.data))
                g = getattr(t, "grad", None)
                t.data = torch.zeros_like(t.data) if g is None else g.detach().clone()
        out = mv.pack_vision(mz, L)
        for t, d in saved:
            t.data = d
    return out


def record_masks(mz, sink):
    """forward hooks on every Dropout2d: append the (B, C) survival bits of the call"""
    def hook(_mod, inp, out):
        x, y = inp[0].detach(), out.detach()
        alive = (y != 0).flatten(2).any(dim=2)
        dead_input = ~(x != 0).flatten(2).any(dim=2)      # an all-zero channel: the mask cannot matter
        sink.append((alive | dead_input).to(torch.uint8).numpy().copy())
    hs = []
    for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                hs.append(m.register_forward_hook(hook))
    return hs


def synth_batch(rs, A, K, B, obs_seed, is_mode):
    obs = np.random.RandomState(obs_seed).uniform(0.0, 1.0, size=(B, 3, 98, 98)).astype(np.float32)
    act_idx = rs.randint(0, A, size=(K, B))
    actions = np.zeros((K, B, A), np.float32)
    for k in range(K):
        actions[k, np.arange(B), act_idx[k]] = 1.0
    value = rs.uniform(-3.0, 12.0, size=(K, B)).astype(np.float32)
    reward = rs.choice([0.0, 1.0, 1.0, -1.0, 2.5], size=(K, B)).astype(np.float32)
    policy = rs.dirichlet([0.7] * A, size=(K, B)).astype(np.float32)
    # one absorbing tail (replay_buffer.py:120-129): zero action vector, value 0, zero policy
    actions[K - 1:, 0] = 0.0
    value[K - 1:, 0] = 0.0
    policy[K - 1:, 0] = 0.0
    is_ratio = rs.uniform(0.3, 1.4, size=B) if is_mode == "vector" else np.array([0])
    return dict(actions=actions, target_value=value, target_reward=reward, target_policy=policy,
                is_ratio=np.asarray(is_ratio, np.float64)), obs


def to_reference_batch(bt, obs):
    K, B, A = bt["actions"].shape
    batch = []
    for b in range(B):
        acts = [bt["actions"][k, b].astype(np.float64) for k in range(K)]
        tg = [[float(bt["target_value"][k, b]), float(bt["target_reward"][k, b]),
               bt["target_policy"][k, b].astype(np.float64)] for k in range(K)]
        batch.append((torch.tensor(obs[b:b + 1]), acts, tg))
    pos = np.stack([np.arange(B), np.arange(B)], axis=1)
    return (batch, bt["is_ratio"], pos)


def run_case(name, A, Sd, H, L, K, B, n_steps, seed, lr, opt, is_mode):
    torch.manual_seed(seed)
    mz = mv.vision_model(A, Sd, H, L, seed)
    # vision_model() builds with the defaults; rebuild the optimizer for this case's hyper-parameters
    mz.k_hypothetical_steps, mz.lr, mz.opt, mz.sch, mz.loss_type = K, lr, opt, None, "general"
    mz.init_criterion_and_optimizer()
    rs = np.random.RandomState(seed)
    out = dict(dims=np.array([98, A, Sd, H, L], np.int32), K=K, B=B, n_steps=n_steps, lr=lr, optimizer=opt,
               blob0=mv.pack_vision(mz, L))
    for s in range(n_steps):
        obs_seed = 7000 + 10 * seed + s
        bt, obs = synth_batch(rs, A, K, B, obs_seed, is_mode)
        masks = []
        hooks = record_masks(mz, masks)
        prio, _ = mz.train(to_reference_batch(bt, obs))
        for h in hooks:
            h.remove()
        for k, v in bt.items():
            out[f"s{s}_{k}"] = v
        out[f"s{s}_obs_seed"] = np.int64(obs_seed)
        out[f"s{s}_masks"] = np.concatenate([m.ravel() for m in masks])
        out[f"s{s}_mask_calls"] = np.array([m.shape[1] for m in masks], np.int32)   # channels of every call
        out[f"s{s}_losses"] = np.array([float(x) for x in mz.store_loss[-1]], np.float64)
        out[f"s{s}_priority"] = np.stack([p[:, 0] for p in prio]).astype(np.float32)
        out[f"s{s}_grad"] = pack_vision_grads(mz, L)
        out[f"s{s}_blob"] = mv.pack_vision(mz, L)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "losses =", out["s0_losses"][:5], "|g|max =", np.abs(out["s0_grad"]).max(),
          "mask calls =", len(out["s0_mask_calls"]), "alive =", out["s0_masks"].mean())


def main():
    torch.set_num_threads(1)
    run_case("train_visA_K3_B6.npz", 2, 11, 16, 1, 3, 6, 2, 51, 0.01, "adam", "vector")
    run_case("train_visB_K2_B4.npz", 4, 9, 12, 2, 2, 4, 2, 52, 0.05, "sgd", "scalar")


if __name__ == "__main__":
    main()
