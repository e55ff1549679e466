#!/usr/bin/env python3
"""Golden vectors of the training step of the 'lstm_model' (container-only; never runs on the GPU box).

Same recipe as make_train_fixtures.py: the reference's own ``Muzero.train`` (muzero_model.py:791-811)
on ``Muzero(model_structure='lstm_model')`` (neural_network_lstm_model.py), seeded synthetic batches,
recorded per step: loss terms, new_priority, every parameter's ``.grad`` and the weights after the
optimizer step, in blob order (= state_dict order, include/mzs_layout.h mzs_lstm_layout).

What this pins (and what looks odd is the reference's behaviour, kept as is): every head is
Linear -> nn.LSTM fed a 2-D (B, H) tensor, i.e. ONE unbatched sequence of length B per call with a
zero initial state, so sample b's outputs depend on samples 0..b-1 of the batch.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_lstm_train_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_fixtures as mf  # noqa: E402
import make_lstm_fixtures as mlf  # noqa: E402
import make_train_fixtures as mtf  # noqa: E402

mm = mf.mm


def nets(mz):
    return (mz.representation_function, mz.dynamics_function, mz.prediction_function)


def pack_lstm_grads(mz):
    parts = []
    for net in nets(mz):
        for p in net.parameters():
            parts.append(np.zeros(tuple(p.shape), np.float32).ravel() if p.grad is None
                         else p.grad.detach().numpy().astype(np.float32).ravel())
    return np.concatenate(parts)


def make_model(dims, K, lr, opt, sch, epoch, seed, std):
    """std None: the reference's own initialisation (weights_init on the Linear layers, torch's
    default U(-1/sqrt(n), 1/sqrt(n)) on the LSTM tensors); else every tensor ~ seeded N(0, std)."""
    obs, A, Sd, H, L = dims
    torch.manual_seed(seed)
    mz = mm.Muzero(model_structure="lstm_model", observation_space_dimensions=mf.Box((obs,)),
                   action_space_dimensions=mf.Discrete(A), state_space_dimensions=Sd,
                   hidden_layer_dimensions=H, number_of_hidden_layer=L, k_hypothetical_steps=K,
                   learning_rate=lr, optimizer=opt, lr_scheduler=sch, loss_type="general",
                   num_of_epoch=epoch, device="cpu")
    if std is not None:
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for net in nets(mz):
                for p in net.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * std)
    return mz


def run_case(name, mz, dims, K, B, n_steps, seed, absorbing_frac, is_mode, hyper):
    rs = np.random.RandomState(seed)
    out = dict(dims=np.array(list(dims) + [1], np.int32), K=K, B=B, n_steps=n_steps,
               blob0=mlf.pack_lstm(mz), **{k: np.asarray(v) for k, v in hyper.items()})
    for s in range(n_steps):
        bt = mtf.synth_batch(rs, dims, K, B, absorbing_frac, is_mode)
        prio, pos = mz.train(mtf.to_reference_batch(bt))
        for k, v in bt.items():
            out[f"s{s}_{k}"] = v
        out[f"s{s}_losses"] = np.array([float(x) for x in mz.store_loss[-1]], np.float64)
        out[f"s{s}_priority"] = np.stack([p[:, 0] for p in prio]).astype(np.float32)   # [K, B]
        out[f"s{s}_grad"] = pack_lstm_grads(mz)
        out[f"s{s}_blob"] = mlf.pack_lstm(mz)
        out[f"s{s}_lr_next"] = np.float64(mz.optimizer.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "losses[0] =", out["s0_losses"][:4], "|g| =", np.abs(out["s0_grad"]).max())


def main():
    torch.set_num_threads(1)
    # A: BASELINE config C4 shape (CartPole, S_dim 31 / H 64 / one LSTM layer), the reference's own
    #    initialisation, unroll 5, batch 128, Adam + cosine annealing, prioritised sampling
    dims = (4, 2, 31, 64, 1)
    hyper = dict(lr=0.01, optimizer="adam", scheduler="cosineannealinglr", epoch=10000)
    mz = make_model(dims, 5, 0.01, "adam", "cosineannealinglr", 10000, seed=31, std=None)
    run_case("train_lstmA_K5_B128.npz", mz, dims, 5, 128, 3, 41, 0.1, "vector", hyper)
    # B: two LSTM layers, four actions, wide-ish weights so that the recurrence and every gate matter,
    #    SGD momentum + StepLR, uniform sampling
    dims = (8, 4, 16, 32, 2)
    hyper = dict(lr=0.05, optimizer="sgd", scheduler="steplr", epoch=100)
    mz = make_model(dims, 3, 0.05, "sgd", "steplr", 100, seed=32, std=0.3)
    run_case("train_lstmB_K3_B32.npz", mz, dims, 3, 32, 3, 42, 0.25, "scalar", hyper)
    # C: same shape as A, scaled weights (every tensor N(0, 0.25)), Adam, no scheduler
    dims = (4, 2, 31, 64, 1)
    hyper = dict(lr=0.003, optimizer="adam", scheduler="none", epoch=100)
    mz = make_model(dims, 4, 0.003, "adam", None, 100, seed=33, std=0.25)
    run_case("train_lstmC_K4_B48.npz", mz, dims, 4, 48, 2, 43, 0.2, "vector", hyper)


if __name__ == "__main__":
    main()
