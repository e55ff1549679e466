#!/usr/bin/env python3
"""Golden vectors of the 'lstm_model' (container-only; never runs on the GPU box).

Builds the reference's own LSTM networks (neural_network_lstm_model.py through
``Muzero(model_structure='lstm_model')``, imported by path as make_fixtures.py does), and records
with make_fixtures' recorders: the inference wrappers on random inputs (L2) and whole searches
(``Monte_carlo_tree_search.run``) with the consumed random stream (L1 / L3).

The reference initialises only the Linear layers (weights_init, N(0, 1/137)); that makes every
LSTM input ~1e-2 and every policy ~uniform, so the fixtures scale all tensors to a seeded
N(0, std) instead: the architecture, not its initialisation, is what is pinned.

Blob order = state_dict order of the three modules (include/mzs_layout.h, mzs_lstm_layout).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_lstm_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_fixtures as mf  # noqa: E402

mm = mf.mm


def lstm_model(obs, A, Sd, H, L, seed, std):
    mz = mm.Muzero(model_structure="lstm_model", observation_space_dimensions=mf.Box((obs,)),
                   action_space_dimensions=mf.Discrete(A), state_space_dimensions=Sd,
                   hidden_layer_dimensions=H, number_of_hidden_layer=L, k_hypothetical_steps=5, device="cpu")
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
            for p in net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return mz


def pack_lstm(mz):
    parts = []
    for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
        for name, t in net.state_dict().items():
            parts.append(t.detach().numpy().astype(np.float32).ravel())
    sd = mz.dynamics_function.state_dict()
    assert list(sd)[:2] == ["reward.0.weight", "reward.0.bias"] and list(sd)[2] == "reward.1.weight_ih_l0"
    return np.concatenate(parts)


def main():
    torch.set_num_threads(1)
    models = {"lstmA": lstm_model(obs=4, A=2, Sd=31, H=64, L=1, seed=21, std=0.35),
              "lstmB": lstm_model(obs=8, A=4, Sd=16, H=32, L=2, seed=22, std=0.5)}
    for name, mz in models.items():
        dims = mf.model_dims(mz)
        mf.save(f"weights_{name}.npz", blob=pack_lstm(mz), kind=np.int64(1), **{k: np.int64(v) for k, v in dims.items()})
        mv = mf.model_vectors(mz, 48, seed=31)
        mf.save(f"model_L2_{name}.npz", **mv)
    g = 0.997
    mf.save("search_lstmA_S50.npz", **mf.run_searches(models["lstmA"], mf.cartpole_obs(48, 2001, scale=1.0),
                                                     20000 + np.arange(48), 50, True, g))
    mf.save("search_lstmB_S30.npz", **mf.run_searches(models["lstmB"], mf.cartpole_obs(32, 2002, dim=8, scale=1.0),
                                                     21000 + np.arange(32), 30, True, g))


if __name__ == "__main__":
    main()
