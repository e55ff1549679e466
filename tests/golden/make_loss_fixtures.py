#!/usr/bin/env python3
"""Golden vectors of the training step under the reference's other loss types (container-only).

loss_type 'general_kkc' (KL divergence on value / reward), 'game' (MSE on value, zero loss on
reward) and 'game_mmc' (MSE on both), muzero_model.py:364-403 with Loss_function
(neural_network_mlp_model.py:236-262); the policy term is the cross-entropy in all of them.
Same recipe and file layout as make_train_fixtures.py (the reference's own Muzero.train).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_loss_fixtures.py
"""
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
import make_train_fixtures as mt  # noqa: E402

mf, mm = mt.mf, mt.mm


def make_model(dims, K, lr, loss_type, seed, std=0.3, sch=None, opt="adam", epoch=100):
    obs, A, Sd, H, L = dims
    torch.manual_seed(seed)
    mz = mm.Muzero(model_structure="mlp_model", observation_space_dimensions=mf.Box((obs,)),
                   action_space_dimensions=mf.Discrete(A), state_space_dimensions=Sd,
                   hidden_layer_dimensions=H, number_of_hidden_layer=L, k_hypothetical_steps=K,
                   learning_rate=lr, optimizer=opt, lr_scheduler=sch, loss_type=loss_type,
                   num_of_epoch=epoch, device="cpu")
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in (mz.representation_function, mz.dynamics_function, mz.prediction_function):
            for p in net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return mz


def main():
    torch.set_num_threads(1)
    dims = (4, 2, 31, 64, 1)
    synth = mt.synth_batch
    for i, lt in enumerate(("general_kkc", "game", "game_mmc")):
        hyper = dict(lr=0.003, optimizer="adam", scheduler="none", epoch=100, loss_type=lt)
        mz = make_model(dims, 3, 0.003, lt, seed=21 + i)
        if lt == "general_kkc":
            # the reference's KL term takes log(target + 1e-9), and transform_with_support gives a
            # NEGATIVE weight to a negative target (muzero_model.py:553): NaN losses there.  The
            # fixture keeps to non-negative targets, where the term is defined.
            def positive(*a, **k):
                bt = synth(*a, **k)
                bt["target_value"] = np.abs(bt["target_value"])
                bt["target_reward"] = np.abs(bt["target_reward"])
                return bt
            mt.synth_batch = positive
        else:
            mt.synth_batch = synth
        mt.run_case(f"train_loss_{lt}_K3_B24.npz", mz, dims, 3, 24, 2, 31 + i, 0.25, "vector", hyper)
    # OneCycleLR (muzero_model.py:433): the rate AND Adam's beta1 / SGD's momentum change every step
    mt.synth_batch = synth
    for opt, lr in (("adam", 0.01), ("sgd", 0.05)):
        hyper = dict(lr=lr, optimizer=opt, scheduler="onecyclelr", epoch=12)
        mz = make_model(dims, 3, lr, "general", seed=41, sch="onecyclelr", opt=opt, epoch=12)
        mt.run_case(f"train_onecycle_{opt}_K3_B24.npz", mz, dims, 3, 24, 6, 51, 0.25, "vector", hyper)


if __name__ == "__main__":
    main()
