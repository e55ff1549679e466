"""The training-step oracle (oracle/mz_train_oracle.c) pinned to the reference's own
Muzero.train (tests/golden/train_*.npz, made by tests/golden/make_train_fixtures.py), and the
host-side pieces of the training path (lr schedules, reshape_batch)."""
import numpy as np
import pytest

from conftest import load_golden

MLP_TRAIN_FIXTURES = ["train_tag123_K5_B128.npz", "train_tag133_K3_B32.npz", "train_synA4_K4_B16.npz",
                      "train_init_K10_B64.npz"]
# 'lstm_model' (tests/golden/make_lstm_train_fixtures.py): every head's LSTM runs along the batch axis
LSTM_TRAIN_FIXTURES = ["train_lstmA_K5_B128.npz", "train_lstmB_K3_B32.npz", "train_lstmC_K4_B48.npz"]
# the reference's other loss types (tests/golden/make_loss_fixtures.py): KL / KL, MSE / zero, MSE / MSE
LOSS_TRAIN_FIXTURES = ["train_loss_general_kkc_K3_B24.npz", "train_loss_game_K3_B24.npz",
                       "train_loss_game_mmc_K3_B24.npz"]
# OneCycleLR (muzero_model.py:433): rate and Adam beta1 / SGD momentum change every step; six steps of twelve
ONECYCLE_FIXTURES = ["train_onecycle_adam_K3_B24.npz", "train_onecycle_sgd_K3_B24.npz"]
TRAIN_FIXTURES = MLP_TRAIN_FIXTURES + LSTM_TRAIN_FIXTURES + LOSS_TRAIN_FIXTURES + ONECYCLE_FIXTURES


def loss_type_of(fx):
    return str(fx["loss_type"]) if "loss_type" in fx else "general"
# fp32 tolerances of the training path (north_star: 1e-5 relative in fp32)
LOSS_REL = 1e-5          # every loss term and loss_nn
GRAD_REL = 1e-5          # max |g - g_ref| <= GRAD_REL * max |g_ref| per step
PRIO_REL = 1e-5          # |priority - ref| <= PRIO_REL * (1 + |ref|) + one inverse-transform quantum


def step_inputs(fx, s):
    isr = fx[f"s{s}_is_ratio"]
    return dict(obs=fx[f"s{s}_obs"], actions=fx[f"s{s}_actions"], target_value=fx[f"s{s}_target_value"],
                target_reward=fx[f"s{s}_target_reward"], target_policy=fx[f"s{s}_target_policy"],
                is_ratio=None if isr.size == 1 else isr.astype(np.float32))


def prio_tol(ref, target):
    # priority = |v - target|; the reference's fp32 inverse transform quantises the predicted
    # value v to ~1.2e-4 * sqrt(|v| + 1) (tests/test_oracle_golden.py::value_err)
    vmag = np.abs(ref) + np.abs(target)
    return PRIO_REL * (1 + vmag) + 2.4e-4 * np.sqrt(vmag + 1)


def check_against_fixture(fx, s, out, blob_after=None, lr=None):
    L = fx[f"s{s}_losses"]
    assert abs(out["total_loss"] - L[0]) <= LOSS_REL * abs(L[0])
    assert np.all(np.abs(out["loss_terms"] - L[1:]) <= LOSS_REL * np.abs(L[1:]) + 1e-7)
    g = fx[f"s{s}_grad"]
    assert np.abs(out["grad"] - g).max() <= GRAD_REL * np.abs(g).max()
    pr = fx[f"s{s}_priority"]
    assert np.all(np.abs(out["priority"] - pr) <= prio_tol(pr, fx[f"s{s}_target_value"]))
    if blob_after is not None:
        ref = fx[f"s{s}_blob"]
        # Adam's first steps move every weight by ~lr * g / (|g| + eps): where |g| is tiny the update
        # direction is ill-conditioned, so compare in units of the step size.  The LSTM fixtures hold
        # gradient entries of ~1e-9, a tenth of Adam's eps (far ends of the batch-axis recurrence):
        # there an error of 1e-7 * max|g| already turns the step by half a percent
        frac = 1e-2 if len(fx["dims"]) > 5 and int(fx["dims"][5]) == 1 else 2e-3
        # the MSE terms of 'game' / 'game_mmc' leave gradient entries of ~3e-9 in the dynamics trunk (the
        # reward head has no loss at all under 'game'): the same conditioning, measured 1.0e-2 * lr
        if "loss_type" in fx:
            frac = 2e-2
        assert np.abs(blob_after - ref).max() <= frac * lr + 1e-6


@pytest.mark.parametrize("fname", TRAIN_FIXTURES)
def test_train_oracle_matches_reference(fname):
    from muzero_amd.train import scheduled_lr, scheduled_momentum
    from oracle import mz_oracle as mzo
    fx = load_golden(fname)
    d = mzo.dims(*fx["dims"])
    blob = fx["blob0"].copy()
    m, v = np.zeros_like(blob), np.zeros_like(blob)
    opt, sch = str(fx["optimizer"]), str(fx["scheduler"])
    for s in range(int(fx["n_steps"])):
        out = mzo.train_grad(d, blob, **step_inputs(fx, s), loss_type=loss_type_of(fx))
        lr = scheduled_lr(sch, float(fx["lr"]), int(fx["epoch"]), s)
        mzo.set_momentum(scheduled_momentum(sch, int(fx["epoch"]), s))
        if opt == "adam":
            mzo.adam_step(blob, m, v, out["grad"], s + 1, lr)
        else:
            mzo.sgd_step(blob, m, out["grad"], s + 1, lr)
        check_against_fixture(fx, s, out, blob, lr)
        # the reference's scheduler state after this step = the rate of the next one
        assert abs(scheduled_lr(sch, float(fx["lr"]), int(fx["epoch"]), s + 1) - float(fx[f"s{s}_lr_next"])) \
            <= 1e-12
        blob = fx[f"s{s}_blob"].copy()      # re-anchor: later steps test the step, not the drift
    mzo.set_momentum(0.9)


def test_data_parallel_shards_sum_to_the_full_batch():
    """Two shards with inv_batch = 1/global give gradients / losses that add up to the one-rank
    result (what the all-reduce relies on)."""
    from oracle import mz_oracle as mzo
    fx = load_golden("train_tag123_K5_B128.npz")
    d = mzo.dims(*fx["dims"])
    inp = step_inputs(fx, 0)
    full = mzo.train_grad(d, fx["blob0"], **inp)
    B = inp["obs"].shape[0]
    parts = []
    for sl in (slice(0, B // 2), slice(B // 2, B)):
        sh = dict(obs=inp["obs"][sl], actions=inp["actions"][:, sl], target_value=inp["target_value"][:, sl],
                  target_reward=inp["target_reward"][:, sl], target_policy=inp["target_policy"][:, sl],
                  is_ratio=inp["is_ratio"][sl])
        parts.append(mzo.train_grad(d, fx["blob0"], global_batch=B, **sh))
    g = parts[0]["grad"] + parts[1]["grad"]
    assert np.abs(g - full["grad"]).max() <= 1e-6 * np.abs(full["grad"]).max()
    assert abs(parts[0]["total_loss"] + parts[1]["total_loss"] - full["total_loss"]) <= 1e-6 * full["total_loss"]
    assert np.allclose(parts[0]["loss_terms"] + parts[1]["loss_terms"], full["loss_terms"], rtol=1e-6)
    assert np.array_equal(np.concatenate([parts[0]["priority"], parts[1]["priority"]], axis=1), full["priority"])


def test_reshape_batch_layout():
    import torch
    from muzero_amd.train import reshape_batch
    K, A = 3, 2
    batch = []
    for b in range(4):
        acts = [np.eye(A)[(b + k) % A] for k in range(K)]
        tg = [[10.0 * b + k, float(k), np.array([0.25, 0.75])] for k in range(K)]
        batch.append((torch.full((1, 4), float(b)), acts, tg))
    obs, actions, tv, tr, tp, ratio, pos = reshape_batch((batch, np.array([0]), "pos"), 4, A, K)
    assert obs.shape == (4, 4) and np.all(obs[2] == 2.0)
    assert actions.shape == (K, 4, A) and actions[1, 3].tolist() == [1.0, 0.0]
    assert tv[2, 1] == 12.0 and tr[2, 1] == 2.0 and tp.shape == (K, 4, A)
    assert ratio is None and pos == "pos"
    _, _, _, _, _, ratio, _ = reshape_batch((batch, np.arange(4) + 1.0, None), 4, A, K)
    assert ratio.tolist() == [1.0, 2.0, 3.0, 4.0]


def test_scheduled_lr_closed_forms():
    import math
    from muzero_amd.train import scheduled_lr
    assert scheduled_lr("cosineannealinglr", 0.01, 100, 0) == 0.01
    assert scheduled_lr("cosineannealinglr", 0.01, 100, 1) == 0.01
    assert abs(scheduled_lr("cosineannealinglr", 0.01, 100, 51) - 0.005) < 1e-12
    assert abs(scheduled_lr("steplr", 1.0, 10, 3) - 0.99 ** 2) < 1e-15
    assert scheduled_lr(None, 0.3, 10, 7) == 0.3
    assert abs(scheduled_lr("cosineannealinglrwarmrestarts", 1.0, 10, 12) - (1 + math.cos(math.pi * 0.1)) / 2) < 1e-12
    # OneCycleLR(max_lr=0.01, total_steps=50): values read off the reference's optimizer (lr, beta1) after
    # train() calls 1..4 (tests/golden/make_loss_fixtures.py's recipe)
    from muzero_amd.train import scheduled_momentum
    for count, lr, b1 in ((0, 0.0004, 0.95), (1, 0.0003999999999999993, 0.95), (2, 0.0005203460215272467, 0.9487463956090911),
                          (3, 0.0008753494340683881, 0.945048443395121), (4, 0.0014472088841534555, 0.9390915741234015)):
        assert abs(scheduled_lr("onecyclelr", 0.01, 50, count) - lr) < 1e-15
        assert abs(scheduled_momentum("onecyclelr", 50, count) - b1) < 1e-15
    assert scheduled_momentum("steplr", 50, 3) == 0.9
    with pytest.raises(ValueError):
        scheduled_lr("onecyclelr", 0.01, 5, 7)            # torch raises past total_steps
    # the end of the cycle against torch itself, stepped the reference's way (optimizer.step(), then
    # scheduler.step(epoch=count), muzero_model.py:764-786): train total_steps + 1 still runs, at the rate
    # of one_cycle(total_steps), and only then torch raises
    import warnings
    import torch
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=0.01)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.01, total_steps=10)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for count in range(12):
            g = opt.param_groups[0]
            assert abs(scheduled_lr("onecyclelr", 0.01, 10, count) - g["lr"]) < 1e-15
            assert abs(scheduled_momentum("onecyclelr", 10, count) - g["betas"][0]) < 1e-15
            p.grad = torch.ones(1)
            opt.step()
            if count < 11:
                sch.step(epoch=count)
            else:
                with pytest.raises(ValueError):
                    sch.step(epoch=count)
    with pytest.raises(ValueError):
        scheduled_lr("onecyclelr", 0.01, 10, 12)
    with pytest.raises(NotImplementedError):
        scheduled_lr("exponentiallr", 0.1, 10, 1)
