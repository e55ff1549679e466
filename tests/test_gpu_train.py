"""GPU parity of the training step (HIP kernels through the C-ABI) against the CPU oracle and
the reference's own Muzero.train fixtures.  fp32 tolerances (north_star: 1e-5 relative)."""
import numpy as np
import pytest

from conftest import load_golden
from test_train_oracle import (GRAD_REL, LOSS_REL, TRAIN_FIXTURES, check_against_fixture, loss_type_of,
                               step_inputs)

pytestmark = pytest.mark.gpu


def make_trainer(fx):
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import Trainer
    dims = tuple(int(x) for x in fx["dims"])
    sch = str(fx["scheduler"])
    eng = SearchEngine(dims, max_games=1, max_simulations=0)
    eng.set_weights(fx["blob0"])
    tr = Trainer(eng, int(fx["K"]), int(fx["B"]), str(fx["optimizer"]), float(fx["lr"]),
                 None if sch == "none" else sch, int(fx["epoch"]), loss_type_of(fx))
    return eng, tr


def gpu_out(tr, B):
    import torch
    torch.cuda.synchronize()
    loss = tr.loss.cpu().numpy()
    return dict(grad=tr.grad.cpu().numpy(), loss_terms=loss[:-1], total_loss=float(loss[-1]),
                priority=tr.priority[:, :B].cpu().numpy())


@pytest.mark.parametrize("fname", TRAIN_FIXTURES)
def test_train_step_vs_oracle_and_reference(fname):
    from oracle import mz_oracle as mzo
    fx = load_golden(fname)
    d = mzo.dims(*fx["dims"])
    eng, tr = make_trainer(fx)
    B = int(fx["B"])
    for s in range(int(fx["n_steps"])):
        inp = step_inputs(fx, s)
        blob_before = tr.weights()
        lr = tr.current_lr()
        tr.grad_step(inp["obs"], inp["actions"], inp["target_value"], inp["target_reward"],
                     inp["target_policy"], inp["is_ratio"])
        got = gpu_out(tr, B)
        ref = mzo.train_grad(d, blob_before, **inp, loss_type=loss_type_of(fx))
        assert abs(got["total_loss"] - ref["total_loss"]) <= LOSS_REL * abs(ref["total_loss"])
        assert np.all(np.abs(got["loss_terms"] - ref["loss_terms"]) <= LOSS_REL * np.abs(ref["loss_terms"]) + 1e-7)
        assert np.abs(got["grad"] - ref["grad"]).max() <= GRAD_REL * np.abs(ref["grad"]).max()
        assert np.abs(got["priority"] - ref["priority"]).max() <= 1e-5 * (1 + np.abs(ref["priority"]).max())
        tr.apply()
        blob_after = tr.weights()
        if s == 0:
            check_against_fixture(fx, 0, got, blob_after, lr)      # the reference itself (same weights)
        # optimizer kernel vs the oracle's optimizer on the same gradient
        m = getattr(test_train_step_vs_oracle_and_reference, "_m", None)
        if s == 0:
            m, v = np.zeros_like(blob_before), np.zeros_like(blob_before)
        else:
            m, v = test_train_step_vs_oracle_and_reference._m, test_train_step_vs_oracle_and_reference._v
        w = blob_before.copy()
        from muzero_amd.train import scheduled_momentum
        mzo.set_momentum(scheduled_momentum(tr.sch, tr.epoch, s))   # OneCycleLR cycles it, else 0.9
        if str(fx["optimizer"]) == "adam":
            mzo.adam_step(w, m, v, got["grad"], s + 1, lr)
        else:
            mzo.sgd_step(w, m, got["grad"], s + 1, lr)
        mzo.set_momentum(0.9)
        test_train_step_vs_oracle_and_reference._m, test_train_step_vs_oracle_and_reference._v = m, v
        assert np.abs(blob_after - w).max() <= 1e-6 + 1e-5 * lr
        if str(fx["scheduler"]) == "onecyclelr":                   # all six steps against the reference's weights
            assert np.abs(blob_after - fx[f"s{s}_blob"]).max() <= (s + 1) * 2e-3 * float(fx["lr"]) + 1e-6
    eng.close()


def test_training_updates_what_the_search_reads():
    """After mzs_train_apply the inference kernels use the new weights (the transposed store is
    refreshed on the device): inference on the trainer's engine == oracle inference with the
    read-back blob, bit for bit."""
    from oracle import mz_oracle as mzo
    fx = load_golden("train_tag123_K5_B128.npz")
    d = mzo.dims(*fx["dims"])
    eng, tr = make_trainer(fx)
    inp = step_inputs(fx, 0)
    tr.step(inp["obs"], inp["actions"], inp["target_value"], inp["target_reward"], inp["target_policy"],
            inp["is_ratio"])
    blob = tr.weights()
    assert not np.array_equal(blob, fx["blob0"])
    hid, pol, val = [t.cpu().numpy() for t in eng.initial_inference(inp["obs"])]
    oh, op, ov = mzo.initial_inference(d, blob, inp["obs"])
    assert np.array_equal(hid.view(np.uint32), oh.view(np.uint32))
    assert np.array_equal(pol.view(np.uint32), op.view(np.uint32))
    assert np.array_equal(val.view(np.uint32), ov.view(np.uint32))
    eng.close()


def test_lstm_training_updates_what_the_search_reads():
    """'lstm_model': after mzs_train_apply the device store (Linear, i / g / o gate rows, both bias
    vectors) holds the new weights — inference == oracle inference with the read-back blob, bit for
    bit — and a search runs on them."""
    from oracle import mz_oracle as mzo
    fx = load_golden("train_lstmB_K3_B32.npz")
    d = mzo.dims(*fx["dims"])
    eng, tr = make_trainer(fx)
    inp = step_inputs(fx, 0)
    tr.step(inp["obs"], inp["actions"], inp["target_value"], inp["target_reward"], inp["target_policy"],
            inp["is_ratio"])
    blob = tr.weights()
    assert np.count_nonzero(blob != fx["blob0"]) > 0.9 * blob.size
    hid, pol, val = [t.cpu().numpy() for t in eng.initial_inference(inp["obs"])]
    oh, op, ov = mzo.initial_inference(d, blob, inp["obs"])
    assert np.array_equal(hid.view(np.uint32), oh.view(np.uint32))
    assert np.array_equal(pol.view(np.uint32), op.view(np.uint32))
    assert np.array_equal(val.view(np.uint32), ov.view(np.uint32))
    eng.close()


def test_lstm_partial_batch_and_repeat():
    """A batch smaller than max_batch (rows are strided by the capacity, not the batch) and a second
    call on the same handle: both == the oracle on the same inputs."""
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import Trainer
    from oracle import mz_oracle as mzo
    fx = load_golden("train_lstmC_K4_B48.npz")
    d = mzo.dims(*fx["dims"])
    eng = SearchEngine(tuple(int(x) for x in fx["dims"]), max_games=1, max_simulations=0)
    eng.set_weights(fx["blob0"])
    tr = Trainer(eng, int(fx["K"]), 64, "adam", 0.003, None, 100)
    inp = step_inputs(fx, 0)
    for B in (48, 17, 48):
        sub = dict(obs=inp["obs"][:B], actions=inp["actions"][:, :B], target_value=inp["target_value"][:, :B],
                   target_reward=inp["target_reward"][:, :B], target_policy=inp["target_policy"][:, :B],
                   is_ratio=inp["is_ratio"][:B])
        prio = tr.grad_step(sub["obs"], sub["actions"], sub["target_value"], sub["target_reward"],
                            sub["target_policy"], sub["is_ratio"])
        import torch
        torch.cuda.synchronize()
        ref = mzo.train_grad(d, fx["blob0"], **sub)
        loss = tr.loss.cpu().numpy()
        assert abs(float(loss[-1]) - ref["total_loss"]) <= LOSS_REL * abs(ref["total_loss"])
        assert np.abs(tr.grad.cpu().numpy() - ref["grad"]).max() <= GRAD_REL * np.abs(ref["grad"]).max()
        assert np.abs(prio.cpu().numpy() - ref["priority"]).max() <= 1e-5 * (1 + np.abs(ref["priority"]).max())
    eng.close()


def test_muzero_train_drop_in_lstm():
    """Muzero(model_structure='lstm_model').train with the reference's batch format against the
    reference fixture (Adam + cosine annealing, unroll 5, batch 128)."""
    import torch
    from muzero_amd.model import Muzero
    fx = load_golden("train_lstmA_K5_B128.npz")
    mz = Muzero(model_structure="lstm_model", observation_space_dimensions=4,
                action_space_dimensions=2, state_space_dimensions=31, hidden_layer_dimensions=64,
                number_of_hidden_layer=1, k_hypothetical_steps=5, learning_rate=0.01, optimizer="adam",
                lr_scheduler="cosineannealinglr", num_of_epoch=10000)
    mz.set_weights(fx["blob0"])
    for s in range(2):
        K, B, A = fx[f"s{s}_actions"].shape
        batch = []
        for b in range(B):
            acts = [fx[f"s{s}_actions"][k, b] for k in range(K)]
            tg = [[float(fx[f"s{s}_target_value"][k, b]), float(fx[f"s{s}_target_reward"][k, b]),
                   fx[f"s{s}_target_policy"][k, b]] for k in range(K)]
            batch.append((torch.tensor(fx[f"s{s}_obs"][b:b + 1]), acts, tg))
        pos = np.stack([np.arange(B), np.arange(B)], axis=1)
        prio, gp = mz.train((batch, fx[f"s{s}_is_ratio"], pos))
        assert gp is pos and len(prio) == K and prio[0].shape == (B, 1)
        L = fx[f"s{s}_losses"]
        got = np.array([float(x) for x in mz.store_loss[-1]])
        if s == 0:
            assert np.all(np.abs(got - L) <= 1e-5 * np.abs(L) + 1e-7)
        else:                      # second step starts from slightly different weights (Adam step 1)
            assert np.all(np.abs(got - L) <= 5e-3 * np.abs(L) + 1e-4)
        assert np.abs(mz.weights_blob() - fx[f"s{s}_blob"]).max() <= (s + 1) * 1e-2 * 0.01 + 1e-6
    assert mz.count == 2


def test_muzero_train_drop_in():
    """Muzero.train with the reference's batch format: losses, priorities and weights after two
    steps against the reference fixture (Adam + cosine annealing, unroll 5, batch 128)."""
    import torch
    from muzero_amd.model import Muzero
    fx = load_golden("train_tag123_K5_B128.npz")
    mz = Muzero(model_structure="mlp_model", observation_space_dimensions=4,
                action_space_dimensions=2, state_space_dimensions=31, hidden_layer_dimensions=64,
                number_of_hidden_layer=0, k_hypothetical_steps=5, learning_rate=0.01, optimizer="adam",
                lr_scheduler="cosineannealinglr", num_of_epoch=10000)
    mz.set_weights(fx["blob0"])
    for s in range(2):
        K, B, A = fx[f"s{s}_actions"].shape
        batch = []
        for b in range(B):
            acts = [fx[f"s{s}_actions"][k, b] for k in range(K)]
            tg = [[float(fx[f"s{s}_target_value"][k, b]), float(fx[f"s{s}_target_reward"][k, b]),
                   fx[f"s{s}_target_policy"][k, b]] for k in range(K)]
            batch.append((torch.tensor(fx[f"s{s}_obs"][b:b + 1]), acts, tg))
        pos = np.stack([np.arange(B), np.arange(B)], axis=1)
        prio, gp = mz.train((batch, fx[f"s{s}_is_ratio"], pos))
        assert gp is pos and len(prio) == K and prio[0].shape == (B, 1)
        L = fx[f"s{s}_losses"]
        got = np.array([float(x) for x in mz.store_loss[-1]])
        if s == 0:
            assert np.all(np.abs(got - L) <= 1e-5 * np.abs(L) + 1e-7)
        else:                      # second step starts from slightly different weights (Adam step 1)
            assert np.all(np.abs(got - L) <= 2e-3 * np.abs(L) + 1e-4)
        assert np.abs(mz.weights_blob() - fx[f"s{s}_blob"]).max() <= (s + 1) * 2e-3 * 0.01 + 1e-6
    assert mz.count == 2


@pytest.mark.parametrize("loss_type", ["general_kkc", "game", "game_mmc"])
def test_muzero_train_drop_in_other_loss_types(loss_type):
    """Muzero(loss_type=...).train: the reference's KL / MSE / zero criteria (muzero_model.py:364-403)
    through the drop-in surface: the first step's store_loss entry and priorities against the
    reference's own Muzero.train on the same batch and weights."""
    import torch
    from muzero_amd.model import Muzero
    fx = load_golden(f"train_loss_{loss_type}_K3_B24.npz")
    mz = Muzero(model_structure="mlp_model", observation_space_dimensions=4, action_space_dimensions=2,
                state_space_dimensions=31, hidden_layer_dimensions=64, number_of_hidden_layer=1,
                k_hypothetical_steps=3, learning_rate=0.003, optimizer="adam", lr_scheduler=None, num_of_epoch=100,
                loss_type=loss_type)
    mz.set_weights(fx["blob0"])
    K, B, A = fx["s0_actions"].shape
    batch = []
    for b in range(B):
        acts = [fx["s0_actions"][k, b] for k in range(K)]
        tg = [[float(fx["s0_target_value"][k, b]), float(fx["s0_target_reward"][k, b]), fx["s0_target_policy"][k, b]]
              for k in range(K)]
        batch.append((torch.tensor(fx["s0_obs"][b:b + 1]), acts, tg))
    pos = np.stack([np.arange(B), np.arange(B)], axis=1)
    prio, _ = mz.train((batch, fx["s0_is_ratio"], pos))
    L = fx["s0_losses"]
    got = np.array([float(x) for x in mz.store_loss[-1]])
    assert np.all(np.abs(got - L) <= 1e-5 * np.abs(L) + 1e-7)
    if loss_type == "game":
        assert np.all(got[[5, 8]] == 0)                      # zero_loss on the reward terms
    ref = fx["s0_priority"]
    assert np.all(np.abs(np.stack([p[:, 0] for p in prio]) - ref) <= 1e-5 * (1 + np.abs(ref)) + 1e-3)


def test_randomised_training_sweep_within_tolerance():
    """Eight seconds of tests/sweeps/train_sweep.py: random mlp_model / lstm_model shapes, unroll lengths, batch
    sizes, loss types, targets and weights — losses, gradient and priorities of mzs_train_grad within the
    fp32 tolerances of the oracle's train_grad (the long run: profiles/r2d_train_sweep.md)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweeps"))
    import train_sweep
    out = train_sweep.run(8.0, seed=7, log=lambda s: None)
    assert out["summary"] == "all within tolerance", out
    assert out["cases"] >= 50


def test_training_refreshes_the_weight_stream():
    """The weight-stream search kernels read chunk images, not the store: a search after a training
    step must see the new weights (k_build_stream runs before every such search).  Wide shape
    (S_dim 61 / H 128 / L 1): search, train one step on a synthetic batch, search again — both
    searches == the oracle's search with the weights read back at that moment, bit for bit."""
    from muzero_amd import rng
    from muzero_amd.engine import SearchEngine
    from muzero_amd.model import blob_layout
    from muzero_amd.train import Trainer
    from oracle import mz_oracle as mzo
    shape, G, S, K, B = (4, 2, 61, 128, 1), 37, 20, 3, 32
    d = mzo.dims(*shape)
    rs = np.random.RandomState(4242)
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob0 = (rs.standard_normal(n_w) * 0.2).astype(np.float32)
    obs = rs.uniform(-1, 1, size=(G, 4)).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 9, 2, S, 0.25, True)
    eng = SearchEngine(shape, max_games=G, max_simulations=S, discount=0.997)
    eng.set_weights(blob0)
    tr = Trainer(eng, K, B, "adam", 0.01, None, 1000)

    def check(blob):
        eng.search(obs, dirichlet, noise, S, True)
        out = eng.results_host()
        ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet, noise=noise,
                         train=True, want=("visits", "root_value"))
        assert np.array_equal(out["visits"], ref["visits"])
        assert np.array_equal(np.asarray(out["root_value"]).view(np.uint32), ref["root_value"].view(np.uint32))

    check(blob0)
    acts = np.zeros((K, B, 2), np.float32)
    acts[np.arange(K)[:, None], np.arange(B)[None, :], rs.randint(0, 2, size=(K, B))] = 1.0
    pol = rs.dirichlet([1.0, 1.0], size=(K, B)).astype(np.float32)
    tr.step(rs.uniform(-1, 1, size=(B, 4)).astype(np.float32), acts,
            rs.uniform(0, 5, size=(K, B)).astype(np.float32), rs.uniform(0, 1, size=(K, B)).astype(np.float32), pol)
    blob1 = tr.weights()
    assert not np.array_equal(blob1, blob0)
    check(blob1)
    eng.close()
