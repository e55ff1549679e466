"""Training step of the 'vision_model' on the GPU (muzero_amd/csrc/mzs_vision_train.hip) through the C-ABI:
against the float64 oracle (oracle/mz_vision_train_oracle.py) and against the reference's own Muzero.train
(tests/golden/train_vis*.npz: BatchNorm on batch statistics, the Dropout2d masks the reference drew injected
through mzs_train_set_dropout).

Floating-point bar (north_star: 1e-5 for fp32): loss terms 1e-5 relative; a tensor's gradient
|d| <= 1e-5 x the blob's largest gradient + 1e-4 x the tensor's own largest (the bound the oracle meets against
the float32 reference: tests/test_vision_train_oracle.py); priorities 1e-5 + one grid step of the float32
inverse transform; weights and running statistics after the step 1e-5 absolute + the gradient bound x the rate.
"""
import numpy as np
import pytest

from conftest import load_golden
from test_vision_train_oracle import CASES, check_grad, priority_close, step_inputs, vo

pytestmark = pytest.mark.gpu


def make_trainer(fx, max_batch=None, max_games=4, max_simulations=0):
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import Trainer
    obs, A, Sd, H, L = [int(v) for v in fx["dims"]]
    eng = SearchEngine((obs, A, Sd, H, L, 2), max_games=max_games, max_simulations=max_simulations, discount=0.997)
    eng.set_weights(fx["blob0"])
    tr = Trainer(eng, int(fx["K"]), max_batch or int(fx["B"]), str(fx["optimizer"]), float(fx["lr"]))
    return eng, tr


def run_step(tr, kw, inject=True):
    if inject:
        tr.set_dropout(kw["masks"])
    prio = tr.step(kw["obs"], kw["actions"], kw["tv"], kw["tr"], kw["tpol"], kw["is_ratio"])
    import torch
    torch.cuda.synchronize()
    return dict(grad=tr.grad.cpu().numpy().astype(np.float64), losses=tr.loss.cpu().numpy().astype(np.float64),
                priority=prio.cpu().numpy().astype(np.float64), blob=tr.weights())


@pytest.mark.parametrize("case", CASES)
def test_vision_train_step_vs_oracle_and_reference(case):
    fx = load_golden(case + ".npz")
    dims, K, B = fx["dims"], int(fx["K"]), int(fx["B"])
    eng, tr = make_trainer(fx)
    assert tr.dropout_count(B) == fx["s0_masks"].size
    blob = fx["blob0"]
    opt = vo.Optimizer(str(fx["optimizer"]), blob.size, dims)
    stat = opt.is_stat
    for s in range(int(fx["n_steps"])):
        kw = step_inputs(fx, s)
        got = run_step(tr, kw)
        want = vo.train_grad(blob, dims, K, **kw)
        # against the oracle
        assert np.abs(got["losses"] - want["losses"]).max() <= 1e-5 * np.abs(want["losses"]).max()
        assert priority_close(got["priority"], want["priority"], kw["tv"])
        g = got["grad"].copy()
        g[stat] = 0.0                                   # those slots carry the running statistics' change
        check_grad(dims, g, want["grad"], f"{case} step {s} vs oracle")
        # up to fifteen float32 moving-average updates per statistic against the oracle's float64 ones
        assert np.abs(blob[stat] + got["grad"][stat] - want["running"][stat]).max() <= 5e-6
        # against the reference itself
        ref = fx[f"s{s}_losses"]
        assert abs(got["losses"][-1] - ref[0]) <= 1e-5 * abs(ref[0])
        assert np.abs(got["losses"][:-1] - ref[1:]).max() <= 1e-5 * np.abs(ref[1:]).max()
        assert priority_close(got["priority"], fx[f"s{s}_priority"], kw["tv"])
        check_grad(dims, g, fx[f"s{s}_grad"], f"{case} step {s} vs reference")
        # weights after the step: the gradient's bound times the rate (SGD: lr 0.05 on gradients of ~17)
        wtol = 1e-5 + 2.0 * float(fx["lr"]) * (s + 1) * 1e-5 * max(1.0, np.abs(fx[f"s{s}_grad"]).max())
        assert np.abs(got["blob"] - fx[f"s{s}_blob"]).max() <= wtol
        blob = opt.apply(blob, want["grad"], want["running"], float(fx["lr"]))
        assert np.abs(got["blob"] - blob).max() <= wtol
        blob = got["blob"]                               # the next step starts from the device's weights
        opt.m, opt.v = opt.m, opt.v
    eng.close()


def test_vision_training_updates_what_the_search_reads():
    """After mzs_train_apply the inference and search kernels read the new weights (convolutions and
    batchnorms from the blob, the Linear layers from their two refreshed stores) and the new running
    statistics: inference and a whole search == the oracle on the read-back blob, bit for bit."""
    from oracle import mz_oracle as mzo
    fx = load_golden(CASES[0] + ".npz")
    obs_dim, A, Sd, H, L = [int(v) for v in fx["dims"]]
    eng, tr = make_trainer(fx, max_games=8, max_simulations=12)
    kw = step_inputs(fx, 0)
    got = run_step(tr, kw)
    blob = got["blob"]
    assert not np.array_equal(blob, fx["blob0"])
    d = mzo.dims(obs_dim, A, Sd, H, L, 2)
    obs = kw["obs"]
    hid, pol, val = [t.cpu().numpy() for t in eng.initial_inference(obs)]
    oh = mzo.representation(d, blob, obs)
    op, ov = mzo.prediction(d, blob, oh)
    assert np.array_equal(hid.view(np.uint32), oh.view(np.uint32))
    assert np.array_equal(pol.view(np.uint32), op.view(np.uint32))
    assert np.array_equal(val.view(np.uint32), ov.view(np.uint32))
    G, S = obs.shape[0], 12
    seeds = np.arange(G, dtype=np.int64) + 77
    eng.search_streams(obs, seeds, S, True)
    out = eng.results_host()
    dirichlet, noise = eng.fill_streams(seeds, S, True)
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet.cpu().numpy(),
                     noise=noise.cpu().numpy(), train=True, want=("visits", "root_value"))
    assert np.array_equal(out["visits"], ref["visits"])
    assert np.array_equal(out["root_value"].view(np.uint32), ref["root_value"].view(np.uint32))
    eng.close()


def test_vision_device_dropout_is_seeded_and_partial_batches_work():
    """Without injected masks the device draws them from (seed, call number): two trainers with one seed
    stay bit-identical over two steps, another seed gives other weights; a batch smaller than max_batch runs
    (and equals the same batch on a trainer sized for it: the arena layout does not enter the result)."""
    fx = load_golden(CASES[1] + ".npz")
    kw0, kw1 = step_inputs(fx, 0), step_inputs(fx, 1)
    blobs = []
    for seed, mb in ((5, None), (5, 7), (6, None)):
        eng, tr = make_trainer(fx, max_batch=mb)
        tr.set_dropout(None, seed=seed)
        run_step(tr, kw0, inject=False)
        blobs.append(run_step(tr, kw1, inject=False)["blob"])
        eng.close()
    assert np.array_equal(blobs[0], blobs[1])
    assert not np.array_equal(blobs[0], blobs[2])
    assert np.isfinite(blobs[2]).all()


def test_vision_wrong_mask_count_is_refused():
    from muzero_amd import _lib
    fx = load_golden(CASES[1] + ".npz")
    eng, tr = make_trainer(fx)
    kw = step_inputs(fx, 0)
    kw["masks"] = kw["masks"][:-3]
    with pytest.raises(_lib.MzsError):
        run_step(tr, kw)
    eng.close()


def test_vision_muzero_train_drop_in():
    """Muzero(model_structure='vision_model').train(batch) with the reference's batch format; the masks come
    from the device generator, so the check is on behaviour: losses finite and falling over a few steps on one
    batch, priorities shaped as the reference's, the search running on the trained weights."""
    import torch
    from muzero_amd.model import Muzero
    from muzero_amd.search import Monte_carlo_tree_search
    fx = load_golden(CASES[0] + ".npz")
    _, A, Sd, H, L = [int(v) for v in fx["dims"]]
    K, B = int(fx["K"]), int(fx["B"])
    mz = Muzero(model_structure="vision_model", observation_space_dimensions=(400, 600, 3), action_space_dimensions=A,
                state_space_dimensions=Sd, hidden_layer_dimensions=H, number_of_hidden_layer=L, k_hypothetical_steps=K,
                learning_rate=0.01, optimizer="adam")
    mz.set_weights(fx["blob0"])
    kw = step_inputs(fx, 0)
    batch = []
    for b in range(B):
        acts = [kw["actions"][k, b].astype(np.float64) for k in range(K)]
        tg = [[float(kw["tv"][k, b]), float(kw["tr"][k, b]), kw["tpol"][k, b].astype(np.float64)] for k in range(K)]
        batch.append((torch.tensor(kw["obs"][b:b + 1]), acts, tg))
    pos = np.stack([np.arange(B), np.arange(B)], axis=1)
    first = None
    for _ in range(12):
        prio, gp = mz.train((batch, np.array([0]), pos))
        loss = float(mz.store_loss[-1][0])
        assert np.isfinite(loss)
        first = loss if first is None else first
    assert len(prio) == K and prio[0].shape == (B, 1) and gp is pos
    assert loss < first
    res = Monte_carlo_tree_search(discount=0.997).run_batch(kw["obs"][:3], 6, mz, True, seeds=np.arange(3))
    assert np.all(res.policy.sum(axis=1) == 6)


def random_vision_blob(rs, A, Sd, H, L):
    parts = []
    for name, shape in vo.blob_layout(A, Sd, H, L):
        n = int(np.prod(shape))
        if name.endswith("_var"):
            v = 0.5 + rs.uniform(size=n)
        elif name.endswith("_mean"):
            v = 0.3 * rs.standard_normal(n)
        elif name.endswith("bn_w"):
            v = 0.5 + rs.uniform(size=n)
        elif name.endswith(("_b", "bn_b")):
            v = 0.1 * rs.standard_normal(n)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rs.standard_normal(n) * (1.3 / np.sqrt(fan_in))
        parts.append(v.astype(np.float32))
    return np.concatenate(parts)


@pytest.mark.parametrize("seed", range(8))
def test_vision_training_random_shapes_vs_oracle(seed):
    """Random head shapes, unroll lengths (K = 1: only forward-only dynamics), batch sizes (B = 1: batch statistics
    of one sample; B = 96 and 257: every reduction in several chunks, finished by the waves of the next kernel), the
    four loss types, with and without importance weights, absorbing actions — one step against the oracle, masks
    drawn here and injected into both."""
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import LOSS_TYPES, Trainer
    rs = np.random.RandomState(1000 + seed)
    A, Sd, H, L = int(rs.randint(2, 7)), int(rs.randint(3, 32)), int(rs.choice([8, 12, 33, 64])), int(rs.randint(1, 3))
    K, B = (1, 5) if seed == 0 else (int(rs.randint(2, 5)), 1 if seed == 1 else int(rs.randint(2, 11)))
    if seed >= 6:
        K, B = 2, (96, 257)[seed - 6]
    loss_type = ["general", "general", "game_mmc", "game", "general_kkc", "general", "general", "game_mmc"][seed]
    blob = random_vision_blob(rs, A, Sd, H, L)
    obs = rs.uniform(0.0, 1.0, size=(B, 3, 98, 98)).astype(np.float32)
    actions = np.zeros((K, B, A), np.float32)
    actions[np.arange(K)[:, None], np.arange(B)[None, :], rs.randint(0, A, size=(K, B))] = 1.0
    actions[K - 1:, 0] = 0.0                                                      # an absorbing tail
    tv = rs.uniform(0.0 if loss_type == "general_kkc" else -3.0, 10.0, size=(K, B)).astype(np.float32)
    tr = rs.choice([0.0, 1.0, 2.5], size=(K, B)).astype(np.float32)
    tpol = rs.dirichlet([0.7] * A, size=(K, B)).astype(np.float32)
    isr = rs.uniform(0.3, 1.4, size=B).astype(np.float32) if seed % 2 else None
    eng = SearchEngine((98, A, Sd, H, L, 2), max_games=1, max_simulations=0)
    eng.set_weights(blob)
    tr_ = Trainer(eng, K, B + (seed % 3), "adam", 0.01, loss_type=loss_type)
    masks = (rs.uniform(size=tr_.dropout_count(B)) < 0.5).astype(np.uint8)
    tr_.set_dropout(masks)
    prio = tr_.step(obs, actions, tv, tr, tpol, isr)
    import torch
    torch.cuda.synchronize()
    lv, lr_ = LOSS_TYPES[loss_type]
    want = vo.train_grad(blob, (98, A, Sd, H, L), K, obs, actions, tv, tr, tpol, isr, masks, loss_value=lv,
                         loss_reward=lr_)
    got_loss = tr_.loss.cpu().numpy().astype(np.float64)
    assert np.abs(got_loss - want["losses"]).max() <= 1e-5 * max(1.0, np.abs(want["losses"]).max())
    assert priority_close(prio.cpu().numpy().astype(np.float64), want["priority"], tv)
    stat = vo.Optimizer("adam", blob.size, (98, A, Sd, H, L)).is_stat
    g = tr_.grad.cpu().numpy().astype(np.float64)
    assert np.abs(blob[stat] + g[stat] - want["running"][stat]).max() <= 5e-6
    g[stat] = 0.0
    # MSE criteria differentiate through softmax probabilities near 0/1: same bound, seen 1e-6
    check_grad((98, A, Sd, H, L), g, want["grad"], f"seed {seed} ({loss_type}, A {A} Sd {Sd} H {H} L {L} K {K} B {B})")
    eng.close()
