"""GPU tests of the drop-in Python surface (muzero_amd.search / muzero_amd.model) against
the reference fixtures: same return types, same global-RNG consumption, same results."""
import numpy as np
import pytest
import torch

from conftest import fixture_weights, load_golden
from test_oracle_golden import REL, hidden_err, rel_err, value_err

pytestmark = pytest.mark.gpu


def model_for(tag):
    from muzero_amd.model import Muzero
    d, blob = fixture_weights(tag)
    mz = Muzero(model_structure="lstm_model" if d.model_kind == 1 else "mlp_model",
                observation_space_dimensions=d.obs_dim, action_space_dimensions=d.n_actions,
                state_space_dimensions=d.state_dim, hidden_layer_dimensions=d.hidden_dim,
                number_of_hidden_layer=d.n_mid_layers, k_hypothetical_steps=5)
    mz.set_weights(blob)
    return mz


@pytest.mark.parametrize("fname,tag,n", [("search_tag123_S11.npz", "tag123", 24),
                                         ("search_tag123_S50.npz", "tag123", 12),
                                         ("search_synA4_S50.npz", "synA4", 8),
                                         ("search_tag123_S50_notrain.npz", "tag123", 6),
                                         ("search_tag123_S0.npz", "tag123", 4),
                                         ("search_lstmA_S50.npz", "lstmA", 6)])
def test_run_is_drop_in(fname, tag, n):
    """Monte_carlo_tree_search.run: reference signature, global np.random stream."""
    from muzero_amd.search import Monte_carlo_tree_search
    fx = load_golden(fname)
    S, A, train = int(fx["S"]), int(fx["A"]), bool(fx["train"])
    mz = model_for(tag)
    mcts = Monte_carlo_tree_search(discount=float(fx["gamma"]))
    for g in range(n):
        np.random.seed(int(fx["seed"][g]))
        obs = torch.tensor(fx["obs"][g:g + 1])
        policy, root, action = mcts.run(observation=obs, model=mz, num_simulations=S, train=train)
        # the caller's stream continues where the reference's would
        rs = np.random.RandomState(int(fx["seed"][g]))
        if train and S > 0:
            rs.dirichlet([0.25] * A)
        rs.uniform(1e-7, 2e-7, size=int(fx["n_draws"][g]))
        assert np.random.random_sample() == rs.random_sample()
        assert policy.dtype == np.float64 and policy.shape == (A,)
        assert isinstance(action, np.ndarray) and list(action) == list(range(A))
        assert list(root.children.keys()) == list(range(A))
        if S == 0:
            assert rel_err(policy, fx["visits"][g]).max() < REL     # priors as float64
            assert root.visit_count == 0 and root.value() == 0
            continue
        assert np.array_equal(policy, fx["visits"][g])          # every search, near-ties included
        # what Game.store_search_statistics reads (game.py:154-168)
        vc = np.array([c.visit_count for c in root.children.values()], dtype=np.float64)
        assert np.array_equal(vc, fx["visits"][g]) and vc.sum() == S
        assert rel_err(root.value(), fx["root_value"][g]) < REL      # S >= 11: plain 1e-5
        assert isinstance(root.value(), np.floating)
        pri = [root.children[a].prior for a in range(A)]
        assert rel_err(pri, fx["root_prior"][g]).max() < REL
        assert type(pri[0]) is (np.float64 if train else np.float32)
        assert root.visit_count == S and root.expanded()
        assert tuple(root.hidden_state.shape) == (1, int(fx["S_dim"]))
        assert hidden_err(root.hidden_state.numpy()[0], fx["root_hidden"][g]).max() < REL
        # one level deeper: grandchildren exist under the most visited child
        best = root.children[int(np.argmax(vc))]
        assert best.expanded() and sum(c.visit_count for c in best.children.values()) == best.visit_count - 1


def test_run_batch_equals_run():
    from muzero_amd.search import Monte_carlo_tree_search
    fx = load_golden("search_tag123_S11.npz")
    mz = model_for("tag123")
    mcts = Monte_carlo_tree_search(discount=0.997)
    res = mcts.run_batch(fx["obs"], 11, mz, True, seeds=fx["seed"])
    assert np.array_equal(res.policy, fx["visits"])
    assert np.array_equal(res.consumed, fx["n_draws"])
    np.random.seed(int(fx["seed"][3]))
    policy, root, _ = mcts.run(torch.tensor(fx["obs"][3:4]), 11, mz, True)
    assert np.array_equal(policy, res.policy[3])
    assert root.value() == res.root(3).value()


def test_inference_wrappers_match_reference_types_and_values():
    """Muzero.*_inference (muzero_model.py:828-905): types, shapes, values."""
    mv = load_golden("model_L2_tag123.npz")
    mz = model_for("tag123")
    h = mz.representation_function_inference(torch.tensor(mv["obs"][:1]))
    assert isinstance(h, torch.Tensor) and h.device.type == "cpu" and tuple(h.shape) == (1, 31)
    assert hidden_err(h.numpy(), mv["repr_hidden"][:1]).max() < REL
    h2 = mz.representation_function_inference(mv["obs"][0])          # ndarray input (tensor_test)
    assert torch.equal(h, h2)
    r, nh = mz.dynamics_function_inference(torch.tensor(mv["hidden"][:1]), int(mv["action"][0]))
    assert isinstance(r, np.float32) and tuple(nh.shape) == (1, 31) and nh.device.type == "cpu"
    assert value_err(r, mv["dyn_reward"][0]) <= 1
    assert hidden_err(nh.numpy(), mv["dyn_hidden"][:1]).max() < REL
    p, v = mz.prediction_function_inference(torch.tensor(mv["hidden"][:1]))
    assert isinstance(p, np.ndarray) and p.shape == (1, 2) and p.dtype == np.float32
    assert isinstance(v, np.float32)
    assert rel_err(p, mv["pred_policy"][:1]).max() < REL and value_err(v, mv["pred_value"][0]) <= 1
    p, v = mz.prediction_function_inference(torch.tensor(mv["hidden"][:5]), reanalyze=True)
    assert v.shape == (5,) and p.shape == (5, 2)


def test_errors_are_loud():
    from muzero_amd import _lib
    from muzero_amd.engine import SearchEngine
    eng = SearchEngine((4, 2, 31, 64, 0), 4, 10)
    obs = np.zeros((4, 4), np.float32)
    noise = np.zeros((4, 110), np.float64)
    with pytest.raises(_lib.MzsError, match="mzs_set_weights"):
        eng.search(obs, np.zeros((4, 2)), noise, 10, True)
    eng.set_weights(fixture_weights("tag123")[1])
    with pytest.raises(_lib.MzsError, match="noise stream"):
        eng.search(obs, np.zeros((4, 2)), noise[:, :50], 10, True)
    with pytest.raises(_lib.MzsError, match="max_simulations"):
        eng.search(obs, np.zeros((4, 2)), noise, 11, True)
    with pytest.raises(_lib.MzsError, match="dirichlet"):
        eng.search(obs, None, noise, 10, True)
    with pytest.raises(_lib.MzsError, match="weight blob"):
        eng.set_weights(np.zeros(7, np.float32))
    eng.close()
