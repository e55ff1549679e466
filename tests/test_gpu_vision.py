"""'vision_model' (ResNet-v2, neural_network_vision_model.py) on the GPU: the three inference entry
points and a whole search through the C-ABI against the CPU oracle (bit for bit: same arithmetic
contract, oracle/mz_vision_oracle.c) and against the reference's own outputs (1e-5)."""
import numpy as np
import pytest

from conftest import fixture_weights, load_golden
from test_vision_oracle import VISION_MODELS, close, value_close, vision_obs

pytestmark = pytest.mark.gpu


def engine_for(d, max_games, max_simulations=0, **kw):
    from muzero_amd.engine import SearchEngine
    return SearchEngine((d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers, 2),
                        max_games=max_games, max_simulations=max_simulations, **kw)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", VISION_MODELS)
def test_vision_inference_equals_oracle_and_reference(name):
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(name)
    fx = load_golden(f"model_L2_{name}.npz")
    n = fx["hidden"].shape[0]
    obs = vision_obs(fx, n)
    eng = engine_for(d, n)
    eng.set_weights(blob)
    # initial inference: representation (Down_sample + residual block) + prediction
    hid, pol, val = [t.cpu().numpy() for t in eng.initial_inference(obs)]
    oh = mzo.representation(d, blob, obs)
    op, ov = mzo.prediction(d, blob, oh)
    assert np.array_equal(bits(hid), bits(oh))
    assert np.array_equal(bits(pol), bits(op)) and np.array_equal(bits(val), bits(ov))
    assert close(hid, fx["repr_hidden"].reshape(n, -1)) and close(pol, fx["ini_policy"])
    assert value_close(val, fx["ini_value"])
    # recurrent inference on the fixture's hidden states
    h = fx["hidden"].reshape(n, -1)
    rew, nh, pol, val = [t.cpu().numpy() for t in eng.recurrent_inference(h, fx["action"])]
    orew, onh = mzo.dynamics(d, blob, h, fx["action"])
    op, ov = mzo.prediction(d, blob, onh)
    assert np.array_equal(bits(rew), bits(orew)) and np.array_equal(bits(nh), bits(onh))
    assert np.array_equal(bits(pol), bits(op)) and np.array_equal(bits(val), bits(ov))
    assert close(nh, fx["dyn_hidden"].reshape(n, -1)) and value_close(rew, fx["dyn_reward"])
    assert close(pol, fx["rec_policy"]) and value_close(val, fx["rec_value"])
    # prediction alone
    pol, val = [t.cpu().numpy() for t in eng.prediction_inference(h)]
    op, ov = mzo.prediction(d, blob, h)
    assert np.array_equal(bits(pol), bits(op)) and np.array_equal(bits(val), bits(ov))
    assert close(pol, fx["pred_policy"]) and value_close(val, fx["pred_value"])
    eng.close()


@pytest.mark.parametrize("engine", ["fused", "generic"])
@pytest.mark.parametrize("name,G,S", [("visA", 21, 30), ("visB", 9, 12), ("visA", 4, 5), ("visB", 1, 7)])
def test_vision_search_equals_oracle(name, G, S, engine):
    """Whole searches (Monte_carlo_tree_search.run on the vision networks) through mzs_search_streams:
    visit counts, root values, consumed draws and the hidden state of every expansion == the oracle's
    search on the same images and streams, bit for bit.  engine "fused": k_search_vis4 (four games per block,
    the head MLPs on the matrix core; ragged game counts leave waves that replay the last game); "generic":
    k_search_fused (one game per wavefront, lane-per-output Linear layers) — two implementations, one oracle."""
    from muzero_amd import _lib, rng
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(name)
    obs = np.random.RandomState(7).uniform(0.0, 1.0, size=(G, 3, 98, 98)).astype(np.float32)
    seeds = np.arange(G, dtype=np.int64) + 321
    eng = engine_for(d, G, S, discount=0.997,
                     engine=_lib.ENGINE_FUSED if engine == "fused" else _lib.ENGINE_FUSED_GENERIC)
    eng.set_weights(blob)
    assert eng.search_kernel_name(S) == ("k_search_vis4" if engine == "fused" else "k_search_fused<false,vision>")
    eng.search_streams(obs, seeds, S, True)
    out, tree = eng.results_host(), eng.tree_host()
    dirichlet, noise = eng.fill_streams(seeds, S, True)
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet.cpu().numpy(),
                     noise=noise.cpu().numpy(), train=True, want=("visits", "root_value", "consumed", "hidden"))
    assert np.array_equal(out["visits"], ref["visits"])
    assert np.array_equal(bits(out["root_value"]), bits(ref["root_value"]))
    assert np.array_equal(out["consumed"], ref["consumed"])
    assert np.array_equal(bits(tree["hidden"]), bits(ref["hidden"]))
    assert out["visits"].sum() == G * S and np.all(tree["status"] == 0)
    assert G == 1 or len({tuple(v) for v in out["visits"]}) > 1          # the searches are not all alike
    eng.close()


@pytest.mark.parametrize("engine", ["fused", "generic"])
def test_vision_search_many_games_equals_oracle(engine):
    """More games than two blocks of k_search_vis4 per CU hold (8 x CUs + 256): search_launch switches to the
    build of k_search_fused compiled for four waves per SIMD (128 registers, 32-row Linear batches), under
    either engine.  Results — copies of twelve distinct games — equal the oracle's for those twelve, bit for
    bit, in every copy."""
    import torch
    from muzero_amd import _lib, rng  # noqa: F401
    from oracle import mz_oracle as mzo
    name, K, S = "visA", 12, 12
    d, blob = fixture_weights(name)
    G = 8 * torch.cuda.get_device_properties(0).multi_processor_count + 256
    obs_k = np.random.RandomState(11).uniform(0.0, 1.0, size=(K, 3, 98, 98)).astype(np.float32)
    seeds_k = np.arange(K, dtype=np.int64) + 77
    idx = np.arange(G) % K
    eng = engine_for(d, G, S, discount=0.997,
                     engine=_lib.ENGINE_FUSED if engine == "fused" else _lib.ENGINE_FUSED_GENERIC)
    eng.set_weights(blob)
    eng.search_streams(torch.as_tensor(obs_k).cuda()[torch.as_tensor(idx).cuda()], seeds_k[idx], S, True)
    out, tree = eng.results_host(), eng.tree_host()
    dirichlet, noise = eng.fill_streams(seeds_k, S, True)
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs_k, dirichlet=dirichlet.cpu().numpy(),
                     noise=noise.cpu().numpy(), train=True, want=("visits", "root_value", "consumed", "hidden"))
    assert np.array_equal(out["visits"], ref["visits"][idx])
    assert np.array_equal(bits(out["root_value"]), bits(ref["root_value"][idx]))
    assert np.array_equal(out["consumed"], ref["consumed"][idx])
    assert np.array_equal(bits(tree["hidden"]), bits(ref["hidden"][idx]))
    assert np.all(tree["status"] == 0)
    eng.close()


def test_vision_search_two_blocks_per_cu_equals_oracle():
    """k_search_vis4 with two resident blocks on every CU (8 x CUs games, the largest batch it is dispatched for):
    copies of twelve distinct games, each == the oracle's, bit for bit, over two launches in a row."""
    import torch
    from oracle import mz_oracle as mzo
    name, K, S = "visB", 12, 10
    d, blob = fixture_weights(name)
    G = 8 * torch.cuda.get_device_properties(0).multi_processor_count
    obs_k = np.random.RandomState(13).uniform(0.0, 1.0, size=(K, 3, 98, 98)).astype(np.float32)
    seeds_k = np.arange(K, dtype=np.int64) + 177
    idx = np.arange(G) % K
    eng = engine_for(d, G, S, discount=0.997)
    assert eng.search_kernel_name(S) == "k_search_vis4"
    dirichlet, noise = eng.fill_streams(seeds_k, S, True)
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs_k, dirichlet=dirichlet.cpu().numpy(),
                     noise=noise.cpu().numpy(), train=True, want=("visits", "root_value", "consumed", "hidden"))
    eng.set_weights(blob)
    for _ in range(2):
        eng.search_streams(torch.as_tensor(obs_k).cuda()[torch.as_tensor(idx).cuda()], seeds_k[idx], S, True)
        out, tree = eng.results_host(), eng.tree_host()
        assert np.array_equal(out["visits"], ref["visits"][idx])
        assert np.array_equal(bits(out["root_value"]), bits(ref["root_value"][idx]))
        assert np.array_equal(out["consumed"], ref["consumed"][idx])
        assert np.array_equal(bits(tree["hidden"]), bits(ref["hidden"][idx]))
        assert np.all(tree["status"] == 0)
    eng.close()


@pytest.mark.parametrize("A,Sd,H,L", [(3, 17, 70, 1), (2, 9, 30, 2), (5, 31, 130, 1)])
def test_vision_odd_head_shapes_equal_oracle(A, Sd, H, L):
    """Head MLPs whose width is neither a multiple of four (the weight store's last rows behind the quads)
    nor within one wavefront (several passes over the outputs), three to five actions: inference and a short
    search on seeded random weights == oracle, bit for bit (no reference run of these shapes: parity vs the
    oracle only, which tests/test_vision_oracle.py pins to the reference on the fixture shapes)."""
    from muzero_amd.engine import SearchEngine
    from muzero_amd.model import vision_blob_layout
    from oracle import mz_oracle as mzo
    rs = np.random.RandomState(A * 1000 + H)
    parts = []
    for name, shp in vision_blob_layout(A, Sd, H, L):
        n = int(np.prod(shp))
        if name.endswith("bn_var"):
            v = rs.uniform(0.5, 1.5, n)
        elif name.endswith("bn_w"):
            v = rs.uniform(0.8, 1.2, n)
        elif "conv" in name and not name.endswith("_b"):
            v = rs.standard_normal(n) * 0.35
        else:
            v = rs.standard_normal(n) * (1.0 / np.sqrt(max(shp[-1], 1)) if len(shp) == 2 else 0.1)
        parts.append(v.astype(np.float32))
    blob = np.concatenate(parts)
    d = mzo.dims(98, A, Sd, H, L, 2)
    G, S = 7, 9
    obs = rs.uniform(0.0, 1.0, size=(G, 3, 98, 98)).astype(np.float32)
    eng = SearchEngine((98, A, Sd, H, L, 2), max_games=G, max_simulations=S, discount=0.997)
    eng.set_weights(blob)
    hid, pol, val = [t.cpu().numpy() for t in eng.initial_inference(obs)]
    oh = mzo.representation(d, blob, obs)
    op, ov = mzo.prediction(d, blob, oh)
    assert np.array_equal(bits(hid), bits(oh)) and np.array_equal(bits(pol), bits(op))
    assert np.array_equal(bits(val), bits(ov)) and np.isfinite(ov).all()
    act = (np.arange(G) % A).astype(np.int32)
    rew, nh, pol, val = [t.cpu().numpy() for t in eng.recurrent_inference(oh, act)]
    orew, onh = mzo.dynamics(d, blob, oh, act)
    op, ov = mzo.prediction(d, blob, onh)
    assert np.array_equal(bits(rew), bits(orew)) and np.array_equal(bits(nh), bits(onh))
    assert np.array_equal(bits(pol), bits(op)) and np.array_equal(bits(val), bits(ov))
    seeds = np.arange(G, dtype=np.int64) + 5
    eng.search_streams(obs, seeds, S, True)
    out = eng.results_host()
    dirichlet, noise = eng.fill_streams(seeds, S, True)
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet.cpu().numpy(),
                     noise=noise.cpu().numpy(), train=True, want=("visits", "root_value", "consumed"))
    assert np.array_equal(out["visits"], ref["visits"])
    assert np.array_equal(bits(out["root_value"]), bits(ref["root_value"]))
    assert np.array_equal(out["consumed"], ref["consumed"])
    eng.close()


def test_vision_muzero_drop_in():
    """Muzero(model_structure='vision_model'): the reference's wrapper surface (muzero_model.py:828-905)
    — (1, 3, 98, 98) observations, (1, 3, 7, 7) hidden states — against the reference's own outputs, a
    batched search through Monte_carlo_tree_search.run_batch (train(): tests/test_gpu_vision_train.py)."""
    import torch
    from muzero_amd.model import Muzero
    from muzero_amd.search import Monte_carlo_tree_search
    d, blob = fixture_weights("visA")
    fx = load_golden("model_L2_visA.npz")
    mz = Muzero(model_structure="vision_model", observation_space_dimensions=(400, 600, 3),
                action_space_dimensions=2, state_space_dimensions=31, hidden_layer_dimensions=64,
                number_of_hidden_layer=1, k_hypothetical_steps=5)
    assert mz.is_RGB and mz.observation_dimension == (98, 98, 3) and mz.dims == (98, 2, 31, 64, 1, 2)
    fresh = mz.named_weights()                      # torch's default initialisation, not weights_init
    assert np.all(fresh["r_res_bn_w"] == 1) and np.all(fresh["d_bn_var"] == 1) and np.all(fresh["p_res_bn_mean"] == 0)
    assert np.abs(fresh["d_fc_in_w"]).max() <= 1 / np.sqrt(147) and np.abs(fresh["d_conv"]).max() <= 1 / 6
    mz.set_weights(blob)
    obs = vision_obs(fx, fx["hidden"].shape[0])
    h0 = mz.representation_function_inference(torch.tensor(obs[:1]))
    assert tuple(h0.shape) == (1, 3, 7, 7) and close(h0.numpy(), fx["repr_hidden"][:1])
    pol, val = mz.prediction_function_inference(h0)
    assert pol.shape == (1, 2) and close(pol, fx["ini_policy"][:1]) and value_close(val, fx["ini_value"][0])
    rew, h1 = mz.dynamics_function_inference(torch.tensor(fx["hidden"][3:4]), int(fx["action"][3]))
    assert tuple(h1.shape) == (1, 3, 7, 7) and close(h1.numpy(), fx["dyn_hidden"][3:4])
    assert value_close(rew, fx["dyn_reward"][3])
    res = Monte_carlo_tree_search(discount=0.997).run_batch(obs[:5], 10, mz, True, seeds=np.arange(5) + 3)
    assert res.policy.shape == (5, 2) and np.all(res.policy.sum(axis=1) == 10)
