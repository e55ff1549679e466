"""CPU tests of the host-side logic (no GPU, no HIP calls)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import ROOT, fixture_weights, load_golden


def test_pbc_table_matches_fixture():
    from muzero_amd import rng
    fx = load_golden("search_tag123_S50.npz")
    assert np.array_equal(rng.pbc_table(50), fx["pbc_table"])


def test_noise_streams_equal_reference_consumption():
    """RandomState(seed): Dirichlet then uniforms == what the reference drew from the global
    stream after np.random.seed(seed) (recorded in the fixture)."""
    from muzero_amd import rng
    fx = load_golden("search_tag123_S11.npz")
    dirichlet, noise = rng.noise_streams(fx["seed"], 2, 11, 0.25, True)
    off = fx["draw_offsets"]
    for g in range(len(fx["seed"])):
        n = int(fx["n_draws"][g])
        assert np.array_equal(noise[g, :n], fx["draws"][off[g]:off[g] + n])
    assert np.array_equal(dirichlet, fx["dirichlet"])
    assert noise.shape[1] == rng.max_draws(2, 11) == 2 * 11 * 12 // 2
    # no-train: no Dirichlet draw, the uniforms start the stream
    d2, n2 = rng.noise_streams(fx["seed"][:2], 2, 11, 0.25, False)
    assert not d2.any()
    assert n2[0, 0] == np.random.RandomState(int(fx["seed"][0])).uniform(1e-7, 2e-7)


def test_sequential_draws_equal_batched_draw():
    a = np.random.RandomState(5)
    b = np.random.RandomState(5)
    seq = np.array([a.uniform(low=1e-7, high=2e-7, size=1)[0] for _ in range(100)])
    assert np.array_equal(seq, b.uniform(1e-7, 2e-7, size=100))


def test_blob_layout_and_model_wrapper():
    from muzero_amd.model import Muzero, blob_layout
    for tag in ("tag123", "tag133", "synA4"):
        d, blob = fixture_weights(tag)
        dims = (d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers)
        assert sum(int(np.prod(s)) for _, s in blob_layout(*dims)) == blob.size
    mz = Muzero(observation_space_dimensions=4, action_space_dimensions=2, state_space_dimensions=31,
                hidden_layer_dimensions=64, number_of_hidden_layer=0, k_hypothetical_steps=5)
    assert mz.dims == (4, 2, 31, 64, 0, 0)
    assert mz.action_dictionnary == [0, 1] and mz.action_dimension == 2
    w = mz.weights_blob()
    assert w.size == 12734 and abs(w.std() - 1 / 137.035999) < 5e-4   # weights_init
    named = mz.named_weights()
    assert named["d_in_w"].shape == (64, 33) and named["p_pol_w"].shape == (2, 64)
    v = mz.weights_version
    mz.set_weights(fixture_weights("tag123")[1])
    assert mz.weights_version == v + 1
    with pytest.raises(ValueError):
        mz.set_weights(np.zeros(10, np.float32))
    with pytest.raises(NotImplementedError):
        Muzero(model_structure="vision_conv_lstm_model", observation_space_dimensions=4, action_space_dimensions=2)
    vz = Muzero(model_structure="vision_model", observation_space_dimensions=(210, 160, 3), action_space_dimensions=4,
                state_space_dimensions=21, hidden_layer_dimensions=32, number_of_hidden_layer=2)
    assert vz.dims == (98, 4, 21, 32, 2, 2) and vz.is_RGB and vz.observation_dimension == (98, 98, 3)
    vw = vz.named_weights()
    assert vw["r_ds_conv_in"].shape == (1, 3, 3, 3) and vw["d_conv"].shape == (3, 4, 3, 3)
    assert vw["p_p_out_w"].shape == (4, 32) and vw["d_fc_in_w"].shape == (32, 147) and "d_fc_mid_w" in vw
    assert np.all(vw["r_ds_res_in_bn_var"] == 1) and np.all(vw["p_res_bn_b"] == 0)
    lz = Muzero(model_structure="lstm_model", observation_space_dimensions=4, action_space_dimensions=2,
                state_space_dimensions=31, hidden_layer_dimensions=64, number_of_hidden_layer=2)
    assert lz.dims == (4, 2, 31, 64, 2, 1)
    nw = lz.named_weights()
    assert nw["d_rew_w_ih_l1"].shape == (124, 31) and nw["p_pol_w_hh_l0"].shape == (8, 2)
    assert np.abs(nw["p_val_w_ih_l0"]).max() <= 1 / np.sqrt(31) and abs(nw["r_w"].std() - 1 / 137.035999) < 2e-3
    lz.loss_type = "kldiv"           # train(): not one of the reference's loss types
    with pytest.raises(NotImplementedError):
        lz.train(None)


def test_model_checkpoint_roundtrip(tmp_path):
    from muzero_amd.model import Muzero
    mz = Muzero(observation_space_dimensions=4, action_space_dimensions=2, state_space_dimensions=31,
                hidden_layer_dimensions=64, number_of_hidden_layer=0, k_hypothetical_steps=5)
    mz.save_model(directory=str(tmp_path), tag=77, model_update_condition=True)
    iv = json.load(open(tmp_path / "77_muzero_init_variables.json"))
    assert iv["state_space_dimensions"] == 31 and iv["action_map"] == [0, 1]
    m2 = Muzero(load=True)
    m2.load_model(model_directory=str(tmp_path), tag=77)
    assert m2.dims == mz.dims and np.array_equal(m2.weights_blob(), mz.weights_blob())


def test_mcts_attributes_and_reset():
    from muzero_amd.search import Monte_carlo_tree_search
    m = Monte_carlo_tree_search(discount=0.997)
    assert (m.pb_c_base, m.pb_c_init, m.discount, m.root_dirichlet_alpha,
            m.root_exploration_fraction) == (19652, 1.25, 0.997, 0.25, 0.25)
    m.reset()
    assert m.discount == 0.95 and m.node is None and m.model is None and m.overall_graph == []


def test_config_schema():
    from muzero_amd import config
    cfg = config.load_config(os.path.join(ROOT, "configs", "cartpole_mi355x.json"))
    assert cfg["mi355x"]["concurrent_games"] == 4096
    assert config.mcts_kwargs(cfg)["discount"] == 0.997
    kw = config.muzero_kwargs(cfg, 4, 2)
    assert kw["state_space_dimensions"] == 31 and kw["k_hypothetical_steps"] == 5
    # the reference's own config files (same schema, no mi355x section) load unchanged
    for path in sorted(glob.glob("/root/reference/config/experiment_*.json")):
        c = config.load_config(path)
        assert c["mi355x"] == config.MI355X_DEFAULTS
        config.mcts_kwargs(c), config.muzero_kwargs(c, 4, 2)


def test_lstm_blob_layout_matches_header_and_fixture():
    """'lstm_model' blob: the Python layout, the C header's mzs_lstm_layout (through the oracle's
    blob size check) and the reference's state_dict sizes (fixture) agree."""
    from muzero_amd.model import blob_layout
    for tag in ("lstmA", "lstmB"):
        w = load_golden(f"weights_{tag}.npz")
        dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (1,)
        lay = blob_layout(*dims)
        assert sum(int(np.prod(s)) for _, s in lay) == w["blob"].size
        names = [n for n, _ in lay]
        assert names[:4] == ["r_w", "r_b", "d_rew_lin_w", "d_rew_lin_b"] and names[4] == "d_rew_w_ih_l0"
