"""GPU parity of the bf16 variants (Muzero(type_format=torch.bfloat16); include/mzs_numeric.h
"dot_bf16x3" = precision 2, the default, and "dot_bf16" = precision 1, bf16_mode="fast"): the HIP kernels — v_mfma_f32_4x4x4_16B_bf16
in k_search_quad<..., bf16>, in the weight-stream kernels and (one game per wave) in the any-shape kernels — against the bf16
oracle (whose mzs_dot4_bf16 restates that instruction), bit for bit,
through the C-ABI.  Accuracy against the reference (1e-2) is the oracle's: tests/test_bf16_oracle.py."""
import numpy as np
import pytest

from conftest import MODEL_TAGS, fixture_streams, fixture_weights, load_golden
from test_bf16_oracle import bf16_dims
from test_gpu_parity import FULL_WANT, assert_search_equals_oracle, same_bits

pytestmark = pytest.mark.gpu


PRECISIONS = [1, 2]          # MZS_PRECISION_BF16 (fast), MZS_PRECISION_BF16X3 (compensated)
SUFFIX = {1: ",bf16>", 2: ",bf16x3>"}


def make_engine(d, G, S, gamma=0.997, frac=0.25, engine="fused", prec=1):
    from muzero_amd import _lib
    from muzero_amd.engine import SearchEngine
    return SearchEngine((d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers, d.model_kind, prec),
                        max_games=G, max_simulations=S, discount=gamma, root_exploration_fraction=frac,
                        engine=_lib.ENGINE_FUSED if engine == "fused" else _lib.ENGINE_FUSED_GENERIC)


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("tag", MODEL_TAGS)
def test_bf16_inference_kernels_equal_oracle(tag, prec):
    """initial / recurrent / prediction inference (any-shape kernels: the matrix instruction with one game per wave, mzs_device.h dev_dot4_mfma),
    MLP and LSTM models, both bf16 contracts."""
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(tag)
    db = bf16_dims(d, prec)
    mv = load_golden(f"model_L2_{tag}.npz")
    eng = make_engine(d, 1, 0, prec=prec)
    eng.set_weights(blob)
    hid, pol, val = [t.cpu().numpy() for t in eng.initial_inference(mv["obs"])]
    oh, op, ov = mzo.initial_inference(db, blob, mv["obs"])
    assert same_bits(hid, oh) and same_bits(pol, op) and same_bits(val, ov)
    rew, nh, pol, val = [t.cpu().numpy() for t in eng.recurrent_inference(mv["hidden"], mv["action"])]
    orr, onh, op, ov = mzo.recurrent_inference(db, blob, mv["hidden"], mv["action"])
    assert same_bits(rew, orr) and same_bits(nh, onh) and same_bits(pol, op) and same_bits(val, ov)
    # and it is not the fp32 path in disguise
    r32 = mzo.recurrent_inference(d, blob, mv["hidden"], mv["action"])
    assert not same_bits(pol, r32[2])
    eng.close()


# tag -> the weight-stream kernel (mzs_quadw_bf16.h: bf16 stream, hi + lo stream) of the wide MLP and the one-layer LSTM
WIDE = {"tag127": "k_search_quadw2<61,2,128", "lstmA": "k_search_quadl<31,2,64"}

BF16_SEARCHES = [("search_tag123_S50.npz", "tag123"), ("search_tag123_S11.npz", "tag123"),
                 ("search_tag123_S1.npz", "tag123"), ("search_tag123_S0.npz", "tag123"),
                 ("search_tag123_S50_notrain.npz", "tag123"), ("search_tag123_S50_wide.npz", "tag123"),
                 ("search_tag133_S50.npz", "tag133"), ("search_synA4_S50.npz", "synA4"),
                 ("search_tag127_S50.npz", "tag127"), ("search_synA3_S30.npz", "synA3"),
                 ("search_lstmA_S50.npz", "lstmA"), ("search_lstmB_S30.npz", "lstmB")]


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("engine", ["fused", "generic"])
@pytest.mark.parametrize("fname,tag", BF16_SEARCHES)
def test_bf16_search_equals_oracle(fname, tag, engine, prec):
    """Whole searches on the fixtures' inputs: the MFMA kernel (tag123: L = 0; tag133: tied mid
    layers; synA4: four actions), the weight-stream kernels (tag127, lstmA) and the any-shape kernel (synA3, and every shape under
    engine="generic") == bf16 oracle on every output bit, hidden states included."""
    from oracle import mz_oracle as mzo
    fx = load_golden(fname)
    d, blob = fixture_weights(tag)
    S, G = int(fx["S"]), len(fx["seed"])
    dirichlet, noise = fixture_streams(fx)
    train = bool(fx["train"])
    eng = make_engine(d, G, S, float(fx["gamma"]), float(fx["frac"]), engine, prec)
    eng.set_weights(blob)
    name = eng.search_kernel_name(S)
    if engine == "fused" and tag in ("tag123", "tag133", "synA4"):
        assert name.startswith("k_search_quad<") and name.endswith(SUFFIX[prec]), name
    elif engine == "fused" and tag in WIDE:
        assert name == WIDE[tag] + SUFFIX[prec], name
    else:
        assert name == "k_search_fused<false>", name
    eng.search(fx["obs"], dirichlet if train and S > 0 else None, noise if S > 0 else None, S, train)
    out, tree = eng.results_host(), eng.tree_host()
    ref = mzo.search(bf16_dims(d, prec), blob, S, float(fx["gamma"]), float(fx["frac"]), fx["pbc_table"], obs=fx["obs"],
                     dirichlet=dirichlet, noise=noise, train=train, want=FULL_WANT)
    assert not tree["status"].any()
    assert_search_equals_oracle(out, tree, ref)
    eng.close()


@pytest.mark.parametrize("shape", [(8, 4, 31, 64, 0), (4, 2, 31, 64, 0), (4, 2, 31, 64, 4), (8, 4, 31, 64, 1),
                                   (64, 2, 31, 64, 0), (1, 2, 31, 64, 2), (6, 3, 31, 64, 0), (6, 3, 31, 64, 2),
                                   (4, 2, 11, 16, 0), (3, 2, 11, 16, 3)])
@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("G,S", [(37, 50), (5, 7), (1, 50), (11, 120)])
def test_bf16_quad_shapes_vs_oracle(shape, G, S, prec):
    """Every shape the bf16 MFMA kernel is instantiated for, ragged game counts (partial waves /
    blocks), observation widths 1 .. 64 (partial and full groups of four in the input layer), a
    deep search (hidden states in the global array), seeded random weights: == oracle on every bit."""
    from muzero_amd import rng
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo
    d = mzo.dims(*shape)
    rs = np.random.RandomState(1000 * sum(shape) + 10 * G + S)
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * 0.3).astype(np.float32)
    A = shape[1]
    obs = rs.uniform(-1, 1, size=(G, shape[0])).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 77, A, S, 0.25, True)
    eng = make_engine(d, G, S, prec=prec)
    eng.set_weights(blob)
    assert eng.search_kernel_name(S).endswith(SUFFIX[prec])
    eng.search(obs, dirichlet, noise, S, True)
    out, tree = eng.results_host(), eng.tree_host()
    ref = mzo.search(bf16_dims(d, prec), blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, want=FULL_WANT)
    assert not tree["status"].any()
    assert_search_equals_oracle(out, tree, ref)
    eng.close()


@pytest.mark.parametrize("prec", PRECISIONS)
def test_bf16_full_size_4096_games_vs_oracle(prec):
    """BASELINE workload size in bf16: 4096 games x 50 simulations, three launches in a row (race
    screen), each == oracle on visit counts, root-value bits, expanded leaves and value sums."""
    from muzero_amd import rng
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights("tag123")
    G, S, A = 4096, 50, 2
    obs = np.random.RandomState(1000).uniform(-0.05, 0.05, size=(G, 4)).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G), A, S, 0.25, True)
    ref = mzo.search(bf16_dims(d, prec), blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, n_threads=8, want=("visits", "root_value", "consumed", "node_vsum", "sim_leaf"))
    eng = make_engine(d, G, S, prec=prec)
    eng.set_weights(blob)
    for _ in range(3):
        eng.search(obs, dirichlet, noise, S, True)
        out, tree = eng.results_host(), eng.tree_host()
        assert not tree["status"].any()
        assert np.all(out["visits"].sum(axis=1) == S)
        assert np.array_equal(out["visits"], ref["visits"])
        assert same_bits(out["root_value"], ref["root_value"])
        assert np.array_equal(out["consumed"], ref["consumed"])
        assert np.array_equal(tree["sim_leaf"], ref["sim_leaf"])
        assert same_bits(tree["value_sum"], ref["node_vsum"])
    eng.close()


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("tag,G", [("tag127", 4096), ("tag127", 77), ("lstmA", 4096), ("lstmA", 5)])
def test_bf16_stream_kernels_repeated_launches_vs_oracle(tag, G, prec):
    """The bf16 weight-stream kernels (k_search_quadw2<..,bf16 | bf16x3>: pairs of waves behind block barriers;
    k_search_quadl<..>) at full size and on a ragged count: three launches in a row (race screen), each ==
    the bf16 oracle on visit counts, root-value bits, consumed draws, leaves and value sums."""
    from muzero_amd import rng
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(tag)
    S, A = 30, d.n_actions
    obs = np.random.RandomState(17).uniform(-0.5, 0.5, size=(G, d.obs_dim)).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 9, A, S, 0.25, True)
    ref = mzo.search(bf16_dims(d, prec), blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, n_threads=8, want=("visits", "root_value", "consumed", "node_vsum", "sim_leaf", "hidden"))
    eng = make_engine(d, G, S, prec=prec)
    eng.set_weights(blob)
    assert eng.search_kernel_name(S) == WIDE[tag] + SUFFIX[prec]
    for _ in range(3):
        eng.search(obs, dirichlet, noise, S, True)
        out, tree = eng.results_host(), eng.tree_host()
        assert not tree["status"].any()
        assert np.array_equal(out["visits"], ref["visits"])
        assert same_bits(out["root_value"], ref["root_value"])
        assert np.array_equal(out["consumed"], ref["consumed"])
        assert np.array_equal(tree["sim_leaf"], ref["sim_leaf"])
        assert same_bits(tree["value_sum"], ref["node_vsum"])
        assert same_bits(tree["hidden"], ref["hidden"])
    eng.close()


@pytest.mark.parametrize("shape,S", [((4, 2, 31, 64, 0), 50), ((8, 4, 31, 64, 0), 24), ((4, 2, 31, 64, 4), 30),
                                     ((8, 4, 31, 64, 1), 20)])
def test_bf16_two_waves_per_simd_vs_oracle(shape, S):
    """More than sixteen games per CU: mzs_search takes the bf16 kernel compiled for two wavefronts per
    SIMD (two blocks share a CU; a few registers of the prologue live in scratch).  Ragged game count,
    two launches in a row, == oracle on visit counts, root-value bits, leaves and value sums."""
    from muzero_amd import rng
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo
    d = mzo.dims(*shape)
    G, A = 8192 + 37, shape[1]
    rs = np.random.RandomState(5 + S)
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * 0.3).astype(np.float32)
    obs = rs.uniform(-1, 1, size=(G, shape[0])).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 3, A, S, 0.25, True)
    ref = mzo.search(bf16_dims(d), blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, n_threads=8, want=("visits", "root_value", "consumed", "node_vsum", "sim_leaf"))
    eng = make_engine(d, G, S)
    eng.set_weights(blob)
    assert eng.search_kernel_name(S).endswith("bf16,2 waves/SIMD>"), eng.search_kernel_name(S)
    for _ in range(2):
        eng.search(obs, dirichlet, noise, S, True)
        out, tree = eng.results_host(), eng.tree_host()
        assert not tree["status"].any()
        assert np.array_equal(out["visits"], ref["visits"])
        assert same_bits(out["root_value"], ref["root_value"])
        assert np.array_equal(out["consumed"], ref["consumed"])
        assert np.array_equal(tree["sim_leaf"], ref["sim_leaf"])
        assert same_bits(tree["value_sum"], ref["node_vsum"])
    eng.close()


def test_bf16_handle_trains_in_fp32_and_refreshes_rounded_weights():
    """A bf16 handle keeps fp32 master weights: the training step equals the fp32 handle's bit for bit,
    and after the optimizer step the search / inference kernels read the NEW weights rounded to
    bfloat16 (device-side refresh): inference == bf16 oracle on the read-back blob."""
    import torch
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import Trainer
    from oracle import mz_oracle as mzo
    from test_train_oracle import step_inputs
    fx = load_golden("train_tag123_K5_B128.npz")
    dims = tuple(int(x) for x in fx["dims"])
    inp = step_inputs(fx, 0)
    outs = []
    for prec in (0, 1, 2):
        eng = SearchEngine(dims[:5] + (0, prec), max_games=1, max_simulations=0)
        eng.set_weights(fx["blob0"])
        tr = Trainer(eng, int(fx["K"]), int(fx["B"]), "adam", 0.01)
        tr.step(inp["obs"], inp["actions"], inp["target_value"], inp["target_reward"], inp["target_policy"],
                inp["is_ratio"])
        torch.cuda.synchronize()
        outs.append((tr.grad.cpu().numpy(), tr.weights(), eng))
    mv = load_golden("model_L2_tag123.npz")
    for prec in (1, 2):
        assert same_bits(outs[0][0], outs[prec][0]) and same_bits(outs[0][1], outs[prec][1])
        blob1, eng = outs[prec][1], outs[prec][2]
        assert not np.array_equal(blob1, fx["blob0"])
        d = mzo.dims(*dims[:5], 0, prec)
        rew, nh, pol, val = [t.cpu().numpy() for t in eng.recurrent_inference(mv["hidden"], mv["action"])]
        orr, onh, op, ov = mzo.recurrent_inference(d, blob1, mv["hidden"], mv["action"])
        assert same_bits(rew, orr) and same_bits(nh, onh) and same_bits(pol, op) and same_bits(val, ov)
    for o in outs:
        o[2].close()


@pytest.mark.parametrize("mode,prec", [("x3", 2), ("fast", 1)])
def test_bf16_muzero_drop_in(mode, prec):
    """Muzero(type_format=torch.bfloat16[, bf16_mode="fast"]) + Monte_carlo_tree_search.run: the reference's
    call sequence, bf16 kernels underneath, results == bf16 oracle; within 1e-2 of the fp32 reference."""
    import torch
    from muzero_amd.model import Muzero
    from muzero_amd.search import Monte_carlo_tree_search
    from oracle import mz_oracle as mzo
    from test_oracle_golden import rel_err
    d, blob = fixture_weights("tag123")
    fx = load_golden("search_tag123_S50.npz")
    mz = Muzero(observation_space_dimensions=4, action_space_dimensions=2, state_space_dimensions=31,
                hidden_layer_dimensions=64, number_of_hidden_layer=0, k_hypothetical_steps=5,
                type_format=torch.bfloat16, bf16_mode=mode)
    assert mz.use_amp and mz.dims[-1] == prec
    mz.set_weights(blob)
    mcts = Monte_carlo_tree_search(discount=0.997)
    dirichlet, noise = fixture_streams(fx)
    ref = mzo.search(bf16_dims(d, prec), blob, 50, 0.997, 0.25, fx["pbc_table"], obs=fx["obs"][:8], dirichlet=dirichlet[:8],
                     noise=noise[:8], train=True)
    for g in range(8):
        np.random.seed(int(fx["seed"][g]))
        policy, root, action = mcts.run(observation=torch.tensor(fx["obs"][g:g + 1]), model=mz, num_simulations=50,
                                        train=True)
        assert np.array_equal(policy, ref["visits"][g].astype(np.float64))
        assert np.float32(root.value()).view(np.uint32) == ref["root_value"][g].view(np.uint32)
        if np.array_equal(policy, fx["visits"][g]):
            assert rel_err(root.value(), fx["root_value"][g]) < 1e-2
    p, v = mz.prediction_function_inference(torch.tensor(fx["root_hidden"][:1]))
    assert p.dtype == np.float32 and isinstance(v, np.float32)


def test_bf16_on_any_shape_kernel_warns():
    """A bf16 engine whose shape has no four-games-per-wave kernel says so (it runs at about twice the fp32 time);
    a specialised shape, fp32 and the forced generic engine stay silent."""
    import warnings
    from oracle import mz_oracle as mzo
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        make_engine(mzo.dims(6, 3, 21, 32, 2), 8, 10, prec=2).close()
    assert len(rec) == 1 and "k_search_fused" in str(rec[0].message)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        make_engine(mzo.dims(4, 2, 31, 64, 0), 8, 10, prec=2).close()
        make_engine(mzo.dims(6, 3, 21, 32, 2), 8, 10, prec=0).close()
        make_engine(mzo.dims(6, 3, 21, 32, 2), 8, 10, engine="generic", prec=1).close()
    assert not rec
