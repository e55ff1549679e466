"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle and the golden
fixtures recorded from the reference.  Run on the MI355X box: pytest -m gpu.

Bars (north_star): visit counts / expanded leaves / draws consumed bit-exact; with
the arithmetic contract of include/mzs_numeric.h the fp32 outputs are bit-exact
against the oracle as well (compared as raw bits), and within 1e-5 relative of the
reference fixtures (value scalars: plus one quantum of the reference's own fp32
inverse transform, see tests/test_oracle_golden.py::value_err).
"""
import numpy as np
import pytest

from conftest import (MODEL_TAGS, SEARCH_FIXTURES, fixture_obs, fixture_streams, fixture_weights, load_golden)
from test_oracle_golden import (REL, check_L3_against_reference, hidden_err, leaves_of, rel_err, table_of,
                                value_err)

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64) if a.dtype == np.float64 else a


def same_bits(a, b):
    return np.array_equal(bits(np.asarray(a)), bits(np.asarray(b)))


ENGINES = ["fused", "generic"]   # fused = k_search_quad where the shape has one; generic = k_search_fused


def make_engine(d, G, S, gamma, frac=0.25, engine="fused"):
    from muzero_amd import _lib
    from muzero_amd.engine import SearchEngine
    return SearchEngine((d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers, d.model_kind),
                        max_games=G, max_simulations=S, discount=gamma,
                        root_exploration_fraction=frac,
                        engine=_lib.ENGINE_FUSED if engine == "fused" else _lib.ENGINE_FUSED_GENERIC)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("fname,tag", SEARCH_FIXTURES)
def test_L1_search_injected_bit_exact(fname, tag, engine):
    """Search kernels alone: the reference's own model outputs injected."""
    fx = load_golden(fname)
    d, _ = fixture_weights(tag)
    S, G = int(fx["S"]), len(fx["seed"])
    dirichlet, noise = fixture_streams(fx)
    eng = make_engine(d, G, S, float(fx["gamma"]), float(fx["frac"]), engine)
    assert np.array_equal(eng.pbc[: S + 1], fx["pbc_table"][: S + 1])
    train = bool(fx["train"])
    eng.search_injected(table_of(fx), dirichlet if train and S > 0 else None,
                        noise if S > 0 else None, S, train)
    out = eng.results_host()
    tree = eng.tree_host()
    if S > 0:
        assert np.array_equal(out["visits"].astype(np.float64), fx["visits"])
    assert same_bits(out["root_prior"], fx["root_prior"])
    assert same_bits(out["root_value"], fx["root_value"])
    assert np.array_equal(out["consumed"], fx["n_draws"])
    assert np.array_equal(out["path_len"], fx["total_path_len"])
    assert np.array_equal(tree["sim_leaf"], leaves_of(fx))
    assert np.array_equal(tree["visit"], fx["child_visit"])
    assert same_bits(tree["value_sum"], fx["child_vsum"])
    assert not tree["status"].any()
    eng.close()


@pytest.mark.parametrize("tag", MODEL_TAGS)
def test_L2_inference_kernels(tag):
    """initial / recurrent / prediction inference: bit-exact vs the oracle, within
    tolerance of the reference wrappers."""
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(tag)
    mv = load_golden(f"model_L2_{tag}.npz")
    eng = make_engine(d, 1, 0, 0.997)
    eng.set_weights(blob)
    hid, pol, val = [t.cpu().numpy() for t in eng.initial_inference(mv["obs"])]
    oh, op, ov = mzo.initial_inference(d, blob, mv["obs"])
    assert same_bits(hid, oh) and same_bits(pol, op) and same_bits(val, ov)
    assert hidden_err(hid, mv["repr_hidden"]).max() < REL
    assert rel_err(pol, mv["ini_policy"]).max() < REL
    assert value_err(val, mv["ini_value"]).max() <= 1
    rew, nh, pol, val = [t.cpu().numpy() for t in eng.recurrent_inference(mv["hidden"], mv["action"])]
    orr, onh, op, ov = mzo.recurrent_inference(d, blob, mv["hidden"], mv["action"])
    assert same_bits(rew, orr) and same_bits(nh, onh) and same_bits(pol, op) and same_bits(val, ov)
    assert value_err(rew, mv["dyn_reward"]).max() <= 1
    assert hidden_err(nh, mv["dyn_hidden"]).max() < REL
    assert rel_err(pol, mv["rec_policy"]).max() < REL
    assert value_err(val, mv["rec_value"]).max() <= 1
    pol, val = [t.cpu().numpy() for t in eng.prediction_inference(mv["hidden"])]
    op, ov = mzo.prediction(d, blob, mv["hidden"])
    assert same_bits(pol, op) and same_bits(val, ov)
    assert rel_err(pol, mv["pred_policy"]).max() < REL
    assert value_err(val, mv["pred_value"]).max() <= 1
    eng.close()


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("fname,tag", SEARCH_FIXTURES)
def test_L3_search_end_to_end(fname, tag, engine):
    """Full search on the GPU: bit-exact vs the oracle on every game; vs the reference,
    identical visit counts on EVERY search (near-ties included; counts pinned in
    tests/golden/parity_table.json) and root values within a plain 1e-5 relative."""
    from oracle import mz_oracle as mzo
    fx = load_golden(fname)
    d, blob = fixture_weights(tag)
    S, G = int(fx["S"]), len(fx["seed"])
    dirichlet, noise = fixture_streams(fx)
    train = bool(fx["train"])
    eng = make_engine(d, G, S, float(fx["gamma"]), float(fx["frac"]), engine)
    eng.set_weights(blob)
    eng.search(fixture_obs(fx), dirichlet if train and S > 0 else None, noise if S > 0 else None, S, train)
    out = eng.results_host()
    tree = eng.tree_host()
    ref = mzo.search(d, blob, S, float(fx["gamma"]), float(fx["frac"]), fx["pbc_table"],
                     obs=fixture_obs(fx), dirichlet=dirichlet, noise=noise, train=train,
                     want=("visits", "root_value", "root_prior", "consumed", "path_len", "sim_leaf",
                           "node_visit", "node_vsum", "node_reward", "node_prior", "child_base",
                           "minmax", "hidden"))
    assert np.array_equal(out["visits"], ref["visits"])
    assert same_bits(out["root_value"], ref["root_value"])
    assert same_bits(out["root_prior"], ref["root_prior"])
    assert np.array_equal(out["consumed"], ref["consumed"])
    assert np.array_equal(out["path_len"], ref["path_len"])
    assert np.array_equal(tree["sim_leaf"], ref["sim_leaf"])
    assert np.array_equal(tree["visit"], ref["node_visit"])
    assert np.array_equal(tree["child_base"], ref["child_base"])
    assert same_bits(tree["value_sum"], ref["node_vsum"])
    assert same_bits(tree["reward"], ref["node_reward"])
    assert same_bits(tree["prior"], ref["node_prior"])
    assert same_bits(tree["minmax"], ref["minmax"])
    assert same_bits(tree["hidden"], ref["hidden"])
    # against the reference run itself
    assert rel_err(out["root_prior"], fx["root_prior"]).max() < REL
    if S > 0:
        m = check_L3_against_reference(fname, fx, out["visits"], out["root_value"])
        print(f"\n{fname} [{engine}]: visit counts identical {m['matched']}/{m['searches']} (near-ties "
              f"{m['near_ties']}), root value max rel err {m['root_value_rel']:.2e}")
        assert np.array_equal(np.argmax(out["visits"], axis=1), fx["t0_action"])
    eng.close()


FULL_WANT = ("visits", "root_value", "root_prior", "consumed", "path_len", "sim_leaf", "node_visit",
             "node_vsum", "node_reward", "node_prior", "child_base", "minmax", "hidden")


def assert_search_equals_oracle(out, tree, ref):
    assert np.array_equal(out["visits"], ref["visits"])
    assert same_bits(out["root_value"], ref["root_value"])
    assert same_bits(out["root_prior"], ref["root_prior"])
    assert np.array_equal(out["consumed"], ref["consumed"])
    assert np.array_equal(out["path_len"], ref["path_len"])
    assert np.array_equal(tree["sim_leaf"], ref["sim_leaf"])
    assert np.array_equal(tree["visit"], ref["node_visit"])
    assert np.array_equal(tree["child_base"], ref["child_base"])
    assert same_bits(tree["value_sum"], ref["node_vsum"])
    assert same_bits(tree["reward"], ref["node_reward"])
    assert same_bits(tree["prior"], ref["node_prior"])
    assert same_bits(tree["minmax"], ref["minmax"])
    assert same_bits(tree["hidden"], ref["hidden"])


@pytest.mark.parametrize("shape", [(8, 4, 31, 64, 0), (4, 2, 31, 64, 0), (4, 2, 11, 16, 0),
                                   (4, 2, 31, 64, 4), (4, 2, 31, 64, 1), (4, 2, 11, 16, 1),
                                   (6, 3, 31, 64, 0), (6, 3, 31, 64, 2), (8, 4, 31, 64, 1)])
@pytest.mark.parametrize("G,S", [(37, 50), (5, 7), (1, 50)])
def test_quad_shapes_vs_oracle(shape, G, S):
    """Every shape k_search_quad is instantiated for (incl. the LunarLander-like A = 4 one, which no
    shipped checkpoint covers), ragged game counts (partial waves / blocks), seeded random
    weights: fused engine == oracle on every output bit.  The oracle is pinned to the reference
    by tests/test_oracle_golden.py; these weights have no reference run (parity vs oracle only)."""
    from muzero_amd import rng
    from oracle import mz_oracle as mzo
    d = mzo.dims(*shape)
    rs = np.random.RandomState(1000 * sum(shape) + 10 * G + S)
    from muzero_amd.model import blob_layout
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * 0.3).astype(np.float32)
    A = shape[1]
    obs = rs.uniform(-1, 1, size=(G, shape[0])).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 77, A, S, 0.25, True)
    eng = make_engine(d, G, S, 0.997)
    eng.set_weights(blob)
    eng.search(obs, dirichlet, noise, S, True)
    out, tree = eng.results_host(), eng.tree_host()
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, want=FULL_WANT)
    assert not tree["status"].any()
    assert_search_equals_oracle(out, tree, ref)
    eng.close()


@pytest.mark.parametrize("shape,S", [((4, 2, 31, 64, 0), 120), ((4, 2, 31, 64, 0), 230), ((8, 4, 31, 64, 1), 110),
                                     ((4, 2, 31, 64, 0), 400)])
def test_deep_searches_vs_oracle(shape, S):
    """More simulations than the benchmark's 50: the per-game LDS footprint grows, so the quad kernel
    switches to the variant with the hidden states in the global array (S = 120, 230: still four
    waves per block; S = 400: fewer); every variant == oracle on every output bit."""
    from muzero_amd import rng
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo
    d = mzo.dims(*shape)
    G = 11
    rs = np.random.RandomState(S + sum(shape))
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * 0.3).astype(np.float32)
    obs = rs.uniform(-1, 1, size=(G, shape[0])).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 5, shape[1], S, 0.25, True)
    eng = make_engine(d, G, S, 0.997)
    eng.set_weights(blob)
    eng.search(obs, dirichlet, noise, S, True)
    out, tree = eng.results_host(), eng.tree_host()
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, want=FULL_WANT)
    assert not tree["status"].any()
    assert_search_equals_oracle(out, tree, ref)
    assert out["visits"].sum(1).tolist() == [S] * G
    eng.close()


@pytest.mark.parametrize("shape", [(4, 2, 31, 64, 0), (4, 2, 31, 64, 2), (6, 3, 21, 32, 1)])
def test_saturated_logits_bit_exact(shape):
    """Weights 10x the usual scale: logit spreads of several hundred, so softmax terms hit the
    exp clamp and quotients can go subnormal — the corner of the contract's shared-reciprocal
    quotient (mzs_div_r2) where it may differ from a correctly rounded division.  GPU (quad and
    generic kernels) and oracle follow the same operation sequence: still bit-identical."""
    from muzero_amd import rng
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo
    d = mzo.dims(*shape)
    G, S = 23, 12
    rs = np.random.RandomState(sum(shape))
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * 3.0).astype(np.float32)
    obs = rs.uniform(-1, 1, size=(G, shape[0])).astype(np.float32)
    pol = mzo.prediction(d, blob, mzo.representation(d, blob, obs))[0]
    assert pol.min() < 1e-30 and (pol > 0).all()               # saturated: terms at the exp clamp
    dirichlet, noise = rng.noise_streams(np.arange(G) + 9, shape[1], S, 0.25, True)
    for engine in ENGINES:
        eng = make_engine(d, G, S, 0.997, engine=engine)
        eng.set_weights(blob)
        eng.search(obs, dirichlet, noise, S, True)
        out, tree = eng.results_host(), eng.tree_host()
        ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet, noise=noise,
                         train=True, want=FULL_WANT)
        assert_search_equals_oracle(out, tree, ref)
        eng.close()



@pytest.mark.parametrize("shape,S", [((4, 2, 31, 64, 0), 50), ((8, 4, 31, 64, 0), 24)])
def test_two_waves_per_simd_kernel_vs_oracle(shape, S):
    """More than sixteen games per CU (8192 + 37 games): mzs_search takes the kernel compiled for two
    wavefronts per SIMD — two blocks share a CU, hidden states in the global array, a few dozen
    register reloads from scratch per simulation.  Two launches in a row, == oracle bit for bit."""
    from muzero_amd import rng
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo
    d = mzo.dims(*shape)
    G, A = 8192 + 37, shape[1]
    rs = np.random.RandomState(11 + S)
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * 0.3).astype(np.float32)
    obs = rs.uniform(-1, 1, size=(G, shape[0])).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 9, A, S, 0.25, True)
    ref = mzo.search(d, blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, n_threads=8, want=("visits", "root_value", "consumed", "node_vsum", "sim_leaf"))
    eng = make_engine(d, G, S, 0.997)
    eng.set_weights(blob)
    assert eng.search_kernel_name(S).endswith(",2 waves/SIMD>"), eng.search_kernel_name(S)
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


@pytest.mark.parametrize("tag", ["tag123", "tag133", "tag127", "lstmA"])
def test_full_size_4096_games_vs_oracle(tag):
    """BASELINE config C2: 4096 games x 50 simulations against the oracle on all games (bit-exact),
    plus size-independent invariants — tag-123 weights (the benchmark: k_search_quad), and the same
    size on the tied-mid-layer variant (tag 133), the wide kernel (tag 127: k_search_quadw) and the
    LSTM model (k_search_quadl)."""
    from muzero_amd import rng
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(tag)
    G, S, A = 4096, 50, 2
    obs = np.random.RandomState(1000).uniform(-0.05, 0.05, size=(G, 4)).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G), A, S, 0.25, True)
    eng = make_engine(d, G, S, 0.997)
    eng.set_weights(blob)
    eng.search(obs, dirichlet, noise, S, True)
    out = eng.results_host()
    tree = eng.tree_host()
    assert not tree["status"].any()
    # invariants: every simulation visits exactly one root child; the root is visited S times
    assert np.all(out["visits"].sum(axis=1) == S)
    assert np.all(tree["visit"][:, 0] == S)
    assert np.all(out["consumed"] == A * out["path_len"])
    # each expanded node's visit count = 1 + sum of its children's
    cb, vis = tree["child_base"], tree["visit"]
    exp = cb > 0
    child_sum = np.zeros_like(vis)
    gi, ni = np.nonzero(exp)
    for a in range(A):
        child_sum[gi, ni] += vis[gi, cb[gi, ni] + a]
    assert np.all(vis[gi, ni] == child_sum[gi, ni] + (ni > 0))
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, n_threads=8,
                     want=("visits", "root_value", "consumed", "node_vsum", "sim_leaf"))
    assert np.array_equal(out["visits"], ref["visits"])
    assert same_bits(out["root_value"], ref["root_value"])
    assert np.array_equal(out["consumed"], ref["consumed"])
    assert np.array_equal(tree["sim_leaf"], ref["sim_leaf"])
    assert same_bits(tree["value_sum"], ref["node_vsum"])
    eng.close()


@pytest.mark.parametrize("tag", ["tag123", "tag133", "tag127", "lstmA", "synA4"])
def test_repeated_launches_agree_with_generic_kernel(tag):
    """Race screen (tools/cmp_engines.py): three launches of the shape's fast kernel in a row, each
    compared with the generic one-game-per-wave kernel on all 4096 games — a data race between
    waves or against an asynchronous copy shows as mismatches that move from launch to launch."""
    from muzero_amd import rng
    d, blob = fixture_weights(tag)
    G, S, A = 4096, 50, d.n_actions     # synA4: four actions = hidden states in the global array
    obs = np.random.RandomState(5).uniform(-0.05, 0.05, size=(G, d.obs_dim)).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G), A, S, 0.25, True)
    gen = make_engine(d, G, S, 0.997, engine="generic")
    gen.set_weights(blob)
    gen.search(obs, dirichlet, noise, S, True)
    ref, ref_tree = gen.results_host(), gen.tree_host()
    gen.close()
    eng = make_engine(d, G, S, 0.997)
    eng.set_weights(blob)
    for _ in range(3):
        eng.search(obs, dirichlet, noise, S, True)
        out, tree = eng.results_host(), eng.tree_host()
        assert np.array_equal(out["visits"], ref["visits"])
        assert same_bits(out["root_value"], ref["root_value"])
        assert same_bits(tree["value_sum"], ref_tree["value_sum"])
    eng.close()


@pytest.mark.parametrize("shape", [(4, 2, 61, 128, 0, 0), (4, 2, 61, 128, 1, 0), (4, 2, 61, 128, 2, 0),
                                   (17, 2, 61, 128, 1, 0), (4, 2, 61, 128, 15, 0), (4, 2, 31, 64, 1, 1),
                                   (9, 2, 31, 64, 1, 1)])
@pytest.mark.parametrize("G,S", [(37, 50), (5, 7), (3, 230)])
def test_stream_kernel_shapes_vs_oracle(shape, G, S):
    """The weight-stream kernels (k_search_quadw: S_dim 61 / H 128 with 0, 1, 2 tied mid layers — the
    chunk count per net decides which register buffer a unit reads; 15 mid layers = more chunks than
    the image table holds, i.e. the generic kernel; k_search_quadl: the one-layer LSTM), ragged game
    counts, a deep search (fewer waves per block), seeded random weights: == oracle on every bit."""
    from muzero_amd import rng
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo
    d = mzo.dims(*shape)
    rs = np.random.RandomState(1000 * sum(shape) + 10 * G + S)
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * (0.1 if shape[4] > 4 else 0.3)).astype(np.float32)
    obs = rs.uniform(-1, 1, size=(G, shape[0])).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G) + 31, shape[1], S, 0.25, True)
    eng = make_engine(d, G, S, 0.997)
    eng.set_weights(blob)
    eng.search(obs, dirichlet, noise, S, True)
    out, tree = eng.results_host(), eng.tree_host()
    ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=dirichlet, noise=noise,
                     train=True, want=FULL_WANT)
    assert not tree["status"].any()
    assert_search_equals_oracle(out, tree, ref)
    eng.close()


def test_randomised_sweep_equals_oracle():
    """A few seconds of tests/sweeps/parity_sweep.py: random shapes (every specialised kernel and the generic
    one, fp32 and bf16), weight scales, game and simulation counts, with and without root noise, short
    stream prefixes that force the device-side redo pass — every game == oracle, bit for bit.  The long
    runs are recorded in profiles/r2d_parity_sweep.md."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweeps"))
    import parity_sweep
    out = parity_sweep.run(8.0, seed=2024, threads=8, max_games=1024, log=lambda s: None)
    assert out["summary"] == "all equal", out
    assert out["cases"] >= 5


@pytest.mark.parametrize("tag", ["tag123", "tag127"])
def test_result_sink_matches_device_results(tag):
    """mzs_set_result_sink: the visit counts and root values the search kernel stores into pinned
    host memory are the handle's own results, bit for bit (both kernel families), and switching the
    sink off leaves the host tensors alone."""
    import torch
    from muzero_amd import rng
    d, blob = fixture_weights(tag)
    G, S, A = 1000, 25, 2
    obs = np.random.RandomState(3).uniform(-0.05, 0.05, size=(G, 4)).astype(np.float32)
    dirichlet, noise = rng.noise_streams(np.arange(G), A, S, 0.25, True)
    eng = make_engine(d, G, S, 0.997)
    eng.set_weights(blob)
    vis_h = torch.full((G, A), -1, dtype=torch.int32).pin_memory()
    val_h = torch.full((G,), -1.0, dtype=torch.float32).pin_memory()
    eng.set_result_sink(vis_h, val_h)
    eng.search(obs, dirichlet, noise, S, True)
    out = eng.results_host()
    assert np.array_equal(vis_h.numpy(), out["visits"])
    assert same_bits(val_h.numpy(), out["root_value"])
    eng.set_result_sink(None, None)
    vis_h.fill_(-7)
    eng.search(obs, dirichlet, noise, S, True)
    out2 = eng.results_host()
    assert np.array_equal(out2["visits"], out["visits"])
    assert (vis_h.numpy() == -7).all()
    with pytest.raises(ValueError):
        eng.set_result_sink(torch.zeros((G, A), dtype=torch.int32), val_h)   # not pinned
    eng.close()


def test_device_reciprocals_are_correctly_rounded():
    """quad_rcp_count (hardware rcp + two Newton steps, f64) == 1.0 / d for every visit count a tree
    can hold (d <= 65536): the one place where the search kernels do not take a quotient from
    IEEE division or a host-made table."""
    import ctypes as C
    from muzero_amd import _lib
    lib = _lib.load()
    bad = C.c_int32(-1)
    _lib.check(lib.mzs_selftest_reciprocals(0, 65536, C.byref(bad)), "mzs_selftest_reciprocals")
    assert bad.value == 0
