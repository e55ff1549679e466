"""Device-side random streams (mzs_fill_streams) against numpy.random.RandomState — the
generator the reference draws from (monte_carlo_tree_search.py:140, :163)."""
import numpy as np
import pytest

from conftest import fixture_weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("A,S,train", [(2, 50, True), (4, 20, True), (3, 7, True), (2, 11, False),
                                       (2, 0, True), (4, 1, True)])
def test_streams_equal_randomstate(A, S, train):
    from muzero_amd import rng
    from muzero_amd.engine import SearchEngine
    from oracle import mz_oracle as mzo
    eng = SearchEngine((4, A, 31, 64, 0), max_games=64, max_simulations=max(S, 1))
    seeds = np.concatenate([np.arange(40), [2 ** 32 - 1, 2 ** 31, 123456789, 7, 7],
                            np.random.RandomState(5).randint(0, 2 ** 31, size=19)]).astype(np.int64)
    dirichlet, noise = eng.fill_streams(seeds, S, train)
    ref_d, ref_n = rng.noise_streams(seeds, A, S, 0.25, train)
    noise = noise.cpu().numpy()
    n_u = rng.max_draws(A, S)
    # MT19937 words -> doubles -> lo + range * d: integer and exactly rounded fp64 work, bit for bit
    assert np.array_equal(noise[:, :n_u].view(np.uint64), ref_n[:, :n_u].view(np.uint64))
    if train and S > 0:
        d = dirichlet.cpu().numpy()
        # legacy gamma sampler with log / pow from include/mzs_crmath.h: the stream oracle's rows bit for bit ...
        od, ou = mzo.streams(seeds, A, S, 0.25, True)
        assert np.array_equal(d.view(np.uint64), od.view(np.uint64))
        assert np.array_equal(noise[:, :n_u].view(np.uint64), ou[:, :n_u].view(np.uint64))
        # ... and NumPy's own (glibc's log / pow beneath it) on every row glibc rounds correctly: at most one
        # of these 64 rows may differ, by one ulp (tests/test_crmath.py measures 4 in 4096)
        ulps = np.abs(d - ref_d) / np.spacing(np.abs(ref_d))
        rows = int(np.all(d.view(np.uint64) == ref_d.view(np.uint64), axis=1).sum())
        print(f"\nDirichlet A={A} S={S}: {rows}/{len(seeds)} rows bit-identical to NumPy, max {ulps.max():.0f} ulp")
        assert ulps.max() <= 2 and rows >= len(seeds) - 1
        assert np.allclose(d.sum(axis=1), 1.0, atol=1e-15)
    else:
        assert dirichlet is None
    eng.close()


@pytest.mark.parametrize("alpha", [0.03, 0.25, 0.5, 0.9, 1.0])
@pytest.mark.parametrize("A,S", [(2, 30), (5, 12), (18, 6), (64, 3)])
def test_streams_other_alphas_and_action_counts(alpha, A, S):
    """root_dirichlet_alpha from 0.03 to 1 (1: the exponential branch of the legacy gamma sampler) and up to 64
    actions — with 64, fewer than A of the 64 rejection attempts a wave evaluates at once are accepted, so the
    scalar continuation of the sampler runs too — on 600 random seeds: uniforms bit for bit vs NumPy, Dirichlet
    rows bit for bit vs the stream oracle and vs NumPy except where glibc's log / pow misround."""
    from muzero_amd import rng
    from muzero_amd.engine import SearchEngine
    G = 600
    eng = SearchEngine((4, A, 31, 64, 0), max_games=G, max_simulations=S, root_dirichlet_alpha=alpha)
    seeds = np.random.RandomState(int(alpha * 100) + A).randint(0, 2 ** 32 - 1, size=G).astype(np.int64)
    dirichlet, noise = eng.fill_streams(seeds, S, True)
    ref_d, ref_n = rng.noise_streams(seeds, A, S, alpha, True)
    n_u = rng.max_draws(A, S)
    assert np.array_equal(noise.cpu().numpy()[:, :n_u].view(np.uint64), ref_n[:, :n_u].view(np.uint64))
    d = dirichlet.cpu().numpy()
    from oracle import mz_oracle as mzo
    od, ou = mzo.streams(seeds, A, S, alpha, True)
    assert np.array_equal(d.view(np.uint64), od.view(np.uint64))             # device == stream oracle, every bit
    big = ref_d > 1e-300                      # alpha 0.03: components underflow towards 0, ulps lose meaning there
    ulps = np.abs(d - ref_d)[big] / np.spacing(np.abs(ref_d[big]))
    rows = int(np.all(d.view(np.uint64) == ref_d.view(np.uint64), axis=1).sum())
    print(f"\nalpha {alpha} A {A}: {rows}/{G} rows bit-identical to NumPy, max {ulps.max():.0f} ulp")
    # a row glibc misrounds once: that gamma is an ulp off, the shared 1 / sum may be too, and the two products
    # round again: up to 3 ulp on a component of an 18-action row, 1 on nearly all
    assert ulps.max() <= 4, ulps.max()
    assert rows >= G - max(2, int(0.01 * G * A))  # glibc misrounds ~0.07 % of its log / pow calls
    assert np.allclose(d[~big], ref_d[~big], atol=1e-300)
    eng.close()


def test_run_batch_with_device_streams_matches_oracle():
    """run_batch(seeds=...) end to end: streams made on the GPU, search == oracle search fed the
    very same streams, and == the host-stream search wherever the Dirichlet bits agree."""
    from muzero_amd import rng
    from muzero_amd.model import Muzero
    from muzero_amd.search import Monte_carlo_tree_search
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights("tag123")
    mz = Muzero(observation_space_dimensions=4, action_space_dimensions=2, state_space_dimensions=31,
                hidden_layer_dimensions=64, number_of_hidden_layer=0, k_hypothetical_steps=5)
    mz.set_weights(blob)
    mcts = Monte_carlo_tree_search(discount=0.997)
    G, S = 100, 50
    obs = np.random.RandomState(3).uniform(-0.05, 0.05, size=(G, 4)).astype(np.float32)
    seeds = np.arange(G) + 1000
    res = mcts.run_batch(obs, S, mz, True, seeds=seeds)
    eng = mcts._engine_for(mz, G, S)
    dirichlet, noise = eng.fill_streams(seeds, S, True)
    ref = mzo.search(d, blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=dirichlet.cpu().numpy(),
                     noise=noise.cpu().numpy(), train=True)
    assert np.array_equal(res.policy, ref["visits"].astype(np.float64))
    assert np.array_equal(res.root_value.view(np.uint32), ref["root_value"].view(np.uint32))
    hd, hn = rng.noise_streams(seeds, 2, S, 0.25, True)
    host = mzo.search(d, blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=hd, noise=hn, train=True)
    same_bits = np.all(dirichlet.cpu().numpy().view(np.uint64) == hd.view(np.uint64), axis=1)
    assert same_bits.sum() >= G - 1                 # NumPy's own rows, but for glibc's rare misroundings
    assert np.array_equal(ref["visits"][same_bits], host["visits"][same_bits])
    od, ou = mzo.streams(seeds, 2, S, 0.25, True)   # the stream oracle: every game, every bit
    orc = mzo.search(d, blob, S, 0.997, 0.25, rng.pbc_table(S), obs=obs, dirichlet=od, noise=ou, train=True)
    assert np.array_equal(res.policy, orc["visits"].astype(np.float64))
    assert np.array_equal(res.root_value.view(np.uint32), orc["root_value"].view(np.uint32))


def _search_both_ways(shape, blob, G, S, prefix, seed0=500, scale=0.05):
    """(fill_streams + search) and search_streams(prefix) on the same handle: results and trees.
    prefix "median": the median number of draws the searches consume (half of them overflow)."""
    from muzero_amd.engine import SearchEngine
    eng = SearchEngine(shape, max_games=G, max_simulations=S, discount=0.997)
    eng.set_weights(blob)
    obs = np.random.RandomState(seed0).uniform(-scale, scale, size=(G, shape[0])).astype(np.float32)
    seeds = np.arange(G, dtype=np.int64) + seed0
    dirichlet, noise = eng.fill_streams(seeds, S, True)
    eng.search(obs, dirichlet, noise, S, True)
    full, full_tree = eng.results_host(), eng.tree_host()
    if prefix == "median":
        prefix = int(np.median(full["consumed"]))
    eng.search_streams(obs, seeds, S, True, prefix_draws=prefix)
    lazy, lazy_tree = eng.results_host(), eng.tree_host()
    name = eng.search_kernel_name(S)
    eng.close()
    return full, full_tree, lazy, lazy_tree, name, prefix


@pytest.mark.parametrize("model,G,S,prefixes", [
    ("tag123", 203, 50, (0, "median", 8)),      # k_search_quad (register-resident weights)
    ("synA4", 45, 20, (0, "median", 4)),        # four actions
    ("lstmA", 37, 30, (0, "median")),           # k_search_quadl (weight stream: redo skips k_build_stream)
])
def test_search_streams_equals_fill_plus_search(model, G, S, prefixes):
    """mzs_search_streams writes a prefix of every stream and redoes the searches that ask for more:
    the end state must be that of mzs_fill_streams + mzs_search bit for bit, whether no game, some
    games or every game overflows its prefix."""
    d, blob = fixture_weights(model)
    shape = (d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers, d.model_kind)
    for prefix in prefixes:
        asked = prefix
        full, ft, lazy, lt, name, prefix = _search_both_ways(shape, blob, G, S, prefix)
        over = int((full["consumed"] > (prefix if prefix > 0 else 9 * d.n_actions * S)).sum())
        print(f"\n{model} {name} prefix={prefix}: {over}/{G} searches redone, "
              f"draws consumed mean {full['consumed'].mean():.0f} max {full['consumed'].max()}")
        assert np.all(lt["status"] == 0) and np.all(ft["status"] == 0)
        for k in ("visits", "consumed", "path_len"):
            assert np.array_equal(full[k], lazy[k]), (prefix, k)
        assert np.array_equal(full["root_value"].view(np.uint32), lazy["root_value"].view(np.uint32))
        assert np.array_equal(full["root_prior"].view(np.uint64), lazy["root_prior"].view(np.uint64))
        for k in ("visit", "child_base", "sim_leaf"):
            assert np.array_equal(ft[k], lt[k]), (prefix, k)
        for k in ("value_sum", "prior", "reward", "minmax", "hidden"):
            assert np.array_equal(ft[k].view(np.uint32), lt[k].view(np.uint32)), (prefix, k)
        if asked in (4, 8):
            assert over == G                 # every search was redone
        if asked == "median":
            assert 0 < over < G              # a mixed case: redone and untouched quads side by side


def test_search_streams_generic_kernel_and_bf16():
    """The one-game-per-wave kernel (a shape without a quad kernel) and the bf16 quad kernel."""
    from muzero_amd.model import blob_layout
    rs = np.random.RandomState(77)
    shape = (6, 3, 20, 48, 1)                      # no compile-time instantiation: k_search_fused
    n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(*shape))
    blob = (rs.standard_normal(n_w) * 0.3).astype(np.float32)
    for prefix in (0, "median"):
        full, ft, lazy, lt, name, _ = _search_both_ways(shape, blob, 29, 16, prefix, scale=1.0)
        assert "fused" in name
        assert np.array_equal(full["visits"], lazy["visits"]) and np.array_equal(full["consumed"], lazy["consumed"])
        assert np.array_equal(full["root_value"].view(np.uint32), lazy["root_value"].view(np.uint32))
        assert np.array_equal(ft["visit"], lt["visit"]) and np.all(lt["status"] == 0)
    d, blob = fixture_weights("tag123")
    shape = (d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers, 0, 1)
    for prefix in (0, "median"):
        full, ft, lazy, lt, name, _ = _search_both_ways(shape, blob, 64, 50, prefix)
        assert np.array_equal(full["visits"], lazy["visits"]) and np.array_equal(full["consumed"], lazy["consumed"])
        assert np.array_equal(full["root_value"].view(np.uint32), lazy["root_value"].view(np.uint32))
        assert np.array_equal(ft["visit"], lt["visit"]) and np.all(lt["status"] == 0)
