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
        # legacy gamma sampler through the device's fp64 log / pow: same accept/reject path
        # (else the uniforms above would be shifted); every component within a few ulp OF ITSELF
        # (pow and log are each good to an ulp or two, the normalisation adds the sum's error)
        ulps = np.abs(d - ref_d) / np.spacing(np.abs(ref_d))
        print(f"\nDirichlet A={A} S={S}: max error {ulps.max():.1f} ulp")
        assert ulps.max() <= 8
        assert np.allclose(d.sum(axis=1), 1.0, atol=1e-15)
    else:
        assert dirichlet is None
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
    assert same_bits.mean() > 0.5
    assert np.array_equal(ref["visits"][same_bits], host["visits"][same_bits])
