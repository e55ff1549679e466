import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def fixture_weights(tag):
    """(oracle dims, blob) of tests/golden/weights_<tag>.npz"""
    from oracle import mz_oracle as mzo
    w = load_golden(f"weights_{tag}.npz")
    d = mzo.dims(w["obs_dim"], w["A"], w["S_dim"], w["H"], w["L"], w.get("kind", 0))
    return d, w["blob"]


def fixture_streams(fx):
    """Rebuild dirichlet / noise of a search fixture from its seeds and check them
    against what the reference consumed (draws stored ragged in the fixture)."""
    from muzero_amd import rng
    A, S = int(fx["A"]), int(fx["S"])
    dirichlet, noise = rng.noise_streams(fx["seed"], A, S, float(fx["alpha"]), bool(fx["train"]))
    off = fx["draw_offsets"]
    for g in range(len(fx["seed"])):
        n = int(fx["n_draws"][g])
        assert np.array_equal(noise[g, :n], fx["draws"][off[g]:off[g] + n])
    if bool(fx["train"]) and S > 0:
        assert np.array_equal(dirichlet, fx["dirichlet"])
    return dirichlet, noise


def fixture_obs(fx):
    """Observations of a search fixture: stored, or — 'vision_model' fixtures, 115 KB per image —
    regenerated from the seed tests/golden/make_vision_fixtures.py drew them from."""
    if "obs" in fx:
        return fx["obs"]
    n = len(fx["seed"])
    return np.random.RandomState(int(fx["obs_seed"])).uniform(0.0, 1.0, size=(n, 3, 98, 98)).astype(np.float32)


SEARCH_FIXTURES = [
    ("search_tag123_S50.npz", "tag123"), ("search_tag123_S11.npz", "tag123"),
    ("search_tag123_S1.npz", "tag123"), ("search_tag123_S0.npz", "tag123"),
    ("search_tag123_S50_notrain.npz", "tag123"), ("search_tag123_S50_wide.npz", "tag123"),
    ("search_tag127_S50.npz", "tag127"), ("search_tag133_S50.npz", "tag133"),
    ("search_tag137_S25.npz", "tag137"), ("search_synA4_S50.npz", "synA4"),
    ("search_synA3_S30.npz", "synA3"),
    ("search_lstmA_S50.npz", "lstmA"), ("search_lstmB_S30.npz", "lstmB"),      # 'lstm_model'
    ("search_visA_S30.npz", "visA"), ("search_visB_S20.npz", "visB"),          # 'vision_model'
    ("search_visA_S12_notrain.npz", "visA"),
]
MODEL_TAGS = ["tag123", "tag127", "tag133", "tag137", "synA4", "synA3", "lstmA", "lstmB"]
