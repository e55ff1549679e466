"""The 'vision_model' inference oracle (oracle/mz_vision_oracle.c) pinned to the reference's own
ResNet-v2 networks (tests/golden/model_L2_vis*.npz, made by tests/golden/make_vision_fixtures.py
from neural_network_vision_model.py in eval mode)."""
import numpy as np
import pytest

from conftest import fixture_weights, load_golden

VISION_MODELS = ["visA", "visB"]
REL = 1e-5   # fp32 tolerance of the model functions (north_star), relative to the largest output


def vision_obs(fx, n):
    """the fixture's observations, regenerated from its seed (make_vision_fixtures.vision_vectors)"""
    return np.random.RandomState(int(fx["obs_seed"])).uniform(0.0, 1.0, size=(n, 3, 98, 98)).astype(np.float32)


def close(a, b, rel=REL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rel * max(1.0, np.abs(b).max())


def value_close(got, ref):
    # |error| of the scalar behind inverse_transform_with_support: the reference's own fp32 evaluation of
    # the inverse transform is quantised to ~1.2e-4 * sqrt(|v| + 1) (tests/test_oracle_golden.py::value_err)
    tol = 1e-5 * (1 + np.abs(ref)) + 2.4e-4 * np.sqrt(np.abs(ref) + 1)
    return np.all(np.abs(got - ref) <= tol)


@pytest.mark.parametrize("name", VISION_MODELS)
def test_vision_layout_matches_python(name):
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(name)
    assert d.model_kind == 2 and blob.size == mzo.blob_size(d)
    names = [n for n, _ in blob_layout(d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers, 2)]
    assert len(names) == len(set(names))


@pytest.mark.parametrize("name", VISION_MODELS)
def test_vision_oracle_matches_reference(name):
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights(name)
    fx = load_golden(f"model_L2_{name}.npz")
    n = fx["hidden"].shape[0]
    obs = vision_obs(fx, n)
    hid = mzo.representation(d, blob, obs)
    assert hid.shape == (n, 147)
    err = np.abs(hid - fx["repr_hidden"].reshape(n, -1)).max() / np.abs(fx["repr_hidden"]).max()
    print(f"\n{name}: representation max error {err:.2e} of the largest activation")
    assert close(hid, fx["repr_hidden"].reshape(n, -1))
    pol, val = mzo.prediction(d, blob, hid)
    assert close(pol, fx["ini_policy"]) and value_close(val, fx["ini_value"])
    h = fx["hidden"].reshape(n, -1)
    rew, nh = mzo.dynamics(d, blob, h, fx["action"])
    assert close(nh, fx["dyn_hidden"].reshape(n, -1)) and value_close(rew, fx["dyn_reward"])
    pol, val = mzo.prediction(d, blob, h)
    assert close(pol, fx["pred_policy"]) and value_close(val, fx["pred_value"])
    pol, val = mzo.prediction(d, blob, nh)
    assert close(pol, fx["rec_policy"]) and value_close(val, fx["rec_value"])


def test_vision_downsample_output_matches_reference():
    """Down_sample alone (conv, residual blocks, average pools: 98 -> 49 -> 25 -> 13 -> 7): its last
    stage as the reference computed it for the first observation."""
    import ctypes as C
    from oracle import mz_oracle as mzo
    d, blob = fixture_weights("visA")
    fx = load_golden("model_L2_visA.npz")
    obs = vision_obs(fx, fx["hidden"].shape[0])[:1]
    hid = np.empty((1, 147), np.float32)
    stage = np.empty(147, np.float32)
    mzo.lib().mzo_vision_representation(C.byref(d), mzo._p(mzo._f32(blob)), mzo._p(obs), C.c_int32(1), mzo._p(hid),
                                        mzo._p(stage))
    ref = fx["ds_stage10"].ravel()
    assert ref.size == 147 and close(stage, ref)
