"""The 'vision_model' inference oracle (oracle/mz_vision_oracle.c) pinned to the reference's own
ResNet-v2 networks (tests/golden/model_L2_vis*.npz, made by tests/golden/make_vision_fixtures.py
from neural_network_vision_model.py in eval mode)."""
import numpy as np
import pytest

from conftest import fixture_weights, load_golden

from test_oracle_golden import REL, rel_err, value_err

VISION_MODELS = ["visA", "visB"]


def vision_obs(fx, n):
    """the fixture's observations, regenerated from its seed (make_vision_fixtures.vision_vectors)"""
    return np.random.RandomState(int(fx["obs_seed"])).uniform(0.0, 1.0, size=(n, 3, 98, 98)).astype(np.float32)


def close(a, b, rel=REL):
    """Feature maps and policies, ELEMENT-WISE at the MLP tests' 1e-5: |a - b| <= 1e-5 max(|b|, 1) for every
    element (the maps are not min-max scaled like the MLP's hidden states: entries reach ~5; policies are
    below 1, so theirs is an absolute 1e-5 and test_vision_policies_relative adds the MLP's relative bar).
    Measured maximum: 8e-7."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= rel * np.maximum(np.abs(b), 1.0)))


def value_close(got, ref):
    """Scalars behind inverse_transform_with_support: the MLP tests' bar (test_oracle_golden.value_err):
    1e-5 relative plus ONE step of the grid the reference's own fp32 inverse transform puts its outputs on."""
    return bool(np.all(value_err(got, ref) <= 1))


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
    assert rel_err(pol, fx["rec_policy"]).max() < REL          # the MLP tests' relative bar on policies
    pol, _ = mzo.prediction(d, blob, h)
    assert rel_err(pol, fx["pred_policy"]).max() < REL


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
