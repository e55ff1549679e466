"""Pins the CPU oracle (oracle/) against the reference's own outputs.

Fixtures: tests/golden/*.npz, produced by tests/golden/make_fixtures.py which
imports /root/reference in the build container.  CPU-only.
"""
import numpy as np
import pytest

from conftest import (MODEL_TAGS, SEARCH_FIXTURES, fixture_obs, fixture_streams, fixture_weights, load_golden)
from oracle import mz_oracle as mzo

import json
import os

REL = 1e-5          # north_star: root values and policies within 1e-5 relative (fp32)
MARGIN_TAU = 1e-4   # decisions closer than this count as near-ties in the printed summary

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_table.json")) as _f:
    PARITY_TABLE = json.load(_f)


def check_L3_against_reference(fname, fx, visits, root_value):
    """The L3 bar shared by the CPU (oracle) and GPU tests: EVERY search of the fixture reproduces
    the reference's visit counts (near-ties included: the count per file is pinned in
    tests/golden/parity_table.json, so one flipped search fails), and root values are within a plain
    1e-5 relative of the reference's, no grid allowance — except the S = 1 fixture, whose root value
    IS a single leaf scalar (root.value() = the one backed-up value) and carries value_err's grid
    step like every other single-leaf scalar.  Returns the measured maxima for the test's printout."""
    row = PARITY_TABLE[fname]
    same = np.all(np.asarray(visits).astype(np.float64) == fx["visits"], axis=1)
    assert len(same) == row["searches"]
    assert int(same.sum()) == row["visits_equal"], f"{fname}: {int(same.sum())}/{len(same)} searches match"
    rv = rel_err(root_value, fx["root_value"]).max()
    if row.get("single_leaf"):
        assert value_err(root_value, fx["root_value"]).max() <= 1
    elif row.get("leaf_grid"):
        # 'vision_model' fixtures (seeded random weights): rewards of +-8 and values of -1..-40 cancel in a
        # root value of -5, so ONE grid step of a reward next to the root (value_err's q at |r| = 8: 3.6e-4;
        # 1-3 % of the leaves differ from the reference by exactly one step, none by more) is 8e-5 of the
        # root value.  Bar: 1e-5 relative plus one step of the grid at the search's largest leaf scalar.
        big = np.maximum(np.abs(fx["sim_value"]).max(axis=1), np.abs(fx["sim_reward"]).max(axis=1))
        q = 2.0 * 2.0 ** -23 / 0.002 * np.sqrt(big + 1.0)
        err = np.abs(np.asarray(root_value, np.float64) - fx["root_value"])
        assert np.all(err <= REL * np.abs(fx["root_value"]) + 1.05 * q), f"{fname}: root value error {err.max():.3e}"
    else:
        assert rv < REL, f"{fname}: root value rel err {rv:.3e}"
    assert rv <= row["root_value_rel_max"], f"{fname}: root value rel err {rv:.3e} above the pinned bound"
    near = int((fx["min_margin"] < MARGIN_TAU).sum())
    return dict(matched=int(same.sum()), searches=len(same), near_ties=near, root_value_rel=float(rv))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-3)


def value_err(a, b):
    """Error of a value/reward scalar in units of its tolerance (<= 1 passes).

    Used for SINGLE-LEAF scalars only (a value / reward straight out of one inverse transform);
    root values of searches with S >= 11 are held to a plain 1e-5 (check_L3_against_reference).
    Measured maxima on the fixtures (oracle == GPU bits): value 7.96e-5 rel (tag123), 7.1e-5
    (tag127), 2.0e-5 (tag133), reward 3.9e-5 (synA3) — each exactly one grid step — policies
    <= 4.8e-6, hidden states <= 1.0e-6 absolute.

    inverse_transform_with_support (muzero_model.py:585-586) evaluates
    ((sqrt(1 + 0.004(|y|+1.001)) - 1)/0.002)^2 - 1 in fp32: sqrt(t) lies in [1, 2)
    where one ulp is 2^-23, so the reference's own outputs sit on a grid of spacing
    q(v) = 2 * 2^-23 / 0.002 * sqrt(|v|+1) = 1.19e-4 * sqrt(|v|+1)  (8e-5 relative at
    v = 3).  A 1-ulp difference in the support expectation can move the result to the
    neighbouring grid point, so scalars are compared to 1e-5 relative plus one such step."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    q = 2.0 * 2.0 ** -23 / 0.002 * np.sqrt(np.abs(b) + 1.0)
    return np.abs(a - b) / (REL * np.abs(b) + 1.05 * q)


def hidden_err(a, b):
    """Hidden states are min-max scaled to [0, 1] (neural_network_mlp_model.py:135-143):
    the error is measured against the row scale (1.0), not per element."""
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))


def table_of(fx):
    return dict(root_policy=fx["root_policy"], sim_reward=fx["sim_reward"],
                sim_value=fx["sim_value"], sim_prior=fx["sim_prior"])


def leaves_of(fx):
    """node index expanded at each simulation, from the fixture's (parent, action)"""
    A, S = int(fx["A"]), int(fx["S"])
    N = len(fx["seed"])
    leaf = np.zeros((N, S), np.int32)
    for g in range(N):
        base = {0: 1}
        for s in range(S):
            par, act = int(fx["sim_parent"][g, s]), int(fx["sim_action"][g, s])
            leaf[g, s] = base[par] + act
            base[leaf[g, s]] = 1 + (s + 1) * A
    return leaf


@pytest.mark.parametrize("fname,tag", SEARCH_FIXTURES)
def test_L1_search_injected_bit_exact(fname, tag):
    """Search logic alone (reference model outputs injected): everything bit-exact."""
    fx = load_golden(fname)
    d, _ = fixture_weights(tag)
    S = int(fx["S"])
    dirichlet, noise = fixture_streams(fx)
    out = mzo.search(d, None, S, float(fx["gamma"]), float(fx["frac"]), fx["pbc_table"],
                     dirichlet=dirichlet, noise=noise, train=bool(fx["train"]),
                     table=table_of(fx),
                     want=("visits", "root_value", "root_prior", "consumed", "path_len",
                           "sim_leaf", "node_visit", "node_vsum", "min_margin"))
    if S > 0:
        assert np.array_equal(out["visits"].astype(np.float64), fx["visits"])
    assert np.array_equal(out["root_prior"], fx["root_prior"])
    assert np.array_equal(out["root_value"], fx["root_value"])          # bitwise fp32
    assert np.array_equal(out["consumed"], fx["n_draws"])
    assert np.array_equal(out["path_len"], fx["total_path_len"])
    assert np.array_equal(out["sim_leaf"], leaves_of(fx))
    assert np.array_equal(out["node_visit"], fx["child_visit"])
    assert np.array_equal(out["node_vsum"], fx["child_vsum"])
    if S > 1:
        m = np.isfinite(fx["min_margin"])
        assert np.allclose(out["min_margin"][m], fx["min_margin"][m], rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag", MODEL_TAGS)
def test_L2_model_functions(tag):
    """Model half under the mzs_numeric.h contract vs the reference wrappers."""
    d, blob = fixture_weights(tag)
    mv = load_golden(f"model_L2_{tag}.npz")
    h0 = mzo.representation(d, blob, mv["obs"])
    assert hidden_err(h0, mv["repr_hidden"]).max() < REL
    p, v = mzo.prediction(d, blob, mv["repr_hidden"])
    assert rel_err(p, mv["ini_policy"]).max() < REL
    assert value_err(v, mv["ini_value"]).max() <= 1
    r, nh = mzo.dynamics(d, blob, mv["hidden"], mv["action"])
    assert value_err(r, mv["dyn_reward"]).max() <= 1
    assert hidden_err(nh, mv["dyn_hidden"]).max() < REL
    p, v = mzo.prediction(d, blob, mv["hidden"])
    assert rel_err(p, mv["pred_policy"]).max() < REL
    assert value_err(v, mv["pred_value"]).max() <= 1
    # chained, as the search uses it
    r, nh, p, v = mzo.recurrent_inference(d, blob, mv["hidden"], mv["action"])
    assert rel_err(p, mv["rec_policy"]).max() < REL
    assert value_err(v, mv["rec_value"]).max() <= 1
    # building blocks
    assert hidden_err(mzo.scale_to_bound(mv["scale_in"]), mv["scale_out"]).max() < REL
    assert value_err(mzo.support_to_scalar(mv["inv_in"]), mv["inv_out"]).max() <= 1


def test_exp_contract_accuracy():
    xs = np.concatenate([np.linspace(-87, 0, 20001), np.linspace(0, 10, 2001)]).astype(np.float32)
    got = np.array([mzo.lib().mzo_expf(float(x)) for x in xs], np.float64)
    want = np.exp(xs.astype(np.float64))
    assert (np.abs(got - want) / want).max() < 2.5e-7
    # below the clamp the result stays e^-87 (1.6e-38): negligible against a softmax's leading 1
    assert mzo.lib().mzo_expf(-100.0) == mzo.lib().mzo_expf(-87.0) < 2e-38
    assert mzo.lib().mzo_elu(-200.0) == -1.0
    xs = -np.logspace(-8, 1.9, 4000).astype(np.float32)
    got = np.array([mzo.lib().mzo_elu(float(x)) for x in xs], np.float64)
    want = np.expm1(xs.astype(np.float64))
    assert (np.abs(got - want) / np.abs(want)).max() < 4e-7
    assert mzo.lib().mzo_elu(3.5) == 3.5 and mzo.lib().mzo_elu(0.0) == 0.0


@pytest.mark.parametrize("fname,tag", SEARCH_FIXTURES)
def test_L3_search_end_to_end(fname, tag):
    """Full search (oracle model + oracle search) vs the reference run."""
    fx = load_golden(fname)
    d, blob = fixture_weights(tag)
    S = int(fx["S"])
    dirichlet, noise = fixture_streams(fx)
    out = mzo.search(d, blob, S, float(fx["gamma"]), float(fx["frac"]), fx["pbc_table"],
                     obs=fixture_obs(fx), dirichlet=dirichlet, noise=noise, train=bool(fx["train"]))
    assert rel_err(out["root_prior"], fx["root_prior"]).max() < REL
    if S == 0:
        return
    m = check_L3_against_reference(fname, fx, out["visits"], out["root_value"])
    print(f"\n{fname}: visit counts identical {m['matched']}/{m['searches']} (of which near-ties "
          f"{m['near_ties']}), root value max rel err {m['root_value_rel']:.2e}")


def test_oracle_threads_agree():
    fx = load_golden("search_tag123_S11.npz")
    d, blob = fixture_weights("tag123")
    dirichlet, noise = fixture_streams(fx)
    kw = dict(obs=fx["obs"], dirichlet=dirichlet, noise=noise, train=True)
    a = mzo.search(d, blob, 11, 0.997, 0.25, fx["pbc_table"], n_threads=1, **kw)
    b = mzo.search(d, blob, 11, 0.997, 0.25, fx["pbc_table"], n_threads=4, **kw)
    for k in a:
        assert np.array_equal(a[k], b[k])


def test_oracle_sanitized():
    """the CPU restatement under -fsanitize=address,undefined.  Builds
    oracle/libmzoracle_san.so and runs the oracle pin tests (search, model, training, self-play, vision,
    bf16) in a child interpreter that preloads the sanitizer runtimes; any report aborts the child."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("MZO_ORACLE_SO"):
        pytest.skip("already inside the sanitized child")
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    rts = []
    for name in ("libasan.so", "libubsan.so"):
        path = subprocess.check_output([gcc, f"-print-file-name={name}"], text=True).strip()
        if not os.path.isabs(path) or not os.path.exists(path):
            pytest.skip(f"{name} not installed")
        rts.append(os.path.realpath(path))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    odir = os.path.join(root, "oracle")
    subprocess.check_call(["make", "-C", odir, "libmzoracle_san.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update(MZO_ORACLE_SO=os.path.join(odir, "libmzoracle_san.so"), LD_PRELOAD=":".join(rts),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    files = ["test_oracle_golden.py", "test_train_oracle.py", "test_selfplay_oracle.py", "test_vision_oracle.py",
             "test_bf16_oracle.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"]
                       + [os.path.join(root, "tests", f) for f in files],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
