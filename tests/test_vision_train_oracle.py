"""The vision training oracle (oracle/mz_vision_train_oracle.py, numpy float64) against the reference's own
Muzero.train on its ResNet-v2 networks: tests/golden/train_vis*.npz (make_vision_train_fixtures.py — two
steps each, BatchNorm on batch statistics, Dropout2d masks as the reference drew them).

Tolerances (float32 reference against a float64 restatement): loss terms 1e-5 relative; a tensor's
gradient |d| <= 1e-5 x the largest gradient of the blob + 1e-4 x the tensor's own largest (the reference's
float32 sums over 49 x 49 x B terms carry that much noise: measured 4e-5 of a BatchNorm bias gradient);
priorities 1e-5 + one grid step of the float32 inverse transform; the blob after the optimizer step (running statistics included) 1e-5 absolute.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mz_vision_train_oracle as vo  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["train_visA_K3_B6", "train_visB_K2_B4"]


def step_inputs(fx, s):
    B = int(fx["B"])
    obs = np.random.RandomState(int(fx[f"s{s}_obs_seed"])).uniform(0.0, 1.0, size=(B, 3, 98, 98)).astype(np.float32)
    isr = fx[f"s{s}_is_ratio"]
    return dict(obs=obs, actions=fx[f"s{s}_actions"], tv=fx[f"s{s}_target_value"], tr=fx[f"s{s}_target_reward"],
                tpol=fx[f"s{s}_target_policy"], is_ratio=None if isr.size == 1 else isr.astype(np.float32),
                masks=fx[f"s{s}_masks"])


def priority_close(got, want, target):
    """|h^-1(value) - target| to 1e-5 + ONE grid step of the reference's float32 inverse transform: its
    sqrt(1 + 0.004 (|y| + 1.001)) is a float32 next to 1, so h^-1 moves in steps of 2 sqrt(|v| + 1) x 2^-23 / 0.002"""
    v = np.abs(target) + np.abs(want)
    return bool((np.abs(got - want) <= 1e-5 * np.maximum(1.0, v) + 2.1 * np.sqrt(v + 1.0) * 2.0 ** -23 / 0.002).all())


def check_grad(dims, got, want, what):
    gmax = np.abs(want).max()
    p = 0
    for name, shape in vo.blob_layout(*[int(v) for v in dims[1:]]):
        n = int(np.prod(shape))
        e = np.abs(got[p:p + n] - want[p:p + n]).max()
        bound = 1e-5 * gmax + 1e-4 * np.abs(want[p:p + n]).max()
        assert e <= bound, f"{what}: gradient of {name}: |d| = {e:.3e} > {bound:.3e}"
        p += n


@pytest.mark.parametrize("case", CASES)
def test_vision_train_oracle_vs_reference(case):
    fx = np.load(os.path.join(GOLDEN, case + ".npz"))
    dims, K = fx["dims"], int(fx["K"])
    blob = fx["blob0"]
    opt = vo.Optimizer(str(fx["optimizer"]), blob.size, dims)
    for s in range(int(fx["n_steps"])):
        r = vo.train_grad(blob, dims, K, **step_inputs(fx, s))
        ref = fx[f"s{s}_losses"]                                   # [loss_nn, terms...]
        assert abs(r["losses"][-1] - ref[0]) <= 1e-5 * abs(ref[0])
        assert np.abs(r["losses"][:-1] - ref[1:]).max() <= 1e-5 * np.abs(ref[1:]).max()
        assert priority_close(r["priority"], fx[f"s{s}_priority"], fx[f"s{s}_target_value"]), f"{case} step {s}"
        check_grad(dims, r["grad"], fx[f"s{s}_grad"], f"{case} step {s}")
        blob = opt.apply(blob, r["grad"], r["running"], float(fx["lr"]))
        assert np.abs(blob - fx[f"s{s}_blob"]).max() <= 1e-5, f"{case} step {s}: weights after the step"
        # the running statistics moved (training mode), and the unused down_sampling=False path's
        # convolution moved too: it sits inside the reference's zip()-truncated l2 term
        lay = {n: i for i, (n, _) in enumerate(vo.blob_layout(*[int(v) for v in dims[1:]]))}
        assert "r_conv" in vo.l2_names(int(dims[4])) and "r_res_in_bn_mean" in lay


def test_l2_set_is_the_first_fifteen_parameters_of_each_network():
    names = vo.l2_names(1)
    assert len(names) == 45
    assert "r_conv" in names and "r_bn_w" not in names           # representation: 17 parameters, 15 counted
    assert "d_fc_out_b" in names                                  # dynamics: all 15
    assert "p_p_in_w" in names and "p_p_in_b" not in names        # prediction: 20 parameters, 15 counted


def test_masks_must_be_consumed_exactly():
    fx = np.load(os.path.join(GOLDEN, CASES[1] + ".npz"))
    kw = step_inputs(fx, 0)
    kw["masks"] = kw["masks"][:-1]
    with pytest.raises(AssertionError):
        vo.train_grad(fx["blob0"], fx["dims"], int(fx["K"]), **kw)
