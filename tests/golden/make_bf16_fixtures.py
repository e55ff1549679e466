#!/usr/bin/env python3
"""Reference run with type_format=torch.bfloat16 (container-only; imports /root/reference by path
through make_fixtures.py's harness; no reference source is copied).

What the reference does in that mode (muzero_model.py:194-229, :320-324, :828-905): "bfloat16"
contains the substring "float16", so use_amp becomes True, the three nets KEEP fp32 parameters
(model_without_amp is skipped) and every inference wrapper runs under
torch.autocast(device_type="cpu", dtype=torch.bfloat16).  Rewards go through
inverse_transform_with_support in fp32 (`reward.type(torch.float)`, :866), values do NOT
(:903: the transform runs on the bfloat16 logits outside autocast) and collapse to -1.0:
sqrt(1 + 0.004 (|y| + 1.001)) rounds to 1.0 in bfloat16.

The fixture records those outputs next to the inputs of model_L2_tag123.npz so the tests can show
(a) the bf16 contract of this repository (include/mzs_numeric.h) agrees with the reference's
bf16 policies / rewards / hidden states to bf16 accuracy, and (b) the reference's bf16 values are
the constant -1 — which is why the bf16 variant is held to the reference's fp32 outputs (1e-2),
as BASELINE.json's north_star words it, and not to its bf16 ones.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_bf16_fixtures.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fixtures as mf  # noqa: E402  (installs the gymnasium stub, imports the reference)


def load_checkpoint_bf16(tag):
    mz = mf.mm.Muzero(load=True, type_format=torch.bfloat16)
    cwd = os.getcwd()
    os.chdir(mf.REF)
    try:
        with torch.serialization.safe_globals(mf.ALLOW):
            mz.load_model(tag=tag, observation_space_dimensions=None, device="cpu", type_format=torch.bfloat16)
    finally:
        os.chdir(cwd)
    return mz


def main():
    torch.set_num_threads(1)
    for tag in (123, 133):
        mz = load_checkpoint_bf16(tag)
        assert mz.use_amp and mz.type_format is torch.bfloat16
        assert next(mz.dynamics_function.parameters()).dtype is torch.float32
        torch.set_default_dtype(torch.float32)
        mv = dict(np.load(os.path.join(mf.OUT, f"model_L2_tag{tag}.npz")))
        n = len(mv["action"])
        out = dict(obs=mv["obs"], hidden=mv["hidden"], action=mv["action"],
                   repr_hidden=np.zeros_like(mv["repr_hidden"]), dyn_reward=np.zeros(n, np.float32),
                   dyn_hidden=np.zeros_like(mv["dyn_hidden"]), pred_policy=np.zeros_like(mv["pred_policy"]),
                   pred_value=np.zeros(n, np.float32), rec_policy=np.zeros_like(mv["rec_policy"]),
                   rec_value=np.zeros(n, np.float32))
        for i in range(n):
            h0 = mz.representation_function_inference(mv["obs"][i:i + 1])
            assert h0.dtype is torch.bfloat16
            out["repr_hidden"][i] = h0.float().numpy()[0]
            r, h2 = mz.dynamics_function_inference(torch.tensor(mv["hidden"][i:i + 1]), int(mv["action"][i]))
            out["dyn_reward"][i], out["dyn_hidden"][i] = r, h2.float().numpy()[0]
            p, v = mz.prediction_function_inference(torch.tensor(mv["hidden"][i:i + 1]))
            out["pred_policy"][i], out["pred_value"][i] = p[0], v
            p, v = mz.prediction_function_inference(h2)
            out["rec_policy"][i], out["rec_value"][i] = p[0], v
        mf.save(f"model_L2_tag{tag}_refbf16.npz", **out)
        print(f"tag{tag}: reference bf16 values: min {out['pred_value'].min()} max {out['pred_value'].max()}")


if __name__ == "__main__":
    main()
