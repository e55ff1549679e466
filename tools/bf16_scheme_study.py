#!/usr/bin/env python3
"""Which bf16 scheme keeps ONE recurrent inference within 1e-2 of the reference's fp32 outputs on all six
weight sets?  Emulation in float64 (accumulation error is not the question; operand rounding is):
  w16/x16   weights and layer inputs rounded to bf16 (round 2's contract)
  w16/xhl   weights bf16, inputs as hi + lo bf16 pair (two products per weight)
  whl/xhl   weights hi + lo, inputs hi + lo, the lo*lo product dropped (three products per weight)
Prints the maxima per tag (policy / value / reward relative, hidden absolute).  CPU only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from muzero_amd.model import blob_layout  # noqa: E402


def bf(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)


def split(x, n):
    """x as a sum of n bf16 terms (float64 arrays)"""
    out, r = [], np.asarray(x, np.float64)
    for _ in range(n):
        h = bf(r)
        out.append(h)
        r = r - h
    return out


def dense(W, b, x, nw, nx):
    ws, xs = split(W, nw), split(x, nx)
    y = np.zeros((x.shape[0], W.shape[0]))
    for i, w in enumerate(ws):
        for j, xx in enumerate(xs):
            if i + j < max(nw, nx):          # drop lo*lo
                y += xx @ w.T
    return y + b


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def softmax(z):
    e = np.exp(z - z.max(1, keepdims=True))
    return e / e.sum(1, keepdims=True)


def inv(logits):
    S = logits.shape[1]
    y = (softmax(logits) * (np.arange(S) - S // 2)).sum(1)
    return np.sign(y) * (((np.sqrt(1 + 0.004 * (np.abs(y) + 1.001)) - 1) / 0.002) ** 2 - 1)


def scale(x):
    mn, mx = x.min(1, keepdims=True), x.max(1, keepdims=True)
    sc = mx - mn
    sc = np.where(sc < 1e-5, sc + 1e-5, sc)
    return (x - mn) / sc


def recurrent(w, L, A, hidden, action, nw, nx):
    x = np.concatenate([hidden, np.eye(A)[action]], 1)
    # the one-hot columns stay fp32 in the contract: add them exactly
    Win = w["d_in_w"]
    t = dense(Win[:, :-A], w["d_in_b"], hidden, nw, nx) + Win[:, -A:][:, action].T
    t = elu(t)
    for _ in range(L):
        t = elu(dense(w["d_mid_w"], w["d_mid_b"], t, nw, nx))
    rew = inv(dense(w["d_rew_w"], w["d_rew_b"], t, nw, nx))
    nh = scale(dense(w["d_st_w"], w["d_st_b"], t, nw, nx))
    t = elu(dense(w["p_in_w"], w["p_in_b"], nh, nw, nx))
    for _ in range(L):
        t = elu(dense(w["p_mid_w"], w["p_mid_b"], t, nw, nx))
    pol = softmax(dense(w["p_pol_w"], w["p_pol_b"], t, nw, nx))
    val = inv(dense(w["p_val_w"], w["p_val_b"], t, nw, nx))
    return rew, nh, pol, val


def rel(a, b):
    return (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max()


def main():
    for tag in ("tag123", "tag133", "tag127", "tag137", "synA4", "synA3"):
        wz = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{tag}.npz"))
        dims = [int(wz[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")]
        w, off = {}, 0
        for name, shp in blob_layout(*dims, 0):
            n = int(np.prod(shp))
            w[name] = wz["blob"][off:off + n].reshape(shp).astype(np.float64)
            off += n
        mv = np.load(os.path.join(ROOT, "tests", "golden", f"model_L2_{tag}.npz"))
        line = [f"{tag:7s} |w|max {max(np.abs(v).max() for v in w.values()):6.2f}"]
        for nw, nx in ((8, 8), (1, 1), (1, 2), (2, 2), (1, 3)):
            r, nh, p, v = recurrent(w, dims[4], dims[1], mv["hidden"].astype(np.float64), mv["action"], nw, nx)
            e = max(rel(p, mv["rec_policy"]), rel(v, mv["rec_value"]), rel(r, mv["dyn_reward"]),
                    np.abs(nh - mv["dyn_hidden"]).max())
            line.append(f"w{nw}x{nx}: {e:.2e}")
        print("  ".join(line))


if __name__ == "__main__":
    main()
