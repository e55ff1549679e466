#!/usr/bin/env python3
"""Throughput of the vision_model (ResNet-v2) search on one MI355X: G games x S simulations through
mzs_search_streams, timed with events on the launch stream.  Not the headline benchmark (bench.py);
prints one JSON line.  Usage: python tools/bench_vision.py [--games 1024] [--sims 50] [--weights visA]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muzero_amd.engine import SearchEngine  # noqa: E402


def vision_flops(A, Sd, H, L):
    """(FLOP of one root, FLOP of one simulation) of the vision_model from its layout (2 per multiply-add):
    3 x 3 convolutions of the Down_sample stack and the residual blocks (three per block, the batchnorms and
    pools not counted), the 1 x 1 head convolutions, the three head MLPs (147 -> H -> (H ->) x L -> n_out).
    neural_network_vision_model.py:41-300."""
    C = 3
    conv = lambda side, ci, co: 2 * side * side * 9 * ci * co  # noqa: E731
    res = lambda side, c: 3 * conv(side, c, c)                  # noqa: E731
    root = conv(49, 3, C // 2) + 2 * res(49, C // 2) + conv(25, C // 2, C) + 2 * res(25, C) + 3 * res(13, C) + res(7, C)
    mlp = lambda n_out: 2 * (147 * H + L * H * H + H * n_out)   # noqa: E731
    pred = L * res(7, C) + 2 * (2 * 49 * C * C) + mlp(Sd) + mlp(A)
    dyn = conv(7, C + 1, C) + L * res(7, C) + 2 * 49 * (C + 1) * C + mlp(Sd)
    return root + pred, dyn + pred


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--weights", default="visA")
    ap.add_argument("--generic", action="store_true", help="k_search_fused (one game per wave) instead of k_search_vis4")
    ap.add_argument("--phases", action="store_true",
                    help="also time the three inference kernels on the same batch (us per call): where a search's time goes")
    a = ap.parse_args()
    w = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{a.weights}.npz"))
    dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (int(w["kind"]),)
    G, S = a.games, a.sims
    from muzero_amd import _lib
    eng = SearchEngine(dims, max_games=G, max_simulations=S, discount=0.997,
                       engine=_lib.ENGINE_FUSED_GENERIC if a.generic else _lib.ENGINE_FUSED)
    eng.set_weights(w["blob"])
    obs = torch.as_tensor(np.random.RandomState(7).uniform(0.0, 1.0, size=(G,) + eng.obs_shape).astype(np.float32)).cuda()
    seeds = torch.arange(G, dtype=torch.int64).cuda()
    eng.search_streams(obs, seeds, S, True)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.steps):
        eng.search_streams(obs, seeds, S, True)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.steps
    out = dict(model="vision_model", weights=a.weights, dims=dims, games=G, sims=S, ms_per_search=ms,
               sims_per_s=G * S / (ms * 1e-3), kernel=eng.search_kernel_name(S))
    f_root, f_sim = vision_flops(dims[1], dims[2], dims[3], dims[4])
    tf = G * (f_root + S * f_sim) / (ms * 1e-3) / 1e12
    # fp32 vector and fp32 matrix peaks of the MI355X are both 157.3 TFLOP/s (MI355X_MICROARCH.md): the convolutions
    # run on the vector ALUs, the head MLPs (59 % of a simulation's FLOP at H = 64) on the matrix core
    out["roofline"] = dict(bound="mfma", achieved=tf, peak=157.3, unit="TFLOP/s", frac=tf / 157.3, traffic=None,
                           flop_per_root=f_root, flop_per_simulation=f_sim,
                           note="both launches of a search (roots + simulations), events on the launch stream")
    if a.phases:
        def timed(fn, n=20):
            fn()
            torch.cuda.synchronize()
            t0.record()
            for _ in range(n):
                fn()
            t1.record()
            torch.cuda.synchronize()
            return 1e3 * t0.elapsed_time(t1) / n
        hid, _, _ = eng.initial_inference(obs)
        act = torch.zeros(G, dtype=torch.int32).cuda()
        out["us_initial"] = timed(lambda: eng.initial_inference(obs))
        out["us_representation"] = timed(lambda: eng.initial_inference(obs, want_prediction=False))
        out["us_recurrent"] = timed(lambda: eng.recurrent_inference(hid, act))
        out["us_dynamics"] = timed(lambda: eng.recurrent_inference(hid, act, want_prediction=False))
        out["us_prediction"] = timed(lambda: eng.prediction_inference(hid))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
