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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--weights", default="visA")
    ap.add_argument("--phases", action="store_true",
                    help="also time the three inference kernels on the same batch (us per call): where a search's time goes")
    a = ap.parse_args()
    w = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{a.weights}.npz"))
    dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (int(w["kind"]),)
    G, S = a.games, a.sims
    eng = SearchEngine(dims, max_games=G, max_simulations=S, discount=0.997)
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
