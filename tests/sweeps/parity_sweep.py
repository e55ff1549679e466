#!/usr/bin/env python3
"""Randomised parity sweep on one MI355X: for a time budget, draw a model shape, a weight scale, a game
count, a simulation count and seeds; run the search through the C-ABI with device-made streams
(mzs_search_streams: prefix + redo path included) and compare every game with the CPU oracle on the same
streams — visit counts, root values, consumed draws, expansion order and every node's value sum, bit for
bit.  Test infrastructure like tests/: the oracle is the checker, never the product path.
Prints one JSON line per case and a summary line; exits non-zero on the first mismatch.
Usage: python tests/sweeps/parity_sweep.py [--seconds 120] [--seed 0] [--threads 16]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# (obs_dim, A, S_dim, H, L, kind): the shapes the specialised kernels exist for, and others that fall to the generic one
SHAPES = [
    (4, 2, 31, 64, 0, 0), (4, 2, 31, 64, 1, 0), (4, 2, 31, 64, 4, 0), (8, 4, 31, 64, 0, 0), (8, 4, 31, 64, 1, 0),
    (6, 3, 31, 64, 0, 0), (4, 2, 11, 16, 0, 0), (4, 2, 11, 16, 2, 0), (4, 2, 61, 128, 1, 0), (4, 2, 61, 128, 0, 0),
    (4, 2, 31, 64, 1, 1), (4, 2, 31, 64, 2, 1),                       # lstm_model: stream kernel, generic kernel
    (5, 7, 19, 24, 1, 0), (3, 2, 128, 40, 0, 0), (9, 64, 17, 33, 2, 0), (2, 5, 2, 8, 3, 0),   # generic kernel
    # seventh entry 1: bf16 model half (weights and activations rounded to bf16, fp32 accumulation; the
    # oracle restates the same contract)
    (4, 2, 31, 64, 0, 0, 1), (8, 4, 31, 64, 0, 0, 1), (4, 2, 31, 64, 2, 0, 1), (8, 4, 31, 64, 1, 0, 1),
    (5, 7, 19, 24, 1, 0, 1), (6, 3, 31, 64, 0, 0, 1), (4, 2, 11, 16, 2, 0, 1),
    (4, 2, 61, 128, 1, 0, 1), (4, 2, 31, 64, 1, 1, 1),                # bf16 on the weight-stream kernels (wide MLP, LSTM)
    (4, 2, 31, 64, 2, 1, 1), (4, 2, 31, 64, 3, 1, 2), (3, 2, 128, 40, 0, 0, 2),   # bf16 on the any-shape kernel: deeper LSTMs, wide state
    # seventh entry 2: bf16x3 (hi + lo bfloat16 pairs, three products: what type_format=torch.bfloat16 selects)
    (4, 2, 31, 64, 0, 0, 2), (8, 4, 31, 64, 0, 0, 2), (4, 2, 31, 64, 2, 0, 2), (4, 2, 11, 16, 0, 0, 2),
    (4, 2, 61, 128, 1, 0, 2), (4, 2, 31, 64, 1, 1, 2), (5, 7, 19, 24, 1, 0, 2), (6, 3, 31, 64, 1, 0, 2),
    # vision_model (kind 2: obs_dim = the image side): two launches per search (roots, simulations); small game
    # counts, the oracle's ResNet runs on the host
    (98, 2, 31, 64, 1, 2), (98, 4, 21, 32, 2, 2), (98, 3, 17, 70, 1, 2),
]


def random_blob(rs, shape, scale):
    """Seeded random weights of a shape; vision_model: positive batchnorm variances, gains near 1."""
    from muzero_amd.model import blob_layout, vision_blob_layout
    obs_dim, A, Sd, H, L, kind = shape[:6]
    if kind != 2:
        n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(obs_dim, A, Sd, H, L, kind))
        return (rs.standard_normal(n_w) * scale).astype(np.float32)
    parts = []
    for name, shp in vision_blob_layout(A, Sd, H, L):
        n = int(np.prod(shp))
        if name.endswith("bn_var"):
            v = rs.uniform(0.5, 1.5, n)
        elif name.endswith("bn_w"):
            v = rs.uniform(0.8, 1.2, n)
        elif "conv" in name and not name.endswith("_b"):
            v = rs.standard_normal(n) * 0.35
        else:
            v = rs.standard_normal(n) * (1.0 / np.sqrt(max(shp[-1], 1)) if len(shp) == 2 else 0.1)
        parts.append(v.astype(np.float32))
    return np.concatenate(parts)


def run(seconds, seed=0, threads=16, max_games=4096, log=print):
    """The sweep; returns the summary dict (summary == "all equal" or "MISMATCH" + the failing case)."""
    a = argparse.Namespace(seconds=seconds, seed=seed, threads=threads, max_games=max_games)
    from muzero_amd import rng  # noqa: F401
    from muzero_amd.engine import SearchEngine
    from muzero_amd.model import blob_layout
    from oracle import mz_oracle as mzo

    rs = np.random.RandomState(a.seed)
    t_end = time.time() + a.seconds
    cases = games = sims = 0
    kernels = {}
    while time.time() < t_end:
        shape = SHAPES[rs.randint(len(SHAPES))]
        obs_dim, A, Sd, H, L, kind = shape[:6]
        prec = shape[6] if len(shape) > 6 else 0
        scale = float(rs.choice([0.1, 0.3, 1.0, 3.0]))
        S = int(rs.choice([1, 2, 7, 25, 50, 50, 50, 80, 120]))
        if A > 8:
            S = min(S, 25)          # 64 children per node: keep the oracle's share of the budget small
        G = int(rs.choice([1, 3, 17, 64, 255, 1024, a.max_games, a.max_games, 2 * a.max_games + 37]))
        if A > 8:
            G = min(G, 255)
        if kind == 2:
            G, S = min(G, int(rs.choice([1, 5, 17, 40]))), min(S, 25)
        train = bool(rs.randint(4) != 0)
        d = mzo.dims(obs_dim, A, Sd, H, L, kind, prec)
        blob = random_blob(rs, shape, scale)
        obs = (rs.uniform(0, 1, size=(G, 3, obs_dim, obs_dim)) if kind == 2
               else rs.uniform(-1, 1, size=(G, obs_dim))).astype(np.float32)
        seeds = rs.randint(0, 2 ** 31 - 1, size=G).astype(np.int64)
        eng = SearchEngine(shape, max_games=G, max_simulations=S, discount=0.997)
        eng.set_weights(blob)
        name = eng.search_kernel_name(S)
        # a short prefix on some cases, so that the device-side redo pass runs too
        prefix = int(rs.choice([0, 0, 2 * A * max(S // 4, 1)]))
        eng.search_streams(obs, seeds, S, train, prefix_draws=prefix or None)
        out, tree = eng.results_host(), eng.tree_host()
        dirichlet, noise = eng.fill_streams(seeds, S, train)
        ref = mzo.search(d, blob, S, 0.997, 0.25, eng.pbc, obs=obs, dirichlet=None if dirichlet is None else dirichlet.cpu().numpy(),
                         noise=noise.cpu().numpy(), train=train, n_threads=a.threads,
                         want=("visits", "root_value", "consumed", "node_vsum", "sim_leaf"))
        ok = (np.array_equal(out["visits"], ref["visits"])
              and np.array_equal(out["root_value"].view(np.uint32), ref["root_value"].view(np.uint32))
              and np.array_equal(out["consumed"], ref["consumed"])
              and np.array_equal(tree["sim_leaf"][:, :S], ref["sim_leaf"][:, :S])
              and np.array_equal(np.ascontiguousarray(tree["value_sum"]).view(np.uint32),
                                 np.ascontiguousarray(ref["node_vsum"]).view(np.uint32))
              and not tree["status"].any())
        eng.close()
        rec = dict(case=cases, shape=shape, scale=scale, games=G, sims=S, train=train, prefix=prefix, kernel=name,
                   ok=bool(ok))
        log(json.dumps(rec))
        if not ok:
            return dict(summary="MISMATCH", seed=a.seed, **rec)
        cases += 1
        games += G
        sims += G * S
        kernels[name] = kernels.get(name, 0) + G
    return dict(summary="all equal", seed=a.seed, cases=cases, games=games, simulations=sims, games_per_kernel=kernels)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--max-games", type=int, default=4096)
    a = ap.parse_args()
    out = run(a.seconds, a.seed, a.threads, a.max_games, log=lambda s: print(s, flush=True))
    print(json.dumps(out), flush=True)
    sys.exit(0 if out["summary"] == "all equal" else 1)


if __name__ == "__main__":
    main()
