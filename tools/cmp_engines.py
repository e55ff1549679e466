#!/usr/bin/env python3
"""Race screen: the shape's fast search kernel against the generic one-game-per-wave kernel on the
same inputs, three launches in a row (a data race shows as mismatches that move between launches).
usage: python tools/cmp_engines.py <weights tag> <games>"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from muzero_amd import rng, _lib
from muzero_amd.engine import SearchEngine
tag = sys.argv[1]; G = int(sys.argv[2]); S = 50
w = np.load(f"tests/golden/weights_{tag}.npz")
kind = int(w["kind"]) if "kind" in w.files else 0
dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (kind,)
obs = np.random.RandomState(5).uniform(-0.05, 0.05, size=(G, dims[0])).astype(np.float32)
dirichlet, noise = rng.noise_streams(np.arange(G), dims[1], S, 0.25, True)
res = {}
for name, engine in (("fast", _lib.ENGINE_FUSED), ("generic", _lib.ENGINE_FUSED_GENERIC)):
    eng = SearchEngine(dims, G, S, discount=0.997, engine=engine)
    eng.set_weights(w["blob"])
    for it in range(3):
        eng.search(obs, dirichlet, noise, S, True)
        r = eng.results_host()
        res[name, it] = np.array(r["visits"]).copy()
for it in range(3):
    bad = np.nonzero((res["fast", it] != res["generic", 0]).any(axis=1))[0]
    print(tag, "iter", it, "mismatching games:", len(bad), bad[:40], "waves:", sorted(set((bad // 4) % 4))[:8], "blocks:", len(set(bad // 16)))
