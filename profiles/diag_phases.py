#!/usr/bin/env python3
"""Diagnostic (not a benchmark): per-phase share of wave cycles in k_search_quad / k_search_quadw,
from a -DMZS_DIAG build of the library (s_memtime stamps).  Usage on the GPU box:
MZS_LIB=$PWD/muzero_amd/libmzsearch_diag.so python profiles/diag_phases.py [weights tag] [precision]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import _lib, rng  # noqa: E402
from muzero_amd.engine import SearchEngine  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "tag123"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # 0 fp32, 1 fast bf16, 2 bf16x3
w = np.load(f"tests/golden/weights_{tag}.npz")
dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (int(w["kind"]) if "kind" in w.files else 0, prec)
G, S = 4096, 50
obs = np.random.RandomState(1000).uniform(-0.05, 0.05, size=(G, dims[0])).astype(np.float32)
dirichlet, noise = rng.noise_streams(np.arange(G), 2, S, 0.25, True)
eng = SearchEngine(dims, G, S, discount=0.997)
eng.set_weights(w["blob"])
lib = _lib.load()
out = (C.c_ulonglong * 8)()
for it in range(3):
    eng.search(obs, dirichlet, noise, S, True)
    eng.results_host()
    lib.mzs_debug_counters(out)
v = np.array(list(out), dtype=np.float64)
# one wave = 4 games; counters are summed over waves (lane 0 of each)
if tag == "tag123":
    v[3] -= v[2]   # k_search_quad stamps dynamics + prediction together in slot 3
names = ["-", "select", "dynamics", "prediction", "expand+backup", "-", "-", "-"]
if tag == "tag127":   # k_search_quadw2: slots 6 / 7 are parts of slot 3
    names = ["    of the units: wait for weights / x rows", "select", "wait for the block (tree imbalance)", "dynamics + prediction", "expand+backup", "    of the units: matrix blocks",
             "  of it: barrier waits", "  of it: dense units (x loads, requests, MFMAs)"]
tot = v[1:5].sum()
W = G / 4
for n, x in zip(names, v):
    if x:
        print(f"{n:22s} {x / (W * S):10.1f} cycles per wave-sim (4 games)   {100 * x / tot:5.1f} %")
print(f"total {tot / (W * S):.1f} cycles per wave-sim (stamps cost ~40 cycles each)")
