#!/usr/bin/env python3
"""Secondary measurement: time of mzs_fill_streams (device-side RandomState streams) for the
benchmark shape, next to the host loop it replaces and to the search itself."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import rng  # noqa: E402
from muzero_amd.engine import SearchEngine  # noqa: E402

G, S, A = 4096, 50, 2
w = np.load("tests/golden/weights_tag123.npz")
dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L"))
eng = SearchEngine(dims, G, S, discount=0.997)
eng.set_weights(w["blob"])
seeds = torch.arange(G, dtype=torch.int64, device=eng.device)
obs = torch.as_tensor(np.random.RandomState(1).uniform(-0.05, 0.05, (G, 4)).astype(np.float32)).to(eng.device)
for _ in range(3):
    d, n = eng.fill_streams(seeds, S, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    d, n = eng.fill_streams(seeds, S, True)
torch.cuda.synchronize()
fill_ms = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter()
for _ in range(20):
    d, n = eng.fill_streams(seeds, S, True)
    eng.search(obs, d, n, S, True)
torch.cuda.synchronize()
both_ms = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter()
rng.noise_streams(np.arange(256), A, S, 0.25, True)
host_ms = (time.perf_counter() - t0) * 1e3 * G / 256
print(json.dumps({"games": G, "sims": S, "fill_streams_ms": fill_ms, "fill_plus_search_ms": both_ms,
                  "host_randomstate_loop_ms_extrapolated": host_ms,
                  "sims_per_s_with_device_streams": G * S / (both_ms * 1e-3)}))
