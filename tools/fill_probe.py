"""Diagnostic: mzs_fill_streams time per call (host-timed, async enqueues: the floor of ~25 us is the
host's own enqueue cost) against the number of draws and with / without the Dirichlet draws."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from muzero_amd.engine import SearchEngine
w = np.load("tests/golden/weights_tag123.npz")
dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L"))
G = 4096
eng = SearchEngine(dims, G, 50, discount=0.997)
eng.set_weights(w["blob"])
seeds = torch.arange(G, dtype=torch.int64, device="cuda")
for S in (0, 10, 25, 35, 50):
    for train in (False, True):
        for _ in range(3): eng.fill_streams(seeds, S, train)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): eng.fill_streams(seeds, S, train)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        from muzero_amd import rng
        print(S, train, "draws", rng.max_draws(2, S), "us", round(dt * 1e6, 1))
