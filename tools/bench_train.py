#!/usr/bin/env python3
"""Secondary measurement (not the headline metric): wall time of one training step
(mzs_train_grad + mzs_train_apply) on one MI355X, BASELINE config C1 shape: CartPole MLP,
batch 128, unroll 5 (and the reference config's unroll 10).  Prints one JSON line per case."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd.engine import SearchEngine  # noqa: E402
from muzero_amd.train import Trainer  # noqa: E402


def main():
    w = np.load("tests/golden/weights_tag123.npz")
    dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L"))
    for K, B in ((5, 128), (10, 128), (5, 1024)):
        rs = np.random.RandomState(0)
        eng = SearchEngine(dims, 1, 0)
        eng.set_weights(w["blob"])
        tr = Trainer(eng, K, B, "adam", 0.01, "cosineannealinglr", 10000)
        dev = eng.device
        obs = torch.as_tensor(rs.uniform(-1, 1, (B, dims[0])).astype(np.float32)).to(dev)
        act = torch.zeros((K, B, dims[1]), device=dev)
        act[..., 0] = 1.0
        tv = torch.as_tensor(rs.uniform(0, 50, (K, B)).astype(np.float32)).to(dev)
        trw = torch.ones((K, B), device=dev)
        tp = torch.full((K, B, dims[1]), 1.0 / dims[1], device=dev)
        isr = torch.ones(B, device=dev)
        for _ in range(5):
            tr.step(obs, act, tv, trw, tp, isr)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(obs, act, tv, trw, tp, isr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(json.dumps({"metric": "training step wall time", "unroll": K, "batch": B,
                          "ms_per_step": 1e3 * dt, "samples_per_s": B / dt,
                          "note": "forward+losses+backward+wgrad+Adam, 1 GPU, inputs resident"}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
