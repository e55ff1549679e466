#!/usr/bin/env python3
"""A short loop of 'lstm_model' training steps (config C4 shape, unroll 5, batch 128) for rocprofv3."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import Trainer
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_lstmA.npz"))
    dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L")) + (1,)
    K, B = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 128
    eng = SearchEngine(dims, 1, 0)
    eng.set_weights(w["blob"])
    tr = Trainer(eng, K, B, "adam", 0.01, None, 100)
    rs = np.random.RandomState(0)
    obs = rs.uniform(-1, 1, (B, dims[0])).astype(np.float32)
    act = np.zeros((K, B, dims[1]), np.float32)
    act[..., 0] = 1.0
    tv = rs.uniform(0, 50, (K, B)).astype(np.float32)
    trw = np.ones((K, B), np.float32)
    tp = np.full((K, B, dims[1]), 1.0 / dims[1], np.float32)
    for _ in range(20):
        tr.step(obs, act, tv, trw, tp)
    torch.cuda.synchronize()
    eng.close()


if __name__ == "__main__":
    main()
