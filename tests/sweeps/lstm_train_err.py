#!/usr/bin/env python3
"""GPU-vs-oracle error of the 'lstm_model' training step on the three reference fixtures (the margin
under the 1e-5 tolerances of tests/test_gpu_train.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from muzero_amd.engine import SearchEngine
    from muzero_amd.train import Trainer
    from oracle import mz_oracle as mzo
    from test_train_oracle import LSTM_TRAIN_FIXTURES, step_inputs
    for f in LSTM_TRAIN_FIXTURES:
        fx = np.load(os.path.join(ROOT, "tests", "golden", f))
        d = mzo.dims(*fx["dims"])
        eng = SearchEngine(tuple(int(x) for x in fx["dims"]), max_games=1, max_simulations=0)
        eng.set_weights(fx["blob0"])
        tr = Trainer(eng, int(fx["K"]), int(fx["B"]), "adam", 0.01, None, 100)
        inp = step_inputs(fx, 0)
        tr.grad_step(inp["obs"], inp["actions"], inp["target_value"], inp["target_reward"], inp["target_policy"],
                     inp["is_ratio"])
        torch.cuda.synchronize()
        ref = mzo.train_grad(d, fx["blob0"], **inp)
        g, gr, gf = tr.grad.cpu().numpy(), ref["grad"], fx["s0_grad"]
        loss = tr.loss.cpu().numpy()
        print(f, "grad vs oracle %.2e  vs reference %.2e  loss vs oracle %.2e  terms %.2e" % (
            np.abs(g - gr).max() / np.abs(gr).max(), np.abs(g - gf).max() / np.abs(gf).max(),
            abs(loss[-1] - ref["total_loss"]) / ref["total_loss"],
            (np.abs(loss[:-1] - ref["loss_terms"]) / np.abs(ref["loss_terms"])).max()), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
