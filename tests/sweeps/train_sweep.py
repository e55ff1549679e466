#!/usr/bin/env python3
"""Randomised parity sweep of the training step on one MI355X: for a time budget, draw a model shape
(mlp_model and lstm_model), an unroll length, a batch size, a loss type, targets and weights; run
mzs_train_grad through the C-ABI and compare loss terms, total loss, the flat gradient and the priorities
with the CPU oracle's train_grad on the same inputs (fp32 tolerances of tests/test_train_oracle.py:
1e-5 relative on the losses, 1e-5 of the largest gradient entry — 1e-3 under the MSE criteria, see below).  Test infrastructure like tests/.
(A three-entry hidden state — S_dim 3 — was in the list once: its min-max normalisation is so ill-conditioned
that rounding differences reached 3e-5 of the largest gradient entry after five unroll steps; replaced by S_dim 9.)
Prints one JSON line per case and a summary; exits non-zero on the first case out of tolerance.
Usage: python tests/sweeps/train_sweep.py [--seconds 60] [--seed 0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

LOSS_REL, GRAD_REL = 1e-5, 1e-5
# (obs_dim, A, S_dim, H, L, kind)
SHAPES = [
    (4, 2, 31, 64, 0, 0), (4, 2, 31, 64, 1, 0), (4, 2, 31, 64, 4, 0), (8, 4, 31, 64, 1, 0), (4, 2, 11, 16, 2, 0),
    (4, 2, 61, 128, 1, 0), (5, 7, 19, 24, 1, 0), (3, 2, 63, 40, 0, 0), (9, 33, 17, 33, 2, 0), (2, 5, 9, 8, 3, 0),
    (4, 2, 31, 64, 1, 1), (4, 2, 31, 64, 2, 1), (6, 3, 15, 32, 1, 1),
]


def minmax_tie(mzo, d, blob, obs, act, eps=1e-6):
    """mlp_model: True if along the unroll some hidden state attains its minimum or maximum twice (to eps of
    the range).  scale_to_bound's backward routes the gradient of min / max to ONE entry; at a tie the choice —
    and with it a whole branch of the gradient — hangs on the last bit of the forward pass, which the training
    kernels do not share with the oracle bit for bit (seen: 1 case in 24 000, exact tie in the oracle's forward,
    17 % difference in the state head's gradient).  Such cases are counted and skipped, not compared."""
    h = mzo.representation(d, blob, obs)
    for k in range(act.shape[0] + 1):
        s = np.sort(h, axis=1)
        if (s[:, 1] - s[:, 0]).min() < eps or (s[:, -1] - s[:, -2]).min() < eps:
            return True
        if k < act.shape[0] - 1:
            h = mzo.dynamics(d, blob, h, act[k].astype(np.int32))[1]
        else:
            break
    return False


def run(seconds, seed=0, log=print, only_case=None, keep_going=False):
    from muzero_amd.engine import SearchEngine
    from muzero_amd.model import blob_layout
    from muzero_amd.train import LOSS_TYPES, Trainer
    from oracle import mz_oracle as mzo
    import torch

    t_end = time.time() + seconds
    cases, ties, worst = 0, 0, dict(loss=0.0, grad=0.0, prio=0.0)
    if only_case is not None:
        cases = int(only_case)
    while time.time() < t_end:
        rs = np.random.RandomState((seed * 1000003 + cases) % (2 ** 31))   # a case can be re-run alone (--case)
        shape = SHAPES[rs.randint(len(SHAPES))]
        obs_dim, A, Sd, H, L, kind = shape
        lstm = kind == 1
        K = int(rs.choice([1, 2, 3, 5, 6]))
        B = int(rs.choice([1, 2, 7, 24, 32, 128, 200] if not lstm else [1, 2, 7, 24, 32, 64]))
        loss_type = "general" if lstm else str(rs.choice(sorted(LOSS_TYPES)))
        # weights of per-layer gain 0.5-1.5 (std = gain / sqrt(H)): beyond that a tied stack of four mid layers,
        # unrolled K times, amplifies rounding differences past any fixed tolerance (seen: 1.9e-5 at gain 2.4)
        scale = float(rs.choice([0.5, 1.0, 1.5])) / float(np.sqrt(H))
        d = mzo.dims(obs_dim, A, Sd, H, L, kind)
        n_w = sum(int(np.prod(shp)) for _, shp in blob_layout(obs_dim, A, Sd, H, L, kind))
        blob = (rs.standard_normal(n_w) * scale).astype(np.float32)
        obs = rs.uniform(-1, 1, size=(B, obs_dim)).astype(np.float32)
        act = rs.randint(0, A, size=(K, B))
        actions = np.zeros((K, B, A), np.float32)
        actions[np.arange(K)[:, None], np.arange(B)[None, :], act] = 1.0
        half = (Sd - 1) // 2
        # scalar targets inside the support's range; general_kkc takes the log of a single-bin weight that is
        # negative for negative targets (NaN in the reference too): keep to non-negative ones there
        lo = 0.0 if loss_type == "general_kkc" else -0.8 * half
        tv = rs.uniform(lo, 0.8 * half, size=(K, B)).astype(np.float32)
        tr_ = rs.uniform(lo if loss_type == "general_kkc" else -1.0, 1.0, size=(K, B)).astype(np.float32)
        tp = rs.dirichlet(np.ones(A), size=(K, B)).astype(np.float32)
        isr = None if rs.randint(2) else rs.uniform(0.2, 1.0, size=B).astype(np.float32)
        if not lstm and only_case is None and minmax_tie(mzo, d, blob, obs, act):
            log(json.dumps(dict(case=cases, shape=shape, K=K, B=B, skipped="min/max tie in a hidden state")))
            ties += 1
            cases += 1
            continue
        eng = SearchEngine(shape, max_games=1, max_simulations=0)
        eng.set_weights(blob)
        tr = Trainer(eng, K, B, "adam", 1e-3, None, 100, loss_type)
        tr.grad_step(obs, actions, tv, tr_, tp, isr)
        torch.cuda.synchronize()
        loss = tr.loss.cpu().numpy()
        got = dict(grad=tr.grad.cpu().numpy(), loss_terms=loss[:-1], total_loss=float(loss[-1]),
                   priority=tr.priority[:, :B].cpu().numpy())
        ref = mzo.train_grad(d, blob, obs, actions, tv, tr_, tp, isr, loss_type=loss_type)
        # a term is a sum over the batch of per-sample losses of either sign (the reference's single-bin
        # targets are negative for negative scalars): where they cancel, the term's own size says nothing
        # about the size of what was added up — measure against the largest term as well
        tscale = 0.1 * float(np.abs(ref["loss_terms"]).max())
        e_loss = max(abs(got["total_loss"] - ref["total_loss"]) / max(abs(ref["total_loss"]), tscale),
                     float(np.max(np.abs(got["loss_terms"] - ref["loss_terms"]) /
                                  (np.abs(ref["loss_terms"]) + tscale))))
        gmax = float(np.abs(ref["grad"]).max())
        e_grad = float(np.abs(got["grad"] - ref["grad"]).max()) / max(gmax, 1e-30)
        # priority = |v - target| with v from the reference's fp32 inverse transform, which quantises v to
        # ~1.2e-4 * sqrt(|v| + 1) (tests/test_train_oracle.py prio_tol): one ulp in the expectation moves v
        # by one such step; the allowance is two steps, expressed in units of it
        vmag = np.abs(ref["priority"]) + np.abs(tv)
        e_prio = float(np.max(np.abs(got["priority"] - ref["priority"]) /
                              (1e-5 * (1 + vmag) + 4.8e-4 * np.sqrt(vmag + 1)))) * 1e-5
        finite = bool(np.isfinite(ref["grad"]).all() and np.isfinite(got["grad"]).all())
        # MSE criteria ('game', 'game_mmc') differentiate (v - target)^2 with v from the same quantised inverse
        # transform: one step in v (above) is a relative change of step / |v - target| in that term's whole
        # gradient — unbounded as v nears its target (seen: 1.9e-5 typically, 1.7e-4 of the largest entry with
        # |v - target| = 0.04), so their bound is 1e-3: still far below what a wrong formula would show
        grad_rel = 1e-3 if loss_type in ("game", "game_mmc") else GRAD_REL
        ok = finite and e_loss <= LOSS_REL and e_grad <= grad_rel and e_prio <= 1e-5
        eng.close()
        rec = dict(case=cases, shape=shape, K=K, B=B, loss_type=loss_type, scale=scale, is_ratio=isr is not None,
                   e_loss=e_loss, e_grad=e_grad, e_prio=e_prio, ok=bool(ok))
        log(json.dumps(rec))
        if not ok or only_case is not None:
            per, off = {}, 0
            for name, shp in blob_layout(obs_dim, A, Sd, H, L, kind):
                n = int(np.prod(shp))
                g, r = got["grad"][off:off + n], ref["grad"][off:off + n]
                per[name] = [float(np.abs(g - r).max()), float(np.abs(r).max())]
                off += n
            rec["per_tensor_maxerr_maxref"] = {k: v for k, v in per.items() if v[0] > 1e-5 * gmax}
            if keep_going and only_case is None:
                log(json.dumps(dict(FAIL=True, **rec)))
                cases += 1
                continue
            return dict(summary="all within tolerance" if ok else "OUT OF TOLERANCE", seed=seed, got_terms=got["loss_terms"].tolist(),
                        ref_terms=ref["loss_terms"].tolist(), got_total=got["total_loss"], ref_total=ref["total_loss"],
                        **rec)
        worst = dict(loss=max(worst["loss"], e_loss), grad=max(worst["grad"], e_grad), prio=max(worst["prio"], e_prio))
        cases += 1
    return dict(summary="all within tolerance", seed=seed, cases=cases, skipped_minmax_ties=ties, worst_rel_loss=worst["loss"],
                worst_rel_grad=worst["grad"], worst_rel_priority=worst["prio"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--case", type=int, default=None, help="run this case of the seed alone, with per-tensor errors")
    ap.add_argument("--keep-going", action="store_true", help="log cases out of tolerance and carry on")
    a = ap.parse_args()
    out = run(a.seconds, a.seed, log=lambda s: print(s, flush=True), only_case=a.case, keep_going=a.keep_going)
    print(json.dumps(out), flush=True)
    sys.exit(0 if out["summary"] == "all within tolerance" else 1)


if __name__ == "__main__":
    main()
