#!/usr/bin/env python3
"""End-to-end sanity of the batched learning cycle on one MI355X: CartPole-v1, the MLP of
experiment_123 (obs 4, A 2, S_dim 31, H 64, L 0), self-play -> replay -> training in a loop.
Prints mean episode reward per iteration; a working system climbs well above the ~22 of a
random policy.  Usage: python tools/learn_cartpole.py [--iterations 30] [--games 2048] ..."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd.engine import SearchEngine  # noqa: E402
from muzero_amd.model import Muzero  # noqa: E402
from muzero_amd.selfplay import Arena, ReplayStore, SelfPlay, temperature_scheduler  # noqa: E402
from muzero_amd.train import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--games", type=int, default=2048, help="games per iteration")
    ap.add_argument("--slots", type=int, default=2048)
    ap.add_argument("--sims", type=int, default=25)
    ap.add_argument("--train-steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--unroll", type=int, default=5)
    ap.add_argument("--td", type=int, default=10)
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--window", type=int, default=8192)
    ap.add_argument("--temperature", default="static_one_temperature", help="self_play.py temperature_scheduler mode")
    ap.add_argument("--scheduler", default=None, help="lr scheduler: cosineannealinglr, steplr, ...")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16x3"],
                    help="search / inference precision (training stays fp32): bf16x3 = type_format=torch.bfloat16's default")
    a = ap.parse_args()
    torch.manual_seed(a.seed)
    np.random.seed(a.seed)
    mz = Muzero(observation_space_dimensions=4, action_space_dimensions=2, state_space_dimensions=31,
                hidden_layer_dimensions=64, number_of_hidden_layer=0, k_hypothetical_steps=a.unroll,
                **({} if a.dtype == "f32" else dict(type_format=torch.bfloat16,
                                                    bf16_mode="fast" if a.dtype == "bf16" else "x3")))
    eng = SearchEngine(mz.dims, max_games=a.slots, max_simulations=a.sims, discount=0.997)
    eng.set_weights(mz.weights_blob())
    tr = Trainer(eng, a.unroll, a.batch, "adam", a.lr, a.scheduler, a.iterations * a.train_steps)
    store = ReplayStore(a.window, a.batch, a.unroll, a.td, "priority", "priority", discount=0.997)
    n_blocks = -(-(a.window + 1) // a.games) + 1
    arena = Arena(eng, a.slots, n_blocks * a.games, 500)
    loop = SelfPlay(mz, eng, tr, arena, store, a.sims, games_per_iteration=a.games, seed=a.seed)
    t0 = time.perf_counter()
    curve = []
    for it in range(1, a.iterations + 1):
        rewards = loop.play(temperature_scheduler(a.iterations + 1, it, a.temperature))
        losses = [float(loop.train_step().item()) for _ in range(a.train_steps)]
        curve.append(dict(iteration=it, mean_reward=float(rewards.mean()), max_reward=float(rewards.max()),
                          loss=float(np.mean(losses)), seconds=time.perf_counter() - t0,
                          lockstep_moves=loop.lockstep_moves))
        print(json.dumps(curve[-1]), flush=True)


if __name__ == "__main__":
    main()
