#!/usr/bin/env python3
"""Throughput of the batched self-play loop on one MI355X: G games x S simulations per move,
fill_streams -> search -> arena step per move, then sample / gather / train steps.
Prints one JSON line.  Usage: python tools/bench_selfplay.py [--games 4096] [--sims 50] [--gpus N]

--gpus N > 1 starts N ranks itself (muzero_amd.shard.launch_ranks: one process per GPU, RCCL): every rank
owns an arena and a replay store and plays its own games (seeds offset by rank, no collective in the play
path, as self_play.py:212-236 hands whole games to Ray workers); the training steps are data-parallel —
one all_reduce of the gradient bucket per step (muzero_amd.train) — so the replicas' weights stay
identical, which the run asserts at the end.  Rank 0 prints the whole job's figures: game moves and
simulations summed over the ranks, divided by the slowest rank's time."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--max-moves", type=int, default=500)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--games-per-round", type=int, default=0, help="default 8 x --games")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--train-steps", type=int, default=50)
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16", "bf16x3"),
                    help="precision of the model half of the search (bf16 = the fast contract, bf16x3 = hi + lo pairs)")
    ap.add_argument("--rolling", action="store_true",
                    help="queue the next round's games while the current one drains (SelfPlay(rolling=True))")
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); > 1 starts them itself")
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    a = ap.parse_args()

    from muzero_amd import shard
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(shard.launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    if shard.rank_world()[1] != a.gpus:
        sys.exit(f"bench_selfplay.py: --gpus {a.gpus} but WORLD_SIZE={shard.rank_world()[1]}")
    if shard.visible_gpus() < a.gpus:
        sys.exit(f"bench_selfplay.py: --gpus {a.gpus} but only {shard.visible_gpus()} GPU(s) are visible")
    import torch.distributed as dist
    from muzero_amd.engine import SearchEngine
    from muzero_amd.selfplay import Arena, ReplayStore, SelfPlay, play_round
    from muzero_amd.train import Trainer
    rank, world = shard.init(a.backend)
    local = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    torch.cuda.set_device(local)

    G, S, T = a.games, a.sims, a.max_moves
    dims = (4, 2, 31, 64, 0, 0, {"f32": 0, "bf16": 1, "bf16x3": 2}[a.dtype])
    eng = SearchEngine(dims, max_games=G, max_simulations=S, discount=0.997, device=local)
    n = eng.weight_count()
    w0 = (np.random.RandomState(0).standard_normal(n) * 0.1).astype(np.float32)
    eng.set_weights(w0)
    K, td = 5, 10
    tr = Trainer(eng, K, a.batch, "adam", 1e-3)
    N = a.games_per_round or 8 * G

    def run_once():
        eng.set_weights(w0)        # both passes play the same games (the optimizer state carries over, harmless)
        store = ReplayStore(window_size=N, batch_size=a.batch, num_unroll=K, td_steps=td, discount=0.997)
        ar = Arena(eng, G, (4 if a.rolling else 3) * N, T)
        loop = SelfPlay(None, eng, tr, ar, store, S, games_per_iteration=N, seed=1, rolling=a.rolling, rank=rank)
        np.random.seed(rank)
        out = []
        for r in range(a.rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if r == a.rounds - 1 and not a.rolling and world == 1:   # last round: time the device loop alone
                seeds = (np.arange(N, dtype=np.int64) * 2654435761 + 99) & 0x7FFFFFFF
                mv = play_round(eng, ar, seeds, 1000 + r, S, 1.0, loop.block * N)
                torch.cuda.synchronize()
                tl = time.perf_counter() - t0
                n_moves = int(ar.arrays["length"][loop.block * N:(loop.block + 1) * N].sum().item())
                out.append(dict(loop_only=True, loop_s=tl, loop_moves=mv, ms_per_lockstep_move=tl * 1e3 / mv,
                                game_moves=n_moves, sims_per_s=n_moves * S / tl))
                t0 = time.perf_counter()
            rewards = loop.play(1.0, 1.0 if (a.rolling and r < a.rounds - 1) else None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            moves = int(rewards.sum())
            longest = loop.lockstep_moves
            for _ in range(a.train_steps):
                loss = loop.train_step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out.append(dict(round=r, play_s=t1 - t0, game_moves=moves, lockstep_moves=longest,
                            ms_per_lockstep_move=(t1 - t0) * 1e3 / longest,
                            train_ms_per_step=(t2 - t1) * 1e3 / a.train_steps,
                            mean_reward=float(rewards.mean()), loss=float(loss.item())))
        ar.close()
        return out

    run_once()            # warm-up pass: module load, kernel attribute calls, allocator growth
    if world > 1:
        dist.barrier()
    out = run_once()
    # Totals over ALL rounds of the measured pass.  With rolling rounds a round's lock-step moves also
    # advance the next round's early games, so only the totals mean something: game moves recorded /
    # (lock-step moves x slots); the last round's tail drains unrefilled and is part of the figure.
    rr = [o for o in out if "round" in o]
    tot_moves = sum(o["game_moves"] for o in rr)
    tot_lock = sum(o["lockstep_moves"] for o in rr)
    tot_play = sum(o["play_s"] for o in rr)
    tot_train = sum(o["train_ms_per_step"] for o in rr) * a.train_steps * 1e-3
    job_moves, job_lock = tot_moves, tot_lock
    if world > 1:
        # whole-job figures: work summed over the ranks, time of the slowest rank (the ranks meet in every
        # training step's all-reduce, so their rounds stay in step)
        t = torch.tensor([tot_moves, tot_lock], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        job_moves, job_lock = int(t[0].item()), int(t[1].item())
        tot_play = shard.max_over_ranks(tot_play, eng.device)
        tot_train = shard.max_over_ranks(tot_train, eng.device)
        shard.assert_replicas_identical(tr.weights(), eng.device)      # the data-parallel steps kept them equal
    if rank == 0:
        print(json.dumps(dict(n_gpus=world, games_per_round=N, games_per_round_whole_job=N * world, slots=G, sims=S,
                              max_moves=T, batch=a.batch, global_batch=a.batch * world, rolling=a.rolling,
                              dtype=a.dtype, kernel=eng.search_kernel_name(S),
                              rounds_timed=len(rr), slot_utilisation=job_moves / (job_lock * G),
                              useful_sims_per_s=job_moves * S / tot_play,
                              search_sims_per_s=job_lock * G * S / tot_play,
                              ms_per_lockstep_move=tot_play * 1e3 / tot_lock,
                              train_ms_per_step=tot_train * 1e3 / (a.train_steps * len(rr)), rounds=out)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
