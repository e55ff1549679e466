#!/usr/bin/env python3
"""Randomised parity sweep of the self-play arena on one MI355X: for a time budget, draw a slot count, a
round size, an episode limit, a number of rolling rounds (each with its own temperature and key), feed the
arena (mzs_arena_begin / enqueue / step through the C-ABI) and the CPU oracle's arena the same synthetic
search results move by move, and compare: after every move the status entry (queue state, live slots) and the
observations, after every round all trajectory arrays of its rows, then Game.make_priority (padded and packed)
and a random batch of make_target positions (mzs_arena_make_batch) — integer, fp32 and fp64 work, bit for bit.
Test infrastructure like tests/.  Prints one JSON line per case and a summary; exits non-zero on a mismatch.
Usage: python tests/sweeps/arena_sweep.py [--seconds 60] [--seed 0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def fake_results(rs, G, A, dev):
    """Search results of every kind the step has to deal with: ordinary visit counts, almost none, exact ties."""
    import torch
    visits = rs.multinomial(50, rs.dirichlet(np.ones(A)), G).astype(np.int32)
    few = rs.random_sample(G) < 0.1
    visits[few] = rs.multinomial(2, np.ones(A) / A, int(few.sum()))
    tie = rs.random_sample(G) < 0.1
    visits[tie] = 25
    rv = rs.standard_normal(G).astype(np.float32) * 3
    pr = rs.dirichlet(np.ones(A), G) * rs.uniform(0.5, 1.5, (G, 1))
    res = dict(visits=torch.as_tensor(visits).to(dev), root_value=torch.as_tensor(rv).to(dev),
               root_prior=torch.as_tensor(pr).to(dev))
    return (visits, rv, pr), res


def one_case(rs, engines):
    import torch
    from muzero_amd.engine import SearchEngine
    from muzero_amd.selfplay import Arena, discount_powers
    from oracle import mz_oracle as mzo
    host = lambda t: t.cpu().numpy()                                            # noqa: E731
    G = int(rs.choice([1, 2, 7, 64, 65, 300, 1000, 2500]))
    T = int(rs.choice([1, 2, 5, 12, 30, 60]))
    N = max(1, int(G * rs.choice([0.5, 1.0, 1.0, 2.3, 4.0])) + int(rs.randint(0, 3)))
    rounds = int(rs.choice([1, 2, 4]))
    R = 3 * N
    temps = [float(rs.choice([0.0, 0.25, 0.5, 0.7, 1.0])) for _ in range(rounds)]
    td, K = int(rs.choice([1, 3, 10, 50])), int(rs.choice([1, 5, 10]))
    rec = dict(slots=G, limit=T, games_per_round=N, rounds=rounds, temperatures=temps, td_steps=td, unroll=K)
    eng = engines.get(G)
    if eng is None:
        eng = engines[G] = SearchEngine((4, 2, 31, 64, 0), max_games=G, max_simulations=0)
    ar = Arena(eng, G, R, T)
    orc = mzo.ArenaOracle(G, T, n_rows=R)
    seeds = lambda k: (rs.randint(0, 2 ** 31 - 1) + np.arange(N, dtype=np.int64) * 104729 + 13 * k)   # noqa: E731
    s0 = seeds(0)
    obs = ar.begin(0, s0, episode=0)
    orc.begin(s0, 0, row0=0)
    moves = 0

    def fail(what):
        rec.update(ok=False, what=what, move=moves)
        ar.close()
        return rec

    for k in range(rounds):
        if k + 1 < rounds:                       # keep one round ahead (rolling rounds)
            sk = seeds(k + 1)
            ar.enqueue(sk, k + 1, temps[k + 1])
            orc.enqueue(sk, k + 1, temps[k + 1])
        while orc.done_upto < (k + 1) * N:
            (v, rv, pr), res = fake_results(rs, G, 2, eng.device)
            ar.step(temps[0], res, num_simulations=50)
            n_alive = orc.step(v, rv, pr, temps[0])
            torch.cuda.synchronize()
            st = ar.status(moves)
            moves += 1
            if (st["next"], st["end"], st["done_upto"], st["alive"]) != (*orc.queue, orc.done_upto, n_alive):
                return fail("status entry")
            if not np.array_equal(host(obs), orc.obs):
                return fail("observations")
            live = orc.alive == 1
            if not (np.array_equal(host(ar.arrays["alive"]), orc.alive)
                    and np.array_equal(host(ar.arrays["slot_row"])[live], orc.row[live])
                    and np.array_equal(host(ar.arrays["stream_seed"])[live], orc.stream_seed[live])):
                return fail("slot bookkeeping")
            if moves > 4 * (rounds * N // G + 2) * T + 16:
                return fail("did not finish")
        got = {name: host(t) for name, t in ar.arrays.items()}
        blk = slice((k % 3) * N, (k % 3 + 1) * N)
        for name, w in (("length", orc.length), ("obs", orc.traj_obs), ("action", orc.traj_action),
                        ("reward", orc.traj_reward), ("root_value", orc.traj_root), ("policy", orc.traj_policy)):
            if not np.array_equal(got[name][blk], w[blk]):
                return fail(f"trajectory array {name} of round {k}")
        # Game.make_priority of the round's rows, padded and packed; make_target on a random batch of them
        row0 = (k % 3) * N
        lengths = got["length"][blk]
        dp = discount_powers(0.997, td)
        wantp = mzo.make_priority(got["length"], got["reward"], got["root_value"], td, dp)[blk]
        longest = int(lengths.max())
        if not np.array_equal(host(ar.priority(row0, N, td, 0.997, longest)), wantp[:, :longest]):
            return fail("priority")
        packed = host(ar.priority_packed(row0, ar.arrays["length"][blk], int(lengths.sum()), longest, td, 0.997))
        mask = np.arange(wantp.shape[1])[None, :] < lengths[:, None]
        if not np.array_equal(packed, wantp[mask]):
            return fail("packed priority")
        B = int(rs.choice([1, 33, 400]))
        rows = row0 + rs.randint(0, N, B)
        pos = np.array([rs.randint(0, got["length"][r]) for r in rows])
        pos[: B // 4] = got["length"][rows[: B // 4]] - 1          # last positions: absorbing targets
        row_pos = np.stack([rows, pos], 1).astype(np.int32)
        out = ar.make_batch(row_pos, K, td, 0.997)
        torch.cuda.synchronize()
        want = mzo.make_batch(got["length"], got["obs"], got["action"], got["reward"], got["root_value"],
                              got["policy"], row_pos, K, td, dp)
        for name, g, w in zip(("obs", "actions", "target_value", "target_reward", "target_policy"), out, want):
            if not np.array_equal(host(g), w):
                return fail(f"make_batch {name}")
    ar.close()
    rec.update(ok=True, moves=moves)
    return rec


def run(seconds, seed=0, log=print):
    t_end = time.time() + seconds
    cases = moves = games = 0
    engines = {}
    while time.time() < t_end:
        rs = np.random.RandomState((seed * 1000003 + cases) % (2 ** 31))
        rec = one_case(rs, engines)
        rec["case"] = cases
        log(json.dumps(rec))
        if not rec["ok"]:
            return dict(summary="MISMATCH", seed=seed, **rec)
        cases += 1
        moves += rec["moves"]
        games += rec["rounds"] * rec["games_per_round"]
    for e in engines.values():
        e.close()
    return dict(summary="all equal", seed=seed, cases=cases, lockstep_moves=moves, games=games)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    out = run(a.seconds, a.seed, log=lambda s: print(s, flush=True))
    print(json.dumps(out), flush=True)
    sys.exit(0 if out["summary"] == "all equal" else 1)


if __name__ == "__main__":
    main()
