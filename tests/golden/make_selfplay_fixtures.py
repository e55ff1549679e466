#!/usr/bin/env python3
"""Golden vectors of the self-play bookkeeping (container-only; never runs on the GPU box).

Imports the reference's own ``Game`` (game.py) and ``ReplayBuffer`` (replay_buffer.py) by path
from /root/reference — torchvision is not installed and is stubbed (Game only touches it for RGB
observations) — and drives them with a scripted environment through the loop of
``play_game`` (self_play.py:62-73): observation -> policy_step -> store_search_statistics.
``self_play`` needs ray and matplotlib (not installed); it is imported with both stubbed only to
tabulate its ``temperature_scheduler``.

Stored per game: the inputs (visit counts, priors, root values, the uniform number
np.random.choice consumed, scripted observations / rewards) and the reference's outputs
(selected actions, child_visits, make_target for every position, make_priority), plus
ReplayBuffer runs (save_game / sample_batch / update_value) with the uniforms they consumed.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_selfplay_fixtures.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tv.transforms = tvt
sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})
sys.path.insert(0, REF)
from game import Game                      # noqa: E402
from monte_carlo_tree_search import Node   # noqa: E402
from replay_buffer import ReplayBuffer     # noqa: E402


def reference_temperature_scheduler():
    """self_play.py imports ray, matplotlib and gymnasium at module level (none installed): stub
    the three names it needs at import time and take its temperature_scheduler."""
    sys.path.insert(0, OUT)
    import make_fixtures  # noqa: F401  (installs the gymnasium stub the reference's model file needs)
    ray = types.ModuleType("ray")
    ray.remote = lambda f: f
    mpl, plt = types.ModuleType("matplotlib"), types.ModuleType("matplotlib.pyplot")
    mpl.pyplot = plt
    sys.modules.update({"ray": ray, "matplotlib": mpl, "matplotlib.pyplot": plt})
    import self_play
    return self_play.temperature_scheduler


class ScriptedEnv:
    """Plays back scripted observations / rewards; `done` after the last scripted move."""
    metadata = {"render_fps": 30}

    def __init__(self, obs0, obs, rewards):
        self.obs0, self.obs, self.rewards, self.t = obs0, obs, rewards, 0

    def reset(self, seed=None):
        self.t = 0
        return self.obs0, {}

    def step(self, action):
        o, r = self.obs[self.t], self.rewards[self.t]
        self.t += 1
        return o, r, self.t == len(self.obs), False, {}

    def close(self):
        pass


def fake_root(visits, priors, value_sum):
    root = Node(0)
    for a, (v, p) in enumerate(zip(visits, priors)):
        root.children[a] = Node(prior=p)
        root.children[a].visit_count = int(v)
    root.visit_count = int(sum(visits)) + 1
    root.value_sum = value_sum             # np.float32, as the reference's backup leaves it
    return root


def play(rs, A, L, limit, temperature, discount, unit_reward, seed_base, from_prior=False):
    """One episode through the reference's Game; returns inputs + the Game."""
    obs0 = rs.standard_normal(4).astype(np.float32)
    obs = rs.standard_normal((L, 4)).astype(np.float32)
    rewards = [1.0] * L if unit_reward else [float(x) for x in rs.integers(-8, 9, L) * 0.25]
    game = Game(gym_env=ScriptedEnv(obs0, obs, rewards), discount=discount, limit_of_game_play=limit,
                observation_dimension=4, action_dimension=A, rgb_observation=False,
                action_map={i: i for i in range(A)})
    visits = np.zeros((L, A), np.int32)
    priors = np.zeros((L, A), np.float64)
    roots = np.zeros(L, np.float32)
    us = np.zeros(L, np.float64)
    counter, feedback = 0, None
    while not game.terminal and counter < game.limit_of_game_play:
        game.observation(iteration=counter, feedback=feedback)
        if rs.random() < 0.15:               # fewer than 3 visits: the prior branch of store_search_statistics
            v = rs.multinomial(int(rs.integers(1, 3)), np.ones(A) / A)
        elif rs.random() < 0.2:              # all equal: the np.random.choice branch at temperature 0
            v = np.full(A, 7)
        else:
            v = rs.multinomial(50, rs.dirichlet(np.ones(A)))
        p = rs.dirichlet(np.ones(A)) * rs.uniform(0.5, 1.5)
        visits[counter], priors[counter] = v, p
        value_sum = np.float32(rs.standard_normal() * 20.0)
        root = fake_root(v, p, value_sum)
        roots[counter] = root.value()
        np.random.seed(seed_base + counter)
        us[counter] = np.random.RandomState(seed_base + counter).random_sample()
        # MCTS.run hands policy_step the visit counts, or the priors when num_simulations == 0
        # (monte_carlo_tree_search.py:218-221)
        if from_prior:
            v = np.zeros(A, np.int64)
            visits[counter] = v
            root = fake_root(v, p, np.float32(0))
            roots[counter] = root.value()
        feedback = game.policy_step(policy=np.asarray(p, np.float64) if from_prior else v.astype(np.float64),
                                    action=np.arange(A), temperature=temperature)
        game.store_search_statistics(root)
        counter += 1
    n = counter
    return game, dict(visits=visits[:n], priors=priors[:n], root_values=roots[:n], u=us[:n],
                      obs=obs[:n], rewards=np.asarray(rewards[:n], np.float64))


def targets_of(game, K, td):
    n = game.game_length
    tvs, trs, tps = np.zeros((n, K), np.float32), np.zeros((n, K), np.float32), np.zeros((n, K, game.action_space_size), np.float32)
    for pos in range(n):
        for k, (v, r, p) in enumerate(game.make_target(pos, K, td)):
            tvs[pos, k], trs[pos, k], tps[pos, k] = np.float32(v), np.float32(r), np.asarray(p, np.float32)
    return tvs, trs, tps


def main():
    rs = np.random.default_rng(20260101)
    out = {}
    cases = []
    cfgs = [  # A, L scripted, limit, temperature, discount, unit rewards
        (2, 30, 500, 0.0, 0.997, True), (2, 12, 500, 1.0, 0.997, True), (2, 40, 25, 0.5, 0.997, True),
        (3, 17, 500, 0.25, 0.95, False), (4, 9, 500, 0.7, 0.997, False), (2, 1, 500, 0.0, 0.8, True),
        (2, 3, 500, 1.0, 0.997, True), (3, 22, 500, 0.0, 0.997, False), (2, 60, 500, 1.0, 0.997, True),
        (4, 33, 500, 2.0, 0.9, True), (2, 14, 500, 0.5, 0.8, False),
    ]
    games = []
    cfgs += [(2, 15, 500, 0.0, 0.997, True, True), (3, 11, 500, 1.0, 0.997, True, True),
             (2, 9, 500, 0.5, 0.997, True, True)]          # 7th field: policy = priors (0 simulations)
    for ci, cfg in enumerate(cfgs):
        A, L, limit, temp, disc, unit = cfg[:6]
        from_prior = len(cfg) > 6
        game, inp = play(rs, A, L, limit, temp, disc, unit, seed_base=1000 * (ci + 1), from_prior=from_prior)
        games.append(game)
        n = game.game_length
        assert n == min(L, limit)
        pre = f"g{ci}_"
        for k, v in inp.items():
            out[pre + k] = v
        out[pre + "cfg"] = np.array([A, limit, temp, disc, float(from_prior)], np.float64)
        out[pre + "actions"] = np.array([int(np.argmax(a)) for a in game.action_history], np.int32)
        out[pre + "child_visits"] = np.array(game.child_visits, np.float64)
        out[pre + "stored_root"] = np.array(game.root_values, np.float32)
        out[pre + "stored_obs"] = np.concatenate([o.numpy() for o in game.observations], 0).astype(np.float32)
        for td in (3, 10):
            tvs, trs, tps = targets_of(game, 5, td)
            out[pre + f"tv_td{td}"], out[pre + f"tr_td{td}"], out[pre + f"tp_td{td}"] = tvs, trs, tps
            prio_pos, prio_game = game.make_priority(td)
            out[pre + f"prio_td{td}"] = np.asarray(prio_pos, np.float64)
            out[pre + f"priogame_td{td}"] = np.float64(prio_game)
        cases.append(ci)
    out["n_games"] = np.int64(len(cases))

    # ---- ReplayBuffer: priority / priority and priority / uniform sampling, window overflow,
    # update_value.  Unit-reward games with A == 2 and length >= 3 only (as CartPole gives).
    ok = lambda g, c: c[0] == 2 and g.game_length >= 3 and c[4] == 0.997 and len(c) == 6  # noqa: E731
    pool = [g for g, c in zip(games, cfgs) if ok(g, c)]
    pool_ids = [i for i, (g, c) in enumerate(zip(games, cfgs)) if ok(g, c)]
    out["rb_pool"] = np.array(pool_ids, np.int64)
    for ri, (pos_mode, window, td) in enumerate([("priority", 3, 3), ("uniform", 10, 10)]):
        rb = ReplayBuffer(window_size=window, batch_size=6, num_unroll=5, td_steps=td,
                          game_sampling="priority", position_sampling=pos_mode)
        order = [0, 1, 2, 3, 4, 0, 2][: len(pool) + 2]
        order = [o % len(pool) for o in order]
        # the reference identifies games by object (buffer.index): give every save its own object
        import copy
        saved = []
        log_pos, log_ratio, log_tv, log_state = [], [], [], []
        for step, o in enumerate(order):
            g = copy.copy(pool[o])
            rb.save_game(g)
            saved.append(o)
            np.random.seed(77 + 13 * step + ri)
            batch, ratio, game_pos = rb.sample_batch()
            log_pos.append(np.asarray(game_pos, np.int64))
            log_ratio.append(np.asarray(ratio, np.float64))
            log_tv.append(np.array([[t[0] for t in b[2]] for b in batch], np.float32))
            # new priorities as Muzero.train returns them: [K][B][1]
            newp = np.abs(np.random.RandomState(5 + step).standard_normal((5, 6, 1)))
            rb.update_value(newp, game_pos)
            log_state.append(np.array(rb.prio_game, np.float64))
        pre = f"rb{ri}_"
        out[pre + "cfg"] = np.array([window, td, 6, 5, pos_mode == "priority"], np.int64)
        out[pre + "order"] = np.array(order, np.int64)
        out[pre + "game_pos"] = np.stack(log_pos)
        out[pre + "ratio"] = np.stack(log_ratio)
        out[pre + "tv"] = np.stack(log_tv)
        out[pre + "buffer_len"] = np.int64(len(rb.buffer))
        out[pre + "big_n"] = np.int64(rb.big_n_of_importance_sampling_ratio)
        for si, s in enumerate(log_state):
            out[pre + f"prio_game_{si}"] = s
        for bi, pp in enumerate(rb.prio_position):
            out[pre + f"final_prio_position_{bi}"] = np.asarray(pp, np.float64)
    ts = reference_temperature_scheduler()
    modes = ["static_temperature", "static_one_temperature", "linear_decrease_temperature",
             "extreme_temperature", "reversal_tanh_temperature"]
    epochs = 41
    out["temp_epochs"] = np.int64(epochs)
    for m in modes:
        out["temp_" + m] = np.array([float(np.asarray(ts(epochs, e, mode=m)).reshape(-1)[0])
                                     for e in range(1, epochs)], np.float64)
    np.savez_compressed(os.path.join(OUT, "selfplay_ref.npz"), **out)
    print("wrote selfplay_ref.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
