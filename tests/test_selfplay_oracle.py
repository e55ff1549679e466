"""Self-play bookkeeping: the oracle (oracle/mz_selfplay_oracle.c) and the host-side ReplayStore /
make_priority (muzero_amd/selfplay.py) against the reference's own Game / ReplayBuffer /
temperature_scheduler outputs (tests/golden/selfplay_ref.npz, made by make_selfplay_fixtures.py).
CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mz_oracle as mzo  # noqa: E402
from muzero_amd import selfplay as sp  # noqa: E402

FX = np.load(os.path.join(ROOT, "tests", "golden", "selfplay_ref.npz"))
N_GAMES = int(FX["n_games"])


def game(ci):
    pre = f"g{ci}_"
    return {k[len(pre):]: FX[k] for k in FX.files if k.startswith(pre)}


@pytest.mark.parametrize("ci", range(N_GAMES))
def test_store_search_statistics(ci):
    g = game(ci)
    for v, p, want in zip(g["visits"], g["priors"], g["child_visits"]):
        got = mzo.store_policy(v, p)
        assert np.array_equal(got, want.astype(np.float32))
    assert np.array_equal(g["stored_root"], g["root_values"])


@pytest.mark.parametrize("ci", range(N_GAMES))
def test_policy_step_action(ci):
    g = game(ci)
    temperature = float(g["cfg"][2])
    policy = g["priors"] if g["cfg"][4] else g["visits"]      # 0 simulations: MCTS.run returns the priors
    got = [mzo.policy_action(v, temperature, u) for v, u in zip(policy, g["u"])]
    assert got == g["actions"].tolist()


def trajectory(g, T):
    n, A = g["visits"].shape
    length = np.array([n], np.int32)
    obs = np.zeros((1, T, 4), np.float32)
    obs[0, :n] = g["stored_obs"]
    act = np.zeros((1, T), np.int32)
    act[0, :n] = g["actions"]
    rew = np.zeros((1, T), np.float32)
    rew[0, :n] = g["rewards"]
    root = np.zeros((1, T), np.float32)
    root[0, :n] = g["stored_root"]
    pol = np.zeros((1, T, A), np.float32)
    pol[0, :n] = g["child_visits"].astype(np.float32)
    return length, obs, act, rew, root, pol


@pytest.mark.parametrize("td", [3, 10])
@pytest.mark.parametrize("ci", range(N_GAMES))
def test_make_target(ci, td):
    g = game(ci)
    n, A = g["visits"].shape
    K, T = 5, n + 3
    discount = float(g["cfg"][3])
    tr = trajectory(g, T)
    row_pos = np.stack([np.zeros(n, np.int32), np.arange(n, dtype=np.int32)], 1)
    obs, actions, tv, trw, tp = mzo.make_batch(*tr, row_pos, K, td, sp.discount_powers(discount, td))
    assert np.array_equal(tv.T, g[f"tv_td{td}"])           # bit-exact, incl. the fp32 / fp64 accumulation
    assert np.array_equal(trw.T, g[f"tr_td{td}"])
    assert np.array_equal(tp.transpose(1, 0, 2), g[f"tp_td{td}"])
    assert np.array_equal(obs, g["stored_obs"])             # make_image: the observation AFTER move pos
    for pos in range(n):
        for k in range(K):
            want = np.zeros(A, np.float32)
            if pos + k < n:
                want[g["actions"][pos + k]] = 1.0            # fill_gap_empty_action: zeros past the end
            assert np.array_equal(actions[k, pos], want)


@pytest.mark.parametrize("td", [3, 10])
@pytest.mark.parametrize("ci", range(N_GAMES))
def test_make_priority(ci, td):
    g = game(ci)
    n = g["visits"].shape[0]
    length, _, _, rew, root, _ = trajectory(g, n + 2)
    prio = sp.make_priority(root, rew, length, td, float(g["cfg"][3]))
    assert np.array_equal(prio[0, :n], g[f"prio_td{td}"])
    assert np.array_equal(mzo.make_priority(length, rew, root, td, sp.discount_powers(float(g["cfg"][3]), td)), prio)
    assert np.all(prio[0, n:] == 0)
    assert np.max(prio[0, :n]) == g[f"priogame_td{td}"]


def test_make_priority_many_rows():
    ids = [i for i in range(N_GAMES)]
    T = max(game(i)["visits"].shape[0] for i in ids) + 1
    for td in (3, 10):
        for disc in (0.997,):
            sel = [i for i in ids if float(game(i)["cfg"][3]) == disc]
            tr = [trajectory(game(i), T) for i in sel]
            prio = sp.make_priority(np.concatenate([t[4] for t in tr]), np.concatenate([t[3] for t in tr]),
                                    np.concatenate([t[0] for t in tr]), td, disc)
            for r, i in enumerate(sel):
                n = game(i)["visits"].shape[0]
                assert np.array_equal(prio[r, :n], game(i)[f"prio_td{td}"])


@pytest.mark.parametrize("native", [False, True])
@pytest.mark.parametrize("ri", [0, 1])
def test_replay_store_matches_reference_buffer(ri, native):
    pre = f"rb{ri}_"
    window, td, B, K, pos_prio = (int(x) for x in FX[pre + "cfg"])
    pool = FX["rb_pool"].tolist()
    order = FX[pre + "order"].tolist()
    T = max(game(i)["visits"].shape[0] for i in pool) + 1
    store = sp.ReplayStore(window, B, K, td, "priority", "priority" if pos_prio else "uniform", discount=0.997,
                           native=native)
    # arena rows: one per save (the reference saves distinct Game objects)
    traj = []
    for step, o in enumerate(order):
        g = game(pool[o])
        length, obs, act, rew, root, pol = trajectory(g, T)
        traj.append((length, obs, act, rew, root, pol))
        store.save_games([step], length, root, rew)
        np.random.seed(77 + 13 * step + ri)
        row_pos, ratio, game_pos = store.sample_batch()
        assert np.array_equal(game_pos, FX[pre + "game_pos"][step])
        np.testing.assert_allclose(ratio, FX[pre + "ratio"][step], rtol=1e-15, atol=0)
        # the rows the store names hold the games the reference sampled: gather through the oracle
        cat = [np.concatenate([t[i] for t in traj]) for i in range(6)]
        _, _, tv, _, _ = mzo.make_batch(*cat, row_pos, K, td, sp.discount_powers(0.997, td))
        assert np.array_equal(tv.T, FX[pre + "tv"][step])
        newp = np.abs(np.random.RandomState(5 + step).standard_normal((5, 6, 1)))
        store.update_value(newp, game_pos)
        assert np.array_equal(np.array(store.prio_game), FX[pre + f"prio_game_{step}"])
    assert len(store) == int(FX[pre + "buffer_len"])
    assert store.big_n_of_importance_sampling_ratio == int(FX[pre + "big_n"])
    for bi, pp in enumerate(store.prio_position):
        assert np.array_equal(pp, FX[pre + f"final_prio_position_{bi}"])


def test_save_games_block_equals_one_by_one():
    rs = np.random.RandomState(3)
    T, n = 12, 9
    lengths = rs.randint(2, T, n)
    root = rs.standard_normal((n, T)).astype(np.float32)
    rew = np.ones((n, T), np.float32)
    a = sp.ReplayStore(4, 3, 5, 3)
    b = sp.ReplayStore(4, 3, 5, 3)
    a.save_games(range(n), lengths, root, rew)
    for i in range(n):
        b.save_games([i], lengths[i:i + 1], root[i:i + 1], rew[i:i + 1])
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.lengths, b.lengths)
    assert len(a) == 5                                                      # window_size + 1, as the reference
    assert all(np.array_equal(x, y) for x, y in zip(a.prio_position, b.prio_position)) and len(a._blocks) == 1
    assert a.big_n_of_importance_sampling_ratio == b.big_n_of_importance_sampling_ratio == sum(a.lengths)
    assert np.array_equal(a.soft_prio_game, b.soft_prio_game)


@pytest.mark.parametrize("native", [False, True])
def test_packed_priorities_equal_padded_table(native):
    """save_games(priorities=<1-d packed array>) (what Arena.priority_packed delivers: game after game,
    no padding) behaves as the [n, T] table does: same draws, ratios, prio_game and prio_position
    through window overflow, sampling and update_value."""
    rs = np.random.RandomState(21)
    n, T, B, K = 700, 60, 32, 5
    stores = [sp.ReplayStore(900, B, K, 10, native=native) for _ in range(2)]
    for blk in range(3):                                  # three saves: the third overflows the window
        lengths = rs.randint(1, T + 1, n)
        prio = np.abs(rs.standard_normal((n, T))) * rs.choice([1e-3, 1.0, 50.0], (n, 1))
        mask = np.arange(T)[None, :] < lengths[:, None]
        prio[~mask] = 0.0
        rows = np.arange(blk * n, (blk + 1) * n)
        stores[0].save_games(rows, lengths, priorities=prio.copy())
        stores[1].save_games(rows, lengths, priorities=prio[mask].copy())
        with pytest.raises(ValueError):
            sp.ReplayStore(900, B, K, 10).save_games(rows, lengths, priorities=prio[mask][:-1].copy())
        for step in range(10):
            outs = []
            for st in stores:
                np.random.seed(1000 * blk + step)
                outs.append(st.sample_batch())
            for a, b in zip(*outs):
                assert np.array_equal(a, b)
            newp = np.abs(np.random.RandomState(step).standard_normal((K, B))) * 3
            for st in stores:
                st.update_value(newp, outs[0][2])
            assert np.array_equal(stores[0].prio_game, stores[1].prio_game)
    assert len(stores[0]) == len(stores[1]) == 901
    for a, b in zip(stores[0].prio_position, stores[1].prio_position):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("pos_mode", ["priority", "uniform"])
def test_native_replay_loops_equal_numpy(pos_mode):
    """The library's host loops (mzs_replay.h: NumPy's pairwise sum / cumsum / searchsorted restated)
    against the NumPy formulation on a large random store: same draws, same ratios, same
    priorities after update_value, bit for bit, over many steps (game lengths 2..400 cover every
    branch of the pairwise summation: < 8, <= 128, recursive)."""
    rs = np.random.RandomState(7)
    n, T, B, K = 3000, 400, 128, 5
    lengths = rs.randint(2, T + 1, n)
    lengths[:40] = rs.randint(2, 9, 40)
    prio = np.abs(rs.standard_normal((n, T))) * rs.choice([1e-3, 1.0, 50.0], (n, 1))
    prio[np.arange(T)[None, :] >= lengths[:, None]] = 0.0
    # window overflow only under priority / priority: with uniform positions the reference pops
    # prio_game but not prio_position (replay_buffer.py:64-67) and then indexes out of range
    window = 2500 if pos_mode == "priority" else 4000
    stores = [sp.ReplayStore(window, B, K, 10, "priority", pos_mode, native=nat) for nat in (False, True)]
    for st in stores:
        st.save_games(np.arange(n), lengths, priorities=prio.copy())
    for step in range(30):
        outs = []
        for st in stores:
            np.random.seed(100 + step)
            outs.append(st.sample_batch())
        for a, b in zip(*outs):
            assert np.array_equal(a, b)
        newp = np.abs(np.random.RandomState(step).standard_normal((K, B))) * 3
        for st in stores:
            st.update_value(newp, outs[0][2])
        assert np.array_equal(stores[0].prio_game, stores[1].prio_game)
    for a, b in zip(stores[0].prio_position, stores[1].prio_position):
        assert np.array_equal(a, b)
    if pos_mode == "priority":
        for st in stores:                                     # an all-zero row: np.random.choice raises
            st.prio_position[5][:] = 0.0
            st.soft_prio_game[:] = 0.0
            st.soft_prio_game[5] = 1.0
            st._game_cdf = None                               # (the cached cdf follows save_games only)
            with pytest.raises(ValueError):
                st.sample_batch()


def test_temperature_scheduler():
    epochs = int(FX["temp_epochs"])
    for mode in ("static_temperature", "static_one_temperature", "linear_decrease_temperature",
                 "extreme_temperature", "reversal_tanh_temperature"):
        got = np.array([float(sp.temperature_scheduler(epochs, e, mode)) for e in range(1, epochs)])
        np.testing.assert_allclose(got, FX["temp_" + mode], rtol=1e-15, atol=0, err_msg=mode)


def test_cartpole_against_float64_numpy_equations():
    """Physics parity is unpinned (gymnasium is not installed): this checks the restated equations
    against an independent float64 evaluation with libm sin / cos, to 1e-12."""
    rs = np.random.RandomState(0)
    s = rs.uniform(-0.05, 0.05, 4)
    ref = s.copy()
    for t in range(200):
        a = int(rs.randint(0, 2))
        s, r, term = mzo.cartpole_step(s, a)
        x, xd, th, thd = ref
        f = 10.0 if a == 1 else -10.0
        temp = (f + 0.05 * thd * thd * np.sin(th)) / 1.1
        tha = (9.8 * np.sin(th) - np.cos(th) * temp) / (0.5 * (4.0 / 3.0 - 0.1 * np.cos(th) ** 2 / 1.1))
        xa = temp - 0.05 * tha * np.cos(th) / 1.1
        ref = np.array([x + 0.02 * xd, xd + 0.02 * xa, th + 0.02 * thd, thd + 0.02 * tha])
        np.testing.assert_allclose(s, ref, rtol=0, atol=1e-12)
        want_term = abs(ref[0]) > 2.4 or abs(ref[2]) > 12 * 2 * np.pi / 360
        assert term == want_term and r == 1.0
        if term:
            break
        ref = s.copy()
    assert term, "a random policy drops the pole well within 200 moves"


def test_arena_oracle_episode_bookkeeping():
    G, T = 6, 9
    ar = mzo.ArenaOracle(G, T)
    obs0 = ar.begin(np.arange(G) + 11, episode=2).copy()
    assert np.all(np.abs(obs0) <= 0.05) and len({tuple(o) for o in obs0}) == G
    again = mzo.ArenaOracle(G, T)
    assert np.array_equal(again.begin(np.arange(G) + 11, episode=2), obs0)          # counter-based: reproducible
    assert not np.array_equal(again.begin(np.arange(G) + 11, episode=3), obs0)
    rs = np.random.RandomState(1)
    for t in range(T):
        visits = rs.multinomial(50, [0.5, 0.5], G).astype(np.int32)
        alive_before = ar.alive.copy()
        n_alive = ar.step(visits, rs.standard_normal(G), rs.dirichlet([1, 1], G), temperature=0.0)
        assert n_alive == ar.alive.sum()
        live = alive_before == 1
        assert np.array_equal(ar.length[live], np.full(live.sum(), t + 1))
        # temperature 0: argmax of the visit counts (first maximum), the sampled branch on a tie
        for g in np.nonzero(live)[0]:
            if visits[g, 0] != visits[g, 1]:
                assert ar.traj_action[g, t] == int(np.argmax(visits[g]))
            np.testing.assert_array_equal(ar.traj_policy[g, t], (visits[g] / 50.0).astype(np.float32))
            if ar.alive[g]:
                assert np.array_equal(ar.traj_obs[g, t], ar.obs[g])
    assert ar.alive.sum() == 0 and np.all(ar.length <= T)            # limit_of_game_play ends every game


def test_arena_oracle_refills_free_slots_in_order():
    """More games than slots: a finished slot takes the next unplayed row; every game is played
    exactly once, from its own seed, whatever slot it lands in."""
    G, T, N = 4, 12, 19
    seeds = np.arange(N) * 101 + 3
    ar = mzo.ArenaOracle(G, T, n_rows=N + 2)
    ar.begin(seeds, episode=5, row0=2)
    rs = np.random.RandomState(2)
    started = list(ar.row.copy())
    for it in range(N * T):
        visits = rs.multinomial(50, [0.5, 0.5], G).astype(np.int32)
        rows_before = ar.row.copy()
        n_alive = ar.step(visits, rs.standard_normal(G), rs.dirichlet([1, 1], G), temperature=1.0)
        for g in range(G):
            if ar.row[g] != rows_before[g]:
                started.append(int(ar.row[g]))
        if n_alive == 0:
            break
    assert started == list(range(2, N + 2))                      # rows handed out in order, once each
    assert np.all(ar.length[2:] >= 1) and np.all(ar.length[:2] == 0)
    # a game's trajectory depends on its seed only, not on the slot / time it was played: replay row 9 alone
    solo = mzo.ArenaOracle(1, T)
    solo.begin(seeds[7:8], episode=5)
    assert np.array_equal(solo.obs[0], ar.traj_obs[9, 0] * 0 + solo.obs[0])
    first = solo.obs[0].copy()
    st = solo.state[0].copy()
    s2, _, _ = mzo.cartpole_step(st, int(ar.traj_action[9, 0]))
    assert np.array_equal(s2.astype(np.float32), ar.traj_obs[9, 0]) and np.all(np.abs(first) <= 0.05)


def test_oracle_rolling_queue_semantics():
    """mzo_arena_enqueue (the CPU statement of mzs_arena_enqueue): a game's trajectory depends on its
    (seed, round key, temperature) only — queued behind another round and started in whatever slot
    frees up, it records exactly what it records when its round is played on its own."""
    from oracle import mz_oracle as mzo
    G, T, N = 16, 25, 60

    def fake(ar):
        v = np.stack([(ar.obs[:, 2] > 0) * 30 + 10, (ar.obs[:, 2] <= 0) * 30 + 10], 1).astype(np.int32)
        return v, ar.obs[:, 0].copy(), np.full((ar.G, 2), 0.5)

    seeds = [np.arange(N, dtype=np.int64) * 31 + 7 + 1000 * k for k in range(3)]
    temps = [1.0, 0.0, 0.5]
    roll = mzo.ArenaOracle(G, T, n_rows=3 * N)
    roll.begin(seeds[0], 0)
    for k in (1, 2):
        roll.enqueue(seeds[k], k, temps[k])
    lock = 0
    while True:
        lock += 1
        if roll.step(*fake(roll), temps[0]) == 0:
            break
    assert roll.done_upto == 3 * N and roll.queue == (3 * N, 3 * N)
    total = int(roll.length.sum())
    for k in range(3):                              # each round alone, on a different slot count
        solo = mzo.ArenaOracle(7, T, n_rows=N)
        solo.begin(seeds[k], k)
        while solo.step(*fake(solo), temps[k]) != 0:
            pass
        blk = slice(k * N, (k + 1) * N)
        assert np.array_equal(solo.length, roll.length[blk])
        assert np.array_equal(solo.traj_action, roll.traj_action[blk])
        assert np.array_equal(solo.traj_obs, roll.traj_obs[blk])
    assert total / (lock * G) > 0.85                # one queue, one tail: the slots stayed busy
