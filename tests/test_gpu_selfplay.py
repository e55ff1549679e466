"""GPU parity of the batched self-play path (mzs_arena_* through the C-ABI) against the CPU oracle
(oracle/mz_selfplay_oracle.c, itself pinned to the reference's Game / ReplayBuffer by
tests/test_selfplay_oracle.py).  Integer / index work and the fp32 / fp64 targets: bit-exact."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

TAG = "tag123"     # CartPole checkpoint shape: obs 4, 2 actions


def make_engine(G, S, discount=0.997):
    from muzero_amd.engine import SearchEngine
    w = load_golden(f"weights_{TAG}.npz")
    dims = tuple(int(w[k]) for k in ("obs_dim", "A", "S_dim", "H", "L"))
    eng = SearchEngine(dims, max_games=G, max_simulations=S, discount=discount)
    eng.set_weights(w["blob"])
    return eng, dims, w["blob"]


def host(t):
    return t.cpu().numpy()


def arena_host(ar, rows=None):
    import torch
    torch.cuda.synchronize()
    out = {k: host(v) for k, v in ar.arrays.items()}
    return out


def fake_results(rs, G, A, dev, small=0.1):
    import torch
    visits = rs.multinomial(50, rs.dirichlet(np.ones(A)), G).astype(np.int32)
    few = rs.random_sample(G) < small
    visits[few] = rs.multinomial(2, np.ones(A) / A, int(few.sum()))
    tie = rs.random_sample(G) < small
    visits[tie] = 25
    rv = rs.standard_normal(G).astype(np.float32) * 3
    pr = rs.dirichlet(np.ones(A), G) * rs.uniform(0.5, 1.5, (G, 1))
    res = dict(visits=torch.as_tensor(visits).to(dev), root_value=torch.as_tensor(rv).to(dev),
               root_prior=torch.as_tensor(pr).to(dev))
    return (visits, rv, pr), res


@pytest.mark.parametrize("temperature", [0.0, 0.25, 0.5, 1.0, 0.7])
def test_arena_step_vs_oracle(temperature):
    import torch
    from muzero_amd.selfplay import Arena
    from oracle import mz_oracle as mzo
    G, T = 300, 24
    N = G if temperature != 1.0 else 3 * G + 17          # more games than slots: refill in slot order
    R = N + 5
    eng, dims, _ = make_engine(G, 0)
    ar = Arena(eng, G, R, T)
    orc = mzo.ArenaOracle(G, T, n_rows=R)
    seeds = np.arange(N, dtype=np.int64) * 7919 + 5
    obs = ar.begin(5, seeds, episode=3)
    want = orc.begin(seeds, 3, row0=5)
    torch.cuda.synchronize()
    assert np.array_equal(host(obs), want)
    rs = np.random.RandomState(int(temperature * 100))
    for t in range(N * T):
        (v, rv, pr), res = fake_results(rs, G, 2, eng.device)
        ar.step(temperature, res, num_simulations=50)
        n_alive = orc.step(v, rv, pr, temperature)
        torch.cuda.synchronize()
        assert int(ar.alive_count.item()) == n_alive
        st = ar.status(t)                                       # the move's entry in pinned host memory
        assert (st["move"], st["alive"], st["max_consumed"]) == (t, n_alive, 0)
        assert np.array_equal(host(obs), orc.obs)
        assert np.array_equal(host(ar.arrays["alive"]), orc.alive)
        live = orc.alive == 1
        assert np.array_equal(host(ar.arrays["slot_row"])[live], orc.row[live])
        assert np.array_equal(host(ar.arrays["stream_seed"])[live], orc.stream_seed[live])
        if n_alive == 0:
            break
    got = arena_host(ar)
    assert np.array_equal(got["length"], orc.length)
    assert np.array_equal(got["state"], orc.state)              # fp64 physics, bit for bit
    for k, w in (("obs", orc.traj_obs), ("action", orc.traj_action), ("reward", orc.traj_reward),
                 ("root_value", orc.traj_root), ("policy", orc.traj_policy)):
        assert np.array_equal(got[k], w), k
    assert n_alive == 0 and np.all(orc.length[5:] >= 1) and np.all(orc.length[:5] == 0)
    ar.close(), eng.close()


def test_arena_rolling_queue_vs_oracle():
    """mzs_arena_enqueue: three rounds queued one ahead of each other, rows wrapping round the ring,
    each round with its own temperature and random-number key; slots freed by round k's last games
    start round k + 1's — slot rows, stream seeds, queue state (next, end, done_upto) and every
    trajectory array equal the oracle's after every move."""
    import torch
    from muzero_amd.selfplay import Arena
    from oracle import mz_oracle as mzo
    G, T, N = 64, 20, 150
    R = 3 * N                                   # ring: round 3 reuses round 0's rows
    eng, dims, _ = make_engine(G, 0)
    ar = Arena(eng, G, R, T)
    orc = mzo.ArenaOracle(G, T, n_rows=R)
    seeds = lambda k: np.arange(N, dtype=np.int64) * 104729 + 13 * k + 1   # noqa: E731
    temps = [1.0, 0.5, 0.0, 0.25]
    obs = ar.begin(0, seeds(0), episode=0)
    orc.begin(seeds(0), 0, row0=0)
    rs = np.random.RandomState(11)
    moves = full = 0
    for k in range(4):
        if k < 3:                               # keep one round ahead
            ar.enqueue(seeds(k + 1), k + 1, temps[k + 1])
            orc.enqueue(seeds(k + 1), k + 1, temps[k + 1])
        while orc.done_upto < (k + 1) * N:
            (v, rv, pr), res = fake_results(rs, G, 2, eng.device)
            ar.step(temps[0], res, num_simulations=50)          # rows of begin(): NaN -> this argument
            n_alive = orc.step(v, rv, pr, temps[0])
            torch.cuda.synchronize()
            moves += 1
            full += n_alive == G
            assert int(ar.alive_count.item()) == n_alive
            assert ar.queue_state() == (*orc.queue, orc.done_upto)
            st = ar.status(moves - 1)
            assert (st["next"], st["end"], st["done_upto"], st["alive"]) == (*orc.queue, orc.done_upto, n_alive)
            live = orc.alive == 1
            assert np.array_equal(host(ar.arrays["alive"]), orc.alive)
            assert np.array_equal(host(ar.arrays["slot_row"])[live], orc.row[live])
            assert np.array_equal(host(ar.arrays["stream_seed"])[live], orc.stream_seed[live])
            assert np.array_equal(host(obs), orc.obs)
        got = arena_host(ar)
        blk = slice((k % 3) * N, (k % 3 + 1) * N)               # round k's rows are complete now
        assert np.all(orc.length[blk] >= 1)
        for name, w in (("length", orc.length), ("obs", orc.traj_obs), ("action", orc.traj_action),
                        ("root_value", orc.traj_root), ("policy", orc.traj_policy)):
            assert np.array_equal(got[name][blk], w[blk]), (k, name)
    assert orc.done_upto == 4 * N and n_alive == 0
    # the point of the queue: except for the last round's tail the slots were all busy
    assert full / moves > 0.7
    ar.close(), eng.close()


def test_large_arena_vs_oracle():
    """More slots than the refill block has threads x 8: several slots per thread in the scan; same
    oracle, same status entries."""
    import torch
    from muzero_amd.selfplay import Arena
    from oracle import mz_oracle as mzo
    G, T = 8300, 5
    N = G + 700
    eng, dims, _ = make_engine(G, 0)
    ar = Arena(eng, G, N, T)
    orc = mzo.ArenaOracle(G, T, n_rows=N)
    seeds = np.arange(N, dtype=np.int64) * 31 + 2
    obs = ar.begin(0, seeds, episode=1)
    orc.begin(seeds, 1, row0=0)
    rs = np.random.RandomState(4)
    for t in range(3 * T):
        (v, rv, pr), res = fake_results(rs, G, 2, eng.device)
        ar.step(1.0, res, num_simulations=50)
        n_alive = orc.step(v, rv, pr, 1.0)
        torch.cuda.synchronize()
        st = ar.status(t)
        assert (st["next"], st["end"], st["done_upto"], st["alive"]) == (*orc.queue, orc.done_upto, n_alive)
        assert np.array_equal(host(obs), orc.obs)
        if n_alive == 0:
            break
    got = arena_host(ar)
    assert n_alive == 0 and np.array_equal(got["length"], orc.length)
    for k, w in (("obs", orc.traj_obs), ("action", orc.traj_action), ("policy", orc.traj_policy)):
        assert np.array_equal(got[k], w), k
    with pytest.raises(Exception):
        ar.status(t + 1)                                        # a move that has not been played
    ar.close(), eng.close()


def test_randomised_arena_sweep_equals_oracle():
    """Eight seconds of tests/sweeps/arena_sweep.py: random slot counts, episode limits, ragged round sizes, rolling
    rounds on a ring of rows, temperatures — status entries and observations after every move, trajectories,
    priorities and make_target batches after every round == the oracle's (profiles/r2d_arena_sweep.md)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweeps"))
    import arena_sweep
    out = arena_sweep.run(8.0, seed=5, log=lambda s: None)
    assert out["summary"] == "all equal", out
    assert out["cases"] >= 20


def test_priority_packed_equals_padded():
    """mzs_arena_priority_packed == mzs_arena_priority without the zeros, bit for bit."""
    import torch
    from muzero_amd.selfplay import Arena
    G, T, N = 96, 30, 250
    eng, dims, _ = make_engine(G, 0)
    ar = Arena(eng, G, N + 10, T)
    ar.begin(10, np.arange(N, dtype=np.int64) * 17 + 3, episode=2)
    rs = np.random.RandomState(9)
    for t in range(N * T):
        _, res = fake_results(rs, G, 2, eng.device)
        ar.step(1.0, res, num_simulations=50)
        if t % 8 == 7 and int(ar.alive_count.item()) == 0:
            break
    lengths_d = ar.arrays["length"][10:10 + N]
    lengths = host(lengths_d)
    for td in (3, 10):
        padded = host(ar.priority(10, N, td, 0.997, int(lengths.max())))
        packed = host(ar.priority_packed(10, lengths_d, int(lengths.sum()), int(lengths.max()), td, 0.997))
        mask = np.arange(padded.shape[1])[None, :] < lengths[:, None]
        assert packed.shape == (int(lengths.sum()),) and np.array_equal(packed, padded[mask])
        assert np.all(padded[~mask] == 0)
    ar.close(), eng.close()


def test_rolling_learning_cycle_keeps_slots_busy():
    """SelfPlay(rolling=True) end to end with the real search: rewards / losses finite, the store
    follows the window rule, and the slots' utilisation over the run beats the synchronous loop's."""
    from muzero_amd.selfplay import Arena, ReplayStore, SelfPlay
    from muzero_amd.train import Trainer
    util = {}
    for rolling in (False, True):
        G, T, S, B, K, td, N = 128, 48, 10, 64, 5, 10, 512
        eng, dims, blob = make_engine(G, S)
        tr = Trainer(eng, K, B, "adam", 1e-3)
        store = ReplayStore(window_size=2 * N, batch_size=B, num_unroll=K, td_steps=td, discount=0.997)
        ar = Arena(eng, G, (5 if rolling else 4) * N, T)
        np.random.seed(0)
        loop = SelfPlay(None, eng, tr, ar, store, S, games_per_iteration=N, seed=42, rolling=rolling)
        lock = moves = 0
        rewards = []
        for ep in range(5):
            r = loop.play(1.0, 1.0 if ep < 4 else None)
            rewards.append(float(r.mean()))
            lock += loop.lockstep_moves
            moves += int(r.sum())
            for _ in range(2):
                assert np.isfinite(float(loop.train_step().item()))
        util[rolling] = moves / (lock * G)
        assert all(1 <= x <= T for x in rewards) and len(store) == 2 * N + 1
        length = host(ar.arrays["length"])
        assert all(length[r] == n for r, n in zip(store.rows, store.lengths))
        ar.close(), eng.close()
    print(f"\nslot utilisation: synchronous {util[False]:.2f}, rolling {util[True]:.2f}")
    assert util[True] > util[False] + 0.05 and util[True] > 0.8


def test_rolling_learning_cycle_called_twice():
    """A second learning_cycle (or play) after a rolling run whose last iteration queued nothing ahead: the
    round is handed to the arena then, every saved row is a finished game of THAT round, and the store's
    lengths are the arena's."""
    from muzero_amd.selfplay import Arena, ReplayStore, SelfPlay
    from muzero_amd.train import Trainer
    G, T, S, B, K, td, N = 64, 40, 8, 32, 3, 5, 160
    eng, dims, blob = make_engine(G, S)
    tr = Trainer(eng, K, B, "adam", 1e-3)
    store = ReplayStore(window_size=3 * N, batch_size=B, num_unroll=K, td_steps=td, discount=0.997)
    ar = Arena(eng, G, 6 * N, T)
    np.random.seed(1)
    loop = SelfPlay(None, eng, tr, ar, store, S, games_per_iteration=N, seed=7, rolling=True)
    r1, _ = loop.learning_cycle(2, 1)
    assert loop.episode == 2 and loop.queued == 2
    r2, _ = loop.learning_cycle(2, 1)                       # round 2 was never queued by the first call
    r3 = loop.play(1.0)                                     # and a bare play() after that
    assert loop.episode == 5 and loop.queued == 5
    assert all(1 <= x <= T for x in r1 + r2) and np.all(r3 >= 1)
    length = host(ar.arrays["length"])
    assert np.all(length[:5 * N] >= 1) and np.all(length[5 * N:] == 0)
    assert len(store) == 3 * N + 1
    assert all(length[r] == n for r, n in zip(store.rows, store.lengths))
    assert ar.queue_state()[2] == 5 * N
    ar.close(), eng.close()


@pytest.mark.parametrize("temperature", [0.0, 1.0])
def test_arena_step_zero_simulations(temperature):
    """number_of_mcts_simulation = 0 (experiment_130..133 configs): policy_step gets the root priors
    (monte_carlo_tree_search.py:218-221); through the real search kernel with S = 0."""
    import torch
    from muzero_amd.selfplay import Arena
    from oracle import mz_oracle as mzo
    G, T, N = 100, 30, 260
    eng, dims, _ = make_engine(G, 0)
    ar = Arena(eng, G, N, T)
    orc = mzo.ArenaOracle(G, T, n_rows=N)
    seeds = np.arange(N, dtype=np.int64) + 77
    obs = ar.begin(0, seeds, 1)
    orc.begin(seeds, 1)
    for t in range(N * T):
        eng.search(obs, None, None, 0, True)
        r = {k: host(v) for k, v in eng.results_view().items()}
        assert np.all(r["visits"] == 0)
        ar.step(temperature)
        n_alive = orc.step(r["visits"], r["root_value"], r["root_prior"], temperature, from_prior=True)
        assert int(ar.alive_count.item()) == n_alive
        if n_alive == 0:
            break
    got = arena_host(ar)
    assert np.array_equal(got["length"], orc.length) and orc.length.min() >= 1
    for k, w in (("obs", orc.traj_obs), ("action", orc.traj_action), ("root_value", orc.traj_root),
                 ("policy", orc.traj_policy)):
        assert np.array_equal(got[k], w), k
    ar.close(), eng.close()


@pytest.mark.parametrize("td,K", [(3, 5), (10, 5), (50, 10)])
def test_make_batch_vs_oracle(td, K):
    import torch
    from muzero_amd.selfplay import Arena, discount_powers
    from oracle import mz_oracle as mzo
    G, T = 128, 64
    eng, dims, _ = make_engine(G, 0)
    ar = Arena(eng, G, 2 * G, T)
    rs = np.random.RandomState(td)
    for block in range(2):                       # two blocks of rows, different seeds
        ar.begin(block * G, np.arange(G, dtype=np.int64) + 1000 * block, episode=block)
        for t in range(T):
            _, res = fake_results(rs, G, 2, eng.device)
            ar.step(1.0, res, num_simulations=50)
        torch.cuda.synchronize()
        assert int(ar.alive_count.item()) == 0
    got = arena_host(ar)
    assert got["length"].min() >= 1 and got["length"].max() <= T
    B = 777
    rows = rs.randint(0, 2 * G, B)
    pos = np.array([rs.randint(0, got["length"][r]) for r in rows])
    pos[:G] = got["length"][rows[:G]] - 1        # last position: mostly absorbing targets
    row_pos = np.stack([rows, pos], 1).astype(np.int32)
    out = ar.make_batch(row_pos, K, td, 0.997)
    torch.cuda.synchronize()
    want = mzo.make_batch(got["length"], got["obs"], got["action"], got["reward"], got["root_value"],
                          got["policy"], row_pos, K, td, discount_powers(0.997, td))
    for name, g, w in zip(("obs", "actions", "tv", "tr", "tp"), out, want):
        assert np.array_equal(host(g), w), name
    with pytest.raises(ValueError):
        ar.make_batch(np.array([[2 * G, 0]]), K, td, 0.997)
    # Game.make_priority on the device: full width and trimmed to the longest game
    wantp = mzo.make_priority(got["length"], got["reward"], got["root_value"], td, discount_powers(0.997, td))
    assert np.array_equal(host(ar.priority(0, 2 * G, td, 0.997, T)), wantp)
    longest = int(got["length"][G:].max())
    assert np.array_equal(host(ar.priority(G, G, td, 0.997, longest)), wantp[G:, :longest])
    ar.close(), eng.close()


def test_play_round_real_search_vs_oracle():
    """The whole per-move loop on the GPU (fill_streams -> search -> arena step): the oracle arena,
    fed with the same search results move by move, records identical trajectories."""
    import torch
    from muzero_amd.selfplay import Arena
    from oracle import mz_oracle as mzo
    G, T, S, N = 192, 40, 20, 500
    eng, dims, _ = make_engine(G, S)
    ar = Arena(eng, G, N, T)
    orc = mzo.ArenaOracle(G, T, n_rows=N)
    seeds = np.arange(N, dtype=np.int64) * 31 + 7
    obs = ar.begin(0, seeds, 0)
    orc.begin(seeds, 0)
    for t in range(N * T):
        dirichlet, noise = eng.fill_streams(ar.arrays["stream_seed"], S, True)
        eng.search(obs, dirichlet, noise, S, True)
        r = {k: host(v) for k, v in eng.results_view().items()}
        torch.cuda.synchronize()
        assert np.all(r["visits"].sum(1) == S)
        ar.step(1.0)
        n_alive = orc.step(r["visits"], r["root_value"], r["root_prior"], 1.0)
        assert int(ar.alive_count.item()) == n_alive
        if n_alive == 0:
            break
    got = arena_host(ar)
    assert np.array_equal(got["length"], orc.length)
    for k, w in (("obs", orc.traj_obs), ("action", orc.traj_action), ("root_value", orc.traj_root),
                 ("policy", orc.traj_policy)):
        assert np.array_equal(got[k], w), k
    assert 8 <= got["length"].mean() <= T         # CartPole under a random-ish policy: 10-40 moves
    ar.close(), eng.close()


def test_play_round_helper_and_learning_cycle():
    """learning_cycle end to end on one GPU: play, save, sample, gather, train; the search of the
    next round reads the updated weights."""
    import torch
    from muzero_amd.selfplay import Arena, ReplayStore, SelfPlay
    from muzero_amd.train import Trainer
    G, T, S, B, K, td, N = 128, 48, 10, 64, 5, 10, 300
    eng, dims, blob = make_engine(G, S)
    tr = Trainer(eng, K, B, "adam", 1e-3)
    store = ReplayStore(window_size=2 * N, batch_size=B, num_unroll=K, td_steps=td, discount=0.997)
    ar = Arena(eng, G, 4 * N, T)
    np.random.seed(0)
    loop = SelfPlay(None, eng, tr, ar, store, S, games_per_iteration=N, seed=42)
    rewards, losses = loop.learning_cycle(4, number_of_training_before_self_play=3,
                                          temperature_type="static_one_temperature")
    assert len(rewards) == 4 and all(1 <= r <= T for r in rewards)
    assert all(np.isfinite(l) and l > 0 for l in losses)
    assert len(store) == 2 * N + 1                                  # the reference's window rule
    assert store.big_n_of_importance_sampling_ratio == sum(store.lengths)
    assert not np.array_equal(tr.weights(), blob)
    # rows of the store are live arena rows with matching lengths
    length = host(ar.arrays["length"])
    assert all(length[r] == n for r, n in zip(store.rows, store.lengths))
    with pytest.raises(ValueError):
        SelfPlay(None, eng, tr, Arena(eng, G, 2 * N, T), store, S, N)   # too few rows for the window
    ar.close(), eng.close()


def test_learning_cycle_from_reference_config(tmp_path):
    """A reference-schema experiment config (the CartPole / mlp_model shape every shipped config has)
    drives the batched cycle; td_steps 50 and unroll 10 as experiment_123."""
    import json
    from muzero_amd.config import load_config
    from muzero_amd.selfplay import SelfPlay
    cfg = {"game": {"env": "CartPole-v1", "render": None}, "random_seed": {"np_random_seed": 0, "torch_manual_seed": 0, "env_seed": 0},
           "muzero": {"model_structure": "mlp_model", "state_space_dimensions": 31, "hidden_layer_dimensions": 64,
                      "number_of_hidden_layer": 0, "k_hypothetical_steps": 10, "optimizer": "adam",
                      "lr_scheduler": "cosineannealinglr", "learning_rate": 0.01, "loss_type": "general",
                      "num_of_epoch": 10000, "device": "cpu", "load": False, "use_amp": False, "scaler_on": False,
                      "bin_method": "uniform_bin", "bin_decomposition_number": 10},
           "replaybuffer": {"window_size": 500, "batch_size": 128, "td_steps": 50, "game_sampling": "priority",
                            "position_sampling": "priority"},
           "monte_carlo_tree_search": {"pb_c_base": 19652, "pb_c_init": 1.25, "discount": 0.997,
                                       "root_dirichlet_alpha": 0.25, "root_exploration_fraction": 0.25},
           "gameplay": {"discount": 0.997, "limit_of_game_play": 500},
           "learning_cycle": {"number_of_iteration": 10000, "number_of_self_play_before_training": 1,
                              "number_of_training_before_self_play": 1, "number_of_mcts_simulation": 11,
                              "tempererature_type": "linear_decrease_temperature", "model_tag_number": 123,
                              "verbose": True, "number_of_worker_selfplay": 0},
           "play_game_from_checkpoint": {}}
    path = tmp_path / "experiment.json"
    path.write_text(json.dumps(cfg))
    np.random.seed(1)
    loop = SelfPlay.from_config(load_config(str(path)), n_slots=256, games_per_iteration=600, window_size=1500)
    assert loop.arena.G == 256 and loop.arena.T == 500 and loop.S == 11 and loop.store.td_steps == 50
    mz = loop.mz
    w0 = mz.weights_blob().copy()
    rewards, losses = loop.run_config(number_of_iteration=3, model_tag_number=777,
                                      checkpoint_directory=str(tmp_path / "ckpt"))
    assert len(rewards) == 3 and all(np.isfinite(losses)) and len(loop.store) == 1501
    assert loop.trainer.count == 3
    # the drop-in model object follows the trainer it shares a handle with: step count,
    # store_loss, and the weights save_model / a later Monte_carlo_tree_search.run(model=mz) see
    assert mz.count == 3 and len(mz.store_loss) == 3 and len(mz.store_loss[-1]) == 3 * 10
    assert abs(float(mz.store_loss[-1][0]) - losses[-1]) <= 1e-6 * abs(losses[-1])
    trained = loop.trainer.weights()
    assert not np.array_equal(trained, w0) and np.array_equal(mz.weights_blob(), trained)
    # the best-reward checkpoint was written before that iteration's training step (self_play.py:248-262)
    assert (tmp_path / "ckpt" / "777_muzero_init_variables.json").exists()
    mz.save_model(directory=str(tmp_path / "final"), tag=778, model_update_condition=True)
    assert np.array_equal(np.load(tmp_path / "final" / "778_muzero_weights.npz")["blob"], trained)
    from muzero_amd.model import Muzero
    back = Muzero(load=True)
    back.load_model(str(tmp_path / "final"), tag=778)
    assert np.array_equal(back.weights_blob(), trained) and back.dims == mz.dims
    # Muzero.train after the loop continues the schedule instead of rewinding it
    row_pos, ratio, game_pos = loop.store.sample_batch()
    assert mz._trainer is loop.trainer and mz._trainer.count == 3
    loop.arena.close(), loop.eng.close()


def test_arena_errors_are_loud():
    """Bad arguments come back as error codes with a message (never a fault on the GPU)."""
    import ctypes as C
    import torch
    from muzero_amd import _lib
    from muzero_amd.selfplay import Arena
    lib = _lib.load()
    eng, dims, _ = make_engine(8, 0)
    h = C.c_void_p()
    for cfg in (_lib.ArenaConfig(0, 7, 8, 8, 10), _lib.ArenaConfig(0, 0, 0, 8, 10), _lib.ArenaConfig(0, 0, 8, 4, 10),
                _lib.ArenaConfig(0, 0, 8, 8, 0)):
        assert lib.mzs_arena_create(C.byref(cfg), C.byref(h)) != 0 and lib.mzs_last_error()
    ar = Arena(eng, 8, 16, 10)
    with pytest.raises(ValueError):
        ar.begin(10, np.arange(8), 0)                       # rows 10..17 do not exist
    with pytest.raises(ValueError):
        ar.begin(0, np.zeros(0, np.int64), 0)
    seeds = torch.arange(8, dtype=torch.int64, device=eng.device)
    assert lib.mzs_arena_begin(ar.h, 12, 8, C.c_void_p(seeds.data_ptr()), 0, C.c_void_p(ar.obs.data_ptr()), None) != 0
    assert lib.mzs_arena_begin(ar.h, 0, 8, None, 0, C.c_void_p(ar.obs.data_ptr()), None) != 0
    assert lib.mzs_arena_step(ar.h, None, None, None, C.c_double(0.0), 50, None, None, None) != 0
    assert lib.mzs_arena_priority(ar.h, 8, 16, 3, None, 10, None, None) != 0
    assert lib.mzs_arena_make_batch(ar.h, 4, None, 5, 3, None, None, None, None, None, None, None) != 0
    assert lib.mzs_arena_array(ar.h, 99) is None
    with pytest.raises(ValueError):
        Arena(eng, 9, 16, 10)                               # more slots than the engine searches at once
    ar.close(), eng.close()
