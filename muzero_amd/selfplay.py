"""Batched self-play on the MI355X: the callers on either side of the search
(/root/reference self_play.py, game.py, replay_buffer.py).

The reference plays ONE game at a time through Python objects (Game lists, one MCTS per move,
ReplayBuffer of Game objects).  Here G episodes advance in lock step on one GPU:

  per move   mzs_search_streams (streams + G trees) -> mzs_arena_step (store_search_statistics,
             policy_step, env.step and the trajectory append, all in HBM)       self_play.py:62-73
  per round  finished trajectories stay in HBM rows; the host keeps only what the reference's
             ReplayBuffer needs to SAMPLE: lengths and priorities (ReplayStore)  replay_buffer.py
  per batch  the host draws (game, position) pairs exactly as ReplayBuffer.sample_batch does,
             mzs_arena_make_batch gathers observations / actions / make_target straight into the
             buffers of the training kernels                                     replay_buffer.py:131-166

Nothing here computes on the CPU what the reference computes per move; the host work is
O(batch) index arithmetic on NumPy's own generator (so seeding np.random reproduces the
reference's draws).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .engine import _DevView, _ptr


# ---------------------------------------------------------------------------- schedules
def _scaler(x, newmin=0.0, newmax=1.0):
    """self_play.py:81-97."""
    x = np.asarray(x, np.float64)
    oldmin, oldmax = x.min(), x.max()
    if oldmax == oldmin:
        return np.full(x.shape, newmin if oldmin < newmin else (newmax if oldmin > newmax else oldmin))
    return (x - oldmin) * ((newmax - newmin) / (oldmax - oldmin)) + newmin


def temperature_scheduler(epoch=1, actual_epoch=1, mode="static_temperature"):
    """self_play.py:102-139: the action-selection temperature of iteration `actual_epoch`."""
    if mode == "reversal_tanh_temperature":
        arr = np.arange(1, epoch + 1)
        return float(((1 - np.tanh(_scaler(arr, 0.001, 0.75))) * 1.1)[actual_epoch - 1])
    if mode == "extreme_temperature":
        for frac, t in ((100, 3), (200, 2), (300, 1), (400, .7), (500, .5), (600, .4), (700, .0625)):
            if actual_epoch < epoch * (frac / 700):
                return t
        return None
    if mode == "linear_decrease_temperature":
        return 1 if epoch * 0.5 > actual_epoch else (0.5 if epoch * 0.75 > actual_epoch else 0.2)
    if mode == "static_temperature":
        return 0.0
    if mode == "static_one_temperature":
        return 1
    raise ValueError(f"temperature mode {mode!r}")


# ---------------------------------------------------------------------------- priorities
def discount_powers(discount, n):
    """[discount ** i for i in 0..n] evaluated as the reference does (Python float pow)."""
    return np.array([float(discount) ** i for i in range(n + 1)], np.float64)


def make_priority(root_values, rewards, lengths, td_steps, discount):
    """Game.make_priority (game.py:262-284) for many games at once.

    root_values / rewards: [R, T] float32 rows of the arena, lengths [R].  Returns the
    [R, T] float64 |root_value - n-step target| (entries past a game's length are 0) with the
    reference's arithmetic under NumPy 2: the target accumulates in float32 when a bootstrap
    value exists and in float64 otherwise (see mzs_selfplay.h target_value)."""
    rv = np.ascontiguousarray(root_values, np.float32)
    rw = np.ascontiguousarray(rewards, np.float32)
    ln = np.asarray(lengths, np.int64)
    R, T = rv.shape
    td = int(td_steps)
    dp = discount_powers(discount, td)
    pos = np.arange(T)
    has = (pos[None, :] + td) < ln[:, None]
    boot = rv[:, np.minimum(pos + td, T - 1)] * np.float32(dp[td])
    v32 = np.where(has, boot, np.float32(0)).astype(np.float32)
    v64 = np.zeros((R, T), np.float64)
    for i in range(td):
        idx = pos + i
        ok = idx[None, :] < ln[:, None]
        term = rw[:, np.minimum(idx, T - 1)].astype(np.float64) * dp[i]
        v32 = np.where(ok, v32 + term.astype(np.float32), v32)
        v64 = np.where(ok, v64 + term, v64)
    target = np.where(has, v32.astype(np.float64), v64)
    prio = np.abs(rv.astype(np.float64) - target)
    prio[pos[None, :] >= ln[:, None]] = 0.0
    return prio


class ReplayStore:
    """ReplayBuffer (replay_buffer.py) over arena rows: same window rule, same priorities, same
    draws from NumPy's generator, but a "game" is a trajectory row in HBM and a batch is a
    [B, 2] table of (row, position) handed to mzs_arena_make_batch.

    Differences that are deliberate and stated:
      * game_sampling="uniform" raises TypeError in the reference (replay_buffer.py:96 indexes the
        buffer with a Game object); here it draws a uniform buffer index.
      * the per-sample np.random.choice calls are issued as one random_sample(B) per stage, which
        consumes the generator identically (pinned by tests/test_selfplay_oracle.py)."""

    def __init__(self, window_size, batch_size, num_unroll, td_steps, game_sampling="priority",
                 position_sampling="priority", discount=0.997, rng=None, native=True):
        for m in (game_sampling, position_sampling):
            if m not in ("uniform", "priority"):
                raise ValueError(f"sampling mode {m!r}")
        self.window_size, self.batch_size = int(window_size), int(batch_size)
        self.num_unroll, self.td_steps = int(num_unroll), int(td_steps)
        self.game_sampling, self.position_sampling = game_sampling, position_sampling
        self.discount = float(discount)
        self.rng = np.random if rng is None else rng
        self.rows = np.zeros(0, np.int64)      # buffer order -> arena row
        self.lengths = np.zeros(0, np.int64)   # buffer order -> game_length
        self.prio_game = np.zeros(0, np.float64)
        self.soft_prio_game = np.zeros(0)
        self.big_n_of_importance_sampling_ratio = 0
        # prio_position: one [n, width] float64 block per save_games call + (block, index, length) per entry
        self._blocks, self._next_block = {}, 0
        self._blk, self._bidx, self._plen, self._addr = (np.zeros(0, np.int64) for _ in range(4))
        # native=True: the per-sample loops of sample_batch / update_value run in the library
        # (muzero_amd/csrc/mzs_replay.h: NumPy's float64 arithmetic restated, same results bit for
        # bit); native=False: the NumPy formulation below, kept as the checked definition
        self.lib = _lib.load() if native else None

    def __len__(self):
        return len(self.rows)

    @property
    def uses_priority(self):
        return "priority" in (self.game_sampling, self.position_sampling)

    class _PositionPriorities:
        """prio_position of the reference: entry g is the writable float64 [length] view."""

        def __init__(self, store):
            self.s = store

        def __len__(self):
            return len(self.s._blk)

        def __getitem__(self, g):
            s = self.s
            blk, i, n = s._blocks[int(s._blk[g])], int(s._bidx[g]), int(s._plen[g])
            return blk[i: i + n] if blk.ndim == 1 else blk[i, :n]     # packed block: i = the game's offset

        def __iter__(self):
            return (self[g] for g in range(len(self)))

    @property
    def prio_position(self):
        return self._PositionPriorities(self)

    def save_games(self, rows, lengths, root_values=None, rewards=None, priorities=None):
        """save_game (replay_buffer.py:59-84) for a block of finished games, in order, without a
        per-game Python loop.  priorities: [n, >= longest game] float64 from Arena.priority
        (mzs_arena_priority), zero past each game's end, or the packed 1-d form of
        Arena.priority_packed (game after game, sum(lengths) entries, no padding); without it they are
        computed here from [n, T] float32 host copies of root_values / rewards."""
        rows = np.asarray(rows, np.int64).reshape(-1)
        lengths = np.asarray(lengths, np.int64).reshape(-1)
        m, n0 = len(rows), len(self.rows)
        # one pop per save once the buffer holds window_size + 1 games (replay_buffer.py:61-63)
        pops = n0 + m - min(n0 + m, self.window_size + 1)
        if self.uses_priority:
            if np.any(lengths == 0):
                raise ValueError("zero-size array to reduction operation maximum which has no identity")
            prio = priorities if priorities is not None else make_priority(root_values, rewards, lengths,
                                                                          self.td_steps, self.discount)
            prio = np.ascontiguousarray(prio, np.float64)
            # as the reference: each list is popped only under its own mode although both are
            # appended whenever either mode is "priority" (replay_buffer.py:64-67, :70-79)
            pg = pops if self.game_sampling == "priority" else 0
            pp = pops if self.position_sampling == "priority" else 0
            if prio.ndim == 1:          # packed: game i at [off[i], off[i] + lengths[i])
                off = np.cumsum(lengths) - lengths
                if prio.shape[0] != int(lengths.sum()):
                    raise ValueError("packed priorities must hold sum(lengths) entries")
                game_max, index, stride = np.maximum.reduceat(prio, off), off, prio.strides[0]
            else:
                game_max, index, stride = prio.max(axis=1), np.arange(m, dtype=np.int64), prio.strides[0]
            self.prio_game = np.concatenate([self.prio_game, game_max])[pg:]
            bid, self._next_block = self._next_block, self._next_block + 1
            self._blocks[bid] = prio
            self._blk = np.concatenate([self._blk, np.full(m, bid, np.int64)])[pp:]
            self._bidx = np.concatenate([self._bidx, index])[pp:]
            self._plen = np.concatenate([self._plen, lengths])[pp:]
            self._addr = np.concatenate([self._addr, prio.ctypes.data + index * stride])[pp:]
            for old in [k for k in self._blocks if len(self._blk) and k < int(self._blk[0])]:
                del self._blocks[old]
            self.soft_prio_game = self.prio_game / np.sum(self.prio_game)
            self._game_cdf = None          # cdf of soft_prio_game, built by the first sample_batch after a save
        self.rows = np.concatenate([self.rows, rows])[pops:]
        self.lengths = np.concatenate([self.lengths, lengths])[pops:]
        self.big_n_of_importance_sampling_ratio = int(self.lengths.sum())

    @staticmethod
    def _choice(p, u):
        """RandomState.choice(len(p), p=p) for the uniform numbers u it would draw."""
        if not np.all(np.isfinite(p)):
            raise ValueError("probabilities contain NaN")
        if np.any(p < 0):
            raise ValueError("probabilities are not non-negative")
        if abs(float(p.sum()) - 1.0) > 1.5e-8:
            raise ValueError("probabilities do not sum to 1")
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        return cdf.searchsorted(u, side="right")

    def sample_batch(self):
        """sample_batch (replay_buffer.py:131-166) without the gather: returns
        (row_pos int32 [B, 2] = (arena row, position), importance_sampling_ratio, game_pos [B, 2] =
        (buffer index, position) as the reference returns it)."""
        B = self.batch_size
        if self.game_sampling == "priority":
            # np.random.choice(range(n), p=soft_prio_game) B times; soft_prio_game only changes in
            # save_games (replay_buffer.py:80), so its validated cdf is kept between batches
            if getattr(self, "_game_cdf", None) is None or len(self._game_cdf) != len(self.soft_prio_game):
                self._choice(self.soft_prio_game, np.zeros(1))
                cdf = np.cumsum(self.soft_prio_game)
                cdf /= cdf[-1]
                self._game_cdf = cdf
            gi = self._game_cdf.searchsorted(self.rng.random_sample(B), side="right")
        else:
            gi = self.rng.randint(0, len(self.rows), size=B)
        pos = np.zeros(B, np.int64)
        mouve = np.zeros(B, np.float64)
        if np.any(self.lengths[gi] == 0):
            raise Exception("Game need to return at least one reward")
        prio_pos = self.position_sampling == "priority"
        u = self.rng.random_sample(B) if prio_pos else None
        if self.lib is not None and self.uses_priority:
            if not prio_pos:
                for b in range(B):
                    pos[b] = self.rng.randint(0, self.lengths[gi[b]] - 1)
            rows = np.ascontiguousarray(self._addr[gi])
            rlen = np.ascontiguousarray(self._plen[gi], np.int32)
            p_ = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
            rc = self.lib.mzs_replay_sample_positions(B, p_(rows), p_(rlen), p_(u), p_(pos), p_(mouve))
            if rc != 0:
                raise ValueError(self.lib.mzs_last_error().decode())
            return self._finish_batch(gi, pos, mouve)
        pp, cache = self.prio_position, {}
        for b in range(B):
            g = int(gi[b])
            if self.uses_priority:
                if g not in cache:      # soft_prio_position (replay_buffer.py:102) and its cdf, once per game
                    p = pp[g]
                    with np.errstate(invalid="ignore", divide="ignore"):   # an all-zero row -> NaN, rejected below as the reference's np.random.choice does
                        soft = p / p.sum()
                    cdf = None
                    if prio_pos:
                        if not np.isfinite(soft[-1]) or soft.min() < 0:     # what RandomState.choice rejects
                            raise ValueError("probabilities contain NaN")
                        cdf = soft.cumsum()
                        cdf /= cdf[-1]
                    cache[g] = (soft, cdf)
                soft, cdf = cache[g]
            if prio_pos:
                pos[b] = cdf.searchsorted(u[b], side="right")
            else:
                pos[b] = self.rng.randint(0, self.lengths[g] - 1)
            if self.uses_priority:
                mouve[b] = soft[pos[b]]
        return self._finish_batch(gi, pos, mouve)

    def _finish_batch(self, gi, pos, mouve):
        game_pos = np.stack([gi, pos], 1)
        row_pos = np.stack([self.rows[gi], pos], 1).astype(np.int32)
        if self.uses_priority:
            priority = self.soft_prio_game[gi] * mouve
            ratio = 1 / (self.big_n_of_importance_sampling_ratio * priority)
            return row_pos, ratio, game_pos
        return row_pos, np.array([0]), game_pos

    def update_value(self, new_value, position):
        """update_value (replay_buffer.py:169-177); new_value [K][B] or [K][B][1]."""
        nv = np.asarray(new_value, np.float64)
        nv = np.ascontiguousarray(nv.reshape(nv.shape[0], nv.shape[1]))
        position = np.asarray(position, np.int64)
        if self.lib is not None:
            gi, pos = np.ascontiguousarray(position[:, 0]), np.ascontiguousarray(position[:, 1])
            rows = np.ascontiguousarray(self._addr[gi])
            rlen = np.ascontiguousarray(self._plen[gi], np.int32)
            glen = np.ascontiguousarray(self.lengths[gi], np.int32)
            p_ = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            _lib.check(self.lib.mzs_replay_update_value(len(gi), self.num_unroll, p_(rows), p_(rlen), p_(glen),
                                                        p_(gi), p_(pos), p_(nv), p_(self.prio_game)),
                       "mzs_replay_update_value")
            return
        pp = self.prio_position
        for count, (g, p) in enumerate(position):
            hi = min(self.num_unroll + p, self.lengths[g] - 1)
            row = pp[g]
            if hi > p:
                row[p:hi] = nv[: hi - p, count]
            self.prio_game[g] = np.max(row)


# ---------------------------------------------------------------------------- device arena
class Arena:
    """Owner of one mzs_arena (include/mzsearch.h): G environment slots + trajectory rows in HBM."""

    ARRAYS = dict(length=(_lib.ARENA_LENGTH, "<i4"), obs=(_lib.ARENA_OBS, "<f4"),
                  action=(_lib.ARENA_ACTION, "<i4"), reward=(_lib.ARENA_REWARD, "<f4"),
                  root_value=(_lib.ARENA_ROOT_VALUE, "<f4"), policy=(_lib.ARENA_POLICY, "<f4"),
                  alive=(_lib.ARENA_ALIVE, "<i4"), state=(_lib.ARENA_STATE, "<f8"),
                  stream_seed=(_lib.ARENA_STREAM_SEED, "<i8"), slot_row=(_lib.ARENA_SLOT_ROW, "<i4"),
                  queue=(_lib.ARENA_QUEUE, "<i4"))

    def __init__(self, engine, n_slots, n_rows=None, max_moves=500, env=_lib.ENV_CARTPOLE):
        self.eng, self.lib = engine, engine.lib
        self.G, self.T = int(n_slots), int(max_moves)
        self.n_rows = int(n_rows) if n_rows is not None else self.G
        if self.G > engine.max_games:
            raise ValueError(f"{self.G} slots exceed the engine's max_games {engine.max_games}")
        cfg = _lib.ArenaConfig(engine.device.index, int(env), self.G, self.n_rows, self.T)
        h = C.c_void_p()
        _lib.check(self.lib.mzs_arena_create(C.byref(cfg), C.byref(h)), "mzs_arena_create")
        self.h = h
        self.obs_dim = int(self.lib.mzs_arena_obs_dim(h))
        self.A = int(self.lib.mzs_arena_n_actions(h))
        if (self.obs_dim, self.A) != (engine.obs_dim, engine.A):
            raise ValueError(f"environment is obs {self.obs_dim} / {self.A} actions, the model "
                             f"{engine.obs_dim} / {engine.A}")
        dev = engine.device
        self.obs = torch.zeros((self.G, self.obs_dim), dtype=torch.float32, device=dev)
        self.alive_count = torch.zeros(1, dtype=torch.int32, device=dev)
        shapes = dict(length=(self.n_rows,), obs=(self.n_rows, self.T, self.obs_dim),
                      action=(self.n_rows, self.T), reward=(self.n_rows, self.T),
                      root_value=(self.n_rows, self.T), policy=(self.n_rows, self.T, self.A),
                      alive=(self.G,), state=(self.G, 4), stream_seed=(self.G,), slot_row=(self.G,), queue=(3,))
        self.arrays = {}
        for k, (which, ts) in self.ARRAYS.items():
            p = self.lib.mzs_arena_array(h, which)
            self.arrays[k] = torch.as_tensor(_DevView(p, shapes[k], ts), device=dev)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mzs_arena_destroy(self.h)
            self.h = None

    __del__ = close

    def begin(self, row0, seeds, episode):
        """Start a round of len(seeds) games recorded in rows [row0, row0 + len(seeds)): the first
        G start now, the others as slots free up.  Returns the device observation tensor
        [G, obs_dim] that the moves keep updating in place."""
        seeds = self.eng._dev(np.asarray(seeds, np.int64) if not torch.is_tensor(seeds) else seeds, torch.int64)
        n = int(seeds.shape[0])
        if seeds.dim() != 1 or n < 1 or row0 < 0 or row0 + n > self.n_rows:
            raise ValueError(f"{n} games from row {row0} do not fit {self.n_rows} rows")
        self.seeds, self.episode, self.n_games = seeds, int(episode), n
        self.moves_issued, self.pending = 0, []
        _lib.check(self.lib.mzs_arena_begin(self.h, int(row0), n, _ptr(seeds), int(episode), _ptr(self.obs),
                                            self.eng._stream()), "mzs_arena_begin")
        return self.obs

    def enqueue(self, seeds, episode, temperature=None):
        """mzs_arena_enqueue: len(seeds) more games behind the queue's end; they start in the slots
        finished games free, so the lock-step searches stay full across the boundary.  temperature:
        these games' action-selection temperature (None: whatever each step() is given)."""
        seeds = self.eng._dev(np.asarray(seeds, np.int64) if not torch.is_tensor(seeds) else seeds, torch.int64)
        n = int(seeds.shape[0])
        self._keep_q = seeds
        _lib.check(self.lib.mzs_arena_enqueue(self.h, n, _ptr(seeds), int(episode),
                                              float("nan") if temperature is None else float(temperature),
                                              self.eng._stream()), "mzs_arena_enqueue")

    def queue_state(self):
        """(next unplayed, queue end, done_upto) as virtual rows — one small device->host read."""
        q = self.arrays["queue"].cpu().numpy()
        return int(q[0]), int(q[1]), int(q[2])

    def step(self, temperature, results=None, num_simulations=None):
        """One move of every live slot from the engine's latest search results (num_simulations:
        the simulation count of that search; 0 makes policy_step use the root priors)."""
        S = self.eng.num_simulations if num_simulations is None else int(num_simulations)
        r = self.eng.results_view() if results is None else results
        if r["visits"].shape != (self.G, self.A):
            raise ValueError("the last search did not cover every slot")
        _lib.check(self.lib.mzs_arena_step(self.h, _ptr(r["visits"]), _ptr(r["root_value"]),
                                           _ptr(r["root_prior"]), float(temperature), S, _ptr(self.obs),
                                           _ptr(self.alive_count), self.eng._stream()), "mzs_arena_step")
        self.moves_issued += 1

    def mark(self):
        """An event behind the moves issued so far + the index of the last of them: once the event
        has completed, status(index) is readable (no stream synchronisation, later moves keep running)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.eng.device))
        self.pending.append((ev, self.moves_issued - 1))
        return self.pending[-1]

    def status(self, move):
        """mzs_arena_status: what lock-step move `move` left behind, read from pinned host memory:
        dict(move, next, end, done_upto, alive, max_consumed)."""
        out = (C.c_int32 * 6)()
        _lib.check(self.lib.mzs_arena_status(self.h, int(move), out), "mzs_arena_status")
        return dict(zip(("move", "next", "end", "done_upto", "alive", "max_consumed"), (int(v) for v in out)))

    def play(self, n_moves, num_simulations, temperature, train=True, lo=1e-7, hi=2e-7):
        """n_moves x (search with device-made streams, step) in one library call (mzs_arena_play):
        nothing but kernel launches between the moves."""
        e, S = self.eng, int(num_simulations)
        dirichlet, noise = e.stream_scratch(self.G, S)
        _lib.check(self.lib.mzs_arena_play(self.h, e.h, int(n_moves), S, int(bool(train)), e.alpha, float(lo),
                                           float(hi), int(e.stream_prefix), float(temperature), _ptr(self.obs),
                                           _ptr(dirichlet), _ptr(noise), noise.shape[1], _ptr(self.alive_count),
                                           e._stream()), "mzs_arena_play")
        e.n_games, e.num_simulations = self.G, S
        self.moves_issued += int(n_moves)

    def priority(self, row0, n_rows, td_steps, discount, longest):
        """Game.make_priority for rows [row0, row0 + n_rows) on the device: float64 tensor
        [n_rows, longest] (longest = the longest game among those rows)."""
        dev = self.eng.device
        dp_d = torch.as_tensor(discount_powers(discount, int(td_steps))).to(dev)
        stride = max(1, min(int(longest), self.T))
        out = torch.empty((int(n_rows), stride), dtype=torch.float64, device=dev)
        self._keep_p = dp_d
        _lib.check(self.lib.mzs_arena_priority(self.h, int(row0), int(n_rows), int(td_steps), _ptr(dp_d), stride,
                                               _ptr(out), self.eng._stream()), "mzs_arena_priority")
        return out

    def priority_packed(self, row0, lengths_d, total, longest, td_steps, discount):
        """Game.make_priority for rows [row0, row0 + len(lengths_d)) without the padding: one float64
        device tensor of `total` = sum(lengths) entries, game after game (mzs_arena_priority_packed).
        lengths_d: the rows' lengths on the device."""
        dev = self.eng.device
        dp_d = torch.as_tensor(discount_powers(discount, int(td_steps))).to(dev)
        ln = lengths_d.to(torch.int64)
        off = torch.cumsum(ln, 0) - ln
        out = torch.empty(int(total), dtype=torch.float64, device=dev)
        self._keep_p = (dp_d, off)
        _lib.check(self.lib.mzs_arena_priority_packed(self.h, int(row0), int(ln.shape[0]), int(td_steps), _ptr(dp_d),
                                                      max(1, int(longest)), _ptr(off), _ptr(out),
                                                      self.eng._stream()), "mzs_arena_priority_packed")
        return out

    def make_batch(self, row_pos, num_unroll, td_steps, discount):
        """ReplayBuffer.sample_batch's gather + Game.make_target on the device, in the layout
        muzero_amd.train.Trainer.grad_step takes: obs [B, obs], actions [K, B, A], target value /
        reward [K, B], target policy [K, B, A]."""
        rp = np.ascontiguousarray(row_pos, np.int32)
        B, K = rp.shape[0], int(num_unroll)
        if rp.ndim != 2 or rp.shape[1] != 2:
            raise ValueError("row_pos must be [B, 2]")
        if B and (rp[:, 0].min() < 0 or rp[:, 0].max() >= self.n_rows or rp[:, 1].min() < 0 or
                  rp[:, 1].max() >= self.T):
            raise ValueError("row_pos out of range")
        dev = self.eng.device
        rp_d = torch.as_tensor(rp).to(dev)
        dp_d = torch.as_tensor(discount_powers(discount, int(td_steps))).to(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        obs, actions = torch.empty((B, self.obs_dim), **f32), torch.empty((K, B, self.A), **f32)
        tv, tr, tp = torch.empty((K, B), **f32), torch.empty((K, B), **f32), torch.empty((K, B, self.A), **f32)
        self._keep = (rp_d, dp_d)
        _lib.check(self.lib.mzs_arena_make_batch(self.h, B, _ptr(rp_d), K, int(td_steps), _ptr(dp_d), _ptr(obs),
                                                 _ptr(actions), _ptr(tv), _ptr(tr), _ptr(tp),
                                                 self.eng._stream()), "mzs_arena_make_batch")
        return obs, actions, tv, tr, tp


def play_round(engine, arena, seeds, episode, num_simulations, temperature, row0=0, check_every=16,
               train=True):
    """learning_cycle's self-play phase (self_play.py:230-236: len(seeds) play_game calls, each the
    loop of self_play.py:62-73) on arena.G slots at once.  Returns the number of lock-step moves.
    Everything between the alive checks is enqueued without a host sync."""
    obs = arena.begin(row0, seeds, episode)
    S = int(num_simulations)
    moves = 0
    while moves < arena.n_games * arena.T:
        arena.play(check_every, S, temperature, train)
        ev, last = arena.mark()
        arena.pending.pop()
        moves += check_every
        ev.synchronize()
        st = arena.status(last)
        engine.adapt_stream_prefix(st["max_consumed"])
        if st["alive"] == 0:
            break
    return moves


def play_moves(engine, arena, n_moves, num_simulations, temperature, train=True):
    """n_moves lock-step moves, enqueued without a host sync."""
    arena.play(n_moves, num_simulations, temperature, train)


def play_until(engine, arena, done_upto, num_simulations, temperature, check_every=16, train=True, max_moves=None,
               depth=2):
    """Lock-step moves until every game below virtual row `done_upto` has finished (rolling rounds:
    the slots those games free go to the queued games of the next round).  Returns the number of
    lock-step moves examined, the status the last of them left (Arena.status) and the event behind it.

    The GPU's queue is never drained to look: while games are still queued, `depth` chunks of
    check_every moves are kept in flight and the host waits on the event behind the OLDEST chunk,
    then reads that chunk's status entry from pinned memory.  When the round is seen complete, the
    younger chunks are already running the next round's games (arena.pending keeps their events
    for the next call).  In the final drain (queue empty) one chunk at a time is issued, so that no
    search runs on an arena that has nothing left to play."""
    S = int(num_simulations)
    if check_every * depth > 256:
        raise ValueError("check_every * depth must stay within the 256 status entries the arena keeps")
    moves, st, ev = 0, None, None
    limit = max_moves if max_moves is not None else (done_upto + arena.G) * arena.T
    draining = False
    while moves < limit:
        while len(arena.pending) < (1 if draining else depth):
            play_moves(engine, arena, check_every, S, temperature, train)
            arena.mark()
        ev, last = arena.pending.pop(0)
        ev.synchronize()
        st = arena.status(last)
        moves += check_every
        engine.adapt_stream_prefix(st["max_consumed"])
        draining = st["next"] >= st["end"]
        if st["done_upto"] >= done_upto:
            break
        if st["alive"] == 0 and draining:
            break                       # nothing running, nothing queued
    return moves, st, ev


def game_seed_index(rank, games_per_rank):
    """Global index of the games rank `rank` plays in one iteration (muzero_amd.shard.game_range)."""
    from .shard import game_range
    return game_range(rank, games_per_rank)


class SelfPlay:
    """learning_cycle (self_play.py:144-281) with batched self-play: every iteration plays
    `games_per_iteration` (number_of_self_play_before_training) episodes on arena.G slots, saves
    them in the ReplayStore in game order and runs the training steps on the same GPU, same weights."""

    def __init__(self, muzero, engine, trainer, arena, store, num_simulations, games_per_iteration=None,
                 seed=0, rank=None, rolling=False):
        """Data-parallel use (one process per GPU, torch.distributed initialised): every rank plays
        its own `games_per_iteration` games (seeds offset by rank, no collective), keeps its own
        replay store, and Trainer.step sums the gradients of the per-rank batches with one
        all-reduce, so the replicas' weights stay identical.

        rolling=True: while the last games of an iteration finish, the slots they free already play
        the NEXT iteration's games (mzs_arena_enqueue), so the lock-step searches stay full — 90+ %
        of the slots busy instead of 62-69 %.  The price is stated: those early games make their
        first moves under the weights of before the training steps of the iteration they overlap
        (the reference plays every game of an iteration after the previous iteration's training,
        self_play.py:217-262; rolling=False keeps exactly that)."""
        from .shard import rank_world
        self.rank = rank_world()[0] if rank is None else int(rank)
        self.mz, self.eng, self.trainer, self.arena, self.store = muzero, engine, trainer, arena, store
        self.S = int(num_simulations)
        self.N = int(games_per_iteration) if games_per_iteration else arena.G
        self.seed = int(seed)
        self.block = 0
        self.episode = 0
        self.rolling = bool(rolling)
        self.queued = 0                 # iterations whose games have been handed to the arena (rolling)
        self._side = None
        self.n_blocks = arena.n_rows // self.N
        need = -(-(store.window_size + 1) // self.N) + (2 if rolling else 1)
        if self.n_blocks < need:
            raise ValueError(f"arena needs {need * self.N} rows for a window of {store.window_size} games "
                             f"played {self.N} per iteration (has {arena.n_rows})")

    @classmethod
    def from_config(cls, cfg, n_slots=None, games_per_iteration=None, window_size=None, seed=0, device=0,
                    weights=None, rolling=None):
        """Build engine, trainer, arena and store from a reference experiment config
        (config/experiment_*.json as muzero_cli.py:48-114 consumes it; muzero_amd.config.load_config).
        Every shipped config is CartPole-v1 + mlp_model.  The reference plays
        `number_of_self_play_before_training` games (1 or 10) per iteration; pass
        games_per_iteration / n_slots (or the optional "mi355x" section) to fill the GPU, and a
        window_size to match, since the config's replay window (500 games) is sized for that."""
        from .config import engine_id
        from .engine import SearchEngine
        from .model import Muzero
        from .train import Trainer
        if cfg["game"]["env"] != "CartPole-v1":
            raise NotImplementedError(f"arena environment {cfg['game']['env']!r}: only CartPole-v1 "
                                      "(include/mzs_env.h); other Gym environments go through the "
                                      "drop-in Monte_carlo_tree_search.run, stepped on the host")
        m, rb, mc, gp, lc = (cfg[k] for k in ("muzero", "replaybuffer", "monte_carlo_tree_search", "gameplay",
                                              "learning_cycle"))
        extra = cfg.get("mi355x", {})
        N = int(games_per_iteration or extra.get("games_per_iteration") or lc["number_of_self_play_before_training"])
        # 8192 slots by default: two blocks of the search kernel share a CU from 4097 games on (the kernels
        # compiled for two wavefronts per SIMD), measured 8.7e8 useful sims/s against 7.5e8 on 4096 slots
        G = min(int(n_slots or extra.get("concurrent_games", 8192)), N)
        W = int(window_size or extra.get("window_size") or rb["window_size"])
        S = int(lc["number_of_mcts_simulation"])
        mz = Muzero(model_structure=m["model_structure"], observation_space_dimensions=4,
                    action_space_dimensions=2, state_space_dimensions=m["state_space_dimensions"],
                    k_hypothetical_steps=m["k_hypothetical_steps"], learning_rate=m["learning_rate"],
                    optimizer=m["optimizer"], lr_scheduler=m["lr_scheduler"], loss_type=m["loss_type"],
                    num_of_epoch=m["num_of_epoch"], hidden_layer_dimensions=m["hidden_layer_dimensions"],
                    number_of_hidden_layer=m["number_of_hidden_layer"], bin_method=m["bin_method"],
                    bin_decomposition_number=m["bin_decomposition_number"])
        if weights is not None:
            mz.set_weights(weights)
        eng = SearchEngine(mz.dims, max_games=G, max_simulations=S, discount=mc["discount"],
                           root_exploration_fraction=mc["root_exploration_fraction"], pb_c_base=mc["pb_c_base"],
                           pb_c_init=mc["pb_c_init"], root_dirichlet_alpha=mc["root_dirichlet_alpha"],
                           device=device, engine=engine_id(cfg) if "mi355x" in cfg else _lib.ENGINE_FUSED)
        eng.set_weights(mz.weights_blob())
        mz._engine = eng
        trainer = Trainer(eng, m["k_hypothetical_steps"], rb["batch_size"], m["optimizer"], m["learning_rate"],
                          m["lr_scheduler"], m["num_of_epoch"], m["loss_type"])
        mz._trainer = trainer
        from .shard import rank_world
        if rank_world()[1] > 1:      # Trainer broadcast rank 0's weights into the handle: mz follows
            mz._blob_stale = True
            mz.weights_version += 1
        store = ReplayStore(W, rb["batch_size"], m["k_hypothetical_steps"], rb["td_steps"], rb["game_sampling"],
                            rb["position_sampling"], discount=gp["discount"])
        rolling = bool(extra.get("rolling", False) if rolling is None else rolling)
        n_blocks = -(-(W + 1) // N) + (2 if rolling else 1)
        arena = Arena(eng, G, n_blocks * N, gp["limit_of_game_play"])
        loop = cls(mz, eng, trainer, arena, store, S, games_per_iteration=N, seed=seed, rolling=rolling)
        loop.cfg = cfg
        return loop

    def run_config(self, number_of_iteration=None, verbose=False, model_tag_number=None,
                   checkpoint_directory="model_checkpoint"):
        """learning_cycle with the config's own iteration counts and temperature schedule."""
        lc = self.cfg["learning_cycle"]
        return self.learning_cycle(int(number_of_iteration or lc["number_of_iteration"]),
                                   lc["number_of_training_before_self_play"], lc["tempererature_type"], verbose,
                                   model_tag_number=model_tag_number, checkpoint_directory=checkpoint_directory)

    def _seeds(self):
        return (game_seed_index(self.rank, self.N) * 2654435761 + self.seed) & 0x7FFFFFFF

    def play(self, temperature, next_temperature=None):
        """One round of N games into the next block of rows; returns their total rewards.
        rolling: next_temperature = the temperature of the FOLLOWING iteration's games, queued now so
        that they fill the slots this iteration's tail frees (None: last iteration, the tail drains)."""
        a, N = self.arena, self.N
        row0 = self.block * N
        seeds = self._seeds()
        if not self.rolling:
            self.lockstep_moves = play_round(self.eng, a, seeds, self.episode, self.S, temperature, row0)
        else:
            k = self.episode
            if self.queued == 0:                      # very first iteration: nothing was queued ahead
                if a.n_rows % N:
                    raise ValueError("rolling rounds need n_rows to be a multiple of games_per_iteration")
                a.begin(0, seeds, 0)
                self.queued = 1
            if self.queued == k:
                # this round was never handed to the arena: the previous call was a last iteration
                # (next_temperature=None), and learning_cycle / play is now called again
                a.enqueue(seeds, k, temperature)
                self.queued += 1
            if next_temperature is not None and self.queued == k + 1:
                a.enqueue(seeds, k + 1, next_temperature)
                self.queued += 1
            # the step's temperature argument only reaches the games of the very first round
            # (mzs_arena_begin); every queued game carries its own
            moves, st, ev = play_until(self.eng, a, (k + 1) * N, self.S, temperature)
            self.lockstep_moves = moves
            if st is None or st["done_upto"] < (k + 1) * N:
                raise RuntimeError(f"rolling self-play: round {k} incomplete (games below row "
                                   f"{0 if st is None else st['done_upto']} of {(k + 1) * N} finished, "
                                   f"{self.queued} rounds queued): nothing saved")
            # The host now saves this round (device->host copies, NumPy bookkeeping).  Its rows were
            # complete when `ev` fired; the chunks play_until left in flight are already playing the
            # next round's games on the main stream.  Read the rows on a side stream that waits for
            # `ev` only, so that neither side waits for the other.
            dev = self.eng.device
            if self._side is None:
                self._side = torch.cuda.Stream(dev)
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                out = self._save_round(row0)
            self._side.synchronize()
            return out
        return self._save_round(row0)

    def _save_round(self, row0):
        a, N = self.arena, self.N
        lengths_d = a.arrays["length"][row0:row0 + N]
        lengths = lengths_d.cpu().numpy()
        longest, total = int(lengths.max()), int(lengths.sum())
        prio = None
        if self.store.uses_priority:
            # game after game without padding (a round of 32 768 CartPole games: 6 MB instead of the
            # 75 MB of a [N, longest] table), into pinned memory the store then keeps
            prio_d = a.priority_packed(row0, lengths_d, total, longest, self.store.td_steps, self.store.discount)
            pinned = torch.empty(total, dtype=torch.float64, pin_memory=True)
            pinned.copy_(prio_d, non_blocking=True)
            torch.cuda.current_stream(self.eng.device).synchronize()
            prio = pinned.numpy()
        self.store.save_games(np.arange(row0, row0 + N), lengths, priorities=prio)
        self.block = (self.block + 1) % self.n_blocks
        self.episode += 1
        rw = a.arrays["reward"][row0:row0 + N, :longest]
        mask = torch.arange(longest, device=rw.device)[None, :] < lengths_d[:, None]
        return (rw * mask).sum(1).cpu().numpy()                       # sum(g.rewards), self_play.py:240

    def train_step(self):
        """sample_batch -> make_batch -> Muzero.train -> update_value (self_play.py:259-262)."""
        st = self.store
        row_pos, ratio, game_pos = st.sample_batch()
        obs, actions, tv, tr, tp = self.arena.make_batch(row_pos, st.num_unroll, st.td_steps, st.discount)
        isr = None if np.size(ratio) == 1 else np.asarray(ratio, np.float32)
        prio = self.trainer.step(obs, actions, tv, tr, tp, isr)
        loss = self.trainer.loss.clone()            # the 3K-1 terms then loss_nn (muzero_model.py:734), on the device
        if st.uses_priority:
            st.update_value(prio.detach().cpu().numpy(), game_pos)
        if self.mz is not None:
            # keep the drop-in model object in step with the trainer it shares a handle with: step
            # count (lr schedule), store_loss, and a weight blob that save_model / a later
            # Monte_carlo_tree_search.run(model=mz) will fetch from the device
            self.mz.note_trained(self.trainer, loss.cpu().numpy())
        return loss[-1]

    def learning_cycle(self, number_of_iteration, number_of_training_before_self_play=1,
                       temperature_type="static_temperature", verbose=False, model_tag_number=None,
                       checkpoint_directory="model_checkpoint"):
        """model_tag_number: as the reference (self_play.py:248-255), the model is saved whenever an
        iteration's mean self-play reward is the best so far; None (default) writes nothing."""
        rewards, losses = [], []
        for ep in range(1, number_of_iteration + 1):
            t = temperature_scheduler(number_of_iteration + 1, ep, temperature_type)
            t_next = (temperature_scheduler(number_of_iteration + 1, ep + 1, temperature_type)
                      if ep < number_of_iteration else None)
            rewards.append(float(np.mean(self.play(t, t_next))))
            if self.mz is not None and model_tag_number is not None:
                self.mz.save_model(directory=checkpoint_directory, tag=model_tag_number,
                                   model_update_condition=bool(rewards[-1] == max(rewards)))
            cache = [self.train_step() for _ in range(number_of_training_before_self_play)]
            losses.append(float(torch.stack(cache).mean().item()))
            if verbose:
                print(f"EPOCH {ep} || selfplay reward: {rewards[-1]} || training loss: {losses[-1]}||")
        return rewards, losses
