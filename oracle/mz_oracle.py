"""ctypes front end of the CPU oracle (oracle/libmzoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by muzero_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Dims(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("n_actions", C.c_int32), ("state_dim", C.c_int32),
                ("hidden_dim", C.c_int32), ("n_mid_layers", C.c_int32), ("model_kind", C.c_int32),
                ("precision", C.c_int32)]


class SearchCfg(C.Structure):
    _fields_ = [("dims", Dims), ("num_simulations", C.c_int32), ("train", C.c_int32),
                ("n_threads", C.c_int32), ("_pad", C.c_int32), ("discount", C.c_double),
                ("exploration_fraction", C.c_double), ("pbc_table", C.c_void_p)]


class TableModel(C.Structure):
    _fields_ = [("root_policy", C.c_void_p), ("sim_reward", C.c_void_p),
                ("sim_value", C.c_void_p), ("sim_prior", C.c_void_p)]


_OUT_FIELDS = ["visits", "root_value", "root_prior", "consumed", "path_len", "sim_leaf",
               "node_visit", "node_vsum", "node_reward", "node_prior", "child_base", "minmax",
               "min_margin", "hidden"]


class SearchOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _OUT_FIELDS]


def build(force=False):
    """Path of the oracle library, (re)built when a source is newer.  MZO_ORACLE_SO names another build of
    the same sources (the sanitizer build of tests/test_oracle_golden.py::test_oracle_sanitized)."""
    override = os.environ.get("MZO_ORACLE_SO")
    if override:
        return override
    so = os.path.join(_HERE, "libmzoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mz_oracle.c", "mz_train_oracle.c", "mz_selfplay_oracle.c",
                                             "mz_vision_oracle.c", "mz_oracle.h", "../include/mzs_env.h",
                                             "../include/mzs_numeric.h", "../include/mzs_layout.h")]
    stale = os.path.exists(so) and all(os.path.exists(f) for f in srcs) and \
        os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs)
    if force or stale or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "libmzoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.mzo_search.restype = C.c_int
        _LIB.mzo_support_to_scalar.restype = C.c_float
        _LIB.mzo_support_to_scalar.argtypes = [C.c_void_p, C.c_int32]
        _LIB.mzo_expf.restype = C.c_float
        _LIB.mzo_expf.argtypes = [C.c_float]
        _LIB.mzo_elu.restype = C.c_float
        _LIB.mzo_elu.argtypes = [C.c_float]
        _LIB.mzo_cr_log.restype = C.c_double
        _LIB.mzo_cr_log.argtypes = [C.c_double]
        _LIB.mzo_cr_pow.restype = C.c_double
        _LIB.mzo_cr_pow.argtypes = [C.c_double, C.c_double]
        _LIB.mzo_stream.restype = C.c_int
    return _LIB


def dims(obs_dim, A, S_dim, H, L, kind=0, precision=0):
    """kind: 0 = 'mlp_model', 1 = 'lstm_model'; precision: 0 = fp32, 1 = bf16 (include/mzs_layout.h)."""
    return Dims(int(obs_dim), int(A), int(S_dim), int(H), int(L), int(kind), int(precision))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def dot4_bf16(a16, b16):
    """a16, b16: uint16 [n, 4] bf16 bit patterns -> float32 [n] (mzs_dot4_bf16)."""
    a = np.ascontiguousarray(a16, np.uint16).reshape(-1, 4)
    b = np.ascontiguousarray(b16, np.uint16).reshape(-1, 4)
    out = np.empty(len(a), np.float32)
    lib().mzo_dot4_bf16(_p(a), _p(b), C.c_int32(len(a)), _p(out))
    return out


def round_bf16(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().mzo_round_bf16(_p(x), C.c_int32(x.size), _p(out))
    return out


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def streams(seeds, n_actions, num_simulations, alpha=0.25, train=True, lo=1e-7, hi=2e-7, n_uniform=None):
    """mz_stream_oracle.c: RandomState(seed).dirichlet([alpha] * A) then .uniform(lo, hi, n) per seed, with
    log / pow from include/mzs_crmath.h: (dirichlet [G, A] or zeros, noise [G, n])."""
    seeds = np.ascontiguousarray(seeds, np.int64).ravel()
    A, S = int(n_actions), int(num_simulations)
    n = A * S * (S + 1) // 2 if n_uniform is None else int(n_uniform)
    d = np.zeros((len(seeds), A), np.float64)
    u = np.zeros((len(seeds), max(n, 1)), np.float64)
    tr = int(bool(train) and S > 0)
    for g, seed in enumerate(seeds):
        rc = lib().mzo_stream(C.c_int64(int(seed)), C.c_int(A), C.c_double(alpha), C.c_int(tr), C.c_double(lo),
                              C.c_double(hi), C.c_int64(n), _p(d[g]), _p(u[g]))
        if rc != 0:
            raise ValueError("mzo_stream: alpha must be <= 1")
    return d, u


VISION_HIDDEN = 3 * 7 * 7


def hidden_floats(d):
    """floats of one hidden state: S_dim, or the 3 x 7 x 7 feature map of the 'vision_model' (kind 2)"""
    return VISION_HIDDEN if d.model_kind == 2 else d.state_dim


def representation(d, blob, obs):
    obs = _f32(obs)
    B = obs.shape[0]
    hid = np.empty((B, hidden_floats(d)), np.float32)
    lib().mzo_representation(C.byref(d), _p(_f32(blob)), _p(obs), C.c_int32(B), _p(hid))
    return hid


def dynamics(d, blob, hidden, action):
    hidden = _f32(hidden)
    action = np.ascontiguousarray(action, dtype=np.int32)
    B = hidden.shape[0]
    rew = np.empty(B, np.float32)
    nh = np.empty((B, hidden_floats(d)), np.float32)
    lib().mzo_dynamics(C.byref(d), _p(_f32(blob)), _p(hidden), _p(action), C.c_int32(B), _p(rew), _p(nh))
    return rew, nh


def prediction(d, blob, hidden):
    hidden = _f32(hidden)
    B = hidden.shape[0]
    pol = np.empty((B, d.n_actions), np.float32)
    val = np.empty(B, np.float32)
    lib().mzo_prediction(C.byref(d), _p(_f32(blob)), _p(hidden), C.c_int32(B), _p(pol), _p(val))
    return pol, val


def initial_inference(d, blob, obs):
    h = representation(d, blob, obs)
    p, v = prediction(d, blob, h)
    return h, p, v


def recurrent_inference(d, blob, hidden, action):
    r, nh = dynamics(d, blob, hidden, action)
    p, v = prediction(d, blob, nh)
    return r, nh, p, v


def scale_to_bound(x):
    x = _f32(x)
    y = np.empty_like(x)
    for i in range(x.shape[0]):
        lib().mzo_scale_to_bound(_p(x[i]), C.c_int32(x.shape[1]), _p(y[i]))
    return y


def support_to_scalar(logits):
    logits = _f32(logits)
    return np.array([lib().mzo_support_to_scalar(_p(logits[i]), logits.shape[1])
                     for i in range(logits.shape[0])], np.float32)


def search(d, blob, S, gamma, frac, pbc_table, obs=None, dirichlet=None, noise=None, train=True,
           table=None, n_threads=1, want=("visits", "root_value", "root_prior", "consumed",
                                          "path_len", "sim_leaf", "min_margin")):
    """Run G searches. ``noise`` is [G, stride] float64 tie-break draws, ``dirichlet``
    [G, A] float64. ``table`` = dict(root_policy, sim_reward, sim_value, sim_prior)
    switches to the injected-model (L1) mode. Returns a dict of numpy arrays."""
    A = d.n_actions
    if table is not None:
        G = np.asarray(table["root_policy"]).shape[0]
    else:
        obs = _f32(obs)
        G = obs.shape[0]
    N = 1 + A * (S + 1)
    pbc = np.ascontiguousarray(pbc_table, dtype=np.float64)
    assert pbc.shape[0] >= max(S, 1)
    if noise is not None:
        noise = np.ascontiguousarray(noise, dtype=np.float64)
        assert noise.ndim == 2 and noise.shape[0] == G
    if dirichlet is not None:
        dirichlet = np.ascontiguousarray(dirichlet, dtype=np.float64)
    cfg = SearchCfg(d, int(S), int(bool(train)), int(n_threads), 0, float(gamma), float(frac),
                    _p(pbc).value)
    shapes = dict(visits=((G, A), np.int32), root_value=((G,), np.float32),
                  root_prior=((G, A), np.float64), consumed=((G,), np.int64),
                  path_len=((G,), np.int64), sim_leaf=((G, S), np.int32),
                  node_visit=((G, N), np.int32), node_vsum=((G, N), np.float32),
                  node_reward=((G, N), np.float32), node_prior=((G, N), np.float32),
                  child_base=((G, N), np.int32), minmax=((G, 2), np.float32),
                  min_margin=((G,), np.float64), hidden=((G, S + 1, hidden_floats(d)), np.float32))
    res = {k: np.zeros(*shapes[k]) for k in want}
    out = SearchOut(*[(_p(res[k]).value if k in res else None) for k in _OUT_FIELDS])
    tbl = None
    keep = []
    if table is not None:
        keep = [_f32(table[k]) for k in ("root_policy", "sim_reward", "sim_value", "sim_prior")]
        tbl = TableModel(*[_p(a).value for a in keep])
    blob_c = None if blob is None else _f32(blob)
    rc = lib().mzo_search(C.byref(cfg), _p(blob_c), C.byref(tbl) if tbl is not None else None,
                          _p(obs), _p(dirichlet), _p(noise),
                          C.c_int64(noise.shape[1] if noise is not None else 0), C.c_int32(G),
                          C.byref(out))
    if rc != 0:
        raise RuntimeError(f"mzo_search failed: {rc}")
    return res


# ------------------------------------------------------------------ training step
class TrainBatch(C.Structure):
    _fields_ = [("B", C.c_int32), ("K", C.c_int32), ("obs", C.c_void_p), ("actions", C.c_void_p),
                ("target_value", C.c_void_p), ("target_reward", C.c_void_p),
                ("target_policy", C.c_void_p), ("is_ratio", C.c_void_p), ("inv_batch", C.c_float)]


def blob_size(d):
    from muzero_amd.model import blob_layout
    return sum(int(np.prod(shp)) for _, shp in
               blob_layout(d.obs_dim, d.n_actions, d.state_dim, d.hidden_dim, d.n_mid_layers, d.model_kind))


# loss_type -> (criterion of the value term, of the reward term), muzero_model.py:350-403; MZO_LOSS_*
LOSS_MODES = {"general": (0, 0), "general_kkc": (1, 1), "game": (2, 3), "game_mmc": (2, 2)}


def train_grad(d, blob, obs, actions, target_value, target_reward, target_policy, is_ratio=None,
               global_batch=None, loss_type="general"):
    """Forward + losses + backward of one training step on a batch (shard).  Returns
    dict(grad [P], loss_terms [3K-1], total_loss, priority [K, B])."""
    lib().mzo_train_set_loss(*LOSS_MODES[loss_type])
    obs, actions = _f32(obs), _f32(actions)
    tv, tr, tp = _f32(target_value), _f32(target_reward), _f32(target_policy)
    K, B = tv.shape
    isr = None if is_ratio is None else _f32(is_ratio)
    bt = TrainBatch(B, K, _p(obs).value, _p(actions).value, _p(tv).value, _p(tr).value, _p(tp).value,
                    None if isr is None else _p(isr).value, 1.0 / float(global_batch or B))
    blob = _f32(blob)
    n = blob_size(d)
    assert blob.size == n
    grad = np.zeros(n, np.float32)
    terms = np.zeros(3 * K - 1, np.float32)
    total = C.c_double(0.0)
    prio = np.zeros((K, B), np.float32)
    f = lib().mzo_train_grad
    f.restype = C.c_int
    rc = f(C.byref(d), _p(blob), C.byref(bt), _p(grad), _p(terms), C.byref(total), _p(prio))
    if rc != 0:
        raise RuntimeError(f"mzo_train_grad failed: {rc}")
    return dict(grad=grad, loss_terms=terms, total_loss=total.value, priority=prio)


def set_momentum(momentum=0.9):
    """Adam's beta1 / SGD's momentum for the following steps (OneCycleLR cycles it)."""
    f = lib().mzo_set_momentum
    f.restype = None
    f(C.c_double(momentum))


def adam_step(w, m, v, g, step, lr):
    f = lib().mzo_adam_step
    f.restype = None
    f(_p(w), _p(m), _p(v), _p(_f32(g)), C.c_size_t(w.size), C.c_int(step), C.c_double(lr))


def sgd_step(w, buf, g, step, lr):
    f = lib().mzo_sgd_step
    f.restype = None
    f(_p(w), _p(buf), _p(_f32(g)), C.c_size_t(w.size), C.c_int(step), C.c_double(lr))


# ------------------------------------------------------------------ self-play bookkeeping
def store_policy(visits, prior):
    v = np.ascontiguousarray(visits, np.int32)
    pr = np.ascontiguousarray(prior, np.float64)
    out = np.zeros(v.size, np.float32)
    lib().mzo_store_policy(C.c_int32(v.size), _p(v), _p(pr), _p(out))
    return out


def policy_action(policy, temperature, u):
    v = np.ascontiguousarray(policy, np.float64)
    f = lib().mzo_policy_action
    f.restype = C.c_int32
    return int(f(C.c_int32(v.size), _p(v), C.c_double(temperature), C.c_double(u)))


def cartpole_step(state, action):
    s = np.array(state, np.float64)
    term = C.c_int32(0)
    f = lib().mzo_cartpole_step
    f.restype = C.c_double
    r = f(_p(s), C.c_int32(action), C.byref(term))
    return s, float(r), bool(term.value)


class _Arena(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("G", "n_rows", "T", "A", "episode", "next_row", "row_end", "done_upto")] + \
               [(n, C.c_void_p) for n in ("state", "alive", "row", "stream_seed", "obs", "row_seed", "length",
                                          "traj_obs", "traj_action", "traj_reward", "traj_root", "traj_policy",
                                          "row_episode", "row_temp", "vrow")]


class ArenaOracle:
    """G slots / n_rows trajectory rows on the CPU: the arrays the device arena keeps."""

    def __init__(self, G, T, A=2, obs_dim=4, n_rows=None):
        R = G if n_rows is None else n_rows
        self.G, self.T, self.A, self.obs_dim, self.n_rows = G, T, A, obs_dim, R
        self.state = np.zeros((G, 4), np.float64)
        self.alive = np.zeros(G, np.int32)
        self.row = np.zeros(G, np.int32)
        self.stream_seed = np.zeros(G, np.int64)
        self.obs = np.zeros((G, obs_dim), np.float32)
        self.row_seed = np.zeros(R, np.int64)
        self.length = np.zeros(R, np.int32)
        self.traj_obs = np.zeros((R, T, obs_dim), np.float32)
        self.traj_action = np.zeros((R, T), np.int32)
        self.traj_reward = np.zeros((R, T), np.float32)
        self.traj_root = np.zeros((R, T), np.float32)
        self.traj_policy = np.zeros((R, T, A), np.float32)
        self.row_episode = np.zeros(R, np.int32)
        self.row_temp = np.full(R, np.nan, np.float64)
        self.vrow = np.zeros(G, np.int32)
        self.c = _Arena(G, R, T, A, 0, 0, 0, 0, *(a.ctypes.data for a in (
            self.state, self.alive, self.row, self.stream_seed, self.obs, self.row_seed, self.length,
            self.traj_obs, self.traj_action, self.traj_reward, self.traj_root, self.traj_policy,
            self.row_episode, self.row_temp, self.vrow)))

    def begin(self, seeds, episode, row0=0):
        seeds = np.ascontiguousarray(seeds, np.int64)
        assert row0 + len(seeds) <= self.n_rows
        lib().mzo_arena_begin(C.byref(self.c), C.c_int32(row0), C.c_int32(len(seeds)), _p(seeds), C.c_int32(episode))
        return self.obs

    def enqueue(self, seeds, episode, temperature=float("nan")):
        """More games behind the queue's end (mzs_arena_enqueue); rows wrap modulo n_rows."""
        seeds = np.ascontiguousarray(seeds, np.int64)
        lib().mzo_arena_enqueue(C.byref(self.c), C.c_int32(len(seeds)), _p(seeds), C.c_int32(episode),
                                C.c_double(temperature))

    @property
    def done_upto(self):
        return int(self.c.done_upto)

    @property
    def queue(self):
        return int(self.c.next_row), int(self.c.row_end)

    def step(self, visits, root_value, root_prior, temperature, from_prior=False):
        v = np.ascontiguousarray(visits, np.int32)
        rv = _f32(root_value)
        rp = np.ascontiguousarray(root_prior, np.float64)
        f = lib().mzo_arena_step
        f.restype = C.c_int32
        return f(C.byref(self.c), _p(v), _p(rv), _p(rp), C.c_double(temperature), C.c_int32(int(from_prior)))


def make_batch(length, traj_obs, traj_action, traj_reward, traj_root, traj_policy, row_pos, K, td, disc_pow):
    T, A, obs_dim = traj_obs.shape[1], traj_policy.shape[2], traj_obs.shape[2]
    rp = np.ascontiguousarray(row_pos, np.int32)
    B = rp.shape[0]
    dp = np.ascontiguousarray(disc_pow, np.float64)
    obs = np.zeros((B, obs_dim), np.float32)
    actions = np.zeros((K, B, A), np.float32)
    tv, tr = np.zeros((K, B), np.float32), np.zeros((K, B), np.float32)
    tp = np.zeros((K, B, A), np.float32)
    lib().mzo_make_batch(C.c_int32(T), C.c_int32(A), C.c_int32(obs_dim), _p(np.ascontiguousarray(length, np.int32)),
                         _p(_f32(traj_obs)), _p(np.ascontiguousarray(traj_action, np.int32)), _p(_f32(traj_reward)),
                         _p(_f32(traj_root)), _p(_f32(traj_policy)), C.c_int32(B), _p(rp), C.c_int32(K),
                         C.c_int32(td), _p(dp), _p(obs), _p(actions), _p(tv), _p(tr), _p(tp))
    return obs, actions, tv, tr, tp


def make_priority(length, traj_reward, traj_root, td, disc_pow):
    rw, rv = _f32(traj_reward), _f32(traj_root)
    n, T = rv.shape
    out = np.zeros((n, T), np.float64)
    lib().mzo_make_priority(C.c_int32(n), C.c_int32(T), _p(np.ascontiguousarray(length, np.int32)), _p(rw), _p(rv),
                            C.c_int32(td), _p(np.ascontiguousarray(disc_pow, np.float64)), _p(out))
    return out
