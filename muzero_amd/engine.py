"""SearchEngine — thin owner of one mzs_handle (one GPU, one stream).

Device memory, streams and host<->device copies come from PyTorch-ROCm; all
compute goes through the C-ABI of libmzsearch.so (include/mzsearch.h).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, rng


class _DevView:
    """Expose a raw device pointer of the handle to torch (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


class SearchEngine:
    """Batched Monte_carlo_tree_search over `max_games` concurrent games.

    dims = (obs_dim, n_actions, state_dim, hidden_dim, n_mid_layers[, model_kind[, precision]]) — the
    shapes the reference passes to Muzero(...) (muzero_model.py:16-35; model_kind 0 = 'mlp_model', 1 =
    'lstm_model'; precision 0 = fp32, 1 = bf16 (fast), 2 = bf16x3 (compensated)); discount / pb_c_* /
    root_exploration_fraction are the Monte_carlo_tree_search ctor arguments
    (monte_carlo_tree_search.py:38).
    """

    def __init__(self, dims, max_games, max_simulations, discount=0.95,
                 root_exploration_fraction=0.25, pb_c_base=19652, pb_c_init=1.25,
                 root_dirichlet_alpha=0.25, device=0, engine=_lib.ENGINE_FUSED):
        if not torch.cuda.is_available():
            raise _lib.MzsError("no GPU visible: the search path has no CPU fallback")
        self.lib = _lib.load()
        self.dims = tuple(int(x) for x in dims)
        self.obs_dim, self.A, self.S_dim, self.H, self.L = self.dims[:5]
        self.model_kind = self.dims[5] if len(self.dims) > 5 else _lib.MODEL_MLP
        self.precision = self.dims[6] if len(self.dims) > 6 else _lib.PRECISION_F32
        vision = self.model_kind == _lib.MODEL_VISION
        # one observation / one hidden state as the kernels see them: the vision model takes a
        # (3, 98, 98) image and keeps a 3 x 7 x 7 feature map (147 floats) per node
        self.obs_shape = (3, self.obs_dim, self.obs_dim) if vision else (self.obs_dim,)
        self.hidden_floats = 147 if vision else self.S_dim
        self.max_games, self.max_simulations = int(max_games), int(max_simulations)
        self.discount = float(discount)
        self.frac = float(root_exploration_fraction)
        self.alpha = float(root_dirichlet_alpha)
        self.pb_c_base, self.pb_c_init = pb_c_base, pb_c_init
        self.device = torch.device("cuda", int(device))
        self.pbc = rng.pbc_table(self.max_simulations, pb_c_base, pb_c_init)
        cfg = _lib.Config(_lib.Dims(*self.dims), self.max_games, self.max_simulations,
                          int(device), int(engine), self.discount, self.frac,
                          self.pbc.ctypes.data_as(C.c_void_p))
        h = C.c_void_p()
        _lib.check(self.lib.mzs_create(C.byref(cfg), C.byref(h)), "mzs_create")
        self.h = h
        self.n_games = 0
        self.num_simulations = 0
        self.stream_prefix = 0          # search_streams: draws written before the search (0: library default)
        self.num_simulations = 0
        self.has_weights = False
        if self.precision != _lib.PRECISION_F32 and int(engine) == _lib.ENGINE_FUSED:
            name = self.search_kernel_name()
            if name.startswith("k_search_fused"):
                import warnings
                warnings.warn(f"bf16 search of shape {self.dims[:6]} runs on {name}: no four-games-per-wave kernel "
                              "is instantiated for it, and on the any-shape kernel bf16 costs about twice the fp32 "
                              "search (DESIGN.md section 4); fp32 is the fast mode for this shape", stacklevel=2)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mzs_destroy(self.h)
            self.h = None

    __del__ = close

    # ------------------------------------------------------------------ weights
    def weight_count(self):
        d = _lib.Dims(*self.dims)
        return int(self.lib.mzs_weight_count(C.byref(d)))

    def set_weights(self, blob):
        """blob: packed fp32 weights (include/mzs_layout.h), numpy array or torch tensor."""
        if isinstance(blob, torch.Tensor):
            t = blob.detach().to(torch.float32).contiguous()
            ptr, n = C.c_void_p(t.data_ptr()), t.numel()
            if t.is_cuda:
                torch.cuda.synchronize(t.device)
        else:
            t = np.ascontiguousarray(blob, dtype=np.float32)
            ptr, n = t.ctypes.data_as(C.c_void_p), t.size
        _lib.check(self.lib.mzs_set_weights(self.h, ptr, n), "mzs_set_weights")
        self.has_weights = True

    # ------------------------------------------------------------------- search
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype):
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def search(self, obs, dirichlet, noise, num_simulations, train=True):
        """Enqueue one search per row of `obs` on the current stream (no host sync).
        obs [G, obs_dim] f32, dirichlet [G, A] f64 or None, noise [G, stride] f64."""
        obs = self._dev(obs, torch.float32)
        dirichlet = self._dev(dirichlet, torch.float64)
        noise = self._dev(noise, torch.float64)
        G = obs.shape[0]
        assert tuple(obs.shape) == (G,) + self.obs_shape
        stride = 0 if noise is None else noise.shape[1]
        if noise is not None:
            assert noise.shape[0] == G
        if dirichlet is not None:
            assert dirichlet.shape == (G, self.A)
        self._keep = (obs, dirichlet, noise)
        _lib.check(self.lib.mzs_search(self.h, G, int(num_simulations), int(bool(train)),
                                       _ptr(obs), _ptr(dirichlet), _ptr(noise), stride,
                                       self._stream()), "mzs_search")
        self.n_games, self.num_simulations = G, int(num_simulations)

    def search_kernel_name(self, num_simulations=None):
        """The kernel mzs_search dispatches to for this handle (mzs_search_kernel_name)."""
        S = self.max_simulations if num_simulations is None else int(num_simulations)
        buf = C.create_string_buffer(128)
        _lib.check(self.lib.mzs_search_kernel_name(self.h, S, buf, 128), "mzs_search_kernel_name")
        return buf.value.decode()

    def fill_streams(self, seeds, num_simulations, train=True, lo=1e-7, hi=2e-7):
        """Per-game streams generated on the device (mzs_fill_streams): returns device tensors
        (dirichlet [G, A] f64 or None, noise [G, max_draws] f64) equal to
        rng.noise_streams(seeds, ...) — uniforms bit for bit, Dirichlet to an ulp or two."""
        seeds = self._dev(np.asarray(seeds, dtype=np.int64) if not isinstance(seeds, torch.Tensor) else seeds,
                          torch.int64)
        G, S = seeds.shape[0], int(num_simulations)
        stride = max(1, rng.max_draws(self.A, S))
        do_dir = bool(train) and S > 0
        dirichlet = torch.empty((G, self.A), dtype=torch.float64, device=self.device) if do_dir else None
        noise = torch.empty((G, stride), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mzs_fill_streams(self.h, G, S, int(bool(train)), _ptr(seeds), self.alpha,
                                             float(lo), float(hi), _ptr(dirichlet), _ptr(noise), stride,
                                             self._stream()), "mzs_fill_streams")
        self._keep_seeds = seeds
        return dirichlet, noise

    def stream_scratch(self, n_games, num_simulations):
        """(dirichlet [G, A], noise [G, max_draws]) f64 device scratch of search_streams, reused move
        after move."""
        stride = max(1, rng.max_draws(self.A, int(num_simulations)))
        key = (int(n_games), stride)
        if getattr(self, "_stream_scratch_key", None) != key:
            self._stream_scratch = (torch.empty((key[0], self.A), dtype=torch.float64, device=self.device),
                                    torch.empty(key, dtype=torch.float64, device=self.device))
            self._stream_scratch_key = key
        return self._stream_scratch

    def adapt_stream_prefix(self, used=None):
        """Host-side tuning of search_streams' prefix.  A redo pass costs a whole search however few
        games need it, so the prefix should cover every game of a move almost always: 1.5 x the
        largest consumption of the last search, in whole refreshes of the generator state (312
        draws), never below the library default.  Purely a speed knob: any prefix gives the same
        results.  used: that largest consumption if the caller already has it (Arena.status);
        None reads it back from the device (a host sync: call it where the loop already waits)."""
        if self.n_games == 0 or self.num_simulations == 0:
            return self.stream_prefix
        if used is None:
            used = int(self.results()["consumed"][: self.n_games].max().item())
        want = -(-int(1.5 * used) // 312) * 312
        self.stream_prefix = max(want, 9 * self.A * self.num_simulations)
        return self.stream_prefix

    def search_streams(self, obs, seeds, num_simulations, train=True, lo=1e-7, hi=2e-7, prefix_draws=None):
        """fill_streams + search as one call that writes only a prefix of every game's stream and
        redoes, on the device, the few searches that ask for more (mzs_search_streams).  Same results
        as search(obs, *fill_streams(seeds, ...)), bit for bit.  Returns the scratch tensors
        (dirichlet, noise): only the prefix of a noise row is defined unless its game was redone.
        prefix_draws None: self.stream_prefix (0 = the library default, see adapt_stream_prefix)."""
        if prefix_draws is None:
            prefix_draws = self.stream_prefix
        obs = self._dev(obs, torch.float32)
        seeds = self._dev(np.asarray(seeds, dtype=np.int64) if not isinstance(seeds, torch.Tensor) else seeds,
                          torch.int64)
        G, S = obs.shape[0], int(num_simulations)
        assert tuple(obs.shape) == (G,) + self.obs_shape and seeds.shape == (G,)
        dirichlet, noise = self.stream_scratch(G, S)
        stride = noise.shape[1]
        self._keep = (obs, seeds)
        _lib.check(self.lib.mzs_search_streams(self.h, G, S, int(bool(train)), _ptr(obs), _ptr(seeds), self.alpha,
                                               float(lo), float(hi), int(prefix_draws), _ptr(dirichlet), _ptr(noise),
                                               stride, self._stream()), "mzs_search_streams")
        self.n_games, self.num_simulations = G, S
        return dirichlet, noise

    def search_injected(self, table, dirichlet, noise, num_simulations, train=True):
        """L1 mode: `table` = dict(root_policy [G,A], sim_reward [G,S], sim_value [G,S],
        sim_prior [G,S,A]) holds the model outputs to use instead of computing them."""
        tb = {k: self._dev(table[k], torch.float32)
              for k in ("root_policy", "sim_reward", "sim_value", "sim_prior")}
        dirichlet = self._dev(dirichlet, torch.float64)
        noise = self._dev(noise, torch.float64)
        G = tb["root_policy"].shape[0]
        stride = 0 if noise is None else noise.shape[1]
        self._keep = (tb, dirichlet, noise)
        _lib.check(self.lib.mzs_search_injected(
            self.h, G, int(num_simulations), int(bool(train)), _ptr(tb["root_policy"]),
            _ptr(tb["sim_reward"]), _ptr(tb["sim_value"]), _ptr(tb["sim_prior"]),
            _ptr(dirichlet), _ptr(noise), stride, self._stream()), "mzs_search_injected")
        self.n_games, self.num_simulations = G, int(num_simulations)

    def results(self):
        """Device tensors: visits i32 [G,A], root_value f32 [G], root_prior f64 [G,A],
        consumed i64 [G], path_len i64 [G] (enqueued copies; caller synchronises)."""
        G, A, dev = self.n_games, self.A, self.device
        out = dict(visits=torch.empty((G, A), dtype=torch.int32, device=dev),
                   root_value=torch.empty(G, dtype=torch.float32, device=dev),
                   root_prior=torch.empty((G, A), dtype=torch.float64, device=dev),
                   consumed=torch.empty(G, dtype=torch.int64, device=dev),
                   path_len=torch.empty(G, dtype=torch.int64, device=dev))
        _lib.check(self.lib.mzs_results(self.h, G, _ptr(out["visits"]), _ptr(out["root_value"]),
                                        _ptr(out["root_prior"]), _ptr(out["consumed"]),
                                        _ptr(out["path_len"]), self._stream()), "mzs_results")
        return out

    def results_view(self):
        """Zero-copy torch views of the handle's result arrays (same keys as results()); valid
        until the next search on this engine."""
        G, A = self.n_games, self.A
        spec = dict(visits=(_lib.TREE_ROOT_VISITS, (G, A), "<i4"),
                    root_value=(_lib.TREE_ROOT_VALUE, (G,), "<f4"),
                    root_prior=(_lib.TREE_ROOT_PRIOR, (G, A), "<f8"),
                    consumed=(_lib.TREE_CONSUMED, (G,), "<i8"),
                    path_len=(_lib.TREE_PATH_LEN, (G,), "<i8"))
        if not hasattr(self, "_views") or self._views[0] != G:
            out = {}
            for k, (which, shape, ts) in spec.items():
                p = self.lib.mzs_tree_array(self.h, which)
                out[k] = torch.as_tensor(_DevView(p, shape, ts), device=self.device)
            self._views = (G, out)
        return self._views[1]

    def set_result_sink(self, visits=None, root_value=None):
        """mzs_set_result_sink: every later search also stores its visit counts / root values in
        these pinned host tensors (int32 [max_games, A], float32 [max_games]); they are complete
        when the search's stream work is (synchronise or wait on an event before reading).
        None, None switches it off."""
        if visits is not None:
            ok = (visits.is_pinned() and root_value.is_pinned() and visits.is_contiguous() and
                  root_value.is_contiguous() and visits.dtype == torch.int32 and
                  root_value.dtype == torch.float32 and visits.numel() >= self.max_games * self.A and
                  root_value.numel() >= self.max_games)
            if not ok:
                raise ValueError("result sink: pinned contiguous int32 [max_games, A] and float32 [max_games]")
        self._sink = (visits, root_value)   # keep the memory alive while the handle points at it
        _lib.check(self.lib.mzs_set_result_sink(self.h, None if visits is None else visits.data_ptr(),
                                                None if root_value is None else root_value.data_ptr()),
                   "mzs_set_result_sink")

    def results_packed_view(self):
        """One int32 view [max_games * (A + 1)] over visit counts (max_games x A) followed by the
        root values' bits (max_games): a single device->host copy reads both."""
        n = self.max_games * (self.A + 1)
        p = self.lib.mzs_tree_array(self.h, _lib.TREE_ROOT_VISITS)
        return torch.as_tensor(_DevView(p, (n,), "<i4"), device=self.device)

    def results_host(self):
        out = self.results()
        torch.cuda.synchronize(self.device)
        return {k: v.cpu().numpy() for k, v in out.items()}

    def tree(self):
        """Zero-copy torch views of the handle's SoA tree (valid until the next search)."""
        G, Smax = self.max_games, self.max_simulations
        st = int(self.lib.mzs_tree_stride(self.h))
        hs = int(self.lib.mzs_hidden_stride(self.h))
        spec = dict(visit=(_lib.TREE_VISIT, (G, st), "<i4"),
                    value_sum=(_lib.TREE_VALUE_SUM, (G, st), "<f4"),
                    prior=(_lib.TREE_PRIOR, (G, st), "<f4"),
                    reward=(_lib.TREE_REWARD, (G, st), "<f4"),
                    child_base=(_lib.TREE_CHILD_BASE, (G, st), "<i4"),
                    sim_leaf=(_lib.TREE_SIM_LEAF, (G, max(Smax, 1)), "<i4"),
                    minmax=(_lib.TREE_MINMAX, (G, 2), "<f4"),
                    hidden=(_lib.TREE_HIDDEN, (G, Smax + 1, hs), "<f4"),
                    status=(_lib.TREE_STATUS, (G,), "<i4"))
        out = {}
        for k, (which, shape, ts) in spec.items():
            p = self.lib.mzs_tree_array(self.h, which)
            out[k] = torch.as_tensor(_DevView(p, shape, ts), device=self.device)
        return out

    def tree_host(self):
        torch.cuda.synchronize(self.device)
        n = 1 + self.A * (self.num_simulations + 1)
        t = {k: v[: self.n_games].cpu().numpy() for k, v in self.tree().items()}
        for k in ("visit", "value_sum", "prior", "reward", "child_base"):
            t[k] = t[k][:, :n]
        t["sim_leaf"] = t["sim_leaf"][:, : self.num_simulations]
        t["hidden"] = t["hidden"][:, : self.num_simulations + 1, : self.hidden_floats]
        return t

    def run_batch(self, observations, seeds=None, num_simulations=None, train=True,
                  dirichlet=None, noise=None):
        """Convenience wrapper: draw per-game streams from `seeds` (RandomState(seed):
        Dirichlet, then the tie-break uniforms — what the reference's global stream yields
        after np.random.seed(seed)), search, and return host arrays."""
        S = self.max_simulations if num_simulations is None else int(num_simulations)
        if noise is None and 0.0 < self.alpha <= 1.0:
            self.search_streams(observations, seeds, S, train)             # streams made on the device
            return self.results_host()
        elif noise is None:
            dirichlet, noise = rng.noise_streams(seeds, self.A, S, self.alpha, train)
        self.search(observations, dirichlet if (train and S > 0) else None,
                    noise if S > 0 else None, S, train)
        return self.results_host()

    # ---------------------------------------------------------------- inference
    def initial_inference(self, obs, want_prediction=True):
        obs = self._dev(obs, torch.float32)
        B = obs.shape[0]
        hid = torch.empty((B, self.hidden_floats), dtype=torch.float32, device=self.device)
        pol = torch.empty((B, self.A), dtype=torch.float32, device=self.device) if want_prediction else None
        val = torch.empty(B, dtype=torch.float32, device=self.device) if want_prediction else None
        _lib.check(self.lib.mzs_initial_inference(self.h, B, _ptr(obs), _ptr(hid), _ptr(pol),
                                                  _ptr(val), self._stream()), "mzs_initial_inference")
        return hid, pol, val

    def recurrent_inference(self, hidden, action, want_prediction=True):
        hidden = self._dev(hidden, torch.float32)
        action = self._dev(action, torch.int32)
        B = hidden.shape[0]
        rew = torch.empty(B, dtype=torch.float32, device=self.device)
        nh = torch.empty((B, self.hidden_floats), dtype=torch.float32, device=self.device)
        pol = torch.empty((B, self.A), dtype=torch.float32, device=self.device) if want_prediction else None
        val = torch.empty(B, dtype=torch.float32, device=self.device) if want_prediction else None
        _lib.check(self.lib.mzs_recurrent_inference(self.h, B, _ptr(hidden), _ptr(action), _ptr(rew),
                                                    _ptr(nh), _ptr(pol), _ptr(val), self._stream()),
                   "mzs_recurrent_inference")
        return rew, nh, pol, val

    def prediction_inference(self, hidden):
        hidden = self._dev(hidden, torch.float32)
        B = hidden.shape[0]
        pol = torch.empty((B, self.A), dtype=torch.float32, device=self.device)
        val = torch.empty(B, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mzs_prediction_inference(self.h, B, _ptr(hidden), _ptr(pol), _ptr(val),
                                                     self._stream()), "mzs_prediction_inference")
        return pol, val
