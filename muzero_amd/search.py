"""`Monte_carlo_tree_search` — the reference's search class (monte_carlo_tree_search.py:37)
on top of the batched HIP engine.

Drop-in surface: constructor / `reset` arguments and attributes
(monte_carlo_tree_search.py:38-86) and
    run(observation, num_simulations, model, train) -> (policy, root, action)
(monte_carlo_tree_search.py:224-265) with
    policy  np.float64[A]  raw root-child visit counts (priors when num_simulations == 0)
    root    object with .children {a: node(.visit_count, .prior, ...)} and .value()
            — what Game.store_search_statistics / policy_step read (game.py:154-188)
    action  np.ndarray of action ids 0..A-1
`run` consumes NumPy's *global* legacy stream exactly as the reference does: one
Dirichlet sample (if train and num_simulations > 0) and one uniform(1e-7, 2e-7) per
child scored, so the caller's next np.random draw is the same number.

Additive: `run_batch` searches many games at once with per-game RandomState streams.
"""
import numpy as np
import torch

from . import rng
from .engine import SearchEngine


class NodeView:
    """Read-only view of one node of a finished search (monte_carlo_tree_search.py:6-20)."""

    def __init__(self, tree, g, index, prior):
        self._t, self._g, self._i = tree, g, index
        self.prior = prior
        self.visit_count = int(tree["visit"][g, index])
        self.value_sum = tree["value_sum"][g, index] if self.visit_count else 0
        self.reward = tree["reward"][g, index]
        self.to_play = -1
        self._children = None

    def expanded(self):
        return int(self._t["child_base"][self._g, self._i]) > 0

    def value(self):
        return 0 if self.visit_count == 0 else self.value_sum / self.visit_count

    @property
    def children(self):
        if self._children is None:
            self._children = {}
            base = int(self._t["child_base"][self._g, self._i])
            if base > 0:
                A = self._t["n_actions"]
                for a in range(A):
                    prior = self._t["prior"][self._g, base + a]
                    if self._i == 0:  # root children carry the float64 prior after noise
                        prior = self._t["root_prior"][self._g, a]
                        if not self._t["noisy"]:
                            prior = np.float32(prior)
                    self._children[a] = NodeView(self._t, self._g, base + a, prior)
        return self._children

    @property
    def hidden_state(self):
        base = int(self._t["child_base"][self._g, self._i])
        if base <= 0:
            return 0
        e = (base - 1) // self._t["n_actions"]
        return torch.from_numpy(self._t["hidden"][self._g, e][None, :].copy())


class BatchResult:
    """Host copy of one batched search."""

    def __init__(self, out, tree, n_actions, num_simulations, noisy):
        self.visits = out["visits"]
        self.root_value = out["root_value"]
        self.root_prior = out["root_prior"]
        self.consumed = out["consumed"]
        self.path_len = out["path_len"]
        self.num_simulations = num_simulations
        self._tree = dict(tree, root_prior=out["root_prior"], n_actions=n_actions, noisy=noisy)

    @property
    def policy(self):
        """generate_action_and_policy_base_on_node_count (mcts.py:215-222)"""
        if self.num_simulations >= 1:
            return self.visits.astype(np.float64)
        return self.root_prior.astype(np.float64)

    def root(self, g=0):
        return NodeView(self._tree, g, 0, 0)


class Monte_carlo_tree_search:
    def __init__(self, pb_c_base=19652, pb_c_init=1.25, discount=0.95, root_dirichlet_alpha=0.25,
                 root_exploration_fraction=0.25):
        self._engines = {}
        self.reset(pb_c_base, pb_c_init, discount, root_dirichlet_alpha, root_exploration_fraction)

    def reset(self, pb_c_base=19652, pb_c_init=1.25, discount=0.95, root_dirichlet_alpha=0.25,
              root_exploration_fraction=0.25):
        changed = (getattr(self, "pb_c_base", None), getattr(self, "pb_c_init", None),
                   getattr(self, "discount", None), getattr(self, "root_exploration_fraction", None)
                   ) != (pb_c_base, pb_c_init, discount, root_exploration_fraction)
        self.pb_c_base = pb_c_base
        self.pb_c_init = pb_c_init
        self.discount = discount
        self.root_dirichlet_alpha = root_dirichlet_alpha
        self.root_exploration_fraction = root_exploration_fraction
        self.node = None
        self.model = None
        self.overall_graph = []
        if changed:
            for e, _ in self._engines.values():
                e.close()
            self._engines = {}

    # ------------------------------------------------------------------ engine
    def _engine_for(self, model, n_games, num_simulations):
        # capacity grows in powers of two so a self-play loop reuses one handle
        cap_g = 1 << max(0, (int(n_games) - 1).bit_length())
        key = (model.dims, cap_g, int(num_simulations))
        hit = self._engines.get(key)
        if hit is None:
            eng = SearchEngine(model.dims, cap_g, num_simulations, discount=self.discount,
                               root_exploration_fraction=self.root_exploration_fraction,
                               pb_c_base=self.pb_c_base, pb_c_init=self.pb_c_init,
                               root_dirichlet_alpha=self.root_dirichlet_alpha)
            hit = [eng, (None, -1)]
            self._engines[key] = hit
        eng, (mid, ver) = hit
        if mid != id(model) or ver != model.weights_version:
            eng.set_weights(model.weights_blob())
            hit[1] = (id(model), model.weights_version)
        return eng

    @staticmethod
    def _obs_batch(observation, obs_dim):
        if torch.is_tensor(observation):
            o = observation.detach().to(torch.float32).cpu().numpy()
        else:
            o = np.asarray(observation, dtype=np.float32)
        return np.ascontiguousarray(o.reshape(-1, obs_dim))

    # --------------------------------------------------------------------- run
    def run_batch(self, observations=None, num_simulations=10, model=None, train=True, seeds=None,
                  dirichlet=None, noise=None, want_tree=True):
        """Search every row of `observations` at once. Streams come from `seeds`
        (RandomState(seed) per game: Dirichlet, then tie-break uniforms) unless
        `dirichlet` [G, A] / `noise` [G, >= mzs_max_draws] are given."""
        S = int(num_simulations)
        if getattr(model, "is_RGB", False):          # vision model: (G, 3, 98, 98) images
            obs = model._as_images(observations).numpy()
        else:
            obs = self._obs_batch(observations, model.dims[0])
        G, A = obs.shape[0], model.dims[1]
        noisy = bool(train) and S > 0
        eng = self._engine_for(model, G, S)
        searched = False
        if S > 0 and noise is None:
            if seeds is None:
                raise ValueError("run_batch needs seeds or explicit dirichlet/noise streams")
            if 0.0 < self.root_dirichlet_alpha <= 1.0:
                # generated on the GPU (mzs_search_streams): the host loop over RandomState objects
                # costs ~0.16 ms per game, three orders of magnitude more than the search itself
                eng.search_streams(obs, seeds, S, train)
                searched = True
            else:
                dirichlet, noise = rng.noise_streams(seeds, A, S, self.root_dirichlet_alpha, train)
        if not searched:
            eng.search(obs, dirichlet if noisy else None, noise if S > 0 else None, S, train)
        out = eng.results_host()
        tree = eng.tree_host() if want_tree else {}
        if want_tree and tree["status"].any():
            raise RuntimeError("tie-break stream exhausted (noise shorter than mzs_max_draws)")
        return BatchResult(out, tree, A, S, noisy)

    def run(self, observation=None, num_simulations=10, model=None, train=True):
        """One search, global np.random stream, reference return value."""
        self.model = model
        self.num_simulations = S = int(num_simulations)
        A = model.dims[1]
        noisy = bool(train) and S > 0
        dirichlet = noise = None
        if S > 0:
            # Draw ahead on the global stream, search, then rewind and consume exactly what
            # the reference would have (the draw count depends on the paths taken).
            state = np.random.get_state()
            d, u = rng.draw_stream(np.random, A, S, self.root_dirichlet_alpha, train)
            dirichlet, noise = d[None, :], u[None, :]
        res = self.run_batch(observation, S, model, train, dirichlet=dirichlet, noise=noise)
        if S > 0:
            np.random.set_state(state)
            rng.draw_stream(np.random, A, S, self.root_dirichlet_alpha, train,
                            n_uniform=int(res.consumed[0]))
        self.root = res.root(0)
        self.node = self.root
        policy = res.policy[0]
        action = np.array(list(range(A)))
        return policy, self.root, action
