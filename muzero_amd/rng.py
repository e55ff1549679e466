"""Host-side random streams for the batched search.

The reference draws from NumPy's *global* legacy MT19937 stream
(monte_carlo_tree_search.py:140 ``np.random.dirichlet``, :163
``np.random.uniform(1e-7, 2e-7)`` once per child scored).  The device kernels
consume a pre-drawn stream per game instead; these helpers draw it with the same
generator (``numpy.random.RandomState``), so a game seeded with ``seed`` sees the
numbers the reference would see after ``np.random.seed(seed)``.
"""
import numpy as np


def max_draws(n_actions, num_simulations):
    """Upper bound on tie-break draws of one search (include/mzs_layout.h mzs_max_draws)."""
    return int(n_actions) * int(num_simulations) * (int(num_simulations) + 1) // 2


def pbc_table(num_simulations, pb_c_base=19652, pb_c_init=1.25):
    """sqrt(N) * pb_c(N) for parent visit counts N = 0..S, evaluated exactly as
    ``ucb_score`` does (monte_carlo_tree_search.py:156-157): python-float quotient,
    ``np.log``/``np.sqrt`` in float64.  Kept on the host so the device needs no
    fp64 libm."""
    tab = np.zeros(int(num_simulations) + 2, np.float64)
    for n in range(tab.shape[0]):
        pb_c = np.log((n + pb_c_base + 1) / pb_c_base) + pb_c_init
        tab[n] = np.sqrt(n) * pb_c
    return tab


def draw_stream(rs, n_actions, num_simulations, alpha, train, n_uniform=None):
    """Dirichlet sample (only if ``train`` and S > 0, monte_carlo_tree_search.py:135-141)
    followed by ``n_uniform`` tie-break draws from RandomState ``rs``."""
    if n_uniform is None:
        n_uniform = max_draws(n_actions, num_simulations)
    if train and num_simulations > 0:
        dirichlet = rs.dirichlet([alpha] * n_actions)
    else:
        dirichlet = np.zeros(n_actions, np.float64)
    noise = rs.uniform(1e-7, 2e-7, size=n_uniform) if n_uniform else np.zeros(0, np.float64)
    return dirichlet, noise


def noise_streams(seeds, n_actions, num_simulations, alpha=0.25, train=True, stride=None):
    """Per-game streams for ``seeds``: returns (dirichlet [G, A] f64, noise [G, stride] f64)."""
    seeds = np.asarray(seeds).ravel()
    if stride is None:
        stride = max(1, max_draws(n_actions, num_simulations))
    G = seeds.shape[0]
    dirichlet = np.zeros((G, n_actions), np.float64)
    noise = np.zeros((G, stride), np.float64)
    n_u = min(stride, max_draws(n_actions, num_simulations))
    for g in range(G):
        rs = np.random.RandomState(int(seeds[g]))
        d, u = draw_stream(rs, n_actions, num_simulations, alpha, train, n_u)
        dirichlet[g] = d
        noise[g, :n_u] = u
    return dirichlet, noise
