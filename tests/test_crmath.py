"""include/mzs_crmath.h (double-double log / pow) and the stream oracle built on it
(oracle/mz_stream_oracle.c: numpy.random.RandomState(seed).dirichlet + .uniform restated).

Three pins:
  * mzs_cr_log / mzs_cr_pow against decimal arithmetic at 60 digits: the correctly rounded double on every
    sample (glibc's own log / pow, which NumPy's legacy sampler calls, miss it on ~0.05 % of them);
  * the stream oracle's uniforms == RandomState's, bit for bit;
  * the stream oracle's Dirichlet rows vs RandomState.dirichlet itself (monte_carlo_tree_search.py:140):
    never more than one ulp apart, and bit-identical on all but the handful of rows where glibc misrounds —
    the exact count measured in this image is printed and bounded.
CPU-only."""
import math
from decimal import Decimal, getcontext

import numpy as np
import pytest

from muzero_amd import rng
from oracle import mz_oracle as mzo


def cr_log(x):
    return float(Decimal(x).ln())


def cr_pow(x, y):
    return float((Decimal(x).ln() * Decimal(y)).exp())


def test_cr_log_is_correctly_rounded():
    getcontext().prec = 60
    rs = np.random.RandomState(2)
    n = 4000
    xs = np.concatenate([rs.random_sample(n), 1.0 - rs.random_sample(n) * 2.0 ** -rs.randint(0, 50, n),
                         np.exp(rs.uniform(-40, 40, n)), 1 + rs.uniform(-1, 1, n) * 2.0 ** -rs.randint(1, 40, n),
                         [0.75, 1.5, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, 2.0 ** -53, 0.9999, 1.0001, 4.0]])
    xs = xs[xs > 0]
    L = mzo.lib()
    bad = [x for x in xs if L.mzo_cr_log(float(x)) != cr_log(float(x))]
    glibc = sum(1 for x in xs if math.log(x) != cr_log(float(x)))
    print(f"\nlog: {len(bad)} of {len(xs)} not correctly rounded (glibc: {glibc})")
    assert not bad
    assert L.mzo_cr_log(1.0) == 0.0


@pytest.mark.parametrize("y", [4.0, 1 / 0.3, 2.0, 1 / 0.03, 1 / 0.9])
def test_cr_pow_is_correctly_rounded(y):
    getcontext().prec = 60
    rs = np.random.RandomState(int(y * 10))
    n = 2500
    xs = np.concatenate([rs.random_sample(n), 0.75 + rs.random_sample(n) * 5, rs.random_sample(200) * 2.0 ** -40,
                         [2.0 ** -53, 1.0 - 2.0 ** -53]])
    L = mzo.lib()
    bad = [x for x in xs if L.mzo_cr_pow(float(x), y) != cr_pow(float(x), y)]
    print(f"\npow(., {y:.4g}): {len(bad)} of {len(xs)} not correctly rounded")
    assert not bad
    assert L.mzo_cr_pow(0.0, y) == 0.0 and L.mzo_cr_pow(1.0, y) == 1.0


# (alpha, A, seeds): rows of RandomState.dirichlet the oracle must reproduce
STREAM_CASES = [(0.25, 2, 4096), (0.25, 4, 2048), (0.3, 3, 1024), (0.03, 5, 512), (0.9, 2, 512), (1.0, 4, 512),
                (0.5, 18, 256)]


@pytest.mark.parametrize("alpha,A,n", STREAM_CASES)
def test_stream_oracle_vs_randomstate(alpha, A, n):
    S = 6
    seeds = np.concatenate([np.arange(n - 24), [2 ** 32 - 1, 2 ** 31, 123456789, 7],
                            np.random.RandomState(int(alpha * 100) + A).randint(0, 2 ** 31, size=20)]).astype(np.int64)
    d, u = mzo.streams(seeds, A, S, alpha, True)
    ref_d, ref_u = rng.noise_streams(seeds, A, S, alpha, True)
    assert np.array_equal(u.view(np.uint64), ref_u.view(np.uint64))          # same accept / reject path, same words
    big = ref_d > 1e-300
    ulps = np.abs(d - ref_d)[big] / np.spacing(np.abs(ref_d[big]))
    rows_equal = int(np.all(d.view(np.uint64) == ref_d.view(np.uint64), axis=1).sum())
    print(f"\nalpha {alpha} A {A}: {rows_equal}/{n} Dirichlet rows bit-identical to NumPy, max {ulps.max():.0f} ulp")
    # one gamma off by an ulp moves the shared 1 / sum: the row's other components follow by <= 1 ulp each
    assert ulps.max() <= 2
    assert rows_equal >= n - max(2, int(0.02 * n * A / 2))
    assert np.allclose(d.sum(axis=1), 1.0, atol=1e-15)


def test_stream_oracle_no_noise_and_alpha_range():
    d, u = mzo.streams([5, 6], 2, 0, 0.25, True)
    assert np.all(d == 0)
    d, u = mzo.streams([5, 6], 2, 4, 0.25, False)
    ref_d, ref_u = rng.noise_streams([5, 6], 2, 4, 0.25, False)
    assert np.all(d == 0) and np.array_equal(u, ref_u)
    with pytest.raises(ValueError):
        mzo.streams([1], 2, 4, 1.5, True)
