/*
 * mzs_replay.h — host-side (CPU, no GPU work) inner loops of the replay store: the per-sample
 * arithmetic of ReplayBuffer.sample_position / sample_batch / update_value
 * (/root/reference replay_buffer.py:100-118, :153-157, :169-177) for a whole batch per call.
 *
 * The reference does this with NumPy calls per sample (p / p.sum(), np.random.choice's
 * cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(u, 'right'), np.max).  To stay a drop-in —
 * np.random.seed reproduces the reference's batches, bit for bit — the loops below restate
 * NumPy's own float64 arithmetic: ndarray.sum() of a contiguous 1-D array is NumPy's pairwise
 * summation (numpy/core/src/umath/loops_utils.h.src: < 8 elements sequential, blocks of <= 128
 * with 8 running sums, halves split at a multiple of 8 above that), cumsum is sequential.  The
 * uniform numbers come from NumPy's generator (drawn by the caller), not from here.
 * tests/test_selfplay_oracle.py checks these loops against the NumPy implementation in
 * muzero_amd/selfplay.py (itself pinned to the reference's ReplayBuffer) on random stores.
 * Included by mzsearch.hip (one library).
 */
#ifndef MZS_REPLAY_H
#define MZS_REPLAY_H

#include <stdint.h>

#include <cmath>
#include <vector>

namespace mzs {

/* ndarray.sum() of n contiguous doubles */
inline double numpy_pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise_sum(a, n2) + numpy_pairwise_sum(a + n2, n - n2);
}

/* returns 0, or 1 + index of the first sample whose priorities do not form a distribution
 * (NumPy's choice would raise "probabilities contain NaN") */
inline int replay_sample_positions(int32_t batch, const double* const* rows, const int32_t* row_len,
                                   const double* u, int64_t* pos, double* mouve) {
    std::vector<double> cdf;
    for (int32_t b = 0; b < batch; ++b) {
        const double* p = rows[b];
        const int64_t n = row_len[b];
        if (n < 1) return 1 + b;
        const double s = numpy_pairwise_sum(p, n);
        if (u) {
            cdf.resize((size_t)n);
            double c = 0.0;
            bool ok = true;
            for (int64_t i = 0; i < n; ++i) {
                const double soft = p[i] / s;
                ok = ok && soft >= 0.0; /* false for NaN too */
                c = i == 0 ? soft : c + soft;
                cdf[(size_t)i] = c;
            }
            if (!ok || !std::isfinite(cdf[(size_t)n - 1])) return 1 + b;
            const double last = cdf[(size_t)n - 1];
            /* searchsorted(cdf / last, u, side='right'): number of entries <= u */
            int64_t lo = 0, hi = n;
            while (lo < hi) {
                const int64_t mid = lo + (hi - lo) / 2;
                if (cdf[(size_t)mid] / last <= u[b])
                    lo = mid + 1;
                else
                    hi = mid;
            }
            pos[b] = lo;
        }
        if (pos[b] < 0 || pos[b] >= n) return 1 + b;
        mouve[b] = p[pos[b]] / s;
    }
    return 0;
}

inline void replay_update_value(int32_t batch, int32_t num_unroll, double* const* rows, const int32_t* row_len,
                                const int32_t* game_len, const int64_t* game_idx, const int64_t* pos,
                                const double* new_value, double* prio_game) {
    for (int32_t b = 0; b < batch; ++b) { /* in sample order: a later sample overwrites an earlier one */
        double* p = rows[b];
        const int64_t n = row_len[b];
        int64_t hi = pos[b] + num_unroll;
        if (hi > (int64_t)game_len[b] - 1) hi = (int64_t)game_len[b] - 1;
        if (hi > n) hi = n;
        for (int64_t h = pos[b]; h < hi; ++h) p[h] = new_value[(size_t)(h - pos[b]) * batch + b];
        double m = p[0];
        for (int64_t i = 1; i < n; ++i) m = p[i] > m ? p[i] : m;
        prio_game[game_idx[b]] = m;
    }
}

}  // namespace mzs
#endif
