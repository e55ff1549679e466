/*
 * mzsearch_x3w.hip — the weight-stream kernels (mzs_wide_kernels.h) on the two bf16 streams of mzs_quadw_bf16.h:
 * contract "dot_bf16" (WideStreamB) and contract "dot_bf16x3" (WideStreamX: hi + lo images).  Built like
 * mzsearch_bf16.hip with -amdgpu-mfma-vgpr-form (every matrix result feeds a VALU add) and, this unit only, with
 * -amdgpu-sched-strategy=max-ilp: A / B on one box the fast-bf16 stream kernels gain (wide MLP 0.602 -> 0.570 ms per
 * 4096 x 50 search, LSTM 0.590 -> 0.579), the hi + lo ones measure the same, and every k_search_quad build LOSES 2-7 %
 * under it (so mzsearch_bf16.hip and mzsearch_f32v.hip keep the default strategy).
 * A translation unit of its own also because the hi + lo pair kernel sits at the edge of the register file (128
 * stream registers + 32 split inputs + 32 chain registers): while its matrix work was not yet fenced group by group
 * (mzs_quadw_bf16.h wide_quarter_x) this compiler's "Rewrite AGPR-Copy-MFMA" pass crashed on it; kept apart so that such
 * a failure names this unit.  Part of libmzsearch.so.
 */
#include <hip/hip_runtime.h>

#include "mzs_search_kernels.h"
#include "mzs_wide_kernels.h"

namespace mzs {

/* the wide MLP of experiments 127 / 128 and the one-layer 'lstm_model' shape (the caller checked the shape) */
quad_kernel_t pick_wide_kernel_bf16(const mzs_dims& d) {
    if (d.precision == MZS_PRECISION_BF16X3) return pick_wide_kernel_x3(d);
    if (d.model_kind == MZS_MODEL_MLP && d.state_dim == 61 && d.hidden_dim == 128 && d.n_actions == 2)
        return k_search_quadw2<61, 2, 128, 1>;
    if (d.model_kind == MZS_MODEL_LSTM && d.state_dim == 31 && d.hidden_dim == 64 && d.n_actions == 2)
        return k_search_quadl<31, 2, 64, 1>;
    return nullptr;
}

quad_kernel_t pick_wide_kernel_x3(const mzs_dims& d) {
    if (d.model_kind == MZS_MODEL_MLP && d.state_dim == 61 && d.hidden_dim == 128 && d.n_actions == 2)
        return k_search_quadw2<61, 2, 128, 2>;
    if (d.model_kind == MZS_MODEL_LSTM && d.state_dim == 31 && d.hidden_dim == 64 && d.n_actions == 2)
        return k_search_quadl<31, 2, 64, 2>;
    return nullptr;
}

#ifdef MZS_DIAG
bool diag_drain_x3w(unsigned long long* out8) { return diag_drain(out8); }
#endif

}  // namespace mzs
