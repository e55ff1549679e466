/*
 * mzsearch_x3w.hip — the weight-stream kernels (mzs_wide_kernels.h) on the hi + lo bf16 stream of
 * mzs_quadw_bf16.h (contract "dot_bf16x3"), built like mzsearch_bf16.hip with -amdgpu-mfma-vgpr-form (every
 * matrix result feeds a VALU add).  A translation unit of their own: the pair kernel sits at the edge of the
 * register file (128 stream registers + 32 split inputs + 32 chain registers), and while its matrix work was not
 * yet fenced group by group (mzs_quadw_bf16.h wide_quarter_x) this compiler's "Rewrite AGPR-Copy-MFMA" pass
 * crashed on it; kept apart so that such a failure names this unit.  Part of libmzsearch.so.
 */
#include <hip/hip_runtime.h>

#include "mzs_search_kernels.h"
#include "mzs_wide_kernels.h"

namespace mzs {

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
