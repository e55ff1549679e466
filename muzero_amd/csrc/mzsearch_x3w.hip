/*
 * mzsearch_x3w.hip — the weight-stream kernels (mzs_wide_kernels.h) on the hi + lo bf16 stream of
 * mzs_quadw_bf16.h (contract "dot_bf16x3").  A translation unit of their own, built WITHOUT mzsearch_bf16.hip's
 * -amdgpu-mfma-vgpr-form: with 64 stream registers per buffer the wide-MLP kernel needs the accumulator file,
 * and this compiler's "Rewrite AGPR-Copy-MFMA" pass crashes on it under that flag.  Part of libmzsearch.so.
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

}  // namespace mzs
