/*
 * mzsearch_f32v.hip — the fp32 search kernels that gain from -amdgpu-mfma-vgpr-form (matrix results written to
 * VGPRs: no v_accvgpr_read in front of the ELU / softmax code that consumes every layer's result):
 * k_search_quad without mid layers (benchmark shape: 0.2084 -> 0.2050 ms per 4096 x 50 search, A / B on one box),
 * k_search_quadw2 (0.638 -> 0.623), k_search_quadl (0.635 -> 0.628).  The tied-mid-layer kernels stay in
 * mzsearch.hip: their weights fill the accumulator file and they lose 10 % under the flag (tag 133: 0.386 -> 0.426);
 * the two-wave and S_dim 11 kernels measure the same either way.  Part of libmzsearch.so.
 */
#include <hip/hip_runtime.h>

#include "mzs_search_kernels.h"
#include "mzs_wide_kernels.h"

namespace mzs {

template <bool kHidG>
static quad_kernel_t pick_t(const mzs_dims& d) {
    switch (d.n_actions) {
        case 2: return k_search_quad<31, 2, 64, false, false, kHidG>;
        case 3: return k_search_quad<31, 3, 64, false, false, kHidG>;
        case 4: return k_search_quad<31, 4, 64, false, false, kHidG>;
        default: return nullptr;
    }
}

/* S_dim 31 / H 64, no mid layers (the caller checked the shape) */
quad_kernel_t pick_quad_kernel_f32v(const mzs_dims& d, bool hid_global) {
    return hid_global ? pick_t<true>(d) : pick_t<false>(d);
}

/* the weight-stream kernels in fp32 (the caller checked the shape) */
quad_kernel_t pick_wide_kernel_f32v(const mzs_dims& d) {
    if (d.model_kind == MZS_MODEL_LSTM) return k_search_quadl<31, 2, 64>;
    return k_search_quadw2<61, 2, 128>;
}

#ifdef MZS_DIAG
bool diag_drain_f32v(unsigned long long* out8) { return diag_drain(out8); }
#endif

}  // namespace mzs
