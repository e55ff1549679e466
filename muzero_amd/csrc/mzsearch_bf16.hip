/*
 * mzsearch_bf16.hip — the bf16 instantiations of k_search_quad (mzs_search_kernels.h,
 * mzs_quad_bf16.h), a translation unit of their own because they want MFMA results in VGPRs
 * (-mllvm -amdgpu-mfma-vgpr-form, see muzero_amd/csrc/Makefile); part of libmzsearch.so.  The bf16 weight-stream
 * kernels live in mzsearch_x3w.hip.
 */
#include <hip/hip_runtime.h>

#include "mzs_search_kernels.h"

namespace mzs {

/* bf16 (contract "dot_bf16", kPrec 1; "dot_bf16x3", kPrec 2): every shape the fp32 kernel is instantiated for
 * (mzsearch.hip pick_quad_kernel_t), with or without tied mid layers */
template <bool kHidG, int kPrec>
static quad_kernel_t pick_quad_kernel_b16_t(const mzs_dims& d) {
    if (d.obs_dim > 64 || d.model_kind != MZS_MODEL_MLP) return nullptr;
    const bool mid = d.n_mid_layers != 0;
    if (d.state_dim == 31 && d.hidden_dim == 64) {
        switch (d.n_actions) {
            case 2: return mid ? k_search_quad<31, 2, 64, false, true, kHidG, kPrec> : k_search_quad<31, 2, 64, false, false, kHidG, kPrec>;
            case 3: return mid ? k_search_quad<31, 3, 64, false, true, kHidG, kPrec> : k_search_quad<31, 3, 64, false, false, kHidG, kPrec>;
            case 4: return mid ? k_search_quad<31, 4, 64, false, true, kHidG, kPrec> : k_search_quad<31, 4, 64, false, false, kHidG, kPrec>;
            default: return nullptr;
        }
    }
    if (d.state_dim == 11 && d.n_actions == 2 && d.hidden_dim == 16)
        return mid ? k_search_quad<11, 2, 16, false, true, kHidG, kPrec> : k_search_quad<11, 2, 16, false, false, kHidG, kPrec>;
    return nullptr;
}
/* the same kernels compiled for two wavefronts per SIMD (hidden states in the global array) */
static quad_kernel_t pick_quad_kernel_b16_occ2(const mzs_dims& d) {
    if (d.obs_dim > 64 || d.model_kind != MZS_MODEL_MLP) return nullptr;
    const bool mid = d.n_mid_layers != 0; /* packed, even the tied mid layers fit 256 registers (~12 reloads per simulation) */
    if (d.state_dim == 31 && d.hidden_dim == 64) {
        if (d.n_actions == 2) return mid ? k_search_quad<31, 2, 64, false, true, true, 1, 2> : k_search_quad<31, 2, 64, false, false, true, 1, 2>;
        if (d.n_actions == 4) return mid ? k_search_quad<31, 4, 64, false, true, true, 1, 2> : k_search_quad<31, 4, 64, false, false, true, 1, 2>;
    }
    return nullptr;
}
quad_kernel_t pick_quad_kernel_bf16(const mzs_dims& d, bool hid_global, bool two_per_simd) {
    if (d.precision == MZS_PRECISION_BF16X3) { /* hi + lo weight pairs: the fp32 kernel's register count, one wave per SIMD
                                                * (built for two: 392 registers spilled, 8192 games 0.584 ms against 0.489) */
        if (two_per_simd) return nullptr;
        return hid_global ? pick_quad_kernel_b16_t<true, 2>(d) : pick_quad_kernel_b16_t<false, 2>(d);
    }
    if (two_per_simd) return hid_global ? pick_quad_kernel_b16_occ2(d) : nullptr;
    return hid_global ? pick_quad_kernel_b16_t<true, 1>(d) : pick_quad_kernel_b16_t<false, 1>(d);
}

#ifdef MZS_DIAG
bool diag_drain_bf16(unsigned long long* out8) { return diag_drain(out8); }
#endif

}  // namespace mzs
