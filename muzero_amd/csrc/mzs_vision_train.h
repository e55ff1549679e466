/*
 * mzs_vision_train.h — interface between mzsearch.hip (the C-ABI) and mzs_vision_train.hip (the training
 * step of the 'vision_model': batched layer kernels + a host-side tape).  Not part of the public ABI.
 */
#ifndef MZS_VISION_TRAIN_H
#define MZS_VISION_TRAIN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/mzs_layout.h"

namespace mzs {

struct VisTrain;

/* allocates the activation / gradient arenas for batches of up to max_batch samples and unroll K */
int vis_train_create(VisTrain** out, const mzs_dims& d, int max_batch, int K, int device, std::string* err);
void vis_train_destroy(VisTrain* vt);

/* Dropout2d mask bytes one step on `batch` samples consumes (one per (call, sample, channel), call order) */
int64_t vis_train_mask_count(VisTrain* vt, int batch);

/* device array [n_params]: 1 where the blob element is a BatchNorm running statistic (mzs_train_apply adds
 * the "gradient" slot to it instead of running the optimizer on it) */
const unsigned char* vis_train_stat_flags(const VisTrain* vt);

/* forward (training mode) + losses + backward; arguments as mzs_train_grad (include/mzsearch.h).
 * quad_store_dev: the search kernels' store of the Linear layers (mzs_vision.h vis_lin_offsets), which the forward
 * pass reads too.
 * masks_dev: vis_train_mask_count(batch) bytes, or null = drawn on the device from (seed, step). */
int vis_train_grad(VisTrain* vt, const float* blob_dev, const float* quad_store_dev, int batch, const float* obs_dev, const float* actions_dev,
                   const float* tv_dev, const float* tr_dev, const float* tp_dev, const float* is_ratio_dev,
                   float inv_global_batch, int loss_value, int loss_reward, const unsigned char* masks_dev,
                   uint64_t seed, uint64_t step, float* grad_dev, float* loss_dev, float* priority_dev,
                   hipStream_t st, std::string* err);

}  // namespace mzs
#endif
