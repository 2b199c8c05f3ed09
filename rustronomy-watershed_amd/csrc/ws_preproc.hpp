// ws_preproc.hpp -- launch wrapper of the pre-processor kernels (ws_preproc.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace wsk {

constexpr int PREPROC_BLOCKS = 4096;          // partial (min, max) pairs: scratch = 2 * PREPROC_BLOCKS doubles

size_t preproc_elem_size(int dtype);          // 0 for an unknown dtype
// dtype: ws_dtype of include/ws_hip.h.  scratch[0..1] hold (min, max) afterwards.
hipError_t preprocess(hipStream_t s, const void *data, int dtype, size_t n, uint8_t maxv, double *scratch, uint8_t *out);

}  // namespace wsk
