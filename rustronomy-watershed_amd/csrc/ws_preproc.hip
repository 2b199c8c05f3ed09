// ws_preproc.hip -- WatershedUtils::pre_processor / pre_processor_with_max on the GPU
// (lib.rs:1081-1173): any numeric array -> u8 in [0, MAX], with the reference's special cases.
//
// Restated exactly, quirks included:
//   min = fold(0, |acc, x| if x < acc && x is finite { x } else { acc })        lib.rs:1147-1151
//   max = fold(0, |acc, x| if x > acc && x is finite { x } else { acc })        lib.rs:1152-1156
//     -> both folds are seeded with ZERO, so min <= 0 <= max whatever the data
//   for every x (as f64):                                                       lib.rs:1159-1172
//     is_normal(x)            -> trunc(((x - min) / (max - min)) * MAX)   (two roundings, f64)
//     x == +inf               -> ALWAYS_FILL (0)     (the reference's comment says -inf; the code says +inf)
//     anything else           -> NEVER_FILL (255)    (NaN, -inf, subnormals and exact 0)
// f64 add/sub/mul/div are IEEE correctly rounded on gfx950 in the default (non fast-math) build, so
// the result is bit-identical to the reference's f64 arithmetic.
#include "ws_common.hpp"
#include "ws_preproc.hpp"

#include <math.h>

namespace wsk {

template <typename T>
__device__ __forceinline__ double to_f64(T v) { return (double)v; }

__device__ __forceinline__ bool finite_f64(double v) { return fabs(v) <= 1.7976931348623157e308; }   // false for NaN, +-inf

// a value the fold keeps: strictly below/above the running bound and finite
template <typename T>
__global__ __launch_bounds__(256) void k_minmax(const T *__restrict__ data, size_t n, double *partial) {
  __shared__ double s_min[4], s_max[4];
  double mn = 0.0, mx = 0.0;                                 // lib.rs:1149 / 1154: T::zero()
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const double v = to_f64(data[i]);
    if (finite_f64(v)) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double a = __shfl_down(mn, off, 64), b = __shfl_down(mx, off, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_min[wave] = mn; s_max[wave] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) { mn = s_min[k] < mn ? s_min[k] : mn; mx = s_max[k] > mx ? s_max[k] : mx; }
    partial[2 * blockIdx.x] = mn;
    partial[2 * blockIdx.x + 1] = mx;
  }
}

__global__ __launch_bounds__(256) void k_minmax_final(double *partial, int nblocks) {
  __shared__ double s_min[256], s_max[256];
  double mn = 0.0, mx = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    mn = partial[2 * i] < mn ? partial[2 * i] : mn;
    mx = partial[2 * i + 1] > mx ? partial[2 * i + 1] : mx;
  }
  s_min[threadIdx.x] = mn;
  s_max[threadIdx.x] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s_min[threadIdx.x] = s_min[threadIdx.x + off] < s_min[threadIdx.x] ? s_min[threadIdx.x + off] : s_min[threadIdx.x];
      s_max[threadIdx.x] = s_max[threadIdx.x + off] > s_max[threadIdx.x] ? s_max[threadIdx.x + off] : s_max[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[0] = s_min[0]; partial[1] = s_max[0]; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_quantise(const T *__restrict__ data, size_t n, const double *__restrict__ minmax,
                                                  double maxv, uint8_t *__restrict__ out) {
  const double mn = minmax[0], range = minmax[1] - minmax[0];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const double v = to_f64(data[i]);
    const double a = fabs(v);
    uint8_t q;
    if (a >= 2.2250738585072014e-308 && a <= 1.7976931348623157e308) {      // f64::is_normal
      const double normal = (v - mn) / range;                               // lib.rs:1163
      q = (uint8_t)(normal * maxv);                                         // lib.rs:1164: to_u8 truncates; in [0, MAX] by construction
    } else if (v == INFINITY) {
      q = 0;                                                                // lib.rs:1165-1167
    } else {
      q = 255;                                                              // lib.rs:1168-1170
    }
    out[i] = q;
  }
}

template <typename T>
static hipError_t run(hipStream_t s, const void *data, size_t n, uint8_t maxv, double *scratch, uint8_t *out) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 2047) / 2048 < PREPROC_BLOCKS ? (n + 2047) / 2048 : PREPROC_BLOCKS);
  k_minmax<T><<<blocks, 256, 0, s>>>((const T *)data, n, scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  k_minmax_final<<<1, 256, 0, s>>>(scratch, blocks);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int qblocks = (int)((n + 1023) / 1024 < 16384 ? (n + 1023) / 1024 : 16384);
  k_quantise<T><<<qblocks, 256, 0, s>>>((const T *)data, n, scratch, (double)maxv, out);
  return hipGetLastError();
}

size_t preproc_elem_size(int dtype) {
  switch (dtype) {
    case 0: return 4;   // f32
    case 1: return 8;   // f64
    case 2: return 4;   // i32
    case 3: return 2;   // u16
    case 4: return 2;   // i16
    case 5: return 1;   // u8
    default: return 0;
  }
}

hipError_t preprocess(hipStream_t s, const void *data, int dtype, size_t n, uint8_t maxv, double *scratch, uint8_t *out) {
  switch (dtype) {
    case 0: return run<float>(s, data, n, maxv, scratch, out);
    case 1: return run<double>(s, data, n, maxv, scratch, out);
    case 2: return run<int32_t>(s, data, n, maxv, scratch, out);
    case 3: return run<uint16_t>(s, data, n, maxv, scratch, out);
    case 4: return run<int16_t>(s, data, n, maxv, scratch, out);
    case 5: return run<uint8_t>(s, data, n, maxv, scratch, out);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wsk
