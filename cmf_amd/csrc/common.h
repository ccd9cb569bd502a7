// Shared helpers for the gfx950 kernels of libcmf_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include "cmf_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CMF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e_ = hipGetLastError();             \
    if (e_ != hipSuccess) return (int)e_;          \
  } while (0)

// runtime.hip: per-(device, kernel) memo of the dynamic-LDS attribute and the per-device CU count (keyed by the calling
// thread's current device; see the file header for why a process-wide flag is wrong)
hipError_t cmf_set_dynamic_lds(const void* fn, int bytes);
int cmf_device_cus();

static inline int cmf_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// sum over the 64 lanes of a wavefront (every lane gets the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blocks of up to 1024 threads; `red` is >= 16 floats of LDS; result in every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
