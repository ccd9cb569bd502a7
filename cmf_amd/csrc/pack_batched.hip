// Batched weight re-packing: every stale pack of a model in ONE launch.
//
// A training step invalidates all packed weight copies (the optimiser rewrites the parameters); the next forward / backward pass
// needs each conv / linear weight in up to four kernel layouts (fp32 and split-precision, forward and adjoint): ~780 launches of
// cmf_pack_weight / cmf_pack_weight_bf16x3_t of ~4 us each per step for the MNIST / CIFAR models (3.3 ms, rocprofv3
// profiles/r03_c5_train_kernel_stats.csv).  cmf_pack_weights_batched walks a device-resident table of (source, destination, shape,
// layout) descriptors: blockIdx.y = table entry, blockIdx.x strides over its elements.  The element formulas are the ones of
// pack_weight_kernel (elementwise.hip) and pack_weight_bf16x3_kernel (conv_tangent_bf16x3.hip); tests/test_gpu_round3.py checks the
// batched packs bit for bit against the single-weight entry points.
#include "common.h"

namespace {

__device__ __forceinline__ float pack_f32_elem(const cmf_pack_desc& e, long long i) {
  const int ncout = e.transpose ? e.cin : e.cout, ncin = e.transpose ? e.cout : e.cin;
  const int ncin_pad = (ncin + 7) / 8 * 8;
  const int col = (int)(i & 63);
  long long t = i >> 6;
  const int ci = (int)(t % ncin_pad);
  t /= ncin_pad;
  const int tap = (int)(t % e.taps), cog = (int)(t / e.taps);
  const int co = cog * 64 + col;
  if (co >= ncout || ci >= ncin) return 0.f;
  return e.transpose ? e.w[((long long)ci * e.cin + co) * e.taps + (e.taps - 1 - tap)] : e.w[((long long)co * e.cin + ci) * e.taps + tap];
}

// cout / cin here are the PACKED operator's (the adjoint's when transpose: the caller swaps them, like cmf_pack_weight_bf16x3_t's)
__device__ __forceinline__ unsigned short pack_bf16x3_elem(const cmf_pack_desc& e, long long i) {
  const int j = (int)(i & 7), col = (int)((i >> 3) & 15), kq = (int)((i >> 7) & 3), cot = (int)((i >> 9) & 3);
  long long t = i >> 11;
  const int s = (int)(t % 3);
  t /= 3;
  const int hl = (int)(t & 1);
  t >>= 1;
  const int nchunks = e.cin / 8;
  const int ch = (int)(t % nchunks), cog = (int)(t / nchunks);
  const int co = cog * 64 + cot * 16 + col;
  int ci = ch * 8 + j, tap = s == 0 ? kq : 5 + kq;
  if (s == 2) {
    tap = (ch & 3) == 3 ? 4 : 9;
    ci = (ch - 3 + kq) * 8 + j;
  }
  float v = 0.f;
  if (co < e.cout && tap < 9)
    v = e.transpose ? e.w[((long long)ci * e.cout + co) * 9 + (8 - tap)] : e.w[((long long)co * e.cin + ci) * 9 + tap];
  const __bf16 h = (__bf16)v;
  const __bf16 r = hl ? (__bf16)(v - (float)h) : h;
  return __builtin_bit_cast(unsigned short, r);
}

__global__ __launch_bounds__(256) void pack_batched_kernel(const cmf_pack_desc* __restrict__ table) {
  const cmf_pack_desc e = table[blockIdx.y];
  const long long stride = (long long)gridDim.x * 256;
  if (e.kind == 0) {
    float* out = reinterpret_cast<float*>(e.out);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < e.total; i += stride) out[i] = pack_f32_elem(e, i);
  } else {
    unsigned short* out = reinterpret_cast<unsigned short*>(e.out);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < e.total; i += stride) out[i] = pack_bf16x3_elem(e, i);
  }
}

}  // namespace

extern "C" int cmf_pack_weights_batched(const cmf_pack_desc* table, int n, void* stream) {
  if (!table || n <= 0 || n > 65535) return CMF_EINVAL;
  hipLaunchKernelGGL(pack_batched_kernel, dim3(32, n), dim3(256), 0, (hipStream_t)stream, table);
  CMF_LAUNCH_CHECK();
  return 0;
}
