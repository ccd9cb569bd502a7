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
__device__ __forceinline__ unsigned short pack_bf16x3_elem(const cmf_pack_desc& e, long long i, float scale) {
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
  if (e.kind == 2) {                                   // cmf_pack_weight_f16x3: fp16 halves of w 2^k
    v *= scale;
    const _Float16 h = (_Float16)v;
    const _Float16 r = hl ? (_Float16)(v - (float)h) : h;
    return __builtin_bit_cast(unsigned short, r);
  }
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
    float scale = 1.f;
    if (e.kind == 2) {
      // fp16 pack: max |w| 2^k in [2^11, 2^12) (pack_f16_scale_kernel's rule).  EVERY block takes the maximum of the whole weight
      // itself (147 KB for a 64 -> 64 layer, L2-resident): no hand-off between blocks; block 0 writes the 16-byte trailer.
      __shared__ float red[4];
      const long long n = (long long)e.cout * e.cin * 9;
      float m = 0.f;
      for (long long i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(e.w[i]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
      __syncthreads();
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      const int ex = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
      int k = (ex == 0 || ex == 255) ? 0 : 12 - (ex - 126);
      k = k < -60 ? -60 : k > 60 ? 60 : k;
      scale = __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        float* tr = reinterpret_cast<float*>(out + e.total);
        tr[0] = scale;
        tr[1] = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
        tr[2] = tr[3] = 0.f;
      }
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < e.total; i += stride) out[i] = pack_bf16x3_elem(e, i, scale);
  }
}

}  // namespace

extern "C" int cmf_pack_weights_batched(const cmf_pack_desc* table, int n, void* stream) {
  if (!table || n <= 0 || n > 65535) return CMF_EINVAL;
  hipLaunchKernelGGL(pack_batched_kernel, dim3(32, n), dim3(256), 0, (hipStream_t)stream, table);
  CMF_LAUNCH_CHECK();
  return 0;
}
