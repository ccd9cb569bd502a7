// MFMA issue rate vs number of interleaved independent accumulator chains (dependent-MFMA latency), one wave.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N>
__device__ long long run(bf16x8 a, bf16x8 b, f32x4* acc, int iters) {
  long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / N; ++r)
#pragma unroll
      for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  return t1 - t0;
}
extern "C" __global__ void mfmachain(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
  long long c1 = run<1>(a8, b8, acc, iters), c2 = run<2>(a8, b8, acc, iters), c4 = run<4>(a8, b8, acc, iters), c8 = run<8>(a8, b8, acc, iters);
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[threadIdx.x & 63] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1; cyc[1] = c2; cyc[2] = c4; cyc[3] = c8; }
}
extern "C" int run_mfmachain(float* out, long long* cyc, int iters, int nblocks, int threads, void* stream) {
  hipLaunchKernelGGL(mfmachain, dim3(nblocks), dim3(threads), 0, (hipStream_t)stream, out, cyc, iters);
  return (int)hipGetLastError();
}
