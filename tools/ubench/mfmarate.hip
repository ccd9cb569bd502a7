// Issue rate of bf16 MFMA shapes on one SIMD: N independent accumulators, ITER rounds, timed with s_memtime.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern "C" __global__ void mfmarate(float* out, long long* cyc, int iters) {
  f32x4 acc[8]; f32x4 acc2[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; acc2[i] = f32x4{0, 0, 0, 0}; }
  bf16x8 a8, b8; s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(i * 3); }
  long long t0, t1, t2;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc2[i], 0, 0, 0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc2[i][1];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
extern "C" int run_mfmarate(float* out, long long* cyc, int iters, void* stream) {
  hipLaunchKernelGGL(mfmarate, dim3(1), dim3(64), 0, (hipStream_t)stream, out, cyc, iters);
  return (int)hipGetLastError();
}
