// What the matrix pipes sustain on LIVE data: every SIMD of the chip issues back-to-back v_mfma_f32_16x16x32_bf16 whose
// operands change with every instruction (eight A and eight B fragments of random bf16 values per wave, eight accumulators),
// or the same operand pair every time (mode 0: what mfmachain measures).  Operand toggling is what the multiplier arrays
// burn power on, so the clock the chip holds -- and with it the reachable fraction of the 2.4 GHz peak -- depends on it.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int LIVE>
__global__ __launch_bounds__(256) void mfmapower(const bf16x8* __restrict__ src, float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    acc[i] = f32x4{0, 0, 0, 0};
    a[i] = src[(threadIdx.x * 16 + i) & 4095];
    b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
  }
  long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LIVE ? a[(i + r) & 7] : a[0], LIVE ? b[(i + 3 * r) & 7] : b[0], acc[i], 0, 0, 0);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[(blockIdx.x * 256 + threadIdx.x) & 1023] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// the same load on v_mfma_f32_32x32x16_bf16: half as many operand-register reads per flop (A 32x16 + B 16x32 per 32 Kflop against
// twice 16x32 + 32x16), four accumulators of 16 registers
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LIVE>
__global__ __launch_bounds__(256) void mfmapower32(const bf16x8* __restrict__ src, float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  bf16x8 a[8], b[8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(threadIdx.x * 16 + i) & 4095];
    b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
  }
  long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LIVE ? a[(i + r) & 7] : a[0], LIVE ? b[(i + 3 * r) & 7] : b[0], acc[i], 0, 0, 0);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[(blockIdx.x * 256 + threadIdx.x) & 1023] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
extern "C" int run_mfmapower32(const void* src, float* out, long long* cyc, int iters, int nblocks, int live, void* stream) {
  if (live) hipLaunchKernelGGL(mfmapower32<1>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src, out, cyc, iters);
  else hipLaunchKernelGGL(mfmapower32<0>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src, out, cyc, iters);
  return (int)hipGetLastError();
}
extern "C" int run_mfmapower(const void* src, float* out, long long* cyc, int iters, int nblocks, int live, void* stream) {
  if (live) hipLaunchKernelGGL(mfmapower<1>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src, out, cyc, iters);
  else hipLaunchKernelGGL(mfmapower<0>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src, out, cyc, iters);
  return (int)hipGetLastError();
}
