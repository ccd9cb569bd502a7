// Streaming-read micro-benchmark: every lane loads 16 B; a wave-instruction covers 1 KiB made of contiguous
// segments of SEG bytes placed STRIDE bytes apart (SEG = 64: the tangent-conv staging pattern).
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void segread(const float* __restrict__ in, float* __restrict__ out, int seg, int stride, long long rows_per_wave, int iters) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lanes_per_seg = seg / 16;
  const long long nseg = 64 / lanes_per_seg;              // segments per wave-instruction
  f4 acc = {0, 0, 0, 0};
  const char* base = (const char*)in + wave * rows_per_wave * stride;
  for (long long r = 0; r < rows_per_wave; r += nseg * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long row = r + u * nseg + lane / lanes_per_seg;
      const f4 v = *(const f4*)(base + row * stride + (lane % lanes_per_seg) * 16);
      acc += v;
    }
  }
  if (acc[0] == 12345.f) out[0] = acc[1];
}
extern "C" int run_segread(const float* in, float* out, int seg, int stride, long long rows_per_wave, int nblocks, int threads, void* stream) {
  hipLaunchKernelGGL(segread, dim3(nblocks), dim3(threads), 0, (hipStream_t)stream, in, out, seg, stride, rows_per_wave, 1);
  return (int)hipGetLastError();
}
