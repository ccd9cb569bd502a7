// Per-CU VMEM throughput for the tangent-conv ACCESS SHAPE: one wave-instruction = 16 segments of 64 B, each in a
// different channel plane (`plane` bytes apart), vs one contiguous KiB.  mode 0 load / 1 store, seg 0 contiguous / 1 planes.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void curate2(float* __restrict__ buf, float* __restrict__ out, int mode, int seg, long long plane, int iters) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  // wave-private region: 16 planes x (iters*8) pixels x 256 B (nc = 64 floats per pixel row); this wave's slice = 64 B of each row
  char* base = (char*)buf + wave * 16 * plane;
  const long long loff = seg ? (long long)(lane & 15) * plane + (lane >> 4) * 16 : (long long)lane * 16;
  f4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long pix = (long long)i * 8 + u;
      f4* p = (f4*)(base + loff + (seg ? pix * 256 : pix * 1024));
      if (mode == 0) acc += *p; else *p = acc;
    }
  }
  if (acc[0] == 12345.f) out[0] = acc[1];
}
extern "C" int run_curate2(float* buf, float* out, int mode, int seg, long long plane, int iters, int nblocks, int threads, void* stream) {
  hipLaunchKernelGGL(curate2, dim3(nblocks), dim3(threads), 0, (hipStream_t)stream, buf, out, mode, seg, plane, iters);
  return (int)hipGetLastError();
}
