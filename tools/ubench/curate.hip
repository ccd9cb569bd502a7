// Per-CU vector-memory throughput: NB workgroups (one per CU when NB <= #CUs) of `threads` threads; every wave streams
// `iters` x 8 KiB (8 dwordx4 wave-instructions of 1 KiB, fully coalesced) over a private window of `window` bytes
// (window << L2: L2-resident; window = whole stream: HBM).  mode 0 = loads, 1 = stores.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void curate(float* __restrict__ buf, float* __restrict__ out, int mode, long long window, int iters) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  char* base = (char*)buf + wave * window;
  f4 acc = {0, 0, 0, 0};
  long long off = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f4* p = (f4*)(base + off + u * 1024 + lane * 16);
      if (mode == 0) acc += *p; else *p = acc;
    }
    off += 8192;
    if (off >= window) off = 0;
  }
  if (acc[0] == 12345.f) out[0] = acc[1];
}
extern "C" int run_curate(float* buf, float* out, int mode, long long window, int iters, int nblocks, int threads, void* stream) {
  hipLaunchKernelGGL(curate, dim3(nblocks), dim3(threads), 0, (hipStream_t)stream, buf, out, mode, window, iters);
  return (int)hipGetLastError();
}
