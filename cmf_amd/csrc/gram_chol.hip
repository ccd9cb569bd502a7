// Fused J^T J Gram (fp32 MFMA) + Cholesky + log-det + g_ij / g_kk L1 terms for gfx950 (MI355X).
//
// Replaces, per sample, the reference's  jac = stack(cols, 2); bmm(jac^T, jac)
// (cmf/models/components/densities/non_square.py:307-308),  torch.linalg.cholesky + 2*sum(log diag)
// (:280-294) and the metric regularisers  sum_{i!=j}|G_ij| / sum_k|G_kk|  (:87-100).
//
// One 256-thread workgroup per sample.  The J panel T(b, r, 0:nc) is streamed once from HBM in
// 32-row slabs through LDS (register prefetch of the next slab under the MFMAs of the current one);
// G = J^T J is accumulated as NT x NT tiles of v_mfma_f32_16x16x4_f32 (A and B operands are the SAME
// slab: A[i][k] = J[k][i], B[k][j] = J[k][j], both conflict-free ds_read_b32 of 16 consecutive
// columns), parked in LDS, factorised in place by a right-looking Cholesky, and reduced with
// wavefront shuffles.  Algorithmic traffic: (n_rows*nc + d*d + 3) * 4 bytes per sample.
//
// Whole-batch jitter retry (non_square.py:284-288) without a host round trip: attempt 0 raises
// fail[0] when any sample hits a non-positive pivot; cmf_cholesky_retry(a) is enqueued unconditionally
// and exits at once unless fail[a-1] is set.
#include "common.h"

namespace {

constexpr int KC = 32;  // J rows per LDS slab


// In-place lower Cholesky of the leading d x d block of G (row stride ldg) by the whole workgroup.
// Returns 0 or (k+1) for a non-positive / non-finite pivot at column k; *logdet = 2*sum(log L_kk).
__device__ int block_cholesky(float* G, int ldg, int d, float* logdet) {
  const int tid = threadIdx.x, nt = blockDim.x;
  float ld = 0.f;
  int info = 0;
  for (int k = 0; k < d; ++k) {
    const float piv = G[k * ldg + k];
    if (!(piv > 0.f) || !(piv < 3.0e38f)) {   // uniform: every thread reads the same LDS word
      info = k + 1;
      break;
    }
    const float lkk = sqrtf(piv);
    ld += logf(lkk);
    const float inv = 1.f / lkk;
    __syncthreads();                           // everyone has read the pivot before it is overwritten
    for (int i = k + 1 + tid; i < d; i += nt) G[i * ldg + k] *= inv;
    if (tid == 0) G[k * ldg + k] = lkk;
    __syncthreads();
    const int m = d - k - 1;
    for (int idx = tid; idx < m * m; idx += nt) {
      const int i = k + 1 + idx / m, j = k + 1 + idx % m;
      if (j <= i) G[i * ldg + j] -= G[i * ldg + k] * G[j * ldg + k];
    }
    __syncthreads();
  }
  *logdet = 2.f * ld;
  return info;
}

__device__ void report(int b, int info, float logdet, float* logdet_out, int* info_out, int* fail_slot) {
  if (threadIdx.x == 0) {
    info_out[b] = info;
    logdet_out[b] = info ? __builtin_nanf("") : logdet;
    if (info) atomicOr(fail_slot, 1);
  }
}

template <int NT>
__global__ __launch_bounds__(256) void gram_chol_kernel(const float* __restrict__ t, long long t_b, long long t_r,
                                                         int n_rows, int d, float* __restrict__ jtj,
                                                         float* __restrict__ logdet, float* __restrict__ l1_off,
                                                         float* __restrict__ l1_diag, int* __restrict__ info,
                                                         int* __restrict__ fail) {
  constexpr int NC = NT * 16;
  constexpr int LDJ = (NC % 32 == 0) ? NC + 16 : NC;
  constexpr int LDG = NC + 1;
  constexpr int NI = (NT + 3) / 4;                 // row tiles per wave
  constexpr int ITEMS = KC * NC / 4;               // float4 per slab
  constexpr int NIT = (ITEMS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Js = smem;                                // [KC][LDJ]
  float* G = smem + KC * LDJ;                      // [NC][LDG]
  __shared__ float red[16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;
  const int b = blockIdx.x;
  const float* tb = t + (long long)b * t_b;

  f32x4 acc[NI][NT];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 pr[NIT];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + 256 * it;
      i = i < ITEMS ? i : ITEMS - 1;
      const int row = i / (NC / 4), c4 = i % (NC / 4);
      const bool ok = (k0 + row) < n_rows;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (ok ? (long long)(k0 + row) * t_r : 0) + c4 * 4);
      pr[it] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 256 * it;
      if (i < ITEMS) *reinterpret_cast<f32x4*>(Js + (i / (NC / 4)) * LDJ + (i % (NC / 4)) * 4) = pr[it];
    }
  };

  prefetch(0);
  for (int k0 = 0; k0 < n_rows; k0 += KC) {
    commit();
    __syncthreads();
    if (k0 + KC < n_rows) prefetch(k0 + KC);
#pragma unroll
    for (int kg = 0; kg < KC / 4; ++kg) {
      const float* row = Js + (kg * 4 + kq) * LDJ + cl;
      float bv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = row[j * 16];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int it = wave + 4 * i;
        if (it < NT) {                             // wave-uniform
          const float av = row[it * 16];
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // park G in LDS: D[row = kq*4 + r][col = cl]
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int it = wave + 4 * i;
    if (it < NT) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) G[(it * 16 + kq * 4 + r) * LDG + j * 16 + cl] = acc[i][j][r];
    }
  }
  __syncthreads();

  // J^T J out + L1 terms (before the factorisation overwrites the lower triangle)
  float so = 0.f, sd = 0.f;
  float* gout = jtj + (long long)b * d * d;
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx % d;
    const float v = G[i * LDG + j];
    gout[idx] = v;
    if (i == j) sd += fabsf(v); else so += fabsf(v);
  }
  so = block_sum(so, red);
  sd = block_sum(sd, red);
  if (tid == 0) {
    l1_off[b] = so;
    l1_diag[b] = sd;
  }
  float ld;
  const int inf = block_cholesky(G, LDG, d, &ld);
  report(b, inf, ld, logdet, info, fail + 0);
}

__global__ __launch_bounds__(256) void chol_retry_kernel(float* __restrict__ jtj, int d, int attempt, float eps,
                                                          float* __restrict__ logdet, float* __restrict__ l1_diag,
                                                          int* __restrict__ info, int* __restrict__ fail) {
  if (fail[attempt - 1] == 0) return;              // previous attempt succeeded for the whole batch
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[16];
  float* G = smem;
  const int ldg = d + 1, tid = threadIdx.x, b = blockIdx.x;
  float* gb = jtj + (long long)b * d * d;
  float sd = 0.f;
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx % d;
    float v = gb[idx];
    if (i == j) {                                   // jitter EVERY sample: non_square.py:286
      v += eps;
      gb[idx] = v;
      sd += fabsf(v);
    }
    G[i * ldg + j] = v;
  }
  sd = block_sum(sd, red);
  if (tid == 0) l1_diag[b] = sd;
  __syncthreads();
  float ld;
  const int inf = block_cholesky(G, ldg, d, &ld);
  report(b, inf, ld, logdet, info, fail + attempt);
}

__global__ void zero_flags_kernel(int* __restrict__ fail) {
  if (threadIdx.x < 8) fail[threadIdx.x] = 0;
}

template <int NT>
int launch_gram(const float* t, long long t_b, long long t_r, int n_rows, int d, int B, float* jtj, float* logdet,
                float* l1_off, float* l1_diag, int* info, int* fail, hipStream_t s) {
  constexpr int NC = NT * 16;
  constexpr int LDJ = (NC % 32 == 0) ? NC + 16 : NC;
  const size_t lds = (size_t)(KC * LDJ + NC * (NC + 1)) * sizeof(float);
  auto k = gram_chol_kernel<NT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(B), dim3(256), lds, s, t, t_b, t_r, n_rows, d, jtj, logdet, l1_off, l1_diag, info, fail);
  CMF_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int cmf_gram_cholesky(const float* t, long long t_b, long long t_r, int n_rows, int nc, int d, int B,
                                 float* jtj, float* logdet, float* l1_off, float* l1_diag, int* info, int* fail,
                                 void* stream) {
  if (!t || !jtj || !logdet || !l1_off || !l1_diag || !info || !fail) return CMF_EINVAL;
  if (n_rows <= 0 || B <= 0 || d <= 0 || d > nc || nc % 16 || nc > 128) return CMF_EINVAL;
  if ((t_b | t_r) % 4 || (uintptr_t)t % 16) return CMF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // A kernel node, not hipMemsetAsync: a captured 32-byte memset node replayed with garbage from the second
  // hipGraphLaunch on (ROCm 7.2, observed as pointer-like values in the flags), which silently armed every retry.
  hipLaunchKernelGGL(zero_flags_kernel, dim3(1), dim3(64), 0, s, fail);
  CMF_LAUNCH_CHECK();
#define CMF_GRAM_CASE(N) \
  case N: return launch_gram<N>(t, t_b, t_r, n_rows, d, B, jtj, logdet, l1_off, l1_diag, info, fail, s);
  switch (nc / 16) {
    CMF_GRAM_CASE(1) CMF_GRAM_CASE(2) CMF_GRAM_CASE(3) CMF_GRAM_CASE(4)
    CMF_GRAM_CASE(5) CMF_GRAM_CASE(6) CMF_GRAM_CASE(7) CMF_GRAM_CASE(8)
  }
#undef CMF_GRAM_CASE
  return CMF_EINVAL;
}

extern "C" int cmf_cholesky_retry(float* jtj, int d, int B, int attempt, float eps0, float* logdet, float* l1_diag,
                                  int* info, int* fail, void* stream) {
  if (!jtj || !logdet || !l1_diag || !info || !fail || d <= 0 || d > 128 || B <= 0 || attempt < 1 || attempt > 7)
    return CMF_EINVAL;
  float eps = eps0;
  for (int i = 1; i < attempt; ++i) eps *= 10.f;
  const size_t lds = (size_t)d * (d + 1) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)chol_retry_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(chol_retry_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, jtj, d, attempt, eps, logdet,
                     l1_diag, info, fail);
  CMF_LAUNCH_CHECK();
  return 0;
}
