// Reverse of the Gram / log-det / metric-L1 head: the cotangent of the Jacobian stack.
//
// The training loss of the reference differentiates through  jtj = J^T J, logdet = 2 sum log diag chol(jtj)
// and the metric terms sum |jtj_ij| (non_square.py:280-294, :307-308, :87-100) with torch.autograd.  Here the
// same derivative is one kernel per sample:
//
//   dG = g_logdet * G^-1 + g_l1off * sign(G)[i != j] + g_l1diag * sign(G)[i == j]      (symmetric)
//   dJ = J (dG + dG^T) = 2 J dG
//
// G is the matrix the forward kernel left in `jtj` (jitter included: a constant shift of the diagonal has no
// derivative of its own).  G^-1 comes from an in-place Gauss-Jordan sweep in LDS (no pivoting: G is SPD,
// the forward pass has already factorised it), the product runs on the vector ALUs: 2 D d^2 FLOP per sample
// is < 1 % of the reverse sweep it feeds (profiles/LABBOOK.md section 9, f1), so this kernel is written for clarity, not MFMA.
#include "common.h"

namespace {

constexpr int SLAB = 32;     // rows of J staged per pass

__global__ __launch_bounds__(256) void gram_backward_kernel(const float* __restrict__ t, long long t_b, long long t_r,
                                                            int n_rows, int nc, int d, const float* __restrict__ jtj,
                                                            const float* __restrict__ g_logdet,
                                                            const float* __restrict__ g_l1off,
                                                            const float* __restrict__ g_l1diag, float* __restrict__ dt,
                                                            long long dt_b, long long dt_r, const float* __restrict__ m_in) {
  extern __shared__ __align__(16) float lds[];
  const int LD = nc + 4;                    // 16-byte aligned rows for the b128 reads of the product phase
  float* A = lds;                           // [d][LD]   G -> G^-1 -> M = 2 dG  (columns >= d stay 0)
  float* rowk = A + (size_t)d * LD;         // [nc]
  float* colk = rowk + nc;                  // [nc]
  float* Js = colk + nc;                    // [SLAB][nc]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  const float* G = jtj + (size_t)b * d * d;

  if (m_in) {
    // explicit cotangent of the Gram matrix (Hutchinson surrogate: M = mean_s u_s eps_s^T, not symmetric): dJ = J (M + M^T)
    const float* M = m_in + (size_t)b * d * d;
    for (int i = ti; i < d; i += 16)
      for (int j = tj; j < LD; j += 16) A[i * LD + j] = j < d ? M[i * d + j] + M[j * d + i] : 0.f;
    __syncthreads();
  } else {
  for (int i = ti; i < d; i += 16)
    for (int j = tj; j < LD; j += 16) A[i * LD + j] = j < d ? G[i * d + j] : 0.f;
  __syncthreads();

  for (int k = 0; k < d; ++k) {
    const float p = 1.f / A[k * LD + k];
    if (tid < d) {
      rowk[tid] = A[k * LD + tid] * p;
      colk[tid] = A[tid * LD + k];
    }
    __syncthreads();
    for (int i = ti; i < d; i += 16) {
      const float f = colk[i];
      for (int j = tj; j < d; j += 16) {
        float v;
        if (i == k) v = j == k ? p : rowk[j];
        else if (j == k) v = -f * p;
        else v = A[i * LD + j] - f * rowk[j];
        A[i * LD + j] = v;
      }
    }
    __syncthreads();
  }

  // M = 2 dG in place (elementwise: no hazard).  The sweep's rounding asymmetry (a few ulp) is left as it is.
  const float ga = g_logdet ? g_logdet[b] : 0.f;
  const float go = g_l1off ? g_l1off[b] : 0.f;
  const float gd = g_l1diag ? g_l1diag[b] : 0.f;
  for (int i = ti; i < d; i += 16)
    for (int j = tj; j < d; j += 16) {
      const float g = G[i * d + j];
      const float sg = g > 0.f ? 1.f : g < 0.f ? -1.f : 0.f;
      A[i * LD + j] = 2.f * (ga * A[i * LD + j] + (i == j ? gd : go) * sg);
    }
  __syncthreads();
  }

  // dJ = J M, SLAB rows at a time: thread (rg, q) owns columns 4q..4q+3 of rows rg, rg + rpp, ...
  const int nq = nc >> 2, rpp = 256 / nq;
  const int q = tid % nq, rg = tid / nq;
  const float* tb = t + (size_t)b * t_b;
  float* db = dt + (size_t)b * dt_b;
  for (int r0 = 0; r0 < n_rows; r0 += SLAB) {
    const int nr = min(SLAB, n_rows - r0);
    for (int e = tid; e < nr * nq; e += 256) {
      const int r = e / nq, c = e - r * nq;
      reinterpret_cast<f32x4*>(Js)[r * nq + c] = *reinterpret_cast<const f32x4*>(tb + (size_t)(r0 + r) * t_r + 4 * c);
    }
    __syncthreads();
    if (rg < rpp)
      for (int r = rg; r < nr; r += rpp) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* jr = Js + r * nc;
        for (int k = 0; k < d; ++k) acc += jr[k] * *reinterpret_cast<const f32x4*>(A + k * LD + 4 * q);
        *reinterpret_cast<f32x4*>(db + (size_t)(r0 + r) * dt_r + 4 * q) = acc;
      }
    __syncthreads();
  }
}

}  // namespace

extern "C" int cmf_gram_backward(const float* t, long long t_b, long long t_r, int n_rows, int nc, int d, int B,
                                 const float* jtj, const float* g_logdet, const float* g_l1off,
                                 const float* g_l1diag, float* dt, long long dt_b, long long dt_r, void* stream) {
  if (!t || !jtj || !dt) return CMF_EINVAL;
  if (n_rows <= 0 || B <= 0 || d <= 0 || d > nc || nc % 16 || nc > 128) return CMF_EINVAL;
  if ((t_b | t_r | dt_b | dt_r) % 4 || (uintptr_t)t % 16 || (uintptr_t)dt % 16) return CMF_EINVAL;
  const size_t lds = ((size_t)d * (nc + 4) + 2 * nc + (size_t)SLAB * nc) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = cmf_set_dynamic_lds((const void*)gram_backward_kernel, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(gram_backward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, t, t_b, t_r, n_rows, nc, d, jtj,
                     g_logdet, g_l1off, g_l1diag, dt, dt_b, dt_r, (const float*)nullptr);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_gram_backward_matrix(const float* t, long long t_b, long long t_r, int n_rows, int nc, int d, int B,
                                        const float* m, float* dt, long long dt_b, long long dt_r, void* stream) {
  if (!t || !m || !dt) return CMF_EINVAL;
  if (n_rows <= 0 || B <= 0 || d <= 0 || d > nc || nc % 16 || nc > 128) return CMF_EINVAL;
  if ((t_b | t_r | dt_b | dt_r) % 4 || (uintptr_t)t % 16 || (uintptr_t)dt % 16) return CMF_EINVAL;
  const size_t lds = ((size_t)d * (nc + 4) + 2 * nc + (size_t)SLAB * nc) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = cmf_set_dynamic_lds((const void*)gram_backward_kernel, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(gram_backward_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, t, t_b, t_r, n_rows, nc, d, m,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dt, dt_b, dt_r, m);
  CMF_LAUNCH_CHECK();
  return 0;
}
