// NSF prior layers of the low-dimensional flow for gfx950: the elementwise monotone rational-quadratic
// spline with linear tails, the LU-parameterised linear map, and the MADE masks.
//
// Reference call sites: cmf/models/components/bijections/nsf.py:86-113 (MaskedPiecewiseRationalQuadraticAutoregressive-
// Transform: tails = 'linear', residual blocks, relu), bijections/linear.py:12-34 (LULinear, identity_init), config/
// schemas.py:87-103 (8 bins, tail bound 3).  The arithmetic lives in jrmcornish/nsf @ 8e3fe75, which is NOT vendored under
// the reference tree: PARITY UNPINNED.  Implemented from the published algorithm (Durkan, Bekasov, Murray, Papamakarios,
// "Neural Spline Flows", NeurIPS 2019, eqs. 4-8 and the public code base's nde/transforms/splines/rational_quadratic.py,
// lu.py, made.py); the CPU restatement it is tested against is oracle/cmf_oracle.py (rq_spline, lu_linear_matrices,
// made_masks).
//
// All three are HBM / latency-bound elementwise work on (B, d <= 128) tensors: one pass over the operands, no reuse.
#include "common.h"

namespace {

constexpr float MIN_BIN = 1e-3f, MIN_DER = 1e-3f;   // nsf defaults (min_bin_width / min_bin_height / min_derivative)
constexpr int MAX_BINS = 16;

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// One wavefront per sample, lanes stride over the features.  params: [B][D][3 bins - 1] = bins widths, bins heights,
// bins - 1 inner derivatives (the two boundary derivatives are the constant that makes the tails C^1: softplus^-1(1 - min)).
// out (may alias x) = spline(x) or spline^-1(x); lj[b] += sum_f log|d out / d x|.
__global__ __launch_bounds__(64) void rq_spline_kernel(const float* __restrict__ x, long long x_b, const float* __restrict__ params,
                                                        int D, int bins, float inv_sqrt_hidden, float tail, int inverse,
                                                        float* __restrict__ out, long long o_b, float* __restrict__ lj) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int K = 3 * bins - 1;
  float acc = 0.f;
  for (int f = lane; f < D; f += 64) {
    const float xv = x[b * x_b + f];
    float yv = xv, lad = 0.f;
    if (xv >= -tail && xv <= tail) {
      const float* p = params + ((long long)b * D + f) * K;
      float cw[MAX_BINS + 1], ch[MAX_BINS + 1], der[MAX_BINS + 1];
      // softmax -> bin widths / heights with the minimum size, cumulated into knot positions on [-tail, tail]
      float mw = -3.0e38f, mh = -3.0e38f;
      for (int i = 0; i < bins; ++i) {
        mw = fmaxf(mw, p[i] * inv_sqrt_hidden);
        mh = fmaxf(mh, p[bins + i] * inv_sqrt_hidden);
      }
      float sw = 0.f, sh = 0.f;
      for (int i = 0; i < bins; ++i) {
        cw[i + 1] = expf(p[i] * inv_sqrt_hidden - mw);
        ch[i + 1] = expf(p[bins + i] * inv_sqrt_hidden - mh);
        sw += cw[i + 1];
        sh += ch[i + 1];
      }
      const float scale = 1.f - MIN_BIN * bins;
      float aw = 0.f, ah = 0.f;
      cw[0] = ch[0] = -tail;
      for (int i = 1; i <= bins; ++i) {
        aw += MIN_BIN + scale * (cw[i] / sw);
        ah += MIN_BIN + scale * (ch[i] / sh);
        cw[i] = 2.f * tail * aw - tail;
        ch[i] = 2.f * tail * ah - tail;
      }
      cw[bins] = ch[bins] = tail;
      der[0] = der[bins] = 1.f;                     // min + softplus(log(exp(1 - min) - 1)) = 1 exactly in exact arithmetic
      for (int i = 1; i < bins; ++i) der[i] = MIN_DER + softplusf(p[2 * bins + i - 1]);
      // bin search: last knot k with x >= knot_k (the top knot is moved up by 1e-6 so that x == tail lands in the last bin)
      const float* loc = inverse ? ch : cw;
      int k = 0;
      for (int i = 1; i < bins; ++i) k += xv >= loc[i] ? 1 : 0;
      const float w = cw[k + 1] - cw[k], h = ch[k + 1] - ch[k], delta = h / w, d0 = der[k], d1 = der[k + 1];
      const float s2 = d0 + d1 - 2.f * delta;
      float th;
      if (inverse) {
        const float yy = xv - ch[k];
        const float a = yy * s2 + h * (delta - d0), bq = h * d0 - yy * s2, c = -delta * yy;
        th = (2.f * c) / (-bq - sqrtf(bq * bq - 4.f * a * c));
        yv = th * w + cw[k];
      } else {
        th = (xv - cw[k]) / w;
      }
      const float t1 = th * (1.f - th);
      const float den = delta + s2 * t1;
      if (!inverse) yv = ch[k] + h * (delta * th * th + d0 * t1) / den;
      const float dnum = delta * delta * (d1 * th * th + 2.f * delta * t1 + d0 * (1.f - th) * (1.f - th));
      lad = logf(dnum) - 2.f * logf(den);
      if (inverse) lad = -lad;
    }
    out[b * o_b + f] = yv;
    acc += lad;
  }
  acc = wave_sum(acc);
  if (lane == 0 && lj) lj[b] += acc;
}

// W = L U and logabsdet = sum log diag(U) from the LULinear parameters; entries in np.tril_indices(n, -1) / np.triu_indices(n, 1)
// order (row-major over the strict triangles), diag(U) = softplus(unconstrained) + eps.
__global__ __launch_bounds__(256) void lu_weights_kernel(const float* __restrict__ lower, const float* __restrict__ upper,
                                                          const float* __restrict__ udiag, int n, float eps, float* __restrict__ W,
                                                          float* __restrict__ logdet) {
  __shared__ float red[16];
  // strict-lower entry (i, j), j < i, sits at i (i - 1) / 2 + j; strict-upper (i, j), j > i, at i n - i (i + 1) / 2 + (j - i - 1)
  auto Lij = [&](int i, int j) { return j > i ? 0.f : (j == i ? 1.f : lower[i * (i - 1) / 2 + j]); };
  auto Uij = [&](int i, int j) {
    return j < i ? 0.f : (j == i ? softplusf(udiag[i]) + eps : upper[i * n - i * (i + 1) / 2 + (j - i - 1)]);
  };
  for (int idx = threadIdx.x; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    float acc = 0.f;
    const int kmax = i < j ? i : j;
    for (int k = 0; k <= kmax; ++k) acc += Lij(i, k) * Uij(k, j);
    W[idx] = acc;
  }
  float ld = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) ld += logf(softplusf(udiag[i]) + eps);
  ld = block_sum(ld, red);
  if (threadIdx.x == 0) logdet[0] = ld;
}

// out = w * mask with the MADE mask computed from the unit degrees (random_mask = False):
//   kind 0 (input -> hidden):  deg_out(o) = o % max(1, F - 1) + min(1, F - 1),  deg_in(i) = i + 1,          mask = out >= in
//   kind 1 (hidden -> hidden): both sides hidden degrees,                                                 mask = out >= in
//   kind 2 (hidden -> output): deg_out(o) = o / multiplier + 1 (each feature's parameters contiguous),     mask = out >  in
__global__ void made_mask_kernel(const float* __restrict__ w, float* __restrict__ out, int n_out, int n_in, int kind, int F,
                                 int multiplier) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_out * n_in) return;
  const int o = idx / n_in, i = idx % n_in;
  const int mx = F - 1 > 1 ? F - 1 : 1, mn = F - 1 < 1 ? F - 1 : 1;
  const int hid_o = o % mx + mn, hid_i = i % mx + mn;
  const int dout = kind == 2 ? o / multiplier + 1 : hid_o;
  const int din = kind == 0 ? i + 1 : hid_i;
  const bool keep = kind == 2 ? dout > din : dout >= din;
  out[idx] = keep ? w[idx] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Training: first derivatives of the spline by forward-mode dual numbers.  z(x, theta) and lad(x, theta) = log dz/dx are
// evaluated 3 bins times with one seeded input each (x and the 3 bins - 1 parameters of the feature): ~25 evaluations of a
// ~150-flop function per (sample, feature) on (B, d <= 128) tensors -- negligible next to the decode, and no hand-derived
// softmax / cumulative-sum / quotient-rule chains to get wrong.  The bin index is piecewise constant (as in autograd).
// ---------------------------------------------------------------------------------------------------------------------
struct Dual {
  float v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const float q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual dexp(Dual a) {
  const float e = expf(a.v);
  return {e, e * a.d};
}
__device__ __forceinline__ Dual dlog(Dual a) { return {logf(a.v), a.d / a.v}; }
__device__ __forceinline__ Dual dsoftplus(Dual a) {
  const float sg = 1.f / (1.f + expf(-a.v));        // d softplus = sigmoid
  return {softplusf(a.v), sg * a.d};
}

// forward spline on duals; seed: -1 = x, k >= 0 = parameter k.  Returns z and lad (inside the tails only; caller checks).
__device__ void spline_dual(float xv, const float* __restrict__ p, int bins, float isq, float tail, int seed, Dual& z, Dual& lad) {
  Dual cw[MAX_BINS + 1], ch[MAX_BINS + 1], der[MAX_BINS + 1];
  float mw = -3.0e38f, mh = -3.0e38f;
  for (int i = 0; i < bins; ++i) {
    mw = fmaxf(mw, p[i] * isq);
    mh = fmaxf(mh, p[bins + i] * isq);
  }
  auto par = [&](int k, float scale) { return Dual{p[k] * scale, seed == k ? scale : 0.f}; };
  Dual sw{0.f, 0.f}, sh{0.f, 0.f};
  for (int i = 0; i < bins; ++i) {
    cw[i + 1] = dexp(par(i, isq) - Dual{mw, 0.f});
    ch[i + 1] = dexp(par(bins + i, isq) - Dual{mh, 0.f});
    sw = sw + cw[i + 1];
    sh = sh + ch[i + 1];
  }
  const float scale = 1.f - MIN_BIN * bins;
  Dual aw{0.f, 0.f}, ah{0.f, 0.f};
  cw[0] = ch[0] = Dual{-tail, 0.f};
  for (int i = 1; i <= bins; ++i) {
    aw = aw + (MIN_BIN + scale * (cw[i] / sw));
    ah = ah + (MIN_BIN + scale * (ch[i] / sh));
    cw[i] = (2.f * tail) * aw - Dual{tail, 0.f};
    ch[i] = (2.f * tail) * ah - Dual{tail, 0.f};
  }
  cw[bins] = ch[bins] = Dual{tail, 0.f};
  der[0] = der[bins] = Dual{1.f, 0.f};
  for (int i = 1; i < bins; ++i) der[i] = MIN_DER + dsoftplus(par(2 * bins + i - 1, 1.f));
  int k = 0;
  for (int i = 1; i < bins; ++i) k += xv >= cw[i].v ? 1 : 0;
  const Dual x{xv, seed < 0 ? 1.f : 0.f};
  const Dual w = cw[k + 1] - cw[k], h = ch[k + 1] - ch[k], delta = h / w, d0 = der[k], d1 = der[k + 1];
  const Dual s2 = d0 + d1 - 2.f * delta;
  const Dual th = (x - cw[k]) / w;
  const Dual one{1.f, 0.f};
  const Dual t1 = th * (one - th);
  const Dual den = delta + s2 * t1;
  z = ch[k] + h * (delta * th * th + d0 * t1) / den;
  const Dual dnum = delta * delta * (d1 * th * th + 2.f * (delta * t1) + d0 * (one - th) * (one - th));
  lad = dlog(dnum) - 2.f * dlog(den);
}

// dx[b][f] = dz dz/dx + dlj[b] dlad/dx;  dparams[b][f][k] = dz dz/dtheta_k + dlj[b] dlad/dtheta_k.  One thread per (b, f, seed).
__global__ void rq_spline_backward_kernel(const float* __restrict__ x, long long x_b, const float* __restrict__ params, int D,
                                          int bins, float isq, float tail, int B, const float* __restrict__ dz, long long dz_b,
                                          const float* __restrict__ dlj, float* __restrict__ dx, long long dx_b,
                                          float* __restrict__ dparams) {
  const int K = 3 * bins - 1, S = K + 1;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * D * S) return;
  const int seed = (int)(i % S) - 1;                 // -1 = x
  const long long bf = i / S;
  const int f = (int)(bf % D);
  const long long b = bf / D;
  const float xv = x[b * x_b + f];
  const float gz = dz[b * dz_b + f], gl = dlj ? dlj[b] : 0.f;
  float out;
  if (xv >= -tail && xv <= tail) {
    Dual z, lad;
    spline_dual(xv, params + (b * D + f) * K, bins, isq, tail, seed, z, lad);
    out = gz * z.d + gl * lad.d;
  } else {
    out = seed < 0 ? gz : 0.f;                       // identity tails: dz/dx = 1, lad = 0, no parameter dependence
  }
  if (seed < 0) dx[b * dx_b + f] = out;
  else dparams[(b * D + f) * K + seed] = out;
}

// LULinear backward from dW = sum_b dy (x) x (n x n):  dL = dW U^T (strict lower),  dU = L^T dW (upper incl. diagonal);
// diag(U) = softplus(u) + eps  ->  g_u[i] += (dU_ii + dlj_sum / diag_i) sigmoid(u_i)   (log-jac = sum log diag(U)).
__global__ __launch_bounds__(256) void lu_backward_kernel(const float* __restrict__ dW, const float* __restrict__ lower,
                                                           const float* __restrict__ upper, const float* __restrict__ udiag, int n,
                                                           float eps, const float* __restrict__ dlj_sum, float* __restrict__ g_lower,
                                                           float* __restrict__ g_upper, float* __restrict__ g_udiag) {
  auto Lij = [&](int i, int j) { return j > i ? 0.f : (j == i ? 1.f : lower[i * (i - 1) / 2 + j]); };
  auto Uij = [&](int i, int j) {
    return j < i ? 0.f : (j == i ? softplusf(udiag[i]) + eps : upper[i * n - i * (i + 1) / 2 + (j - i - 1)]);
  };
  const float gl = dlj_sum ? dlj_sum[0] : 0.f;
  for (int idx = threadIdx.x; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    if (j < i) {                                     // dL_ij = sum_k dW_ik U_jk
      float acc = 0.f;
      for (int k = j; k < n; ++k) acc += dW[i * n + k] * Uij(j, k);
      g_lower[i * (i - 1) / 2 + j] += acc;
    } else {                                         // dU_ij = sum_k L_ki dW_kj
      float acc = 0.f;
      for (int k = i; k < n; ++k) acc += Lij(k, i) * dW[k * n + j];
      if (j == i) {
        const float sg = 1.f / (1.f + expf(-udiag[i]));
        g_udiag[i] += (acc + gl / (softplusf(udiag[i]) + eps)) * sg;
      } else {
        g_upper[i * n - i * (i + 1) / 2 + (j - i - 1)] += acc;
      }
    }
  }
}

// du[b][f] = -dlow[b] u[b][f]: cotangent of log N(u; 0, I) scaled by the cotangent of the low-dimensional elbo
__global__ void gaussian_backward_kernel(const float* __restrict__ u, const float* __restrict__ dlow, int n, long long total,
                                         float* __restrict__ du) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) du[i] = -dlow[i / n] * u[i];
}

// single-block sum of B floats (the LULinear log-jac cotangent: the same constant is added to every sample's log-jac)
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = acc;
}

}  // namespace

extern "C" int cmf_rq_spline_backward(const float* x, long long x_b, const float* params, int D, int bins, int hidden,
                                      float tail_bound, int B, const float* dz, long long dz_b, const float* dlj, float* dx,
                                      long long dx_b, float* dparams, void* stream) {
  if (!x || !params || !dz || !dx || !dparams || D <= 0 || B <= 0 || bins < 2 || bins > MAX_BINS || hidden <= 0 || !(tail_bound > 0.f))
    return CMF_EINVAL;
  const long long total = (long long)B * D * (3 * bins);
  if (total > 0x7fffffffLL * 256) return CMF_ERANGE;
  hipLaunchKernelGGL(rq_spline_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_b,
                     params, D, bins, 1.f / sqrtf((float)hidden), tail_bound, B, dz, dz_b, dlj, dx, dx_b, dparams);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_lu_backward(const float* dW, const float* lower, const float* upper, const float* unconstrained_diag, int n,
                               float eps, const float* dlj, int B, float* scratch1, float* g_lower, float* g_upper, float* g_udiag,
                               void* stream) {
  if (!dW || !unconstrained_diag || !g_udiag || n <= 0 || n > 1024 || (n > 1 && (!lower || !upper || !g_lower || !g_upper)))
    return CMF_EINVAL;
  if (dlj && (!scratch1 || B <= 0)) return CMF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dlj) {
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, dlj, B, scratch1);
    CMF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(lu_backward_kernel, dim3(1), dim3(256), 0, s, dW, lower, upper, unconstrained_diag, n, eps,
                     dlj ? scratch1 : nullptr, g_lower, g_upper, g_udiag);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_gaussian_backward(const float* u, const float* dlow, int n, int B, float* du, void* stream) {
  if (!u || !dlow || !du || n <= 0 || B <= 0) return CMF_EINVAL;
  const long long total = (long long)n * B;
  hipLaunchKernelGGL(gaussian_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, dlow, n,
                     total, du);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_rq_spline(const float* x, long long x_b, const float* params, int D, int bins, int hidden, float tail_bound,
                             int inverse, int B, float* out, long long out_b, float* lj, void* stream) {
  if (!x || !params || !out || D <= 0 || B <= 0 || bins < 2 || bins > MAX_BINS || hidden <= 0 || !(tail_bound > 0.f))
    return CMF_EINVAL;
  hipLaunchKernelGGL(rq_spline_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, x_b, params, D, bins,
                     1.f / sqrtf((float)hidden), tail_bound, inverse, out, out_b, lj);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_lu_weights(const float* lower, const float* upper, const float* unconstrained_diag, int n, float eps, float* W,
                              float* logdet, void* stream) {
  if (!unconstrained_diag || !W || !logdet || n <= 0 || n > 1024 || (n > 1 && (!lower || !upper))) return CMF_EINVAL;
  hipLaunchKernelGGL(lu_weights_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lower, upper, unconstrained_diag, n, eps, W,
                     logdet);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_made_mask_weight(const float* w, float* out, int n_out, int n_in, int kind, int features, int multiplier,
                                    void* stream) {
  if (!w || !out || n_out <= 0 || n_in <= 0 || kind < 0 || kind > 2 || features <= 0 || multiplier <= 0) return CMF_EINVAL;
  const long long total = (long long)n_out * n_in;
  if (total > 0x7fffffffLL) return CMF_ERANGE;
  hipLaunchKernelGGL(made_mask_kernel, dim3(cmf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, out, n_out, n_in,
                     kind, features, multiplier);
  CMF_LAUNCH_CHECK();
  return 0;
}
