// Fused optimiser step over ONE flat fp32 buffer (all parameters of the density).
//
// The reference builds torch.optim.{SGD, Adam, Adamax}(params, lr, weight_decay) per objective and optionally clips the
// global gradient norm first (experiment.py:515-534, trainer.py:213-221).  Per-parameter optimiser loops are launch-bound
// for this model family (486 tensors, 24 MB in all): here parameters, gradients and both moments live in flat buffers
// (cmf_amd/optim.py re-points every Parameter into them), so a step is two launches -- the squared-norm reduction when
// clipping is on, and one elementwise pass that streams 4 arrays in and 3 out (HBM-bound: 28 B per parameter).
// Arithmetic follows torch's single-tensor kernels (torch/optim/{sgd,adam,adamax}.py, no amsgrad / momentum / maximize):
//   g  <- coef * g  (coef = min(1, max_norm / (||g|| + 1e-6)), clip_grad_norm_, in place);  below g means g + wd * p
//   SGD:    p <- p - lr g
//   Adam:   m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  p <- p - (lr / (1-b1^t)) m / (sqrt(v) / sqrt(1-b2^t) + eps)
//   Adamax: m <- b1 m + (1-b1) g;  u <- max(b2 u, |g| + eps);  p <- p - (lr / (1-b1^t)) m / u
#include "common.h"
#include <cmath>

namespace {

constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ ws) {
  __shared__ float red[16];
  float s = 0.f;
  const long long n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)NORM_BLOCKS * 256) {
    const f32x4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ ws, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < NORM_BLOCKS; i += 256) s += ws[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

struct StepArgs {
  float *p, *g, *m, *v;
  long long n;
  float lr, b1, b2, omb1, omb2, eps, wd, step_size, inv_sqrt_bc2, max_norm;   // omb = 1 - beta, rounded from double
  const float* sqnorm;
};

template <int KIND>
__device__ __forceinline__ void update(float& p, float& g, float& m, float& v, const StepArgs& a, float coef) {
  g *= coef;                                      // what clip_grad_norm_ leaves in .grad
  const float ge = g + a.wd * p;                  // weight decay is added to a temporary
  if (KIND == CMF_OPT_SGD) {
    p -= a.lr * ge;
  } else if (KIND == CMF_OPT_ADAM) {
    m = a.b1 * m + a.omb1 * ge;
    v = a.b2 * v + a.omb2 * ge * ge;
    p -= a.step_size * (m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
  } else {
    m = a.b1 * m + a.omb1 * ge;
    v = fmaxf(a.b2 * v, fabsf(ge) + a.eps);
    p -= a.step_size * (m / v);
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void optimizer_step_kernel(StepArgs a) {
  float coef = 1.f;
  if (a.sqnorm) coef = fminf(1.f, a.max_norm / (sqrtf(a.sqnorm[0]) + 1e-6f));
  const long long n4 = a.n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 p = reinterpret_cast<f32x4*>(a.p)[i], g = reinterpret_cast<f32x4*>(a.g)[i], m = {0, 0, 0, 0}, v = {0, 0, 0, 0};
    if (KIND != CMF_OPT_SGD) {
      m = reinterpret_cast<f32x4*>(a.m)[i];
      v = reinterpret_cast<f32x4*>(a.v)[i];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float pp = p[c], gg = g[c], mm = m[c], vv = v[c];
      update<KIND>(pp, gg, mm, vv, a, coef);
      p[c] = pp, g[c] = gg, m[c] = mm, v[c] = vv;
    }
    reinterpret_cast<f32x4*>(a.p)[i] = p;
    if (a.sqnorm) reinterpret_cast<f32x4*>(a.g)[i] = g;            // clip_grad_norm_ rescales the gradients in place
    if (KIND != CMF_OPT_SGD) {
      reinterpret_cast<f32x4*>(a.m)[i] = m;
      reinterpret_cast<f32x4*>(a.v)[i] = v;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float pp = a.p[i], gg = a.g[i], mm = KIND != CMF_OPT_SGD ? a.m[i] : 0.f, vv = KIND != CMF_OPT_SGD ? a.v[i] : 0.f;
    update<KIND>(pp, gg, mm, vv, a, coef);
    a.p[i] = pp;
    if (a.sqnorm) a.g[i] = gg;
    if (KIND != CMF_OPT_SGD) a.m[i] = mm, a.v[i] = vv;
  }
}

}  // namespace

extern "C" int cmf_grad_sqnorm(const float* g, long long n, float* ws, float* out, void* stream) {
  if (!g || !ws || !out || n <= 0 || (uintptr_t)g % 16) return CMF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, g, n, ws);
  CMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, ws, out);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_optimizer_step(int kind, float* p, float* g, float* m, float* v, long long n, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, int step, const float* sqnorm,
                                  float max_norm, void* stream) {
  if (!p || !g || n <= 0 || step < 1 || ((uintptr_t)p | (uintptr_t)g) % 16) return CMF_EINVAL;
  if (kind != CMF_OPT_SGD && (!m || !v || ((uintptr_t)m | (uintptr_t)v) % 16)) return CMF_EINVAL;
  // hyper-parameters arrive as doubles: torch forms 1 - beta and lr / (1 - beta^t) in double before rounding to fp32
  StepArgs a{p, g, m, v, n, (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
             (float)weight_decay, 0.f, 0.f, max_norm, sqnorm};
  if (kind != CMF_OPT_SGD) {                                       // bias corrections in double, like torch's python scalars
    a.step_size = (float)(lr / (1.0 - std::pow(beta1, step)));
    a.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(1.0 - std::pow(beta2, step)));
  }
  const int blocks = (int)std::min<long long>(2048, (n / 4 + 255) / 256 + 1);
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case CMF_OPT_SGD: hipLaunchKernelGGL(optimizer_step_kernel<CMF_OPT_SGD>, dim3(blocks), dim3(256), 0, s, a); break;
    case CMF_OPT_ADAM: hipLaunchKernelGGL(optimizer_step_kernel<CMF_OPT_ADAM>, dim3(blocks), dim3(256), 0, s, a); break;
    case CMF_OPT_ADAMAX: hipLaunchKernelGGL(optimizer_step_kernel<CMF_OPT_ADAMAX>, dim3(blocks), dim3(256), 0, s, a); break;
    default: return CMF_EINVAL;
  }
  CMF_LAUNCH_CHECK();
  return 0;
}
