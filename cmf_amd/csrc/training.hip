// Small training-side kernels: pieces of the primal backward of a coupler network that are not convolutions.
#include "common.h"

namespace {

// ScaledTanh2dModule (networks.py:96-113): y = w tanh(u) + b and, for the tangent pass, g = w (1 - tanh(u)^2).
// Given the cotangents of y and of g (the latter from the coupling layer's cross terms):
//   du = dy g + dg dg/du,  dg/du = -2 tanh(u) g;   dw[c] += sum dy tanh + dg (1 - tanh^2);   db[c] += sum dy
// One block per channel (a handful of channels, B*HW elements each); tanh(u) is recovered as (y - b) / w.
__global__ __launch_bounds__(1024) void stanh_backward_kernel(const float* __restrict__ dy, const float* __restrict__ dg,
                                                              const float* __restrict__ y, const float* __restrict__ g,
                                                              const float* __restrict__ sw, const float* __restrict__ sb,
                                                              float* __restrict__ du, float* __restrict__ dsw,
                                                              float* __restrict__ dsb, int B, int C, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const float w = sw[c], b = sb[c], iw = w != 0.f ? 1.f / w : 0.f;
  float aw = 0.f, ab = 0.f;
  const long long n = (long long)B * HW;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const long long e = (i / HW) * C * HW + (long long)c * HW + i % HW;
    const float t = (y[e] - b) * iw, gv = g[e], dyv = dy[e], dgv = dg ? dg[e] : 0.f;
    du[e] = dyv * gv - 2.f * dgv * t * gv;
    aw += dyv * t + dgv * (gv * iw);                // 1 - tanh^2 = g / w: no cancellation when the tanh saturates
    ab += dyv;
  }
  aw = block_sum(aw, red);
  ab = block_sum(ab, red);
  if (threadIdx.x == 0) {
    if (dsw) dsw[c] += aw;
    if (dsb) dsb[c] += ab;
  }
}

// out[c] += sum over (n, px, col) of t(n, c, px, col): the bias gradient of a conv whose output cotangent is kept in a
// tangent-layout tensor (16 samples in the column slots for primal data).  One block per channel.
__global__ __launch_bounds__(1024) void channel_sum_kernel(const float* __restrict__ t, long long t_np, long long t_c,
                                                           long long t_px, long long t_sl, int np, int npx, int nc,
                                                           float* __restrict__ out) {
  __shared__ float red[16];
  const int c = blockIdx.x, nq = nc / 4;
  const long long total = (long long)np * npx * nq;
  float acc = 0.f;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    const int q = (int)(i % nq);
    const long long r = i / nq;
    const int px = (int)(r % npx);
    const long long n = r / npx;
    const f32x4 v = *reinterpret_cast<const f32x4*>(t + n * t_np + (long long)c * t_c + (long long)px * t_px + (q / 4) * t_sl + (q % 4) * 4);
    acc += (v.x + v.y) + (v.z + v.w);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[c] += acc;
}

// the same for up to CMF_WGRAD_MAX_BATCH tensors of one shape in one launch (blockIdx.y = tensor): the 2 K + 1 bias gradients of a
// coupler's primal backward were 2 K + 1 launches of ~5 us
struct ChannelSumBatch {
  const float* t[CMF_WGRAD_MAX_BATCH];
  float* out[CMF_WGRAD_MAX_BATCH];
};
__global__ __launch_bounds__(1024) void channel_sum_batched_kernel(ChannelSumBatch P, long long t_np, long long t_c, long long t_px,
                                                                   long long t_sl, int np, int npx, int nc) {
  __shared__ float red[16];
  const float* __restrict__ t = P.t[blockIdx.y];
  const int c = blockIdx.x, nq = nc / 4;
  const long long total = (long long)np * npx * nq;
  float acc = 0.f;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    const int q = (int)(i % nq);
    const long long r = i / nq;
    const int px = (int)(r % npx);
    const long long n = r / npx;
    const f32x4 v = *reinterpret_cast<const f32x4*>(t + n * t_np + (long long)c * t_c + (long long)px * t_px + (q / 4) * t_sl + (q % 4) * 4);
    acc += (v.x + v.y) + (v.z + v.w);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) P.out[blockIdx.y][c] += acc;
}

// MLP couplers (tanh layers): the tangent rule of layer i+1 reads phi_i = 1 - h_i^2 of the PRIMAL activation, so the reverse
// sweep has a second-order term.  Given the unmasked cotangent ct = W_{i+1}^T c_{i+1} (rows = features, fmajor / panel tangent
// layout) and the saved raw tangent x_i of the same rows:   c_i = phi_i ct  (in place),   dh_i[b][f] += -2 h_i sum_col ct x_i.
// One wavefront per (sample, feature).
__global__ __launch_bounds__(256) void tanh_cross_terms_kernel(float* __restrict__ c, long long c_b, long long c_r,
                                                               const float* __restrict__ x, long long x_b, long long x_r,
                                                               const float* __restrict__ h, float* __restrict__ dh, int F, int nc,
                                                               long long n_rows) {
  const long long bf = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bf >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int f = (int)(bf % F);
  const long long b = bf / F;
  const float hv = h[b * F + f], phi = 1.f - hv * hv;
  float* cp = c + b * c_b + (long long)f * c_r;
  const float* xp = x + b * x_b + (long long)f * x_r;
  float acc = 0.f;
  for (int k = lane; k < nc; k += 64) {
    const float ct = cp[k];
    acc += ct * xp[k];
    cp[k] = phi * ct;
  }
  acc = wave_sum(acc);
  if (lane == 0) dh[b * F + f] += -2.f * hv * acc;
}

// relu' bit mask of an activation tensor (B, C, HW) in the CMF_F_RELU_BITS layout [B][HW][C / 8]: bit j of a byte = [act > 0]
// of channel 8 * octet + j.  (The eval path gets these bits from the primal conv's epilogue; training keeps float activations
// for the weight gradients and derives the bits for the split-precision transposed convs here.)
__global__ __launch_bounds__(256) void relu_bits_kernel(const float* __restrict__ act, unsigned char* __restrict__ out, int C, int HW,
                                                        long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int noct = C / 8;
  const int px = (int)(i % HW);
  const long long r = i / HW;
  const int oct = (int)(r % noct);
  const long long b = r / noct;
  const float* p = act + (b * C + oct * 8) * (long long)HW + px;
  unsigned bits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) bits |= (p[(long long)j * HW] > 0.f ? 1u : 0u) << j;
  out[(b * HW + px) * noct + oct] = (unsigned char)bits;
}

// MLP coupler primal backward, elementwise stage (networks.py:206-224: h = tanh(W h' + b)): the cotangent of a hidden
// layer's pre-activation from the cotangent of its output, d = (dh + extra) (1 - a^2), a = the tanh output; `extra` (may be
// NULL) is the second-order term the tangent pass adds to dh (cmf_tanh_cross_terms).  Flat over n elements, any layout.
__global__ void tanh_backward_kernel(const float* __restrict__ dh, const float* __restrict__ a, const float* __restrict__ extra,
                                     long long n, float* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float av = a[i], e = extra ? extra[i] : 0.f;
    out[i] = (dh[i] + e) * (1.f - av * av);
  }
}

// AffineBijection backward (affine.py:24-34, u = x e^{ls} + sh, log-jac = sum ls): per feature f (one thread each, the
// batch loop reads coalesced across features)  g_ls[f] += sum_b dz x e^{ls} + sum_b dlj[b],  g_sh[f] += sum_b dz,
// dz <- dz e^{ls} in place.
__global__ void affine_prior_backward_kernel(float* __restrict__ dz, long long dz_b, const float* __restrict__ x, long long x_b,
                                             const float* __restrict__ log_scale, int n, int B, const float* __restrict__ dlj,
                                             float* __restrict__ g_ls, float* __restrict__ g_sh) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float es = expf(log_scale[f]);
  float als = 0.f, ash = 0.f, alj = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = dz[b * dz_b + f];
    als += d * x[b * x_b + f] * es;
    ash += d;
    if (dlj) alj += dlj[b];
    dz[b * dz_b + f] = d * es;
  }
  g_ls[f] += als + alj;
  g_sh[f] += ash;
}

}  // namespace

extern "C" int cmf_tanh_backward(const float* dh, const float* a, const float* extra, long long n, float* out, void* stream) {
  if (!dh || !a || !out || n <= 0) return CMF_EINVAL;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(tanh_backward_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, (hipStream_t)stream,
                     dh, a, extra, n, out);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_affine_prior_backward(float* dz, long long dz_b, const float* x, long long x_b, const float* log_scale, int n,
                                         int B, const float* dlj, float* g_ls, float* g_sh, void* stream) {
  if (!dz || !x || !log_scale || !g_ls || !g_sh || n <= 0 || B <= 0) return CMF_EINVAL;
  hipLaunchKernelGGL(affine_prior_backward_kernel, dim3(cmf_ceil_div(n, 64)), dim3(64), 0, (hipStream_t)stream, dz, dz_b, x, x_b,
                     log_scale, n, B, dlj, g_ls, g_sh);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_stanh_backward(const float* dy, const float* dg, const float* y, const float* g, const float* sw,
                                  const float* sb, float* du, float* dsw, float* dsb, int B, int C, int HW, void* stream) {
  if (!dy || !y || !g || !sw || !sb || !du || B <= 0 || C <= 0 || HW <= 0) return CMF_EINVAL;
  hipLaunchKernelGGL(stanh_backward_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, dy, dg, y, g, sw, sb, du, dsw, dsb, B, C, HW);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_channel_sum(const float* t, long long t_np, long long t_c, long long t_px, long long t_sl, int np, int C,
                               int npx, int nc, float* out, void* stream) {
  if (!t || !out || np <= 0 || C <= 0 || npx <= 0 || nc <= 0 || nc % 16) return CMF_EINVAL;
  if ((uintptr_t)t % 16 || (t_np | t_c | t_px | t_sl) % 4) return CMF_EINVAL;
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, t, t_np, t_c, t_px, t_sl ? t_sl : 16, np, npx,
                     nc, out);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_channel_sum_batched(const float* const* t, float* const* out, int n, long long t_np, long long t_c, long long t_px,
                                       long long t_sl, int np, int C, int npx, int nc, void* stream) {
  if (!t || !out || n < 1 || n > CMF_WGRAD_MAX_BATCH || np <= 0 || C <= 0 || npx <= 0 || nc <= 0 || nc % 16) return CMF_EINVAL;
  if ((t_np | t_c | t_px | t_sl) % 4) return CMF_EINVAL;
  ChannelSumBatch P;
  for (int i = 0; i < CMF_WGRAD_MAX_BATCH; ++i) {
    const int q = i < n ? i : 0;
    if (!t[q] || !out[q] || (uintptr_t)t[q] % 16) return CMF_EINVAL;
    P.t[i] = t[q], P.out[i] = out[q];
  }
  hipLaunchKernelGGL(channel_sum_batched_kernel, dim3(C, n), dim3(1024), 0, (hipStream_t)stream, P, t_np, t_c, t_px, t_sl ? t_sl : 16, np,
                     npx, nc);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_tanh_cross_terms(float* c, long long c_b, long long c_r, const float* x, long long x_b, long long x_r,
                                    const float* h, float* dh, int F, int B, int nc, void* stream) {
  if (!c || !x || !h || !dh || F <= 0 || B <= 0 || nc <= 0) return CMF_EINVAL;
  const long long n_rows = (long long)B * F;
  hipLaunchKernelGGL(tanh_cross_terms_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c, c_b, c_r, x,
                     x_b, x_r, h, dh, F, nc, n_rows);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_relu_bits(const float* act, void* out, int B, int C, int HW, void* stream) {
  if (!act || !out || B <= 0 || C <= 0 || C % 8 || HW <= 0) return CMF_EINVAL;
  const long long total = (long long)B * (C / 8) * HW;
  hipLaunchKernelGGL(relu_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, act,
                     (unsigned char*)out, C, HW, total);
  CMF_LAUNCH_CHECK();
  return 0;
}
