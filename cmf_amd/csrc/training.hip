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

}  // namespace

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
