// Weight gradient of the tangent convolution (the parameter-gradient half of the training step).
//
// Forward (conv_tangent.hip):  y(n, co, px, :) = sum_{ci,tap} W[co][ci][tap] * F(n, ci, px+tap) * x(n, ci, px+tap, :)
// Reverse wrt W:               dW[co][ci][tap] += sum_{n, px, col} gy(n, co, px, col) * F(n, ci, px+tap) * x(n, ci, px+tap, col)
//
// A GEMM with M = cout, N = cin * taps and K = samples * pixels * columns (C3 hidden conv: 64 x 576 x 25.7 M): as many
// FLOPs as the forward conv, output of a few hundred KB.  The reference gets it from autograd through the d
// column-by-column JVP graphs (jvp_layers.py:49-64 under loss.backward(), trainer.py:213).
//
// This first version: fp32 MFMA (v_mfma_f32_16x16x4_f32), K split over persistent workgroups.  One K block =
// (sample, pixel, 16-column slice); a lane loads 4 consecutive columns (16 B) of one channel, and the MFMA K index is
// spread over the slice as  k-group q = lane / 16, K-step j  <->  column 4 q + j  for BOTH operands, so the 16-byte
// loads feed four MFMAs without any shuffling.  A 512-thread workgroup keeps a 64 x 64 x taps block of dW in
// accumulators: wave w owns output-channel tile w & 3 and half of the (tap, input-channel tile) pairs (18 tiles = 72
// VGPRs for 3x3).  Operands come straight from global memory (the nine taps and the four co-tile waves re-read an
// input pixel from L1 / L2).  Partial blocks go to a workspace and a second kernel sums them in a fixed order and
// accumulates into dW (deterministic; no float atomics).  Larger cout / cin are tiled by the host over 64 x 64 blocks.
#include "common.h"

namespace {

constexpr int WG_MAX = 256;

__device__ __forceinline__ float factor_of(float f, int fmode) {
  switch (fmode) {
    case CMF_F_RELU: return f > 0.f ? 1.f : 0.f;
    case CMF_F_TANH: return 1.f - f * f;
    case CMF_F_RAW: return f;
    default: return 1.f;
  }
}

template <int TAPS>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(cmf_conv_tangent_args a, const float* __restrict__ gy,
                                                            float* __restrict__ ws, int co0, int ci0, long long nkb) {
  constexpr int PAIRS = TAPS == 9 ? 18 : 2;                        // (tap, ci tile) pairs per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = wave & 3, h = wave >> 2;
  const int r = lane & 15, q = lane >> 4;
  const int HW = a.H * a.W, nsl = a.nc / 16;
  const long long xsl = a.x_sl ? a.x_sl : 16, ysl = a.y_sl ? a.y_sl : 16;
  const int fgrp = a.f_group > 1 ? a.f_group : 1;
  const bool has_f = a.f != nullptr && a.fmode != CMF_F_NONE;

  const int co = co0 + c * 16 + r;
  const long long gy_lane = (long long)(co < a.cout ? co : a.cout - 1) * a.y_co + 4 * q;   // rows >= cout: dropped by the reduction
  long long x_lane[4], f_lane[4];
  bool ci_ok[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = ci0 + it * 16 + r, cic = ci < a.cin ? ci : a.cin - 1;
    ci_ok[it] = ci < a.cin;
    x_lane[it] = (long long)cic * a.x_ci + 4 * q;
    f_lane[it] = (long long)cic * a.f_ci;
  }

  f32x4 acc[PAIRS];
#pragma unroll
  for (int j = 0; j < PAIRS; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long per = (nkb + gridDim.x - 1) / gridDim.x;
  const long long kb0 = (long long)blockIdx.x * per, kb1 = kb0 + per < nkb ? kb0 + per : nkb;
  // K-block cursor (slice fastest, then x, y, sample): decoded once, then advanced with SALU compares
  int sl = 0, xx = 0, yy = 0, n = 0;
  if (kb0 < kb1) {
    sl = (int)(kb0 % nsl);
    const long long t = kb0 / nsl;
    const int px = (int)(t % HW);
    n = (int)(t / HW);
    yy = px / a.W;
    xx = px - yy * a.W;
  }
  for (long long kb = kb0; kb < kb1; ++kb) {
    const int px = yy * a.W + xx;
    // every load is unconditional (clamped address) and validity is a 0 / 1 multiplier: one basic block, so the
    // compiler can keep all of a K block's loads in flight ahead of its MFMAs
    const f32x4 g = *reinterpret_cast<const f32x4*>(gy + (long long)n * a.y_np + (long long)px * a.y_px + sl * ysl + gy_lane);
    const float* xb = a.x + (long long)n * a.x_np + sl * xsl;
    const float* fb = has_f ? a.f + (long long)(n / fgrp) * a.f_np + (n % fgrp) : a.x;
    f32x4 v[PAIRS];
    float m[PAIRS];
    bool ok[PAIRS];
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
      const int jp = h * PAIRS + j;
      const int tap = TAPS == 9 ? jp / 4 : 0, it = TAPS == 9 ? jp % 4 : jp;
      const int dy = TAPS == 9 ? tap / 3 - 1 : 0, dx = TAPS == 9 ? tap % 3 - 1 : 0;
      const int y2 = yy + dy, x2 = xx + dx;
      const bool in = y2 >= 0 && y2 < a.H && x2 >= 0 && x2 < a.W;  // wave-uniform
      const int p2 = in ? y2 * a.W + x2 : px;
      v[j] = *reinterpret_cast<const f32x4*>(xb + (long long)p2 * a.x_px + x_lane[it]);
      m[j] = has_f ? fb[(long long)p2 * a.f_px + f_lane[it]] : 1.f;
      ok[j] = in && ci_ok[it];
    }
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
      const float mj = has_f ? factor_of(m[j], a.fmode) : 1.f;
      f32x4 w;                                                     // a select, not a 0-multiplier: 0 * inf would be NaN
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = ok[j] ? v[j][k] * mj : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[k], w[k], acc[j], 0, 0, 0);
    }
    if (++sl == nsl) {
      sl = 0;
      if (++xx == a.W) {
        xx = 0;
        if (++yy == a.H) yy = 0, ++n;
      }
    }
  }

  // D[row = 4 q + i][col = r] = dW[co tile row][ci tile col] of pair j  ->  ws[wg][co 64][ci 64][TAPS]
  float* out = ws + (size_t)blockIdx.x * 64 * 64 * TAPS;
#pragma unroll
  for (int j = 0; j < PAIRS; ++j) {
    const int jp = h * PAIRS + j;
    const int tap = TAPS == 9 ? jp / 4 : 0, it = TAPS == 9 ? jp % 4 : jp;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[((c * 16 + 4 * q + i) * 64 + it * 16 + r) * TAPS + tap] = acc[j][i];
  }
}

// dw[(co0 + co)][ci0 + ci][tap] += sum_wg ws[wg][co][ci][tap]   (fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nwg,
                                                           int taps, int co0, int ci0, int cout, int cin) {
  const int e = blockIdx.x * 256 + threadIdx.x, per = 64 * 64 * taps;
  if (e >= per) return;
  const int tap = e % taps, ci = (e / taps) % 64 + ci0, co = e / (taps * 64) + co0;
  if (co >= cout || ci >= cin) return;
  float s = 0.f;
  for (int g = 0; g < nwg; ++g) s += ws[(size_t)g * per + e];
  dw[((size_t)co * cin + ci) * taps + tap] += s;
}

}  // namespace

extern "C" long long cmf_conv_tangent_wgrad_ws(const cmf_conv_tangent_args* a) {
  if (!a || (a->taps != 9 && a->taps != 1)) return 0;
  return (long long)WG_MAX * 64 * 64 * a->taps * sizeof(float);
}

extern "C" int cmf_conv_tangent_wgrad(const cmf_conv_tangent_args* a, const float* gy, float* dw, float* ws,
                                      long long ws_bytes, void* stream) {
  if (!a || !a->x || !gy || !dw || !ws) return CMF_EINVAL;
  if (a->taps != 9 && a->taps != 1) return CMF_EINVAL;
  if (a->np <= 0 || a->cin <= 0 || a->cout <= 0 || a->H <= 0 || a->W <= 0 || a->nc <= 0 || a->nc % 16) return CMF_EINVAL;
  if (a->fmode == CMF_F_RELU_BITS || a->fmode == CMF_F_SELF_RELU) return CMF_EINVAL;   // fp32 factor tensors only
  if (a->fmode != CMF_F_NONE && !a->f) return CMF_EINVAL;
  if (ws_bytes < cmf_conv_tangent_wgrad_ws(a)) return CMF_EINVAL;
  // 16-byte column quads on both operands
  if (((uintptr_t)a->x | (uintptr_t)gy) % 16 || (a->x_np | a->x_ci | a->x_px | a->x_sl | a->y_np | a->y_co | a->y_px | a->y_sl) % 4)
    return CMF_EINVAL;
  const long long nkb = (long long)a->np * a->H * a->W * (a->nc / 16);
  const int grid = (int)(nkb < WG_MAX ? nkb : WG_MAX);
  hipStream_t s = (hipStream_t)stream;
  for (int co0 = 0; co0 < a->cout; co0 += 64)
    for (int ci0 = 0; ci0 < a->cin; ci0 += 64) {
      if (a->taps == 9) hipLaunchKernelGGL(conv_wgrad_kernel<9>, dim3(grid), dim3(512), 0, s, *a, gy, ws, co0, ci0, nkb);
      else hipLaunchKernelGGL(conv_wgrad_kernel<1>, dim3(grid), dim3(512), 0, s, *a, gy, ws, co0, ci0, nkb);
      CMF_LAUNCH_CHECK();
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cmf_ceil_div(64 * 64 * a->taps, 256)), dim3(256), 0, s, ws, dw, grid,
                         a->taps, co0, ci0, a->cout, a->cin);
      CMF_LAUNCH_CHECK();
    }
  return 0;
}
