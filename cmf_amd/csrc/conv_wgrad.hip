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
#include <cstdlib>
#include <type_traits>

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
  // a wave whose output-channel tile or whose input-channel tiles lie entirely past cout / cin has nothing to add (the
  // reduction never reads those entries): the 1 -> 64 / 64 -> 2 convs at the ends of a coupler keep 1 - 4 of 8 waves
  if (co0 + c * 16 >= a.cout || ci0 + (TAPS == 9 ? 0 : h * PAIRS) * 16 >= a.cin) return;
  const int r = lane & 15, q = lane >> 4;
  const int HW = a.H * a.W, nsl = a.nc / 16;
  const long long xsl = a.x_sl ? a.x_sl : 16, ysl = a.y_sl ? a.y_sl : 16;
  const int fgrp = a.f_group > 1 ? a.f_group : 1;
  const bool self = a.fmode == CMF_F_SELF_RELU;                    // x's own relu, elementwise (primal data in the column slots)
  const bool has_f = a.f != nullptr && a.fmode != CMF_F_NONE && !self;

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
      for (int k = 0; k < 4; ++k) w[k] = ok[j] ? (self ? fmaxf(v[j][k], 0.f) : v[j][k] * mj) : 0.f;
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

// THIN shapes: the first 3x3 conv of a coupler (1 .. 7 input channels -> 64) and its last 1x1 conv (64 -> 2 .. 16 channels).  In
// the kernels above a wave owns whole 16-channel tiles, so these shapes keep one to four waves of a workgroup busy, on MFMA tiles
// that are 1/16 .. 1/4 real (3x3, one input channel: 72 MFMAs per K block for 9 useful columns) -- 0.75 - 1.5 TB/s on what is a
// pure streaming job (one pass over the 64-channel tensor).  Here
//   * the N index of a 3x3 block is n = ci * 9 + tap (NT = ceil(9 cin / 16) tiles instead of 9 tiles of one or two real columns):
//     lane r of tile t fetches ITS tap's pixel of ITS channel -- a per-lane address, still one 16-byte load;
//   * all eight waves split K (wave w takes the workgroup's K blocks w, w + 8, ...) and each keeps the whole MT x NT block (at most
//     16 accumulator tiles); the eight partial blocks are summed through LDS at the end, in a fixed order.
template <int TAPS, int MT, int NT>
__global__ __launch_bounds__(512, 2) void conv_wgrad_thin_kernel(cmf_conv_tangent_args a, const float* __restrict__ gy,
                                                                 float* __restrict__ ws, int co0, int ci0, long long nkb) {
  extern __shared__ __attribute__((aligned(16))) float thin_part[];     // [8 waves][MT * NT tiles][256]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int W = a.W, HW = a.H * a.W, nsl = a.nc / 16;
  const long long xsl = a.x_sl ? a.x_sl : 16, ysl = a.y_sl ? a.y_sl : 16;
  const int fgrp = a.f_group > 1 ? a.f_group : 1;
  const bool self = a.fmode == CMF_F_SELF_RELU;
  const bool has_f = a.f != nullptr && a.fmode != CMF_F_NONE && !self;

  long long gy_lane[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int co = co0 + mt * 16 + r;
    gy_lane[mt] = (long long)(co < a.cout ? co : a.cout - 1) * a.y_co + 4 * q;      // rows >= cout: dropped by the reduction
  }
  // column n = nt * 16 + r of the block: TAPS == 9: (ci, tap) = (n / 9, n % 9); TAPS == 1: ci = n
  long long x_lane[NT], f_lane[NT];
  int n_dy[NT], n_dx[NT];
  bool n_ok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + r, ci = ci0 + (TAPS == 9 ? n / 9 : n), tap = TAPS == 9 ? n % 9 : 0;
    const int cic = ci < a.cin ? ci : a.cin - 1;
    n_ok[nt] = ci < a.cin;
    n_dy[nt] = TAPS == 9 ? tap / 3 - 1 : 0;
    n_dx[nt] = TAPS == 9 ? tap % 3 - 1 : 0;
    x_lane[nt] = (long long)cic * a.x_ci + 4 * q;
    f_lane[nt] = (long long)cic * a.f_ci;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long per = (nkb + gridDim.x - 1) / gridDim.x;
  const long long kb0 = (long long)blockIdx.x * per, kb1 = kb0 + per < nkb ? kb0 + per : nkb;
  for (long long kb = kb0 + wave; kb < kb1; kb += 8) {
    const int sl = (int)(kb % nsl);
    const long long t = kb / nsl;
    const int px = (int)(t % HW), n = (int)(t / HW);
    const int yy = px / W, xx = px - yy * W;
    f32x4 g[MT], v[NT];
    float m[NT];
    bool ok[NT];
    const float* gb = gy + (long long)n * a.y_np + (long long)px * a.y_px + sl * ysl;
    const float* xb = a.x + (long long)n * a.x_np + sl * xsl;
    const float* fb = has_f ? a.f + (long long)(n / fgrp) * a.f_np + (n % fgrp) : a.x;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) g[mt] = *reinterpret_cast<const f32x4*>(gb + gy_lane[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {                                // every load unconditional (clamped address), validity a select
      const int y2 = yy + n_dy[nt], x2 = xx + n_dx[nt];
      const bool in = y2 >= 0 && y2 < a.H && x2 >= 0 && x2 < W;
      const int p2 = in ? y2 * W + x2 : px;
      v[nt] = *reinterpret_cast<const f32x4*>(xb + (long long)p2 * a.x_px + x_lane[nt]);
      m[nt] = has_f ? fb[(long long)p2 * a.f_px + f_lane[nt]] : 1.f;
      ok[nt] = in && n_ok[nt];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float mj = has_f ? factor_of(m[nt], a.fmode) : 1.f;
      f32x4 w;
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = ok[nt] ? (self ? fmaxf(v[nt][k], 0.f) : v[nt][k] * mj) : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[mt][k], w[k], acc[mt][nt], 0, 0, 0);
    }
  }
  // the eight partial blocks -> LDS -> summed in wave order -> ws[wg][co 64][ci 64][TAPS]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      *reinterpret_cast<f32x4*>(thin_part + ((wave * MT * NT + mt * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * 64 * 64 * TAPS;
  for (int e = threadIdx.x; e < MT * NT * 256; e += 512) {
    const int i = e & 3, ln = (e >> 2) & 63, tile = e >> 8;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += thin_part[(w * MT * NT + tile) * 256 + ln * 4 + i];
    const int mt = tile / NT, nt = tile % NT, rr = ln & 15, qq = ln >> 4;
    const int co = mt * 16 + 4 * qq + i, n = nt * 16 + rr;       // D[row = 4 q + i][col = r]
    const int ci = TAPS == 9 ? n / 9 : n, tap = TAPS == 9 ? n % 9 : 0;
    if (ci < 64) out[(co * 64 + ci) * TAPS + tap] = sum;
  }
}

// 3x3 with a sliding window.  The simple kernel above issues 19 sixteen-byte loads per K block and wave, nine of them
// for pixels the same wave loaded one and two steps earlier.  Here wave w owns output-channel tile w & 3 and
// input-channel tiles {2 (w >> 2), 2 (w >> 2) + 1} for ALL nine taps, walks along image rows, and keeps four image
// columns (3 rows x 2 channel tiles each) in a register ring: per step ONE new column (6 loads + 6 factor words) and one
// gy quad are fetched, one step (72 MFMAs) ahead of their first use.  An image row is a stream of
// column slots with the zero-padding columns included (as zeros), rows follow each other without a pipeline restart, and
// everything is branch-free: validity is a select on the loaded values.  (A row is W + 1 slots: one zero column between rows.)
struct WCol {
  f32x4 v[2][3];                                                   // [ci tile][row dy]
  float f[2][3];
};

template <bool SELF>   // SELF: the input's own relu instead of a factor tensor (a run-time select here cost 100+ branches and vmcnt(0) waits)
__global__ __launch_bounds__(512, 2) void conv_wgrad3x3_kernel(cmf_conv_tangent_args a, const float* __restrict__ gy,
                                                               float* __restrict__ ws, int co0, int ci0, int nrows) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = wave & 3, h = wave >> 2;
  if (co0 + c * 16 >= a.cout || ci0 + 2 * h * 16 >= a.cin) return;  // nothing in range for this wave (no barriers in this kernel)
  const int r = lane & 15, q = lane >> 4;
  const int W = a.W, H = a.H, nsl = a.nc / 16;
  const long long xsl = a.x_sl ? a.x_sl : 16, ysl = a.y_sl ? a.y_sl : 16;
  const int fgrp = a.f_group > 1 ? a.f_group : 1;
  const bool has_f = !SELF && a.f != nullptr && a.fmode != CMF_F_NONE;
  // factor = c0 + c1 [f > 0] + c2 f + c3 f^2  (NONE: 1, RELU: [f > 0], TANH: 1 - f^2, RAW: f) -- branch-free
  const float fc0 = (!has_f || a.fmode == CMF_F_TANH) ? 1.f : 0.f, fc1 = (has_f && a.fmode == CMF_F_RELU) ? 1.f : 0.f;
  const float fc2 = (has_f && a.fmode == CMF_F_RAW) ? 1.f : 0.f, fc3 = (has_f && a.fmode == CMF_F_TANH) ? -1.f : 0.f;

  const int co = co0 + c * 16 + r;
  const long long gy_lane = (long long)(co < a.cout ? co : a.cout - 1) * a.y_co + 4 * q;
  long long x_lane[2], f_lane[2];
  bool ci_ok[2];
#pragma unroll
  for (int il = 0; il < 2; ++il) {
    const int ci = ci0 + (2 * h + il) * 16 + r, cic = ci < a.cin ? ci : a.cin - 1;
    ci_ok[il] = ci < a.cin;
    x_lane[il] = (long long)cic * a.x_ci + 4 * q;
    f_lane[il] = (long long)cic * a.f_ci;
  }

  f32x4 acc[2][9];
#pragma unroll
  for (int il = 0; il < 2; ++il)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[il][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this workgroup's image rows (row id = (sample * H + y) * nsl + slice) and its slot stream
  const int per = (nrows + gridDim.x - 1) / gridDim.x;
  const int row0 = blockIdx.x * per, row1 = row0 + per < nrows ? row0 + per : nrows;
  // W + 1 slots per row: the zero column right of one row doubles as the zero column left of the next (column -1 only)
  const int nslots = row1 > row0 ? (row1 - row0) * (W + 1) : 0;     // slots past the end are dead = zero columns

  // load cursor: the slot the next fetch brings in
  int l_slot = 0, l_col = -1, l_sl = 0, l_yy = 0, l_n = 0;
  if (row1 > row0) {
    l_sl = row0 % nsl;
    const int t = row0 / nsl;
    l_yy = t % H;
    l_n = t / H;
  }
  WCol ring[4];
  f32x4 gring[4];
  int okrow[4], okg[4];                                            // wave-uniform validity bits of the ring entries
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    okrow[i] = okg[i] = 0;
    gring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int il = 0; il < 2; ++il)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) ring[i].v[il][dy] = f32x4{0.f, 0.f, 0.f, 0.f}, ring[i].f[il][dy] = 0.f;
  }

  auto fetch = [&](WCol& cs, f32x4& g, int& rowbits, int& gbit) __attribute__((always_inline)) {
    const bool alive = l_slot < nslots;
    const bool colok = alive && l_col >= 0 && l_col < W;
    const int colc = l_col < 0 ? 0 : l_col >= W ? W - 1 : l_col;
    const float* xb = a.x + (long long)l_n * a.x_np + l_sl * xsl;
    const float* fb = has_f ? a.f + (long long)(l_n / fgrp) * a.f_np + (l_n % fgrp) : a.x;
    rowbits = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y2 = l_yy + dy - 1;
      const bool rok = colok && y2 >= 0 && y2 < H;
      const int p2 = (y2 < 0 ? 0 : y2 >= H ? H - 1 : y2) * W + colc;
      rowbits |= rok ? 1 << dy : 0;
#pragma unroll
      for (int il = 0; il < 2; ++il) {
        cs.v[il][dy] = *reinterpret_cast<const f32x4*>(xb + (long long)p2 * a.x_px + x_lane[il]);
        cs.f[il][dy] = fb[has_f ? (long long)p2 * a.f_px + f_lane[il] : 0];   // !has_f: any finite word (fc1..3 = 0)
      }
    }
    g = *reinterpret_cast<const f32x4*>(gy + (long long)l_n * a.y_np + (long long)(l_yy * W + colc) * a.y_px + l_sl * ysl + gy_lane);
    gbit = colok ? 1 : 0;
    // advance (branch-free, SALU selects); past the end the cursor stays on the last slot, whose addresses are valid
    const int more = (alive && l_slot + 1 < nslots) ? 1 : 0;
    l_slot += alive ? 1 : 0;
    const int wrap_c = more && l_col == W - 1;
    l_col = wrap_c ? -1 : l_col + more;
    const int wrap_s = wrap_c && l_sl + 1 == nsl;
    l_sl = wrap_s ? 0 : l_sl + wrap_c;
    const int wrap_y = wrap_s && l_yy + 1 == H;
    l_yy = wrap_y ? 0 : l_yy + wrap_s;
    l_n += wrap_y;
  };
  auto settle = [&](WCol& cs, f32x4& g, int rowbits, int gbit) __attribute__((always_inline)) {   // first use: factor + validity
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int il = 0; il < 2; ++il) {
        const bool ok = ((rowbits >> dy) & 1) && ci_ok[il];
        const float f = cs.f[il][dy];
        const float m = fc0 + fc1 * (f > 0.f ? 1.f : 0.f) + f * (fc2 + fc3 * f);
#pragma unroll
        for (int k = 0; k < 4; ++k) cs.v[il][dy][k] = ok ? (SELF ? fmaxf(cs.v[il][dy][k], 0.f) : cs.v[il][dy][k] * m) : 0.f;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = gbit ? g[k] : 0.f;
  };

  if (nslots > 0) {
    fetch(ring[0], gring[0], okrow[0], okg[0]);
    fetch(ring[1], gring[1], okrow[1], okg[1]);
    settle(ring[0], gring[0], okrow[0], okg[0]);
  }
  // step i of an unrolled group: centre = ring[i], left = ring[i-1], right = ring[i+1] (settled here), fetch into ring[i+2]
  auto step = [&](auto I) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    constexpr int L = (i + 3) % 4, R = (i + 1) % 4, N = (i + 2) % 4;
    settle(ring[R], gring[R], okrow[R], okg[R]);                   // waits for the fetch of the PREVIOUS step ...
    fetch(ring[N], gring[N], okrow[N], okg[N]);                    // ... before this one is issued: it flies during the MFMAs
    const f32x4 g = gring[i];
#pragma unroll
    for (int il = 0; il < 2; ++il)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[il][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[k], ring[L].v[il][dy][k], acc[il][dy * 3 + 0], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[il][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[k], ring[i].v[il][dy][k], acc[il][dy * 3 + 1], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[il][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[k], ring[R].v[il][dy][k], acc[il][dy * 3 + 2], 0, 0, 0);
      }
  };
  for (int s = 0; s < nslots; s += 4) {                            // slots past the end are dead (all-zero operands)
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
  }

  float* out = ws + (size_t)blockIdx.x * 64 * 64 * 9;
#pragma unroll
  for (int il = 0; il < 2; ++il)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[((c * 16 + 4 * q + i) * 64 + (2 * h + il) * 16 + r) * 9 + t] = acc[il][t][i];
}

// dw[(co0 + co)][ci0 + ci][tap] += sum_wg ws[wg][co][ci][tap]   (fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nwg,
                                                           int taps, int co0, int ci0, int cout, int cin) {
  const int e = blockIdx.x * 256 + threadIdx.x, per = 64 * 64 * taps;
  if (e >= per) return;
  const int tap = e % taps, ci = (e / taps) % 64 + ci0, co = e / (taps * 64) + co0;
  if (co >= cout || ci >= cin) return;
  // eight independent partial sums (the loads of one chain would each wait for the previous add), combined in a fixed order
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int g = 0;
  for (; g + 8 <= nwg; g += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] += ws[(size_t)(g + j) * per + e];
  for (; g < nwg; ++g) s8[0] += ws[(size_t)g * per + e];
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
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
  if (a->fmode == CMF_F_RELU_BITS) return CMF_EINVAL;                                    // fp32 factor tensors only
  if (a->fmode != CMF_F_NONE && a->fmode != CMF_F_SELF_RELU && !a->f) return CMF_EINVAL;
  if (ws_bytes < cmf_conv_tangent_wgrad_ws(a)) return CMF_EINVAL;
  // 16-byte column quads on both operands
  if (((uintptr_t)a->x | (uintptr_t)gy) % 16 || (a->x_np | a->x_ci | a->x_px | a->x_sl | a->y_np | a->y_co | a->y_px | a->y_sl) % 4)
    return CMF_EINVAL;
  const long long nkb = (long long)a->np * a->H * a->W * (a->nc / 16);
  const long long nrows = (long long)a->np * a->H * (a->nc / 16);
  if (nrows > 0x7fffffffLL / (a->W + 2)) return CMF_ERANGE;
  // CMF_WGRAD_SIMPLE: diagnostic, the one-K-block-per-step kernel for 3x3 too.  (Tried as the choice for launches with few image
  // rows -- primal data, 56 rows at 32 samples: 0.136 ms against 0.098 ms for the row-walking kernel on 56 workgroups.)
  static const bool simple = getenv("CMF_WGRAD_SIMPLE") != nullptr;
  const long long units = (a->taps == 9 && !simple) ? nrows : nkb;
  const int grid = (int)(units < WG_MAX ? units : WG_MAX);
  hipStream_t s = (hipStream_t)stream;
  // thin shapes (first 3x3 / last 1x1 conv of a coupler): all eight waves on K, n = ci * 9 + tap
  const bool thin_in = a->taps == 9 && a->cin * 9 <= 64 && !simple, thin_out = a->taps == 1 && a->cout <= 16;
  if (thin_in || thin_out) {
    const int nt = thin_in ? (a->cin * 9 + 15) / 16 : 4;
    const void* fn = thin_out ? (const void*)conv_wgrad_thin_kernel<1, 1, 4>
                   : nt == 1 ? (const void*)conv_wgrad_thin_kernel<9, 4, 1>
                   : nt == 2 ? (const void*)conv_wgrad_thin_kernel<9, 4, 2>
                             : (const void*)conv_wgrad_thin_kernel<9, 4, 4>;
    const int tiles = thin_out ? 4 : 4 * (nt == 3 ? 4 : nt);
    const int lds = 8 * tiles * 256 * (int)sizeof(float);
    const hipError_t e = cmf_set_dynamic_lds(fn, lds);
    if (e != hipSuccess) return (int)e;
    const int tgrid = (int)(nkb < 8 * WG_MAX ? (nkb + 7) / 8 : WG_MAX);
    for (int co0 = 0; co0 < a->cout; co0 += 64)
      for (int ci0 = 0; ci0 < a->cin; ci0 += 64) {
        if (thin_out) hipLaunchKernelGGL((conv_wgrad_thin_kernel<1, 1, 4>), dim3(tgrid), dim3(512), lds, s, *a, gy, ws, co0, ci0, nkb);
        else if (nt == 1) hipLaunchKernelGGL((conv_wgrad_thin_kernel<9, 4, 1>), dim3(tgrid), dim3(512), lds, s, *a, gy, ws, co0, ci0, nkb);
        else if (nt == 2) hipLaunchKernelGGL((conv_wgrad_thin_kernel<9, 4, 2>), dim3(tgrid), dim3(512), lds, s, *a, gy, ws, co0, ci0, nkb);
        else hipLaunchKernelGGL((conv_wgrad_thin_kernel<9, 4, 4>), dim3(tgrid), dim3(512), lds, s, *a, gy, ws, co0, ci0, nkb);
        CMF_LAUNCH_CHECK();
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cmf_ceil_div(64 * 64 * a->taps, 256)), dim3(256), 0, s, ws, dw, tgrid,
                           a->taps, co0, ci0, a->cout, a->cin);
        CMF_LAUNCH_CHECK();
      }
    return 0;
  }
  for (int co0 = 0; co0 < a->cout; co0 += 64)
    for (int ci0 = 0; ci0 < a->cin; ci0 += 64) {
      if (a->taps == 9 && !simple && a->fmode == CMF_F_SELF_RELU)
        hipLaunchKernelGGL(conv_wgrad3x3_kernel<true>, dim3(grid), dim3(512), 0, s, *a, gy, ws, co0, ci0, (int)nrows);
      else if (a->taps == 9 && !simple)
        hipLaunchKernelGGL(conv_wgrad3x3_kernel<false>, dim3(grid), dim3(512), 0, s, *a, gy, ws, co0, ci0, (int)nrows);
      else if (a->taps == 9) hipLaunchKernelGGL(conv_wgrad_kernel<9>, dim3(grid), dim3(512), 0, s, *a, gy, ws, co0, ci0, nkb);
      else hipLaunchKernelGGL(conv_wgrad_kernel<1>, dim3(grid), dim3(512), 0, s, *a, gy, ws, co0, ci0, nkb);
      CMF_LAUNCH_CHECK();
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cmf_ceil_div(64 * 64 * a->taps, 256)), dim3(256), 0, s, ws, dw, grid,
                         a->taps, co0, ci0, a->cout, a->cin);
      CMF_LAUNCH_CHECK();
    }
  return 0;
}
