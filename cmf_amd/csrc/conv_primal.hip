// Primal convolution / linear layer of the coupler networks on fp32 MFMA for gfx950 (MI355X).
//
//     v(b, co, px) = sum_{ci,tap} W[co][ci][tap] * in(b, ci, px+tap) + bias[co] [+ r(b, co, px)]
// with in = x | relu(x) | x*mask fused on load and y = v | tanh(v) | sw*tanh(v)+sb on store: the
// forward of nn.Conv2d / nn.Linear inside get_resnet / get_mlp
// (cmf/models/components/networks.py:50-60, :103-106, :116-161, :206-224) in the standard
// (B, C, H, W) / (B, F) layouts, so the tensors it writes double as the activation-derivative source
// of the tangent kernel (conv_tangent.hip) without any layout change.
//
// GEMM view: M = cout (A = packed weights), N = 16 consecutive pixels along x (one MFMA column block),
// K = (tap, cin) with v_mfma_f32_16x16x4_f32.  A wave owns 4 column blocks ("slots") x COT*16 output
// channels; a workgroup of 4 waves covers 16 slots = (16/NXB) rows x (NXB*16) pixels for 3x3
// (NXB = 1 for W <= 16, else 2) or 256 flat pixels for 1x1 (an MLP layer: pixels = batch samples).
// The primal pass is ~1/(d+2) of the path's FLOPs, so staging uses plain dword
// loads with arbitrary strides rather than vector loads.
#include <type_traits>
#include "common.h"

namespace {

constexpr int CIC = 8;

// SLOTS_W = column blocks (16 pixels each) per wave: 4 normally; 1 for 1x1 layers whose grid would otherwise
// underfill the chip (an MLP layer over B = 4096 samples is only 16 tiles of 256 samples).
template <int TAPS, int COT, int NXB, int SLOTS_W>
struct PCfg {
  static constexpr int ROWS = (TAPS == 9) ? 16 / NXB : 1;
  static constexpr int ROWS_H = (TAPS == 9) ? ROWS + 2 : 1;
  static constexpr int PIX1 = 64 * SLOTS_W;                          // flat pixels per workgroup (1x1)
  static constexpr int RS = (TAPS == 9) ? 16 * NXB + 2 : PIX1;       // LDS row stride (dwords)
  static constexpr int XS_RAW = ROWS_H * RS;
  static constexpr int XS_CI = ((XS_RAW + 15) / 32) * 32 + 16;      // >= XS_RAW and == 16 (mod 32)
  static constexpr int WS_CI = (COT % 2) ? COT * 16 : COT * 16 + 16;
  static constexpr int XS_FLOATS = CIC * XS_CI;
  static constexpr int WS_FLOATS = TAPS * CIC * WS_CI;
  static constexpr int NX_ITEMS = CIC * XS_RAW;
  static constexpr int NXIT = (NX_ITEMS + 255) / 256;
  static constexpr int NW_ITEMS = TAPS * CIC * COT * 4;
  static constexpr int NWIT = (NW_ITEMS + 255) / 256;
};

template <int TAPS, int COT, int NXB, int SLOTS_W>
__global__ __launch_bounds__(256, 2) void conv_primal_kernel(cmf_conv_primal_args a, int tiles_x, int ncog,
                                                               int cin_pad) {
  using C = PCfg<TAPS, COT, NXB, SLOTS_W>;
  static_assert(TAPS == 1 || SLOTS_W == 4, "3x3 tiles are 16 slots");
  static_assert(C::XS_CI >= C::XS_RAW && C::XS_CI % 32 == 16, "LDS channel stride");
  __shared__ __attribute__((aligned(16))) float smem[C::XS_FLOATS + C::WS_FLOATS];
  float* Xs = smem;
  float* Ws = smem + C::XS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;
  const int tile = blockIdx.x, cog = blockIdx.y, b = blockIdx.z;
  const int HW = a.H * a.W;

  int y0 = 0, x0 = 0, p0 = 0;
  if (TAPS == 9) {
    y0 = C::ROWS * (tile / tiles_x);
    x0 = 16 * NXB * (tile % tiles_x);
  } else {
    p0 = tile * C::PIX1;
  }
  const int x_c = (int)a.x_c, x_px = (int)a.x_px, f_c = (int)a.f_c, f_px = (int)a.f_px;
  const float* xb = a.x + (long long)b * a.x_b;
  const float* wb = a.w + (long long)cog * TAPS * cin_pad * 64;

  float xr[C::NXIT];
  f32x4 wr[C::NWIT];

  // loads are unconditional (clamped addresses); validity is applied as a select afterwards
  auto prefetch = [&](int ci0) {
#pragma unroll
    for (int it = 0; it < C::NXIT; ++it) {
      const int i = tid + 256 * it;
      const int ci = i / C::XS_RAW, rem = i % C::XS_RAW;
      bool ok = i < C::NX_ITEMS && (ci0 + ci) < a.cin;
      int gpix;
      if (TAPS == 9) {
        const int hy = rem / C::RS, hx = rem % C::RS;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        gpix = gy * a.W + gx;
      } else {
        gpix = p0 + rem;
        ok = ok && gpix < HW;
      }
      float v = xb[ok ? (ci0 + ci) * x_c + gpix * x_px : 0];
      if (a.imode == CMF_F_RELU) v = fmaxf(v, 0.f);
      else if (a.imode == CMF_F_RAW) v *= a.f[ok ? (ci0 + ci) * f_c + gpix * f_px : 0];
      xr[it] = ok ? v : 0.f;
    }
#pragma unroll
    for (int it = 0; it < C::NWIT; ++it) {
      int i = tid + 256 * it;
      i = i < C::NW_ITEMS ? i : C::NW_ITEMS - 1;
      const int q = i % (COT * 4), rowi = i / (COT * 4);
      const int ci = rowi % CIC, tap = rowi / CIC;
      wr[it] = *reinterpret_cast<const f32x4*>(wb + (tap * cin_pad + ci0 + ci) * 64 + q * 4);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < C::NXIT; ++it) {
      const int i = tid + 256 * it;
      if (i < C::NX_ITEMS) Xs[(i / C::XS_RAW) * C::XS_CI + i % C::XS_RAW] = xr[it];
    }
#pragma unroll
    for (int it = 0; it < C::NWIT; ++it) {
      const int i = tid + 256 * it;
      if (i < C::NW_ITEMS) {
        const int q = i % (COT * 4), rowi = i / (COT * 4);
        *reinterpret_cast<f32x4*>(Ws + rowi * C::WS_CI + q * 4) = wr[it];
      }
    }
  };

  f32x4 acc[SLOTS_W][COT];
#pragma unroll
  for (int p = 0; p < SLOTS_W; ++p)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // slot s = wave*4 + p: 3x3 -> (row s / NXB, x block s % NXB); 1x1 -> flat pixels [16 s, 16 s + 16)
  int soff[SLOTS_W];
#pragma unroll
  for (int p = 0; p < SLOTS_W; ++p) {
    const int s = wave * SLOTS_W + p;
    soff[p] = (TAPS == 9) ? (s / NXB) * C::RS + (s % NXB) * 16 : s * 16;
  }
  const float* a_base = Ws + kq * C::WS_CI + cl;
  const float* b_base = Xs + kq * C::XS_CI + cl;

  const int nchunks = (a.cin + CIC - 1) / CIC;
  prefetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    commit();
    __syncthreads();
    if (ch + 1 < nchunks) prefetch((ch + 1) * CIC);
#pragma unroll
    for (int kg = 0; kg < CIC / 4; ++kg) {
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int toff = (TAPS == 9) ? (tap / 3) * C::RS + (tap % 3) : 0;
        float av[COT];
#pragma unroll
        for (int c = 0; c < COT; ++c) av[c] = a_base[(tap * CIC + kg * 4) * C::WS_CI + c * 16];
#pragma unroll
        for (int p = 0; p < SLOTS_W; ++p) {
          const float bv = b_base[kg * 4 * C::XS_CI + soff[p] + toff];
#pragma unroll
          for (int c = 0; c < COT; ++c)
            acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv, acc[p][c], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue ----
  const int y_c = (int)a.y_c, y_px = (int)a.y_px, r_c = (int)a.r_c, r_px = (int)a.r_px;
  float* yb = a.y + (long long)b * a.y_b;
  float* gb = a.g ? a.g + (long long)b * a.y_b : nullptr;
  const float* rb = a.r ? a.r + (long long)b * a.r_b : nullptr;
#pragma unroll
  for (int p = 0; p < SLOTS_W; ++p) {
    const int s = wave * SLOTS_W + p;
    int gpix;
    bool ok;
    if (TAPS == 9) {
      const int gy = y0 + s / NXB, gx = x0 + (s % NXB) * 16 + cl;
      ok = gy < a.H && gx < a.W;
      gpix = gy * a.W + gx;
    } else {
      gpix = p0 + s * 16 + cl;
      ok = gpix < HW;
    }
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cog * 64 + c * 16 + kq * 4 + r;
        if (ok && co < a.cout) {
          float v = acc[p][c][r];
          if (a.bias) v += a.bias[co];
          if (rb) v += rb[co * r_c + gpix * r_px];
          if (a.omode == CMF_O_TANH) {
            v = tanhf(v);
          } else if (a.omode == CMF_O_STANH) {
            // 1 - tanh^2 as sech^2 = 4 e / (1 + e)^2, e = exp(-2 |v|): the difference form loses eps / (1 - t^2) of relative
            // accuracy when the tanh saturates (logit-scale image inputs drive it there), and g multiplies every tangent
            // and every gradient below this network
            const float t = tanhf(v), w = a.sw[co], e = expf(-2.f * fabsf(v));
            v = w * t + a.sb[co];
            if (gb) gb[co * y_c + gpix * y_px] = w * (4.f * e / ((1.f + e) * (1.f + e)));
          }
          yb[co * y_c + gpix * y_px] = v;
        }
      }
  }
}

template <int TAPS, int COT, int NXB, int SLOTS_W>
int launch(const cmf_conv_primal_args& a, hipStream_t s) {
  using C = PCfg<TAPS, COT, NXB, SLOTS_W>;
  int tiles, tiles_x = 1;
  if (TAPS == 9) {
    tiles_x = cmf_ceil_div(a.W, 16 * NXB);
    tiles = tiles_x * cmf_ceil_div(a.H, C::ROWS);
  } else {
    tiles = cmf_ceil_div((long long)a.H * a.W, C::PIX1);
  }
  const int ncog = cmf_ceil_div(a.cout, 64), cin_pad = (a.cin + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_primal_kernel<TAPS, COT, NXB, SLOTS_W>), dim3(tiles, ncog, a.B), dim3(256), 0, s, a, tiles_x,
                     ncog, cin_pad);
  CMF_LAUNCH_CHECK();
  return 0;
}

template <int TAPS, int NXB, int SLOTS_W>
int launch_cot(const cmf_conv_primal_args& a, hipStream_t s) {
  const int cot = (a.cout >= 64) ? 4 : (a.cout + 15) / 16;
  switch (cot) {
    case 1: return launch<TAPS, 1, NXB, SLOTS_W>(a, s);
    case 2: return launch<TAPS, 2, NXB, SLOTS_W>(a, s);
    case 3: return launch<TAPS, 3, NXB, SLOTS_W>(a, s);
    default: return launch<TAPS, 4, NXB, SLOTS_W>(a, s);
  }
}

inline bool fits_int(long long v) { return v >= 0 && v < (1LL << 31); }

}  // namespace

extern "C" int cmf_conv_primal(const cmf_conv_primal_args* ap, void* stream) {
  if (!ap) return CMF_EINVAL;
  const cmf_conv_primal_args& a = *ap;
  if (!a.x || !a.w || !a.y || a.B <= 0 || a.cin <= 0 || a.cout <= 0 || a.H <= 0 || a.W <= 0) return CMF_EINVAL;
  if (a.taps != 1 && a.taps != 9) return CMF_EINVAL;
  if (a.imode != CMF_F_NONE && a.imode != CMF_F_RELU && a.imode != CMF_F_RAW) return CMF_EINVAL;
  if (a.imode == CMF_F_RAW && !a.f) return CMF_EINVAL;
  if (a.omode < CMF_O_NONE || a.omode > CMF_O_STANH || (a.omode == CMF_O_STANH && (!a.sw || !a.sb))) return CMF_EINVAL;
  if ((uintptr_t)a.w % 16) return CMF_EINVAL;
  const long long HW = (long long)a.H * a.W;
  if (!fits_int((a.cin + 8) * a.x_c + HW * a.x_px) || !fits_int((a.cout + 64) * a.y_c + HW * a.y_px) ||
      (a.r && !fits_int((a.cout + 64) * a.r_c + HW * a.r_px)) || (a.f && !fits_int((a.cin + 8) * a.f_c + HW * a.f_px)) ||
      a.B > 65535 || HW > (1LL << 30))
    return CMF_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  if (a.taps == 9) return (a.W <= 16) ? launch_cot<9, 1, 4>(a, s) : launch_cot<9, 2, 4>(a, s);
  // 1x1: 256-pixel tiles unless that leaves most of the 256 CUs idle, then 64-pixel tiles
  const long long blocks256 = (HW + 255) / 256 * ((a.cout + 63) / 64) * a.B;
  return blocks256 >= 512 ? launch_cot<1, 1, 4>(a, s) : launch_cot<1, 1, 1>(a, s);
}
