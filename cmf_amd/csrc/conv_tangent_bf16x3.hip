// Tangent 3x3 convolution with split-precision bf16 MFMA ("bf16x3") for gfx950 (MI355X).
//
// Same contract, HBM layouts, tiling and epilogue as conv_tangent.hip (fp32 MFMA), for taps == 9 and
// cin % 8 == 0.  Every fp32 operand is split as v = hi + lo with hi = bf16(v), lo = bf16(v - hi) and the
// product is formed as  hi*hi + hi*lo + lo*hi  on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the
// dropped lo*lo term and the representation residual are ~2^-16 relative, i.e. the result is fp32-grade
// (measured end to end against an fp64 evaluation of the reference: log-det / likelihood / g_ij within
// ~1e-6 relative, same as the fp32 path).  Three bf16 MFMAs replace sixteen fp32
// MFMA-cycles' worth of work: 16x16x32 does 16 Kflop in 16 cycles, 16x16x4 f32 does 2 Kflop in 32.
//
// K mapping: one MFMA contracts 32 K-slots = 4 lane groups x 8 contiguous bf16.  A slot group is one
// (tap, 8-channel octet) pair: lane group kq of K-step s carries tap 4*s + kq, so a chunk of 8 input
// channels (the same LDS chunking as the fp32 kernel, 2 workgroups per CU) takes 3 K-steps with taps
// 9..11 zero-weighted (25 % padding; the kernel is HBM-bound at this speed, not MFMA-bound).
//
// LDS images (all reads are 16-byte, linear within a lane group => bank-conflict free):
//   X hi/lo : [pixel][16 columns][8 channels] bf16   -> B fragment = one ds_read_b128 per lane
//   W hi/lo : [K-step][co tile][kq][16 co][8 channels] bf16, pre-split and pre-arranged by
//             cmf_pack_weight_bf16x3, so staging is a straight copy.
// A staging thread owns one (pixel, 4 columns) item: it loads the 8 channels of the octet (8 x 16 B),
// applies the activation-derivative factor, splits, and writes 4 + 4 ds_write_b128.
#include <type_traits>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int COT, int PXW>
struct BCfg {
  static constexpr int TW = 2 * PXW, TWH = TW + 2, PIXH = 4 * TWH;
  static constexpr int XS_BYTES = PIXH * 16 * 16;                 // one of hi / lo
  static constexpr int WS_BYTES = 3 * COT * 4 * 16 * 16;          // one of hi / lo
  static constexpr int W_CHUNK_BYTES = 2 * 3 * 4 * 4 * 16 * 16;   // global: [hl][s][cot 4][kq][co][8] bf16 = 24 KiB
  static constexpr int NX_ITEMS = PIXH * 4;                       // (pixel, column quad)
  static constexpr int NXIT = (NX_ITEMS + 255) / 256;
  static constexpr int NW_ITEMS = 2 * 3 * COT * 4 * 16;           // 16-byte items of the W chunk actually used
  static constexpr int NWIT = (NW_ITEMS + 255) / 256;
};

__device__ __forceinline__ unsigned pack_hi(float a, float b, float& ra, float& rb) {
  // RNE to bf16, return the packed pair and the exact fp32 remainders
  bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  const unsigned bits = __builtin_bit_cast(unsigned, h);
  ra = a - __builtin_bit_cast(float, bits << 16);
  rb = b - __builtin_bit_cast(float, bits & 0xffff0000u);
  return bits;
}
__device__ __forceinline__ unsigned pack_lo(float a, float b) {
  bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, h);
}

template <int COT, int PXW>
__global__ __launch_bounds__(256, 2) void conv_tangent_bf16x3_kernel(cmf_conv_tangent_args a, int tiles_x, int ntiles,
                                                                       int nslices, int ncog) {
  using C = BCfg<COT, PXW>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * C::XS_BYTES + 2 * C::WS_BYTES];
  unsigned char* Xh = smem;
  unsigned char* Xl = smem + C::XS_BYTES;
  unsigned char* Wh = smem + 2 * C::XS_BYTES;
  unsigned char* Wl = Wh + C::WS_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;

  int tile, slice, cog, np;
  {  // XCD-aware work mapping, identical to conv_tangent.hip
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    slice = w % nslices;
    w /= nslices;
    tile = w % ntiles;
    w /= ntiles;
    cog = w % ncog;
    np = w / ncog;
  }
  const int y0 = 2 * (tile / tiles_x), x0 = C::TW * (tile % tiles_x);
  const int nchunks = a.cin / 8;
  const int x_ci = (int)a.x_ci, x_px = (int)a.x_px, f_ci = (int)a.f_ci, f_px = (int)a.f_px;
  const float* xb = a.x + (long long)np * a.x_np + slice * 16;
  const float* fb = a.f ? a.f + (long long)np * a.f_np : a.x;    // never dereferenced when fmode == NONE
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w) + (long long)cog * nchunks * C::W_CHUNK_BYTES;

  // ---- staging plan: item = (pixel, column quad); loads are unconditional (clamped) ----
  int xo[C::NXIT], fo[C::NXIT];
  bool okv[C::NXIT];
#pragma unroll
  for (int it = 0; it < C::NXIT; ++it) {
    const int i = tid + 256 * it;
    const int q = i & 3, pix = i >> 2;
    const int hy = pix / C::TWH, hx = pix % C::TWH;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = i < C::NX_ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int gpix = gy * a.W + gx;
    okv[it] = ok;
    xo[it] = ok ? gpix * x_px + q * 4 : 0;
    fo[it] = ok ? gpix * f_px : 0;
  }
  const float fzero = (a.fmode == CMF_F_TANH) ? 1.f : 0.f;

  f32x4 xr[C::NXIT][8];
  float fr[C::NXIT][8];
  u32x4 wr[C::NWIT];

  auto prefetch = [&](int ch) {
#pragma unroll
    for (int it = 0; it < C::NXIT; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = ch * 8 + j;
        xr[it][j] = *reinterpret_cast<const f32x4*>(xb + ci * x_ci + xo[it]);
        if (a.fmode != CMF_F_NONE) {
          const float fv = fb[ci * f_ci + fo[it]];
          fr[it][j] = okv[it] ? fv : fzero;
        } else {
          fr[it][j] = okv[it] ? 1.f : 0.f;
        }
      }
#pragma unroll
    for (int it = 0; it < C::NWIT; ++it) {
      int i = tid + 256 * it;
      i = i < C::NW_ITEMS ? i : C::NW_ITEMS - 1;
      // item -> (hl, s, cot, rest 64): the global slab always has 4 co tiles per K-step
      const int rest = i & 63, t = i >> 6;
      const int cot = t % COT, s = (t / COT) % 3, hl = t / (3 * COT);
      wr[it] = *reinterpret_cast<const u32x4*>(wb + (long long)ch * C::W_CHUNK_BYTES +
                                               ((((hl * 3 + s) * 4 + cot) * 64 + rest) << 4));
    }
  };

  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < C::NXIT; ++it) {
      const int i = tid + 256 * it;
      if (i < C::NX_ITEMS) {
        float v[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float m = fr[it][j];
          if (a.fmode == CMF_F_RELU) m = m > 0.f ? 1.f : 0.f;
          else if (a.fmode == CMF_F_TANH) m = 1.f - m * m;
#pragma unroll
          for (int c = 0; c < 4; ++c) v[j][c] = xr[it][j][c] * m;
        }
        const int q = i & 3, pix = i >> 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {           // column q*4 + c: 8 channels -> 16 B hi + 16 B lo
          u32x4 h, l;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float ra, rb;
            h[jj] = pack_hi(v[2 * jj][c], v[2 * jj + 1][c], ra, rb);
            l[jj] = pack_lo(ra, rb);
          }
          const int off = ((pix * 16 + q * 4 + c) << 4);
          *reinterpret_cast<u32x4*>(Xh + off) = h;
          *reinterpret_cast<u32x4*>(Xl + off) = l;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < C::NWIT; ++it) {
      const int i = tid + 256 * it;
      if (i < C::NW_ITEMS) {
        const int rest = i & 63, t = i >> 6;
        const int cot = t % COT, s = (t / COT) % 3, hl = t / (3 * COT);
        unsigned char* dst = (hl ? Wl : Wh) + ((((s * COT + cot) * 64) + rest) << 4);
        *reinterpret_cast<u32x4*>(dst) = wr[it];
      }
    }
  };

  f32x4 acc[PXW][COT];
#pragma unroll
  for (int p = 0; p < PXW; ++p)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane B offsets of the three K-steps: lane group kq of step s reads tap 4*s + kq (taps >= 9 read tap 8's
  // data against zero weights)
  const int wrow = wave >> 1, wx = (wave & 1) * PXW;
  int boff[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    int tap = 4 * s + kq;
    tap = tap < 9 ? tap : 8;
    boff[s] = ((((wrow + tap / 3) * C::TWH + wx + tap % 3) * 16 + cl) << 4);
  }
  const int aoff = lane << 4;

  prefetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    commit();
    __syncthreads();
    if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      bf16x8 ah[COT], al[COT];
#pragma unroll
      for (int c = 0; c < COT; ++c) {
        ah[c] = *reinterpret_cast<const bf16x8*>(Wh + (((s * COT + c) * 64) << 4) + aoff);
        al[c] = *reinterpret_cast<const bf16x8*>(Wl + (((s * COT + c) * 64) << 4) + aoff);
      }
#pragma unroll
      for (int p = 0; p < PXW; ++p) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Xh + boff[s] + p * 256);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Xl + boff[s] + p * 256);
#pragma unroll
        for (int c = 0; c < COT; ++c) {
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[c], bh, acc[p][c], 0, 0, 0);
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c], bl, acc[p][c], 0, 0, 0);
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c], bh, acc[p][c], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue (as conv_tangent.hip) ----
  const int y_co = (int)a.y_co, y_px = (int)a.y_px, r_co = (int)a.r_co, r_px = (int)a.r_px;
  float* ybase = a.y + (long long)np * a.y_np + slice * 16 + cl;
  const float* rbase = a.r ? a.r + (long long)np * a.r_np + slice * 16 + cl : nullptr;
  const int co0 = cog * 64 + kq * 4;
  const bool full = (cog * 64 + COT * 16) <= a.cout;
  auto store_all = [&](auto has_res, auto is_full) {
#pragma unroll
    for (int p = 0; p < PXW; ++p) {
      const int gy = y0 + wrow, gx = x0 + wx + p;
      if (!(gy < a.H && gx < a.W)) continue;
      const int gpix = gy * a.W + gx;
      float* yp = ybase + gpix * y_px + co0 * y_co;
      const float* rp = has_res ? rbase + gpix * r_px + co0 * r_co : nullptr;
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (is_full || (co0 + c * 16 + r) < a.cout) {
            float v = acc[p][c][r];
            if (has_res) v += rp[(c * 16 + r) * r_co];
            yp[(c * 16 + r) * y_co] = v;
          }
        }
    }
  };
  if (rbase) {
    if (full) store_all(std::true_type{}, std::true_type{});
    else store_all(std::true_type{}, std::false_type{});
  } else {
    if (full) store_all(std::false_type{}, std::true_type{});
    else store_all(std::false_type{}, std::false_type{});
  }
}

// weight pre-split / pre-arrangement: out[cog][chunk][hl][s][cot 4][kq][co 16][8] bf16
__global__ void pack_weight_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout,
                                          int cin, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7), col = (int)((i >> 3) & 15), kq = (int)((i >> 7) & 3), cot = (int)((i >> 9) & 3);
  long long t = i >> 11;
  const int s = (int)(t % 3);
  t /= 3;
  const int hl = (int)(t & 1);
  t >>= 1;
  const int nchunks = cin / 8;
  const int ch = (int)(t % nchunks), cog = (int)(t / nchunks);
  const int co = cog * 64 + cot * 16 + col, ci = ch * 8 + j, tap = 4 * s + kq;
  float v = 0.f;
  if (co < cout && tap < 9) v = w[((long long)co * cin + ci) * 9 + tap];
  const __bf16 h = (__bf16)v;
  const float hf = (float)h;
  const __bf16 r = hl ? (__bf16)(v - hf) : h;
  out[i] = __builtin_bit_cast(unsigned short, r);
}

template <int COT, int PXW>
int launch(const cmf_conv_tangent_args& a, hipStream_t s) {
  const int tiles_x = cmf_ceil_div(a.W, 2 * PXW), tiles = tiles_x * cmf_ceil_div(a.H, 2);
  const int nslices = a.nc / 16, ncog = cmf_ceil_div(a.cout, 64);
  const long long total = (long long)tiles * nslices * ncog * a.np;
  if (total > 0x7fffffffLL) return CMF_ERANGE;
  hipLaunchKernelGGL((conv_tangent_bf16x3_kernel<COT, PXW>), dim3((unsigned)total), dim3(256), 0, s, a, tiles_x, tiles,
                     nslices, ncog);
  CMF_LAUNCH_CHECK();
  return 0;
}

template <int PXW>
int launch_cot(const cmf_conv_tangent_args& a, hipStream_t s) {
  const int cot = (a.cout >= 64) ? 4 : (a.cout + 15) / 16;
  switch (cot) {
    case 1: return launch<1, PXW>(a, s);
    case 2: return launch<2, PXW>(a, s);
    case 3: return launch<3, PXW>(a, s);
    default: return launch<4, PXW>(a, s);
  }
}

inline bool fits_int(long long v) { return v >= 0 && v < (1LL << 31); }

}  // namespace

extern "C" int cmf_pack_weight_bf16x3(const float* w, void* out, int cout, int cin, long long* out_bytes, void* stream) {
  if (cout <= 0 || cin <= 0 || cin % 8) return CMF_EINVAL;
  const long long total = (long long)((cout + 63) / 64) * (cin / 8) * 2 * 3 * 4 * 4 * 16 * 8;   // bf16 elements
  if (out_bytes) *out_bytes = total * 2;
  if (!out) return 0;
  if (!w) return CMF_EINVAL;
  hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     w, (unsigned short*)out, cout, cin, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_conv_tangent_bf16x3(const cmf_conv_tangent_args* ap, void* stream) {
  if (!ap) return CMF_EINVAL;
  const cmf_conv_tangent_args& a = *ap;
  if (!a.x || !a.w || !a.y || a.np <= 0 || a.cin <= 0 || a.cout <= 0 || a.H <= 0 || a.W <= 0) return CMF_EINVAL;
  if (a.taps != 9 || a.cin % 8 || a.nc <= 0 || a.nc % 16) return CMF_EINVAL;
  if (a.fmode < CMF_F_NONE || a.fmode > CMF_F_RAW || (a.fmode != CMF_F_NONE && !a.f)) return CMF_EINVAL;
  if ((a.x_np | a.x_ci | a.x_px) % 4 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return CMF_EINVAL;
  const long long HW = (long long)a.H * a.W;
  if (!fits_int((a.cin + 8) * a.x_ci + HW * a.x_px + a.nc) || !fits_int((a.cin + 8) * a.f_ci + HW * a.f_px) ||
      !fits_int((a.cout + 64) * a.y_co + HW * a.y_px + a.nc) || (a.r && !fits_int((a.cout + 64) * a.r_co + HW * a.r_px + a.nc)) ||
      HW > (1 << 24))
    return CMF_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  if (a.W % 14) return CMF_EINVAL;      // only the 7-pixel-per-wave tiling is built (14- and 28-wide images)
  return launch_cot<7>(a, s);
}
