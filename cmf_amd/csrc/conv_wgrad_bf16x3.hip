// Split-precision weight gradient of the 3x3 tangent convolution (fp32 version and the math: conv_wgrad.hip):
//   dW[co][ci][tap] += sum_{n, px, col} gy(n, co, px, col) * F(n, ci, px+tap) * x(n, ci, px+tap, col)
// with every operand split v = hi + lo (bf16, RNE) and hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulation
// (the arithmetic of conv_tangent_bf16x3.hip).
//
// Why a second kernel: in the fp32 version each of the four output-channel waves loads and masks the same input columns from
// global memory and keeps them in a register ring.  Here a 512-thread workgroup shares the work through LDS:
//   * K of one MFMA = 32 = two 16-column slices of one pixel; lane (r, kg) of a fragment holds 8 consecutive columns
//     (32 bytes of fp32 in global memory) of channel r, slice 2 sp + kg / 2 -- the same mapping for both operands;
//   * per image column (one "slot", conv_wgrad.hip) 16 fragments are produced ONCE per workgroup -- input rows y-1, y, y+1 x 4
//     channel tiles, masked by relu', and gy of row y x 4 channel tiles -- two per wave: 4 global loads, ~60 VALU for the hi / lo
//     split, 4 ds_write_b128, into a ring of four slots (4 x 32 KB of LDS);
//   * one barrier per slot, then every wave runs its 18 tiles (one output-channel tile x two input-channel tiles x nine taps):
//     38 ds_read_b128 and 54 MFMAs of 16 cycles -- against 72 fp32 MFMAs of 32 cycles for half the K.
// Shapes: taps = 9, cin % 64 == 0, cout % 64 == 0, nc % 32 == 0, factor NONE, RELU from a float tensor or SELF_RELU; everything else
// stays on the fp32 kernel.  Partial blocks and their fixed-order reduction are shared with conv_wgrad.hip.
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_MAX = 256;
constexpr int FRAG_BYTES = 64 * 16;                 // one bf16x8 per lane
constexpr int SLOT_BYTES = 16 * 2 * FRAG_BYTES;     // 16 fragments x (hi, lo)
constexpr int LDS_BYTES = 4 * SLOT_BYTES;           // ring of four slots = 128 KB

__device__ __forceinline__ unsigned pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

struct Raw {                                        // what one wave fetches for its two fragments of one slot
  f32x4 v[2][2];
  float f[2];
  int ok;                                           // bit i: fragment i is inside the image (wave-uniform)
};

// MODE 0: no factor, 1: relu' from a float tensor, 2: SELF -- the input's own relu, elementwise (primal data: samples in the columns)
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_wgrad3x3_split_kernel(cmf_conv_tangent_args a, const float* __restrict__ gy,
                                                                     float* __restrict__ ws, int co0, int ci0, int nrows) {
  constexpr bool HASF = MODE == 1, SELF = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = wave & 3, h = wave >> 2;
  const int r = lane & 15, kg = lane >> 4;
  const int W = a.W, H = a.H, nsp = a.nc / 32;
  const long long xsl = a.x_sl ? a.x_sl : 16, ysl = a.y_sl ? a.y_sl : 16;

  // producer role: fragments 2 wave and 2 wave + 1 of a slot.  0..11: input, row dy = f / 4, channel tile f % 4; 12..15: gy tile f - 12
  long long lane_off[2], f_off[2];
  int p_dy[2];
  bool p_gy[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = 2 * wave + i;
    p_gy[i] = f >= 12;
    p_dy[i] = p_gy[i] ? 1 : f / 4;
    const int tile = p_gy[i] ? f - 12 : f % 4;
    const long long ch = (p_gy[i] ? co0 : ci0) + tile * 16 + r;
    lane_off[i] = ch * (p_gy[i] ? a.y_co : a.x_ci) + (kg >> 1) * (p_gy[i] ? ysl : xsl) + 8 * (kg & 1);
    f_off[i] = p_gy[i] ? 0 : ch * a.f_ci;
  }

  f32x4 acc[2][9];
#pragma unroll
  for (int il = 0; il < 2; ++il)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[il][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this workgroup's image rows (row id = (sample * H + y) * nsp + slice pair); a row is W + 1 slots (conv_wgrad.hip)
  const int per = (nrows + gridDim.x - 1) / gridDim.x;
  const int row0 = blockIdx.x * per, row1 = row0 + per < nrows ? row0 + per : nrows;
  const int nslots = row1 > row0 ? (row1 - row0) * (W + 1) : 0;
  int l_slot = 0, l_col = -1, l_sp = 0, l_yy = 0, l_n = 0;
  if (row1 > row0) {
    l_sp = row0 % nsp;
    const int t = row0 / nsp;
    l_yy = t % H;
    l_n = t / H;
  }

  auto fetch = [&](Raw& raw) __attribute__((always_inline)) {
    const bool alive = l_slot < nslots;
    const bool colok = alive && l_col >= 0 && l_col < W;
    const int colc = l_col < 0 ? 0 : l_col >= W ? W - 1 : l_col;
    raw.ok = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y2 = l_yy + p_dy[i] - 1;
      const bool ok = colok && y2 >= 0 && y2 < H;
      const long long pix = (long long)(y2 < 0 ? 0 : y2 >= H ? H - 1 : y2) * W + colc;
      const float* base = p_gy[i] ? gy + (long long)l_n * a.y_np + pix * a.y_px + 2 * l_sp * ysl
                                  : a.x + (long long)l_n * a.x_np + pix * a.x_px + 2 * l_sp * xsl;
      raw.v[i][0] = *reinterpret_cast<const f32x4*>(base + lane_off[i]);
      raw.v[i][1] = *reinterpret_cast<const f32x4*>(base + lane_off[i] + 4);
      raw.f[i] = (HASF && !p_gy[i]) ? a.f[(long long)l_n * a.f_np + pix * a.f_px + f_off[i]] : 1.f;
      raw.ok |= ok ? 1 << i : 0;
    }
    const int more = (alive && l_slot + 1 < nslots) ? 1 : 0;       // branch-free advance; past the end the cursor stays put
    l_slot += alive ? 1 : 0;
    const int wrap_c = more && l_col == W - 1;
    l_col = wrap_c ? -1 : l_col + more;
    const int wrap_s = wrap_c && l_sp + 1 == nsp;
    l_sp = wrap_s ? 0 : l_sp + wrap_c;
    const int wrap_y = wrap_s && l_yy + 1 == H;
    l_yy = wrap_y ? 0 : l_yy + wrap_s;
    l_n += wrap_y;
  };
  // mask, split hi / lo, park in ring slot `slot`
  auto produce = [&](const Raw& raw, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (raw.ok >> i) & 1;
      const float m = ok ? ((HASF && !p_gy[i]) ? (raw.f[i] > 0.f ? 1.f : 0.f) : 1.f) : 0.f;
      u32x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float e = raw.v[i][j >> 1][(2 * j) & 3], o = raw.v[i][j >> 1][(2 * j + 1) & 3];
        if (SELF && !p_gy[i]) e = fmaxf(e, 0.f), o = fmaxf(o, 0.f);
        e = ok ? e * m : 0.f, o = ok ? o * m : 0.f;
        const unsigned hb = pack2(e, o);
        hi[j] = hb;
        lo[j] = pack2(e - __builtin_bit_cast(float, hb << 16), o - __builtin_bit_cast(float, hb & 0xffff0000u));
      }
      unsigned char* dst = smem + slot * SLOT_BYTES + (2 * wave + i) * 2 * FRAG_BYTES + lane * 16;
      *reinterpret_cast<u32x4*>(dst) = hi;
      *reinterpret_cast<u32x4*>(dst + FRAG_BYTES) = lo;
    }
  };
  auto frag = [&](int slot, int f, int hl) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(smem + slot * SLOT_BYTES + (f * 2 + hl) * FRAG_BYTES + lane * 16);
  };
  // centre slot C with its neighbours L, R: 18 tiles x 3 products
  auto compute = [&](int L, int Cc, int R) __attribute__((always_inline)) {
    const bf16x8 gh = frag(Cc, 12 + c, 0), gl = frag(Cc, 12 + c, 1);
#pragma unroll
    for (int il = 0; il < 2; ++il)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int s = dx == 0 ? L : dx == 1 ? Cc : R;
          const bf16x8 xh = frag(s, dy * 4 + 2 * h + il, 0), xl = frag(s, dy * 4 + 2 * h + il, 1);
          f32x4 t = acc[il][dy * 3 + dx];
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xl, t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, xh, t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, xh, t, 0, 0, 0);
          acc[il][dy * 3 + dx] = t;
        }
  };

  if (nslots > 0) {                                 // (wave-uniform and workgroup-uniform: every wave takes the same barriers)
    // ring slot 3 stands for the column left of slot 0: zeros
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned char* dst = smem + 3 * SLOT_BYTES + (2 * wave + i) * 2 * FRAG_BYTES + lane * 16;
      *reinterpret_cast<u32x4*>(dst) = u32x4{0, 0, 0, 0};
      *reinterpret_cast<u32x4*>(dst + FRAG_BYTES) = u32x4{0, 0, 0, 0};
    }
    Raw raw;
    fetch(raw);
    produce(raw, 0);                                // slot 0
    fetch(raw);                                     // slot 1, parked by step 0
    // step s (centre s): park slot s+1 (fetched one step ago), fetch slot s+2, barrier, compute.  Slot s+1 reuses the ring
    // entry of slot s-3, last read in step s-2 -- every wave finished that before it passed the barrier of step s-1.
    auto step = [&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      produce(raw, (i + 1) & 3);
      fetch(raw);
      __syncthreads();
      compute((i + 3) & 3, i, (i + 1) & 3);
    };
    for (int s = 0; s < nslots; s += 4) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    }
  }

  float* out = ws + (size_t)blockIdx.x * 64 * 64 * 9;
#pragma unroll
  for (int il = 0; il < 2; ++il)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[((c * 16 + 4 * kg + i) * 64 + (2 * h + il) * 16 + r) * 9 + t] = acc[il][t][i];
}

// dw[(co0 + co)][ci0 + ci][tap] += sum_wg ws[wg][co][ci][tap]   (fixed order; same as conv_wgrad.hip)
__global__ __launch_bounds__(256) void wgrad_reduce_split_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nwg, int co0,
                                                                 int ci0, int cin) {
  const int e = blockIdx.x * 256 + threadIdx.x, per = 64 * 64 * 9;
  if (e >= per) return;
  const int tap = e % 9, ci = (e / 9) % 64 + ci0, co = e / (9 * 64) + co0;
  // eight independent partial sums (the loads of one chain would each wait for the previous add), combined in a fixed order
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int g = 0;
  for (; g + 8 <= nwg; g += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] += ws[(size_t)(g + j) * per + e];
  for (; g < nwg; ++g) s8[0] += ws[(size_t)g * per + e];
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  dw[((size_t)co * cin + ci) * 9 + tap] += s;
}

}  // namespace

extern "C" int cmf_conv_tangent_wgrad_bf16x3(const cmf_conv_tangent_args* a, const float* gy, float* dw, float* ws,
                                             long long ws_bytes, void* stream) {
  if (!a || !a->x || !gy || !dw || !ws) return CMF_EINVAL;
  if (a->taps != 9 || a->cin % 64 || a->cout % 64 || a->nc <= 0 || a->nc % 32) return CMF_EINVAL;
  if (a->np <= 0 || a->H <= 0 || a->W <= 0) return CMF_EINVAL;
  if (a->fmode != CMF_F_NONE && a->fmode != CMF_F_RELU && a->fmode != CMF_F_SELF_RELU) return CMF_EINVAL;
  if (a->fmode == CMF_F_RELU && (!a->f || a->f_group > 1)) return CMF_EINVAL;
  if (ws_bytes < (long long)WG_MAX * 64 * 64 * 9 * (long long)sizeof(float)) return CMF_EINVAL;
  if (((uintptr_t)a->x | (uintptr_t)gy) % 16 || (a->x_np | a->x_ci | a->x_px | a->x_sl | a->y_np | a->y_co | a->y_px | a->y_sl) % 4)
    return CMF_EINVAL;
  const long long nrows = (long long)a->np * a->H * (a->nc / 32);
  if (nrows > 0x7fffffffLL / (a->W + 2)) return CMF_ERANGE;
  const int grid = (int)(nrows < WG_MAX ? nrows : WG_MAX);
  hipStream_t s = (hipStream_t)stream;
  {                                                                // per (device, kernel) memo: runtime.hip
    hipError_t e = cmf_set_dynamic_lds((const void*)conv_wgrad3x3_split_kernel<0>, LDS_BYTES);
    if (e == hipSuccess) e = cmf_set_dynamic_lds((const void*)conv_wgrad3x3_split_kernel<1>, LDS_BYTES);
    if (e == hipSuccess) e = cmf_set_dynamic_lds((const void*)conv_wgrad3x3_split_kernel<2>, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
  }
  for (int co0 = 0; co0 < a->cout; co0 += 64)
    for (int ci0 = 0; ci0 < a->cin; ci0 += 64) {
      if (a->fmode == CMF_F_RELU)
        hipLaunchKernelGGL(conv_wgrad3x3_split_kernel<1>, dim3(grid), dim3(512), LDS_BYTES, s, *a, gy, ws, co0, ci0, (int)nrows);
      else if (a->fmode == CMF_F_SELF_RELU)
        hipLaunchKernelGGL(conv_wgrad3x3_split_kernel<2>, dim3(grid), dim3(512), LDS_BYTES, s, *a, gy, ws, co0, ci0, (int)nrows);
      else
        hipLaunchKernelGGL(conv_wgrad3x3_split_kernel<0>, dim3(grid), dim3(512), LDS_BYTES, s, *a, gy, ws, co0, ci0, (int)nrows);
      CMF_LAUNCH_CHECK();
      hipLaunchKernelGGL(wgrad_reduce_split_kernel, dim3(cmf_ceil_div(64 * 64 * 9, 256)), dim3(256), 0, s, ws, dw, grid, co0, ci0, a->cin);
      CMF_LAUNCH_CHECK();
    }
  return 0;
}
