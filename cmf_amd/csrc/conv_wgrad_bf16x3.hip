// Split-precision weight gradient of the 3x3 tangent convolution (fp32 version and the math: conv_wgrad.hip):
//   dW[co][ci][tap] += sum_{n, px, col} gy(n, co, px, col) * F(n, ci, px+tap) * x(n, ci, px+tap, col)
// with every operand split v = hi + lo (bf16, RNE) and hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulation
// (the arithmetic of conv_tangent_bf16x3.hip).
//
// Why a second kernel: in the fp32 version each of the four output-channel waves loads and masks the same input columns from
// global memory and keeps them in a register ring.  Here a 512-thread workgroup shares the work through LDS:
//   * K of one MFMA = 32 = two 16-column slices of one pixel; lane (r, kg) of a fragment holds 8 consecutive columns
//     of channel r, slice 2 sp + kg / 2 -- the same mapping for both operands;
//   * per image column (one "slot", conv_wgrad.hip) 16 fragments are produced ONCE per workgroup -- input rows y-1, y, y+1 x 4
//     channel tiles, masked by relu', and gy of row y x 4 channel tiles -- two per wave: 4 global loads, ~60 VALU for the hi / lo
//     split, 4 ds_write_b128, into a ring of four slots (4 x 32 KB of LDS);
//   * one barrier per slot, then every wave runs its 18 tiles (two output-channel tiles x one input-channel tile x nine taps):
//     22 ds_read_b128 and 54 MFMAs of 16 cycles -- against 72 fp32 MFMAs of 32 cycles for half the K.
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
  // consumer role: wave = 2 ci_t + co_h owns output-channel tiles 2 co_h, 2 co_h + 1 x input-channel tile ci_t x nine taps.  (With
  // one output tile x two input tiles per wave every input fragment was read by four waves: 38 ds_read_b128 per wave and step,
  // 304 KB per step and CU = 2400 cycles of LDS against 1730 of MFMA issue; this split reads 22.)
  const int co_h = wave & 1, ci_t = wave >> 1;
  const int r = lane & 15, kg = lane >> 4;
  const int W = a.W, H = a.H, nsp = a.nc / 32;
  const long long xsl = a.x_sl ? a.x_sl : 16, ysl = a.y_sl ? a.y_sl : 16;

  // producer role: fragments 2 wave and 2 wave + 1 of a slot.  0..11: input, row dy = f / 4, channel tile f % 4; 12..15: gy tile f - 12.
  // A producer lane (lc, lg) = (lane / 4, lane % 4) fetches columns 4 lg .. 4 lg + 3 of channel lc in BOTH slices of the pair (one
  // 16-byte load each): in the slice-major layout the 64 lanes of a load then cover one contiguous KiB (16 channels x 64 B); with
  // the fragment's own lane mapping (8 consecutive columns per lane, two loads) every load touched all sixteen 128-byte lines of the
  // fragment for half of their bytes.  The values reach the fragment layout through the LDS address of the four 8-byte stores.
  const int lc = lane >> 2, lg = lane & 3;
  long long lane_off[2], f_off[2], sl_off[2];
  int p_dy[2];
  bool p_gy[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = 2 * wave + i;
    p_gy[i] = f >= 12;
    p_dy[i] = p_gy[i] ? 1 : f / 4;
    const int tile = p_gy[i] ? f - 12 : f % 4;
    const long long ch = (p_gy[i] ? co0 : ci0) + tile * 16 + lc;
    lane_off[i] = ch * (p_gy[i] ? a.y_co : a.x_ci) + 4 * lg;
    sl_off[i] = p_gy[i] ? ysl : xsl;
    f_off[i] = p_gy[i] ? 0 : ch * a.f_ci;
  }
  // fragment lane (r, kg) holds columns 8 (kg & 1) .. + 7 of slice kg / 2: this lane's four columns of slice s are the (lg & 1) half
  // of fragment lane (lc, 2 s + lg / 2)
  const int st_off = (((lg >> 1) * 16 + lc) << 4) + ((lg & 1) << 3);

  f32x4 acc[2][9];
#pragma unroll
  for (int il = 0; il < 2; ++il)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[il][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // This workgroup's image rows; a row is W + 1 slots (conv_wgrad.hip).  Row id = (sample * nsp + slice pair) * H + y, y FASTEST,
  // and the rows are dealt so that the workgroups running on one XCD (private L2; workgroups go round-robin over the 8 XCDs) work
  // on CONSECUTIVE rows at the same time: XCD k owns the contiguous range [xstart, xstart + xlen) and its j-th workgroup takes rows
  // xstart + j, + nbx, + 2 nbx ...  Input row y is needed by output rows y-1, y, y+1; with each workgroup walking its own block of
  // rows the second and third use came ~1.8 MB later in that workgroup's stream -- 58 MB per XCD against 4 MB of L2 -- and every
  // input row was fetched from HBM three times (6.4 GB per launch at B = 128, the kernel's floor: 1.18 ms with the MFMAs removed).
  // Neighbouring rows in flight together make the other two uses L2 hits.  Speed only: any dealing covers every row exactly once.
  int xstart, xlen, nbx, jx;
  {
    const int G = gridDim.x, P = G < 8 ? G : 8, bid = blockIdx.x, xcd = bid % P;
    const int q = nrows / P, rem = nrows % P;
    xstart = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    xlen = q + (xcd < rem ? 1 : 0);
    nbx = (G - xcd + P - 1) / P;
    jx = bid / P;
  }
  const int my_rows = jx < xlen ? (xlen - jx + nbx - 1) / nbx : 0;
  const int nslots = my_rows * (W + 1);
  int l_slot = 0, l_col = -1, l_rid = xstart + jx, l_sp = 0, l_yy = 0, l_n = 0;
  auto decode_row = [&]() __attribute__((always_inline)) {
    l_yy = l_rid % H;
    const int t = l_rid / H;
    l_sp = t % nsp;
    l_n = t / nsp;
  };
  if (my_rows > 0) decode_row();

  // fetch in three pieces (fragment 0, fragment 1, cursor advance) so that a step can spread them between its MFMAs
  auto fetch_frag = [&](Raw& raw, int i) __attribute__((always_inline)) {
    const bool alive = l_slot < nslots;
    const bool colok = alive && l_col >= 0 && l_col < W;
    const int colc = l_col < 0 ? 0 : l_col >= W ? W - 1 : l_col;
    const int y2 = l_yy + p_dy[i] - 1;
    const bool ok = colok && y2 >= 0 && y2 < H;
    const long long pix = (long long)(y2 < 0 ? 0 : y2 >= H ? H - 1 : y2) * W + colc;
    const float* base = p_gy[i] ? gy + (long long)l_n * a.y_np + pix * a.y_px + 2 * l_sp * ysl
                                : a.x + (long long)l_n * a.x_np + pix * a.x_px + 2 * l_sp * xsl;
#ifdef CMF_DBG_WG_NOFETCH                              // timing-only builds (tools/build_dbg.sh): wrong results by design
    asm volatile("" : "+v"(raw.v[i][0]), "+v"(raw.v[i][1]) : "v"(base));
    raw.f[i] = 1.f;
#else
    raw.v[i][0] = *reinterpret_cast<const f32x4*>(base + lane_off[i]);
    raw.v[i][1] = *reinterpret_cast<const f32x4*>(base + lane_off[i] + sl_off[i]);
    raw.f[i] = (HASF && !p_gy[i]) ? a.f[(long long)l_n * a.f_np + pix * a.f_px + f_off[i]] : 1.f;
#endif
    raw.ok = (i == 0 ? 0 : raw.ok) | (ok ? 1 << i : 0);
  };
  auto fetch_advance = [&]() __attribute__((always_inline)) {
    const bool alive = l_slot < nslots;
    const int more = (alive && l_slot + 1 < nslots) ? 1 : 0;       // branch-free advance; past the end the cursor stays put
    l_slot += alive ? 1 : 0;
    const int wrap_c = more && l_col == W - 1;
    l_col = wrap_c ? -1 : l_col + more;
    l_rid += wrap_c ? nbx : 0;
    decode_row();
  };
  auto fetch = [&](Raw& raw) __attribute__((always_inline)) {
    fetch_frag(raw, 0);
    fetch_frag(raw, 1);
    fetch_advance();
  };
  // mask, split hi / lo, park in ring slot `slot` -- per channel pair j, then the two 16-byte stores
  u32x4 p_hi, p_lo;
  auto produce_pair = [&](const Raw& raw, int i, int j) __attribute__((always_inline)) {
#ifdef CMF_DBG_WG_NOPRODUCE
    asm volatile("" ::"v"(raw.v[i][0]), "v"(raw.v[i][1]), "v"(raw.f[i]));
    return;
#endif
    const bool ok = (raw.ok >> i) & 1;
    const float m = ok ? ((HASF && !p_gy[i]) ? (raw.f[i] > 0.f ? 1.f : 0.f) : 1.f) : 0.f;
    float e = raw.v[i][j >> 1][(2 * j) & 3], o = raw.v[i][j >> 1][(2 * j + 1) & 3];
    if (SELF && !p_gy[i]) e = fmaxf(e, 0.f), o = fmaxf(o, 0.f);
    e = ok ? e * m : 0.f, o = ok ? o * m : 0.f;
    const unsigned hb = pack2(e, o);
    p_hi[j] = hb;
    p_lo[j] = pack2(e - __builtin_bit_cast(float, hb << 16), o - __builtin_bit_cast(float, hb & 0xffff0000u));
  };
  auto produce_store = [&](int slot, int i) __attribute__((always_inline)) {
#ifdef CMF_DBG_WG_NOPRODUCE
    return;
#endif
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned char* dst = smem + slot * SLOT_BYTES + (2 * wave + i) * 2 * FRAG_BYTES + st_off;
    *reinterpret_cast<u32x2*>(dst) = u32x2{p_hi[0], p_hi[1]};                   // slice 0
    *reinterpret_cast<u32x2*>(dst + 512) = u32x2{p_hi[2], p_hi[3]};             // slice 1: fragment lanes kg + 2
    *reinterpret_cast<u32x2*>(dst + FRAG_BYTES) = u32x2{p_lo[0], p_lo[1]};
    *reinterpret_cast<u32x2*>(dst + FRAG_BYTES + 512) = u32x2{p_lo[2], p_lo[3]};
  };
  auto produce = [&](const Raw& raw, int slot, int i) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) produce_pair(raw, i, j);
    produce_store(slot, i);
  };
  auto frag = [&](int slot, int f, int hl) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(smem + slot * SLOT_BYTES + (f * 2 + hl) * FRAG_BYTES + lane * 16);
  };
  // centre slot C with its neighbours L, R: 18 tiles x 3 products = 27 MFMA pairs (two output-channel tiles each).  Between the
  // pairs go the twelve pieces of the producer role (8 channel pairs, 2 stores, 2 fetches), pinned there by sched_barrier: left
  // to itself hipcc clusters the ~150 VALU instructions of the split in front of the MFMAs, and with both waves of a SIMD in
  // phase behind the same barrier the two kinds of work then ran one after the other (timing-only builds: fetch + split alone
  // 1.18 ms, MFMA + split alone 1.12 ms, everything 1.84 ms).
  auto pair_of = [&](int jo, const bf16x8& g, const bf16x8& x, int tap) __attribute__((always_inline)) {
#ifndef CMF_DBG_WG_NOMFMA
    acc[jo][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g, x, acc[jo][tap], 0, 0, 0);
#endif
  };
  auto step_body = [&](int L, int Cc, int R, int P, Raw& raw) __attribute__((always_inline)) {
    bf16x8 gh[2], gl[2], xh[2], xl[2];
#pragma unroll
    for (int jo = 0; jo < 2; ++jo) gh[jo] = frag(Cc, 12 + 2 * co_h + jo, 0), gl[jo] = frag(Cc, 12 + 2 * co_h + jo, 1);
    xh[0] = frag(L, ci_t, 0), xl[0] = frag(L, ci_t, 1);
    auto piece = [&](int c) __attribute__((always_inline)) {      // producer piece c of 12
      if (c < 4) produce_pair(raw, 0, c);
      else if (c == 4) produce_store(P, 0);
      else if (c < 9) produce_pair(raw, 1, c - 5);
      else if (c == 9) produce_store(P, 1);
      else if (c == 10) fetch_frag(raw, 0);
      else if (c == 11) fetch_frag(raw, 1), fetch_advance();
    };
#pragma unroll
    for (int u = 0; u < 9; ++u) {                                   // u = dy * 3 + dx
      if (u + 1 < 9) {
        const int dy = (u + 1) / 3, dx = (u + 1) % 3, s = dx == 0 ? L : dx == 1 ? Cc : R;
        xh[(u + 1) & 1] = frag(s, dy * 4 + ci_t, 0), xl[(u + 1) & 1] = frag(s, dy * 4 + ci_t, 1);
      }
      const bf16x8 &h = xh[u & 1], &l = xl[u & 1];
      pair_of(0, gh[0], l, u), pair_of(1, gh[1], l, u);
      if ((3 * u) % 2 == 0 && 3 * u / 2 < 12) piece(3 * u / 2);
      __builtin_amdgcn_sched_barrier(0);
      pair_of(0, gl[0], h, u), pair_of(1, gl[1], h, u);
      if ((3 * u + 1) % 2 == 0 && (3 * u + 1) / 2 < 12) piece((3 * u + 1) / 2);
      __builtin_amdgcn_sched_barrier(0);
      pair_of(0, gh[0], h, u), pair_of(1, gh[1], h, u);
      if ((3 * u + 2) % 2 == 0 && (3 * u + 2) / 2 < 12) piece((3 * u + 2) / 2);
      __builtin_amdgcn_sched_barrier(0);
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
    // Global loads run FOUR slots ahead of their use (a register set per ring slot): with one set, fetched a single step ahead, a
    // step (~2000 cycles) was all the cover a load had against an HBM round trip of ~5500 under load, and 32 KB in flight per
    // CU bounded the kernel at ~2.4 TB/s of loads.
    Raw raw[4];
    fetch(raw[0]);
    fetch(raw[1]);
    fetch(raw[2]);
    fetch(raw[3]);                                  // slots 0 .. 3
    produce(raw[0], 0, 0), produce(raw[0], 0, 1);   // slot 0
    fetch(raw[0]);                                  // slot 4
    produce(raw[1], 1, 0), produce(raw[1], 1, 1);   // slot 1
    fetch(raw[1]);                                  // slot 5
    // step s (centre s), after ONE barrier: the MFMA work with the parking of slot s+2 (fetched four steps ago) and the fetch of
    // slot s+6 spread between its instructions (step_body).  Slot s+2 reuses the ring entry
    // of slot s-2, last read in step s-1 -- every wave finished that before it passed the barrier of step s; slot s+1 was parked
    // in step s-1.
    auto step = [&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      constexpr int L = (i + 3) & 3, Cc = i, R = (i + 1) & 3, P = (i + 2) & 3;
      __syncthreads();
      step_body(L, Cc, R, P, raw[P]);
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
      for (int i = 0; i < 4; ++i) out[(((2 * co_h + il) * 16 + 4 * kg + i) * 64 + ci_t * 16 + r) * 9 + t] = acc[il][t][i];
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
  const long long nrows = (long long)a->np * (a->nc / 32) * a->H;
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
