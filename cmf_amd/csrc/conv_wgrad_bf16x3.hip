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
//     channel tiles, masked by relu', and gy of row y x 4 channel tiles -- into a ring of four slots (4 x 32 KB of LDS), one
//     barrier per slot;
//   * ROLE-SPECIALISED waves (the forward kernel's structure): waves 4..7 only fetch (16-byte loads four slots ahead of their
//     use), split hi / lo and park -- four fragments each per slot; waves 0..3 only read fragments and issue MFMAs -- wave w owns
//     all four output-channel tiles x input-channel tile w x nine taps = 36 accumulator tiles (144 VGPRs): 26 ds_read_b128 and
//     108 MFMAs of 16 cycles per slot, against 72 fp32 MFMAs of 32 cycles for half the K in conv_wgrad.hip;
//   * rows are dealt XCD-aware so that the three uses of an input row are L2 hits (below).
// History of the structure, each step from a measurement (profiles/LABBOOK.md section 4.6): every wave playing both roles with two fragments and 18
// tiles each ran 210 -> 296 TFLOP/s fp32-equivalent with deeper prefetch, fewer fragment reads and the row dealing, and stopped
// there -- its fetch + split half alone took 1.0 ms, its MFMA half 1.2 ms, together 1.6 ms (B = 128, 28 x 28): a SIMD does not
// overlap one wave's VALU / SALU stream with another wave's MFMAs for free.  Stamps on the role-specialised form then showed the
// MFMA waves computing 2350 cycles of a 4000-cycle step and waiting for the producers the rest: the kernel is bound by the
// producers' INSTRUCTION COUNT.  Row-invariant work hoisted to the row change and one select per value instead of a multiply and
// a select: 560 -> 400 instructions per producer step, both roles at ~2400 cycles per step, 344 TFLOP/s.
// Shapes: taps = 9, cin % 64 == 0, cout % 64 == 0, nc % 32 == 0, factor NONE, RELU from a float tensor or from a bit mask
// (CMF_F_RELU_BITS: one byte per (sample, pixel, 8 channels) -- a fragment's sixteen masks are two neighbouring bytes, where the
// float form touches sixteen cache lines), or SELF_RELU; everything else
// stays on the fp32 kernel.  Partial blocks and their fixed-order reduction are shared with conv_wgrad.hip.
#include "common.h"
#include <type_traits>

#ifdef CMF_DBG_WGSTAMP
// diagnostic build only: phase timestamps of workgroup 0 (wave 0 = MFMA role, wave 4 = producer role), tools/read_wg_stamps.py
__device__ unsigned long long cmf_dbg_wg_stamps[2][64][4];
extern "C" int cmf_debug_read_wg_stamps(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cmf_dbg_wg_stamps), sizeof(cmf_dbg_wg_stamps));
}
#define WSTAMP(role, g, k)                                                                       \
  do {                                                                                            \
    if (blockIdx.x == 0 && (g) < 64 && lane == 0 && wave == ((role) == 1 ? 4 : 0)) {              \
      unsigned long long t_;                                                                      \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
      cmf_dbg_wg_stamps[role][g][k] = t_;                                                         \
    }                                                                                             \
  } while (0)
#else
#define WSTAMP(role, g, k) do {} while (0)
#endif

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

// what one producer wave fetches for its four fragments of one slot
struct Raw4 {
  f32x4 v[4][2];
  float f[4];
  bool ok;                                           // wave-uniform: column and row inside the image
};
// Several problems of ONE shape in one launch (cmf_conv_tangent_wgrad_bf16x3_batched): the weight gradients of the 16 hidden primal
// convs of a coupler at a training shard's 2 - 4 sample groups are 28 - 56 image rows each -- 28 - 56 of 256 workgroups busy, 38 - 43 us
// per launch, 320 launches per step.  Problem p owns workgroups [p wgp, (p + 1) wgp): the row dealing below runs inside that range.
struct WgBatch {
  const float* x[CMF_WGRAD_MAX_BATCH];
  const float* gy[CMF_WGRAD_MAX_BATCH];
  float* dw[CMF_WGRAD_MAX_BATCH];
  int wgp;                                           // workgroups per problem (a multiple of 8: the XCD dealing stays aligned)
};

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_wgrad3x3_roles_kernel(cmf_conv_tangent_args a, WgBatch P, float* __restrict__ ws,
                                                                     int co0, int ci0, int nrows) {
  const int prob = blockIdx.x / P.wgp;
  a.x = P.x[prob];
  const float* __restrict__ gy = P.gy[prob];
  constexpr bool BITS = MODE == 3, HASF = MODE == 1 || BITS, SELF = MODE == 2;   // BITS: relu' from a bit mask (CMF_F_RELU_BITS)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = a.W, H = a.H, nsp = a.nc / 32;

  // This workgroup's image rows; a row is W + 1 slots (conv_wgrad.hip).  Row id = (sample * nsp + slice pair) * H + y, y FASTEST,
  // and the rows are dealt so that the workgroups running on one XCD (private L2; workgroups go round-robin over the 8 XCDs) work
  // on CONSECUTIVE rows at the same time: XCD k owns the contiguous range [xstart, xstart + xlen) and its j-th workgroup takes rows
  // xstart + j, + nbx, + 2 nbx ...  Input row y is needed by output rows y-1, y, y+1; with each workgroup walking its own block of
  // rows the second and third use came ~1.8 MB later in that workgroup's stream -- 58 MB per XCD against 4 MB of L2 -- and every
  // input row was fetched from HBM three times (6.4 GB per launch at B = 128; rocprofv3 FETCH_SIZE now: 3.45 GB against 3.2 GB
  // algorithmic).  Speed only: any dealing covers every row exactly once.
  int xstart, xlen, nbx, jx;
  {
    const int G = P.wgp, NX = G < 8 ? G : 8, bid = blockIdx.x - prob * P.wgp, xcd = bid % NX;
    const int q = nrows / NX, rem = nrows % NX;
    xstart = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    xlen = q + (xcd < rem ? 1 : 0);
    nbx = (G - xcd + NX - 1) / NX;
    jx = bid / NX;
  }
  const int my_rows = jx < xlen ? (xlen - jx + nbx - 1) / nbx : 0;
  const int nslots = my_rows * (W + 1);
  const int nsteps = (nslots + 3) & ~3;              // both roles run exactly this many barriers

  if (wave >= 4) {
    // ================================ producer waves ================================
    const int pw = wave - 4;
    const long long xsl = a.x_sl ? a.x_sl : 16, ysl = a.y_sl ? a.y_sl : 16;
    // A producer lane (lc, lg) = (lane / 4, lane % 4) fetches columns 4 lg .. 4 lg + 3 of channel lc in BOTH slices of the pair (one
    // 16-byte load each): in the slice-major layout the 64 lanes of a load cover one contiguous KiB (16 channels x 64 B).  The values
    // reach the fragment layout (lane (r, kg): columns 8 (kg & 1) .. + 7 of slice kg / 2) through the LDS address of the four 8-byte
    // stores: this lane's columns of slice s are the (lg & 1) half of fragment lane (lc, 2 s + lg / 2).
    const int lc = lane >> 2, lg = lane & 3;
    // the four fragments of a producer wave are the four channel tiles of ONE tensor row: waves 4..6 the input rows y-1, y, y+1
    // (fragments 0..11 = row dy x tile), wave 7 the gy row (fragments 12..15) -- so the row state below is per wave, not per fragment
    const bool is_gy = pw == 3;
    const int dy = is_gy ? 1 : pw;
    const long long sl_off = is_gy ? ysl : xsl;
    const int px_step = (int)(is_gy ? a.y_px : a.x_px), f_px = BITS ? a.cin / 8 : (int)a.f_px;   // BITS: bytes per pixel
    long long lane_off[4], f_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long ch = (is_gy ? co0 : ci0) + i * 16 + lc;
      lane_off[i] = ch * (is_gy ? a.y_co : a.x_ci) + 4 * lg;
      f_off[i] = BITS ? ch >> 3 : ch * a.f_ci;       // BITS: byte of the channel's octet inside the pixel's cin / 8 bytes
    }
    const int f_bit = lc & 7;                        // ... and the channel's bit in it (channel tiles start at multiples of 16)
    const int st_off = (((lg >> 1) * 16 + lc) << 4) + ((lg & 1) << 3);
    // Producer instruction count is what bounds this kernel (stamps: an MFMA wave computes 2350 cycles of a 4000-cycle step and
    // waits for the producers the rest of it): everything that only changes with the image row -- the row decode with its
    // integer divisions, the 64-bit row base, the row validity -- is recomputed at a row change only (a uniform branch), and the
    // relu' / validity mask is ONE select per value (a lane's eight values belong to one channel of one pixel).
    int l_slot = 0, l_col = -1, l_rid = xstart + jx;
    const float* rbase = a.x;
    const float* fbase = a.f;
    bool row_ok = false;
    auto set_row = [&]() __attribute__((always_inline)) {
      const int yy = l_rid % H, t = l_rid / H, sp = t % nsp, n = t / nsp;
      const int y2 = yy + dy - 1;
      const long long rowpix = (long long)(y2 < 0 ? 0 : y2 >= H ? H - 1 : y2) * W;
      rbase = is_gy ? gy + (long long)n * a.y_np + rowpix * a.y_px + 2 * sp * ysl
                    : a.x + (long long)n * a.x_np + rowpix * a.x_px + 2 * sp * xsl;
      if (BITS) fbase = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.f) + (long long)n * a.f_np + rowpix * f_px);
      else if (HASF) fbase = a.f + (long long)n * a.f_np + rowpix * a.f_px;
      row_ok = y2 >= 0 && y2 < H;
    };
    if (my_rows > 0) set_row();
    auto fetch = [&](Raw4& raw) __attribute__((always_inline)) {
      const bool alive = l_slot < nslots;
      const int colc = l_col < 0 ? 0 : l_col >= W ? W - 1 : l_col;
      const float* base = rbase + colc * px_step;                  // 32-bit product: W * pixel stride < 2^31 (launcher)
#pragma unroll
      for (int i = 0; i < 4; ++i) {                                // every load unconditional (clamped address), validity a select
        raw.v[i][0] = *reinterpret_cast<const f32x4*>(base + lane_off[i]);
        raw.v[i][1] = *reinterpret_cast<const f32x4*>(base + lane_off[i] + sl_off);
        if (BITS && !is_gy) {                                      // the raw byte; decoded in produce (touching it here would wait for the load)
          raw.f[i] = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned char*>(fbase)[colc * f_px + f_off[i]]);
        } else {
          raw.f[i] = (HASF && !is_gy) ? fbase[colc * f_px + f_off[i]] : 1.f;
        }
      }
      raw.ok = alive && l_col >= 0 && l_col < W && row_ok;
      const bool more = alive && l_slot + 1 < nslots;              // past the end the cursor stays put
      l_slot += alive ? 1 : 0;
      if (more && l_col == W - 1) {                                // row change (wave-uniform)
        l_col = -1;
        l_rid += nbx;
        set_row();
      } else {
        l_col += more ? 1 : 0;
      }
    };
    auto produce = [&](const Raw4& raw, int slot) __attribute__((always_inline)) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool fon = BITS ? ((__builtin_bit_cast(unsigned, raw.f[i]) >> f_bit) & 1u) != 0 : raw.f[i] > 0.f;
        const bool on = raw.ok && (!(HASF && !is_gy) || fon);
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float e = raw.v[i][j >> 1][(2 * j) & 3], o = raw.v[i][j >> 1][(2 * j + 1) & 3];
          if (SELF && !is_gy) e = fmaxf(e, 0.f), o = fmaxf(o, 0.f);
          e = on ? e : 0.f, o = on ? o : 0.f;
          const unsigned hb = pack2(e, o);
          hi[j] = hb;
          lo[j] = pack2(e - __builtin_bit_cast(float, hb << 16), o - __builtin_bit_cast(float, hb & 0xffff0000u));
        }
        unsigned char* dst = smem + slot * SLOT_BYTES + (4 * pw + i) * 2 * FRAG_BYTES + st_off;
        *reinterpret_cast<u32x2*>(dst) = u32x2{hi[0], hi[1]};                   // slice 0
        *reinterpret_cast<u32x2*>(dst + 512) = u32x2{hi[2], hi[3]};             // slice 1: fragment lanes kg + 2
        *reinterpret_cast<u32x2*>(dst + FRAG_BYTES) = u32x2{lo[0], lo[1]};
        *reinterpret_cast<u32x2*>(dst + FRAG_BYTES + 512) = u32x2{lo[2], lo[3]};
      }
    };
    if (nslots > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {                   // ring slot 3 stands for the column left of slot 0: zeros
        unsigned char* dst = smem + 3 * SLOT_BYTES + (4 * pw + i) * 2 * FRAG_BYTES + lane * 16;
        *reinterpret_cast<u32x4*>(dst) = u32x4{0, 0, 0, 0};
        *reinterpret_cast<u32x4*>(dst + FRAG_BYTES) = u32x4{0, 0, 0, 0};
      }
      Raw4 raw[4];
      fetch(raw[0]);
      fetch(raw[1]);
      fetch(raw[2]);
      fetch(raw[3]);                                  // slots 0 .. 3
      produce(raw[0], 0);
      fetch(raw[0]);                                  // slot 4
      produce(raw[1], 1);
      fetch(raw[1]);                                  // slot 5
      // step s, after its barrier: park slot s+2 (ring entry of slot s-2, last read in step s-1), fetch slot s+6
      int sbase = 0;
      auto pstep = [&](auto I) __attribute__((always_inline)) {
        constexpr int P = (decltype(I)::value + 2) & 3;
        WSTAMP(1, sbase + decltype(I)::value, 0);
        __syncthreads();
        WSTAMP(1, sbase + decltype(I)::value, 1);
        produce(raw[P], P);
        WSTAMP(1, sbase + decltype(I)::value, 2);
        fetch(raw[P]);
        WSTAMP(1, sbase + decltype(I)::value, 3);
      };
      for (int s = 0; s < nsteps; s += 4, sbase += 4) {
        pstep(std::integral_constant<int, 0>{});
        pstep(std::integral_constant<int, 1>{});
        pstep(std::integral_constant<int, 2>{});
        pstep(std::integral_constant<int, 3>{});
      }
    }
    return;
  }

  // ================================== MFMA waves ==================================
  const int ci_t = wave, r = lane & 15, kg = lane >> 4;
  f32x4 acc[4][9];
#pragma unroll
  for (int jo = 0; jo < 4; ++jo)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[jo][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto frag = [&](int slot, int f, int hl) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(smem + slot * SLOT_BYTES + (f * 2 + hl) * FRAG_BYTES + lane * 16);
  };
  int sbase = 0;
  auto mstep = [&](auto I) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    constexpr int L = (i + 3) & 3, Cc = i, R = (i + 1) & 3;
    WSTAMP(0, sbase + i, 0);
    __syncthreads();
    WSTAMP(0, sbase + i, 1);
    bf16x8 gh[4], gl[4], xh[2], xl[2];
    xh[0] = frag(L, ci_t, 0), xl[0] = frag(L, ci_t, 1);
#pragma unroll
    for (int jo = 0; jo < 4; ++jo) gh[jo] = frag(Cc, 12 + jo, 0), gl[jo] = frag(Cc, 12 + jo, 1);
#pragma unroll
    for (int u = 0; u < 9; ++u) {                     // u = dy * 3 + dx; the next unit's fragments are read one unit ahead
      if (u + 1 < 9) {
        const int dy = (u + 1) / 3, dx = (u + 1) % 3, s = dx == 0 ? L : dx == 1 ? Cc : R;
        xh[(u + 1) & 1] = frag(s, dy * 4 + ci_t, 0), xl[(u + 1) & 1] = frag(s, dy * 4 + ci_t, 1);
      }
      const bf16x8 &h = xh[u & 1], &l = xl[u & 1];
#pragma unroll
      for (int jo = 0; jo < 4; ++jo) acc[jo][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[jo], l, acc[jo][u], 0, 0, 0);
#pragma unroll
      for (int jo = 0; jo < 4; ++jo) acc[jo][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl[jo], h, acc[jo][u], 0, 0, 0);
#pragma unroll
      for (int jo = 0; jo < 4; ++jo) acc[jo][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[jo], h, acc[jo][u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u == 0) WSTAMP(0, sbase + i, 2);
    }
    WSTAMP(0, sbase + i, 3);
  };
  if (nslots > 0) {
    for (int s = 0; s < nsteps; s += 4, sbase += 4) {
      mstep(std::integral_constant<int, 0>{});
      mstep(std::integral_constant<int, 1>{});
      mstep(std::integral_constant<int, 2>{});
      mstep(std::integral_constant<int, 3>{});
    }
  }
  float* out = ws + (size_t)blockIdx.x * 64 * 64 * 9;
#pragma unroll
  for (int jo = 0; jo < 4; ++jo)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[((jo * 16 + 4 * kg + i) * 64 + ci_t * 16 + r) * 9 + t] = acc[jo][t][i];
}

// dw[(co0 + co)][ci0 + ci][tap] += sum_wg ws[wg][co][ci][tap]   (fixed order; same as conv_wgrad.hip)
__global__ __launch_bounds__(256) void wgrad_reduce_split_kernel(const float* __restrict__ ws, WgBatch P, int nwg, int co0,
                                                                 int ci0, int cin) {
  const int e = blockIdx.x * 256 + threadIdx.x, per = 64 * 64 * 9;
  if (e >= per) return;
  ws += (size_t)blockIdx.y * nwg * per;               // problem blockIdx.y: its workgroups' partial blocks
  float* __restrict__ dw = P.dw[blockIdx.y];
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

static int wgrad_split_launch(const cmf_conv_tangent_args* a, int nprob, const float* const* xs, const float* const* gys,
                              float* const* dws, float* ws, long long ws_bytes, void* stream) {
  if (!a || !ws || nprob < 1 || nprob > CMF_WGRAD_MAX_BATCH) return CMF_EINVAL;
  if (a->taps != 9 || a->cin % 64 || a->cout % 64 || a->nc <= 0 || a->nc % 32) return CMF_EINVAL;
  if (a->np <= 0 || a->H <= 0 || a->W <= 0) return CMF_EINVAL;
  if (a->fmode != CMF_F_NONE && a->fmode != CMF_F_RELU && a->fmode != CMF_F_SELF_RELU && a->fmode != CMF_F_RELU_BITS) return CMF_EINVAL;
  if ((a->fmode == CMF_F_RELU || a->fmode == CMF_F_RELU_BITS) && (!a->f || a->f_group > 1 || nprob > 1)) return CMF_EINVAL;
  if (a->fmode == CMF_F_RELU_BITS && a->f_np < (long long)a->H * a->W * (a->cin / 8)) return CMF_EINVAL;   // f_np in bytes
  if (ws_bytes < (long long)WG_MAX * 64 * 64 * 9 * (long long)sizeof(float)) return CMF_EINVAL;
  if ((a->x_np | a->x_ci | a->x_px | a->x_sl | a->y_np | a->y_co | a->y_px | a->y_sl) % 4) return CMF_EINVAL;
  WgBatch P;
  for (int p = 0; p < CMF_WGRAD_MAX_BATCH; ++p) {
    const int q = p < nprob ? p : 0;
    if (!xs[q] || !gys[q] || !dws[q] || ((uintptr_t)xs[q] | (uintptr_t)gys[q]) % 16) return CMF_EINVAL;
    P.x[p] = xs[q], P.gy[p] = gys[q], P.dw[p] = dws[q];
  }
  const long long nrows = (long long)a->np * (a->nc / 32) * a->H;
  if (nrows > 0x7fffffffLL / (a->W + 2)) return CMF_ERANGE;
  if ((long long)a->W * a->x_px > 0x7fffffffLL || (long long)a->W * a->y_px > 0x7fffffffLL || (long long)a->W * a->f_px > 0x7fffffffLL)
    return CMF_ERANGE;                                             // the producers form column * pixel-stride in 32 bits
  // one problem: up to WG_MAX workgroups; a batch: WG_MAX / nprob each, rounded down to a multiple of 8 (at least 8)
  int wgp = nprob == 1 ? (int)(nrows < WG_MAX ? nrows : WG_MAX) : (WG_MAX / nprob) & ~7;
  if (nprob > 1 && wgp < 8) return CMF_EINVAL;
  P.wgp = wgp;
  const int grid = wgp * nprob;
  hipStream_t s = (hipStream_t)stream;
  {                                                                // per (device, kernel) memo: runtime.hip
    hipError_t e = cmf_set_dynamic_lds((const void*)conv_wgrad3x3_roles_kernel<0>, LDS_BYTES);
    if (e == hipSuccess) e = cmf_set_dynamic_lds((const void*)conv_wgrad3x3_roles_kernel<1>, LDS_BYTES);
    if (e == hipSuccess) e = cmf_set_dynamic_lds((const void*)conv_wgrad3x3_roles_kernel<2>, LDS_BYTES);
    if (e == hipSuccess) e = cmf_set_dynamic_lds((const void*)conv_wgrad3x3_roles_kernel<3>, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
  }
  for (int co0 = 0; co0 < a->cout; co0 += 64)
    for (int ci0 = 0; ci0 < a->cin; ci0 += 64) {
      if (a->fmode == CMF_F_RELU_BITS)
        hipLaunchKernelGGL(conv_wgrad3x3_roles_kernel<3>, dim3(grid), dim3(512), LDS_BYTES, s, *a, P, ws, co0, ci0, (int)nrows);
      else if (a->fmode == CMF_F_RELU)
        hipLaunchKernelGGL(conv_wgrad3x3_roles_kernel<1>, dim3(grid), dim3(512), LDS_BYTES, s, *a, P, ws, co0, ci0, (int)nrows);
      else if (a->fmode == CMF_F_SELF_RELU)
        hipLaunchKernelGGL(conv_wgrad3x3_roles_kernel<2>, dim3(grid), dim3(512), LDS_BYTES, s, *a, P, ws, co0, ci0, (int)nrows);
      else
        hipLaunchKernelGGL(conv_wgrad3x3_roles_kernel<0>, dim3(grid), dim3(512), LDS_BYTES, s, *a, P, ws, co0, ci0, (int)nrows);
      CMF_LAUNCH_CHECK();
      hipLaunchKernelGGL(wgrad_reduce_split_kernel, dim3(cmf_ceil_div(64 * 64 * 9, 256), nprob), dim3(256), 0, s, ws, P, wgp, co0, ci0, a->cin);
      CMF_LAUNCH_CHECK();
    }
  return 0;
}

extern "C" int cmf_conv_tangent_wgrad_bf16x3(const cmf_conv_tangent_args* a, const float* gy, float* dw, float* ws,
                                             long long ws_bytes, void* stream) {
  if (!a || !a->x || !gy || !dw) return CMF_EINVAL;
  const float* xs[1] = {a->x};
  const float* gys[1] = {gy};
  float* dws[1] = {dw};
  return wgrad_split_launch(a, 1, xs, gys, dws, ws, ws_bytes, stream);
}

extern "C" int cmf_conv_tangent_wgrad_bf16x3_batched(const cmf_conv_tangent_args* a, int nprob, const float* const* x,
                                                     const float* const* gy, float* const* dw, float* ws, long long ws_bytes,
                                                     void* stream) {
  if (!x || !gy || !dw) return CMF_EINVAL;
  return wgrad_split_launch(a, nprob, x, gy, dw, ws, ws_bytes, stream);
}
