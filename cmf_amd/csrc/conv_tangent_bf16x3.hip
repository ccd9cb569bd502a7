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

#ifdef CMF_DBG_STAMP
// diagnostic build only: phase timestamps of workgroup 0 (wave 0 = MFMA role, wave 4 = loader role)
__device__ unsigned long long cmf_dbg_stamps[3][64][4];
extern "C" int cmf_debug_read_stamps(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cmf_dbg_stamps), sizeof(cmf_dbg_stamps));
}
#define STAMP(role, g, k)                                                                        \
  do {                                                                                            \
    if (blockIdx.x == 0 && (g) < 64 && lane == 0 && wave == ((role) == 1 ? 4 : 0)) {                   \
      unsigned long long t_;                                                                      \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
      cmf_dbg_stamps[role][g][k] = t_;                                                            \
    }                                                                                             \
  } while (0)
#else
#define STAMP(role, g, k) do {} while (0)
#endif

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
  static constexpr int BUF_BYTES = 2 * XS_BYTES + 2 * WS_BYTES;   // one pipeline stage
  static constexpr int W_CHUNK_BYTES = 2 * 3 * 4 * 4 * 16 * 16;   // global: [hl][s][cot 4][kq][co][8] bf16 = 24 KiB
  static constexpr int NX_ITEMS = PIXH * 4;                       // (pixel, column quad): one per loader thread
  static constexpr int NW_ITEMS = 2 * 3 * COT * 4 * 16;           // 16-byte items of the W chunk actually used
  static constexpr int NWIT = (NW_ITEMS + 255) / 256;
  static_assert(NX_ITEMS <= 256, "one staging item per loader thread");
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

// 16-byte slot of column `col` inside pixel `pix`'s 256-byte row of the X image.  The permutation makes the eight
// lanes of a ds_write_b128 lane group (4 column quads of two neighbouring pixels, same column-in-quad) cover eight
// distinct 16-byte residues mod 128 B (conflict-free; the natural order is 4-way conflicted), while a 16-lane read
// group still reads a permutation of one 256-byte row (conflict-free ds_read_b128).
__device__ __forceinline__ int xslot(int col, int pix) { return (((col & 3) << 2) + (col >> 2) + ((pix & 1) << 2)) & 15; }

// PERSISTENT, role-specialised kernel: one workgroup of 8 waves per CU streams through its share of the work items
// (item = pixel tile x 16-column slice x co group x sample).  Measured on the way here: with every wave alternating
// staging and MFMA the two phases hardly overlap (MFMA pipe 37 % busy); with one non-persistent 112 KiB workgroup
// per CU the dispatch gap, the first-load latency and the store tail of EVERY tile are exposed.
//   waves 0..3  MFMA waves: LDS fragment reads (software-pipelined ring) + v_mfma only.  wave = 2*row + cohalf owns
//               one tile row (PW = 2*PXW pixels) x CW = COT/2 output-channel tiles = 112 accumulator VGPRs.
//   waves 4..7  loader waves: global -> registers (three register sets: two chunks of memory-latency slack) ->
//               activation-derivative factor, hi/lo split -> LDS stage.  Their chunk stream runs across item
//               boundaries, so only the first chunk of the launch sees memory latency.
// Two LDS stages (2 x 56 KiB), ONE barrier per 8-channel chunk; both roles execute the same number of barriers.
// SELF = the input's own elementwise relu is applied on load (primal data in the column slots); a compile-time switch:
// as a run-time select it made hipcc spill 77 VGPRs in the loader's commit.
template <int COT, int PXW, bool SELF>
__global__ __launch_bounds__(512, 2) void conv_tangent_bf16x3_kernel(cmf_conv_tangent_args a, int tiles_x, int ntiles,
                                                                       int nslices, int ncog, int total) {
  using C = BCfg<COT, PXW>;
  constexpr int CW = COT / 2, PW = 2 * PXW;
  static_assert(COT % 2 == 0, "the co-split wave layout needs an even number of output-channel tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 stages x [Xh | Xl | Wh | Wl]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int kq = lane >> 4, cl = lane & 15;
  const int nchunks = a.cin / 8;

  // XCD-aware item list.  Workgroups are dealt round-robin over the 8 XCDs (private L2 each): XCD k owns the
  // contiguous logical range [xstart, xstart + xlen), ordered slice-fastest, then tile, then co group, then sample,
  // and its workgroups take items j, j + nbx, j + 2 nbx ... so that what runs concurrently on an XCD shares 128-byte
  // lines (column slices) and halos (neighbouring tiles) in its L2.  Speed only, never correctness.
  int xstart, xlen, nbx, jx;
  {
    const int G = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
    const int q = total >> 3, r = total & 7;
    xstart = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xlen = q + (xcd < r ? 1 : 0);
    nbx = (G - xcd + 7) >> 3;
    jx = bid >> 3;
  }
  const int n_items = jx < xlen ? (xlen - jx + nbx - 1) / nbx : 0;
  const int total_chunks = n_items * nchunks;
  auto decode = [&](int item, int& tile, int& slice, int& cog, int& np) {
    int w = xstart + jx + item * nbx;
    slice = w % nslices;
    w /= nslices;
    tile = w % ntiles;
    w /= ntiles;
    cog = w % ncog;
    np = w / ncog;
  };

  if (loader) {
    // =============================== loader waves ===============================
    // Measured with in-kernel stamps: a loader's ~220 VALU + 14 ds_write per chunk took ~4000 cycles beside an MFMA
    // wave on the same SIMD and set the chunk period.  VALU issue is arbitrated by priority, then age: raise the
    // loaders (an MFMA wave needs one issue slot per 16 cycles and barely notices).
    __builtin_amdgcn_s_setprio(3);
    const int lt = tid - 256;                                    // 0..255: staging item = (pixel, column quad)
    const int x_ci = (int)a.x_ci, x_px = (int)a.x_px, f_ci = (int)a.f_ci, f_px = (int)a.f_px;
    const int q = lt & 3, pix = lt >> 2;
    const int hy = pix / C::TWH, hx = pix % C::TWH;
    // NONE: 1   RELU: [f>0]   TANH: 1 - f^2   RAW: f      as  okf * (c0 + c1*[f>0] + c2*f + c3*f*f)
    const float fc0 = (a.fmode == CMF_F_NONE || a.fmode == CMF_F_TANH || a.fmode == CMF_F_SELF_RELU) ? 1.f : 0.f;
    const float fc1 = (a.fmode == CMF_F_RELU) ? 1.f : 0.f;
    const float fc2 = (a.fmode == CMF_F_RAW) ? 1.f : 0.f;
    const float fc3 = (a.fmode == CMF_F_TANH) ? -1.f : 0.f;
    const bool has_f = a.fmode != CMF_F_NONE && a.fmode != CMF_F_SELF_RELU;
    const int fgrp = a.f_group > 1 ? a.f_group : 1;

    // prefetch cursor: which (work item, chunk) the next prefetch fetches, with that item's addressing state.
    // Bases are wave-uniform; per-lane offsets are unsigned 32-bit BYTE offsets (SGPR-base loads, no 64-bit per-lane
    // addresses kept alive).  Validity is applied arithmetically at commit time: every load is unconditional and
    // always consumed (a select on validity lets hipcc sink loads into branches + s_waitcnt vmcnt(0)).
    int cur_item = 0, cur_ch = 0;
    const float* xb = a.x;
    const float* fb = a.x;
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w);
    unsigned xo = 0, fo = 0;
    float okf = 0.f;
    auto set_item = [&](int item) {
      int tile, slice, cog, np;
      decode(item, tile, slice, cog, np);
      const int y0 = 2 * (tile / tiles_x), x0 = C::TW * (tile % tiles_x);
      xb = a.x + (long long)np * a.x_np + slice * 16;
      fb = a.f ? a.f + (long long)(np / fgrp) * a.f_np + (np % fgrp) : a.x;
      wb = reinterpret_cast<const unsigned char*>(a.w) + (long long)cog * nchunks * C::W_CHUNK_BYTES;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = lt < C::NX_ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int gpix = gy * a.W + gx;
      okf = ok ? 1.f : 0.f;
      xo = ok ? 4u * (unsigned)(gpix * x_px + q * 4) : 0u;
      fo = ok ? 4u * (unsigned)(gpix * f_px) : 0u;
    };

    struct Regs {
      f32x4 x[8];
      float f[8];
      u32x4 w[C::NWIT];
      float okf;
    };
    Regs r0, r1, r2;                                               // chunk g of the stream lives in set g % 3
#pragma unroll
    for (int j = 0; j < 8; ++j) r0.f[j] = r1.f[j] = r2.f[j] = 0.f;  // stays 0 when fmode == NONE (never loaded)

    auto prefetch = [&](Regs& r) {
      const int ch = cur_ch;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned off = 4u * (unsigned)((ch * 8 + j) * x_ci) + xo;
#ifndef CMF_DBG_NOLOAD
        r.x[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(xb) + off);
#else
        r.x[j] = f32x4{(float)off, 1.f, 2.f, 3.f};
#endif
      }
      if (has_f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned off = 4u * (unsigned)((ch * 8 + j) * f_ci) + fo;
          r.f[j] = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(fb) + off);
        }
      }
      const unsigned wco = (unsigned)(ch * C::W_CHUNK_BYTES);
#pragma unroll
      for (int it = 0; it < C::NWIT; ++it) {
        int i = lt + 256 * it;
        i = i < C::NW_ITEMS ? i : C::NW_ITEMS - 1;
        unsigned off;
        if (COT == 4) {
          off = (unsigned)i << 4;                                  // the slab is consumed whole: a linear copy
        } else {                                                   // the global slab always has 4 co tiles per K-step
          const int rest = i & 63, t = i >> 6;
          const int cot = t % COT, s = (t / COT) % 3, hl = t / (3 * COT);
          off = (unsigned)((((hl * 3 + s) * 4 + cot) * 64 + rest) << 4);
        }
        r.w[it] = *reinterpret_cast<const u32x4*>(wb + (wco + off));
      }
      r.okf = okf;
      if (++cur_ch == nchunks) {                                   // advance the cursor (wave-uniform)
        cur_ch = 0;
        if (++cur_item < n_items) set_item(cur_item);
      }
    };

    auto commit = [&](Regs& r, int stage) {
#ifdef CMF_DBG_NOCOMMIT
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(r.x[j]), "v"(r.f[j]));
#pragma unroll
      for (int it = 0; it < C::NWIT; ++it) asm volatile("" ::"v"(r.w[it]));
      return;
#endif
      unsigned char* Xh = smem + stage * C::BUF_BYTES;
      unsigned char* Xl = Xh + C::XS_BYTES;
      unsigned char* Wh = Xh + 2 * C::XS_BYTES;
      unsigned char* Wl = Wh + C::WS_BYTES;
      if (lt < C::NX_ITEMS) {
        float v[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = r.f[j];
          const float m = r.okf * (fc0 + fc1 * (f > 0.f ? 1.f : 0.f) + f * (fc2 + fc3 * f));
#pragma unroll
          for (int c = 0; c < 4; ++c) v[j][c] = (SELF ? fmaxf(r.x[j][c], 0.f) : r.x[j][c]) * m;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                              // column q*4 + c: 8 channels -> 16 B hi + 16 B lo
          u32x4 h, l;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float e0, e1;
            h[jj] = pack_hi(v[2 * jj][c], v[2 * jj + 1][c], e0, e1);
            l[jj] = pack_lo(e0, e1);
          }
          const int off = ((pix * 16 + xslot(q * 4 + c, pix)) << 4);
          *reinterpret_cast<u32x4*>(Xh + off) = h;
          *reinterpret_cast<u32x4*>(Xl + off) = l;
        }
      }
#pragma unroll
      for (int it = 0; it < C::NWIT; ++it) {
        const int i = lt + 256 * it;
        if (i < C::NW_ITEMS) {
          const int rest = i & 63, t = i >> 6;
          const int cot = t % COT, s = (t / COT) % 3, hl = t / (3 * COT);
          unsigned char* dst = (hl ? Wl : Wh) + ((((s * COT + cot) * 64) + rest) << 4);
          *reinterpret_cast<u32x4*>(dst) = r.w[it];
        }
      }
    };

    if (n_items > 0) set_item(0);
    if (total_chunks > 0) prefetch(r0);
    if (total_chunks > 1) prefetch(r1);
    if (total_chunks > 2) prefetch(r2);
    if (total_chunks > 0) commit(r0, 0);
    __syncthreads();                                               // stage 0 ready
    // iteration g: the MFMA waves consume stage g&1 (stream chunk g); set g%3 is free -> stream chunk g+3;
    // stage (g+1)&1 <- chunk g+1 from set (g+1)%3.  Exactly one barrier per iteration, matching the MFMA waves.
    auto iter = [&](int g, Regs& freed, Regs& next) {
      STAMP(1, g, 0);
      if (g + 3 < total_chunks) prefetch(freed);
      STAMP(1, g, 1);
#ifdef CMF_DBG_STAMP
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + 8 + C::NWIT));   // the loads of `next` have landed (this set's may fly)
      STAMP(1, g, 2);
#endif
      if (g + 1 < total_chunks) commit(next, (g + 1) & 1);
      STAMP(1, g, 3);
      __syncthreads();
    };
    for (int g = 0; g < total_chunks; g += 3) {
      iter(g, r0, r1);
      if (g + 1 >= total_chunks) break;
      iter(g + 1, r1, r2);
      if (g + 2 >= total_chunks) break;
      iter(g + 2, r2, r0);
    }
    return;
  }

  // ================================= MFMA waves =================================
  const int wrow = wave >> 1, cohalf = wave & 1;
  const int y_co = (int)a.y_co, y_px = (int)a.y_px, r_co = (int)a.r_co, r_px = (int)a.r_px;

  // Per-lane B offsets: lane group kq of K-step s reads tap 4*s + kq (taps >= 9 read tap 8's data against zero
  // weights) of pixel (wrow + dy, p + dx); the slot permutation depends on the pixel's parity, i.e. on (dx + p) & 1
  // (the halo row length TWH is even).
  int boff[3][2];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    int tap = 4 * s + kq;
    tap = tap < 9 ? tap : 8;
    const int pbase = (wrow + tap / 3) * C::TWH + tap % 3;         // staged pixel of p = 0
#pragma unroll
    for (int par = 0; par < 2; ++par) boff[s][par] = ((pbase * 16 + xslot(cl, pbase + par)) << 4);
  }
  const int aoff = (lane << 4) + ((cohalf * CW * 64) << 4);

  // Fragment pipeline: step t = s*PW + p consumes B fragment pair t; pairs are fetched BD-1 steps ahead into a ring
  // of BD register pairs, the A fragments of K-step s+1 are fetched at the start of K-step s into the alternate
  // set.  Everything is unrolled, so ring slots are static registers and hipcc emits counted lgkmcnt waits.
  constexpr int BD = 4, NSTEP = 3 * PW;

  // Item context: output / residual bases and validity of this wave's tile row.
  struct Item {
    float* y;
    const float* r;
    int y0, x0, co0;
    bool full;
  };
  auto make_item = [&](int item) {
    int tile, slice, cog, np;
    decode(item, tile, slice, cog, np);
    Item it;
    it.y0 = 2 * (tile / tiles_x);
    it.x0 = C::TW * (tile % tiles_x);
    // lane (kq, cl) owns columns slice*16 + kq*4 .. +3 of output channel co0 + c*16
    it.y = a.y + (long long)np * a.y_np + slice * 16 + kq * 4;
    it.r = a.r ? a.r + (long long)np * a.r_np + slice * 16 + kq * 4 : nullptr;
    it.co0 = cog * 64 + cohalf * CW * 16 + cl;
    it.full = (cog * 64 + COT * 16) <= a.cout;
    return it;
  };
  f32x4 acc[PW][CW];
  // accumulator initial value of pixel p = residual (or zero)
  auto init_pixel = [&](const Item& it, int p) {
    const int gy = it.y0 + wrow, gx = it.x0 + p;
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it.r && gy < a.H && gx < a.W) {                 // loads go straight into the accumulators (no temporaries)
      const float* rp = it.r + (gy * a.W + gx) * r_px + it.co0 * r_co;
#pragma unroll
      for (int c = 0; c < CW; ++c)
        if (it.full || (it.co0 + c * 16) < a.cout) acc[p][c] = *reinterpret_cast<const f32x4*>(rp + (c * 16) * r_co);
    }
  };
  auto store_pixel = [&](const Item& it, int p) {
    const int gy = it.y0 + wrow, gx = it.x0 + p;
    if (!(gy < a.H && gx < a.W)) return;
    float* yp = it.y + (gy * a.W + gx) * y_px + it.co0 * y_co;
#pragma unroll
    for (int c = 0; c < CW; ++c)
      if (it.full || (it.co0 + c * 16) < a.cout) {
        const float bv = a.bias ? a.bias[it.co0 + c * 16] : 0.f;     // per-channel constant (primal bias), added at store
        *reinterpret_cast<f32x4*>(yp + (c * 16) * y_co) = acc[p][c] + bv;
      }
  };

  // One 8-channel chunk from LDS stage `stage`.  LAST = the item's final chunk: pixel p's accumulators are final
  // after its K-step-2 MFMAs, so its 16-byte stores are issued right there and the NEXT item's residual loads go
  // straight into the freed registers -- the VMEM issue cost of the tail (measured ~335 cycles per store / load
  // instruction, ~9.4k + ~9k cycles per item when done as a block) hides under the remaining MFMAs.
  auto chunk = [&](int stage, auto LAST, const Item& cur, const Item& nxt, bool has_next) {
    const unsigned char* Xh = smem + stage * C::BUF_BYTES;
    const unsigned char* Xl = Xh + C::XS_BYTES;
    const unsigned char* Wh = Xh + 2 * C::XS_BYTES;
    const unsigned char* Wl = Wh + C::WS_BYTES;
    bf16x8 ah[2][CW], al[2][CW], bh[BD], bl[BD];
    auto load_a = [&](int s, int set) {
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        ah[set][c] = *reinterpret_cast<const bf16x8*>(Wh + (((s * COT + c) * 64) << 4) + aoff);
        al[set][c] = *reinterpret_cast<const bf16x8*>(Wl + (((s * COT + c) * 64) << 4) + aoff);
      }
    };
    auto load_b = [&](int t) {
      const int s = t / PW, p = t % PW;
      bh[t % BD] = *reinterpret_cast<const bf16x8*>(Xh + boff[s][p & 1] + p * 256);
      bl[t % BD] = *reinterpret_cast<const bf16x8*>(Xl + boff[s][p & 1] + p * 256);
    };
    load_a(0, 0);
#pragma unroll
    for (int t = 0; t < BD - 1; ++t) load_b(t);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
      const int s = t / PW, p = t % PW;
      if (t + BD - 1 < NSTEP) load_b(t + BD - 1);
      if (p == 0 && s + 1 < 3) load_a(s + 1, (s + 1) & 1);
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        // D[row = Jacobian column][col = output channel] = X-fragment (as A) x W-fragment (as B): each lane then
        // holds 4 CONSECUTIVE columns (rows kq*4 + r) of channel cl -> 16-byte stores / residual loads
        acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t % BD], al[s & 1][c], acc[p][c], 0, 0, 0);
        acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[t % BD], ah[s & 1][c], acc[p][c], 0, 0, 0);
        acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t % BD], ah[s & 1][c], acc[p][c], 0, 0, 0);
      }
      if (LAST && s == 2) {
        store_pixel(cur, p);
        if (has_next) init_pixel(nxt, p);
      }
      // keep this step's reads-then-MFMAs(-then-tail) order: without the fence hipcc's scheduler re-clusters the
      // ds_reads next to their uses (lgkmcnt(0) before most MFMA groups) and the ring no longer hides LDS latency
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int g = 0;                                                       // stream chunk index -> LDS stage g & 1
  Item cur = make_item(0), nxt = cur;
  if (n_items > 0) {
#pragma unroll
    for (int p = 0; p < PW; ++p) init_pixel(cur, p);
  }
  __syncthreads();                                                 // stage 0 ready
  for (int item = 0; item < n_items; ++item) {
    const bool has_next = item + 1 < n_items;
    if (has_next) nxt = make_item(item + 1);
    for (int ch = 0; ch < nchunks - 1; ++ch, ++g) {
      chunk(g & 1, std::false_type{}, cur, nxt, has_next);
      __syncthreads();                                             // stage (g+1)&1 ready, stage g&1 free
    }
    chunk(g & 1, std::true_type{}, cur, nxt, has_next);
    __syncthreads();
    ++g;
    cur = nxt;
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

template <int COT, int PXW, bool SELF>
int launch(const cmf_conv_tangent_args& a, hipStream_t s) {
  using C = BCfg<COT, PXW>;
  const int tiles_x = cmf_ceil_div(a.W, 2 * PXW), tiles = tiles_x * cmf_ceil_div(a.H, 2);
  const int nslices = a.nc / 16, ncog = cmf_ceil_div(a.cout, 64);
  const long long total = (long long)tiles * nslices * ncog * a.np;
  if (total > 0x7fffffffLL) return CMF_ERANGE;
  auto k = conv_tangent_bf16x3_kernel<COT, PXW, SELF>;
  constexpr int lds = 2 * C::BUF_BYTES;
  static int n_cu = 0;                          // idempotent initialisation; a benign race at worst repeats it
  if (n_cu == 0) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    n_cu = cus > 0 ? cus : 256;
  }
  const int grid = (int)(total < n_cu ? total : n_cu);          // persistent: one 112 KiB workgroup per CU
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a, tiles_x, tiles, nslices, ncog, (int)total);
  CMF_LAUNCH_CHECK();
  return 0;
}

template <int PXW>
int launch_cot(const cmf_conv_tangent_args& a, hipStream_t s) {
  if (a.fmode == CMF_F_SELF_RELU) return (a.cout > 32) ? launch<4, PXW, true>(a, s) : launch<2, PXW, true>(a, s);
  return (a.cout > 32) ? launch<4, PXW, false>(a, s) : launch<2, PXW, false>(a, s);   // co tiles per workgroup: 64 / 32 channels
}

inline bool fits_int(long long v) { return v >= 0 && v < (1LL << 29); }   // element offsets; x4 bytes must fit 32 bits

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
  if (a.fmode < CMF_F_NONE || a.fmode > CMF_F_SELF_RELU) return CMF_EINVAL;
  if (a.fmode != CMF_F_NONE && a.fmode != CMF_F_SELF_RELU && !a.f) return CMF_EINVAL;
  if ((a.x_np | a.x_ci | a.x_px) % 4 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return CMF_EINVAL;
  if ((a.y_np | a.y_co | a.y_px) % 4 || ((uintptr_t)a.y % 16)) return CMF_EINVAL;            // 16-byte stores
  if (a.r && ((a.r_np | a.r_co | a.r_px) % 4 || ((uintptr_t)a.r % 16))) return CMF_EINVAL;    // 16-byte residual loads
  const long long HW = (long long)a.H * a.W;
  if (!fits_int((a.cin + 8) * a.x_ci + HW * a.x_px + a.nc) || !fits_int((a.cin + 8) * a.f_ci + HW * a.f_px) ||
      !fits_int((a.cout + 64) * a.y_co + HW * a.y_px + a.nc) || (a.r && !fits_int((a.cout + 64) * a.r_co + HW * a.r_px + a.nc)) ||
      HW > (1 << 24))
    return CMF_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  if (a.W % 14) return CMF_EINVAL;      // only the 7-pixel-per-wave tiling is built (14- and 28-wide images)
  return launch_cot<7>(a, s);
}
