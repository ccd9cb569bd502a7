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
// (tap, 8-channel octet) pair, packed WITHOUT padding: K-steps 0 / 1 of a chunk carry taps 0-3 / 5-8 of its
// octet (lane group kq = tap) and the centre taps of four consecutive octets form one K-step executed in
// every fourth chunk (see the MFMA-wave section below): 9 K-steps per 4 octets.  The kernel is bound by SIMD
// issue in its MFMA waves and by the clock the chip holds under this load, not by HBM: rocprofv3 counts the
// matrix pipes busy 64 % of the kernel's cycles at an effective 1.79 GHz (profiles/r02_pmc_conv_tangent_bf16x3.json).
//
// LDS images (all reads are 16-byte, linear within a lane group => bank-conflict free):
//   X hi/lo : [pixel][16 columns][8 channels] bf16   -> B fragment = one ds_read_b128 per lane
//   W hi/lo : [K-step][co tile][kq][16 co][8 channels] bf16, pre-split and pre-arranged by
//             cmf_pack_weight_bf16x3, so staging is a straight copy.
// A staging thread owns one (pixel, 4 columns) item: it loads the 8 channels of the octet (8 x 16 B),
// applies the activation-derivative factor, splits, and writes 4 + 4 ds_write_b128.
#include <cstdlib>
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
// ... and of EVERY workgroup: s_memrealtime (the chip-wide 100 MHz counter) at entry and at the exit of its MFMA wave 0
// (tools/read_wg_span.py: how evenly the persistent workgroups finish their equal shares)
__device__ unsigned long long cmf_dbg_wg_span[1024][2];
extern "C" int cmf_debug_read_wg_span(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cmf_dbg_wg_span), sizeof(cmf_dbg_wg_span));
}
#define SPAN(k)                                                                                   \
  do {                                                                                            \
    if (lane == 0 && wave == 0 && blockIdx.x < 1024) {                                            \
      unsigned long long t_;                                                                      \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      cmf_dbg_wg_span[blockIdx.x][k] = t_;                                                        \
    }                                                                                             \
  } while (0)
#define STAMP2(w, g, k)                                                                           \
  do {                                                                                            \
    if (blockIdx.x == 0 && lane == 0 && wave == (w)) {                                            \
      unsigned long long t_;                                                                      \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
      cmf_dbg_stamps[2][g][k] = t_;                                                               \
    }                                                                                             \
  } while (0)
#else
#define STAMP(role, g, k) do {} while (0)
#define STAMP2(w, g, k) do {} while (0)
#define SPAN(k) do {} while (0)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int COT, int PXW>
struct BCfg {
  // tile = TH rows x TW pixels: 2 x 14 for 14- / 28-wide images (PXW = 7), 4 x 8 for 16- / 32-wide ones (PXW = 4: 60 halo
  // pixels = 240 staging items, halo overhead 1.9x); an MFMA wave owns RW = TH/2 rows = PW pixels
  static constexpr int TH = PXW == 4 ? 4 : 2, RW = TH / 2, TW = 2 * PXW, PW = RW * TW, TWH = TW + 2, PIXH = (TH + 2) * TWH;
  // LDS pixel index of the wave's pixel p relative to its pixel 0 (rows are TWH apart in the halo image)
  static constexpr int pl(int p) { return (p / TW) * TWH + p % TW; }
  static constexpr int XS_BYTES = PIXH * 16 * 16;                 // one of hi / lo
  static constexpr int WS_BYTES = 3 * COT * 4 * 16 * 16;          // one of hi / lo
  static constexpr int BUF_BYTES = 2 * XS_BYTES + 2 * WS_BYTES;   // one pipeline stage
  // centre ring: the tile's own pixels (no halo) of three octets, hi and lo, for the 4-octet centre-tap K-step
  static constexpr int RING_HALF = TH * TW * 16 * 16;              // one of hi / lo of one octet: TH rows x TW pixels x 256 B
  static constexpr int RING_SLOT = 2 * RING_HALF;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES + 3 * RING_SLOT;
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
// fp16 split of a channel pair (F16 variant): hi = RNE fp16 pair, lo = RNE fp16 of the exact remainders.  The remainder is ONE
// mixed-precision fma per value, e - float(hi) = fma(hi, -1, e) with hi read straight from its half of the packed register
// (v_fma_mix_f32: op_sel_hi marks source 0 as fp16, op_sel picks the half): 4 VALU per pair against 6 for the bf16 form.
__device__ __forceinline__ unsigned split_f16(float e, float o, unsigned& lo) {
  unsigned hb;
  float re, ro;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hb) : "v"(e), "v"(o));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(re) : "v"(hb), "v"(e));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ro) : "v"(hb), "v"(o));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(re), "v"(ro));
  return hb;
}
// 2^e as a float, e clamped to the normal range
__device__ __forceinline__ float pow2i(int e) {
  e = e < -126 ? -126 : e > 127 ? 127 : e;
  return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
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
// F16 (cmf_conv_tangent_f16x3, the primal pass): the same kernel on v_mfma_f32_16x16x32_f16 with fp16 hi / lo halves (11 + 11
// significant bits: ~2^-22 per product instead of 2^-16), exact power-of-two scales around the narrow exponent range (weights
// packed times 2^k, inputs times the power of two derived from *amax_in; the epilogue multiplies by the inverse, a residual is
// scaled when it lands), the sign bits of the stored values written as the next conv's relu' bit mask (mask_out) and max(y) raised
// into *amax_out.  SELF mode only.
// FRES (F16 launches with a residual): the residual is NOT the accumulators' initial value -- every partial sum would be rounded
// at the residual's magnitude, 54 times per output, which cost the primal chain 2-3x the error of the exact-fp32-product kernel
// (measured: median g_ij error of the full-size model 1.3e-6 against 4e-7).  The accumulators start at zero, the item's last chunk
// carries no tail, and after it the wave loads the residual into the registers its fragments occupied and adds it ONCE, after the
// products, like cmf_conv_tangent does.  The round trip is exposed once per item: ~15 % on the 80 launches of a step that have one.
// CKB (a.live = 1 / 2, forward tangent conv with relu' factor): CHECKERBOARD output -- only the pixels with (row + col) % 2 == live - 1 are
// computed: the last hidden conv of a checkerboard coupler's network is read, through the pointwise 1x1 conv, at the (1 - mask)
// pixels alone (acl.py:48-66).  The staging (loader waves, LDS images, centre ring) is the full tile's; an MFMA wave owns the LIVE
// pixels of its rows -- every second one, the phase alternating from row to row: C::PW / 2 accumulator pixels, half the MFMAs and
// B-fragment reads -- loads the residual at those pixels of the FULL image and stores into the COMPACT one (row*(W/2) + col/2).
template <int COT, int PXW, int MODE, bool F16 = false, bool FRES = false, bool CKB = false>
__global__ __launch_bounds__(512, 2) void conv_tangent_bf16x3_kernel(cmf_conv_tangent_args a, int tiles_x, int ntiles,
                                                                       int nslices, int ncog, int total) {
  static_assert(!CKB || (!F16 && !FRES && (MODE == 1 || MODE == 3)), "checkerboard output: the forward tangent conv with a relu' factor");
  static_assert(!F16 || ((MODE == 2 || MODE == 4) && (COT == 4 || COT == 2)), "the fp16 variant is the primal pass: SELF mode (forward) or 4 (backward)");
  // BWD (F16 with MODE 4): the primal BACKWARD's data-gradient convs -- plain input (a cotangent, no relu on load), the packed
  // adjoint operator, and  y = [fo > 0] . conv(x) + r  with fo a float tensor laid out like y (the forward activation whose relu
  // the cotangent passes back through, per column = per sample) and an optional residual: both tiles are loaded in the FRES
  // epilogue; no sign bits are written, *amax_out is raised to max |y|
  constexpr bool BWD = F16 && MODE == 4;
  static_assert(!BWD || FRES, "the backward form uses the epilogue with tile loads");
  // HALF (F16 with COT == 2): an item is 32 output channels of a 64-channel group, `cog` counts HALF groups -- twice the items for
  // launches that would leave most of the chip idle (a 32-sample CIFAR shard: 64 items of 64 channels for 256 CUs).  The weight
  // pack is the 64-channel one: a half reads two of the slab's four channel tiles.
  constexpr bool HALF = F16 && COT == 2;
  constexpr int COG = HALF ? 32 : 64;                            // output channels per item
  static_assert(!FRES || F16, "residual-in-the-epilogue is the fp16 variant's");
  // MODE: 0 = general factor formula, 1 = relu factor (2 instead of 6 VALU per channel in the loader), 2 = SELF (the
  // input's own relu, no factor stream).  Compile-time: every loader VALU instruction delays the MFMA wave it shares
  // a SIMD with.
  constexpr bool SELF = MODE == 2, RELU = MODE == 1, BITS = MODE == 3;   // 3 = relu' from a bit mask (CMF_F_RELU_BITS)
  // 4 = PLAIN: no input factor at all (no factor stream, like SELF, and no relu) and an optional OUTPUT-side relu' bit mask
  // applied at the store -- the reverse (cotangent) sweep: the adjoint of "mask, then conv" is "transposed conv, then mask"
  constexpr bool PLAIN = !F16 && (MODE == 4 || MODE == 5);
  constexpr bool INPLACE = MODE == 5;                          // PLAIN with the output tensor as the (unmasked) residual: below
  constexpr bool NOF = SELF || PLAIN || BWD;                     // no factor stream
  constexpr int NF = NOF ? 0 : BITS ? 1 : 8;                     // factor loads per loader thread and chunk
  using C = BCfg<COT, PXW>;
  // PW = accumulator pixels of an MFMA wave; tp(p) = the p-th one's pixel inside the wave's C::PW tile pixels (CKB: the even ones;
  // the row's phase o is added through per-lane / scalar offsets below), trow(p) = its row within the wave's rows
  constexpr int CW = COT / 2, PW = CKB ? C::PW / 2 : C::PW;
  auto tp = [](int p) constexpr { return CKB ? 2 * p : p; };
  auto trow = [&](int p) constexpr { return tp(p) / C::TW; };
  static_assert(COT % 2 == 0, "the co-split wave layout needs an even number of output-channel tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 stages x [Xh | Xl | Wh | Wl]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  SPAN(0);
  STAMP2(0, 8, 0);                                                 // diagnostic build: kernel entry / exit, prologue, epilogue
  STAMP2(4, 9, 0);                                                 // (tools/read_stamps_f16.py)
  const int kq = lane >> 4, cl = lane & 15;
  const int nchunks = a.cin / 8;
  // F16: exact power-of-two scales (wave-uniform).  xscale puts the largest input in [2^13, 2^14); the packed weights carry
  // 2^k (trailer of cmf_pack_weight_f16x3); accumulators hold 2^k xscale times the result.
  float xscale = 1.f, oscale = 1.f, rscale = 1.f;
  if constexpr (F16) {
    const float amax = a.amax_in ? *a.amax_in : 0.f;
    const int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);     // amax = m 2^(ex - 126), m in [0.5, 1)
    int xs = ex == 0 ? 0 : 14 - (ex - 126);
    xs = xs < -60 ? -60 : xs > 40 ? 40 : xs;
    const float* trailer = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.w) +
                                                          (long long)(HALF ? ncog / 2 : ncog) * nchunks * BCfg<COT, PXW>::W_CHUNK_BYTES);
    const float wscale = trailer[0], winv = trailer[1];
    xscale = pow2i(xs);
    oscale = winv * pow2i(-xs);
    rscale = wscale * xscale;
  }

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
#ifdef CMF_DBG_REPEAT
  // timing only: every workgroup walks its item list CMF_DBG_REPEAT times in ONE launch (same arguments, same outputs) -- what R
  // launches cost without R - 1 of their fills and drains: the ceiling of one persistent launch per coupler network (profiles/LABBOOK.md, round 4: section 9 item 0a)
  const int n_real = jx < xlen ? (xlen - jx + nbx - 1) / nbx : 0;
  const int n_items = n_real * CMF_DBG_REPEAT;
#else
  const int n_items = jx < xlen ? (xlen - jx + nbx - 1) / nbx : 0;
#endif
  const int total_chunks = n_items * nchunks;
  auto decode = [&](int item, int& tile, int& slice, int& cog, int& np) {
#ifdef CMF_DBG_REPEAT
    item = n_real > 0 ? item % n_real : 0;
#endif
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
    // No s_setprio here: with the loaders off the critical path (scalar-VALU split, hand-counted waits) raising their
    // priority only delays the MFMA wave of the same SIMD -- measured 1.5-3 % slower (CMF_DBG_PRIO re-enables it).
#ifdef CMF_DBG_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const int lt = tid - 256;                                    // 0..255: staging item = (pixel, column quad)
    const int x_ci = (int)a.x_ci, x_px = (int)a.x_px, f_ci = (int)a.f_ci, f_px = (int)a.f_px;
    const int q = lt & 3, pix = lt >> 2;
    const int hy = pix / C::TWH, hx = pix % C::TWH;
    // NONE: 1   RELU: [f>0]   TANH: 1 - f^2   RAW: f      as  okf * (c0 + c1*[f>0] + c2*f + c3*f*f)
    const float fc0 = (a.fmode == CMF_F_NONE || a.fmode == CMF_F_TANH || a.fmode == CMF_F_SELF_RELU) ? 1.f : 0.f;
    const float fc1 = (a.fmode == CMF_F_RELU) ? 1.f : 0.f;
    const float fc2 = (a.fmode == CMF_F_RAW) ? 1.f : 0.f;
    const float fc3 = (a.fmode == CMF_F_TANH) ? -1.f : 0.f;
    constexpr bool has_f = !NOF;                                 // compile-time: the wait counts below depend on it
                                                                 // (the launcher rejects fmode NONE without SELF)
    const int fgrp = a.f_group > 1 ? a.f_group : 1;

    // prefetch cursor: which (work item, chunk) the next prefetch fetches, with that item's addressing state.
    // Loads are raw buffer loads: descriptor base = the item's wave-uniform base, voffset = the per-lane byte offset
    // (fixed for the item), soffset = the chunk / channel term (SALU): no VALU address arithmetic per load.  Validity is
    // applied arithmetically at commit time (every load is unconditional and always consumed).
    //
    // The loads are INLINE ASM with hand-counted s_waitcnt: hipcc's own vmcnt bookkeeping collapses on register sets
    // that stay in flight across loop iterations -- it emitted vmcnt(0..12) in front of every commit (and in front
    // of address temporaries it had allocated inside in-flight destination registers), i.e. each iteration drained
    // the prefetch it had just issued and the chunk period contained a full memory latency.  Issue order per
    // iteration g is fixed:   [X,f of chunk g+3]  commit(g+1)  [W slab of chunk g+2]
    // so when commit(g+1) starts the loads issued after X,f(g+1) are W(g), X,f(g+2), W(g+1), X,f(g+3), and after
    // W(g+1) only X,f(g+3).  To keep those counts constant the stream is padded: past the last chunk the cursors stay
    // on the last chunk (re-fetching it, ~3 chunks per workgroup and launch) instead of skipping loads.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto make_rsrc = [&](const void* p) {
      const unsigned long long u = reinterpret_cast<unsigned long long>(p);
      i32x4 d;
      d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
      d[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(u >> 32) & 0xffffu));      // stride 0
      d[2] = -1;                                                                          // no range check
      d[3] = 0x00020000;
      return d;
    };
    int cur_item = 0, cur_ch = 0;
    i32x4 xrs = make_rsrc(a.x), frs = xrs, wrs = make_rsrc(a.w);
    int xo = 0, fo = 0;
    float okf = 0.f;
    auto set_item = [&](int item) {
      int tile, slice, cog, np;
      decode(item, tile, slice, cog, np);
      const int y0 = C::TH * (tile / tiles_x), x0 = C::TW * (tile % tiles_x);
      xrs = make_rsrc(a.x + (long long)np * a.x_np + (long long)slice * (a.x_sl ? a.x_sl : 16));
      frs = BITS ? make_rsrc(reinterpret_cast<const unsigned char*>(a.f) + (long long)np * a.f_np)        // f_np in bytes
                 : make_rsrc(a.f ? a.f + (long long)(np / fgrp) * a.f_np + (np % fgrp) : a.x);
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = lt < C::NX_ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int gpix = gy * a.W + gx;
      okf = ok ? 1.f : 0.f;
      xo = ok ? 4 * (gpix * x_px + q * 4) : 0;
      fo = ok ? (BITS ? gpix * (a.cin / 8) : 4 * (gpix * f_px)) : 0;
    };
    // per-thread byte offsets of its W items inside a chunk slab (loop-invariant).  COT == 4 consumes the slab whole
    // (a linear copy): ONE register, the item stride goes into soffset.
    constexpr int NWOFF = COT == 4 ? 1 : C::NWIT;
    int woff[NWOFF];
    if (COT == 4) {
      woff[0] = lt << 4;
    } else {
#pragma unroll
      for (int it = 0; it < NWOFF; ++it) {                         // the global slab always has 4 co tiles per K-step
        int i = lt + 256 * it;
        i = i < C::NW_ITEMS ? i : C::NW_ITEMS - 1;
        const int rest = i & 63, t = i >> 6;
        const int cot = t % COT, s = (t / COT) % 3, hl = t / (3 * COT);
        woff[it] = (((hl * 3 + s) * 4 + cot) * 64 + rest) << 4;
      }
    }

    struct Regs {
      f32x4 x[8];
      float f[8];
      float okf;
    };
    Regs r0, r1, r2;                                               // chunk g of the stream lives in set g % 3
    // The weight slab has ONE register set: it comes from L2 (the same 192 KiB for every CU), so it is fetched one chunk
    // ahead, right after the previous slab was written to LDS (three sets cost 48 more VGPRs and spilled).
    u32x4 wreg[C::NWIT];
    int wcur_item = 0, wcur_ch = 0;
    [[maybe_unused]] int whalf = 0;                                // HALF: byte offset of the item's two channel tiles inside a K-step
    auto wset_item = [&](int item) {
      int tile, slice, cog, np;
      decode(item, tile, slice, cog, np);
      wrs = make_rsrc(reinterpret_cast<const unsigned char*>(a.w) + (long long)(HALF ? cog >> 1 : cog) * nchunks * C::W_CHUNK_BYTES);
      if constexpr (HALF) whalf = __builtin_amdgcn_readfirstlane((cog & 1) * 2 * 64 * 16);
    };
    // `s_nop 4` opens every load statement: descriptor / soffset SGPRs may have just been written by SALU code.
    auto prefetch_w = [&]() {
      const int wco = wcur_ch * C::W_CHUNK_BYTES;
      // K-step 2 of a slab (the four-octet centre K-step) is consumed in every FOURTH chunk only: in the other chunks its two
      // 4-KiB items (hi, lo) are fetched through a zero-record descriptor -- same instruction count for the hand-counted waits,
      // no L2 / fabric traffic -- and not written to LDS (commit below): 1/3 of the weight bytes in 3 of 4 chunks (round 3: +0.6 ..
      // 1.3 % on the 64 -> 64 launches, results bit-identical)
      i32x4 wrs2 = wrs;
      wrs2[2] = __builtin_amdgcn_readfirstlane((COT != 4 || (wcur_ch & 3) == 3) ? -1 : 0);
#pragma unroll
      for (int it = 0; it < C::NWIT; ++it) {
        const int vo = woff[COT == 4 ? 0 : it % NWOFF];
        const int so = COT == 4 ? wco + it * 4096 : HALF ? wco + whalf : wco;
        if (COT == 4 && it % 3 == 2) {
          asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(wreg[it]) : "v"(vo), "s"(wrs2), "s"(so) : "memory");
          continue;
        }
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(wreg[it]) : "v"(vo), "s"(wrs), "s"(so) : "memory");
      }
      if (++wcur_ch == nchunks) {
        wcur_ch = 0;
        if (wcur_item + 1 < n_items) wset_item(++wcur_item);
        else wcur_ch = nchunks - 1;                                // stream padding: stay on the last slab
      }
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) r0.f[j] = r1.f[j] = r2.f[j] = 0.f;  // stays 0 when fmode == NONE (never loaded)

    auto prefetch = [&](Regs& r) {
      const int ch = cur_ch;
      {
        const int s0 = 4 * (ch * 8 + 0) * x_ci, s1 = 4 * (ch * 8 + 1) * x_ci, s2 = 4 * (ch * 8 + 2) * x_ci,
                  s3 = 4 * (ch * 8 + 3) * x_ci, s4 = 4 * (ch * 8 + 4) * x_ci, s5 = 4 * (ch * 8 + 5) * x_ci,
                  s6 = 4 * (ch * 8 + 6) * x_ci, s7 = 4 * (ch * 8 + 7) * x_ci;
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %8, %9, %10 offen\n\t"
            "buffer_load_dwordx4 %1, %8, %9, %11 offen\n\t"
            "buffer_load_dwordx4 %2, %8, %9, %12 offen\n\t"
            "buffer_load_dwordx4 %3, %8, %9, %13 offen\n\t"
            "buffer_load_dwordx4 %4, %8, %9, %14 offen\n\t"
            "buffer_load_dwordx4 %5, %8, %9, %15 offen\n\t"
            "buffer_load_dwordx4 %6, %8, %9, %16 offen\n\t"
            "buffer_load_dwordx4 %7, %8, %9, %17 offen"
            : "=&v"(r.x[0]), "=&v"(r.x[1]), "=&v"(r.x[2]), "=&v"(r.x[3]), "=&v"(r.x[4]), "=&v"(r.x[5]), "=&v"(r.x[6]),
              "=&v"(r.x[7])
            : "v"(xo), "s"(xrs), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7)
            : "memory");
      }
      if constexpr (BITS) {                                        // one byte: the chunk's 8 relu' bits of this pixel
        asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %1, %2, %3 offen" : "=&v"(r.f[0]) : "v"(fo), "s"(frs), "s"(ch) : "memory");
      } else if (has_f) {
        const int s0 = 4 * (ch * 8 + 0) * f_ci, s1 = 4 * (ch * 8 + 1) * f_ci, s2 = 4 * (ch * 8 + 2) * f_ci,
                  s3 = 4 * (ch * 8 + 3) * f_ci, s4 = 4 * (ch * 8 + 4) * f_ci, s5 = 4 * (ch * 8 + 5) * f_ci,
                  s6 = 4 * (ch * 8 + 6) * f_ci, s7 = 4 * (ch * 8 + 7) * f_ci;
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dword %0, %8, %9, %10 offen\n\t"
            "buffer_load_dword %1, %8, %9, %11 offen\n\t"
            "buffer_load_dword %2, %8, %9, %12 offen\n\t"
            "buffer_load_dword %3, %8, %9, %13 offen\n\t"
            "buffer_load_dword %4, %8, %9, %14 offen\n\t"
            "buffer_load_dword %5, %8, %9, %15 offen\n\t"
            "buffer_load_dword %6, %8, %9, %16 offen\n\t"
            "buffer_load_dword %7, %8, %9, %17 offen"
            : "=&v"(r.f[0]), "=&v"(r.f[1]), "=&v"(r.f[2]), "=&v"(r.f[3]), "=&v"(r.f[4]), "=&v"(r.f[5]), "=&v"(r.f[6]),
              "=&v"(r.f[7])
            : "v"(fo), "s"(frs), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7)
            : "memory");
      }
      r.okf = okf;
      if (++cur_ch == nchunks) {                                   // advance the cursor (wave-uniform)
        cur_ch = 0;
        if (cur_item + 1 < n_items) set_item(++cur_item);
        else cur_ch = nchunks - 1;                                 // stream padding: stay on the last chunk
      }
    };
    // hand-counted waits; the "+v" operands make every consumer of the set depend on the wait statement
    constexpr int NW = C::NWIT;
    auto wait_x = [&](Regs& r) {                                   // X,f of `r` landed; W, X,f, W, X,f may still fly
      if constexpr (BITS)
        asm volatile("s_waitcnt vmcnt(%9)"
                     : "+v"(r.x[0]), "+v"(r.x[1]), "+v"(r.x[2]), "+v"(r.x[3]), "+v"(r.x[4]), "+v"(r.x[5]), "+v"(r.x[6]),
                       "+v"(r.x[7]), "+v"(r.f[0])
                     : "n"(2 * NW + 2 * (8 + NF)));
      else if constexpr (has_f)
        asm volatile("s_waitcnt vmcnt(%16)"
                     : "+v"(r.x[0]), "+v"(r.x[1]), "+v"(r.x[2]), "+v"(r.x[3]), "+v"(r.x[4]), "+v"(r.x[5]), "+v"(r.x[6]),
                       "+v"(r.x[7]), "+v"(r.f[0]), "+v"(r.f[1]), "+v"(r.f[2]), "+v"(r.f[3]), "+v"(r.f[4]), "+v"(r.f[5]),
                       "+v"(r.f[6]), "+v"(r.f[7])
                     : "n"(2 * NW + 32));
      else
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(r.x[0]), "+v"(r.x[1]), "+v"(r.x[2]), "+v"(r.x[3]), "+v"(r.x[4]), "+v"(r.x[5]), "+v"(r.x[6]),
                       "+v"(r.x[7])
                     : "n"(2 * NW + 16));
    };
    auto wait_w = [&]() {                                          // the slab landed; the newest X,f may still fly
      static_assert(C::NWIT == 6 || C::NWIT == 3, "operand list below");
      constexpr int N = 8 + NF;
      if constexpr (C::NWIT == 6)
        asm volatile("s_waitcnt vmcnt(%6)"
                     : "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]), "+v"(wreg[3]), "+v"(wreg[4]), "+v"(wreg[5])
                     : "n"(N));
      else
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]) : "n"(N));
    };

    int com_ch = 0;                                                // channel chunk (within its item) of the next commit
    auto commit = [&](Regs& r, int stage) {
#ifdef CMF_DBG_NOCOMMIT
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(r.x[j]), "v"(r.f[j]));
#pragma unroll
      for (int it = 0; it < C::NWIT; ++it) asm volatile("" ::"v"(wreg[it]));
      return;
#endif
      unsigned char* Xh = smem + stage * C::BUF_BYTES;
      unsigned char* Xl = Xh + C::XS_BYTES;
      unsigned char* Wh = Xh + 2 * C::XS_BYTES;
      unsigned char* Wl = Wh + C::WS_BYTES;
      if (lt < C::NX_ITEMS) {
        // Values are paired ALONG THE COLUMNS (the two halves of a loaded float4 are already register pairs), so the
        // scale and the remainder are v_pk_* ops without the v_mov pairs a channel pairing needed; v_cvt_pk_bf16_f32
        // takes any two registers, so the (channel 2jj, 2jj+1) packing is free.  ~3 VALU per element.
        // SCALAR f32 multiplies / subtractions on purpose (and -fno-slp-vectorize for this file): beside a saturating
        // MFMA wave a v_pk_mul/fma_f32 costs ~20 cycles more than the two scalar ops it replaces.
#ifdef CMF_DBG_RAWCOMMIT
        // timing only: no mask, no split -- truncated bf16 pairs (one v_perm per pair) into both planes
        asm volatile("" ::"v"(r.f[0]));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          u32x4 h;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            h[jj] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r.x[2 * jj + 1][c]),
                                          __builtin_bit_cast(unsigned, r.x[2 * jj][c]), 0x07060302u);
          const int off = ((pix * 16 + xslot(q * 4 + c, pix)) << 4);
          *reinterpret_cast<u32x4*>(Xh + off) = h;
          *reinterpret_cast<u32x4*>(Xl + off) = h;
        }
#else
        float v[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = r.f[j];
          const float m = NOF ? (F16 ? r.okf * xscale : r.okf) : BITS ? (((__builtin_bit_cast(unsigned, r.f[0]) >> j) & 1u) ? r.okf : 0.f)
                               : RELU ? (f > 0.f ? r.okf : 0.f)
                                              : r.okf * (fc0 + fc1 * (f > 0.f ? 1.f : 0.f) + f * (fc2 + fc3 * f));
#pragma unroll
          for (int c = 0; c < 4; ++c) v[j][c] = (SELF ? fmaxf(r.x[j][c], 0.f) : r.x[j][c]) * m;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                              // column q*4 + c: 8 channels -> 16 B hi + 16 B lo
          u32x4 h, l;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float e = v[2 * jj][c], o = v[2 * jj + 1][c];    // even / odd channel of the pair
            if constexpr (F16) {
              unsigned lo_;
              h[jj] = split_f16(e, o, lo_);
              l[jj] = lo_;
            } else {
              const unsigned hb = pack_lo(e, o);                   // v_cvt_pk_bf16_f32 (RNE)
              const float re = e - __builtin_bit_cast(float, hb << 16);
              const float ro = o - __builtin_bit_cast(float, hb & 0xffff0000u);
              h[jj] = hb;
              l[jj] = pack_lo(re, ro);
            }
          }
          const int off = ((pix * 16 + xslot(q * 4 + c, pix)) << 4);
          *reinterpret_cast<u32x4*>(Xh + off) = h;
          *reinterpret_cast<u32x4*>(Xl + off) = l;
          if (c == 1) __builtin_amdgcn_sched_barrier(0);           // bounds the live temporaries
        }
#endif
      }
      wait_w();
      const bool wquad = COT != 4 || (com_ch & 3) == 3;           // the chunk being committed runs the centre K-step
      com_ch = com_ch + 1 == nchunks ? 0 : com_ch + 1;
#pragma unroll
      for (int it = 0; it < C::NWIT; ++it) {
        const int i = lt + 256 * it;
        if (COT == 4 && it % 3 == 2 && !wquad) continue;          // wave-uniform
        if (i < C::NW_ITEMS) {
          const int rest = i & 63, t = i >> 6;
          const int cot = t % COT, s = (t / COT) % 3, hl = t / (3 * COT);
          unsigned char* dst = (hl ? Wl : Wh) + ((((s * COT + cot) * 64) + rest) << 4);
          *reinterpret_cast<u32x4*>(dst) = wreg[it];
        }
      }
    };

    if (n_items > 0) {
      set_item(0);
      wset_item(0);
      prefetch(r0);                                                // chunks 0, 1, 2 (padded past the end of the stream)
      prefetch(r1);
      prefetch(r2);
      prefetch_w();                                                // slab 0
      STAMP2(4, 10, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP2(4, 10, 1);
      wait_x(r0);
      commit(r0, 0);
      prefetch_w();                                                // slab 1
      STAMP2(4, 10, 2);
    }
    __syncthreads();                                               // stage 0 ready
    // iteration g: the MFMA waves consume stage g&1 (stream chunk g); set g%3 is free -> stream chunk g+3;
    // stage (g+1)&1 <- chunk g+1 from set (g+1)%3.  Exactly one barrier per iteration, matching the MFMA waves.
    // Every iteration issues the same loads in the same order (see the wait counts above).
    auto iter = [&](int g, Regs& freed, Regs& next) {
      STAMP(1, g, 0);
      prefetch(freed);
      STAMP(1, g, 1);
      wait_x(next);
      STAMP(1, g, 2);
      commit(next, (g + 1) & 1);
      prefetch_w();                                                // slab of chunk g + 2 into the registers just written out
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // padding loads still in flight
    STAMP2(4, 9, 1);
    return;
  }

  // ================================= MFMA waves =================================
#ifdef CMF_DBG_MFMAPRIO
  __builtin_amdgcn_s_setprio(3);
#endif
  const int wrow = C::RW * (wave >> 1), cohalf = wave & 1;        // first tile row of this wave
  const int y_co = (int)a.y_co, y_px = (int)a.y_px, r_co = (int)a.r_co, r_px = (int)a.r_px;

  // Per-lane B offsets: lane group kq of K-step s reads tap 4*s + kq (taps >= 9 read tap 8's data against zero
  // weights) of pixel (wrow + dy, p + dx); the slot permutation depends on the pixel's parity, i.e. on (dx + p) & 1
  // (the halo row length TWH is even).
  // K mapping WITHOUT padding: K-step 0 = taps 0..3, K-step 1 = taps 5..8 of the chunk's octet (lane group kq = tap);
  // the centre taps (tap 4) of FOUR consecutive octets form one K-step (lane group kq = octet 4m + kq), executed in
  // every fourth chunk: 9 K-steps per 4 octets instead of 12.  The centre pixels of octets 4m .. 4m+2 wait in a small LDS
  // ring (copied from the stage by the MFMA waves while they consume it), octet 4m+3's are read from the live stage.
  int boff[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tap = s == 0 ? kq : 5 + kq;
    const int pbase = (wrow + tap / 3) * C::TWH + tap % 3;         // staged pixel of p = 0
#pragma unroll
    for (int par = 0; par < 2; ++par) boff[s][par] = ((pbase * 16 + xslot(cl, pbase + par)) << 4);
  }
  // centre K-step: per-lane source (ring slot kq for kq < 3, the live stage for kq == 3), relative to smem.  Ring rows
  // are verbatim copies of the stage's pixel rows, so the slot permutation is the stage's (parity of the halo pixel).
  const int cpix = (wrow + 1) * C::TWH + 1;                        // halo index of this wave's pixel p = 0
  const int crow = kq < 3 ? 0 : (C::TWH - C::TW) * 256;            // extra bytes per tile row for the lanes that read the stage
  int coff_ring[2], coff_stage[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int sl = xslot(cl, cpix + par) << 4;
    coff_ring[par] = 2 * C::BUF_BYTES + (kq < 3 ? kq : 0) * C::RING_SLOT + wrow * C::TW * 256 + sl;
    coff_stage[par] = cpix * 256 + sl;
  }
  // CKB: the live pixels of the wave's row k are tile pixels 2 j + o_k, o_k = (live - 1 + wrow + k) & 1 (tile origins are even in both
  // directions).  The phase only moves the pixel by one slot of 256 B and flips the slot permutation's parity: folded into the
  // per-lane offsets, indexed by the row's parity k & 1 instead of the pixel's
  [[maybe_unused]] const int ck_o[2] = {(a.live - 1 + wrow) & 1, (a.live + wrow) & 1};
  if constexpr (CKB) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int b0 = boff[s][0], b1 = boff[s][1];
      boff[s][0] = ck_o[0] ? b1 + 256 : b0;
      boff[s][1] = ck_o[1] ? b1 + 256 : b0;
    }
    const int r0 = coff_ring[0], r1 = coff_ring[1], s0 = coff_stage[0], s1 = coff_stage[1];
    coff_ring[0] = ck_o[0] ? r1 + 256 : r0;
    coff_ring[1] = ck_o[1] ? r1 + 256 : r0;
    coff_stage[0] = ck_o[0] ? s1 + 256 : s0;
    coff_stage[1] = ck_o[1] ? s1 + 256 : s0;
  }
  const int aoff = (lane << 4) + ((cohalf * CW * 64) << 4);

  // Fragment pipeline: step t = s*PW + p consumes B fragment pair t; pairs are fetched BD-1 steps ahead into a ring
  // of BD register pairs, the A fragments of K-step s+1 are fetched at the start of K-step s into the alternate
  // set.  Everything is unrolled, so ring slots are static registers and hipcc emits counted lgkmcnt waits.
#ifdef CMF_DBG_BD
  constexpr int BD = CMF_DBG_BD;
#else
  constexpr int BD = PW > 14 ? 3 : 4;                           // 16-pixel waves (4 x 8 tiles): 128 accumulator VGPRs, shallower ring
#endif

  // Item context.  The launcher guarantees whole tiles (H even, W % TW == 0) and whole channel groups, so the tail has
  // NO per-pixel / per-channel validity tests, and every access is a raw buffer op: descriptor base = the item's
  // wave-uniform base, voffset = the lane's fixed byte offset, soffset = pixel / channel-tile term (SALU).  A missing
  // residual (or "no next item") is a descriptor with zero records: its loads return 0 without touching memory, so
  // the tail is branch-free.  (Before: ~70 instructions and 5-6 branches per pixel, 64-bit address arithmetic,
  // v_mul_lo -- the tail cost ~18k cycles per item, as much as three chunks.)
  typedef unsigned bu32x4 __attribute__((vector_size(16)));        // the type the raw buffer builtins use
  constexpr int RS_FLAGS = 0x00020000;
  const int yvoff = 4 * ((cohalf * CW * 16 + cl) * y_co + kq * 4); // lane (kq, cl): columns kq*4..+3 of channel cl (+16c)
  const int rvoff = 4 * ((cohalf * CW * 16 + cl) * r_co + kq * 4);
  struct Item {
    int ypix, rpix;                                                // byte offset of pixel p = 0 of this wave's tile row
    int pix0, np, cog;                                             // (PLAIN) pixel index of p = 0, sample, channel group: output mask
  };
  auto item_geom = [&](int item, int& np, int& slice, int& cog, Item& it) {
    int tile;
    decode(item, tile, slice, cog, np);
    const int pix0 = (C::TH * (tile / tiles_x) + wrow) * a.W + C::TW * (tile % tiles_x);
    it.ypix = CKB ? 4 * ((C::TH * (tile / tiles_x) + wrow) * (a.W / 2) + C::TW / 2 * (tile % tiles_x)) * y_px    // compact image
                  : 4 * pix0 * y_px;
    it.rpix = 4 * pix0 * r_px;
    it.pix0 = pix0, it.np = np, it.cog = cog;
  };
  // descriptors are built from readfirstlane'd words: hipcc otherwise keeps loop-carried descriptors in VGPRs and
  // wraps every access in a waterfall loop
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto y_rsrc = [&](int np, int slice, int cog) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(a.y + (long long)np * a.y_np + (long long)slice * (a.y_sl ? a.y_sl : 16) + (long long)cog * COG * y_co);
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    // num_records 0x7fffff00: every real offset (a lane's channel / column term, far below 2^31) passes the range check, the
    // offset Y_DROP of a masked-off lane of an in-place store does not -- its store is discarded, still counted by vmcnt
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffff00, RS_FLAGS);
  };
  constexpr int Y_DROP = 0x7ffffff0;
  // F16: relu' bit mask of the stored values (CMF_F_RELU_BITS layout of the next conv): per pixel ONE dword per sample = the 32
  // channel bits of this wave's channel half (HALF: one 16-bit word = its 16 channels), stored by lane s < 16 for sample
  // slice*16 + s; the other lanes (and every lane when mask_out is NULL: zero records) store past the descriptor's range --
  // dropped, still counted by vmcnt
  [[maybe_unused]] auto m_rsrc = [&](int np, int slice, int cog) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(a.mask_out ? a.mask_out : (void*)a.y) +
                                 (unsigned long long)((long long)np * a.nc + slice * 16) * a.mask_np + (unsigned)(cog * (COG / 8) + cohalf * (COG / 16));
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0,
                                             a.mask_out ? 0x7fffff00 : 0, RS_FLAGS);
  };
  const int mvoff = lane < 16 ? lane * (int)a.mask_np : Y_DROP;
  float ymax = 0.f;                                                // F16: running max of the stored values (-> *amax_out)
  auto r_rsrc = [&](int np, int slice, int cog, bool on) {         // words, for the inline-asm loads
    const float* base = a.r ? a.r + (long long)np * a.r_np + (long long)slice * (a.r_sl ? a.r_sl : 16) + (long long)cog * COG * r_co : a.y;
    const unsigned long long u = reinterpret_cast<unsigned long long>(base);
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    d[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(u >> 32) & 0xffffu));
    d[2] = __builtin_amdgcn_readfirstlane((a.r && on) ? -1 : 0);   // zero records: loads return 0, nothing is fetched
    d[3] = RS_FLAGS;
    return d;
  };
  f32x4 acc[PW][CW];
  Item cur, nxt;
  int np_, slice_, cog_;
  item_geom(0, np_, slice_, cog_, cur);
  nxt.ypix = cur.ypix, nxt.rpix = cur.rpix, nxt.pix0 = cur.pix0, nxt.np = cur.np, nxt.cog = cur.cog;   // (field by field: a struct copy
                                                                                                        // left a dead 20-byte stack slot)
  // Per-channel constants (primal bias): a bias implies a single channel group (launcher precondition), so they are
  // fetched ONCE per launch -- by inline asm like the residual, so that hipcc never places a vmcnt wait of its own
  // among the hand-counted ones; a NULL bias is a zero-record descriptor (reads 0).  (A per-store
  // `a.bias ? a.bias[..] : 0` once put a dependent load + s_waitcnt vmcnt(0) in front of EVERY store: ~28k cycles/item.)
  float bias[CW];
  [[maybe_unused]] float bias_h1 = 0.f;                            // HALF: the constant of this lane's channel in the odd halves
  {
    const unsigned long long u = reinterpret_cast<unsigned long long>(a.bias ? a.bias : a.y);
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    d[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(u >> 32) & 0xffffu));
    d[2] = __builtin_amdgcn_readfirstlane(a.bias ? -1 : 0);
    d[3] = RS_FLAGS;
    const int bvo = 4 * (cohalf * CW * 16 + cl);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int so = 4 * c * 16;
      asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(bias[c]) : "v"(bvo), "s"(d), "s"(so) : "memory");
    }
    if constexpr (HALF) {
      const int so = 4 * 32;
      asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(bias_h1) : "v"(bvo), "s"(d), "s"(so) : "memory");
    }
  }
  auto cur_yrs = y_rsrc(np_, slice_, cog_);
  auto cur_mrs = cur_yrs;
  if constexpr (F16) cur_mrs = m_rsrc(np_, slice_, cog_);
  i32x4 nxt_rrs = r_rsrc(np_, slice_, cog_, n_items > 0);
  // Accumulator initial value of pixel p = residual, loaded straight into the accumulators by INLINE ASM with
  // hand-counted waits (wait_res below): hipcc's vmcnt for these loads also counted the interleaved stores'
  // completion, so chunk 0 of every item stalled on HBM round trips.
  auto init_pixel = [&](const Item& it, const i32x4& rrs_in, int p) __attribute__((always_inline)) {
    // hipcc moves a loop-carried descriptor into VGPRs as soon as a lane-dependent select sits near it (the stamp code, the
    // in-place store's offset select) and does not bring an "s"-constrained vector operand back by itself: readfirstlane is a
    // plain copy when the words already are scalars
    i32x4 rrs;
#pragma unroll
    for (int i = 0; i < 4; ++i) rrs[i] = __builtin_amdgcn_readfirstlane(rrs_in[i]);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int so = it.rpix + 4 * ((trow(p) * a.W + tp(p) % C::TW + (CKB ? ck_o[trow(p) & 1] : 0)) * r_px + c * 16 * r_co);
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(acc[p][c]) : "v"(rvoff), "s"(rrs), "s"(so) : "memory");
    }
  };
  // PLAIN: relu' bits of this wave's 32 output channels, one dword per pixel (CMF_F_RELU_BITS layout: cout / 8 bytes per
  // pixel), fetched with SCALAR loads (lgkmcnt, not the hand-counted vmcnt) at the top of the item's last chunk.  One asm
  // statement per load WITH its wait: an s_load writes its SGPR asynchronously, and a destination the compiler believed
  // defined could be spilled, moved or reused before the data arrived.  The address is always valid and the fetch is
  // unconditional (the launcher points a missing mask at the input tensor and clears fomode): no divergent paths here.
  unsigned omask[PW];
  const bool use_omask = PLAIN && a.fomode == CMF_F_RELU_BITS;
  // IN-PLACE skip connection of the reverse sweep,  y <- y + mask . conv(x):  the residual IS the output tensor (launcher: same
  // pointer and strides).  The accumulators start from it as usual, and at the store a lane whose (pixel, channel) is masked off
  // keeps what is in memory -- its store goes past the descriptor's range -- while the others write residual + product.  (The
  // residual as a separate tensor would have to be re-read at the store: the accumulators hold residual + product by then.)
  constexpr bool inplace = INPLACE;                            // launcher: a.r == a.y with y's strides, bit mask given
  // (compile-time: as a run-time flag the 4 x 8-tile PLAIN variant spilled 15 VGPRs; 252 / 244 VGPRs this way)
  auto load_omask = [&](const Item& it) __attribute__((always_inline)) {
    if constexpr (PLAIN) {
      const unsigned long long base = reinterpret_cast<unsigned long long>(a.fo) + (unsigned long long)it.np * a.fo_np +
                                      (unsigned)(it.cog * 8 + cohalf * 4);
      const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
      const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
      const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
      const int stride = a.cout / 8;
#pragma unroll
      for (int p = 0; p < PW; ++p) {
        const int off = __builtin_amdgcn_readfirstlane((it.pix0 + (p / C::TW) * a.W + p % C::TW) * stride);
        unsigned w;
        asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(sb), "s"(off) : "memory");
        omask[p] = use_omask ? w : ~0u;
      }
    }
  };
  auto store_pixel = [&](const Item& it, int p, f32x4 radd0, f32x4 radd1, f32x4 m0 = f32x4{}, f32x4 m1 = f32x4{}) __attribute__((always_inline)) {
    [[maybe_unused]] unsigned mbits = 0;
    [[maybe_unused]] f32x4 vst[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      f32x4 v = acc[p][c];
      int vo = yvoff;
      if constexpr (PLAIN) {
        const bool on = (omask[p] >> (c * 16 + cl)) & 1u;
        if constexpr (inplace) vo = on ? yvoff : Y_DROP;
        else v = on ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (BWD) {                                         // undo the scales, then the per-sample relu' of the forward activation
        v = v * oscale;
        const f32x4 mm = c == 0 ? m0 : m1;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = mm[r] > 0.f ? v[r] : 0.f;
      } else if constexpr (HALF) v = v * oscale + ((it.cog & 1) ? bias_h1 : bias[c]);
      else if constexpr (F16) v = v * oscale + bias[c];            // undo the operand scales (exact), then the bias
      else v += bias[c];                                           // per-channel constant (primal bias)
      if constexpr (FRES) v += c == 0 ? radd0 : radd1;             // the residual, once, after the products
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, v), cur_yrs, vo,
                                             it.ypix + 4 * ((CKB ? trow(p) * (a.W / 2) + tp(p) % C::TW / 2 : (p / C::TW) * a.W + p % C::TW) * y_px +
                                                            c * 16 * y_co), 0);
      if constexpr (F16) {
        vst[c] = v;
        // inline asm on purpose (also the mask code below): written with builtins (fmaxf, __ballot + selects) this epilogue sent
        // hipcc's register allocation from 251 VGPRs to 256 + 118 spilled
        if constexpr (BWD)                                         // signed values feed the next conv: max |v|
          asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(ymax) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        else
          asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "+v"(ymax) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
      }
    }
    if constexpr (BWD) {
      // no sign bits
    } else if constexpr (HALF) {
      // one channel tile: the ballot's 16-bit quarter kq of register r IS sample 4 kq + r's word
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned alo, ahi, d;
        asm volatile(
            "v_cmp_lt_f32 vcc, 0, %4\n\t"
            "s_mov_b32 %1, vcc_lo\n\t"
            "s_mov_b32 %2, vcc_hi\n\t"
            "v_writelane_b32 %0, %1, %5\n\t"
            "s_lshr_b32 %3, %1, 16\n\t"
            "v_writelane_b32 %0, %3, %6\n\t"
            "v_writelane_b32 %0, %2, %7\n\t"
            "s_lshr_b32 %3, %2, 16\n\t"
            "v_writelane_b32 %0, %3, %8"
            : "+v"(mbits), "=&s"(alo), "=&s"(ahi), "=&s"(d)
            : "v"(vst[0][r]), "n"(r), "n"(4 + r), "n"(8 + r), "n"(12 + r)
            : "vcc", "scc");
      }
      __builtin_amdgcn_raw_buffer_store_b16((unsigned short)mbits, cur_mrs, mvoff, (it.pix0 + (p / C::TW) * a.W + p % C::TW) * (a.cout / 8), 0);
    } else if constexpr (F16) {
      // sign bits -> lanes: the ballot of register r of channel tile c holds, in its 16-bit quarter kq, the 16 channel bits of
      // sample 4 kq + r.  SALU packs the quarters of the two tiles into one dword per sample (32 channels), v_writelane parks it
      // in lane 4 kq + r: 8 v_cmp + 16 v_writelane + 24 SALU per pixel, one VGPR, no per-lane selects
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned alo, ahi, d;
        asm volatile(
            "v_cmp_lt_f32 vcc, 0, %4\n\t"
            "s_mov_b32 %1, vcc_lo\n\t"
            "s_mov_b32 %2, vcc_hi\n\t"
            "v_cmp_lt_f32 vcc, 0, %5\n\t"
            "s_pack_ll_b32_b16 %3, %1, vcc_lo\n\t"
            "v_writelane_b32 %0, %3, %6\n\t"
            "s_pack_hh_b32_b16 %3, %1, vcc_lo\n\t"
            "v_writelane_b32 %0, %3, %7\n\t"
            "s_pack_ll_b32_b16 %3, %2, vcc_hi\n\t"
            "v_writelane_b32 %0, %3, %8\n\t"
            "s_pack_hh_b32_b16 %3, %2, vcc_hi\n\t"
            "v_writelane_b32 %0, %3, %9"
            : "+v"(mbits), "=&s"(alo), "=&s"(ahi), "=&s"(d)
            : "v"(vst[0][r]), "v"(vst[1][r]), "n"(r), "n"(4 + r), "n"(8 + r), "n"(12 + r)
            : "vcc");
      }
      __builtin_amdgcn_raw_buffer_store_b32(mbits, cur_mrs, mvoff, (it.pix0 + (p / C::TW) * a.W + p % C::TW) * (a.cout / 8), 0);
    }
  };
  // Pixel p's residual has landed when at most N VMEM operations issued after it are outstanding.  The tail issues,
  // per pixel, CW stores then CW loads, so after pixel p's loads come 2*CW*(PW-1-p) operations of the same tail
  // (+ 2*CW*p of THIS chunk's tail when the item has a single chunk).
  auto wait_res = [&](int p, bool also_last) __attribute__((always_inline)) {
    static_assert(CW == 2 || CW == 1, "operand list below");
    constexpr int NT = 2 * CW + (F16 ? 1 : 0);                     // VMEM operations of one pixel's tail (F16: + the mask store)
    const int n0 = NT * (PW - 1 - p) + (also_last ? NT * p : 0);
    const int n = n0 > 63 ? 63 : n0;                               // vmcnt is a 6-bit field: waiting for more is always safe
    if constexpr (CW == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(acc[p][0]), "+v"(acc[p][1]) : "n"(n));
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(acc[p][0]) : "n"(n));
  };

  // One 8-channel chunk from LDS stage `stage`, PIXEL-MAJOR: all three K-steps of pixel 0, then pixel 1, ...  With
  // the K-step-major order every pixel finished at the very end of the item's last chunk and was needed again at the
  // very start of the next item's first chunk: the residual loads had a third of a chunk to arrive and the MFMA waves
  // sat out an HBM round trip per item (the 15-20k-cycle "tail").  Now pixel p is final after (p+1)/PW of the LAST
  // chunk -- its stores and the next item's residual loads are issued right there -- and is first touched at p/PW of
  // the next FIRST chunk: every residual load has (PW-1)/PW of a chunk period to land.
  // always_inline: instantiated four ways but called eight times -- left as a call, everything captured by reference
  // (accumulators, fragments) went through scratch and flat memory
  int g = 0;                                                       // stream chunk index -> LDS stage g & 1
  auto chunk = [&](int stage, int ring_slot, auto FIRST, auto LAST, auto QUAD) __attribute__((always_inline)) {
#ifdef CMF_DBG_NOMFMA
    if (LAST) {                                    // timing-only build: the MFMA waves only run the tail and the barriers
#pragma unroll
      for (int p = 0; p < PW; ++p) {
        store_pixel(cur, p, f32x4{}, f32x4{});
        init_pixel(nxt, nxt_rrs, p);
      }
    }
    return;
#endif
    constexpr int KS = QUAD ? 3 : 2, NSTEP = KS * PW;              // step t = KS*p + s (pixel-major)
    if (LAST) load_omask(cur);                                     // PLAIN only: scalar loads, land during this chunk
    const unsigned char* Xh = smem + stage * C::BUF_BYTES;
    const unsigned char* Xl = Xh + C::XS_BYTES;
    const unsigned char* Wh = Xh + 2 * C::XS_BYTES;
    const unsigned char* Wl = Wh + C::WS_BYTES;
    bf16x8 ah[KS][CW], al[KS][CW], bh[BD], bl[BD];
    // centre K-step operands: lane groups 0..2 read the ring (hi at +0, lo at +RING_HALF), group 3 the live stage
    const unsigned char* ch_[2];
    const unsigned char* cl_[2];
    if (QUAD) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        ch_[par] = smem + (kq < 3 ? coff_ring[par] : stage * C::BUF_BYTES + coff_stage[par]);
        cl_[par] = smem + (kq < 3 ? coff_ring[par] + C::RING_HALF : stage * C::BUF_BYTES + C::XS_BYTES + coff_stage[par]);
      }
    }
    auto load_b = [&](int t) {
      const int s = t % KS, p = t / KS;
#ifdef CMF_DBG_NOLDS                                // timing-only: no B-fragment LDS reads at all (wrong results)
      asm volatile("" : "=v"(bh[t % BD]), "=v"(bl[t % BD]));
      return;
#endif
      const int q = tp(p), par = CKB ? trow(p) & 1 : p & 1;        // CKB: offsets indexed by the row's parity (phase folded in)
      if (s < 2) {
        bh[t % BD] = *reinterpret_cast<const bf16x8*>(Xh + boff[s][par] + C::pl(q) * 256);
        bl[t % BD] = *reinterpret_cast<const bf16x8*>(Xl + boff[s][par] + C::pl(q) * 256);
      } else {                                     // ring rows are TW pixels long, stage rows TWH: per-lane row step
        bh[t % BD] = *reinterpret_cast<const bf16x8*>(ch_[par] + q * 256 + (q / C::TW) * crow);
        bl[t % BD] = *reinterpret_cast<const bf16x8*>(cl_[par] + q * 256 + (q / C::TW) * crow);
      }
    };
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < CW; ++c) {
#ifdef CMF_DBG_NOA                                  // timing-only: no A-fragment reads either
        asm volatile("" : "=v"(ah[s][c]), "=v"(al[s][c]));
#else
        ah[s][c] = *reinterpret_cast<const bf16x8*>(Wh + (((s * COT + c) * 64) << 4) + aoff);
        al[s][c] = *reinterpret_cast<const bf16x8*>(Wl + (((s * COT + c) * 64) << 4) + aoff);
#endif
      }
    // non-QUAD chunk: park this octet's centre pixels in ring slot `ring_slot`: the two waves of a tile row split hi / lo,
    // 14 pixel rows of 256 B = 224 sixteen-byte units per wave
    u32x4 cp[4];
#ifndef CMF_DBG_NORING
    if (!QUAD) {
      const unsigned char* src = (cohalf ? Xl : Xh) + cpix * 256;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int u = lane + 64 * k;
        if (u < C::PW * 16) cp[k] = *reinterpret_cast<const u32x4*>(src + u * 16 + ((u >> 4) / C::TW) * (C::TWH - C::TW) * 256);
      }
    }
#endif
#pragma unroll
    for (int t = 0; t < BD - 1; ++t) load_b(t);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
      const int s = t % KS, p = t / KS;
      if (t + BD - 1 < NSTEP) load_b(t + BD - 1);
#ifndef CMF_DBG_NORING
      if (!QUAD && t == 2) {                                       // the copy's reads have long landed
        unsigned char* dst = smem + 2 * C::BUF_BYTES + ring_slot * C::RING_SLOT + (cohalf ? C::RING_HALF : 0) + wrow * C::TW * 256;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int u = lane + 64 * k;
          if (u < C::PW * 16) *reinterpret_cast<u32x4*>(dst + u * 16) = cp[k];
        }
      }
#endif
      if (FIRST && s == 0 && FRES) {
#pragma unroll
        for (int c = 0; c < CW; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (FIRST && s == 0 && !FRES) {
#ifndef CMF_DBG_SPREAD
        wait_res(p, false);
#endif
        if constexpr (F16) {                                       // the accumulators hold (2^k xscale) x the result
#pragma unroll
          for (int c = 0; c < CW; ++c) acc[p][c] *= rscale;
        }
      }
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        // D[row = Jacobian column][col = output channel] = X-fragment (as A) x W-fragment (as B): each lane then
        // holds 4 CONSECUTIVE columns (rows kq*4 + r) of channel cl -> 16-byte stores / residual loads
        if constexpr (F16) {
#define CMF_H8(v) __builtin_bit_cast(f16x8, v)                     /* the fragments are fp16 bit patterns */
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CMF_H8(bh[t % BD]), CMF_H8(al[s][c]), acc[p][c], 0, 0, 0);
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CMF_H8(bl[t % BD]), CMF_H8(ah[s][c]), acc[p][c], 0, 0, 0);
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CMF_H8(bh[t % BD]), CMF_H8(ah[s][c]), acc[p][c], 0, 0, 0);
#undef CMF_H8
        } else {
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t % BD], al[s][c], acc[p][c], 0, 0, 0);
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[t % BD], ah[s][c], acc[p][c], 0, 0, 0);
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t % BD], ah[s][c], acc[p][c], 0, 0, 0);
        }
      }
#ifdef CMF_DBG_SPREAD
      // timing only (wrong results): the item's tail -- per wave 28 stores + 28 residual loads -- spread evenly over ALL chunks of
      // the item instead of sitting in the last one: pixel p's pair of stores and loads is issued after its MFMAs in chunk
      // (p mod 8).  The ceiling of any restructuring that de-bursts the tail (second accumulator set, output staging in LDS ...).
      if (s == KS - 1 && !FRES && (p & 7) == (g & 7)) {
        store_pixel(cur, p, f32x4{}, f32x4{});
        init_pixel(nxt, nxt_rrs, p);
      }
#else
      if (LAST && s == KS - 1 && !FRES) {
        store_pixel(cur, p, f32x4{}, f32x4{});
        init_pixel(nxt, nxt_rrs, p);
      }
#endif
      // keep this step's reads-then-MFMAs(-then-tail) order: without the fence hipcc's scheduler re-clusters the
      // ds_reads next to their uses (lgkmcnt(0) before most MFMA groups) and the ring no longer hides LDS latency
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (!FRES) {
#pragma unroll
    for (int p = 0; p < PW; ++p) init_pixel(cur, nxt_rrs, p);
  }
  // item 0: a block of loads, not the tail pattern (+ the bias)
  static_assert(CW == 2 || CW == 1, "operand list below");
  if constexpr (CW == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias[1])::"memory");
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias_h1)::"memory");
  __syncthreads();                                                 // stage 0 ready
  for (int item = 0; item < n_items; ++item) {
    const bool has_next = item + 1 < n_items;
    [[maybe_unused]] auto nxt_yrs = cur_yrs;
    auto next_context = [&]() __attribute__((always_inline)) {                                   // derived right before the chunk that uses it
      int np, slice, cog;
      item_geom(has_next ? item + 1 : item, np, slice, cog, nxt);
      if constexpr (F16) {
        // fp16 variants: the store descriptors of THIS item are derived here, right before its last chunk, instead of riding
        // along as a (current, next) pair through every chunk: 16 SGPRs fewer in the steady state (hipcc had left a dead 16-byte
        // descriptor spill slot + the scavenging slot on the stack of the <2,7,2> form: 20 bytes of scratch that no instruction used)
        int cnp, cslice, ccog, ctile;
        decode(item, ctile, cslice, ccog, cnp);
        cur_yrs = y_rsrc(cnp, cslice, ccog);
        cur_mrs = m_rsrc(cnp, cslice, ccog);
      } else {
        nxt_yrs = y_rsrc(np, slice, cog);
      }
      nxt_rrs = r_rsrc(np, slice, cog, has_next);                  // last item: zero records, nothing is fetched
    };
    // nchunks = 4 G (launcher precondition): chunks 4m, 4m+1, 4m+2 run two K-steps and park their centre pixels, chunk
    // 4m+3 adds the four-octet centre K-step; the item's first chunk waits for the residual, its last one stores.
    // Written out as a fixed sequence (a run-time variant switch inside one loop body spilled 189 VGPRs).
    auto step = [&](auto FIRST, auto LAST, auto QUAD, auto SLOT) __attribute__((always_inline)) {
      STAMP(0, g, 0);
      chunk(g & 1, decltype(SLOT)::value, FIRST, LAST, QUAD);
      STAMP(0, g, 1);
      __syncthreads();                                             // stage (g+1)&1 ready, stage g&1 free
      STAMP(0, g, 2);
      ++g;
    };
    using T = std::true_type;
    using F = std::false_type;
    step(T{}, F{}, F{}, std::integral_constant<int, 0>{});
    step(F{}, F{}, F{}, std::integral_constant<int, 1>{});
    step(F{}, F{}, F{}, std::integral_constant<int, 2>{});
    for (int m = 1; m < nchunks / 4; ++m) {
      step(F{}, F{}, T{}, std::integral_constant<int, 0>{});
      step(F{}, F{}, F{}, std::integral_constant<int, 0>{});
      step(F{}, F{}, F{}, std::integral_constant<int, 1>{});
      step(F{}, F{}, F{}, std::integral_constant<int, 2>{});
    }
    next_context();
    step(F{}, T{}, T{}, std::integral_constant<int, 0>{});
    if constexpr (FRES) {
      // epilogue with the residual: groups of GP pixels -- their residual tiles land in the registers the chunk's fragments just
      // freed (hipcc counts these waits itself: plain builtin loads, nothing else in flight), then scale + bias + residual,
      // stores, sign bits, running maximum per pixel
      // (BWD with 64-channel items: two tiles per pixel and channel tile -- smaller groups still)
      constexpr int GP = BWD && CW == 2 ? (PW > 14 ? 4 : 2) : PW > 14 ? PW / 4 : PW / 2;
      static_assert(PW % GP == 0, "whole groups");
      int np, slice, cog;
      {
        int tile;
        decode(item, tile, slice, cog, np);
      }
      auto tile_rsrc = [&](const float* base, bool on) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(base);
        return __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u)), 0, on ? 0x7fffff00 : 0, RS_FLAGS);
      };
      const long long slo = (long long)slice * (a.r_sl ? a.r_sl : 16);
      // BWD: the residual is optional (zero records: the loads return 0 without touching memory)
      const auto rrs = tile_rsrc(a.r ? a.r + (long long)np * a.r_np + slo + (long long)cog * COG * r_co : a.y, a.r != nullptr);
      [[maybe_unused]] const int fo_co = (int)a.fo_co, fo_px = (int)a.fo_px;
      [[maybe_unused]] const int fovoff = 4 * ((cohalf * CW * 16 + cl) * fo_co + kq * 4);
      [[maybe_unused]] const auto frs = tile_rsrc(BWD ? a.fo + (long long)np * a.fo_np + (long long)slice * (a.y_sl ? a.y_sl : 16) +
                                                            (long long)cog * COG * fo_co : a.y, BWD);
#pragma unroll
      for (int p0 = 0; p0 < PW; p0 += GP) {
        f32x4 rb[GP][CW];
        [[maybe_unused]] f32x4 mb[GP][CW];
#pragma unroll
        for (int q = 0; q < GP; ++q)
#pragma unroll
          for (int c = 0; c < CW; ++c) {
            const int p = p0 + q;
            rb[q][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rrs, rvoff, cur.rpix + 4 * (((p / C::TW) * a.W + p % C::TW) * r_px + c * 16 * r_co), 0));
            if constexpr (BWD)
              mb[q][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                  frs, fovoff, 4 * ((cur.pix0 + (p / C::TW) * a.W + p % C::TW) * fo_px + c * 16 * fo_co), 0));
          }
#pragma unroll
        for (int q = 0; q < GP; ++q) {
          if constexpr (BWD) store_pixel(cur, p0 + q, rb[q][0], rb[q][CW - 1], mb[q][0], mb[q][CW - 1]);
          else store_pixel(cur, p0 + q, rb[q][0], rb[q][CW - 1]);
        }
      }
    }
    cur.ypix = nxt.ypix, cur.rpix = nxt.rpix, cur.pix0 = nxt.pix0, cur.np = nxt.np, cur.cog = nxt.cog;
    if constexpr (!F16) cur_yrs = nxt_yrs;
  }
  STAMP2(0, 11, 0);
  if constexpr (F16) {
    // The last item's "next residual" loads (zero-record descriptor) are still landing in the accumulator registers: drain them
    // and keep those registers allocated until then -- hipcc believes the inline-asm loads completed where they were issued, and
    // reused the dead accumulators for the reduction below (the maximum came out too small, differently from run to run).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
      for (int c = 0; c < CW; ++c) asm volatile("" ::"v"(acc[p][c]));
    if (a.amax_out) {                                              // values >= 0 order like their bit patterns
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ymax = __builtin_fmaxf(ymax, __shfl_xor(ymax, o, 64));
      if (lane == 0 && n_items > 0) atomicMax(reinterpret_cast<int*>(a.amax_out), __builtin_bit_cast(int, ymax));
    }
  }
  STAMP2(0, 11, 1);
  STAMP2(0, 8, 1);
  SPAN(1);
}

// weight pre-split / pre-arrangement: out[cog][chunk][hl][s][cot 4][kq][co 16][8] bf16
// transpose: `w` is the LAYER's weight [cin][cout][3][3] and the pack is of its adjoint operator -- channels swapped, taps
// flipped -- [cout][cin][tap] = w[ci][co][8 - tap]
// f16 != 0 (cmf_pack_weight_f16x3): fp16 halves of w 2^k, the scale read from the trailer pack_f16_scale_kernel wrote behind the
// `total` elements
__global__ void pack_weight_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout,
                                          int cin, long long total, int transpose, int f16) {
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
  // K-step 0: taps 0..3, K-step 1: taps 5..8 of octet ch (lane group kq = tap); K-step 2 (used in chunks ch % 4 == 3
  // only): the CENTRE tap of octets ch-3 .. ch (lane group kq = octet)
  const int co = cog * 64 + cot * 16 + col;
  int ci = ch * 8 + j, tap = s == 0 ? kq : 5 + kq;
  if (s == 2) {
    tap = (ch & 3) == 3 ? 4 : 9;
    ci = (ch - 3 + kq) * 8 + j;
  }
  float v = 0.f;
  if (co < cout && tap < 9) v = transpose ? w[((long long)ci * cout + co) * 9 + (8 - tap)] : w[((long long)co * cin + ci) * 9 + tap];
  if (f16) {
    v *= reinterpret_cast<const float*>(out + total)[0];          // exact: a power of two
    const _Float16 h = (_Float16)v;
    const _Float16 r = hl ? (_Float16)(v - (float)h) : h;
    out[i] = __builtin_bit_cast(unsigned short, r);
    return;
  }
  const __bf16 h = (__bf16)v;
  const float hf = (float)h;
  const __bf16 r = hl ? (__bf16)(v - hf) : h;
  out[i] = __builtin_bit_cast(unsigned short, r);
}

// trailer {2^k, 2^-k, 0, 0} with max |w| 2^k in [2^11, 2^12): one workgroup, no host synchronisation
__global__ __launch_bounds__(1024) void pack_f16_scale_kernel(const float* __restrict__ w, long long n, float* __restrict__ trailer) {
  __shared__ float red[16];
  float m = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    const int ex = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);          // m = f 2^(ex - 126), f in [0.5, 1)
    int k = (ex == 0 || ex == 255) ? 0 : 12 - (ex - 126);
    k = k < -60 ? -60 : k > 60 ? 60 : k;
    trailer[0] = pow2i(k);
    trailer[1] = pow2i(-k);
    trailer[2] = trailer[3] = 0.f;
  }
}

template <int COT, int PXW, int MODE, bool F16 = false, bool FRES = false, bool CKB = false>
int launch(const cmf_conv_tangent_args& a, hipStream_t s) {
  using C = BCfg<COT, PXW>;
  const int tiles_x = cmf_ceil_div(a.W, C::TW), tiles = tiles_x * cmf_ceil_div(a.H, C::TH);
  const int nslices = a.nc / 16, ncog = (F16 && COT == 2) ? a.cout / 32 : cmf_ceil_div(a.cout, 64);
  const long long total = (long long)tiles * nslices * ncog * a.np;
  if (total > 0x7fffffffLL) return CMF_ERANGE;
  auto k = conv_tangent_bf16x3_kernel<COT, PXW, MODE, F16, FRES, CKB>;
  constexpr int lds = C::LDS_BYTES;
  if (hipError_t e = cmf_set_dynamic_lds((const void*)k, lds); e != hipSuccess) return (int)e;   // per device (runtime.hip)
  const int n_cu = cmf_device_cus();
  int grid = (int)(total < n_cu ? total : n_cu);                // persistent: one 112 KiB workgroup per CU
#ifdef CMF_DBG_STAMP
  if (const char* e = getenv("CMF_DBG_GRID")) grid = atoi(e) < grid ? atoi(e) : grid;   // diagnostic: fewer active CUs
#endif
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a, tiles_x, tiles, nslices, ncog, (int)total);
  CMF_LAUNCH_CHECK();
  return 0;
}

template <int PXW>
int launch_cot(const cmf_conv_tangent_args& a, hipStream_t s) {
  // co tiles per workgroup: 64 / 32 channels; factor code specialised for self-relu / relu / anything else
  if (a.fmode == CMF_F_SELF_RELU) return (a.cout > 32) ? launch<4, PXW, 2>(a, s) : launch<2, PXW, 2>(a, s);
  if (a.fmode == CMF_F_RELU) return (a.cout > 32) ? launch<4, PXW, 1>(a, s) : launch<2, PXW, 1>(a, s);
  if (a.fmode == CMF_F_RELU_BITS) return (a.cout > 32) ? launch<4, PXW, 3>(a, s) : launch<2, PXW, 3>(a, s);
  if (a.fmode == CMF_F_NONE && a.fomode == CMF_F_RELU_BITS && a.r) {   // in-place skip connection: 64-channel groups only
    return launch<4, PXW, 5>(a, s);
  }
  if (a.fmode == CMF_F_NONE) return (a.cout > 32) ? launch<4, PXW, 4>(a, s) : launch<2, PXW, 4>(a, s);
  return (a.cout > 32) ? launch<4, PXW, 0>(a, s) : launch<2, PXW, 0>(a, s);
}

inline bool fits_int(long long v) { return v >= 0 && v < (1LL << 29); }   // element offsets; x4 bytes must fit 32 bits

}  // namespace

extern "C" int cmf_pack_weight_bf16x3_t(const float* w, void* out, int cout, int cin, int transpose, long long* out_bytes, void* stream);
extern "C" int cmf_pack_weight_bf16x3(const float* w, void* out, int cout, int cin, long long* out_bytes, void* stream) {
  return cmf_pack_weight_bf16x3_t(w, out, cout, cin, 0, out_bytes, stream);
}
extern "C" int cmf_pack_weight_bf16x3_t(const float* w, void* out, int cout, int cin, int transpose, long long* out_bytes, void* stream) {
  if (cout <= 0 || cin <= 0 || cin % 8) return CMF_EINVAL;
  const long long total = (long long)((cout + 63) / 64) * (cin / 8) * 2 * 3 * 4 * 4 * 16 * 8;   // bf16 elements
  if (out_bytes) *out_bytes = total * 2;
  if (!out) return 0;
  if (!w) return CMF_EINVAL;
  hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     w, (unsigned short*)out, cout, cin, total, transpose ? 1 : 0, 0);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_pack_weight_f16x3(const float* w, void* out, int cout, int cin, int transpose, long long* out_bytes, void* stream) {
  if (cout <= 0 || cin <= 0 || cin % 8) return CMF_EINVAL;
  const long long total = (long long)((cout + 63) / 64) * (cin / 8) * 2 * 3 * 4 * 4 * 16 * 8;   // fp16 elements, then the trailer
  if (out_bytes) *out_bytes = total * 2 + 16;
  if (!out) return 0;
  if (!w || (uintptr_t)out % 16) return CMF_EINVAL;
  hipLaunchKernelGGL(pack_f16_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)cout * cin * 9,
                     reinterpret_cast<float*>((unsigned short*)out + total));
  CMF_LAUNCH_CHECK();
  hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     w, (unsigned short*)out, cout, cin, total, transpose ? 1 : 0, 1);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_conv_tangent_bf16x3(const cmf_conv_tangent_args* ap, void* stream) {
  if (!ap) return CMF_EINVAL;
  const cmf_conv_tangent_args& a = *ap;
  if (!a.x || !a.w || !a.y || a.np <= 0 || a.cin <= 0 || a.cout <= 0 || a.H <= 0 || a.W <= 0) return CMF_EINVAL;
  if (a.taps != 9 || a.cin % 32 || a.nc <= 0 || a.nc % 16) return CMF_EINVAL;   // K packing works on groups of 4 octets
  if (a.fmode < CMF_F_NONE || a.fmode > CMF_F_RELU_BITS) return CMF_EINVAL;
  if (a.fmode != CMF_F_SELF_RELU && a.fmode != CMF_F_NONE && !a.f) return CMF_EINVAL;
  if (a.mask_out) return CMF_EINVAL;                            // sign-bit output: cmf_conv_tangent only
  if (a.fo) {
    // output-side factor: only as a relu' BIT MASK, without input factor and without residual (the residual is the
    // accumulators' initial value and would be masked with the product), whole groups of 64 output channels
    // -- unless the residual IS the output tensor (the in-place skip connection: masked-off lanes keep what is in memory)
    const bool inplace = a.r == a.y && a.r_np == a.y_np && a.r_co == a.y_co && a.r_px == a.y_px && a.r_sl == a.y_sl;
    if (a.fmode != CMF_F_NONE || a.fomode != CMF_F_RELU_BITS || (a.r && !inplace) || a.cout % 64) return CMF_EINVAL;
    if (a.r && a.bias) return CMF_EINVAL;
    if (a.fo_np % 4 || (uintptr_t)a.fo % 4 || a.fo_np < (long long)a.H * a.W * (a.cout / 8)) return CMF_EINVAL;
  }
  if (a.fmode == CMF_F_RELU_BITS && a.f_np < (long long)a.H * a.W * (a.cin / 8)) return CMF_EINVAL;
  if ((a.x_np | a.x_ci | a.x_px | a.x_sl | a.y_sl | a.r_sl) % 4 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return CMF_EINVAL;
  if ((a.y_np | a.y_co | a.y_px) % 4 || ((uintptr_t)a.y % 16)) return CMF_EINVAL;            // 16-byte stores
  if (a.r && ((a.r_np | a.r_co | a.r_px) % 4 || ((uintptr_t)a.r % 16))) return CMF_EINVAL;    // 16-byte residual loads
  const long long HW = (long long)a.H * a.W;
  if (!fits_int((a.cin + 8) * a.x_ci + HW * a.x_px + a.nc) || !fits_int((a.cin + 8) * a.f_ci + HW * a.f_px) ||
      !fits_int((a.cout + 64) * a.y_co + HW * a.y_px + a.nc + 64) ||          // + 64: below the y descriptor's 0x7fffff00 records
      (a.r && !fits_int((a.cout + 64) * a.r_co + HW * a.r_px + a.nc)) ||
      HW > (1 << 24))
    return CMF_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  // the kernel has no partial-tile / partial-channel-group code: whole 2 x 14 tiles (14- and 28-wide images) or whole
  // 4 x 8 tiles (16- / 32-wide: CIFAR) and whole groups of 64 (or exactly 32) output channels only
  const bool t14 = a.W % 14 == 0 && a.H % 2 == 0, t8 = a.W % 8 == 0 && a.H % 4 == 0;
  if (!(t14 || t8) || !(a.cout % 64 == 0 || a.cout == 32)) return CMF_EINVAL;
  if (a.bias && a.cout > 64) return CMF_EINVAL;                 // the per-channel constants are fetched once per launch
  if (a.live) {
    // checkerboard output: the forward tangent conv with a relu' factor, whole 64-channel groups, nothing in the epilogue but the residual
    if (a.live < 0 || a.live > 2 || (a.fmode != CMF_F_RELU && a.fmode != CMF_F_RELU_BITS) || a.fo || a.bias || a.cout % 64 || a.W % 2)
      return CMF_EINVAL;
    if (a.fmode == CMF_F_RELU) return t14 ? launch<4, 7, 1, false, false, true>(a, s) : launch<4, 4, 1, false, false, true>(a, s);
    return t14 ? launch<4, 7, 3, false, false, true>(a, s) : launch<4, 4, 3, false, false, true>(a, s);
  }
  if (a.fmode == CMF_F_NONE && !a.fo) {
    // the PLAIN kernel fetches its output mask unconditionally: without one, let it read (and ignore) the first bytes of
    // every sample's input tensor -- H*W*cout/8 bytes per sample, always inside x (cin*nc*4 >= 2 KiB per pixel)
    cmf_conv_tangent_args b = a;
    b.fo = a.x;
    b.fo_np = 0;
    b.fomode = CMF_F_NONE;
    return t14 ? launch_cot<7>(b, s) : launch_cot<4>(b, s);
  }
  return t14 ? launch_cot<7>(a, s) : launch_cot<4>(a, s);
}

extern "C" int cmf_conv_tangent_f16x3(const cmf_conv_tangent_args* ap, void* stream) {
  return cmf_conv_tangent_f16x3_item(ap, 0, stream);
}

extern "C" int cmf_conv_tangent_f16x3_item(const cmf_conv_tangent_args* ap, int item_channels, void* stream) {
  if (!ap || (item_channels != 0 && item_channels != 32 && item_channels != 64)) return CMF_EINVAL;
  const cmf_conv_tangent_args& a = *ap;
  if (!a.x || !a.w || !a.y || a.np <= 0 || a.cin <= 0 || a.cout <= 0 || a.H <= 0 || a.W <= 0) return CMF_EINVAL;
  if (a.taps != 9 || a.cin % 32 || a.nc <= 0 || a.nc % 16 || a.cout % 64 || a.live) return CMF_EINVAL;
  // forward: the input's own relu, no output factor.  backward (data gradient): plain input, the adjoint pack, y = [fo > 0] . conv
  // + r with fo a float tensor laid out like y (fomode CMF_F_SELF_RELU: cmf_conv_tangent's primal-backward form), no bias, no sign bits
  const bool bwd = a.fmode == CMF_F_NONE && a.fo && a.fomode == CMF_F_SELF_RELU;
  if (!bwd && (a.fmode != CMF_F_SELF_RELU || a.fo)) return CMF_EINVAL;
  if (bwd && (a.bias || a.mask_out || (a.fo_np | a.fo_co | a.fo_px) % 4 || (uintptr_t)a.fo % 16)) return CMF_EINVAL;
  if (a.mask_out && (a.mask_np % 4 || (uintptr_t)a.mask_out % 4 || a.mask_np < (long long)a.H * a.W * (a.cout / 8) ||
                     16 * a.mask_np >= 0x7fffff00LL))
    return CMF_EINVAL;
  if ((a.x_np | a.x_ci | a.x_px | a.x_sl | a.y_sl | a.r_sl) % 4 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return CMF_EINVAL;
  if ((a.y_np | a.y_co | a.y_px) % 4 || ((uintptr_t)a.y % 16)) return CMF_EINVAL;
  if (a.r && ((a.r_np | a.r_co | a.r_px) % 4 || ((uintptr_t)a.r % 16))) return CMF_EINVAL;
  const long long HW = (long long)a.H * a.W;
  if (!fits_int((a.cin + 8) * a.x_ci + HW * a.x_px + a.nc) || !fits_int((a.cout + 64) * a.y_co + HW * a.y_px + a.nc + 64) ||
      (a.r && !fits_int((a.cout + 64) * a.r_co + HW * a.r_px + a.nc)) ||
      (bwd && !fits_int((a.cout + 64) * a.fo_co + HW * a.fo_px + a.nc)) || HW > (1 << 24))
    return CMF_ERANGE;
  const bool t14 = a.W % 14 == 0 && a.H % 2 == 0, t8 = a.W % 8 == 0 && a.H % 4 == 0;
  if (!(t14 || t8)) return CMF_EINVAL;
  if (a.bias && a.cout > 64) return CMF_EINVAL;                 // the per-channel constants are fetched once per launch
  hipStream_t s = (hipStream_t)stream;
  // items of 64 output channels -- or of 32 when that still leaves CUs without one (a 32-sample CIFAR shard: 64 -> 128 items; the
  // 64-sample MNIST shard's 28 x 28 layers: 112 -> 224): a launch of one item per workgroup lasts one item, and half an item's
  // MFMA work is the shorter item
  const int TH = t14 ? 2 : 4, TW = t14 ? 14 : 8;
  const long long items64 = (long long)(a.H / TH) * (a.W / TW) * (a.nc / 16) * (a.cout / 64) * a.np;
  // 4 x 8 tiles (16- / 32-wide images): ALWAYS 32-channel items.  A 64-channel item there is 16 pixels x 2 channel tiles = 128
  // accumulator VGPRs per wave on top of the fp16 epilogue's state: hipcc needed 263 - 278 registers and spilled 7 - 22 of them to
  // scratch (round 4); the two sizes give bit-identical results, so the request for 64 is served by two half items.
  const bool half = !t14 || (item_channels ? item_channels == 32 : 2 * items64 <= cmf_device_cus());
  if (bwd) {
    if (half) return t14 ? launch<2, 7, 4, true, true>(a, s) : launch<2, 4, 4, true, true>(a, s);
    return launch<4, 7, 4, true, true>(a, s);
  }
  if (half) {
    if (a.r) return t14 ? launch<2, 7, 2, true, true>(a, s) : launch<2, 4, 2, true, true>(a, s);
    return t14 ? launch<2, 7, 2, true>(a, s) : launch<2, 4, 2, true>(a, s);
  }
  if (a.r) return launch<4, 7, 2, true, true>(a, s);
  return launch<4, 7, 2, true>(a, s);
}
