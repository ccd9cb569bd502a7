// Tangent convolution / linear layer on fp32 MFMA for gfx950 (MI355X).
//
// Computes, for all NC Jacobian columns of a sample at once,
//     y(np, co, px, :) = sum_{ci,tap} W[co][ci][tap] * F(np, ci, px+tap) * x(np, ci, px+tap, :)  [+ r]
// i.e. the tangent half of the reference's get_conv2d_jvp / get_linear_jvp
// (cmf/models/components/jvp_layers.py:49-64) for every column of the non-square Jacobian, with the
// preceding activation's derivative (jvp_layers.py:38-47) applied while the input tile is staged and
// the residual add of ResidualBlock.jvp (networks.py:62-79) fused into the store.
//
// Mapping to the hardware:
//   * GEMM view per sample: M = cout (A = packed weights), N = (pixel, Jacobian column), K = (tap, cin).
//     The Jacobian column is the contiguous dimension in HBM, so a B-operand row is one 64-byte
//     segment and the activation derivative is constant along N: it is a per-K-row scale.
//   * v_mfma_f32_16x16x4_f32: a wave owns PXW pixels x 16 columns x (COT*16) output channels
//     = PXW*COT accumulator tiles (4 VGPRs each); 4 waves per workgroup cover a 2 x 2*PXW pixel tile
//     (3x3) or 4*PXW flat pixels (1x1).  PXW = 7 fits 14- and 28-wide images exactly, 8 fits 16/32.
//   * K loop in chunks of 8 input channels: the halo tile [8][pix][16] and the weight slab
//     [taps][8][cout] are staged through LDS; both LDS images are padded so that the two 16-lane
//     groups of a ds_read_b32 half-wave hit disjoint banks (row stride == 16 mod 32 dwords).
//   * global -> register prefetch of chunk c+1 is issued before the MFMAs of chunk c and written to
//     LDS after them (issue-early / write-late), so HBM/L2 latency hides under ~16k MFMA cycles.
#include <type_traits>
#include "common.h"

namespace {

constexpr int CIC = 8;  // input channels per LDS chunk

template <int TAPS, int COT, int PXW>
struct Cfg {
  static constexpr int TW = 2 * PXW;                              // 3x3 tile width
  static constexpr int TWH = TW + 2;                              // + halo
  static constexpr int PIXH = (TAPS == 9) ? 4 * TWH : 4 * PXW;    // staged pixels per channel
  static constexpr int XS_CI = PIXH * 16 + 16;                    // dwords; == 16 (mod 32)
  static constexpr int WS_CI = (COT % 2) ? COT * 16 : COT * 16 + 16;
  static constexpr int XS_FLOATS = CIC * XS_CI;
  static constexpr int WS_FLOATS = TAPS * CIC * WS_CI;
  static constexpr int NX_ITEMS = CIC * PIXH * 4;                 // float4 items of the X chunk
  static constexpr int NXIT = (NX_ITEMS + 255) / 256;
  static constexpr int NW_ITEMS = TAPS * CIC * COT * 4;
  static constexpr int NWIT = (NW_ITEMS + 255) / 256;
};

template <int TAPS, int COT, int PXW>
__global__ __launch_bounds__(256, 2) void conv_tangent_kernel(cmf_conv_tangent_args a, int tiles_x, int ntiles,
                                                                int nslices, int ncog, int cin_pad, int nsub) {
  using C = Cfg<TAPS, COT, PXW>;
  __shared__ __attribute__((aligned(16))) float smem[C::XS_FLOATS + C::WS_FLOATS];
  float* Xs = smem;
  float* Ws = smem + C::XS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;

  // XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so
  // with the natural order the 16-column slices of one 128-byte line and the overlapping halos of adjacent
  // pixel tiles would be fetched by different L2s (measured: FETCH_SIZE 3-5x the algorithmic bytes).  Remap
  // so that XCD k works through a contiguous range of logical ids, ordered slice-fastest, then tile, then
  // sample: concurrently resident workgroups of an XCD then share lines and halos in its L2.  Bijective for
  // any grid size (cdna_hip_programming.md section 5, XCD swizzle); affects speed only, never results.
  int tile, slice, cog, np;
  {
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
  // nsub > 1: a 64-channel group is dealt to nsub workgroups of COT*16 channels each (small grids: more, lighter items)
  const int cosub = (cog % nsub) * COT * 16;
  cog /= nsub;
  const int HW = a.H * a.W;

  int y0 = 0, x0 = 0, p0 = 0;
  if (TAPS == 9) {
    y0 = 2 * (tile / tiles_x);
    x0 = C::TW * (tile % tiles_x);
  } else {
    p0 = tile * 4 * PXW;
  }

  const float* xb = a.x + (long long)np * a.x_np + (long long)slice * (a.x_sl ? a.x_sl : 16);
  const int fgrp = a.f_group > 1 ? a.f_group : 1;
  const float* fb = a.f ? a.f + (long long)(np / fgrp) * a.f_np + (np % fgrp) : nullptr;
  const float* wb = a.w + (long long)cog * TAPS * cin_pad * 64 + cosub;  // one co-group slab: < 2^31 floats

  // ---- per-thread staging plan (chunk independent part) ----
  // Every global load below is UNCONDITIONAL (invalid items read element 0 and are zeroed by their
  // factor at commit time): a conditional load makes hipcc branch and drain vmcnt per item.
  const int x_ci = (int)a.x_ci, x_px = (int)a.x_px, f_ci = (int)a.f_ci, f_px = (int)a.f_px;
  int xo[C::NXIT], fo[C::NXIT];
  unsigned okbits = 0;                   // validity per item; applied arithmetically at commit time
#pragma unroll
  for (int it = 0; it < C::NXIT; ++it) {
    const int i = tid + 256 * it;
    const int q = i & 3, rowi = i >> 2;
    const int pix = rowi % C::PIXH, ci = rowi / C::PIXH;
    bool ok = i < C::NX_ITEMS;
    int gpix;
    if (TAPS == 9) {
      const int hy = pix / C::TWH, hx = pix % C::TWH;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      gpix = gy * a.W + gx;
    } else {
      gpix = p0 + pix;
      ok = ok && gpix < HW;
    }
    xo[it] = ok ? 4 * (ci * x_ci + gpix * x_px + q * 4) : 0;     // BYTE offsets against wave-uniform bases
    fo[it] = ok ? 4 * (ci * f_ci + gpix * f_px) : 0;
    okbits |= (ok ? 1u : 0u) << it;
  }

  f32x4 xr[C::NXIT];
  float fr[C::NXIT];
  f32x4 wr[C::NWIT];
#pragma unroll
  for (int it = 0; it < C::NXIT; ++it) fr[it] = 0.f;   // stays 0 when fmode == NONE (never loaded)
  // NONE: 1   RELU: [f>0]   TANH: 1 - f^2   RAW: f
  const float fc0 = (a.fmode == CMF_F_NONE || a.fmode == CMF_F_TANH || a.fmode == CMF_F_SELF_RELU) ? 1.f : 0.f;
  const float fc1 = (a.fmode == CMF_F_RELU) ? 1.f : 0.f;
  const float fc2 = (a.fmode == CMF_F_RAW) ? 1.f : 0.f;
  const float fc3 = (a.fmode == CMF_F_TANH) ? -1.f : 0.f;
  const bool selfrelu = a.fmode == CMF_F_SELF_RELU;          // elementwise relu of the loaded values themselves
  unsigned okchunk = 0;                  // okbits restricted to channels < cin for the chunk being prefetched

  // Loads are unconditional and always consumed (a select on validity lets hipcc sink a load into a branch +
  // s_waitcnt vmcnt(0), serialising the prefetch); invalid items read element 0 and get multiplier 0 at commit.
  auto prefetch = [&](int ci0) {
    okchunk = 0;
#pragma unroll
    for (int it = 0; it < C::NXIT; ++it) {
      const int ci = ((tid + 256 * it) >> 2) / C::PIXH;
      const bool ok = ((okbits >> it) & 1u) && (ci0 + ci) < a.cin;
      okchunk |= (ok ? 1u : 0u) << it;
      const unsigned off = ok ? 4u * (unsigned)(ci0 * x_ci) + (unsigned)xo[it] : 0u;
      xr[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(xb) + off);
    }
    if (a.fmode != CMF_F_NONE && a.fmode != CMF_F_SELF_RELU) {
#pragma unroll
      for (int it = 0; it < C::NXIT; ++it) {
        const unsigned off = ((okchunk >> it) & 1u) ? 4u * (unsigned)(ci0 * f_ci) + (unsigned)fo[it] : 0u;
        fr[it] = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(fb) + off);
      }
    }
#pragma unroll
    for (int it = 0; it < C::NWIT; ++it) {
      int i = tid + 256 * it;
      i = i < C::NW_ITEMS ? i : C::NW_ITEMS - 1;
      const int q = i % (COT * 4), rowi = i / (COT * 4);
      const int ci = rowi % CIC, tap = rowi / CIC;
      const unsigned off = 4u * (unsigned)((tap * cin_pad + ci0 + ci) * 64 + q * 4);
      wr[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(wb) + off);
    }
  };

  auto commit = [&]() {
    const unsigned okcommit = okchunk;   // validity of the chunk whose registers are being committed
#pragma unroll
    for (int it = 0; it < C::NXIT; ++it) {
      const int i = tid + 256 * it;
      if (i < C::NX_ITEMS) {
        const int q = i & 3, rowi = i >> 2;
        const int pix = rowi % C::PIXH, ci = rowi / C::PIXH;
        // branch-free multiplier: ok * (c0 + c1*[f>0] + c2*f + c3*f*f) with wave-uniform mode coefficients
        const float f = fr[it];
        const float m = (((okcommit >> it) & 1u) ? 1.f : 0.f) * (fc0 + fc1 * (f > 0.f ? 1.f : 0.f) + f * (fc2 + fc3 * f));
        f32x4 xv = xr[it];
        if (selfrelu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) xv[c] = fmaxf(xv[c], 0.f);
        }
        *reinterpret_cast<f32x4*>(Xs + ci * C::XS_CI + pix * 16 + q * 4) = xv * m;
      }
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

  // ---- accumulators ----
  f32x4 acc[PXW][COT];
#pragma unroll
  for (int p = 0; p < PXW; ++p)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wrow = wave >> 1, wx = (wave & 1) * PXW;
  const float* a_base = Ws + kq * C::WS_CI + cl;
  const float* b_base =
      Xs + kq * C::XS_CI + cl + ((TAPS == 9) ? (wrow * C::TWH + wx) * 16 : (wave * PXW) * 16);

  const int nchunks = (a.cin + CIC - 1) / CIC;
  prefetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    commit();
    __syncthreads();
    if (ch + 1 < nchunks) prefetch((ch + 1) * CIC);

#pragma unroll
    for (int kg = 0; kg < CIC / 4; ++kg) {
      if (ch * CIC + kg * 4 >= a.cin) break;              // all-padding K group (the 1 -> 64 / 2 -> 64 first convs): skip
      if (TAPS == 9) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          float brow[PXW + 2];
#pragma unroll
          for (int j = 0; j < PXW + 2; ++j) brow[j] = b_base[kg * 4 * C::XS_CI + (dy * C::TWH + j) * 16];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            float av[COT];
#pragma unroll
            for (int c = 0; c < COT; ++c) av[c] = a_base[((dy * 3 + dx) * CIC + kg * 4) * C::WS_CI + c * 16];
#pragma unroll
            for (int p = 0; p < PXW; ++p)
#pragma unroll
              for (int c = 0; c < COT; ++c)
                acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(brow[p + dx], av[c], acc[p][c], 0, 0, 0);
          }
        }
      } else {
        float av[COT];
#pragma unroll
        for (int c = 0; c < COT; ++c) av[c] = a_base[(kg * 4) * C::WS_CI + c * 16];
#pragma unroll
        for (int p = 0; p < PXW; ++p) {
          const float bv = b_base[kg * 4 * C::XS_CI + p * 16];
#pragma unroll
          for (int c = 0; c < COT; ++c)
            acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av[c], acc[p][c], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  The MFMA operands are SWAPPED (x value as A, weight as B): D[row = Jacobian column][col = output
  // channel], so lane (kq, cl) holds the 4 CONSECUTIVE columns kq*4 .. kq*4+3 of channel cl of tile (p, c) -> one 16-byte
  // store / residual load per tile instead of four 4-byte ones (same products, same accumulation order, same bits).
  // The per-channel constant is fetched once, not per store.  Offsets fit 32 bits (host-checked). ----
  const int y_co = (int)a.y_co, y_px = (int)a.y_px, r_co = (int)a.r_co, r_px = (int)a.r_px;
  float* ybase = a.y + (long long)np * a.y_np + (long long)slice * (a.y_sl ? a.y_sl : 16) + kq * 4;
  const float* rbase = a.r ? a.r + (long long)np * a.r_np + (long long)slice * (a.r_sl ? a.r_sl : 16) + kq * 4 : nullptr;
  const int co0 = cog * 64 + cosub + cl;
  const bool full = (cog * 64 + cosub + COT * 16) <= a.cout;   // uniform: no per-channel bound checks needed
  float bias[COT];
#pragma unroll
  for (int c = 0; c < COT; ++c) bias[c] = (a.bias && (full || co0 + c * 16 < a.cout)) ? a.bias[co0 + c * 16] : 0.f;

  auto store_all = [&](auto has_res, auto is_full) {
#pragma unroll
    for (int p = 0; p < PXW; ++p) {
      int gpix;
      bool ok;
      if (TAPS == 9) {
        const int gy = y0 + wrow, gx = x0 + wx + p;
        ok = gy < a.H && gx < a.W;
        gpix = gy * a.W + gx;
      } else {
        gpix = p0 + wave * PXW + p;
        ok = gpix < HW;
      }
      if (!ok) continue;                                  // wave-uniform
      float* yp = ybase + gpix * y_px + co0 * y_co;
      const float* rp = has_res ? rbase + gpix * r_px + co0 * r_co : nullptr;
#pragma unroll
      for (int c = 0; c < COT; ++c) {
        if (is_full || (co0 + c * 16) < a.cout) {
          f32x4 mo = {1.f, 1.f, 1.f, 1.f};                 // output-side factor (reverse sweep): NONE / RELU / TANH / RAW
          if (a.fo && a.fomode == CMF_F_SELF_RELU) {
            // per-COLUMN relu' (primal backward, 16 samples in the column slots): fo is laid out like y
            const f32x4 fv = *reinterpret_cast<const f32x4*>(a.fo + (long long)np * a.fo_np + (long long)slice * (a.y_sl ? a.y_sl : 16) +
                                                             kq * 4 + (long long)(co0 + c * 16) * a.fo_co + (long long)gpix * a.fo_px);
#pragma unroll
            for (int r = 0; r < 4; ++r) mo[r] = fv[r] > 0.f ? 1.f : 0.f;
          } else if (a.fo) {
            const float fv = a.fo[(long long)np * a.fo_np + (long long)(co0 + c * 16) * a.fo_co + (long long)gpix * a.fo_px];
            const float m1 = a.fomode == CMF_F_RELU ? (fv > 0.f ? 1.f : 0.f) : a.fomode == CMF_F_TANH ? 1.f - fv * fv : fv;
            mo = f32x4{m1, m1, m1, m1};
          }
          f32x4 v = acc[p][c] * mo + bias[c];
          if (has_res) v += *reinterpret_cast<const f32x4*>(rp + (c * 16) * r_co);
          *reinterpret_cast<f32x4*>(yp + (c * 16) * y_co) = v;
          if (a.mask_out) {                                // wave-uniform; launcher: cout % 16 == 0, so every lane is here
            // sign bits for the next conv's relu' (CMF_F_RELU_BITS): ballot r covers column kq*4 + r of 16 channels per
            // lane group; lane (kq, cl < 4) writes the 16 channel bits of column kq*4 + cl
            const unsigned long long b0 = __ballot(v[0] > 0.f), b1 = __ballot(v[1] > 0.f), b2 = __ballot(v[2] > 0.f),
                                     b3 = __ballot(v[3] > 0.f);
            const int ccl = cl & 3;
            const unsigned long long bb = ccl == 0 ? b0 : ccl == 1 ? b1 : ccl == 2 ? b2 : b3;
            if (cl < 4) {
              const long long sample = (long long)np * a.nc + slice * 16 + kq * 4 + cl;
              unsigned char* mp = reinterpret_cast<unsigned char*>(a.mask_out) + sample * a.mask_np +
                                  (long long)gpix * (a.cout / 8) + (cog * 64 + cosub + c * 16) / 8;
              *reinterpret_cast<unsigned short*>(mp) = (unsigned short)(bb >> (kq * 16));
            }
          }
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

template <int TAPS, int COT, int PXW>
int launch(const cmf_conv_tangent_args& a, hipStream_t s, int nsub = 1) {
  const int HW = a.H * a.W;
  int tiles, tiles_x = 1;
  if (TAPS == 9) {
    tiles_x = cmf_ceil_div(a.W, 2 * PXW);
    tiles = tiles_x * cmf_ceil_div(a.H, 2);
  } else {
    tiles = cmf_ceil_div(HW, 4 * PXW);
  }
  const int nslices = a.nc / 16, ncog = cmf_ceil_div(a.cout, 64) * nsub;
  const int cin_pad = (a.cin + 7) / 8 * 8;
  const long long total = (long long)tiles * nslices * ncog * a.np;
  if (total > 0x7fffffffLL) return CMF_ERANGE;
  hipLaunchKernelGGL((conv_tangent_kernel<TAPS, COT, PXW>), dim3((unsigned)total), dim3(256), 0, s, a, tiles_x, tiles,
                     nslices, ncog, cin_pad, nsub);
  CMF_LAUNCH_CHECK();
  return 0;
}

template <int TAPS, int PXW>
int launch_cot(const cmf_conv_tangent_args& a, hipStream_t s) {
  // Small grids (the primal pass: 16 samples per column slot, 14 x 14 images = 224 items at B = 512 for 512 resident
  // workgroups): deal every 64-channel group to two workgroups of 32 channels.
  if (a.cout % 64 == 0) {
    const long long tiles = TAPS == 9 ? (long long)cmf_ceil_div(a.W, 2 * PXW) * cmf_ceil_div(a.H, 2)
                                      : (long long)cmf_ceil_div(a.H * a.W, 4 * PXW);
    const long long items = tiles * (a.nc / 16) * (a.cout / 64) * a.np;
    if (items < 160) return launch<TAPS, 1, PXW>(a, s, 4);    // tiny grids (32-sample shards): the launch is one item's
                                                                // latency -- quarter the MFMA work per item
    if (items < 1024) return launch<TAPS, 2, PXW>(a, s, 2);
  }
  const int cot = (a.cout >= 64) ? 4 : (a.cout + 15) / 16;
  switch (cot) {
    case 1: return launch<TAPS, 1, PXW>(a, s);
    case 2: return launch<TAPS, 2, PXW>(a, s);
    case 3: return launch<TAPS, 3, PXW>(a, s);
    default:
      // 2 x 16 tiles with four channel tiles per wave need 267 VGPRs (hipcc spilled 11 to scratch): there a 64-channel group is
      // always dealt to two workgroups of 32 channels (the generic-width / CIFAR path of the exact-fp32 kernel; same results)
      if constexpr (TAPS == 9 && PXW == 8) return launch<TAPS, 2, PXW>(a, s, 2);
      else return launch<TAPS, 4, PXW>(a, s);
  }
}

inline bool fits_int(long long v) { return v >= 0 && v < (1LL << 29); }   // element offsets; x4 bytes must fit 32 bits

// ------------------------------------------------------------------------------------------------------------------------------
// THIN-INPUT 3x3 tangent conv -- a coupler network's FIRST conv (networks.py:40-47: 1 - 2 input channels -> the hidden width) for all
// Jacobian columns: 18 * cin flop per output value against 4 B written, i.e. an HBM WRITE stream (6.6 GB per launch at the headline
// shape), not matrix work.  On the MFMA kernel above its K dimension is mostly padding and it wrote at 3.6 TB/s; a fill of the same bytes
// runs at 5.6.  Here: plain fp32 FMAs.  One WAVE per (sample, image row, 16-column slice, 64-channel group); the three input rows it
// needs (times the input factor, zero rows / columns outside the image) are staged once in the wave's own LDS region and read back as
// 16-byte broadcasts (4 distinct addresses per instruction); lane (cl = lane / 4, q = lane % 4) owns channels cl + 16 j (j = 0..3) x
// columns 4 q .. 4 q + 3 of every pixel, so each of its four stores per pixel is 16 B of a CONTIGUOUS KiB in the slice-major hidden layout
// (y_co = 16).  Weights: 36 * cin registers per lane, read from cmf_pack_weight's image.  No barriers (waves are independent).
template <int CIN, int NJ>
__global__ __launch_bounds__(256) void conv_tangent_thin_kernel(cmf_conv_tangent_args a, int nslices, int ncog, int cin_pad,
                                                                 long long total_waves) {
  // NJ = 16-channel tiles per wave: 4 (a whole 64-channel group) for one input channel, 2 for two -- the weights are
  // 9 * CIN * NJ registers per lane (twice that as hipcc keeps them: pairs for v_pk_fma_f32); `cog` counts groups of 16 NJ channels.
  // Three input channels (the first CIFAR level) were built and measured too: VALU-bound at 3.2 TB/s, no faster than the MFMA kernel.
  extern __shared__ __attribute__((aligned(16))) float thin_rows[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gw = (long long)blockIdx.x * 4 + wave;
  if (gw >= total_waves) return;                                   // whole waves only: nothing below synchronises
  long long t = gw;
  const int slice = (int)(t % nslices);
  t /= nslices;
  const int cog = (int)(t % ncog);
  t /= ncog;
  const int y = (int)(t % a.H), np = (int)(t / a.H);
  const int q = lane & 3, cl = lane >> 2, W = a.W, WP = W + 2;
  float* rows = thin_rows + (long long)wave * CIN * 3 * WP * 16;   // [ci][r][x + 1][16 columns]
  // stage: piece = (ci, r, x, column quad); out-of-image rows and the two border columns are zeros
  const float* xb = a.x + (long long)np * a.x_np + (long long)slice * (a.x_sl ? a.x_sl : 16);
  const bool raw = a.fmode == CMF_F_RAW;
  const float* fb = raw ? a.f + (long long)np * a.f_np : nullptr;
#pragma unroll 2
  for (int i = lane; i < CIN * 3 * WP * 4; i += 64) {
    const int qq = i & 3, xx = (i >> 2) % WP - 1, rr = ((i >> 2) / WP) % 3, ci = (i >> 2) / (3 * WP);
    const int yy = y - 1 + rr;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (yy >= 0 && yy < a.H && xx >= 0 && xx < W) {
      const long long px = (long long)yy * W + xx;
      v = *reinterpret_cast<const f32x4*>(xb + ci * a.x_ci + px * a.x_px + qq * 4);
      if (raw) v *= fb[ci * a.f_ci + px * a.f_px];
    }
    *reinterpret_cast<f32x4*>(rows + ((ci * 3 + rr) * WP + xx + 1) * 16 + qq * 4) = v;
  }
  const int co0 = cog * NJ * 16 + cl;                              // this lane's first output channel (launcher: cout % 64 == 0)
  float w[NJ][CIN][9];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int co = co0 + j * 16;
        w[j][ci][tap] = a.w[(((long long)(co >> 6) * 9 + tap) * cin_pad + ci) * 64 + (co & 63)];
      }
  float* yb = a.y + (long long)np * a.y_np + (long long)slice * (a.y_sl ? a.y_sl : 16) + (long long)co0 * a.y_co + q * 4;
  __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the wave's own LDS writes (no barrier needed)
  __builtin_amdgcn_wave_barrier();
#pragma unroll 1
  for (int x = 0; x < W; ++x) {
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rows + ((ci * 3 + r) * WP + x + c) * 16 + q * 4);
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[j] += w[j][ci][r * 3 + c] * v;
        }
    float* yp = yb + ((long long)y * W + x) * a.y_px;
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(yp + (long long)j * 16 * a.y_co) = acc[j];
  }
}

// shapes the thin kernel takes: everything else stays on the MFMA kernel
inline bool thin_ok(const cmf_conv_tangent_args& a) {
  return a.taps == 9 && a.cin <= 2 && a.cout % 64 == 0 && !a.fo && !a.r && !a.bias && !a.mask_out &&
         (a.fmode == CMF_F_NONE || (a.fmode == CMF_F_RAW && a.f_group <= 1)) &&
         (long long)4 * 3 * (a.W + 2) * a.cin * 64 <= 64 * 1024;      // the four waves' row images: within the default dynamic-LDS limit
}

int launch_thin(const cmf_conv_tangent_args& a, hipStream_t s) {
  const int nj = a.cin == 1 ? 4 : 2;
  const int nslices = a.nc / 16, ncog = a.cout / (16 * nj), cin_pad = (a.cin + 7) / 8 * 8;
  const long long waves = (long long)a.np * a.H * ncog * nslices;
  if ((waves + 3) / 4 > 0x7fffffffLL) return CMF_ERANGE;
  const int lds = 4 * 3 * (a.W + 2) * a.cin * 64;
  const dim3 grid((unsigned)((waves + 3) / 4));
  if (a.cin == 1) hipLaunchKernelGGL((conv_tangent_thin_kernel<1, 4>), grid, dim3(256), lds, s, a, nslices, ncog, cin_pad, waves);
  else hipLaunchKernelGGL((conv_tangent_thin_kernel<2, 2>), grid, dim3(256), lds, s, a, nslices, ncog, cin_pad, waves);
  CMF_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int cmf_conv_tangent(const cmf_conv_tangent_args* ap, void* stream) {
  if (!ap) return CMF_EINVAL;
  const cmf_conv_tangent_args& a = *ap;
  if (!a.x || !a.w || !a.y || a.np <= 0 || a.cin <= 0 || a.cout <= 0 || a.H <= 0 || a.W <= 0) return CMF_EINVAL;
  if (a.taps != 1 && a.taps != 9) return CMF_EINVAL;
  if (a.nc <= 0 || a.nc % 16 || a.live) return CMF_EINVAL;     // checkerboard output: cmf_conv_tangent_bf16x3 only
  if (a.fmode < CMF_F_NONE || a.fmode > CMF_F_SELF_RELU) return CMF_EINVAL;
  if (a.fmode != CMF_F_NONE && a.fmode != CMF_F_SELF_RELU && !a.f) return CMF_EINVAL;
  // 16-byte vector loads of the column slices
  if ((a.x_np | a.x_ci | a.x_px | a.x_sl | a.y_sl | a.r_sl) % 4 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return CMF_EINVAL;
  if ((a.y_np | a.y_co | a.y_px) % 4 || ((uintptr_t)a.y % 16)) return CMF_EINVAL;            // 16-byte stores
  if (a.r && ((a.r_np | a.r_co | a.r_px) % 4 || ((uintptr_t)a.r % 16))) return CMF_EINVAL;    // 16-byte residual loads
  if (a.fo && (a.fomode < CMF_F_RELU || a.fomode > CMF_F_SELF_RELU)) return CMF_EINVAL;
  if (a.fo && a.fomode == CMF_F_SELF_RELU && ((a.fo_np | a.fo_co | a.fo_px) % 4 || (uintptr_t)a.fo % 16)) return CMF_EINVAL;
  if (a.mask_out && (a.cout % 16 || a.mask_np < (long long)a.H * a.W * (a.cout / 8))) return CMF_EINVAL;
  const long long HW = (long long)a.H * a.W;
  // per-sample offsets are held in 32-bit registers
  if (!fits_int((a.cin + 8) * a.x_ci + HW * a.x_px + a.nc) || !fits_int((a.cin + 8) * a.f_ci + HW * a.f_px) ||
      !fits_int((a.cout + 64) * a.y_co + HW * a.y_px + a.nc) || (a.r && !fits_int((a.cout + 64) * a.r_co + HW * a.r_px + a.nc)) ||
      HW > (1 << 24))
    return CMF_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  if (thin_ok(a)) return launch_thin(a, s);                       // a coupler's first conv: an HBM write stream, VALU kernel
  const bool seven = (a.taps == 9) ? (a.W % 14 == 0) : (HW % 28 == 0 && HW % 32 != 0);
  if (a.taps == 9 && !seven && a.W % 8 == 0 && a.cout % 64 == 0) {
    // tiny grids on 8-multiple widths (the primal pass of a 32-sample CIFAR shard: 32 tile items): 2 x 8 tiles instead of 2 x 16 --
    // twice the workgroups, half the serial MFMA work in each: the launch is one item's latency
    const long long items8 = (long long)cmf_ceil_div(a.W, 16) * cmf_ceil_div(a.H, 2) * (a.nc / 16) * (a.cout / 64) * a.np;
    if (items8 * 4 <= 128) return launch_cot<9, 4>(a, s);
  }
  if (a.taps == 9) return seven ? launch_cot<9, 7>(a, s) : launch_cot<9, 8>(a, s);
  return seven ? launch_cot<1, 7>(a, s) : launch_cot<1, 8>(a, s);
}
