// Fused J^T J Gram (fp32 MFMA) + Cholesky + log-det + g_ij / g_kk L1 terms for gfx950 (MI355X).
//
// Replaces, per sample, the reference's  jac = stack(cols, 2); bmm(jac^T, jac)
// (cmf/models/components/densities/non_square.py:307-308),  torch.linalg.cholesky + 2*sum(log diag)
// (:280-294) and the metric regularisers  sum_{i!=j}|G_ij| / sum_k|G_kk|  (:87-100).
//
// One 256-thread workgroup per sample.  The J panel T(b, r, 0:nc) is streamed once from HBM in
// 32-row slabs through LDS (register prefetch of the next slab under the MFMAs of the current one);
// G = J^T J is accumulated as NT x NT tiles of v_mfma_f32_16x16x4_f32 (A and B operands are the SAME
// slab: A[i][k] = J[k][i], B[k][j] = J[k][j], both conflict-free ds_read_b32 of 16 consecutive
// columns), parked in LDS, factorised in place by a right-looking Cholesky, and reduced with
// wavefront shuffles.  Algorithmic traffic: (n_rows*nc + d*d + 3) * 4 bytes per sample.
//
// Whole-batch jitter retry (non_square.py:284-288) without a host round trip: attempt 0 raises
// fail[0] when any sample hits a non-positive pivot; cmf_cholesky_retry(a) is enqueued unconditionally
// and exits at once unless fail[a-1] is set.
#include "common.h"
#include <type_traits>

namespace {

constexpr int KC = 32;  // J rows per LDS slab


// Log-det by symmetric elimination on the lower triangle of the leading d x d block of G (row stride ldg), whole
// workgroup (256 threads as a 16 x 16 grid, no index divisions).  Only the pivots are needed: p_k = G_kk after the
// first k eliminations equals L_kk^2 of the Cholesky factor the reference computes (non_square.py:282,293-294), so
// logdet = sum_k log p_k and "pivot <= 0 or not finite" is exactly torch.linalg.cholesky's failure condition.  No
// square roots, no column scaling, ONE barrier per step:  G_ij -= G_ik * G_jk / p_k  for k < j <= i.
// Returns 0 or (k+1) for a bad pivot at column k.
__device__ int block_cholesky(float* G, int ldg, int d, float* logdet) {
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  float ld = 0.f;
  int info = 0;
  for (int k = 0; k < d; ++k) {
    const float piv = G[k * ldg + k];            // uniform: every thread reads the same LDS word
    if (!(piv > 0.f) || !(piv < 3.0e38f)) {
      info = k + 1;
      break;
    }
    ld += logf(piv);
    for (int i = k + 1 + ti; i < d; i += 16) {
      const float gik = G[i * ldg + k] / piv;      // IEEE division: an exactly singular matrix must give an exact 0 pivot
      for (int j = k + 1 + tj; j <= i; j += 16) G[i * ldg + j] -= gik * G[j * ldg + k];
    }
    __syncthreads();                             // column k is read by this step only; row/col k+1 is final after it
  }
  *logdet = ld;
  return info;
}

// The same elimination with the matrix in REGISTERS: thread (ti, tj) of the 16 x 16 grid keeps the NT x NT elements
// (ti + 16a, tj + 16c).  Per step only column k goes through LDS (double-buffered: one barrier per step): its owners
// publish it, everyone reads the pivot, NT row factors and NT column values and updates its registers.  For j <= i the
// arithmetic is that of block_cholesky (G_ij -= (G_ik / p_k) G_jk); the mirror half is computed but never read.
typedef unsigned int u32x4 __attribute__((vector_size(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));   // the type raw_buffer_load_b128 returns

template <int NT>
__device__ int block_cholesky_reg(const float* G, int ldg, int d, float* colbuf, float* logdet) {
  constexpr int NC = NT * 16;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  float g[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int c = 0; c < NT; ++c) g[a][c] = G[(ti + 16 * a) * ldg + tj + 16 * c];
  float ld = 0.f;
  int info = 0;
#pragma unroll
  for (int c0 = 0; c0 < NT; ++c0) {
    if (info) break;
    for (int kk = 0; kk < 16; ++kk) {
      const int k = c0 * 16 + kk;
      if (k >= d) break;
      float* cb = colbuf + (k & 1) * NC;
      if (tj == kk) {
#pragma unroll
        for (int a = 0; a < NT; ++a) cb[ti + 16 * a] = g[a][c0];
      }
      __syncthreads();
      const float piv = cb[k];
      if (!(piv > 0.f) || !(piv < 3.0e38f)) {
        info = k + 1;
        break;
      }
      ld += logf(piv);
      float l[NT], r[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) l[a] = cb[ti + 16 * a] / piv;
#pragma unroll
      for (int c = 0; c < NT; ++c) r[c] = cb[tj + 16 * c];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) g[a][c] -= l[a] * r[c];
    }
  }
  *logdet = ld;
  return info;
}

__device__ void report(int b, int info, float logdet, float* logdet_out, int* info_out, int* fail_slot) {
  if (threadIdx.x == 0) {
    info_out[b] = info;
    logdet_out[b] = info ? __builtin_nanf("") : logdet;
    if (info) atomicOr(fail_slot, 1);
  }
}

template <int NT>
__global__ __launch_bounds__(256) void gram_chol_kernel(const float* __restrict__ t, long long t_b, long long t_r,
                                                         int n_rows, int d, float* __restrict__ jtj,
                                                         float* __restrict__ logdet, float* __restrict__ l1_off,
                                                         float* __restrict__ l1_diag, int* __restrict__ info,
                                                         int* __restrict__ fail) {
  constexpr int NC = NT * 16;
  constexpr int LDJ = (NC % 32 == 0) ? NC + 16 : NC;
  constexpr int LDG = NC + 1;
  constexpr int NI = (NT + 3) / 4;                 // row tiles per wave
  constexpr int ITEMS = KC * NC / 4;               // float4 per slab
  constexpr int NIT = (ITEMS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Js = smem;                                // [KC][LDJ]
  float* G = smem + KC * LDJ;                      // [NC][LDG]
  __shared__ float red[16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;
  const int b = blockIdx.x;
  const float* tb = t + (long long)b * t_b;

  f32x4 acc[NI][NT];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 pr[NIT];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + 256 * it;
      i = i < ITEMS ? i : ITEMS - 1;
      const int row = i / (NC / 4), c4 = i % (NC / 4);
      const bool ok = (k0 + row) < n_rows;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (ok ? (long long)(k0 + row) * t_r : 0) + c4 * 4);
      pr[it] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 256 * it;
      if (i < ITEMS) *reinterpret_cast<f32x4*>(Js + (i / (NC / 4)) * LDJ + (i % (NC / 4)) * 4) = pr[it];
    }
  };

  prefetch(0);
  for (int k0 = 0; k0 < n_rows; k0 += KC) {
    commit();
    __syncthreads();
    if (k0 + KC < n_rows) prefetch(k0 + KC);
#pragma unroll
    for (int kg = 0; kg < KC / 4; ++kg) {
      const float* row = Js + (kg * 4 + kq) * LDJ + cl;
      float bv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = row[j * 16];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int it = wave + 4 * i;
        if (it < NT) {                             // wave-uniform
          const float av = row[it * 16];
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // park G in LDS: D[row = kq*4 + r][col = cl]
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int it = wave + 4 * i;
    if (it < NT) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) G[(it * 16 + kq * 4 + r) * LDG + j * 16 + cl] = acc[i][j][r];
    }
  }
  __syncthreads();

  // J^T J out + L1 terms (before the factorisation overwrites the lower triangle)
  float so = 0.f, sd = 0.f;
  float* gout = jtj + (long long)b * d * d;
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx % d;
    const float v = G[i * LDG + j];
    gout[idx] = v;
    if (i == j) sd += fabsf(v); else so += fabsf(v);
  }
  so = block_sum(so, red);
  sd = block_sum(sd, red);
  if (tid == 0) {
    l1_off[b] = so;
    l1_diag[b] = sd;
  }
  float ld;
  const int inf = block_cholesky(G, LDG, d, &ld);
  report(b, inf, ld, logdet, info, fail + 0);
}

__global__ __launch_bounds__(256) void chol_retry_kernel(float* __restrict__ jtj, int d, int attempt, float eps,
                                                          float* __restrict__ logdet, float* __restrict__ l1_diag,
                                                          int* __restrict__ info, int* __restrict__ fail) {
  if (fail[attempt - 1] == 0) return;              // previous attempt succeeded for the whole batch
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[16];
  float* G = smem;
  const int ldg = d + 1, tid = threadIdx.x, b = blockIdx.x;
  float* gb = jtj + (long long)b * d * d;
  float sd = 0.f;
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx % d;
    float v = gb[idx];
    if (i == j) {                                   // jitter EVERY sample: non_square.py:286
      v += eps;
      gb[idx] = v;
      sd += fabsf(v);
    }
    G[i * ldg + j] = v;
  }
  sd = block_sum(sd, red);
  if (tid == 0) l1_diag[b] = sd;
  __syncthreads();
  float ld;
  const int inf = block_cholesky(G, ldg, d, &ld);
  report(b, inf, ld, logdet, info, fail + attempt);
}

// Gram without LDS staging or barriers, for NC = 16*NT with NT a multiple of 4 (d <= 64: NT = 4, d <= 128: NT = 8).
// A lane of lane group kq loads NT consecutive columns of row k0 + kq (NT/4 float4: one 4-row group of the J panel is
// 16*NT*4*4 bytes, fully coalesced) and uses component t as its A and B operand element of column tile t: tile t then
// holds the columns {NT*i + t}, a permutation that is the same on both sides of J^T J and is undone when G is written.
// NT = 8: wave w accumulates row tiles it = w, w+4 against all column tiles over ALL rows (the four waves re-read the
// same lines: L1 hits).  NT = 4: all 16 tiles fit one wave's registers, so the ROWS are dealt to the four waves instead
// (4x the bytes in flight per CU -- with shared rows the kernel was load-latency-bound at ~1 TB/s) and the four partial
// Grams are parked in four LDS planes and summed on the way out (ds_add_f32 costs ~64 cycles per wave instruction
// here: measured +44 us).  Loads run PF groups ahead in registers.
template <int NT>
__global__ __launch_bounds__(256) void gram_chol_direct_kernel(const float* __restrict__ t, long long t_b, long long t_r,
                                                                int n_rows, int d, float* __restrict__ jtj,
                                                                float* __restrict__ logdet, float* __restrict__ l1_off,
                                                                float* __restrict__ l1_diag, int* __restrict__ info,
                                                                int* __restrict__ fail) {
  static_assert(NT % 4 == 0, "whole float4 per lane");
  constexpr bool KSPLIT = NT == 4;                 // deal 4-row groups to waves; every wave owns all NT x NT tiles
  constexpr int NC = NT * 16, LDG = NC + 1, NI = KSPLIT ? NT : NT / 4, NV = NT / 4, PF = 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* G = smem;                                 // [NC][LDG]
  __shared__ float red[16];
  __shared__ float colbuf[2 * NC];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;
  const int b = blockIdx.x;
  const float* tb = t + (long long)b * t_b + cl * NT;           // this lane's NT columns

  f32x4 acc[NI][NT];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ngroups_all = (n_rows + 3) >> 2;
  const int ngroups = KSPLIT ? (ngroups_all + 3 - wave) / 4 : ngroups_all;      // this wave's share
  // Rows come through a bounds-checked buffer descriptor over this sample's panel: a row past n_rows reads as zeros, so
  // there is no select after the load (a select made hipcc wait for all PF loads at the top of every iteration).
  // The per-group advance lives in the DESCRIPTOR (scalar base += group, records -= group) and the lane offset VGPR is
  // loop-invariant: a per-step VGPR offset was allocated inside the ring slot being refilled, and rewriting it for
  // the next step then waited for that load (vmcnt(0) every step).
  const char* panel = reinterpret_cast<const char*>(t + (long long)b * t_b);
  const int panel_bytes = (int)((long long)n_rows * t_r * 4);
  const int row_bytes = (int)t_r * 4;
  const int group_bytes = (KSPLIT ? 16 : 4) * row_bytes;
  const int voff = ((KSPLIT ? wave : 0) * 4 + kq) * row_bytes + cl * NT * 4;
  int goff = 0;                                    // wave-uniform byte offset of the next group to fetch
  u32x4 ring[PF][NV];
  auto load = [&](int slot) {
    const int left = panel_bytes - goff;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(panel + goff), 0, left > 0 ? left : 0, 0x00020000);
#pragma unroll
    for (int v = 0; v < NV; ++v) ring[slot][v] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16 * v, 0, 0);
    goff += group_bytes;
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    load(s);
    __builtin_amdgcn_sched_barrier(0);             // issue in slot order, or the loop-header wait degrades to vmcnt(0)
  }
  // W = the wave's first row tile as a compile-time constant: a run-time comp[wave + 4i] became v_cndmask temporaries
  // that were again allocated inside in-flight ring slots.
  auto run = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    for (int g0 = 0; g0 < ngroups; g0 += PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        float comp[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const unsigned u = ring[s][j >> 2][j & 3];     // via a scalar: bit_cast of a vector ELEMENT lvalue reads lane 0
          comp[j] = __builtin_bit_cast(float, u);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp[KSPLIT ? i : W + 4 * i], comp[j], acc[i][j], 0, 0, 0);
        load(s);                                   // refill this slot for PF groups ahead
        __builtin_amdgcn_sched_barrier(0);         // keep consume(s) -> refill(s) order: waits become vmcnt(PF-1)
      }
    }
  };
  if (KSPLIT || wave == 0) run(std::integral_constant<int, 0>{});
  else if (wave == 1) run(std::integral_constant<int, 1>{});
  else if (wave == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});

  // park G in LDS, undoing the column permutation: tile element (row kq*4 + r, col cl) of tile (it, jt) is
  // G[NT*(kq*4 + r) + it][NT*cl + jt]
  constexpr int PLANE = NC * LDG;
  float* Gw = KSPLIT ? G + wave * PLANE : G;       // KSPLIT: one partial-Gram plane per wave
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int it = KSPLIT ? i : wave + 4 * i;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Gw[(NT * (kq * 4 + r) + it) * LDG + NT * cl + j] = acc[i][j][r];
  }
  __syncthreads();

  float so = 0.f, sd = 0.f;
  float* gout = jtj + (long long)b * d * d;
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx % d;
    float v = G[i * LDG + j];
    if (KSPLIT) {
      v = (v + G[PLANE + i * LDG + j]) + (G[2 * PLANE + i * LDG + j] + G[3 * PLANE + i * LDG + j]);
      G[i * LDG + j] = v;
    }
    gout[idx] = v;
    if (i == j) sd += fabsf(v); else so += fabsf(v);
  }
  so = block_sum(so, red);
  sd = block_sum(sd, red);
  if (tid == 0) {
    l1_off[b] = so;
    l1_diag[b] = sd;
  }
  float ld;
  __syncthreads();                                 // plane 0 holds the summed Gram
  const int inf = block_cholesky_reg<NT>(G, LDG, d, colbuf, &ld);
  report(b, inf, ld, logdet, info, fail + 0);
}


#ifdef CMF_DBG_GSTAMP
// diagnostic build only (tools/build_dbg.sh GSTAMP): phase time stamps of every workgroup of the d <= 64 kernel
__device__ unsigned long long cmf_dbg_gram_stamps[4096][4];
extern "C" int cmf_debug_read_gram_stamps(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cmf_dbg_gram_stamps), sizeof(cmf_dbg_gram_stamps));
}
#define GSTAMP(k)                                                                    \
  do {                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 4096) {                                      \
      unsigned long long t_;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      cmf_dbg_gram_stamps[blockIdx.x][k] = t_;                                        \
    }                                                                                 \
  } while (0)
#else
#define GSTAMP(k) do {} while (0)
#endif

// d <= 64 (NC = 64): Gram, cross-wave reduction, J^T J / L1 output and the pivot-only elimination.
// Phase stamps of the round-1 kernel (tools/bench_gram.py with the GSTAMP build; B = 512, D = 784, d = 64, s_memtime ~ 2.1 GHz):
// Gram 56 k cycles for the two co-resident workgroups of a CU = 90 % of the MFMA issue rate, but reduction + output 13 k and
// the elimination 59 k cycles = 920 per step: its ~100 VALU instructions per step (four IEEE divisions, logf, address
// arithmetic of nine LDS reads) were the critical path, not the barrier.  This kernel keeps the Gram loop and rebuilds the rest:
//   Gram      rows dealt to the four waves in 4-row groups (K split), every wave accumulates all 16 tiles (64 accumulator
//             registers); loads run PF groups ahead in a register ring through a bounds-checked buffer descriptor.  Lane
//             (kq, cl) feeds component t of its float4 as A and B operand element of column tile t, so accumulator tile
//             (i, j), register r holds G[16 kq + 4 r + i][4 cl + j].
//   reduce    every wave parks its partial Gram in its own LDS plane in ROW-MAJOR order (row stride 68 floats: a lane's four
//             tiles j = 0..3 of one (i, r) are one ds_write_b128, the 16 lanes of a kq group write one contiguous row).
//             Thread (tr, tc) = (tid / 16, tid % 16) then sums the 4 x 4 block rows 4 tr.., columns 4 tc.. over the four
//             planes (16 conflict-free ds_read_b128) and owns it in registers: J^T J leaves as four 16-byte stores per
//             thread, the L1 terms are register sums.
//   Cholesky  symmetric pivot-only elimination on those registers, one column per barrier, two ds_read_b128 per thread and
//             step (see the loop).  Pivots are kept in LDS and their logs are summed after the loop, off the critical path.
// workgroup barrier for LDS traffic only (``__syncthreads`` also drains vmcnt: outstanding global stores)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int PF>
__global__ __launch_bounds__(256) void gram_chol64_kernel(const float* __restrict__ t, long long t_b, long long t_r,
                                                           int n_rows, int ncm, int d, float* __restrict__ jtj,
                                                           float* __restrict__ logdet, float* __restrict__ l1_off,
                                                           float* __restrict__ l1_diag, int* __restrict__ info,
                                                           int* __restrict__ fail) {
  constexpr int NT = 4, NC = 64, LDP = 68, PLANE = NC * LDP;
  extern __shared__ __attribute__((aligned(16))) float smem[];      // 4 planes x 64 rows x 68 floats = 69.6 KB
  __shared__ __attribute__((aligned(16))) float colB[2][NC];        // column k, double-buffered
  __shared__ float pivs[NC];
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;
  const int b = blockIdx.x;
  GSTAMP(0);

  f32x4 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ngroups_all = (n_rows + 3) >> 2;
  const int ngroups = (ngroups_all + 3 - wave) / 4;                 // this wave's 4-row groups: wave, wave + 4, ...
  const char* panel = reinterpret_cast<const char*>(t + (long long)b * t_b);
  const int panel_bytes = (int)((long long)n_rows * t_r * 4);
  const int row_bytes = (int)t_r * 4;
  const int group_bytes = 16 * row_bytes;
  // Narrower panels (ncm = 16 / 32 / 48 columns in memory, d <= ncm) run on the same 64-column machinery: the lanes whose four
  // columns do not exist read through an offset past every descriptor's range (zeros, no traffic), so their rows / columns
  // of the Gram matrix are exact zeros like the columns >= d; the elimination only visits the first d.
  const int voff = cl * NT < ncm ? (wave * 4 + kq) * row_bytes + cl * NT * 4 : 0x7ffffff0;
  int goff = 0;
  u32x4 ring[PF];
  auto load = [&](int slot) {
    const int left = panel_bytes - goff;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(panel + goff), 0, left > 0 ? left : 0, 0x00020000);
    ring[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    goff += group_bytes;
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    load(s);
    __builtin_amdgcn_sched_barrier(0);
  }
  auto consume = [&](int s) {
    float comp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const unsigned u = ring[s][j];
      comp[j] = __builtin_bit_cast(float, u);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp[i], comp[j], acc[i][j], 0, 0, 0);
  };
  int g0 = 0;
  for (; g0 + PF <= ngroups; g0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      consume(s);
      load(s);                                      // past the panel's end the descriptor returns zeros without traffic
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the remaining ngroups % PF groups, exactly: an all-zero group still costs its 16 MFMAs (49 groups per wave at D = 784:
  // rounding up to a multiple of PF = 8 was 14 % more matrix work)
  const int rem = ngroups - g0;
#pragma unroll
  for (int s = 0; s < PF - 1; ++s)
    if (s < rem) consume(s);                        // uniform
  GSTAMP(1);

  // park the partial Gram of this wave row-major in its plane: G[16 kq + 4 r + i][4 cl + (0..3)]
  float* plane = smem + wave * PLANE;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(plane + (16 * kq + 4 * r + i) * LDP + 4 * cl) =
          f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
  __syncthreads();
  const int tr = tid >> 4, tc = tid & 15;           // this thread: rows 4 tr .. 4 tr + 3 x columns 4 tc .. 4 tc + 3
  float g[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float* src = smem + (4 * tr + i) * LDP + 4 * tc;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(src), p1 = *reinterpret_cast<const f32x4*>(src + PLANE);
    const f32x4 p2 = *reinterpret_cast<const f32x4*>(src + 2 * PLANE), p3 = *reinterpret_cast<const f32x4*>(src + 3 * PLANE);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[i][e] = (p0[e] + p1[e]) + (p2[e] + p3[e]);
  }

  // J^T J out and the L1 terms, from registers
  float so = 0.f, sd = 0.f;
  float* gout = jtj + (long long)b * d * d;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * tr + i;
    if (row < d) {
      if (d == NC) {
        *reinterpret_cast<f32x4*>(gout + row * NC + 4 * tc) = f32x4{g[i][0], g[i][1], g[i][2], g[i][3]};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (4 * tc + c < d) gout[row * d + 4 * tc + c] = g[i][c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float a = fabsf(g[i][c]);             // columns >= d are exact zeros (zero J columns)
        if (row == 4 * tc + c) sd += a; else so += a;
      }
    }
  }
  GSTAMP(2);

  // Pivot-only symmetric elimination in registers, G_ij -= (G_ik / p_k) G_jk.  What bounds a step is LDS traffic and VALU
  // issue, not the barrier (GSTAMP builds: a row x 16-column ownership read 5 b128 per thread and column = 20 KB per
  // workgroup, 600 cycles per column with two workgroups per CU -- and pairing two columns per barrier changed nothing):
  // with 4 x 4 blocks a thread needs its four row values and four column values of column k (two ds_read_b128) plus the
  // pivot.  The row factors are a_i * (1 / p_k) with ONE IEEE division per step (exact for the power-of-two matrices of the
  // retry fixture; on an exactly singular general matrix the sign of the rounding-level pivot is as arbitrary as LAPACK's).
  // The body is branch-free apart from the uniform k < d guard: a `break` on a bad pivot kept hipcc from unrolling the column
  // slots (the published register then went through s_set_gpr_idx and the block was copied twice per step); a bad pivot
  // is remembered and the remaining steps run on garbage.
  int inf = 0;
#pragma unroll 1
  for (int m = 0; m < 16; ++m) {
    if (4 * m >= d) break;                          // uniform
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = 4 * m + c;
      if (k < d) {                                  // uniform
        float* cb = colB[k & 1];
        if (tc == m) {
          *reinterpret_cast<f32x4*>(cb + 4 * tr) = f32x4{g[0][c], g[1][c], g[2][c], g[3][c]};
          if (tr == m) pivs[k] = g[c][c];
        }
        lds_barrier();                              // LDS visibility only: no vmcnt(0) (the J^T J stores are in flight)
        const float piv = cb[k];
        const f32x4 av = *reinterpret_cast<const f32x4*>(cb + 4 * tr);
        const f32x4 rv = *reinterpret_cast<const f32x4*>(cb + 4 * tc);
        if ((!(piv > 0.f) || !(piv < 3.0e38f)) && !inf) inf = k + 1;
        float rp = __builtin_amdgcn_rcpf(piv);      // 1 ulp, then one Newton step: a correctly rounded reciprocal for all
        rp = __builtin_fmaf(__builtin_fmaf(-piv, rp, 1.0f), rp, rp);   // but a few operands; exact for powers of two
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float l = av[i] * rp;
#pragma unroll
          for (int e = 0; e < 4; ++e) g[i][e] -= l * rv[e];
        }
      }
    }
  }
  __syncthreads();
  float ld = (!inf && tid < d) ? logf(pivs[tid]) : 0.f;
  ld = wave_sum(ld);
  so = wave_sum(so);
  sd = wave_sum(sd);
  if (lane == 0) {
    red[wave] = ld;
    red[4 + wave] = so;
    red[8 + wave] = sd;
  }
  __syncthreads();
  ld = (red[0] + red[1]) + (red[2] + red[3]);
  GSTAMP(3);
  if (tid == 0) {
    l1_off[b] = (red[4] + red[5]) + (red[6] + red[7]);
    l1_diag[b] = (red[8] + red[9]) + (red[10] + red[11]);
  }
  report(b, inf, ld, logdet, info, fail + 0);
}

template <int PF>
int launch_gram64(const float* t, long long t_b, long long t_r, int n_rows, int ncm, int d, int B, float* jtj, float* logdet,
                  float* l1_off, float* l1_diag, int* info, int* fail, hipStream_t s) {
  constexpr int lds = 4 * 64 * 68 * 4;
  auto k = gram_chol64_kernel<PF>;
  if (hipError_t e = cmf_set_dynamic_lds((const void*)k, lds); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k, dim3(B), dim3(256), lds, s, t, t_b, t_r, n_rows, ncm, d, jtj, logdet, l1_off, l1_diag, info, fail);
  CMF_LAUNCH_CHECK();
  return 0;
}

template <int NT>
int launch_gram_direct(const float* t, long long t_b, long long t_r, int n_rows, int d, int B, float* jtj, float* logdet,
                       float* l1_off, float* l1_diag, int* info, int* fail, hipStream_t s) {
  constexpr int NC = NT * 16;
  const size_t lds = (size_t)(NC * (NC + 1)) * sizeof(float) * (NT == 4 ? 4 : 1);
  auto k = gram_chol_direct_kernel<NT>;
  if (lds > 48 * 1024) {
    hipError_t e = cmf_set_dynamic_lds((const void*)k, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(B), dim3(256), lds, s, t, t_b, t_r, n_rows, d, jtj, logdet, l1_off, l1_diag, info, fail);
  CMF_LAUNCH_CHECK();
  return 0;
}

__global__ void zero_flags_kernel(int* __restrict__ fail) {
  if (threadIdx.x < 8) fail[threadIdx.x] = 0;
}

template <int NT>
int launch_gram(const float* t, long long t_b, long long t_r, int n_rows, int d, int B, float* jtj, float* logdet,
                float* l1_off, float* l1_diag, int* info, int* fail, hipStream_t s) {
  constexpr int NC = NT * 16;
  constexpr int LDJ = (NC % 32 == 0) ? NC + 16 : NC;
  const size_t lds = (size_t)(KC * LDJ + NC * (NC + 1)) * sizeof(float);
  auto k = gram_chol_kernel<NT>;
  if (lds > 48 * 1024) {
    hipError_t e = cmf_set_dynamic_lds((const void*)k, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(B), dim3(256), lds, s, t, t_b, t_r, n_rows, d, jtj, logdet, l1_off, l1_diag, info, fail);
  CMF_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int cmf_gram_cholesky(const float* t, long long t_b, long long t_r, int n_rows, int nc, int d, int B,
                                 float* jtj, float* logdet, float* l1_off, float* l1_diag, int* info, int* fail,
                                 void* stream) {
  if (!t || !jtj || !logdet || !l1_off || !l1_diag || !info || !fail) return CMF_EINVAL;
  if (n_rows <= 0 || B <= 0 || d <= 0 || d > nc || nc % 16 || nc > 128) return CMF_EINVAL;
  if ((t_b | t_r) % 4 || (uintptr_t)t % 16) return CMF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // A kernel node, not hipMemsetAsync: a captured 32-byte memset node replayed with garbage from the second
  // hipGraphLaunch on (ROCm 7.2, observed as pointer-like values in the flags), which silently armed every retry.
  hipLaunchKernelGGL(zero_flags_kernel, dim3(1), dim3(64), 0, s, fail);
  CMF_LAUNCH_CHECK();
#ifdef CMF_DBG_GRAMOLD
  if (nc == 64) return launch_gram_direct<4>(t, t_b, t_r, n_rows, d, B, jtj, logdet, l1_off, l1_diag, info, fail, s);
#endif
#ifndef CMF_DBG_GRAMPF
#define CMF_DBG_GRAMPF 8                            // measured: 8 groups ahead beat 12 / 16 / 24 by 3 - 12 us at B = 512
#endif
  // nc <= 64: the d <= 64 kernel (narrower panels through lane masking).  The panel of one sample must fit the 32-bit byte
  // offsets of its buffer descriptor (feature-major tensors of large batches do not: they keep the LDS-staged kernel)
#ifndef CMF_DBG_GRAM64_MIN_NC
#define CMF_DBG_GRAM64_MIN_NC 32                    // measured (B = 512, D = 784): nc 48: 75.6 -> 40.4 us, nc 32: 51.4 -> 38.7 us;
                                                    // nc 16 (tabular, B = 4096): the small-LDS staged kernel wins, 19.5 vs 51.9 us
#endif
  if (nc >= CMF_DBG_GRAM64_MIN_NC && nc <= 64 && (long long)n_rows * t_r * 4 < 0x7fffff00LL)
    return launch_gram64<CMF_DBG_GRAMPF>(t, t_b, t_r, n_rows, nc, d, B, jtj, logdet, l1_off, l1_diag, info, fail, s);
  if (nc == 128) return launch_gram_direct<8>(t, t_b, t_r, n_rows, d, B, jtj, logdet, l1_off, l1_diag, info, fail, s);
#define CMF_GRAM_CASE(N) \
  case N: return launch_gram<N>(t, t_b, t_r, n_rows, d, B, jtj, logdet, l1_off, l1_diag, info, fail, s);
  switch (nc / 16) {
    CMF_GRAM_CASE(1) CMF_GRAM_CASE(2) CMF_GRAM_CASE(3) CMF_GRAM_CASE(4)
    CMF_GRAM_CASE(5) CMF_GRAM_CASE(6) CMF_GRAM_CASE(7) CMF_GRAM_CASE(8)
  }
#undef CMF_GRAM_CASE
  return CMF_EINVAL;
}

extern "C" int cmf_cholesky_retry(float* jtj, int d, int B, int attempt, float eps0, float* logdet, float* l1_diag,
                                  int* info, int* fail, void* stream) {
  if (!jtj || !logdet || !l1_diag || !info || !fail || d <= 0 || d > 128 || B <= 0 || attempt < 1 || attempt > 7)
    return CMF_EINVAL;
  float eps = eps0;
  for (int i = 1; i < attempt; ++i) eps *= 10.f;
  const size_t lds = (size_t)d * (d + 1) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = cmf_set_dynamic_lds((const void*)chol_retry_kernel, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(chol_retry_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, jtj, d, attempt, eps, logdet,
                     l1_diag, info, fail);
  CMF_LAUNCH_CHECK();
  return 0;
}
