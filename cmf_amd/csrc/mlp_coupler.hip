// Fused affine-coupling layer with an MLP coupler (2-D / tabular models, low-dimensional prior flows) for gfx950: the whole
// coupler network -- primal AND all Jacobian columns -- plus the coupling update in ONE persistent launch.
//
// Reference: AffineCouplingBijection._z_to_x / _jvp / _x_to_z (cmf/models/components/bijections/acl.py:101-146) around
// get_mlp (networks.py:206-224) evaluated with NN_Sequential_JVP.jvp and the tanh rule (jvp_layers.py:24-53):
//     h_{l+1} = tanh(W_l h_l + b_l),   hdot_{l+1} = (1 - h_{l+1}^2) (W_l hdot_l);   last layer linear: (t, s) = W h + b
//     decode  x_mod = z_mod e^{-s} - t,   xdot_mod = e^{-s} (v_mod - z_mod sdot) - tdot
//     encode  z_mod = (x_mod + t) e^{s},  log-jac = sum s
// The unfused path runs one cmf_conv_primal and one cmf_conv_tangent launch per linear layer plus two coupling kernels:
// ~12 launches of a few microseconds per coupling layer, 150 - 250 per log-density evaluation of a tabular model (C2),
// each far too small to fill the chip.  Here a sample never leaves the CU between layers.
//
// Mapping (fp32 MFMA 16x16x4: exact fp32 products, fp32 accumulation).
//   TANGENT mode (d <= 15, tangents with NC = 16 columns): a wavefront's 16-column N tile holds floor(16 / (d + 1)) SAMPLES,
//     each as d Jacobian columns followed by its PRIMAL column (d = 10: one sample per tile; d = 2: five): bias and tanh act on
//     the primal columns, whose 1 - h^2 scales the sample's other columns.
//   PRIMAL mode (encode pass, sampling): the 16 columns are 16 SAMPLES; bias and tanh on every column.
//   A layer is Y[out, 16] = W[out, in] X[in, 16]: the accumulators of layer l ARE the B operands of layer l + 1 -- in the
//   accumulator layout lane (kq, cl) register (t, r) holds feature 16 t + 4 kq + r of column cl, and the weights are packed
//   (cmf_pack_mlp_layer) so that K-step 4 t + r of the next layer expects exactly that feature from lane group kq: no
//   shuffle, no LDS round trip for the activations.  Only the primal pre-activations of the TANGENT mode cross lanes: the
//   lanes of the primal columns park them in LDS (128 floats per sample), the wave takes one tanh per value in place, and
//   every lane reads back h (and forms 1 - h^2) of its sample.
//   Weights: a layer's A fragments (up to 128 x 128 fp32 = 64 KB) are streamed L2 -> LDS by LDS-DMA (global_load_lds, 16 B
//   per lane, the image is lane-linear by construction of the pack) into a double buffer while the previous layer computes;
//   the 8 waves of a workgroup walk the layers in lockstep (one barrier per layer), a workgroup handles 8 samples (128 in
//   PRIMAL mode) per pass over the weights.
//
// Bound: MFMA (fp32) for 128-wide layers -- 864 MFMAs of 32 cycles per sample and coupling layer at C2b; the launch-bound
// small nets (C1: 10-wide) become one launch per coupling layer.
#include "common.h"

namespace {

constexpr int MAXL = CMF_MLP_MAX_LAYERS;
constexpr int WAVES = 8;
// per-wave LDS scratch: S (128 floats per sample slot: primal pre-activations, then h in place; hidden layers) and, after the
// last layer, Y (out features x 16 columns) share one region of max(128 spt, 16 ceil16(outputs)) floats

__device__ __forceinline__ int ntile(int n) { return (n + 15) >> 4; }

// floats of a packed layer image: A fragments [mt][kg][64 lanes][4] + bias [16 mt]
__host__ __device__ inline long long image_floats(int n_kg, int n_mt) { return (long long)n_mt * n_kg * 256 + 16 * n_mt; }

// Tile counts of layer l: hidden widths are padded to the template's HT tiles (zero weight rows give h = tanh(0) = 0 and feed
// zero weight columns of the next layer), so that the hidden-layer loops carry no guards; the first layer's K groups and the
// last layer's output tiles are what the sizes need.
__host__ __device__ inline int layer_kg(const cmf_mlp_coupler_args& a, int l, int ht) { return l == 0 ? (a.width[0] + 15) / 16 : ht; }
__host__ __device__ inline int layer_mt(const cmf_mlp_coupler_args& a, int l, int ht) {
  return l == a.n_layers - 1 ? (a.width[a.n_layers] + 15) / 16 : ht;
}

// Layer image: fragment (mt, kg) of lane (kq = lane / 16, m = lane % 16), component j = the weight of output feature
// 16 mt + m and input feature  16 kg + 4 j + kq  (first layer: K-step 4 kg + j reads the gathered input rows in natural
// order)  or  16 kg + 4 kq + j  (later layers: K-step (kg, j) is register j of accumulator tile kg of the previous layer).
__global__ void pack_mlp_layer_kernel(const float* __restrict__ w, const float* __restrict__ bias, int out_f, int in_f, int first,
                                      int n_mt, int n_kg, float* __restrict__ out) {
  const long long nfrag = (long long)n_mt * n_kg * 256;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nfrag) {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const long long f = i >> 8;
    const int kg = (int)(f % n_kg), mt = (int)(f / n_kg);
    const int kq = lane >> 4, m = lane & 15;
    const int o = 16 * mt + m, in = first ? 16 * kg + 4 * j + kq : 16 * kg + 4 * kq + j;
    out[i] = (o < out_f && in < in_f) ? w[(long long)o * in_f + in] : 0.f;
  } else if (i < nfrag + 16 * n_mt) {
    const int o = (int)(i - nfrag);
    out[i] = (bias && o < out_f) ? bias[o] : 0.f;
  }
}

// LDS-DMA copy of `nfloat` floats (a multiple of 4) global -> LDS by the whole workgroup, lane-linear
template <int NW = WAVES>
__device__ __forceinline__ void stage_image(const float* __restrict__ src, float* dst, long long nfloat, int tid) {
  const long long nchunk = nfloat >> 2;                 // 16-byte chunks
  for (long long c0 = 0; c0 < nchunk; c0 += NW * 64) {
    const long long c = c0 + tid;                       // wave-instruction: 64 consecutive chunks = 1 KiB
    if (c < nchunk)                                     // destination = wave-uniform base (+ lane x 16 B by the hardware)
      __builtin_amdgcn_global_load_lds(src + 4 * c, dst + 4 * (c - (tid & 63)), 16, 0, 0);
  }
}

// NW waves per workgroup: 8, or 4 where 8-wave tiles would leave CUs without work (launcher)
template <int HT, bool TAN, int NW>
__global__ __launch_bounds__(NW * 64) void mlp_coupler_kernel(cmf_mlp_coupler_args a, int n_tiles, int buf_floats, int scr_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;
  float* scr = smem + 2 * buf_floats + wave * scr_floats;
  float* S = scr, *Y = scr;                              // S [sample in tile][128]: primal pre-activation -> h; Y: output table
  const int L = a.n_layers;
  // TANGENT: c = d + 1 columns per sample, spt samples per 16-column tile; this lane serves column jc of sample slot ks
  const int c = TAN ? a.ncols + 1 : 1, spt = TAN ? 16 / c : 16;
  const int ks = TAN ? cl / c : cl, jc = TAN ? cl - ks * c : 0;
  const bool slot_ok = ks < spt;                         // TANGENT: the trailing 16 - spt c columns of a tile are unused
  const bool is_primal = !TAN || jc == a.ncols;

  int gcount = 0;                                        // layers staged so far: buffer parity
  if ((int)blockIdx.x < n_tiles) stage_image<NW>(a.w + a.w_off[0], smem, image_floats(layer_kg(a, 0, HT), layer_mt(a, 0, HT)), tid);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b0 = (tile * NW + wave) * spt;
    const bool live = b0 < a.B;                          // wave-uniform
    const int bmine = b0 + ks;                           // this lane's sample
    const bool colok = slot_ok && bmine < a.B;
    float bop[HT][4];                                    // B operands of the next hidden layer
    f32x4 acc_o[4];                                      // the output layer's tiles
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) bop[t][r] = 0.f;

    for (int l = 0; l < L; ++l, ++gcount) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this layer's image has landed (issued one layer ago)
      __syncthreads();                                   // ... for every wave; and everyone is done with the other buffer
      {
        const int nl = l + 1 < L ? l + 1 : 0;
        if (l + 1 < L || tile + (int)gridDim.x < n_tiles)
          stage_image<NW>(a.w + a.w_off[nl], smem + ((gcount + 1) & 1) * buf_floats,
                      image_floats(layer_kg(a, nl, HT), layer_mt(a, nl, HT)), tid);
      }
      if (!live) continue;
      const float* W = smem + (gcount & 1) * buf_floats;
      const int n_mt = layer_mt(a, l, HT), n_kg = layer_kg(a, l, HT);
      const float* Wb = W + (long long)n_mt * n_kg * 256;  // bias part
      const bool last = l == L - 1;
      f32x4 acc[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) acc_o[t] = f32x4{0.f, 0.f, 0.f, 0.f};

      // all output tiles against one K group of four K-steps: the fragments of every tile first, then K-step-major so that
      // consecutive MFMAs hit DIFFERENT accumulators (four back-to-back MFMAs on one accumulator each wait out the 40-cycle
      // dependent latency of v_mfma_f32_16x16x4_f32 instead of issuing every 32)
      auto mma = [&](int kg, const float (&b4)[4]) {
        if (!last) {
          f32x4 a4[HT];
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) a4[mt] = *reinterpret_cast<const f32x4*>(W + ((mt * n_kg + kg) * 64 + lane) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < HT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][j], b4[j], acc[mt], 0, 0, 0);
            if (HT > 1) __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          f32x4 a4[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            if (mt < n_mt) a4[mt] = *reinterpret_cast<const f32x4*>(W + ((long long)(mt * n_kg + kg) * 64 + lane) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              if (mt < n_mt) acc_o[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][j], b4[j], acc_o[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };

      if (l == 0) {
        // gathered input rows: feature i = 16 kg + 4 j + kq of the rows the network reads
        for (int kg = 0; kg < n_kg; ++kg) {
          float b4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = 16 * kg + 4 * j + kq;
            float v = 0.f;
            if (i < a.cin && colok) {
              const long long f = a.chan_off + (long long)i * a.chan_step;
              if (is_primal) v = a.z[(long long)bmine * a.z_b + f];
              else v = a.t[f * a.t_f + (long long)bmine * 16 + jc];
            }
            b4[j] = v;
          }
          mma(kg, b4);
        }
      } else {
#pragma unroll
        for (int kg = 0; kg < HT; ++kg) mma(kg, bop[kg]);
      }

      if (!last) {
        if (TAN) {
          // primal pre-activations -> LDS (per sample slot); tanh in place; h back to every lane of the sample
#pragma unroll
          for (int mt = 0; mt < HT; ++mt)
            if (is_primal && slot_ok) {
              const f32x4 b4 = *reinterpret_cast<const f32x4*>(Wb + 16 * mt + 4 * kq);
              *reinterpret_cast<f32x4*>(S + ks * 128 + 16 * mt + 4 * kq) = acc[mt] + b4;
            }
          __builtin_amdgcn_wave_barrier();
          for (int f = lane; f < spt * 128; f += 64)
            if ((f & 127) < 16 * HT) S[f] = tanhf(S[f]);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            const f32x4 hv = slot_ok ? *reinterpret_cast<const f32x4*>(S + ks * 128 + 16 * mt + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)                  // 1 - h^2: the reference's tanh rule (jvp_layers.py:42-44)
              bop[mt][r] = is_primal ? hv[r] : acc[mt][r] * (1.f - hv[r] * hv[r]);
          }
          __builtin_amdgcn_wave_barrier();               // S is rewritten by the next layer
        } else {
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(Wb + 16 * mt + 4 * kq);
#pragma unroll
            for (int r = 0; r < 4; ++r) bop[mt][r] = tanhf(acc[mt][r] + b4[r]);
          }
        }
      } else {
        // network output (t, s) and their tangents -> per-wave LDS table Y[feature][column]
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if (mt < n_mt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(Wb + 16 * mt + 4 * kq);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Y[(16 * mt + 4 * kq + r) * 16 + cl] = acc_o[mt][r] + (is_primal ? b4[r] : 0.f);
          }
      }
    }
    if (!live) continue;
    __builtin_amdgcn_wave_barrier();

    // coupling update: 4 modified elements x 16 columns per step
    float ljacc = 0.f;
    for (int e0 = 0; e0 < a.n_mod; e0 += 4) {
      const int e = e0 + kq;
      if (e < a.n_mod && colok) {
        const int rs = a.si[e], rt = a.ti[e], rz = a.zi[e];
        if (TAN) {
          const int pc = ks * c + a.ncols;               // the primal column of this lane's sample
          const float s = Y[rs * 16 + pc], tt = Y[rt * 16 + pc];
          const float zo = a.z[(long long)bmine * a.z_b + rz];
          const float es = expf(-s);
          if (!is_primal) {
            float* tp = a.t + (long long)rz * a.t_f + (long long)bmine * 16 + jc;
            const float sd = Y[rs * 16 + cl], td = Y[rt * 16 + cl];
            *tp = es * (*tp - zo * sd) - td;             // acl.py:58-60 with this column's (sdot, tdot)
          } else {
            a.z[(long long)bmine * a.z_b + rz] = zo * es - tt;
          }
        } else {
          const float s = Y[rs * 16 + cl], tt = Y[rt * 16 + cl];
          float* zp = a.z + (long long)bmine * a.z_b + rz;
          *zp = a.decode ? (*zp) * expf(-s) - tt : ((*zp) + tt) * expf(s);
          ljacc += s;
        }
      }
    }
    if (!TAN && a.lj) {
      ljacc += __shfl_xor(ljacc, 16, 64);
      ljacc += __shfl_xor(ljacc, 32, 64);
      if (kq == 0 && colok) a.lj[bmine] += a.decode ? -ljacc : ljacc;
    }
    __builtin_amdgcn_wave_barrier();                     // Y is rewritten by the next tile
  }
}


// PRIMAL mode for wide networks (hidden tiles = 8), M split over the waves.  In mlp_coupler_kernel<8, false> a wave pushes its
// 16 samples through a whole 128 x 128 layer alone: 256 dependent-issue MFMAs = 8 k cycles per layer, and a 4096-sample batch
// is only 32 tiles, so 32 of 256 CUs work and a launch takes 35 us of pure latency (15 such launches per C2 evaluation: the
// encode pass and the prior flows).  Here a workgroup owns ONE 16-sample tile (256 tiles: every CU) and wave w computes output
// tile w of a hidden layer (32 MFMAs, two interleaved accumulators); the activations cross the waves through LDS as float4s
// -- in the permuted K order of the packed images a lane's four accumulator registers are exactly the four K-steps lane group
// kq of K group w reads in the next layer, so the exchange is one ds_write_b128 and eight ds_read_b128 per lane and layer.
__global__ __launch_bounds__(WAVES * 64) void mlp_coupler_split_kernel(cmf_mlp_coupler_args a, int n_tiles, int buf_floats) {
  constexpr int HT = 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, cl = lane & 15;
  f32x4* X = reinterpret_cast<f32x4*>(smem + 2 * buf_floats);        // [2][8 kg][4 kq][16 cl] float4
  float* Y = smem + 2 * buf_floats + 2 * HT * 64 * 4;                 // [64 features][16 columns]
  float* LJ = Y + 64 * 16;                                            // [8 waves][16]
  const int L = a.n_layers;

  int gcount = 0;
  if ((int)blockIdx.x < n_tiles) stage_image(a.w + a.w_off[0], smem, image_floats(layer_kg(a, 0, HT), layer_mt(a, 0, HT)), tid);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int bmine = tile * 16 + cl;
    const bool colok = bmine < a.B;
    for (int l = 0; l < L; ++l, ++gcount) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // weights of layer l landed; X of layer l - 1 complete
      {
        const int nl = l + 1 < L ? l + 1 : 0;
        if (l + 1 < L || tile + (int)gridDim.x < n_tiles)
          stage_image(a.w + a.w_off[nl], smem + ((gcount + 1) & 1) * buf_floats,
                      image_floats(layer_kg(a, nl, HT), layer_mt(a, nl, HT)), tid);
      }
      const float* W = smem + (gcount & 1) * buf_floats;
      const int n_mt = layer_mt(a, l, HT), n_kg = layer_kg(a, l, HT);
      const float* Wb = W + (long long)n_mt * n_kg * 256;
      const bool last = l == L - 1;
      if (wave >= n_mt) continue;                        // wave-uniform: the output layer has fewer tiles than waves
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const f32x4* Xin = X + ((l + 1) & 1) * HT * 64;    // written by layer l - 1
      for (int kg = 0; kg < n_kg; ++kg) {
        f32x4 b4;
        if (l == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = 16 * kg + 4 * j + kq;
            b4[j] = (i < a.cin && colok) ? a.z[(long long)bmine * a.z_b + a.chan_off + (long long)i * a.chan_step] : 0.f;
          }
        } else {
          b4 = Xin[(kg * 4 + kq) * 16 + cl];
        }
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(W + ((wave * n_kg + kg) * 64 + lane) * 4);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b4[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], b4[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], b4[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], b4[3], acc1, 0, 0, 0);
      }
      const f32x4 bias4 = *reinterpret_cast<const f32x4*>(Wb + 16 * wave + 4 * kq);
      const f32x4 v = acc0 + acc1 + bias4;
      if (!last) {
        X[(l & 1) * HT * 64 + (wave * 4 + kq) * 16 + cl] = f32x4{tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[(16 * wave + 4 * kq + r) * 16 + cl] = v[r];
      }
    }
    __syncthreads();                                     // Y complete
    float ljacc = 0.f;
    for (int e0 = 0; e0 < a.n_mod; e0 += WAVES * 4) {
      const int e = e0 + wave * 4 + kq;
      if (e < a.n_mod && colok) {
        const int rs = a.si[e], rt = a.ti[e], rz = a.zi[e];
        const float s = Y[rs * 16 + cl], tt = Y[rt * 16 + cl];
        float* zp = a.z + (long long)bmine * a.z_b + rz;
        *zp = a.decode ? (*zp) * expf(-s) - tt : ((*zp) + tt) * expf(s);
        ljacc += s;
      }
    }
    if (a.lj) {                                          // uniform
      ljacc += __shfl_xor(ljacc, 16, 64);
      ljacc += __shfl_xor(ljacc, 32, 64);
      if (kq == 0) LJ[wave * 16 + cl] = ljacc;
      __syncthreads();
      if (wave == 0 && kq == 0 && colok) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += LJ[w * 16 + cl];
        a.lj[bmine] += a.decode ? -t : t;
      }
    }
  }
}

int launch_split(const cmf_mlp_coupler_args& a, int buf_floats, hipStream_t s) {
  const int n_tiles = cmf_ceil_div(a.B, 16);
  const int lds = (2 * buf_floats + 2 * 8 * 64 * 4 + 64 * 16 + WAVES * 16) * (int)sizeof(float);
  if (lds > 160 * 1024) return CMF_ERANGE;
  if (hipError_t e = cmf_set_dynamic_lds((const void*)mlp_coupler_split_kernel, lds); e != hipSuccess) return (int)e;
  const int cus = cmf_device_cus();
  const int grid = n_tiles < cus ? n_tiles : cus;
  hipLaunchKernelGGL(mlp_coupler_split_kernel, dim3(grid), dim3(WAVES * 64), lds, s, a, n_tiles, buf_floats);
  CMF_LAUNCH_CHECK();
  return 0;
}

template <int HT, bool TAN, int NW = WAVES>
int launch(const cmf_mlp_coupler_args& a, int buf_floats, hipStream_t s) {
  const int spt = TAN ? 16 / (a.ncols + 1) : 16;
  const int per_tile = NW * spt;
  const int n_tiles = cmf_ceil_div(a.B, per_tile);
  if constexpr (TAN && NW == 8) {
    // TANGENT mode with several samples per wave tile (small d): 8-wave tiles leave CUs without a workgroup (C2a, d = 2,
    // B = 4096: 103 tiles of 40 samples on 256 CUs, two waves per SIMD on those) -- 4-wave workgroups put one wave on every SIMD of
    // twice as many CUs
    if (n_tiles < cmf_device_cus() && a.B > 4 * spt) return launch<HT, TAN, 4>(a, buf_floats, s);
  }
  const int out_pad = (a.width[a.n_layers] + 15) / 16 * 16;
  const int s_floats = TAN ? 128 * spt : 0;              // S [sample slot][128] and the output table Y share the region
  const int scr_floats = 16 * out_pad > s_floats ? 16 * out_pad : s_floats;
  const int lds = (2 * buf_floats + NW * scr_floats) * (int)sizeof(float);
  if (lds > 160 * 1024) return CMF_ERANGE;
  auto k = mlp_coupler_kernel<HT, TAN, NW>;
  if (hipError_t e = cmf_set_dynamic_lds((const void*)k, lds); e != hipSuccess) return (int)e;
  const int cus = cmf_device_cus();
  // small images leave room for more than one workgroup per CU; the grid is persistent over the tiles
  const int per_cu = lds <= 40 * 1024 ? 2 : 1;
  const int grid = n_tiles < cus * per_cu ? n_tiles : cus * per_cu;
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, s, a, n_tiles, buf_floats, scr_floats);
  CMF_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int cmf_mlp_hidden_tiles(int max_hidden_width) { return max_hidden_width <= 16 ? 1 : (max_hidden_width <= 32 ? 2 : 8); }

extern "C" int cmf_pack_mlp_layer(const float* w, const float* bias, int out_features, int in_features, int first, int out_tiles,
                                  int in_groups, float* out, long long* out_floats, void* stream) {
  if (out_features <= 0 || in_features <= 0 || 16 * out_tiles < out_features || 16 * in_groups < in_features) return CMF_EINVAL;
  const long long n = image_floats(in_groups, out_tiles);
  if (out_floats) *out_floats = n;
  if (!out) return out_floats ? 0 : CMF_EINVAL;
  if (!w) return CMF_EINVAL;
  hipLaunchKernelGGL(pack_mlp_layer_kernel, dim3(cmf_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w, bias, out_features,
                     in_features, first, out_tiles, in_groups, out);
  CMF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cmf_mlp_coupler(const cmf_mlp_coupler_args* a, void* stream) {
  if (!a || !a->z || !a->w || !a->zi || !a->si || !a->ti) return CMF_EINVAL;
  if (a->B <= 0 || a->n_mod <= 0 || a->cin <= 0 || a->n_layers < 2 || a->n_layers > MAXL || a->width[0] != a->cin) return CMF_EINVAL;
  if ((uintptr_t)a->w % 16) return CMF_EINVAL;
  int hmax = 0;
  for (int l = 0; l < a->n_layers; ++l) {
    if (a->width[l + 1] <= 0 || a->w_off[l] % 4) return CMF_EINVAL;
    if (l + 1 < a->n_layers && a->width[l + 1] > hmax) hmax = a->width[l + 1];
  }
  if (hmax > 128 || a->width[a->n_layers] > 64 || a->cin > 128) return CMF_EINVAL;   // 8 hidden tiles, 4 output tiles
  const int ht = cmf_mlp_hidden_tiles(hmax);
  long long buf = 0;
  for (int l = 0; l < a->n_layers; ++l) {
    const long long n = image_floats(layer_kg(*a, l, ht), layer_mt(*a, l, ht));
    if (n > buf) buf = n;
  }
  if (a->t && ((uintptr_t)a->t % 16 || a->t_f % 4 || !a->decode || a->ncols < 1 || a->ncols > 15)) return CMF_EINVAL;
  buf = (buf + 255) / 256 * 256;                       // whole 1 KiB wave-instructions of the LDS-DMA copy
  hipStream_t s = (hipStream_t)stream;
  if (a->t) {
    if (ht == 1) return launch<1, true>(*a, (int)buf, s);
    if (ht == 2) return launch<2, true>(*a, (int)buf, s);
    return launch<8, true>(*a, (int)buf, s);
  }
  if (ht == 1) return launch<1, false>(*a, (int)buf, s);
  if (ht == 2) return launch<2, false>(*a, (int)buf, s);
  // wide networks: M split over the waves while the batch does not fill the chip with 128-sample tiles (latency-bound
  // there); beyond that the one-wave-per-16-samples form streams each weight image once per 128 samples instead of per 16
  if (a->B <= 128 * 2 * cmf_device_cus()) return launch_split(*a, (int)buf, s);
  return launch<8, false>(*a, (int)buf, s);
}
