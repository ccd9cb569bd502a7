// HBM-bound pieces of the non-square log-density path for gfx950: affine-coupling transforms,
// index-map moves (squeeze / split-pad / tail gather-scatter), tangent seeding, pre-head logit chain,
// Gaussian / affine priors, reconstruction error and the final elbo combination.
// Every kernel is a coalesced stream over contiguous Jacobian columns (16-byte accesses on the tangent
// tensors) or over a sample's elements; per-sample sums use wavefront shuffles (64 lanes).
// Reference semantics are cited per entry point in include/cmf_amd.h.
#include "common.h"

namespace {

constexpr int TPB = 256;
inline int nblocks(long long n) { return (int)((n + TPB - 1) / TPB); }

// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps,
                                   int transpose, int ncin_pad, long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i & 63);
  long long t = i >> 6;
  const int ci = (int)(t % ncin_pad);
  t /= ncin_pad;
  const int tap = (int)(t % taps), cog = (int)(t / taps);
  const int co = cog * 64 + col;
  const int ncout = transpose ? cin : cout, ncin = transpose ? cout : cin;
  float v = 0.f;
  if (co < ncout && ci < ncin)
    v = transpose ? w[((long long)ci * cin + co) * taps + (taps - 1 - tap)] : w[((long long)co * cin + ci) * taps + tap];
  out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// acl primal, flat over (b, e); no log-jac
__global__ void acl_primal_flat(float* __restrict__ z, long long z_b, const float* __restrict__ y, long long y_b,
                                const int* __restrict__ zi, const int* __restrict__ si, const int* __restrict__ ti,
                                int n_mod, long long total, int decode) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const long long b = i / n_mod;
  const int e = (int)(i % n_mod);
  const float s = y[b * y_b + si[e]], t = y[b * y_b + ti[e]];
  float* zp = z + b * z_b + zi[e];
  *zp = decode ? (*zp) * expf(-s) - t : ((*zp) + t) * expf(s);
}

// acl primal encode with log-jac: one wavefront per sample
__global__ void acl_primal_lj(float* __restrict__ z, long long z_b, const float* __restrict__ y, long long y_b,
                              const int* __restrict__ zi, const int* __restrict__ si, const int* __restrict__ ti,
                              int n_mod, int B, int decode, float* __restrict__ lj) {
  const int b = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float acc = 0.f;
  for (int e = lane; e < n_mod; e += 64) {
    const float s = y[(long long)b * y_b + si[e]], t = y[(long long)b * y_b + ti[e]];
    float* zp = z + (long long)b * z_b + zi[e];
    *zp = decode ? (*zp) * expf(-s) - t : ((*zp) + t) * expf(s);
    acc += s;
  }
  acc = wave_sum(acc);
  if (lane == 0) lj[b] += decode ? -acc : acc;
}

// acl tangent: flat over (b, e, col4)
__global__ void acl_tangent_kernel(float* __restrict__ t, long long t_b, long long t_r, const float* __restrict__ yt,
                                   long long yt_b, long long yt_r, int nc4, const float* __restrict__ z, long long z_b,
                                   const float* __restrict__ y, long long y_b, const float* __restrict__ g,
                                   const int* __restrict__ zi, const int* __restrict__ si, const int* __restrict__ ti,
                                   int n_mod, long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % nc4);
  const long long be = i / nc4;
  const int e = (int)(be % n_mod);
  const long long b = be / n_mod;
  const int rs = si[e], rt = ti[e], rz = zi[e];
  const float s = y[b * y_b + rs], zo = z[b * z_b + rz];
  const float gs = g ? g[b * y_b + rs] : 1.f, gt = g ? g[b * y_b + rt] : 1.f;
  const float es = expf(-s);
  f32x4* tp = reinterpret_cast<f32x4*>(t + b * t_b + (long long)rz * t_r) + c4;
  const f32x4 v = *tp;
  if (!yt) {                        // the network's tangent is identically zero (its input is structurally zero): s-dot = t-dot = 0
    *tp = es * v;
    return;
  }
  const f32x4 sd = reinterpret_cast<const f32x4*>(yt + b * yt_b + (long long)rs * yt_r)[c4];
  const f32x4 td = reinterpret_cast<const f32x4*>(yt + b * yt_b + (long long)rt * yt_r)[c4];
  *tp = es * (v - (zo * gs) * sd) - gt * td;
}

// acl cotangent (adjoint of acl_tangent_kernel): flat over (b, e, col4)
__global__ void acl_cotangent_kernel(float* __restrict__ c, long long c_b, long long c_r, float* __restrict__ yc,
                                     long long yc_b, long long yc_r, int nc4, const float* __restrict__ z, long long z_b,
                                     const float* __restrict__ y, long long y_b, const float* __restrict__ g,
                                     const int* __restrict__ zi, const int* __restrict__ si, const int* __restrict__ ti,
                                     int n_mod, long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % nc4);
  const long long be = i / nc4;
  const int e = (int)(be % n_mod);
  const long long b = be / n_mod;
  const int rs = si[e], rt = ti[e], rz = zi[e];
  const float s = y[b * y_b + rs], zo = z[b * z_b + rz];
  const float gs = g ? g[b * y_b + rs] : 1.f, gt = g ? g[b * y_b + rt] : 1.f;
  const float es = expf(-s);
  f32x4* cp = reinterpret_cast<f32x4*>(c + b * c_b + (long long)rz * c_r) + c4;
  const f32x4 v = *cp;
  if (yc) {                         // null: nobody reads the cotangent of the network's output (its input rows are dropped)
    reinterpret_cast<f32x4*>(yc + b * yc_b + (long long)rt * yc_r)[c4] = (-gt) * v;
    reinterpret_cast<f32x4*>(yc + b * yc_b + (long long)rs * yc_r)[c4] = (-(es * zo * gs)) * v;
  }
  *cp = es * v;
}

// acl primal backward (training), flat over (b, e), in place on the primal cotangent dx (full tensor; pass-through elements
// keep their value):
//   decode  x_mod = z_mod e^{-s} - t :  dz_mod = dx e^{-s};  dy[s] -= dx z_mod e^{-s};  dy[t] -= dx          (z = tensor BEFORE)
//   encode  z_mod = (x_mod + t) e^{s}:  dx_mod = dz e^{s};   dy[t] += dz e^{s};  dy[s] += dz (x_mod + t) e^{s} + dlj[b]
__global__ void acl_primal_backward_kernel(float* __restrict__ dx, long long dx_b, const float* __restrict__ z, long long z_b,
                                           const float* __restrict__ y, long long y_b, float* __restrict__ dy,
                                           const int* __restrict__ zi, const int* __restrict__ si, const int* __restrict__ ti,
                                           int n_mod, long long total, int decode, const float* __restrict__ dlj) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i % n_mod);
  const long long b = i / n_mod;
  const int rs = si[e], rt = ti[e], rz = zi[e];
  const float s = y[b * y_b + rs], zo = z[b * z_b + rz], d = dx[b * dx_b + rz];
  if (decode) {
    const float es = expf(-s);
    dx[b * dx_b + rz] = d * es;
    dy[b * y_b + rs] -= d * zo * es;
    dy[b * y_b + rt] -= d;
  } else {
    const float es = expf(s), t = y[b * y_b + rt];
    dx[b * dx_b + rz] = d * es;
    dy[b * y_b + rt] += d * es;
    dy[b * y_b + rs] += d * (zo + t) * es + (dlj ? dlj[b] : 0.f);
  }
}

// dst += src over n floats (n % 4 == 0): the skip connection of the reverse sweep, c_h = c_h2 + relu'(a) . conv1^T(c_u), when the
// masked product comes from the split-precision kernel (whose residual input would be masked with the product)
__global__ void accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4) {
  const long long stride = (long long)gridDim.x * TPB;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += stride) {
    f32x4 d = reinterpret_cast<f32x4*>(dst)[i];
    d += reinterpret_cast<const f32x4*>(src)[i];
    reinterpret_cast<f32x4*>(dst)[i] = d;
  }
}

__global__ void accumulate_scalar_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
  const long long stride = (long long)gridDim.x * TPB;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

// acl cross terms (training): the tangent update  out = es (v - zo gs sd) - gt td  also depends on PRIMAL quantities;
// their cotangents are column reductions of c . d(out)/d(.) -- one wavefront per (sample, modified element):
//   d s  = -sum_col c es (v - zo gs sd)     d zo = -sum_col c es gs sd
//   d gs = -sum_col c es zo sd              d gt = -sum_col c td
__global__ __launch_bounds__(256) void acl_cross_terms_kernel(const float* __restrict__ c, long long c_b, long long c_r,
                                                              const float* __restrict__ v, long long v_b, long long v_r,
                                                              const float* __restrict__ yt, long long yt_b, long long yt_r,
                                                              int nc, const float* __restrict__ z, long long z_b,
                                                              const float* __restrict__ y, long long y_b,
                                                              const float* __restrict__ g, const int* __restrict__ zi,
                                                              const int* __restrict__ si, const int* __restrict__ ti, int n_mod,
                                                              long long n_rows, float* __restrict__ dz,
                                                              float* __restrict__ dy, float* __restrict__ dg) {
  const long long be = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (be >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int e = (int)(be % n_mod);
  const long long b = be / n_mod;
  const int rs = si[e], rt = ti[e], rz = zi[e];
  const float s = y[b * y_b + rs], zo = z[b * z_b + rz];
  const float gs = g ? g[b * y_b + rs] : 1.f;
  const float es = expf(-s);
  const float* cp = c + b * c_b + (long long)rz * c_r;
  const float* vp = v + b * v_b + (long long)e * v_r;
  const float* sp = yt + b * yt_b + (long long)rs * yt_r;
  const float* tp = yt + b * yt_b + (long long)rt * yt_r;
  float a_s = 0.f, a_z = 0.f, a_gs = 0.f, a_gt = 0.f;
  if (!yt) {                        // s-dot = t-dot = 0: only the log-scale sees the tangent update, out = es v
    for (int k = lane; k < nc; k += 64) a_s -= cp[k] * es * vp[k];
    a_s = wave_sum(a_s);
    if (lane == 0) dy[b * y_b + rs] += a_s;
    return;
  }
  for (int k = lane; k < nc; k += 64) {
    const float cv = cp[k], sd = sp[k];
    a_s -= cv * es * (vp[k] - zo * gs * sd);
    a_z -= cv * es * gs * sd;
    a_gs -= cv * es * zo * sd;
    a_gt -= cv * tp[k];
  }
  a_s = wave_sum(a_s), a_z = wave_sum(a_z), a_gs = wave_sum(a_gs), a_gt = wave_sum(a_gt);
  if (lane == 0) {
    dy[b * y_b + rs] += a_s;
    dz[b * z_b + rz] += a_z;
    if (dg) {
      dg[b * y_b + rs] += a_gs;
      dg[b * y_b + rt] += a_gt;
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ void gather_primal_kernel(const float* __restrict__ in, long long in_b, float* __restrict__ out,
                                     long long out_b, const int* __restrict__ idx, int n_out, long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const long long b = i / n_out;
  const int r = (int)(i % n_out), s = idx[r];
  out[b * out_b + r] = s >= 0 ? in[b * in_b + s] : 0.f;
}

__global__ void gather_tangent_kernel(const float* __restrict__ in, long long in_b, long long in_r,
                                      float* __restrict__ out, long long out_b, long long out_r,
                                      const int* __restrict__ idx, int n_out, int nc4, long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % nc4);
  const long long br = i / nc4;
  const int r = (int)(br % n_out);
  const long long b = br / n_out;
  const int s = idx[r];
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (s >= 0) v = reinterpret_cast<const f32x4*>(in + b * in_b + (long long)s * in_r)[c4];
  reinterpret_cast<f32x4*>(out + b * out_b + (long long)r * out_r)[c4] = v;
}

__global__ void seed_tangent_kernel(float* __restrict__ t, long long t_b, long long t_r, const int* __restrict__ col_of,
                                    int n_rows, int nc4, const float* __restrict__ eps, int d, int S, long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % nc4);
  const long long br = i / nc4;
  const int r = (int)(br % n_rows);
  const long long b = br / n_rows;
  const int j = col_of[r];
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (j >= 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c4 * 4 + k;
      if (eps) v[k] = c < S ? eps[(b * d + j) * S + c] : 0.f;
      else v[k] = (c == j) ? 1.f : 0.f;
    }
  }
  reinterpret_cast<f32x4*>(t + b * t_b + (long long)r * t_r)[c4] = v;
}

// column expansion of a tangent tensor: out(row, c) = colmap[c] >= 0 ? in(row, colmap[c]) : 0; flat over (row, c4)
__global__ void expand_columns_kernel(const float* __restrict__ in, int nc_in, float* __restrict__ out, int nc4,
                                      const int* __restrict__ colmap, long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % nc4);
  const long long row = i / nc4;
  const float* src = in + row * nc_in;
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = colmap[c4 * 4 + k];
    v[k] = j >= 0 ? src[j] : 0.f;
  }
  reinterpret_cast<f32x4*>(out + row * (long long)nc4 * 4)[c4] = v;
}

// (B, N) <-> (B/16, N, 16): 16 x 16 transposes through registers of one 16-lane group
__global__ void primal_regroup_kernel(const float* __restrict__ in, float* __restrict__ out, int B, long long N,
                                      int to_grouped) {
  __shared__ float tile[16][17];
  const int g = blockIdx.y;                                  // sample group
  const long long n0 = (long long)blockIdx.x * 16;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;    // 256 threads = 16 x 16
  if (to_grouped) {
    if (n0 + tx < N) tile[ty][tx] = in[((long long)g * 16 + ty) * N + n0 + tx];       // row = sample, col = element
    __syncthreads();
    if (n0 + ty < N) out[((long long)g * N + n0 + ty) * 16 + tx] = tile[tx][ty];
  } else {
    if (n0 + ty < N) tile[ty][tx] = in[((long long)g * N + n0 + ty) * 16 + tx];       // row = element, col = sample
    __syncthreads();
    if (n0 + tx < N) out[((long long)g * 16 + ty) * N + n0 + tx] = tile[tx][ty];
  }
}

// ------------------------------------------------------------------------------------------------
// pre-head: one block per sample
__global__ void prehead_kernel(const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ y,
                               float* __restrict__ lj, float a, float c, int logit, int n) {
  __shared__ float red[16];
  const long long base = (long long)blockIdx.x * n;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v = x[base + i];
    if (u) v += u[base + i];
    v = a * v + c;
    if (logit) {
      const float vc = fminf(fmaxf(v, 1e-7f), 1.f - 1e-7f);
      acc += -logf(vc) - logf(1.f - vc);
      v = logf(v) - logf(1.f - v);
    }
    y[base + i] = v;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0 && lj) lj[blockIdx.x] = acc + (float)n * logf(fabsf(a));
}

__global__ void prehead_inverse_kernel(const float* __restrict__ y, float* __restrict__ x, float a, float c, int logit,
                                       long long total) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  float v = y[i];
  if (logit) v = 1.f / (1.f + expf(-v));
  x[i] = (v - c) / a;
}

__global__ void gaussian_logprob_kernel(const float* __restrict__ z, long long z_b, int n, int B, float* __restrict__ lp) {
  const int b = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float acc = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float v = z[(long long)b * z_b + i];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if (lane == 0) lp[b] += -0.5f * (float)n * 1.8378770664093453f - 0.5f * acc;   // log(2 pi)
}

__global__ void affine_prior_kernel(float* __restrict__ z, long long z_b, const float* __restrict__ ls,
                                    const float* __restrict__ sh, int n, int B, int decode, float* __restrict__ lj) {
  const int b = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float acc = 0.f;
  for (int i = lane; i < n; i += 64) {
    float* zp = z + (long long)b * z_b + i;
    *zp = decode ? (*zp - sh[i]) * expf(-ls[i]) : (*zp) * expf(ls[i]) + sh[i];
    acc += ls[i];
  }
  acc = wave_sum(acc);
  if (lane == 0 && lj) lj[b] += decode ? -acc : acc;
}

__global__ void recon_kernel(const float* __restrict__ xh, const float* __restrict__ x, int n, float* __restrict__ rec) {
  __shared__ float red[16];
  const long long base = (long long)blockIdx.x * n;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float dlt = xh[base + i] - x[base + i];
    acc += dlt * dlt;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) rec[blockIdx.x] = acc;
}

__global__ void elbo_combine_kernel(const float* low, const float* logdet, const float* rec, const float* l1,
                                    const float* pre, float wl, float lam, float wm, int B, float* elbo) {
  const int b = blockIdx.x * TPB + threadIdx.x;
  if (b >= B) return;
  float v = 0.f;
  if (low) v += wl * (low[b] - (logdet ? 0.5f * logdet[b] : 0.f));
  if (rec) v -= lam * rec[b];
  if (l1) v -= wm * l1[b];
  if (pre) v += pre[b];
  elbo[b] = v;
}

}  // namespace

extern "C" {

const char* cmf_version(void) { return "cmf_amd 0.1 (gfx950)"; }

int cmf_pack_weight(const float* w, float* out, int cout, int cin, int taps, int transpose, long long* out_floats,
                    void* stream) {
  if (cout <= 0 || cin <= 0 || (taps != 1 && taps != 9)) return CMF_EINVAL;
  const int ncout = transpose ? cin : cout, ncin = transpose ? cout : cin;
  const int ncog = (ncout + 63) / 64, ncin_pad = (ncin + 7) / 8 * 8;
  const long long total = (long long)ncog * taps * ncin_pad * 64;
  if (out_floats) *out_floats = total;
  if (!out) return 0;
  if (!w) return CMF_EINVAL;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, w, out, cout, cin,
                     taps, transpose, ncin_pad, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_primal_regroup(const float* in, float* out, int B, long long N, int to_grouped, void* stream) {
  if (!in || !out || B <= 0 || B % 16 || N <= 0 || (N + 15) / 16 > 0x7fffffffLL) return CMF_EINVAL;
  hipLaunchKernelGGL(primal_regroup_kernel, dim3((unsigned)((N + 15) / 16), B / 16), dim3(256), 0, (hipStream_t)stream, in,
                     out, B, N, to_grouped);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_acl_primal(float* z, long long z_b, const float* y, long long y_b, const int* zi, const int* si, const int* ti,
                   int n_mod, int B, int decode, float* lj, void* stream) {
  if (!z || !y || !zi || !si || !ti || n_mod <= 0 || B <= 0) return CMF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (lj) {
    hipLaunchKernelGGL(acl_primal_lj, dim3((B + 3) / 4), dim3(TPB), 0, s, z, z_b, y, y_b, zi, si, ti, n_mod, B, decode, lj);
  } else {
    const long long total = (long long)B * n_mod;
    hipLaunchKernelGGL(acl_primal_flat, dim3(nblocks(total)), dim3(TPB), 0, s, z, z_b, y, y_b, zi, si, ti, n_mod, total,
                       decode);
  }
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_acl_tangent(float* t, long long t_b, long long t_r, const float* yt, long long yt_b, long long yt_r, int nc,
                    const float* z, long long z_b, const float* y, long long y_b, const float* g, const int* zi,
                    const int* si, const int* ti, int n_mod, int B, void* stream) {
  if (!t || !z || !y || !zi || !si || !ti || n_mod <= 0 || B <= 0 || nc <= 0 || nc % 4) return CMF_EINVAL;
  if ((t_b | t_r) % 4 || (uintptr_t)t % 16 || (yt && ((yt_b | yt_r) % 4 || (uintptr_t)yt % 16))) return CMF_EINVAL;
  const long long total = (long long)B * n_mod * (nc / 4);
  hipLaunchKernelGGL(acl_tangent_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, t, t_b, t_r, yt, yt_b,
                     yt_r, nc / 4, z, z_b, y, y_b, g, zi, si, ti, n_mod, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_acl_cotangent(float* c, long long c_b, long long c_r, float* yc, long long yc_b, long long yc_r, int nc,
                      const float* z, long long z_b, const float* y, long long y_b, const float* g, const int* zi,
                      const int* si, const int* ti, int n_mod, int B, void* stream) {
  if (!c || !z || !y || !zi || !si || !ti || n_mod <= 0 || B <= 0 || nc <= 0 || nc % 4) return CMF_EINVAL;
  if ((c_b | c_r) % 4 || (uintptr_t)c % 16 || (yc && ((yc_b | yc_r) % 4 || (uintptr_t)yc % 16))) return CMF_EINVAL;
  const long long total = (long long)B * n_mod * (nc / 4);
  hipLaunchKernelGGL(acl_cotangent_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, c, c_b, c_r, yc, yc_b,
                     yc_r, nc / 4, z, z_b, y, y_b, g, zi, si, ti, n_mod, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_acl_primal_backward(float* dx, long long dx_b, const float* z, long long z_b, const float* y, long long y_b, float* dy,
                            const int* zi, const int* si, const int* ti, int n_mod, int B, int decode, const float* dlj,
                            void* stream) {
  if (!dx || !z || !y || !dy || !zi || !si || !ti || n_mod <= 0 || B <= 0) return CMF_EINVAL;
  const long long total = (long long)B * n_mod;
  hipLaunchKernelGGL(acl_primal_backward_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, dx, dx_b, z, z_b, y, y_b,
                     dy, zi, si, ti, n_mod, total, decode, dlj);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_accumulate(float* dst, const float* src, long long n, void* stream) {
  if (!dst || !src || n <= 0) return CMF_EINVAL;
  if (n % 4 || ((uintptr_t)dst | (uintptr_t)src) % 16) {          // odd sizes / unaligned views (small gradient tensors): scalar sweep
    const int blocks = (int)(n / TPB + 1 < 16384 ? n / TPB + 1 : 16384);
    hipLaunchKernelGGL(accumulate_scalar_kernel, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream, dst, src, n);
    CMF_LAUNCH_CHECK();
    return 0;
  }
  const long long n4 = n / 4;
  const int blocks = (int)(n4 / TPB + 1 < 16384 ? n4 / TPB + 1 : 16384);
  hipLaunchKernelGGL(accumulate_kernel, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream, dst, src, n4);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_acl_cross_terms(const float* c, long long c_b, long long c_r, const float* v, long long v_b, long long v_r,
                        const float* yt, long long yt_b, long long yt_r, int nc, const float* z, long long z_b,
                        const float* y, long long y_b, const float* g, const int* zi, const int* si, const int* ti,
                        int n_mod, int B, float* dz, float* dy, float* dg, void* stream) {
  if (!c || !v || !z || !y || !zi || !si || !ti || !dy || n_mod <= 0 || B <= 0 || nc <= 0) return CMF_EINVAL;
  if (yt && (!dz || (g == nullptr) != (dg == nullptr))) return CMF_EINVAL;      // yt null: only dy is written
  const long long n_rows = (long long)B * n_mod;
  hipLaunchKernelGGL(acl_cross_terms_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c, c_b, c_r, v, v_b, v_r,
                     yt, yt_b, yt_r, nc, z, z_b, y, y_b, g, zi, si, ti, n_mod, n_rows, dz, dy, dg);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_gather_primal(const float* in, long long in_b, float* out, long long out_b, const int* idx, int n_out, int B,
                      void* stream) {
  if (!in || !out || !idx || n_out <= 0 || B <= 0) return CMF_EINVAL;
  const long long total = (long long)B * n_out;
  hipLaunchKernelGGL(gather_primal_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, in, in_b, out, out_b,
                     idx, n_out, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_gather_tangent(const float* in, long long in_b, long long in_r, float* out, long long out_b, long long out_r,
                       const int* idx, int n_out, int nc, int B, void* stream) {
  if (!in || !out || !idx || n_out <= 0 || B <= 0 || nc <= 0 || nc % 4) return CMF_EINVAL;
  if ((in_b | in_r | out_b | out_r) % 4 || (uintptr_t)in % 16 || (uintptr_t)out % 16) return CMF_EINVAL;
  const long long total = (long long)B * n_out * (nc / 4);
  hipLaunchKernelGGL(gather_tangent_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, in, in_b, in_r, out,
                     out_b, out_r, idx, n_out, nc / 4, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_seed_tangent(float* t, long long t_b, long long t_r, const int* col_of, int n_rows, int nc, const float* eps,
                     int d, int S, int B, void* stream) {
  if (!t || !col_of || n_rows <= 0 || B <= 0 || nc <= 0 || nc % 4 || d <= 0) return CMF_EINVAL;
  if ((t_b | t_r) % 4 || (uintptr_t)t % 16 || (eps && (S <= 0 || S > nc)) || (!eps && d > nc)) return CMF_EINVAL;
  const long long total = (long long)B * n_rows * (nc / 4);
  hipLaunchKernelGGL(seed_tangent_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, t, t_b, t_r, col_of,
                     n_rows, nc / 4, eps, d, S, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_expand_columns(const float* in, int nc_in, float* out, int nc_out, const int* colmap, long long rows, void* stream) {
  if (!in || !out || !colmap || nc_in <= 0 || nc_out <= 0 || nc_out % 4 || rows <= 0 || (uintptr_t)out % 16) return CMF_EINVAL;
  const long long total = rows * (nc_out / 4);
  hipLaunchKernelGGL(expand_columns_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, in, nc_in, out, nc_out / 4,
                     colmap, total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_prehead(const float* x, const float* u, float* y, float* lj, float a, float c, int logit, int n, int B,
                void* stream) {
  if (!x || !y || n <= 0 || B <= 0 || a == 0.f) return CMF_EINVAL;
  hipLaunchKernelGGL(prehead_kernel, dim3(B), dim3(TPB), 0, (hipStream_t)stream, x, u, y, lj, a, c, logit, n);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_prehead_inverse(const float* y, float* x, float a, float c, int logit, long long n_total, void* stream) {
  if (!x || !y || n_total <= 0 || a == 0.f) return CMF_EINVAL;
  hipLaunchKernelGGL(prehead_inverse_kernel, dim3(nblocks(n_total)), dim3(TPB), 0, (hipStream_t)stream, y, x, a, c, logit,
                     n_total);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_gaussian_logprob(const float* z, long long z_b, int n, int B, float* lp, void* stream) {
  if (!z || !lp || n <= 0 || B <= 0) return CMF_EINVAL;
  hipLaunchKernelGGL(gaussian_logprob_kernel, dim3((B + 3) / 4), dim3(TPB), 0, (hipStream_t)stream, z, z_b, n, B, lp);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_affine_prior(float* z, long long z_b, const float* log_scale, const float* shift, int n, int B, int decode,
                     float* lj, void* stream) {
  if (!z || !log_scale || !shift || n <= 0 || B <= 0) return CMF_EINVAL;
  hipLaunchKernelGGL(affine_prior_kernel, dim3((B + 3) / 4), dim3(TPB), 0, (hipStream_t)stream, z, z_b, log_scale, shift,
                     n, B, decode, lj);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_recon_sqerr(const float* xh, const float* x, int n, int B, float* rec, void* stream) {
  if (!xh || !x || !rec || n <= 0 || B <= 0) return CMF_EINVAL;
  hipLaunchKernelGGL(recon_kernel, dim3(B), dim3(TPB), 0, (hipStream_t)stream, xh, x, n, rec);
  CMF_LAUNCH_CHECK();
  return 0;
}

int cmf_elbo_combine(const float* low, const float* logdet, const float* rec, const float* l1, const float* pre,
                     float wl, float lam, float wm, int B, float* elbo, void* stream) {
  if (!elbo || B <= 0) return CMF_EINVAL;
  hipLaunchKernelGGL(elbo_combine_kernel, dim3(nblocks(B)), dim3(TPB), 0, (hipStream_t)stream, low, logdet, rec, l1, pre,
                     wl, lam, wm, B, elbo);
  CMF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// out[0] = max(out[0], max |x|): the input-range word of cmf_conv_tangent_f16x3 (amax_in) for a tensor no conv of that
// family produced (the coupler's first activation).  Non-negative floats order like their bit patterns: integer atomicMax.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = x4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  // ONE atomic per workgroup: with one per wave (8192 on a 100 MB activation) the launch took 100 us -- a single word takes ~12 ns
  // per atomic -- against ~20 us for the sweep itself
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(out), __builtin_bit_cast(int, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
}  // namespace

extern "C" int cmf_absmax(const float* x, long long n, float* out, void* stream) {
  if (!x || !out || n <= 0 || (uintptr_t)x % 16) return CMF_EINVAL;
  const long long blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks)), dim3(256), 0, (hipStream_t)stream, x, n, out);
  CMF_LAUNCH_CHECK();
  return 0;
}
