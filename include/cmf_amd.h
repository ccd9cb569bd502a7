/* cmf_amd.h -- C ABI of libcmf_amd.so: the MI355X (gfx950) kernels of the non-square-flow
 * log-density path of k-flouris/cmf.
 *
 * The reference is pure Python/PyTorch and has no FFI; each entry point below replaces a group of
 * ATen dispatches issued by the reference functions cited next to it (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (or int32 where stated); the caller owns all
 *     buffers (no hidden allocations), sizes/strides are in ELEMENTS, not bytes;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are asynchronous;
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     CMF_E* code for invalid arguments.  Nothing throws across this boundary;
 *   - every call is re-entrant and uses the calling thread's CURRENT device (one Python thread per GPU in one process is
 *     fine).  The only mutable global is a lock-free per-(device, kernel) memo of launch attributes already set
 *     (csrc/runtime.hip: dynamic-LDS limit, CU count); entries are idempotent and never removed.
 *
 * Data layouts
 *   primal tensor   P(b, r)        = p[b*p_b + r*p_r]                  (B, N) row-major, = torch layout
 *   tangent tensor  T(b, r, col)   = t[b*t_b + r*t_r + col]            col = Jacobian column, NC = ceil16(d)
 *       image nets : t_b = N*NC, t_r = NC          ("J panel" per sample: (B, C, H, W, NC))
 *       MLP nets   : t_b = NC,   t_r = B*NC        (feature-major: (F, B, NC))
 */
#ifndef CMF_AMD_H
#define CMF_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define CMF_EINVAL (-1)   /* bad shape / stride / mode                                   */
#define CMF_ERANGE (-2)   /* size exceeds what the kernel's index arithmetic supports    */

/* factor modes: how the per-(channel,pixel) multiplier applied to a conv/linear INPUT is derived
 * from the primal tensor `f` (jvp_layers.py:38-47 activation rules, applied on load) */
#define CMF_F_NONE 0      /* no multiplier                                               */
#define CMF_F_RELU 1      /* (f > 0)           relu'  : jvp_layers.py:40                 */
#define CMF_F_TANH 2      /* 1 - f*f           tanh'  with f = tanh output: :42-44       */
#define CMF_F_RAW  3      /* f itself          checkerboard mask: acl.py:54              */
#define CMF_F_SELF_RELU 4 /* relu of the input itself, elementwise (primal data carried in the column slots) */
#define CMF_F_RELU_BITS 5 /* relu' from a BIT MASK (cmf_conv_tangent_bf16x3 only): f points to bytes, one per (sample, pixel,
                             8-channel octet): bit j of f[np*f_np + px*(cin/8) + ci/8] = [activation(ci = 8*(ci/8)+j) > 0];
                             f_np in BYTES.  Written by cmf_conv_tangent's mask_out (below): 1/32 of the bytes of the
                             activation tensor, one byte load per loader thread and chunk instead of eight dwords */

/* primal epilogues */
#define CMF_O_NONE  0
#define CMF_O_TANH  1     /* y = tanh(v)                         (MLP hidden layer)                  */
#define CMF_O_STANH 2     /* y = w*tanh(v)+b2, g = w*(1-tanh^2)  (ScaledTanh2dModule networks.py:96-113) */

const char* cmf_version(void);

/* ---------------------------------------------------------------------------------------------
 * Weight packing.  w: [cout][cin][kh][kw] (nn.Conv2d / nn.Linear with kh=kw=1).
 * out: [ncog][taps][cin_pad][64] with cin_pad = ceil8(cin), ncog = ceil(cout/64); zero padded.
 * transpose != 0 packs the adjoint operator (cout<->cin swapped, taps flipped) used by the
 * reverse-mode sweep of the Hutchinson path (non_square.py:190-201).
 * Returns the number of floats written through *out_floats when out == NULL (size query).       */
int cmf_pack_weight(const float* w, float* out, int cout, int cin, int taps, int transpose,
                    long long* out_floats, void* stream);

/* Every stale pack of a model in ONE launch (a training step re-packs ~780 weights: round 3).  `table` is a DEVICE array of n
 * descriptors; entry k writes `total` elements of layout `kind` (0: cmf_pack_weight's floats, 1: cmf_pack_weight_bf16x3_t's bf16
 * halves, 2: cmf_pack_weight_f16x3's fp16 halves followed by its 16-byte trailer) of the weight `w` to `out`, exactly as the single-weight entry points do.  For kind 1 with transpose != 0, cout / cin are
 * the ADJOINT operator's (already swapped), as cmf_pack_weight_bf16x3_t expects them. */
typedef struct {
  const float* w; void* out; long long total;
  int cout, cin, taps, transpose, kind, reserved;
} cmf_pack_desc;
int cmf_pack_weights_batched(const cmf_pack_desc* table, int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Tangent convolution / linear layer on fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * Replaces, for all d Jacobian columns at once, the tangent half of get_conv2d_jvp / get_linear_jvp
 * (jvp_layers.py:49-64) fused with the activation rule of the preceding layer (:38-47) and the
 * residual add of ResidualBlock.jvp (networks.py:62-79):
 *     y(np, co, px, :) = sum_{ci,tap} W[co][ci][tap] * F(np, ci, px+tap) * x(np, ci, px+tap, :)  [+ r(np, co, px, :)]
 * taps == 9: 3x3, stride 1, zero padding 1 over an H x W image; taps == 1: 1x1 over H*W flat pixels
 * (an MLP layer is the 1x1 case with pixels = batch samples).                                    */
typedef struct {
  const float* x; long long x_np, x_ci, x_px;   /* input tangent tensor, NC contiguous columns       */
  const float* f; long long f_np, f_ci, f_px;   /* primal tensor the factor is derived from (or NULL) */
  int fmode;                                    /* CMF_F_*                                            */
  const float* w;                               /* packed by cmf_pack_weight                          */
  float* y;       long long y_np, y_co, y_px;
  const float* r; long long r_np, r_co, r_px;   /* residual (same shape as y) or NULL                 */
  int np, cin, cout, H, W, nc, taps;
  const float* bias;                            /* per-output-channel constant added to every column, or NULL  */
  int f_group;                                  /* >1: the factor tensor is sample-grouped: element (np, ci, px)
                                                   lives at f[(np / f_group)*f_np + ci*f_ci + px*f_px + np % f_group]
                                                   (primal activations kept in the "16 samples as columns" layout) */
  long long x_sl, y_sl, r_sl;                   /* element stride between 16-column slices: column col of x lives at
                                                   (col / 16)*x_sl + col % 16 (likewise y, r).  0 means 16 = plain
                                                   contiguous columns.  The slice-major hidden layout
                                                   [pixel][slice][channel][16] (x_px = C*nc, x_sl = C*16, x_ci = 16)
                                                   makes a wave's 16 channels x 16 columns one contiguous KiB        */
  const float* fo; long long fo_np, fo_co, fo_px; int fomode;
                                                /* OUTPUT-side factor (cmf_conv_tangent only; the split kernel rejects
                                                   it): y = Fo(np, co, px) * conv(F * x) + bias + r, Fo derived from fo
                                                   by fomode like F from f (NONE / RELU / TANH / RAW; SELF_RELU: fo is laid
                                                   out like y, columns included, and Fo = [fo > 0] per column -- primal
                                                   backward with 16 samples in the column slots).  This is what the
                                                   reverse (cotangent) sweep needs: the adjoint of "mask, then conv" is
                                                   "transposed conv, then mask" (weights from cmf_pack_weight(transpose=1)) */
  void* mask_out; long long mask_np;            /* cmf_conv_tangent only, cout % 16 == 0: also write the sign bits of the
                                                   stored values, [y > 0], in the CMF_F_RELU_BITS layout of the NEXT conv:
                                                   byte (sample*mask_np + px*(cout/8) + co/8), sample = np*nc + column
                                                   (primal pass: 16 samples in the column slots); NULL = off            */
  const float* amax_in; float* amax_out;        /* cmf_conv_tangent_f16x3 only (else ignored).  amax_in: one device float >= the largest
                                                   |x| the launch reads (NULL or 0: unknown -> no input scaling); the kernel multiplies x
                                                   by the power of two that puts that maximum in [2^13, 2^14) before the fp16 split, so no
                                                   input overflows fp16 and small ones keep their low half.  amax_out: one device float,
                                                   atomically raised to max(y) over the stored values (the caller zeroes it: the next
                                                   conv's amax_in, whose relu-on-load ignores negative values); NULL = off              */
  int live;                                     /* cmf_conv_tangent_bf16x3 only, fmode RELU / RELU_BITS, whole 64-channel groups, no bias / fo:
                                                   CHECKERBOARD output.  0 = every pixel.  1 / 2: only the pixels with (row + col) % 2 ==
                                                   live - 1 are computed -- the (1 - mask) pixels of Checkerboard2dAffineCouplingBijection
                                                   (acl.py:48-66, :68-78), all a coupler network's LAST hidden conv is ever read at, since the
                                                   1x1 conv behind it is pointwise -- and y is COMPACT: pixel (row, col) is stored at pixel
                                                   index row*(W/2) + col/2 (y_px = stride between compact pixels).  x, f and the residual r
                                                   keep the full H x W image; r is read at the live pixels.                              */
} cmf_conv_tangent_args;
/* (A launch with taps == 9, cin <= 2, cout % 64 == 0, no residual / bias / output factor / mask_out and fmode NONE or RAW -- the
 * first conv of a coupler network, networks.py:40-47 -- is an HBM write stream and runs on a VALU kernel instead of the MFMA one:
 * same contract, fp32 FMAs.)                                                                                                     */
int cmf_conv_tangent(const cmf_conv_tangent_args* a, void* stream);

/* Split-precision variant of cmf_conv_tangent for taps == 9, cin % 32 == 0, (W % 14 == 0 and H % 2 == 0) or (W % 8 == 0 and H % 4 == 0), and
 * cout % 64 == 0 or cout == 32 (anything else: CMF_EINVAL, use cmf_conv_tangent): operands are split
 * v = hi + lo (bf16 each) and multiplied as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation (fp32-grade result, ~2^-16 relative per product).  `w` must come from
 * cmf_pack_weight_bf16x3 (out == NULL: size query in bytes through *out_bytes).  Output-side factor: only with
 * fmode CMF_F_NONE, no residual, cout % 64 == 0 and fomode CMF_F_RELU_BITS -- fo is then a relu' bit mask over the OUTPUT
 * channels (byte np*fo_np + px*(cout/8) + co/8, fo_np in bytes; written by cmf_relu_bits or mask_out): the transposed convs of
 * the reverse sweep.  A residual is accepted together with that bit mask only IN PLACE (r == y with y's strides, no bias):  y <- y + Fo . conv(x)  -- the skip connection of the reverse sweep; entries the mask switches off are not written at all
 * (they keep their bits).  mask_out is not supported. */
int cmf_pack_weight_bf16x3(const float* w, void* out, int cout, int cin, long long* out_bytes, void* stream);
/* the same pack of the ADJOINT operator straight from the layer's weight: transpose != 0 reads w as [cin][cout][3][3] and packs
 * [cout][cin][tap] = w[ci][co][8 - tap] (channels swapped, taps flipped: cmf_pack_weight's transpose for the split kernel) */
int cmf_pack_weight_bf16x3_t(const float* w, void* out, int cout, int cin, int transpose, long long* out_bytes, void* stream);
int cmf_conv_tangent_bf16x3(const cmf_conv_tangent_args* a, void* stream);

/* fp16 split-precision variant for the PRIMAL hidden convs of a ResNet coupler (networks.py:50-60 with 16 samples in the column
 * slots): fmode CMF_F_SELF_RELU, taps == 9, cin % 32 == 0, cout % 64 == 0, tiles as cmf_conv_tangent_bf16x3, no output factor.
 * Operands are split v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (11 + 11 significant bits) and multiplied as
 * hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation: ~2^-22 relative per product against bf16x3's 2^-16, so
 * the relu masks taken from these activations flip as rarely as with exact fp32 products (profiles/r04_primal_precision_study.txt)
 * at the bf16 MFMA rate.  fp16's narrow exponent is handled by exact power-of-two scales: the weights are packed times 2^k with
 * max |w| 2^k in [2^11, 2^12) (cmf_pack_weight_f16x3: same layout as cmf_pack_weight_bf16x3_t plus a 16-byte trailer
 * {2^k, 2^-k, 0, 0}; the max is taken on the device, no host synchronisation), the inputs times the power of two derived from
 * amax_in; the epilogue undoes both (exact), the residual enters scaled.  mask_out IS supported (the sign bits of the stored
 * values, as cmf_conv_tangent writes them).  `w` must come from cmf_pack_weight_f16x3.
 * BACKWARD form (the data-gradient convs of the same blocks under loss.backward(), trainer.py:213): fmode CMF_F_NONE, `w` the pack of
 * the adjoint operator (transpose = 1), fo != NULL with fomode CMF_F_SELF_RELU -- a float tensor laid out like y (strides fo_np /
 * fo_co / fo_px, columns included) -- and  y = [fo > 0] * conv(x) + r  per column; no bias, no mask_out; *amax_out = max |y|.          */
int cmf_pack_weight_f16x3(const float* w, void* out, int cout, int cin, int transpose, long long* out_bytes, void* stream);
int cmf_conv_tangent_f16x3(const cmf_conv_tangent_args* a, void* stream);
/* the same with the work-item size chosen by the caller: item_channels = 64 (a workgroup's item is a pixel tile x 16 samples x 64
 * output channels), 32 (half of a 64-channel group per item: twice the items, for launches that would leave most CUs without one)
 * or 0 = cmf_conv_tangent_f16x3's own choice (32 when the 64-channel items number at most half the CUs).  Results are
 * bit-identical between the two sizes.  Launches on 4 x 8 pixel tiles (16- / 32-wide images) always run 32-channel items: the
 * 64-channel form does not fit the 256 VGPRs of a wave there (it spilled to scratch).                                          */
int cmf_conv_tangent_f16x3_item(const cmf_conv_tangent_args* a, int item_channels, void* stream);
/* out[0] = max(out[0], max_i |x[i]|) over n floats (out is NOT cleared: the caller zeroes it or chains several tensors).   */
int cmf_absmax(const float* x, long long n, float* out, void* stream);

/* Weight gradient of the tangent convolution (training; the reference gets it from autograd through
 * get_conv2d_jvp / get_linear_jvp, jvp_layers.py:49-64, under loss.backward(), trainer.py:213):
 *     dw[co][ci][tap] += sum_{np, px, col} gy(np, co, px, col) * F(np, ci, px+tap) * x(np, ci, px+tap, col)
 * `a` describes the FORWARD launch: x, f, fmode (NONE / RELU / TANH / RAW, or SELF_RELU: x's own relu, elementwise), f_group, np, cin, cout, H, W, nc, taps and the
 * strides are read from it; gy is the cotangent of y and is addressed like y (y_np, y_co, y_px, y_sl); a->w, y, r, bias,
 * fo and mask_out are ignored.  dw: [cout][cin][taps] fp32 (the nn.Conv2d / nn.Linear weight layout), accumulated into.
 * ws: caller-owned workspace of cmf_conv_tangent_wgrad_ws(a) bytes (partial sums, reduced in a fixed order).         */
long long cmf_conv_tangent_wgrad_ws(const cmf_conv_tangent_args* a);
int cmf_conv_tangent_wgrad(const cmf_conv_tangent_args* a, const float* gy, float* dw, float* ws, long long ws_bytes,
                           void* stream);
/* Split-precision variant (hi*hi + hi*lo + lo*hi on bf16 MFMA, fp32 accumulation, like cmf_conv_tangent_bf16x3) for taps == 9,
 * cin % 64 == 0, cout % 64 == 0, nc % 32 == 0, fmode CMF_F_NONE, CMF_F_RELU (float factor tensor, f_group <= 1), CMF_F_RELU_BITS (f = the
 * relu' bit mask of cmf_conv_tangent_bf16x3's input factor: byte np*f_np + px*(cin/8) + ci/8, f_np in bytes) or CMF_F_SELF_RELU;
 * anything else: CMF_EINVAL, use cmf_conv_tangent_wgrad.  Same arguments and workspace.                                      */
int cmf_conv_tangent_wgrad_bf16x3(const cmf_conv_tangent_args* a, const float* gy, float* dw, float* ws, long long ws_bytes,
                                  void* stream);
/* nprob <= CMF_WGRAD_MAX_BATCH problems of ONE shape in one launch: x[p], gy[p], dw[p] are HOST arrays of device pointers (a->x is
 * ignored; everything else -- strides, np, nc, H, W, fmode NONE or SELF_RELU -- is read from `a` and shared).  The 256 persistent
 * workgroups are split evenly over the problems.  Made for the primal weight gradients of a coupler's hidden convs at a training
 * shard's few sample groups (trainer.py:213 differentiates 16 such convs per coupler): 28 - 56 image rows per problem leave most of the
 * chip idle when launched one by one.  Same workspace as the single-problem call. */
#define CMF_WGRAD_MAX_BATCH 16
int cmf_conv_tangent_wgrad_bf16x3_batched(const cmf_conv_tangent_args* a, int nprob, const float* const* x, const float* const* gy,
                                          float* const* dw, float* ws, long long ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Primal convolution / linear layer on fp32 MFMA: nn.Conv2d / nn.Linear forward of the coupler
 * networks (networks.py:50-60, :103-106, :206-224) with the preceding activation fused on load and
 * bias / residual / tanh epilogues:
 *     v(b, co, px) = sum W[co][ci][tap] * in(b, ci, px+tap) + bias[co] [+ r(b, co, px)]
 *     in = x | relu(x) | x*f        (imode CMF_F_NONE | CMF_F_RELU | CMF_F_RAW with mask f(ci,px))
 * omode CMF_O_NONE: y = v;  CMF_O_TANH: y = tanh(v);  CMF_O_STANH: y = sw[co]*tanh(v)+sb[co] and
 * g = sw[co]*(1-tanh(v)^2) (the tangent multiplier of ScaledTanh2dModule.jvp, networks.py:108-113).
 * Element (b, c, px) of a tensor t lives at t[b*t_b + c*t_c + px*t_px].                           */
typedef struct {
  const float* x; long long x_b, x_c, x_px;
  const float* f; long long f_c, f_px;          /* CMF_F_RAW mask (shared over the batch) or NULL     */
  int imode;
  const float* w; const float* bias;            /* packed weights; bias[cout] or NULL                 */
  const float* sw; const float* sb;             /* CMF_O_STANH per-channel scale / offset             */
  float* y;       long long y_b, y_c, y_px;
  float* g;                                     /* CMF_O_STANH derivative factor (strides of y)|NULL  */
  const float* r; long long r_b, r_c, r_px;
  int omode;
  int B, cin, cout, H, W, taps;
} cmf_conv_primal_args;
int cmf_conv_primal(const cmf_conv_primal_args* a, void* stream);

/* (B, N) row-major  <->  (B/G, N, G) sample-grouped layout, G = 16: the primal hidden activations of the ResNet
 * couplers run through the tangent kernels with 16 samples in the 16 column slots (B % 16 == 0).             */
int cmf_primal_regroup(const float* in, float* out, int B, long long N, int to_grouped, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Affine coupling transforms (acl.py:43-66, :101-146), in place on the primal / tangent tensors.
 * The three mask types (checkerboard acl.py:68-78, split-channel :185-189, alternating :207-214)
 * are all expressed by index maps over the n_mod modified elements of a sample:
 *   zi[e]: element of z that is modified; ti[e] / si[e]: elements of the coupler output holding its
 *   shift / log-scale (couplers.py:52-59 chunking).  Pass-through elements are untouched.
 * encode (x_to_z): z = (x + t) * exp(s);  if lj != NULL: lj[b] += sum_e s       (log-jac, acl.py:22-23)
 * decode (z_to_x): x = z * exp(-s) - t                                                             */
int cmf_acl_primal(float* z, long long z_b, const float* y, long long y_b, const int* zi, const int* si,
                   const int* ti, int n_mod, int B, int decode, float* lj, void* stream);
/* tangent: T(b, zi[e], :) = exp(-s) * (T(b, zi[e], :) - z_old * sdot) - tdot   (acl.py:61-64, :137-144)
 *   sdot = gs * yt(b, si[e], :), tdot = gt * yt(b, ti[e], :), gs/gt = g(b, si[e]) / g(b, ti[e]) or 1 when
 *   g == NULL; z_old = z(b, zi[e]) BEFORE cmf_acl_primal decode is applied; s from y.
 * yt == NULL: the network's tangent is identically zero (its pass-through input is structurally zero, e.g. the channels
 *   SplitDensity.pad_inputs appended, split.py:50-52): T(b, zi[e], :) *= exp(-s).
 * y_b == 0 (here and in cmf_acl_primal): one network output shared by every sample (same situation: the input is zero).  */
/* cotangent (adjoint of cmf_acl_tangent, for J^T w / the reverse sweep): with c = C(b, zi[e], :) on entry,
 *   yc(b, ti[e], :) = -gt * c,   yc(b, si[e], :) = -(exp(-s) * z_old * gs) * c,   C(b, zi[e], :) = exp(-s) * c.
 * yc rows that no element maps to must be zero (the caller clears yc).
 * yc == NULL: only C is updated (the cotangent of the network's input lands on rows that are dropped).   */
int cmf_acl_cotangent(float* c, long long c_b, long long c_r, float* yc, long long yc_b, long long yc_r, int nc,
                      const float* z, long long z_b, const float* y, long long y_b, const float* g, const int* zi,
                      const int* si, const int* ti, int n_mod, int B, void* stream);
/* Primal backward of the coupling update (training), in place on the primal cotangent dx (z-shaped; pass-through elements keep
 * their value), accumulating the cotangent of the network output into dy (y-shaped):
 *   decode != 0 (x_mod = z_mod e^{-s} - t, `z` = the tensor BEFORE the update, acl.py:57-66):
 *       dx[zi] *= e^{-s};  dy[si] -= dx z_mod e^{-s};  dy[ti] -= dx
 *   decode == 0 (z_mod = (x_mod + t) e^{s}, `z` = the layer INPUT x, acl.py:43-46):
 *       dx[zi] *= e^{s};   dy[ti] += dz e^{s};  dy[si] += dz (x_mod + t) e^{s} + dlj[b]   (dlj: cotangent of the log-jacobian, or NULL) */
int cmf_acl_primal_backward(float* dx, long long dx_b, const float* z, long long z_b, const float* y, long long y_b, float* dy,
                            const int* zi, const int* si, const int* ti, int n_mod, int B, int decode, const float* dlj,
                            void* stream);
/* Cross terms of the coupling update for training (autograd through acl.py:48-66, :113-146 in the reference): the tangent
 * update  out(b, zi[e], :) = es (v - zo gs sd) - gt td  of cmf_acl_tangent also depends on primal values.  Given the
 * cotangent c of `out` (rows zi[e], BEFORE cmf_acl_cotangent rewrites them), the saved input rows v (compact: row e at
 * v + b*v_b + e*v_r) and the network's raw tangent yt, accumulates (+=) the column reductions
 *   dy[b][si[e]] += d/ds,   dz[b][zi[e]] += d/d zo,   dg[b][si[e]] += d/d gs,   dg[b][ti[e]] += d/d gt
 * (dg and g both NULL for networks without the ScaledTanh output stage).
 * yt == NULL (network tangent identically zero): only dy[b][si[e]] -= sum_col c es v is accumulated; dz, dg unused.  */
int cmf_acl_cross_terms(const float* c, long long c_b, long long c_r, const float* v, long long v_b, long long v_r,
                        const float* yt, long long yt_b, long long yt_r, int nc, const float* z, long long z_b,
                        const float* y, long long y_b, const float* g, const int* zi, const int* si, const int* ti,
                        int n_mod, int B, float* dz, float* dy, float* dg, void* stream);
int cmf_acl_tangent(float* t, long long t_b, long long t_r, const float* yt, long long yt_b, long long yt_r,
                    int nc, const float* z, long long z_b, const float* y, long long y_b, const float* g,
                    const int* zi, const int* si, const int* ti, int n_mod, int B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Index-map moves: squeeze / unsqueeze (reshaping.py:89-114), split padding (split.py:50-52), tail
 * gather / scatter (non_square.py:381-384, :397-404), layout changes.
 * out(b, r) = idx[r] >= 0 ? in(b, idx[r]) : 0,   r < n_out.                                        */
int cmf_gather_primal(const float* in, long long in_b, float* out, long long out_b, const int* idx,
                      int n_out, int B, void* stream);
int cmf_gather_tangent(const float* in, long long in_b, long long in_r, float* out, long long out_b,
                       long long out_r, const int* idx, int n_out, int nc, int B, void* stream);
/* Column expansion of a contiguous tangent tensor of `rows` rows: out(row, c) = colmap[c] >= 0 ? in(row, colmap[c]) : 0 for c <
 * nc_out (nc_out % 4 == 0).  The first coupling layer of the decode sweep sees one-hot seed tangents (non_square.py:303-304): a
 * Jacobian column whose seed element is not among the elements that layer's network reads has an identically zero network tangent
 * there, so the network runs on the other columns only (nc_in of them, packed) and its output is expanded back with this.    */
int cmf_expand_columns(const float* in, int nc_in, float* out, int nc_out, const int* colmap, long long rows, void* stream);
/* Seed tangents at the tail (non_square.py:303-304, :406-410): col_of[r] = Jacobian column whose unit
 * vector lands on element r (or -1).  eps == NULL: T(b, r, c) = (col_of[r] == c)  (identity seed, exact
 * path); else T(b, r, c) = eps[b][col_of[r]][c] for c < S (Hutchinson probes, non_square.py:204-215).  */
int cmf_seed_tangent(float* t, long long t_b, long long t_r, const int* col_of, int n_rows, int nc,
                     const float* eps, int d, int S, int B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused J^T J Gram (MFMA) + Cholesky + log-det + metric L1 terms: replaces
 * non_square.py:307-308 (stack + bmm), :280-294 (cholesky, 2*sum log diag) and :87-100 (g_kk / g_ij).
 *   jtj   [B][d][d]  out, the (possibly jittered) Gram matrix;  logdet, l1_off, l1_diag: [B]
 *   info  [B] int32: 0 ok, k+1 = non-positive / non-finite pivot at column k
 *   fail  int32[8]: fail[a] != 0 iff attempt a failed for ANY sample (whole-batch retry, :284-288)
 * cmf_gram_cholesky is attempt 0.  cmf_cholesky_retry(attempt = a >= 1) returns immediately on the
 * device unless fail[a-1] != 0; otherwise it adds eps * 10^(a-1) to the diagonal of EVERY sample's
 * jtj (in place) and refactorises.  The host may enqueue all retries without synchronising.       */
int cmf_gram_cholesky(const float* t, long long t_b, long long t_r, int n_rows, int nc, int d, int B,
                      float* jtj, float* logdet, float* l1_off, float* l1_diag, int* info, int* fail,
                      void* stream);
int cmf_cholesky_retry(float* jtj, int d, int B, int attempt, float eps0, float* logdet, float* l1_diag,
                       int* info, int* fail, void* stream);
/* Reverse of the head above for training (autograd through non_square.py:307-308, :280-294, :87-100):
 *   dt(b, r, :) = 2 * t(b, r, :) * (g_logdet[b] * jtj_b^-1 + g_l1off[b] * sign(jtj_b)[i != j]
 *                                   + g_l1diag[b] * sign(jtj_b)[i == j])
 * jtj is the matrix cmf_gram_cholesky / cmf_cholesky_retry left behind; g_* are [B] or NULL (= 0);
 * dt has nc columns per row like t (columns >= d are written as 0) and may not alias t.            */
int cmf_gram_backward(const float* t, long long t_b, long long t_r, int n_rows, int nc, int d, int B,
                      const float* jtj, const float* g_logdet, const float* g_l1off, const float* g_l1diag,
                      float* dt, long long dt_b, long long dt_r, void* stream);
/* The same product for an EXPLICIT cotangent m [B][d][d] of the Gram matrix (not necessarily symmetric):
 * dt = t (m + m^T).  Training on the Hutchinson surrogate (non_square.py:203-258): m = mean_s u_s eps_s^T, u detached.   */
int cmf_gram_backward_matrix(const float* t, long long t_b, long long t_r, int n_rows, int nc, int d, int B,
                             const float* m, float* dt, long long dt_b, long long dt_r, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Small per-sample reductions / elementwise maps.                                                 */
/* pre-head chain (wrapper.py:28-30, math.py:41-105): y = a*(x + u) + c; if logit: y = log(y)-log(1-y) and
 * lj[b] = n*log|a| + sum(-log(yc) - log(1-yc)), yc = clamp(a*(x+u)+c, 1e-7, 1-1e-7).  u may be NULL.     */
int cmf_prehead(const float* x, const float* u, float* y, float* lj, float a, float c, int logit, int n,
                int B, void* stream);
/* inverse of the pre-head chain for sampling (math.py:45-46, :83-84, :100-101): x = (sigmoid(y) - c) / a */
int cmf_prehead_inverse(const float* y, float* x, float a, float c, int logit, long long n_total, void* stream);
/* lp[b] += -n/2 log(2 pi) - 1/2 sum z^2      (gaussian.py:9-22 with mean 0, stddev 1)              */
int cmf_gaussian_logprob(const float* z, long long z_b, int n, int B, float* lp, void* stream);
/* 2-D prior AffineBijection (affine.py:24-34): encode u = z*exp(ls)+sh, lj[b] += sum ls; decode inverse */
int cmf_affine_prior(float* z, long long z_b, const float* log_scale, const float* shift, int n, int B,
                     int decode, float* lj, void* stream);
/* rec[b] = sum_r (xh(b,r) - x(b,r))^2        (non_square.py:111-114)                               */
int cmf_recon_sqerr(const float* xh, const float* x, int n, int B, float* rec, void* stream);
/* elbo[b] = wl*(low[b] - logdet[b]/2) - lam*rec[b] - wm*l1[b] + pre[b]; NULL inputs count as 0
 * (non_square.py:85, :126-129 and exact.py:27 for the pre-head log-jacobians)                      */
int cmf_elbo_combine(const float* low, const float* logdet, const float* rec, const float* l1, const float* pre,
                     float wl, float lam, float wm, int B, float* elbo, void* stream);
/* Hutchinson log-det surrogate (non_square.py:203-258) against the EXPLICIT Gram matrix produced by
 * cmf_gram_cholesky (see hutch_cg.hip for why the explicit form is the cheaper one on this hardware):
 *   w(b,:,s) = G(b) eps(b,:,s);  u = CG(G, eps) with x0 = 0, unit-normalised right-hand sides, at least
 *   min_iter and at most max_iter iterations, stopping a sample when the mean over its S probes of the
 *   relative residual 2-norm drops below tol;  val[b] = mean_s sum_k u*w.   eps, u, w: [B][d][S]; S <= 128
 *   (probes beyond 16 run in chunks of 16 with the stopping rule applied per chunk; iters[b] = the slowest chunk).
 * The reference's solver (gpytorch linear_cg @ fc2053b) is un-vendored: CG iterates are parity-unpinned. */
int cmf_hutch_cg(const float* jtj, const float* eps, int d, int S, int B, int max_iter, int min_iter, float tol,
                 float* u, float* w, float* val, int* iters, void* stream);
/* Metric term on the Hutchinson product W = (J^T J) eps, [B][d][S] with S == d -- the third return value of
 * non_square.py:253-258 fed to :87-100:  l1_diag[b] = sum_k |W_kk|,  l1_off[b] = sum_{i != j} |W_ij|
 * (the reference's masked_select(~eye).view(B, d(d-1)) exists only for S == d: l1_off must be NULL otherwise; the diagonal branch,
 * torch.diagonal of the (B, d, S) product at :87-92, is valid for any S and sums min(d, S) entries).  Either output may be NULL. */
int cmf_hutch_metric(const float* w, int d, int S, int B, float* l1_off, float* l1_diag, void* stream);
/* Cotangent of the Gram matrix for the train-mode Hutchinson objective, u detached (non_square.py:236-247):
 *   M(b) = g_val[b]/S sum_s u_s eps_s^T + sum_s (g_off[b] [i != s] + g_diag[b] [i == s]) sign(W_is) e_i eps_s^T,
 * [B][d][d]; feed it to cmf_gram_backward_matrix (dJ = J (M + M^T)).  g_val / g_off / g_diag: [B] or NULL; the metric
 * terms need w; g_off needs S == d. */
int cmf_hutch_cotangent(const float* u, const float* eps, const float* w, int d, int S, int B, const float* g_val,
                        const float* g_off, const float* g_diag, float* M, void* stream);
/* Low-rank form of the same cotangent for S << d (non_square.py:241-256 builds its graph only through J^T J eps for the S
 * probes): with V = [u_1..u_S | eps_1..eps_S | e_0..e_{K-1}] (K = min(d, S) when g_diag is given, else 0; n = 2S + K columns) and
 * P = J V from ONE n-column tangent sweep, the objective is a sum of inner products of columns of P and its cotangent with
 * respect to P^T P is  cmat[b][s][S+s] = g_val[b]/S,  cmat[b][2S+k][S+k] = g_diag[b] sign(W_kk)  ([B][n][n], zero elsewhere);
 * cmf_gram_backward_matrix(P, cmat) then yields the cotangent of P.  w [B][d][S] is read only with g_diag. */
int cmf_hutch_lowrank_cotangent(const float* w, int d, int S, int B, const float* g_val, const float* g_diag, int n,
                                float* cmat, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused affine-coupling layer with an MLP coupler (2-D / tabular models, low-dimensional prior flows): the whole coupler
 * network (get_mlp, networks.py:206-224; tanh rule jvp_layers.py:38-53) for the primal and all Jacobian columns, and the
 * coupling update (acl.py:101-146), in one persistent launch -- replaces the per-layer cmf_conv_primal / cmf_conv_tangent
 * launches + cmf_acl_tangent + cmf_acl_primal of one coupling layer.  See csrc/mlp_coupler.hip.
 *   t != NULL  TANGENT mode (decode only): tangents feature-major with 16 column slots of which the first `ncols` <= 15 are
 *              Jacobian columns (the rest zero padding): z and the modified rows of t are updated in place exactly as
 *              cmf_acl_primal(decode) + cmf_acl_tangent would.
 *   t == NULL  PRIMAL mode: z updated in place (decode != 0: x = z e^{-s} - t; else z = (x + t) e^{s});
 *              lj[b] += -/+ sum s when lj != NULL.
 * Limits: hidden widths <= 128, 2 cin <= 256, network outputs <= 64, 2 .. CMF_MLP_MAX_LAYERS linear layers.               */
#define CMF_MLP_MAX_LAYERS 8
typedef struct {
  float* z; long long z_b;            /* primal (B, D): element (b, f) at z + b*z_b + f                                   */
  float* t; long long t_f;            /* tangents (D, B, 16): element (f, b, col) at t + f*t_f + b*16 + col; or NULL      */
  const float* w;                     /* layer images (cmf_pack_mlp_layer), layer l at w + w_off[l]; 16-byte aligned      */
  const int* zi; const int* si; const int* ti; int n_mod;   /* coupling maps as for cmf_acl_primal                          */
  int B, cin, chan_off, chan_step;    /* the network reads z[b][chan_off + f*chan_step], f < cin                          */
  int n_layers;                       /* linear layers: tanh after every one but the last                                */
  int width[CMF_MLP_MAX_LAYERS + 1];  /* width[0] = cin, width[l + 1] = output features of layer l                        */
  long long w_off[CMF_MLP_MAX_LAYERS];
  int decode;
  float* lj;
  int ncols;                          /* TANGENT mode: Jacobian columns in use (1 .. 15)                                  */
} cmf_mlp_coupler_args;
int cmf_mlp_coupler(const cmf_mlp_coupler_args* a, void* stream);
/* Layer image for cmf_mlp_coupler from nn.Linear parameters w [out][in], bias [out] (NULL = zeros): MFMA A fragments in the
 * order the kernel's K-steps consume them (first != 0: the layer that reads the gathered input rows) + bias, zero-padded to
 * out_tiles x 16 outputs and in_groups x 16 inputs.  The kernel expects: hidden layers out_tiles = HT, later layers
 * in_groups = HT with HT = cmf_mlp_hidden_tiles(widest hidden layer); first layer in_groups = ceil(cin / 16); last layer
 * out_tiles = ceil(outputs / 16).  Size query: out == NULL returns the number of floats through *out_floats.            */
int cmf_pack_mlp_layer(const float* w, const float* bias, int out_features, int in_features, int first, int out_tiles,
                       int in_groups, float* out, long long* out_floats, void* stream);
int cmf_mlp_hidden_tiles(int max_hidden_width);

/* ---------------------------------------------------------------------------------------------
 * NSF prior of the low-dimensional flow (config/schemas.py:87-103 -> bijections/nsf.py:86-113,
 * bijections/linear.py:12-34).  The arithmetic is jrmcornish/nsf @ 8e3fe75 (un-vendored: PARITY UNPINNED); implemented
 * from Durkan et al., "Neural Spline Flows" (NeurIPS 2019) -- see csrc/nsf.hip.
 * Elementwise monotone rational-quadratic spline with linear tails outside [-tail_bound, tail_bound]:
 *   params [B][D][3 bins - 1] (bins widths, bins heights -- both divided by sqrt(hidden) as the autoregressive transform
 *   does --, bins - 1 inner knot derivatives); out(b, f) = spline(x(b, f)) (inverse != 0: the inverse map);
 *   lj[b] += sum_f log |d out / d x| (NULL to skip).  out may alias x.                                                */
int cmf_rq_spline(const float* x, long long x_b, const float* params, int D, int bins, int hidden, float tail_bound,
                  int inverse, int B, float* out, long long out_b, float* lj, void* stream);
/* Training: cotangents of x and of the spline parameters from dz (B, D) and dlj (B,) (NULL = 0):
 *   dx(b,f) = dz dz/dx + dlj[b] d log|dz/dx| / dx;   dparams [B][D][3 bins - 1] likewise per parameter.  Forward direction only. */
int cmf_rq_spline_backward(const float* x, long long x_b, const float* params, int D, int bins, int hidden, float tail_bound,
                           int B, const float* dz, long long dz_b, const float* dlj, float* dx, long long dx_b,
                           float* dparams, void* stream);
/* LULinear backward from dW [n][n] = sum_b dy (x) x: gradients of the strict triangles and of the unconstrained diagonal
 * (accumulated), including the log-jac term sum_b dlj[b] * d(sum log diag U) (dlj: (B,) or NULL; scratch1: 1 float).      */
int cmf_lu_backward(const float* dW, const float* lower, const float* upper, const float* unconstrained_diag, int n, float eps,
                    const float* dlj, int B, float* scratch1, float* g_lower, float* g_upper, float* g_udiag, void* stream);
/* du(b, f) = -dlow[b] u(b, f): backward of the standard-normal log-density (gaussian.py:9-22)                           */
int cmf_gaussian_backward(const float* u, const float* dlow, int n, int B, float* du, void* stream);
/* LULinear: W [n][n] = L U (L unit lower from `lower`, U upper from `upper` with diagonal softplus(unconstrained_diag) +
 * eps; entry order of np.tril_indices(n, -1) / np.triu_indices(n, 1)); logdet[0] = sum log diag(U).                   */
int cmf_lu_weights(const float* lower, const float* upper, const float* unconstrained_diag, int n, float eps, float* W,
                   float* logdet, void* stream);
/* out [n_out][n_in] = w * MADE mask (random_mask = False).  kind 0: input -> hidden, 1: hidden -> hidden, 2: hidden ->
 * output with `multiplier` consecutive outputs per feature (strict inequality).  features = autoregressive width D.    */
int cmf_made_mask_weight(const float* w, float* out, int n_out, int n_in, int kind, int features, int multiplier,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Non-convolution pieces of the coupler networks' primal backward.
 * ScaledTanh2dModule (networks.py:96-113), y = sw tanh(u) + sb, g = sw (1 - tanh(u)^2); y, g, dy, dg, du: (B, C, HW):
 *   du = dy g - 2 dg tanh(u) g;   dsw[c] += sum dy tanh(u) + dg (1 - tanh(u)^2);   dsb[c] += sum dy
 * dg (cotangent of g, from cmf_acl_cross_terms) may be NULL; dsw / dsb may be NULL.                                */
int cmf_stanh_backward(const float* dy, const float* dg, const float* y, const float* g, const float* sw,
                       const float* sb, float* du, float* dsw, float* dsb, int B, int C, int HW, void* stream);
/* Second-order term of the tanh MLP couplers' tangent pass (get_linear_jvp + the tanh rule, jvp_layers.py:38-53): layer i+1
 * reads phi_i = 1 - h_i^2 of the primal activation h (B, F).  c holds the UNMASKED cotangent W_{i+1}^T c_{i+1} of the rows
 * (element (b, f, col) at c + b*c_b + f*c_r + col), x the saved raw tangent of the same rows:
 *     c <- phi c  (in place: the cotangent of layer i's raw output);   dh[b][f] += -2 h sum_col c_old x                    */
int cmf_tanh_cross_terms(float* c, long long c_b, long long c_r, const float* x, long long x_b, long long x_r,
                         const float* h, float* dh, int F, int B, int nc, void* stream);
/* MLP coupler primal backward, elementwise stage (networks.py:206-224): out = (dh + extra) (1 - a^2) over n floats, a = the
 * tanh output, extra (or NULL) = the tangent pass's second-order term (cmf_tanh_cross_terms).  out may alias dh.          */
int cmf_tanh_backward(const float* dh, const float* a, const float* extra, long long n, float* out, void* stream);
/* AffineBijection backward (affine.py:24-34): g_ls[f] += sum_b dz x e^{ls} + sum_b dlj[b] (dlj may be NULL), g_sh[f] += sum_b dz,
 * dz <- dz e^{ls} in place; dz, x: (B, n) with row strides dz_b, x_b.                                                     */
int cmf_affine_prior_backward(float* dz, long long dz_b, const float* x, long long x_b, const float* log_scale, int n, int B,
                              const float* dlj, float* g_ls, float* g_sh, void* stream);
/* dst[i] += src[i]: the skip connection of the reverse sweep next to the split-precision kernel (16-byte accesses when n % 4 == 0
 * and both pointers are 16-byte aligned, a scalar sweep otherwise: odd-sized gradient tensors). */
int cmf_accumulate(float* dst, const float* src, long long n, void* stream);
/* relu' bit mask of an activation tensor act (B, C, HW), C % 8 == 0, in the CMF_F_RELU_BITS layout: out[B][HW][C/8] bytes,
 * bit j of byte (b, px, o) = [act(b, 8 o + j, px) > 0].                                                                  */
int cmf_relu_bits(const float* act, void* out, int B, int C, int HW, void* stream);
/* out[c] += sum_{n, px, col} t(n, c, px, col) over a tangent-layout tensor (element at n*t_np + c*t_c + px*t_px +
 * (col/16)*t_sl + col%16, t_sl = 0 meaning 16): the bias gradient of nn.Conv2d / nn.Linear.                        */
int cmf_channel_sum(const float* t, long long t_np, long long t_c, long long t_px, long long t_sl, int np, int C,
                    int npx, int nc, float* out, void* stream);
/* n <= CMF_WGRAD_MAX_BATCH tensors of ONE shape in one launch (t, out: HOST arrays of device pointers): out[k][c] += the sums of t[k]. */
int cmf_channel_sum_batched(const float* const* t, float* const* out, int n, long long t_np, long long t_c, long long t_px,
                            long long t_sl, int np, int C, int npx, int nc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused optimiser step over one flat fp32 buffer: replaces the per-parameter loop of
 * torch.optim.{SGD, Adam, Adamax}(params, lr, weight_decay).step() (experiment.py:515-534) and
 * torch.nn.utils.clip_grad_norm_ (trainer.py:218-219).  `step` counts from 1.                      */
#define CMF_OPT_SGD 0
#define CMF_OPT_ADAM 1
#define CMF_OPT_ADAMAX 2
/* out[0] = sum g^2 (deterministic two-level reduction); ws: 1024 floats of caller-owned workspace   */
int cmf_grad_sqnorm(const float* g, long long n, float* ws, float* out, void* stream);
/* p, g, m, v: n floats each, 16-byte aligned (m, v unused for SGD).  sqnorm != NULL: the gradient is first
 * scaled IN PLACE by min(1, max_norm / (sqrt(sqnorm[0]) + 1e-6)) -- read on the device, no host sync.
 * Hyper-parameters are doubles: torch rounds 1 - beta and lr / (1 - beta^t) to fp32 only after forming them.  */
int cmf_optimizer_step(int kind, float* p, float* g, float* m, float* v, long long n, double lr, double beta1,
                       double beta2, double eps, double weight_decay, int step, const float* sqnorm, float max_norm,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CMF_AMD_H */
