/*
 * aivc_hip.h -- C ABI of libaivc_hip.so: the MI355X (gfx950) native hot path of the
 * AIVC learned video codec (frame transforms, motion compensation, entropy coding).
 *
 * Conventions:
 *   - extern "C", plain pointers and sizes, no framework types.
 *   - every entry point returns AIVC_OK (0) or a negative error code and never throws.
 *   - all pointers are DEVICE pointers owned by the caller unless the name says "host".
 *   - work is enqueued asynchronously on the caller's hipStream_t (passed as void*).
 *   - the library keeps no global state and allocates nothing: scratch comes from the caller.
 *   - results are deterministic: no atomics, fixed accumulation order (see "arithmetic
 *     contract" below), so encoder and decoder agree bit for bit on every GPU.
 *
 * Feature maps are NHWC fp32 ("channels innermost").  The stored channel count of a conv
 * input must be a multiple of 4 (callers zero-pad 3/6/9-channel images to 4/8/12).
 *
 * Each function names the reference code it replaces (paths relative to the upstream
 * repository root, see SURVEY.md for the line-by-line mapping).  The CPU oracle
 * (oracle/aivc_oracle.c) exports the same functions with a `_ref` suffix on HOST pointers;
 * it is test infrastructure only.
 *
 * Arithmetic contract (what "bit exact" means for the fp32 ops):
 *   conv-like ops accumulate   acc = fmaf(a, w, acc)   starting from +0.0f over the reduction
 *   index kk = t * c_in + ci, where t counts the kernel taps in (ky, kx) ascending order (for a
 *   transposed conv: the taps of the output pixel's parity class, out-of-image taps keeping their
 *   place with a zero input) and ci the stored input channels.  kk is walked in groups of 8
 *   (the last group zero padded), inside a group in the order 0, 4, 1, 5, 2, 6, 3, 7
 *   (AIVC_K_ORDER): the order in which v_mfma_f32_32x32x2_f32 consumes an operand row held in
 *   natural K order, so the MFMA kernels need no operand shuffling; the scalar HIP kernels and
 *   the CPU oracle walk the same order and all agree bitwise.  A zero term is an exact no-op.
 *   Then v = acc + bias, then the epilogue in the order documented at aivc_conv2d.
 *   Transcendentals (exp for sigma / sigmoid, expm1 for the Laplace CDF, the factorised
 *   prior's softplus/tanh/sigmoid) are evaluated by a fixed fp64 polynomial scheme
 *   ("det_exp" family) and rounded once to fp32, identically on host and device.
 */
#ifndef AIVC_HIP_H
#define AIVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIVC_CONV_SPARSE4 1

/* position inside a group of 8 reduction indices visited at step i (see the arithmetic contract above) */
#define AIVC_K_ORDER(i) ((((i) & 1) << 2) | ((i) >> 1))

typedef void *aivc_stream_t; /* hipStream_t */

enum {
  AIVC_OK = 0,
  AIVC_ERR_ARG = -1,         /* null pointer / inconsistent sizes */
  AIVC_ERR_UNSUPPORTED = -2, /* shape or option outside what the kernels implement */
  AIVC_ERR_LAUNCH = -3,      /* hipLaunch / runtime error */
  AIVC_ERR_WORKSPACE = -4    /* caller-provided buffer too small */
};

/* ------------------------------------------------------------------------------------------
 * Library identity
 * ---------------------------------------------------------------------------------------- */
/* ABI version, bumped whenever a struct or signature changes (the library and the CPU oracle both return it). */
#define AIVC_ABI_VERSION 17
int aivc_abi_version(void);
/* Last HIP runtime error string seen by this thread's most recent failing call (host). */
const char *aivc_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Convolution family: CustomConvLayer / UpscalingLayer / GDN and the residual compositions
 * built from them (ChengResBlock, ResBlock, AttentionResBlock, SimplifiedAttention).
 * Replaces: src/layers/misc/custom_conv_layers.py:21-253, src/layers/misc/misc_layers.py:113-154,
 *           src/layers/misc/attention.py:22-97 (ATen conv2d / conv_transpose2d / pad kernels).
 * ---------------------------------------------------------------------------------------- */
enum {
  AIVC_MODE_CONV = 0,  /* replicate-pad(pad) + conv k x k stride s             (a1, K1, K2) */
  AIVC_MODE_TCONV = 1, /* ConvTranspose2d k, stride 2, padding=(k+1)/2-1, output_padding 1 (a2, K3) */
  AIVC_MODE_GDN = 2,   /* y = x / sqrt(beta + gamma . x^2)   (w = gamma_eff [C][C], bias = beta_eff) */
  AIVC_MODE_IGDN = 3   /* y = x * sqrt(beta + gamma . x^2) */
};
enum {
  AIVC_ACT_NONE = 0,
  AIVC_ACT_LEAKY = 1,  /* v > 0 ? v : v * 0.01f */
  AIVC_ACT_RELU = 2,   /* v > 0 ? v : 0 */
  AIVC_ACT_SIGMOID = 3 /* 1 / (1 + fp32(det_exp(-v))) */
};
enum { AIVC_ALGO_AUTO = 0, AIVC_ALGO_DIRECT = 1, AIVC_ALGO_MFMA = 2 };

typedef struct aivc_conv_params {
  int32_t mode;   /* AIVC_MODE_* */
  int32_t ksize;  /* 1, 3 or 5 (GDN modes: 1) */
  int32_t stride; /* 1 or 2 (TCONV: must be 2 = upsampling factor) */
  int32_t pad;    /* CONV: replicate padding on each side (0 or ksize/2); TCONV: ignored */
  int32_t n, h_in, w_in, c_in; /* input NHWC; c_in = stored channels, multiple of 4 */
  int32_t h_out, w_out, c_out; /* output NHWC; must match the mode's size formula */
  int32_t act1;                /* applied to acc + bias                */
  int32_t act2;                /* applied after the residual addition   */
  int32_t algo;                /* AIVC_ALGO_* (AUTO picks MFMA when the shape allows) */
  int32_t gdn;                 /* CONV/TCONV only: 0 none, 1 GDN, 2 inverse GDN fused after the bias */
  int32_t flags;               /* AIVC_CONV_SPARSE4: every 4th stored input channel (ci % 4 == 3) is zero in x --
                                * the layout of 3-channel images padded to 4; those terms are exact no-ops of the
                                * fmaf chain and the MFMA kernels skip them (a hint: results never depend on it) */
  const float *x;    /* [n][h_in][w_in][c_in] */
  const float *w;    /* [c_out][ksize][ksize][c_in]  (OHWI; TCONV: w[co][ky][kx][ci] = torch weight[ci][co][ky][kx]) */
  const float *bias; /* [c_out] or NULL */
  const float *mul;  /* [n][h_out][w_out][c_out] or NULL: v = mul * v (after act1) */
  const float *res;  /* [n][h_out][w_out][c_out] or NULL: v = v + res (after mul)  */
  float *y;          /* [n][h_out][w_out][c_out] */
  const float *gdn_beta;  /* [c_out]         effective (re-parameterised) beta,  when gdn != 0 */
  const float *gdn_gamma; /* [c_out][c_out]  effective gamma [i][j],             when gdn != 0 */
  /* Optional fused tail (CONV mode, no gdn): a 1x1 convolution applied to t = act1(acc + bias) in the same launch,
   *   y[.., i] = act2( sum_j (AIVC_K_ORDER) t_j * tail_w[i][j] + tail_bias[i]  (+ res) ),   y is [n][h_out][w_out][tail_c_out]
   * -- bit identical to this conv followed by a 1x1 CONV launch with the same epilogue (the bottleneck blocks of the
   * attention module, src/layers/misc/attention.py:22-42: the c_out-channel intermediate never leaves the chip).
   * mul must be NULL; res / act2 / y then refer to the tail's output.  tail_c_out == 0: no tail. */
  const float *tail_w;    /* [tail_c_out][c_out] */
  const float *tail_bias; /* [tail_c_out] or NULL */
  int32_t tail_c_out;
  int32_t precision; /* AIVC_PREC_FP32 (0): the arithmetic contract above -- bit identical on every kernel and on the CPU
                      * oracle.  AIVC_PREC_BF16X3 (1, ABI 12): a precision MODE for CONV / TCONV with c_in % 32 == 0 and c_out
                      * of 64 or a multiple of 128 (other shapes run the fp32 contract; a fused gdn / 1x1 tail keeps its fp32 GEMM): every fp32 operand is split exactly
                      * into three bf16 terms and a product is six bf16 MFMA products with fp32 accumulation -- within fp32
                      * summation-order noise of the contract's result but NOT its bits (tests/test_gpu_precision.py reports
                      * the error per layer class).  Never the default; bitstreams of the two modes do not interoperate. */
  const void *w_bf16x3; /* ABI 13, optional, read under AIVC_PREC_BF16X3 only: the image of `w` that aivc_split_weights_bf16x3
                         * wrote (c_out * ksize * ksize * c_in * 6 bytes).  Same results bit for bit as with NULL (the
                         * kernels then split the weight fragments in their K loop, ~25 % slower): the terms are the same,
                         * they are only computed once per layer instead of once per tile. */
  const float *w_wino; /* ABI 16, read under AIVC_PREC_FP32_WINO only (required there for the layers aivc_winograd_covers()
                        * names): the image of `w` that aivc_winograd_weights wrote, c_out * 16 * c_in floats. */
} aivc_conv_params;
#define AIVC_PREC_FP32 0
#define AIVC_PREC_BF16X3 1
/* AIVC_PREC_FP32_WINO (ABI 16): the fp32 arithmetic contract, version 2.  Identical to AIVC_PREC_FP32 everywhere except
 * for the stride-1 3x3 convolutions with replicate padding 1, c_in % 32 == 0, c_out % 128 == 0 (not the 64-channel 3x3 of the
 * bottleneck blocks, which the kernels fuse with its 1x1 tail: fused or in two launches, it stays version 1), no fused 1x1 tail and at
 * least AIVC_WINO_MIN_PIXELS input pixels per image (aivc_winograd_covers: a function of the layer and its input size only;
 * src/layers/misc/custom_conv_layers.py:21-180, src/layers/misc/attention.py:22-97 build their residual blocks from
 * them), whose accumulator is the Winograd F(2x2, 3x3) chain below instead of the 9-tap chain: 16 multiplications per
 * 2x2 output pixels, input channel and output channel instead of 36 (the fp32 matrix pipe is the scarce unit of the
 * part).  STILL a fixed-order fp32 chain, bit identical on every kernel and on the CPU oracle; within summation-order
 * noise of version 1 (the per-position chains are 9x shorter), NOT its bits: an encoder and a decoder must run the same
 * version.  For the output tile (ty, tx) = pixels (2 ty + a, 2 tx + b), a, b in {0, 1}:
 *   d[r][c]      = x[clamp(2 ty - 1 + r)][clamp(2 tx - 1 + c)],  r, c = 0..3                     (replicate padding)
 *   position p = 4 i + j, i, j = 0..3, with (A, B, S)[0..3] = (0, 2, -1), (1, 2, +1), (2, 1, -1), (1, 3, -1):
 *   R_i[c][ci]   = fmaf(S[i], d[B[i]][c], d[A[i]][c])                                             (rows first)
 *   V_p[ci]      = fmaf(S[j], R_i[B[j]], R_i[A[j]])
 *   M_p[co]      = fmaf chain from +0 over ci (groups of 8 in AIVC_K_ORDER) of V_p[ci] * U_p[co][ci]
 *   acc[a][b]    = S0 + S1,  S0 = sum over p = 0..7, S1 = sum over p = 8..15, each in ascending order from +0, of
 *                  T[a][i] * T[b][j] * M_p  (terms with a zero coefficient are skipped; the others are one fp32 addition
 *                  or subtraction each),  T[0] = (1, 1, 1, 0), T[1] = (0, 1, -1, -1)
 * then the epilogue of the contract unchanged (bias, fused gdn, act1, mul, res, act2).  U = aivc_winograd_weights(w). */
#define AIVC_PREC_FP32_WINO 2
#define AIVC_WINO_MIN_PIXELS 8000 /* h_in * w_in from which the version applies: below it the 16 x 16-pixel blocks of the kernel
                                   * quantise the image badly and a launch is a handful of blocks per CU (34 x 60: no gain; the 68 x 120
                                   * layers of a 1080p frame are covered: x1.26 ... 1.42 on batches of 16 ... 64 frames) */
#define AIVC_WINO_MIN_PIXELS_TCONV 32768 /* ... of the transposed form (INPUT pixels): a class pass is 16 chunks of 6 positions per wave on
                                         * average, so a block's fixed costs weigh more (68 x 120: x0.93, 272 x 480: x1.2) */
#define AIVC_CONV_WINO_ANY_SIZE 2 /* aivc_conv_params.flags: version 2 whatever the image size (the tests drive the kernel on shapes the oracle checks in seconds) */
/* ... and (ABI 17) for the 5x5 STRIDE-2 convolutions with replicate padding 2, c_in = 32 * 2^k, c_out % 128 == 0, no fused 1x1 tail and
 * at least AIVC_WINO_MIN_PIXELS OUTPUT pixels (the second analysis layer of both networks, src/layers/misc/custom_conv_layers.py:129-180:
 * a quarter of the codec's multiplications), in POLYPHASE form: output pixel (oy, ox) sums, over the four phases (py, px) of the input,
 * a stride-1 3x3 convolution of the phase image  X_ph[y][x] = x[clamp(2 y + py)][clamp(2 x + px)]  (the replicate padding of the ORIGINAL
 * image) with the phase kernel  g_ph[r][l] = w[2 r + py][2 l + px]  (zero where that index would be 5).  Exactly the chain above on a
 * layer of 4 c_in "virtual" input channels  cv = (2 py + px) * c_in + ci  (phases ascending, channels of a phase in groups of 8 in
 * AIVC_K_ORDER), with  d[r][c] = X_ph[2 ty - 1 + r][2 tx - 1 + c]  and  U_p[co][cv] = (G g_ph G^T)[i][j];  the positions whose U is zero
 * by construction -- i == 3 for py = 1, j == 3 for px = 1 -- take no part in M_p: 16 + 12 + 12 + 9 = 49 multiplications per 2x2 outputs,
 * input and output channel instead of 100.  U = aivc_winograd_weights_poly5(w). */
/* ... and for the 5x5 stride-2 TRANSPOSED convolutions with c_in % 32 == 0, c_out % 64 == 0 and at least AIVC_WINO_MIN_PIXELS_TCONV INPUT
 * pixels (the synthesis layers, src/layers/misc/custom_conv_layers.py:183-253), class by class: the outputs of parity class (pyc, pxc),
 * y[2 v + pyc][2 u + pxc], are a stride-1 3x3 correlation of the ZERO-extended input with the class kernel
 * g_c[r][l] = w[pyc + 4 - 2 r][pxc + 4 - 2 l]  (zero where that index would be 5, i.e. r = 0 for pyc = 1, l = 0 for pxc = 1): the chain
 * above with  d[r][c] = x[2 tv - 1 + r][2 tu - 1 + c]  (0 outside the image)  for the tile of grid pixels (2 tv + a, 2 tu + b),
 * U_p[co][ci] = (G g_c G^T)[i][j],  and the positions whose U is zero by construction -- i == 0 for pyc = 1, j == 0 for pxc = 1 -- taking
 * no part in M_p (49 instead of 100 multiplications per 4x4 outputs ...).  U = aivc_winograd_weights_tconv5(w). */
static inline int aivc_winograd_covers(const aivc_conv_params *p) {
  if (p->tail_c_out != 0 || p->act1 == AIVC_ACT_SIGMOID || p->act2 == AIVC_ACT_SIGMOID) return 0;
  if (p->mode == AIVC_MODE_TCONV)
    return p->ksize == 5 && p->stride == 2 && p->c_in % 32 == 0 && p->c_out % 64 == 0 &&
           ((int64_t)p->h_in * p->w_in >= AIVC_WINO_MIN_PIXELS_TCONV || (p->flags & AIVC_CONV_WINO_ANY_SIZE));
  if (p->mode != AIVC_MODE_CONV || p->c_out % 128 != 0) return 0;
  if (p->ksize == 3 && p->stride == 1 && p->pad == 1 && p->c_in % 32 == 0)
    return (int64_t)p->h_in * p->w_in >= AIVC_WINO_MIN_PIXELS || (p->flags & AIVC_CONV_WINO_ANY_SIZE);
  if (p->ksize == 5 && p->stride == 2 && p->pad == 2 && p->c_in >= 32 && (p->c_in & (p->c_in - 1)) == 0)
    return (int64_t)p->h_out * p->w_out >= AIVC_WINO_MIN_PIXELS || (p->flags & AIVC_CONV_WINO_ANY_SIZE);
  return 0;
}
/* Epilogue order:  v = acc + bias;  [mode GDN: v = x / sqrtf(v) | mode IGDN: v = x * sqrtf(v)];
 *                  [fused gdn: with t_j = v_j * v_j over the pixel's channels,
 *                     s_i = fmaf chain over j (in AIVC_K_ORDER) of (t_j, gamma[i][j]) from +0, then + beta[i];
 *                     v_i = gdn == 1 ? v_i / sqrtf(s_i) : v_i * sqrtf(s_i)
 *                   -- bit identical to a CONV launch followed by a GDN/IGDN-mode launch];
 *                  v = act1(v);  if (mul) v = mul * v;  if (res) v = v + res;  v = act2(v).
 * A fused-gdn or fused-tail request the kernels cannot honour (all output channels of a pixel must sit in one
 * workgroup tile: c_out of 64 or 128 on the MFMA path; tail: c_out 64 -> tail_c_out 128, c_in % 32 == 0) returns
 * AIVC_ERR_UNSUPPORTED; callers then issue the two launches (aivc_conv2d_variant tells in advance). */
int aivc_conv2d(const aivc_conv_params *p, aivc_stream_t stream);

/* AIVC_PREC_FP32_WINO: U_p[co][ci], p = 4 i + j, = (G g G^T)[i][j] of the 3x3 kernel g[ky][kx] = w[co][ky][kx][ci], G = (1, 0, 0),
 * (.5, .5, .5), (.5, -.5, .5), (0, 0, 1), evaluated in fp64 in the order  t[i][l] = g[0][l] | .5 * ((g[0][l] + g[1][l]) + g[2][l]) |
 * .5 * ((g[0][l] - g[1][l]) + g[2][l]) | g[2][l],  then the same along l, rounded once to fp32.  w is OHWI [c_out][3][3][c_in]
 * (c_in % 8 == 0, c_out % 64 == 0); u is laid out as the kernels stage it, one contiguous 32 KB image per (block of 64 output
 * channels, chunk of 8 input channels):  u[AIVC_WINO_U_INDEX(co, p, ci, c_in)]. */
#define AIVC_WINO_U_INDEX(co, p, ci, c_in) \
  ((((((size_t)((co) / 64) * (size_t)((c_in) / 8) + (size_t)((ci) / 8)) * 16 + (size_t)(p)) * 2 + (size_t)(((ci) % 8) / 4)) * 64 + \
    (size_t)((co) % 64)) * 4 + (size_t)((ci) % 4))
int aivc_winograd_weights(const float *w, int32_t c_out, int32_t c_in, float *u, aivc_stream_t stream);
/* ABI 17: the polyphase form of a 5x5 stride-2 kernel (see aivc_winograd_covers): w is OHWI [c_out][5][5][c_in], u takes
 * c_out * 16 * 4 * c_in floats -- the chunk images of a layer of 4 c_in virtual input channels, phase-major. */
int aivc_winograd_weights_poly5(const float *w, int32_t c_out, int32_t c_in, float *u, aivc_stream_t stream);
/* ABI 17: the class kernels of a transposed 5x5 stride-2 layer (see aivc_winograd_covers): w is OHWI [c_out][5][5][c_in]
 * (w[co][ky][kx][ci] = torch weight[ci][co][ky][kx]), u takes 4 * c_out * 16 * c_in floats -- class-major blocks of 64 output channels. */
int aivc_winograd_weights_tconv5(const float *w, int32_t c_out, int32_t c_in, float *u, aivc_stream_t stream);

/* AIVC_PREC_BF16X3, weights split ahead of the launches (aivc_conv_params.w_bf16x3): every weight of w [c_out][k_total]
 * (k_total = ksize * ksize * c_in, a multiple of 32: the OHWI rows of aivc_conv2d) as its three bf16 terms
 * h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even; x = h + m + l exactly), laid out for the
 * kernels' loader:  out[((co * k_total / 32 + t) * 3 + term) * 32 + k % 32]  as uint16, t = k / 32.
 * out: c_out * k_total * 6 bytes, 16-byte aligned. */
int aivc_split_weights_bf16x3(const float *w, int32_t c_out, int32_t k_total, void *out, aivc_stream_t stream);

/* Which kernel aivc_conv2d would launch for these parameters (no launch): 0 = scalar kernel,
 * 1 = thin-output VALU kernel (transposed conv to 3 / 6 channels), otherwise 100 + 10 * template-mode (0 conv, 1 tconv, 2 gdn) + tile id (0: 128x128, 1: 64x64,
 * 2: 256x64, 3: 128x32, 4: 256x128, 5: 64x128, 6: 128x64) + 50 with a fused gdn; 190 = conv with a fused 1x1 tail (191: aivc_conv_images).
 * 1000 + that code: the launch the precision mode takes (AIVC_PREC_BF16X3; the tile depends on whether w_bf16x3 is given).
 * Negative = error code.  Used by bench.py to attribute launch times and by callers to ask whether a fusion is available. */
int aivc_conv2d_variant(const aivc_conv_params *p);

/* GDN re-parameterisation, done once per layer instead of once per call:
 *   beta_eff[i]    = max(beta[i],  beta_bound)^2  - pedestal
 *   gamma_eff[i,j] = max(gamma[i,j], gamma_bound)^2 - pedestal       (all fp32)
 * Replaces src/layers/misc/misc_layers.py:131-139. */
int aivc_gdn_reparam(const float *beta, const float *gamma, int32_t c, float beta_bound,
                     float gamma_bound, float pedestal, float *beta_eff, float *gamma_eff,
                     aivc_stream_t stream);

/* Diagnostic (no reference counterpart; nothing on the coded path calls it): the fused (inverse) GDN epilogues compute
 * sqrt and division by the refinement steps of the IEEE sequences alone when a wavefront's operands all lie in
 * [2^-60, 2^60] (aivc_amd/csrc/common.h).  This counts, on the device, the inputs for which those lean sequences differ
 * from the compiler's full ones: mismatch[0] over EVERY float s in the range for the square root, mismatch[1] over
 * n_div_pairs pseudo-random (numerator, denominator) pairs for the division.  Both must come back 0. */
int aivc_selfcheck_gdn_math(uint64_t n_div_pairs, uint32_t seed, uint64_t *mismatch, aivc_stream_t stream);

/* Zero-pad channels: in [npix][c_in] -> out [npix][c_out], c_out >= c_in, extra channels = 0. */
int aivc_pad_channels(const float *in, size_t npix, int32_t c_in, float *out, int32_t c_out,
                      aivc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Frame layout ops
 * ---------------------------------------------------------------------------------------- */
/* InputLayer: planar YUV 4:2:0 (fp32, values k/255) -> 3 channels of an NHWC tensor
 * (nearest x2 on U,V, crop to the Y size).  Writes channels c_off..c_off+2 of
 * out[n][h][w][c_store]; if zero_pad != 0 also writes 0 into channel c_off+3.
 * u,v are [n][ceil(h/2)][ceil(w/2)].  Replaces src/layers/ae/ae_layers.py:17-35. */
int aivc_yuv420_to_444(const float *y, const float *u, const float *v, int32_t n, int32_t h,
                       int32_t w, float *out, int32_t c_store, int32_t c_off, int32_t zero_pad,
                       aivc_stream_t stream);
/* Same from 8-bit planes: value = (float)byte / 255.0f (what to_tensor() of an 8-bit PNG gives,
 * src/func_util/img_processing.py:213-231). */
int aivc_yuv420u8_to_444(const uint8_t *y, const uint8_t *u, const uint8_t *v, int32_t n,
                         int32_t h, int32_t w, float *out, int32_t c_store, int32_t c_off,
                         int32_t zero_pad, aivc_stream_t stream);

/* Input of the first conv of a transform: up to AIVC_MAX_IMAGES 3-channel images side by side, each stored as 4
 * channels (c0, c1, c2, 0): out[n][h][w][4 * n_img].  An image comes from 8-bit 4:2:0 planes (InputLayer
 * semantics as aivc_yuv420u8_to_444), from the first 3 channels of an NHWC float tensor f[n][h][w][f_channels]
 * (the prediction alpha * x_warp), or is all zero (every pointer NULL: the missing reference of a P frame).
 * One launch, every 16-byte store of a wave contiguous; replaces the cat / pad of the inputs of g_a and g_a_ref
 * (src/real_life/decode.py:709-714, 631-636 and the encoder's mirror image). */
#define AIVC_MAX_IMAGES 3
typedef struct {
  const uint8_t *y, *u, *v; /* planes [n][h][w], [n][ceil(h/2)][ceil(w/2)] x2, or NULL */
  const float *f;           /* or an NHWC tensor [n][h][w][f_channels], or NULL */
  int32_t f_channels;
  int32_t reserved;
} aivc_image_src;
int aivc_pack_images(const aivc_image_src *src, int32_t n_img, int32_t n, int32_t h, int32_t w, float *out,
                     aivc_stream_t stream);

/* The first layer of the analysis transforms in one launch: aivc_conv2d(p) over the tensor aivc_pack_images(src, n_img)
 * would produce -- without producing it.  p->x is ignored (may be NULL), p->c_in must be 4 * n_img, p->n / h_in / w_in
 * are the image batch and size, p->w is packed for the 4-channels-per-image layout ([c_out][5][5][4 * n_img], zero
 * columns for the fourth channels).  Bit identical to the two calls.  Implemented for what the codec uses
 * (CustomConvLayer(5, stride 2) to 64 channels with GDN or a cheap activation: ksize 5, stride 2, pad 2, c_out 64, bias /
 * fused gdn / act1 in {none, leaky, relu}, no mul / res / act2 / tail); anything else returns AIVC_ERR_UNSUPPORTED and the
 * caller issues the two calls.  Replaces InputLayer + torch.cat + CustomConvLayer at the head of g_a / g_a_ref
 * (src/layers/ae/ae_layers.py:27-35, src/layers/misc/custom_conv_layers.py:129-180). */
int aivc_conv_images(const aivc_image_src *src, int32_t n_img, const aivc_conv_params *p, aivc_stream_t stream);

/* Reconstruction tail of Decoder.decode: x_hat = x[:, :h, :w, :3] (+ skip), OutputLayer
 * (U,V = bilinear x0.5, align_corners=False = 2x2 mean), replicate-pad U,V to ceil(h/2) x
 * ceil(w/2), then the 8-bit cast round(255*clamp(v,0,1))/255 (half-to-even).
 * x is [n][hx][wx][cx] with hx>=h, wx>=w, cx>=3; skip is [n][h][w][cs] (cs>=3) or NULL.
 * Outputs (each may be NULL): fp32 planes holding 8-bit levels and/or the bytes themselves.
 * Replaces src/real_life/decode.py:549-578, src/layers/ae/ae_layers.py:38-56,
 *          src/func_util/img_processing.py:68-73. */
int aivc_frame_to_yuv420(const float *x, int32_t n, int32_t hx, int32_t wx, int32_t cx,
                         const float *skip, int32_t cs, int32_t h, int32_t w, float *y, float *u,
                         float *v, uint8_t *y8, uint8_t *u8, uint8_t *v8, aivc_stream_t stream);

/* OutputLayer alone (no cast): out[n][j][h/2][w/2] = 2x2 mean of channel ch0+j of x [n][h][w][c]
 * with the bilinear x0.5 operation order  0.5*(0.5*a + 0.5*b) + 0.5*(0.5*c + 0.5*d).
 * Replaces F.interpolate(scale_factor=0.5, mode='bilinear') at src/layers/ae/ae_layers.py:46-52. */
int aivc_downsample2x(const float *x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ch0,
                      int32_t nch, float *out, aivc_stream_t stream);

/* MOFNet output unpack + bi-directional motion compensation + conditional-coding split:
 *   alpha = clamp(m[0]+.5,0,1)  beta = clamp(m[1]+.5,0,1)  v_prev = m[2:4]  v_next = m[4:6]
 *   frame_type P (1): beta = 1, v_next = 0
 *   warp(x, v): bilinear sample of x at (col + v.x, row + v.y), border clamp, align_corners
 *   x_warp = beta * warp(prev, v_prev) + (1 - beta) * warp(next, v_next)
 *   pred   = alpha * x_warp          skip = (1 - alpha) * x_warp
 * mof is the MOFNet synthesis output [n][hm][wm][cm] (hm>=h, wm>=w, cm>=6); prev/next are
 * [n][h][w][cr] (cr>=3).  pred/skip/x_warp are [n][h][w][co] (co>=3, channels >=3 zeroed);
 * alpha_out/beta_out are [n][h][w] (optional, for logging).
 * Replaces src/real_life/decode.py:524-542,729-739 and src/func_util/optical_flow.py:14-55. */
int aivc_warp_blend(const float *mof, int32_t hm, int32_t wm, int32_t cm, const float *prev,
                    const float *next, int32_t cr, int32_t n, int32_t h, int32_t w,
                    int32_t frame_type, float *pred, float *skip, float *x_warp, int32_t co,
                    float *alpha_out, float *beta_out, aivc_stream_t stream);

/* The same for a BAND of output rows [row0, row0 + rows) of the frame (one frame's row bands spread over several
 * GPUs): mof is the band of the MOFNet output, [n][hm][wm][cm] with its local row 0 = frame row row0
 * (hm >= rows); prev / next are the WHOLE references [n][h][w][cr] (motion vectors reach anywhere);
 * pred / skip / x_warp are [n][rows][w][co], alpha_out / beta_out [n][rows][w].  Row r of the outputs is bit-identical
 * to row row0 + r of aivc_warp_blend on the whole frame. */
int aivc_warp_blend_rows(const float *mof, int32_t hm, int32_t wm, int32_t cm, const float *prev,
                         const float *next, int32_t cr, int32_t n, int32_t h, int32_t w, int32_t row0, int32_t rows,
                         int32_t frame_type, float *pred, float *skip, float *x_warp, int32_t co,
                         float *alpha_out, float *beta_out, aivc_stream_t stream);

/* Stand-alone warp (src/func_util/optical_flow.py:14-55): x [n][h][w][c], flow [n][h][w][2]
 * (channel 0 = horizontal, 1 = vertical, pixel units) -> out [n][h][w][c]. */
int aivc_warp(const float *x, const float *flow, int32_t n, int32_t h, int32_t w, int32_t c,
              float *out, aivc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Latent ops
 * ---------------------------------------------------------------------------------------- */
/* PdfParamParameterizer (K = 1): hs is h_s(z_hat) [n][hh][wh][2c]; crop to [h][w];
 * mu = channels [0,c), sigma = fp32(det_exp(0.5 * clamp(channels [c,2c), -18.4207, 10))).
 * Replaces src/layers/misc/misc_layers.py:180-219 + the crop at src/real_life/decode.py:853. */
int aivc_hyper_params(const float *hs, int32_t n, int32_t hh, int32_t wh, int32_t c, int32_t h,
                      int32_t w, float *mu, float *sigma, aivc_stream_t stream);

/* out[p][ch] = in[p][ch] * fabsf(gain[ch])   (GainMatrix.forward, integer idx_rate;
 * src/layers/multi_rate/gain_matrix.py:92-157).  gain may be NULL (copy). */
int aivc_channel_gain(const float *in, const float *gain, size_t npix, int32_t c, float *out,
                      aivc_stream_t stream);

/* Fractional rate index (GainMatrix.interpolate_gain_vector, src/layers/multi_rate/gain_matrix.py:159-194):
 * out[ch] = fp32(|g_r[ch]| ^ l) * fp32(|g_t[ch]| ^ (1 - l)), powers evaluated as det_exp(l * det_log(.)) in
 * fp64 so that encoder and decoder derive the same gains on any machine. */
int aivc_gain_interp(const float *g_r, const float *g_t, int32_t c, float l, float *out,
                     aivc_stream_t stream);

/* Encoder side: q = clamp(rint(y - mu), -256, 255) (half-to-even); y_hat = (q + mu) * |gain_dec|.
 * mu == NULL means mu = 0 (the z latent); gain_dec == NULL means gain 1.  q (int16) and y_hat
 * may each be NULL.  Replaces Quantizer (src/layers/misc/misc_layers.py:162-169) and the
 * centring/gain of src/real_life/decode.py:867-885. */
int aivc_quantize_center(const float *y, const float *mu, const float *gain_dec, size_t npix,
                         int32_t c, int16_t *q, float *y_hat, aivc_stream_t stream);
/* Decoder side: y_hat = ((float)q + mu) * |gain_dec|. */
int aivc_dequantize(const int16_t *q, const float *mu, const float *gain_dec, size_t npix,
                    int32_t c, float *y_hat, aivc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Entropy coding (torchac-compatible 32-bit range coder, 16-bit CDF precision).
 * Replaces src/real_life/bitstream.py:82-184 (CDF build), :241-287 / :426-485 (coding) and the
 * third-party torchac.encode_float_cdf / decode_float_cdf (fab-jul/torchac, unpinned).
 * ---------------------------------------------------------------------------------------- */
#define AIVC_AC_MAX_VAL 256
#define AIVC_LP 514      /* CDF points per symbol: 2*AC_MAX_VAL + 2             */
#define AIVC_MAX_SYMBOL 512 /* torchac's max_symbol = Lp - 2 (value +256): legal, upper bound 2^16 */
#define AIVC_CDF_ROW 520 /* uint16 per stored CDF row (514 used, 1040 B = 65 x 16 B) */
#define AIVC_BALLE_PARAMS 43 /* floats per channel, see aivc_balle_cdf_table */
#define AIVC_MAX_MAPS 256
#define AIVC_RC_MAX_STREAMS 64

/* Factorised-prior CDF table (BallePdfEstim.cdf at k - 256.5, k = 0..513), quantised like
 * torchac:  u16 = (uint16)(rint(cdf * 65023) + k).
 * params[c][43] = matrix_h0[1x3] h1[3x3] h2[3x3] h3[3x1] | bias_b0[3] b1[3] b2[3] b3[1] |
 *                 bias_a0[3] a1[3] a2[3]   (row-major [d][r] for matrix_h).
 * table is [c][AIVC_CDF_ROW]; cdf_f32 (optional) is [c][AIVC_LP].
 * Replaces src/layers/entropy_coding/pdf_estimator.py:204-245 + src/real_life/bitstream.py:82-125. */
int aivc_balle_cdf_table(const float *params, int32_t c, uint16_t *table, float *cdf_f32,
                         aivc_stream_t stream);

/* List of y feature maps that are not identically zero (ascending), src/real_life/bitstream.py:241-255.
 * q is [h*w][c] int16; flags[c] gets 0/1 (device). */
int aivc_nonzero_maps(const int16_t *q, size_t npix, int32_t c, uint8_t *flags,
                      aivc_stream_t stream);
/* The same for the n images of a batch in one launch (ABI 9): q [n][h*w][c], flags [n][c]; n <= 65535. */
int aivc_nonzero_maps_batch(const int16_t *q, int32_t n, size_t npix, int32_t c, uint8_t *flags,
                            aivc_stream_t stream);

typedef struct aivc_map_list {
  int32_t n_maps;
  uint8_t idx[AIVC_MAX_MAPS];
} aivc_map_list;

/* Laplace(0, sigma/sqrt(2)) CDF rows for every coded symbol position, channel-major order
 * (stream position p = (m * npix + pix), channel = maps.idx[m]):
 *   rows[p][k] = (uint16)(rint(cdf(k - 256.5) * 65023) + k),  k = 0..513.
 * sigma is NHWC [npix][c].  Replaces src/real_life/bitstream.py:127-154 without ever holding
 * the fp32 [C,H,W,514] tensor. */
int aivc_laplace_cdf_rows(const float *sigma, size_t npix, int32_t c, const aivc_map_list *maps,
                          uint16_t *rows, aivc_stream_t stream);

/* The same rows restricted to what the range decoder's fast path reads: the AIVC_CDF_WIN entries from AIVC_CDF_WIN0 on
 * (symbols -32 .. +31) of every coded position, 128 B instead of 1040, and sigma of the position (from which
 * aivc_range_decode_windows rebuilds the rest of a row with the same function when a symbol falls outside):
 *   win[p][j] = rows[p][AIVC_CDF_WIN0 + j],  sigma_pos[p] = sigma[pix][maps.idx[m]]. */
#define AIVC_CDF_WIN0 224
#define AIVC_CDF_WIN 64
int aivc_laplace_cdf_windows(const float *sigma, size_t npix, int32_t c, const aivc_map_list *maps, uint16_t *win,
                             float *sigma_pos, aivc_stream_t stream);
/* Encoder: only the two CDF values a symbol needs.  bounds[p] = c_lo | (c_hi << 16) with
 * c_lo = cdf_u16[sym], c_hi = cdf_u16[sym+1], sym = q + 256 in [0, 512].  For sym = 512 (torchac's max_symbol)
 * the upper bound is 2^16 whatever the row holds (torchac: `sym == max_symbol ? 0x10000 : cdf[sym + 1]`); it is
 * packed as c_hi = 0, which no other symbol can have (c_hi > c_lo >= 0), and aivc_range_encode reads 0 as 2^16
 * (ABI 8). */
int aivc_laplace_bounds(const float *sigma, const int16_t *q, size_t npix, int32_t c,
                        const aivc_map_list *maps, uint32_t *bounds, aivc_stream_t stream);
/* Same from a per-channel table (pmf mode, all c channels, channel-major). */
int aivc_table_bounds(const uint16_t *table, const int16_t *q, size_t npix, int32_t c,
                      uint32_t *bounds, aivc_stream_t stream);

typedef struct aivc_rc_stream {
  uint64_t in_off;   /* encode: first element of bounds[]; decode: byte offset of the payload
                        in bytes[] (multiple of 4, zero padded to a multiple of 4 + 8) */
  uint64_t out_off;  /* encode: byte offset in out[] (multiple of 4); decode: offset in sym[] */
  uint64_t row_off;  /* decode: first row (in rows of AIVC_CDF_ROW uint16) */
  uint32_t n_sym;
  uint32_t in_len;   /* decode: payload length in bytes */
  uint32_t out_cap;  /* encode: capacity in bytes at out_off */
  uint32_t plane;    /* decode: 0 -> row = row_off + i (per-symbol rows)
                                 P -> row = row_off + i / P (pmf table, P = h*w) */
} aivc_rc_stream;
typedef struct aivc_rc_batch {
  int32_t n_streams;
  int32_t reserved;
  aivc_rc_stream s[AIVC_RC_MAX_STREAMS];
} aivc_rc_batch;

/* One independent range-coder stream per entry, run concurrently.
 * encode: out_len[i] = number of bytes produced (or 0xFFFFFFFF on overflow of out_cap). */
int aivc_range_encode(const uint32_t *bounds, const aivc_rc_batch *batch, uint8_t *out,
                      uint32_t *out_len, aivc_stream_t stream);
/* decode: sym[out_off + i] = decoded symbol in [0, 512].
 * consumed_bits (optional, [n_streams]): the number of bits stream i shifted in by renormalisation (E1 + E2 + E3 steps)
 * over ALL its n_sym symbols, the last one's update included.  A payload written by the coder of this format holds
 * exactly those bits plus the two of the final flush (the disambiguating bit and one pending bit; further pending bits
 * are E3 steps the decoder counts too), zero padded to a byte:
 *     in_len == (consumed_bits + 2 + 7) / 8
 * for every stream that was decoded with the CDFs it was written with.  A decoder that left the writer's track (other
 * CDF bounds on some coded symbol: another implementation's sigma, a corrupted payload) keeps producing in-alphabet
 * symbols -- torchac's decoder reports nothing either -- but it ends up with a different count: the host-side check of
 * this identity is the product's desynchronisation detector (aivc_amd/real_life/bitstream.py). */
int aivc_range_decode(const uint8_t *bytes, const uint16_t *rows, const aivc_rc_batch *batch,
                      uint16_t *sym, uint32_t *consumed_bits, aivc_stream_t stream);
/* Laplace-mode decode from windows (aivc_laplace_cdf_windows): stream i starts at position row_off of win / sigma_pos,
 * one position per symbol (plane must be 0).  Same symbols and bit counts as aivc_range_decode on the full rows. */
int aivc_range_decode_windows(const uint8_t *bytes, const uint16_t *win, const float *sigma_pos,
                              const aivc_rc_batch *batch, uint16_t *sym, uint32_t *consumed_bits, aivc_stream_t stream);
/* Scatter decoded symbols back to the latent: q[pix][maps.idx[m]] = sym[m*npix + pix] - 256,
 * all other channels 0.  (src/real_life/bitstream.py:458-466) */
int aivc_scatter_symbols(const uint16_t *sym, size_t npix, int32_t c, const aivc_map_list *maps,
                         int16_t *q, aivc_stream_t stream);

/* ---- one launch per frame BATCH for the CDF-bound / scatter kernels (the frames of a dependency level) ----------
 * Per frame f of the batch: which y maps are coded (as aivc_map_list) and where its coded positions start in the
 * concatenated stream-order arrays.  The table lives in DEVICE memory (272 bytes per frame: beyond kernel-argument
 * space for a 64-frame level); sigma / q are the batch tensors [n][npix][c].  Frame f's results are bit-identical to
 * the single-frame call with its map list, written / read at element offset pos_off. */
typedef struct aivc_frame_maps {
  uint64_t pos_off; /* first coded position (symbol) of frame f in bounds / win / sigma_pos / sym */
  int32_t n_maps;
  int32_t reserved;
  uint8_t idx[AIVC_MAX_MAPS];
} aivc_frame_maps;
int aivc_laplace_cdf_windows_batch(const float *sigma, int32_t n, size_t npix, int32_t c,
                                   const aivc_frame_maps *frames, int32_t max_maps, uint16_t *win,
                                   float *sigma_pos, aivc_stream_t stream);
int aivc_laplace_bounds_batch(const float *sigma, const int16_t *q, int32_t n, size_t npix, int32_t c,
                              const aivc_frame_maps *frames, int32_t max_maps, uint32_t *bounds,
                              aivc_stream_t stream);
/* every channel of every frame (pmf mode): bounds[(f * c + ch) * npix + pix] */
int aivc_table_bounds_batch(const uint16_t *table, const int16_t *q, int32_t n, size_t npix, int32_t c,
                            uint32_t *bounds, aivc_stream_t stream);
/* q[f][pix][frames[f].idx[m]] = sym[frames[f].pos_off + m * npix + pix] - 256, every other channel 0 */
int aivc_scatter_symbols_batch(const uint16_t *sym, int32_t n, size_t npix, int32_t c,
                               const aivc_frame_maps *frames, int16_t *q, aivc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Rate estimation (ABI 15) -- logging only, never on the coded path.  The reference prints an estimated rate next to
 * the real one: per section under flag_debug (src/real_life/bitstream.py:307-329), per video in the RESULT lines
 * (src/real_life/encode.py:153-170), from the probabilities of its entropy models
 * (src/layers/entropy_coding/entropy_coder.py:18-30, pdf_estimator.py:27-65 and :185-202).
 * Sums are fp64 and deterministic: lane j of AIVC_RATE_LANES adds elements j, j + L, ... in order, then
 * lanes[j] += lanes[j + s] for s = L/2 .. 1.  `lanes` is caller-owned scratch of AIVC_RATE_LANES doubles.
 * ---------------------------------------------------------------------------------------- */
#define AIVC_RATE_LANES 16384
/* sum[0] = sum over the n packed bounds (aivc_laplace_bounds / aivc_table_bounds) of -log2((c_hi - c_lo) / 2^16): the bits
 * the range coder pays for exactly the CDFs it codes with */
int aivc_bounds_rate(const uint32_t *bounds, size_t n, double *lanes, double *sum, aivc_stream_t stream);
/* EntropyCoder.forward: rate[i] = -log2(clamp(prob[i], p_min, p_max)) (rate may be NULL); sum[0] = their fp64 sum */
int aivc_rate_bits(const float *prob, size_t n, float p_min, float p_max, float *rate, double *lanes, double *sum,
                   aivc_stream_t stream);
/* ParametricPdf.forward, Laplace family, one component: prob[i] = cdf(y + .5) - cdf(y - .5) under
 * Laplace(mu[i], sigma[i] / sqrt(2)); mu NULL = 0 */
int aivc_laplace_prob(const float *y, const float *mu, const float *sigma, size_t n, float *prob, aivc_stream_t stream);
/* BallePdfEstim.forward for integer-valued x in [-256, 256], x laid out [b][c][hw] (NCHW, n = b*c*hw elements):
 * prob[i] = cdf_f32[ch][x + 257] - cdf_f32[ch][x + 256] with cdf_f32 [c][AIVC_LP] from aivc_balle_cdf_table; NaN for
 * values that are not codable symbols */
int aivc_table_prob(const float *x, const float *cdf_f32, size_t n, size_t hw, int32_t c, float *prob, aivc_stream_t stream);

/* ---- quality metrics, fp64 planes [n][h][w] on the device -------------------------
 * Scratch for the three calls below, in bytes (for the largest plane they will see). */
size_t aivc_metrics_workspace(int32_t n, int32_t h, int32_t w);
/* One MS-SSIM scale: out[i][0] = mean SSIM map, out[i][1] = mean contrast-structure (v1/v2) of plane i, with a
 * ws x ws "valid" window win (x) win (HOST pointer, ws <= 11, normalised 1-D Gaussian), C1 = c1, C2 = c2.
 * Replaces ssim() of src/func_util/ms_ssim.py:37-90 (fp32 there) and _SSIMForMultiScale of
 * src/clic21/msssim.py:43-113 (fp64, fftconvolve 'valid'). */
int aivc_ssim_means(const double *a, const double *b, int32_t n, int32_t h, int32_t w, const double *win,
                    int32_t ws, double c1, double c2, double *workspace, double *out, aivc_stream_t stream);
/* 2x2 mean to [n][ceil(h/2)][ceil(w/2)].  Odd sizes read one sample past the end: edge = 0 mirrors without
 * repeating the border (ReflectionPad2d + avg_pool2d, src/func_util/ms_ssim.py:112-121), edge = 1 repeats it
 * (scipy convolve mode='reflect' + [::2], src/clic21/msssim.py:173-175). */
int aivc_pool2x2(const double *in, int32_t n, int32_t h, int32_t w, int32_t edge, double *out,
                 aivc_stream_t stream);
/* out[0] = sum (a[i] - b[i])^2 (src/clic21/metrics.py:58-59, src/model_mngt/loss_function.py:428-433). */
int aivc_sq_err(const double *a, const double *b, size_t count, double *workspace, double *out,
                aivc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AIVC_HIP_H */
