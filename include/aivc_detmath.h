/*
 * aivc_detmath.h -- deterministic transcendentals shared by the HIP kernels and the CPU oracle.
 *
 * The codec's encoder and decoder must derive bit-identical sigma / CDF values, and the HIP path
 * must be checkable bit for bit against the CPU oracle, so nothing here may depend on a libm or
 * on device intrinsics whose rounding differs between hosts and GPUs.  Everything is built from
 * IEEE-754 binary64 add / mul / fma / div and integer bit operations, which are correctly rounded
 * on x86-64 and on gfx950 alike (compile with -ffp-contract=off; every fused operation below is an
 * explicit fma()).  Results are accurate to a few binary64 ulps, i.e. they round to the correctly
 * rounded binary32 value except in ~1e-9 of the cases, which is the accuracy class of the fp32
 * torch/SLEEF kernels the reference uses (src/real_life/bitstream.py:127-154,
 * src/layers/misc/misc_layers.py:203-219, src/layers/entropy_coding/pdf_estimator.py:204-245).
 *
 * This header is NOT part of the oracle: it is a numerical primitive of the product that the
 * oracle re-uses so that "same inputs -> same bits" is a meaningful test.  Its accuracy is pinned
 * independently in tests/test_detmath.py against libm and against torch-generated golden vectors.
 */
#ifndef AIVC_DETMATH_H
#define AIVC_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define AIVC_HD __host__ __device__ static inline
#define AIVC_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define AIVC_RINT(a) __builtin_rint((a))
#define AIVC_FABS(a) __builtin_fabs((a))
#else
#include <math.h>
#define AIVC_HD static inline
#define AIVC_FMA(a, b, c) fma((a), (b), (c))
#define AIVC_RINT(a) rint((a))
#define AIVC_FABS(a) fabs((a))
#endif

AIVC_HD double aivc_bits_to_f64(uint64_t u) {
  union { uint64_t u; double d; } c;
  c.u = u;
  return c.d;
}
AIVC_HD uint64_t aivc_f64_to_bits(double d) {
  union { uint64_t u; double d; } c;
  c.d = d;
  return c.u;
}
/* 2^k for -1022 <= k <= 1023 */
AIVC_HD double aivc_pow2i(int k) { return aivc_bits_to_f64((uint64_t)(k + 1023) << 52); }

/* exp(x) for finite -745 <= x <= 709: the arithmetic of aivc_det_exp without its range tests */
AIVC_HD double aivc_det_exp_core(double x) {
  const double INV_LN2 = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01; /* 0x3FE62E42FEE00000 */
  const double LN2_LO = 1.90821492927058770002e-10; /* 0x3DEA39EF35793C76 */
  const double kd = AIVC_RINT(x * INV_LN2);
  double r = AIVC_FMA(-kd, LN2_HI, x);
  r = AIVC_FMA(-kd, LN2_LO, r);
  /* Taylor, degree 13: |r| <= 0.3466 -> truncation < 2^-57 */
  double p = 1.0 / 6227020800.0;
  p = AIVC_FMA(p, r, 1.0 / 479001600.0);
  p = AIVC_FMA(p, r, 1.0 / 39916800.0);
  p = AIVC_FMA(p, r, 1.0 / 3628800.0);
  p = AIVC_FMA(p, r, 1.0 / 362880.0);
  p = AIVC_FMA(p, r, 1.0 / 40320.0);
  p = AIVC_FMA(p, r, 1.0 / 5040.0);
  p = AIVC_FMA(p, r, 1.0 / 720.0);
  p = AIVC_FMA(p, r, 1.0 / 120.0);
  p = AIVC_FMA(p, r, 1.0 / 24.0);
  p = AIVC_FMA(p, r, 1.0 / 6.0);
  p = AIVC_FMA(p, r, 0.5);
  p = AIVC_FMA(p, r, 1.0);
  p = AIVC_FMA(p, r, 1.0);
  const int k = (int)kd;
  const int k1 = k / 2;
  const int k2 = k - k1;
  return (p * aivc_pow2i(k1)) * aivc_pow2i(k2);
}
/* exp(x), |rel err| < ~2 ulp(binary64) */
AIVC_HD double aivc_det_exp(double x) {
  if (x != x) return x;
  if (x > 709.0) return aivc_bits_to_f64(0x7FF0000000000000ull);
  if (x < -745.0) return 0.0;
  return aivc_det_exp_core(x);
}

/* x * (1 + x/2 + x^2/6 + ...), degree 16 in total: expm1(x) for |x| < 0.34 */
AIVC_HD double aivc_det_expm1_small(double x) {
  double p = 1.0 / 20922789888000.0; /* 1/16! */
  p = AIVC_FMA(p, x, 1.0 / 1307674368000.0);
  p = AIVC_FMA(p, x, 1.0 / 87178291200.0);
  p = AIVC_FMA(p, x, 1.0 / 6227020800.0);
  p = AIVC_FMA(p, x, 1.0 / 479001600.0);
  p = AIVC_FMA(p, x, 1.0 / 39916800.0);
  p = AIVC_FMA(p, x, 1.0 / 3628800.0);
  p = AIVC_FMA(p, x, 1.0 / 362880.0);
  p = AIVC_FMA(p, x, 1.0 / 40320.0);
  p = AIVC_FMA(p, x, 1.0 / 5040.0);
  p = AIVC_FMA(p, x, 1.0 / 720.0);
  p = AIVC_FMA(p, x, 1.0 / 120.0);
  p = AIVC_FMA(p, x, 1.0 / 24.0);
  p = AIVC_FMA(p, x, 1.0 / 6.0);
  p = AIVC_FMA(p, x, 0.5);
  p = AIVC_FMA(p, x, 1.0);
  return p * x;
}
/* expm1(x) */
AIVC_HD double aivc_det_expm1(double x) {
  if (x != x) return x;
  if (AIVC_FABS(x) < 0.34) return aivc_det_expm1_small(x);
  if (x < -60.0) return -1.0;
  return aivc_det_exp(x) - 1.0;
}

/* log(x) for finite x > 0 (normal or subnormal) */
AIVC_HD double aivc_det_log(double x) {
  uint64_t b = aivc_f64_to_bits(x);
  int e = 0;
  if ((b >> 52) == 0) { /* subnormal: scale up */
    x = x * 18014398509481984.0; /* 2^54 */
    b = aivc_f64_to_bits(x);
    e = -54;
  }
  e += (int)((b >> 52) & 0x7FF) - 1023;
  uint64_t mb = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m = aivc_bits_to_f64(mb); /* [1, 2) */
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    e += 1;
  }
  const double f = (m - 1.0) / (m + 1.0); /* |f| <= 0.1716 */
  const double s = f * f;
  /* 2*atanh(f) = 2f (1 + s/3 + s^2/5 + ... + s^12/25) */
  double p = 1.0 / 25.0;
  p = AIVC_FMA(p, s, 1.0 / 23.0);
  p = AIVC_FMA(p, s, 1.0 / 21.0);
  p = AIVC_FMA(p, s, 1.0 / 19.0);
  p = AIVC_FMA(p, s, 1.0 / 17.0);
  p = AIVC_FMA(p, s, 1.0 / 15.0);
  p = AIVC_FMA(p, s, 1.0 / 13.0);
  p = AIVC_FMA(p, s, 1.0 / 11.0);
  p = AIVC_FMA(p, s, 1.0 / 9.0);
  p = AIVC_FMA(p, s, 1.0 / 7.0);
  p = AIVC_FMA(p, s, 1.0 / 5.0);
  p = AIVC_FMA(p, s, 1.0 / 3.0);
  p = AIVC_FMA(p, s, 1.0);
  const double lm = 2.0 * f * p;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double ed = (double)e;
  return AIVC_FMA(ed, LN2_HI, AIVC_FMA(ed, LN2_LO, lm));
}

/* log1p(y) for y >= 0 */
AIVC_HD double aivc_det_log1p(double y) {
  if (y < 1e-5) { /* y - y^2/2 + y^3/3 : rel err < 1e-16 */
    double p = AIVC_FMA(y, 1.0 / 3.0, -0.5);
    p = AIVC_FMA(p, y, 1.0);
    return p * y;
  }
  const double u = 1.0 + y;
  const double c = y - (u - 1.0); /* rounding error of 1+y */
  return aivc_det_log(u) + c / u;
}

/* ---- fp32 wrappers mirroring the reference's fp32 torch ops ---- */
AIVC_HD float aivc_expf_det(float x) { return (float)aivc_det_exp((double)x); }
AIVC_HD float aivc_expm1f_det(float x) {
  /* exp(-17.5) < 2^-25, so fp32(expm1(x)) is exactly -1 below that (saves the fp64 work in the
   * saturated tails of the Laplace CDF) */
  if (x < -17.5f) return -1.0f;
  return (float)aivc_det_expm1((double)x);
}
/* torch.sigmoid(float): 1 / (1 + exp(-x)) evaluated in fp32 */
AIVC_HD float aivc_sigmoidf_det(float x) {
  const float e = aivc_expf_det(-x);
  const float d = 1.0f + e;
  return 1.0f / d;
}
/* torch.tanh(float) */
AIVC_HD float aivc_tanhf_det(float x) {
  const double ax = AIVC_FABS((double)x);
  if (ax > 20.0) return x > 0 ? 1.0f : -1.0f;
  const double em = aivc_det_expm1(2.0 * ax);
  const double t = em / (em + 2.0);
  return (float)(x < 0 ? -t : t);
}
/* torch.nn.functional.softplus(float) with beta = 1, threshold = 20 */
AIVC_HD float aivc_softplusf_det(float x) {
  if (x > 20.0f) return x;
  return (float)aivc_det_log1p(aivc_det_exp((double)x));
}

/* torch.pow(float tensor, python float) for a > 0 */
AIVC_HD float aivc_powf_det(float a, float e) {
  if (e == 0.0f) return 1.0f;
  if (e == 1.0f) return a;
  if (a == 0.0f) return 0.0f;
  return (float)aivc_det_exp((double)e * aivc_det_log((double)a));
}
AIVC_HD float aivc_gain_interp_one(float g_r, float g_t, float l) {
  const float a = g_r < 0.0f ? -g_r : g_r, b = g_t < 0.0f ? -g_t : g_t;
  return aivc_powf_det(a, l) * aivc_powf_det(b, 1.0f - l);
}

/* Laplace(0, sigma/sqrt(2)).cdf(t) in the reference's fp32 op order
 * (torch.distributions.Laplace.cdf: 0.5 - 0.5 * sign(t) * expm1(-|t| / b)). */
AIVC_HD float aivc_laplace_cdf(float t, float sigma) {
  const float b = sigma / 1.41421354f; /* sqrt(fp32 2.0) */
  const float at = t < 0.0f ? -t : t;
  const float a = at / b;
  const float e = aivc_expm1f_det(-a);
  const float hs = t < 0.0f ? -0.5f : (t > 0.0f ? 0.5f : 0.0f);
  return 0.5f - hs * e;
}
/* torchac float -> uint16 CDF quantisation for point k of an Lp = 514 row:
 * int16 cast of round(cdf * (2^16 - (Lp - 1))) then + k, wrapping. */
AIVC_HD uint16_t aivc_cdf_quant(float cdf, int k) {
#if defined(__HIPCC__)
  const float r = __builtin_rintf(cdf * 65023.0f);
#else
  const float r = rintf(cdf * 65023.0f);
#endif
  return (uint16_t)(((int32_t)r + k) & 0xFFFF);
}
AIVC_HD uint16_t aivc_laplace_cdf_u16(int k, float sigma) {
  return aivc_cdf_quant(aivc_laplace_cdf((float)k - 256.5f, sigma), k);
}
/* The scale of the Laplace law as aivc_laplace_cdf derives it from sigma (one IEEE division). */
AIVC_HD float aivc_laplace_scale(float sigma) { return sigma / 1.41421354f; }
/* aivc_laplace_cdf_u16(k, sigma) for b = aivc_laplace_scale(sigma), sigma > 0 and not NaN, 0 <= k <= 512: the same
 * operations on the same values, laid out for a wavefront that evaluates one entry per lane in the range decoder's rare
 * path (csrc/entropy.hip) -- the argument of expm1 is clamped to the saturation point instead of branching around the
 * fp64 work (expm1f(x) = -1 for x < -17.5 either way), and exp() runs without its NaN / overflow / underflow tests
 * (x is in [-17.5, -0.34] there).  Equality with aivc_laplace_cdf_u16 over sigma and k is a test
 * (tests/test_oracle_golden.py::test_laplace_tail_entries_equal_row_entries). */
AIVC_HD uint16_t aivc_laplace_cdf_u16_scale(int k, float b) {
  const float t = (float)k - 256.5f; /* never 0 */
  const float at = t < 0.0f ? -t : t;
  const float a = at / b;
  const float x = -a;
  const float xc = x < -17.5f ? -17.5f : x;
  const double xd = (double)xc;
  const double em = AIVC_FABS(xd) < 0.34 ? aivc_det_expm1_small(xd) : aivc_det_exp_core(xd) - 1.0;
  const float e = x < -17.5f ? -1.0f : (float)em;
  const float hs = t < 0.0f ? -0.5f : 0.5f;
  return aivc_cdf_quant(0.5f - hs * e, k);
}

/* ---- rate estimation (logging only; csrc/rate.hip) ---------------------------------------------------------------
 * -log2 of a probability, the reference's EntropyCoder.forward (src/layers/entropy_coding/entropy_coder.py:25-30):
 * clamp(p, p_min, p_max) then -log2, one rounding to fp32. */
AIVC_HD float aivc_rate_of_prob(float p, float p_min, float p_max) {
  const float c = p < p_min ? p_min : (p > p_max ? p_max : p); /* (a NaN stays a NaN, as torch.clamp) */
  if (!(c == c)) return c;
  return (float)(-(aivc_det_log((double)c) * 1.4426950408889634)); /* 1 / ln 2 */
}
/* bits the range coder pays for a symbol whose CDF bounds are packed as aivc_laplace_bounds / aivc_table_bounds pack
 * them (c_lo | c_hi << 16, c_hi = 0 meaning 2^16): -log2((c_hi - c_lo) / 2^16), fp64 */
AIVC_HD double aivc_rate_of_bounds(uint32_t b) {
  const uint32_t lo = b & 0xFFFFu, hi16 = b >> 16;
  const uint32_t hi = hi16 ? hi16 : 0x10000u;
  if (hi <= lo) return 16.0; /* not a codable symbol: priced as the smallest probability */
  return 16.0 - aivc_det_log((double)(hi - lo)) * 1.4426950408889634;
}
/* P(bin of y) under Laplace(mu, sigma/sqrt(2)): cdf(y + .5) - cdf(y - .5) in the reference's fp32 op order
 * (ParametricPdf.forward, src/layers/entropy_coding/pdf_estimator.py:52-63) */
AIVC_HD float aivc_laplace_bin_prob(float y, float mu, float sigma) {
  const float up = (y + 0.5f) - mu, dn = (y - 0.5f) - mu;
  return aivc_laplace_cdf(up, sigma) - aivc_laplace_cdf(dn, sigma);
}
/* P(bin of x) from a row of the factorised prior's CDF at k - 256.5, k = 0..513 (BallePdfEstim.forward,
 * src/layers/entropy_coding/pdf_estimator.py:196-202, for the integer-valued x of inference): x + .5 is point x + 257,
 * x - .5 point x + 256; values outside [-256, 256] (not codable) or not integers give NaN */
AIVC_HD float aivc_table_bin_prob(float x, const float *cdf_row) {
  if (!(x >= -256.0f && x <= 256.0f)) return __builtin_nanf("");
  const int k = (int)x;
  if ((float)k != x) return __builtin_nanf("");
  return cdf_row[k + 257] - cdf_row[k + 256];
}

#endif /* AIVC_DETMATH_H */
