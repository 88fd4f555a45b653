/*
 * tfx_pow.h - (v / v0) ** delta for exponents that are NOT integers in 1..8: one operation sequence for the HIP
 * kernels and the CPU oracle, so both sides get the same bits.
 *
 * The reference computes `(v / me[v0i]) ** me[deltai]` with NumPy's float32 power (gym_traffic/envs/traffic_env.py:56),
 * a platform SIMD routine within an ulp of the true power; the device's powf is another such routine.  Neither is a
 * contract two implementations can share bit for bit.  This one is: binary64 arithmetic only (+, -, *, /, compares,
 * integer bit moves - each correctly rounded on the host and on gfx950, no fused multiply-add: both sides compile with
 * -ffp-contract=off), fixed evaluation order, one final rounding to binary32.
 *
 *   q = 2^e m, m in [sqrt(1/2), sqrt(2))     (exponent and mantissa taken from the bits)
 *   ln m   = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716: odd series to s^23 (next term < 1e-19)
 *   y      = delta * (e + ln m * log2(e))
 *   2^y    = 2^n * exp(f ln 2), n = nearest integer to y, |f| <= 1/2: Taylor polynomial of degree 13 (next term 4e-18)
 * The binary64 result carries a relative error of a few 1e-16 * max(1, |y|): against binary32's 6e-8 the final rounding
 * is the correctly rounded power except for values within ~1e-8 ulp of a rounding boundary.  The integer exponents 1..8
 * keep their multiply chains (powi_cr): for them this function is not used and nothing changes.
 *
 * Domain: q >= 0 (speeds are clamped at 0, v0 > 0), delta > 0 finite.  q = 0 -> 0, q = 1 -> 1, q = +inf -> +inf,
 * NaN -> NaN; results beyond binary32's range round to +inf / 0 like any conversion.
 */
#ifndef TFX_POW_H
#define TFX_POW_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TFX_POW_FN __host__ __device__ static inline
#else
#define TFX_POW_FN static inline
#endif

TFX_POW_FN double tfx_pow_from_bits(uint64_t b) {
  double x;
  memcpy(&x, &b, sizeof x);
  return x;
}

TFX_POW_FN float tfx_pow_det(float qf, float deltaf) {
  if (!(qf == qf)) return qf;  /* NaN */
  if (qf == 0.0f) return 0.0f;
  if (qf == 1.0f) return 1.0f;
  const double q = (double)qf, delta = (double)deltaf;
  if (q > 3.5e38) return qf;   /* +inf (the largest finite binary32 is 3.4028235e38) */
  /* exponent and mantissa: every positive binary32 - subnormals too - is a NORMAL binary64 */
  uint64_t bits;
  memcpy(&bits, &q, sizeof bits);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  double m = tfx_pow_from_bits((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);  /* [1, 2) */
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    e = e + 1;
  }
  const double s = (m - 1.0) / (m + 1.0);
  const double s2 = s * s;
  double p = 1.0 / 23.0;
  p = p * s2 + 1.0 / 21.0;
  p = p * s2 + 1.0 / 19.0;
  p = p * s2 + 1.0 / 17.0;
  p = p * s2 + 1.0 / 15.0;
  p = p * s2 + 1.0 / 13.0;
  p = p * s2 + 1.0 / 11.0;
  p = p * s2 + 1.0 / 9.0;
  p = p * s2 + 1.0 / 7.0;
  p = p * s2 + 1.0 / 5.0;
  p = p * s2 + 1.0 / 3.0;
  p = p * s2 + 1.0;
  const double lnm = (2.0 * s) * p;
  double y = delta * ((double)e + lnm * 1.4426950408889634);
  if (y > 200.0) y = 200.0;    /* far beyond binary32 either way: the conversion below makes it +inf / 0 */
  if (y < -200.0) y = -200.0;
  const double yn = (y >= 0.0) ? (double)(int)(y + 0.5) : -(double)(int)(0.5 - y);
  const int n = (int)yn;
  const double t = (y - yn) * 0.6931471805599453;
  double r = 1.0 / 6227020800.0;           /* 1/13! */
  r = r * t + 1.0 / 479001600.0;           /* 1/12! */
  r = r * t + 1.0 / 39916800.0;
  r = r * t + 1.0 / 3628800.0;
  r = r * t + 1.0 / 362880.0;
  r = r * t + 1.0 / 40320.0;
  r = r * t + 1.0 / 5040.0;
  r = r * t + 1.0 / 720.0;
  r = r * t + 1.0 / 120.0;
  r = r * t + 1.0 / 24.0;
  r = r * t + 1.0 / 6.0;
  r = r * t + 0.5;
  r = r * t + 1.0;
  r = r * t + 1.0;
  const double scale = tfx_pow_from_bits((uint64_t)(n + 1023) << 52);  /* 2^n, |n| <= 200 */
  return (float)(r * scale);
}

#endif /* TFX_POW_H */
