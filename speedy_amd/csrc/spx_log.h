// Natural log with a fixed operation sequence, shared by the analysis kernel and the unit-level hooks.
#ifndef SPX_LOG_H_
#define SPX_LOG_H_
#include <hip/hip_runtime.h>

// Natural log: the fdlibm operation sequence of DESIGN.md "log spec" (same as oracle/orc_speedy.c orc_log).
//
// Round 4: two halves.  spx_log_main is the sequence for a positive normal argument away from 1 as ONE straight-line
// computation: its four endings (k == 0 or not, the mantissa inside or outside [0x6147a, 0x6b851]) are two selects instead
// of four divergent branches -- in a 64-lane wave nearly every combination occurs, so the branches cost the sum of all
// endings and kept the compiler from interleaving two terms.  Same values bit for bit: with k == 0 the general ending
// computes 0*ln2_hi - ((u + 0) - f) = -(u - f) = f - u, the k == 0 ending's expression (IEEE subtraction is antisymmetric,
// adding 0 and subtracting from 0 are exact), and this path never returns a zero (x == 1 is a `rare` argument), so no sign
// of zero is involved.  spx_log_finish replaces the value for the rare arguments (zero, negative, subnormal, infinite,
// NaN, |x - 1| < 2^-20 after scaling) by the full sequence's; callers with several terms in flight run every main half
// first and the finishes after, so that the terms' instructions interleave.
struct spx_log_parts {
  double res;
  bool rare;
};

__device__ inline double spx_log_full(double x) {   // the sequence as fdlibm writes it, every case
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
               Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
               Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int k, hx, i, j;
  unsigned lx;
  long long bits = __double_as_longlong(x);
  hx = (int)(bits >> 32);
  lx = (unsigned)bits;
  k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_huge_val();
    if (hx < 0) return __builtin_nan("");
    k -= 54;
    x *= two54;
    bits = __double_as_longlong(x);
    hx = (int)(bits >> 32);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  bits = __double_as_longlong(x);
  bits = (bits & 0xffffffffLL) | ((long long)(unsigned)(hx | (i ^ 0x3ff00000)) << 32);
  x = __longlong_as_double(bits);
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {
    if (f == 0.0) {
      if (k == 0) return 0.0;
      dk = (double)k;
      return dk * ln2_hi + dk * ln2_lo;
    }
    R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    dk = (double)k;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  s = f / (2.0 + f);
  dk = (double)k;
  z = s * s;
  i = hx - 0x6147a;
  w = z * z;
  j = 0x6b851 - hx;
  t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  R = t2 + t1;
  if (i > 0) {
    hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

__device__ __forceinline__ spx_log_parts spx_log_main(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               Lg1 = 6.666666666666735130e-01,
               Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
               Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
  long long bits = __double_as_longlong(x);
  const int hx0 = (int)(bits >> 32);
  int k = (hx0 >> 20) - 1023;
  const int hx = hx0 & 0x000fffff;
  const int i = (hx + 0x95f64) & 0x100000;
  bits = (bits & 0xffffffffLL) | ((long long)(unsigned)(hx | (i ^ 0x3ff00000)) << 32);
  const double xm = __longlong_as_double(bits);
  k += (i >> 20);
  const double f = xm - 1.0;
  const double dk = (double)k;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const bool mid = ((hx - 0x6147a) | (0x6b851 - hx)) > 0;
  const double hfsq = 0.5 * f * f;
  const double m = s * (mid ? (hfsq + R) : (f - R));
  const double lo = dk * ln2_lo;
  const double u = mid ? (hfsq - (m + lo)) : (m - lo);
  spx_log_parts p;
  p.res = dk * ln2_hi - (u - f);
  p.rare = (hx0 < 0x00100000) | (hx0 >= 0x7ff00000) | ((0x000fffff & (2 + hx)) < 3);
  return p;
}
__device__ __forceinline__ double spx_log_finish(const spx_log_parts& p, double x) {
  if (__builtin_expect(p.rare, 0)) return spx_log_full(x);
  return p.res;
}

__device__ __forceinline__ double spx_log(double x) { return spx_log_finish(spx_log_main(x), x); }

#endif  // SPX_LOG_H_
