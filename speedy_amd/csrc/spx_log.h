// Natural log with a fixed operation sequence, shared by the analysis kernel and the unit-level hooks: spec v2 (table-driven,
// division-free, for the float arguments the path produces; below) and spec v1 (the fdlibm sequence, every other double).
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
// Correctly rounded n / d WITHOUT the scaling and fix-up halves of the IEEE division sequence (v_div_scale x 2, v_div_fmas'
// scale flag, v_div_fixup): the same reciprocal refinement and the same two correction steps the compiler emits, which IS the
// whole sequence when neither operand is anywhere near the exponent range's ends (the scale steps multiply by 1, v_div_fmas is
// a plain fma, v_div_fixup returns the quotient as it is).  Eight instructions instead of eleven.  Callers guarantee normal,
// finite, non-zero operands whose exponents differ by far less than the format's range.
__device__ __forceinline__ double spx_fdiv64(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = n * r;
  const double res = __builtin_fma(-d, q, n);
  return __builtin_fma(res, r, q);
}
__device__ __forceinline__ float spx_fdiv32(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  float t = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(t, r, q);
  t = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(t, r, q);
}
// (float)sqrt(x) for a double x: the compiler's correctly rounded sequence (v_rsq_f64, one coupled refinement of root and half
// reciprocal root, two corrections of the root) without its scaling by 2^256 for arguments below 2^-767 and without its
// selects for zero / infinity -- arguments outside [2^-767, 2^1000], zero among them, take the library sequence.
__device__ __forceinline__ float spx_sqrt64_to_f32(double x) {
  if (__builtin_expect(!(x >= 0x1p-767 && x <= 0x1p+1000), 0)) return (float)__builtin_sqrt(x);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return (float)g;
}

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
  // (2 + f is in [1.7, 2.42]; |f| >= 2^-20 wherever this value is used: the near-one arguments are `rare`.  For them, for zero
  // and for subnormal x the quotient below is never used -- but it must not trap: f = 0 gives 0 * r = 0, no harm)
  const double s = spx_fdiv64(f, 2.0 + f);
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

__device__ __forceinline__ double spx_log_v1(double x) { return spx_log_finish(spx_log_main(x), x); }

// ---- log spec v2 (round 5; oracle/orc_speedy.c orc_log_v2_f32 is the same sequence) ----
// The argument of every log on the path is a FLOAT quotient promoted to double (speedy.c:716-717).  For a positive normal float
//   bits = pattern of x;  t = bits - 0x3f328000;  k = t >> 23 (arithmetic);  i = (t >> 16) & 127;
//   z = the float with pattern bits - (t & 0xff800000):  x = 2^k z,  z in [0.697265625, 1.39453125), bucket 77 centred on 1.0
//   r = fma(z, invc[i], -1): EXACT (24-bit x 24-bit significands);  |r| <= 2^-8
//   w = fma(k, Ln2hi, logc_hi[i]): EXACT (Ln2hi has 42 significant bits, logc_hi is a multiple of 2^-43)
//   hi = w + r;  lo = (w - hi) + r (exact two-sum);  lo = fma(k, Ln2lo, lo + logc_lo[i])
//   p = fma(1/7, r, -1/6); p = fma(p, r, 1/5); p = fma(p, r, -1/4); p = fma(p, r, 1/3); p = fma(p, r, -1/2)
//   log x = fma(r * r, p, lo) + hi
// No division, no branch: 14 fp64 operations, ten 32-bit ones and one 16-byte table read, against the fdlibm sequence's 45.
// Over ALL 2 130 706 432 positive normal floats it equals glibc's log for 2 130 640 559 and is 1 ulp away for the other 65 873
// (oracle/orc_logcheck.c); the GPU reproduces the oracle's results block for block on the whole domain (spx_debug_log_check).
// Any other double (zero, negative, subnormal-as-float, infinite, NaN, not a float: only the unit-level hooks can feed those)
// takes v1, on both sides.
struct SpxLogEntry { double logc_hi; float invc, logc_lo; };   // 16 bytes: one ds_read_b128 / global_load_dwordx4
#include "spx_log_table.h"
static __device__ const SpxLogEntry spx_log_table_dev[128] = {SPX_LOG_TABLE_ENTRIES};
#define SPX_LOG_TABLE_BYTES (128 * 16)

__device__ __forceinline__ double spx_log_v2_eval(float xf, const SpxLogEntry& e, int k, float z) {
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double r = __builtin_fma((double)z, (double)e.invc, -1.0);
  const double kd = (double)k;
  const double w = __builtin_fma(kd, Ln2hi, e.logc_hi);
  const double hi = w + r;
  double lo = (w - hi) + r;
  lo = __builtin_fma(kd, Ln2lo, lo + (double)e.logc_lo);
  const double r2 = r * r;
  double p = __builtin_fma(0x1.2492492492492p-3, r, -0x1.5555555555555p-3);
  p = __builtin_fma(p, r, 0x1.999999999999ap-3);
  p = __builtin_fma(p, r, -0.25);
  p = __builtin_fma(p, r, 0x1.5555555555555p-2);
  p = __builtin_fma(p, r, -0.5);
  (void)xf;
  return __builtin_fma(r2, p, lo) + hi;
}
// xf: a positive normal float (the caller has checked); tab: the table in LDS or in constant memory
__device__ __forceinline__ double spx_log_v2_f32(float xf, const SpxLogEntry* __restrict__ tab) {
  const unsigned bits = __float_as_uint(xf);
  const unsigned t = bits - 0x3f328000u;
  const int k = (int)t >> 23;
  const float z = __uint_as_float(bits - (t & 0xff800000u));
  return spx_log_v2_eval(xf, tab[(t >> 16) & 127u], k, z);
}
__device__ __forceinline__ bool spx_log_v2_domain(float xf) { return __builtin_amdgcn_classf(xf, 0x100); }   // +normal
// The log of the spec for any double (the one-lane hook kernels): v2 for positive normal floats, v1 otherwise.
__device__ __forceinline__ double spx_log(double x) {
#ifdef SPX_LOG_V1
  return spx_log_v1(x);
#else
  const float xf = (float)x;
  if ((double)xf == x && spx_log_v2_domain(xf)) return spx_log_v2_f32(xf, spx_log_table_dev);
  return spx_log_v1(x);
#endif
}

#endif  // SPX_LOG_H_
