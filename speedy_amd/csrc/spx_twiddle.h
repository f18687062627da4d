// The DFT spec's twiddle factors, machine-independent (host code; round 6).
//
// cos and sin of 2 pi k / n from the INTEGERS (k, n) with IEEE-754 double additions, multiplications, divisions and fused
// multiply-adds only -- no libm call -- so the tables a plan uploads are the same bits on every conforming machine, and a stream's
// output bytes no longer depend on which glibc built the box's sincos (rounds 1-5: "whatever this box's libm returns", on this
// side and in the oracle).  The method (the oracle restates it in C, oracle/orc_twiddle.h; neither includes the other):
//   * 8 k = o n + r: the octant o and the remainder are exact integers; odd octants measure the angle back from the next octant
//     boundary, so the reduced angle theta = (pi / 4)(rho / n) lies in [0, pi / 4];
//   * rho / n, pi / 4 and theta are double-double numbers (an unevaluated sum hi + lo, ~106 bits);
//   * sin theta and cos theta by their Taylor series in double-double, sixteen terms each;
//   * the high word of the normalised result is the table entry -- the correctly rounded value unless the true one lies within
//     ~2^-100 of a rounding boundary (none does for the compiled-in sizes: tools/twiddle_tables.py evaluates every entry with 60
//     digits, tests/test_oracle_twiddle.py compares, spx_twiddle_hashes.h pins the tables and spx_plan_create checks the pins).
// Build requirement: no fp contraction, no fast-math (the Makefile's -ffp-contract=off -fno-fast-math) -- the pins catch a build
// that strays.
#ifndef SPX_TWIDDLE_H_
#define SPX_TWIDDLE_H_
#include <math.h>
#include <stdint.h>

namespace spx_tw {

struct DD {
  double hi, lo;
};
// error-free transformations
static inline DD sum2(double a, double b) {          // a + b exactly, any magnitudes
  const double s = a + b, v = s - a;
  return {s, (a - (s - v)) + (b - v)};
}
static inline DD quick_sum2(double a, double b) {    // a + b exactly, |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
static inline DD prod2(double a, double b) {         // a * b exactly
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
static inline DD operator+(DD a, DD b) {
  DD s = sum2(a.hi, b.hi);
  const DD t = sum2(a.lo, b.lo);
  s.lo = s.lo + t.hi;
  s = quick_sum2(s.hi, s.lo);
  s.lo = s.lo + t.lo;
  return quick_sum2(s.hi, s.lo);
}
static inline DD operator*(DD a, DD b) {
  DD p = prod2(a.hi, b.hi);
  const double x1 = a.hi * b.lo;
  const double x2 = a.lo * b.hi;
  p.lo = p.lo + (x1 + x2);
  return quick_sum2(p.hi, p.lo);
}
static inline DD over(DD a, double d) {               // a / d, d a small integer
  const double q1 = a.hi / d;
  const DD p = prod2(q1, d);
  const double rest = ((a.hi - p.hi) - p.lo) + a.lo;
  return quick_sum2(q1, rest / d);
}
static inline DD neg(DD a) { return {-a.hi, -a.lo}; }

// cos and sin of 2 pi k / n; 0 < n < 2^24, any k >= 0
static inline void sincos_2pi(long k, long n, double* cs, double* sn) {
  k %= n;
  if (k < 0) k += n;
  const long eighths = 8 * k;
  const int octant = (int)(eighths / n);
  const long r = eighths - (long)octant * n;
  const long rho = (octant & 1) ? n - r : r;
  const double q = (double)rho / (double)n;
  const DD frac = quick_sum2(q, fma(-q, (double)n, (double)rho) / (double)n);   // rho / n (the fma's remainder is exact)
  const DD quarter_pi = {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55};
  const DD theta = frac * quarter_pi;
  const DD theta2 = theta * theta;
  DD c = {1.0, 0.0}, s = theta, ct = c, st = s;
  for (int j = 1; j <= 16; j++) {
    ct = neg(over(ct * theta2, (double)((2 * j - 1) * (2 * j))));
    c = c + ct;
    st = neg(over(st * theta2, (double)((2 * j) * (2 * j + 1))));
    s = s + st;
  }
  double cv = c.hi, sv = s.hi;
  if (rho == 0) { cv = 1.0; sv = 0.0; }
  double oc, os;
  switch (octant) {
    case 0: oc = cv; os = sv; break;
    case 1: oc = sv; os = cv; break;
    case 2: oc = -sv; os = cv; break;
    case 3: oc = -cv; os = sv; break;
    case 4: oc = -cv; os = -sv; break;
    case 5: oc = -sv; os = -cv; break;
    case 6: oc = sv; os = -cv; break;
    default: oc = cv; os = -sv; break;
  }
  *cs = oc + 0.0;   // (no negative zeros in the tables)
  *sn = os + 0.0;
}

// one table entry as the plans store it: (cos, -sin)(2 pi t / den)
static inline void entry(long t, long den, double* out2) {
  double c, s;
  sincos_2pi(t, den, &c, &s);
  out2[0] = c;
  out2[1] = 0.0 - s;
}
static inline uint64_t fnv1a(const void* p, size_t bytes) {
  uint64_t h = 0xcbf29ce484222325ull;
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
  return h;
}

}  // namespace spx_tw
#endif  // SPX_TWIDDLE_H_
