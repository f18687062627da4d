/* TEST INFRASTRUCTURE (oracle/): the twiddle factors of the DFT spec, machine-independent (DFT spec v3 tables, round 6).
 *
 * Rounds 1-5 took a twiddle (cos, -sin)(2 pi k / n) from the box's libm (round 5: "one glibc sincos call"), so the same stream's
 * bytes were only guaranteed equal between the GPU library and this oracle ON ONE BOX.  Now an entry is computed from the integers
 * (k, n) with IEEE-754 double additions, multiplications, divisions and fused multiply-adds only -- no libm call -- so every
 * conforming machine produces the same bits:
 *   1. exact octant reduction in integers: 8 k = o n + r (0 <= r < n); odd octants measure the angle back from the next octant
 *      boundary (rho = n - r), so the reduced angle is theta = (pi / 4) (rho / n) in [0, pi / 4];
 *   2. rho / n, pi / 4 and theta as double-double numbers (hi + lo, ~106 bits);
 *   3. sin and cos of theta by their Taylor series in double-double (16 terms each: the first one left out is below 2^-130);
 *   4. the high word of the normalised sum is the entry; sign and the sin <-> cos swap follow from the octant.
 * The result is the correctly rounded value unless the true value lies within ~2^-100 of a rounding boundary;
 * tests/test_oracle_twiddle.py compares every entry of every compiled-in table size with a 60-digit evaluation (Python decimal),
 * and pins a hash of each table; orc_plan_create checks the same hashes (orc_twiddle_hashes.h).  The product holds its own
 * restatement of this routine (speedy_amd/csrc/spx_twiddle.h); neither includes the other. */
#ifndef ORC_TWIDDLE_H_
#define ORC_TWIDDLE_H_
#include <math.h>
#include <stdint.h>

typedef struct { double h, l; } orc_dd;

static inline orc_dd orc_dd_two_sum(double a, double b) {
  const double s = a + b;
  const double bb = s - a;
  const double e = (a - (s - bb)) + (b - bb);
  orc_dd r = {s, e};
  return r;
}
static inline orc_dd orc_dd_fast_two_sum(double a, double b) { /* |a| >= |b| or a == 0 */
  const double s = a + b;
  const double e = b - (s - a);
  orc_dd r = {s, e};
  return r;
}
static inline orc_dd orc_dd_two_prod(double a, double b) {
  const double p = a * b;
  const double e = fma(a, b, -p);
  orc_dd r = {p, e};
  return r;
}
static inline orc_dd orc_dd_add(orc_dd a, orc_dd b) {
  orc_dd s = orc_dd_two_sum(a.h, b.h);
  const orc_dd t = orc_dd_two_sum(a.l, b.l);
  s.l = s.l + t.h;
  s = orc_dd_fast_two_sum(s.h, s.l);
  s.l = s.l + t.l;
  return orc_dd_fast_two_sum(s.h, s.l);
}
static inline orc_dd orc_dd_mul(orc_dd a, orc_dd b) {
  orc_dd p = orc_dd_two_prod(a.h, b.h);
  const double c1 = a.h * b.l;
  const double c2 = a.l * b.h;
  p.l = p.l + (c1 + c2);
  return orc_dd_fast_two_sum(p.h, p.l);
}
static inline orc_dd orc_dd_div_d(orc_dd a, double d) { /* d a small integer: exact as a double */
  const double q1 = a.h / d;
  const orc_dd p = orc_dd_two_prod(q1, d);
  const double r = ((a.h - p.h) - p.l) + a.l;
  const double q2 = r / d;
  return orc_dd_fast_two_sum(q1, q2);
}

/* cos and sin of 2 pi k / n for integers 0 <= k, 0 < n < 2^24 (k is reduced mod n) */
static inline void orc_sincos_2pi(long k, long n, double* cs, double* sn) {
  k %= n;
  if (k < 0) k += n;
  const long m = 8 * k;
  const int o = (int)(m / n);
  const long r = m - (long)o * n;
  const long rho = (o & 1) ? n - r : r;
  /* theta = (pi / 4) (rho / n) */
  const double q1 = (double)rho / (double)n;
  const double rem = fma(-q1, (double)n, (double)rho); /* exact */
  const double q2 = rem / (double)n;
  const orc_dd x = orc_dd_fast_two_sum(q1, q2);
  const orc_dd pi4 = {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55};
  const orc_dd th = orc_dd_mul(x, pi4);
  const orc_dd th2 = orc_dd_mul(th, th);
  orc_dd ssum = th, sterm = th;
  orc_dd one = {1.0, 0.0};
  orc_dd csum = one, cterm = one;
  for (int j = 1; j <= 16; j++) {
    /* cos: term_j = -term_{j-1} theta^2 / ((2j - 1)(2j));  sin: term_j = -term_{j-1} theta^2 / ((2j)(2j + 1)) */
    cterm = orc_dd_div_d(orc_dd_mul(cterm, th2), (double)((2 * j - 1) * (2 * j)));
    cterm.h = -cterm.h; cterm.l = -cterm.l;
    csum = orc_dd_add(csum, cterm);
    sterm = orc_dd_div_d(orc_dd_mul(sterm, th2), (double)((2 * j) * (2 * j + 1)));
    sterm.h = -sterm.h; sterm.l = -sterm.l;
    ssum = orc_dd_add(ssum, sterm);
  }
  double c = csum.h, s = ssum.h;
  if (rho == 0) { c = 1.0; s = 0.0; }
  double oc, os;
  switch (o) {
    case 0: oc = c; os = s; break;     /* theta                */
    case 1: oc = s; os = c; break;     /* pi/2 - theta'        */
    case 2: oc = -s; os = c; break;    /* pi/2 + theta         */
    case 3: oc = -c; os = s; break;    /* pi - theta'          */
    case 4: oc = -c; os = -s; break;   /* pi + theta           */
    case 5: oc = -s; os = -c; break;   /* 3 pi/2 - theta'      */
    case 6: oc = s; os = -c; break;    /* 3 pi/2 + theta       */
    default: oc = c; os = -s; break;   /* 2 pi - theta'        */
  }
  /* no negative zeros in the tables (the libm tables of rounds 1-5 had none either: cos, -sin of small positive angles) */
  *cs = oc + 0.0;
  *sn = os + 0.0;
}

/* FNV-1a (64 bit) over the bytes of a table of `count` entries (cos, -sin)(2 pi t / den) */
static inline uint64_t orc_twiddle_table_hash(long den, long count) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (long t = 0; t < count; t++) {
    double e[2], s;
    orc_sincos_2pi(t, den, &e[0], &s);
    e[1] = 0.0 - s;
    const unsigned char* b = (const unsigned char*)e;
    for (int i = 0; i < 16; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
  }
  return h;
}
#endif /* ORC_TWIDDLE_H_ */
