/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_speedy.h).
 *
 * CPU restatement of the reference's per-frame analysis:  reference speedy.c.
 * Every function cites the reference lines it follows.  Floating-point promotion points are kept
 * exactly as the reference's C expressions evaluate under FLT_EVAL_METHOD == 0 (x86-64 SSE2), with
 * no fused multiply-add (build with -ffp-contract=off): the reference is built with plain
 * `gcc -g` (reference Makefile:13).
 *
 * Two deliberate, documented departures (both far inside the 1e-4 float tolerance of north_star):
 *   1. the FFT: the reference links FFTW3 (double) or kissfft (float) (speedy.c:39-43,438-473),
 *      neither of which exists in this image.  The transform here is the repo's own "DFT spec"
 *      (DESIGN.md): a W-point double-precision mixed-radix Stockham transform of the packed real
 *      frame followed by the real-input untangle, magnitude = (float)sqrt(re*re + im*im).
 *   2. log(): the reference calls libm's log (speedy.c:716) on a FLOAT quotient promoted to double (speedy.c:716-717:
 *      float + float, float / float).  orc_log_spec below is a fixed operation sequence so that the HIP kernel can
 *      reproduce it bit for bit: for a positive normal float argument "log spec v2" (round 5: a 128-entry table, an exact
 *      argument reduction r = z * invc - 1, a degree-7 polynomial with fused multiply-adds), for every other
 *      double "log spec v1" (orc_log: the classic fdlibm sequence, no FMA).  Both agree with glibc's log to <= 1 ulp -- v2
 *      over ALL 2^31 positive floats (oracle/orc_logcheck.c runs them; tests/test_oracle_dft_log.py).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* sincos */
#endif
#include "orc_speedy.h"

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define kFrameRateHz 100.0 /* speedy.c:90 */
#define kMinimumSpeed 0.01 /* speedy.c:92 */
#define ORC_MAX_HYST (12 + 12 + 1)

/* log spec v2: for x = a positive normal float.
 *   bits = pattern of x;  t = bits - 0x3f328000;  k = t >> 23 (arithmetic);  i = (t >> 16) & 127;
 *   z = float with pattern bits - (t & 0xff800000): x = 2^k z, z in [0.697265625, 1.39453125), bucket 77 centred on 1.0
 *   r = fma(z, invc[i], -1): EXACT (24-bit x 24-bit significands; the difference from 1 is below 2^-7.9)
 *   log x = k ln2 + logc[i] + log1p(r),  logc[i] = logc_hi[i] (a multiple of 2^-43) + logc_lo[i] (a float):
 *     w = fma(k, Ln2hi, logc_hi[i])  (EXACT: Ln2hi has 42 significant bits);  hi = w + r;  lo = (w - hi) + r  (exact two-sum: w = 0
 *     or |w| > |r|);  lo = fma(k, Ln2lo, lo + logc_lo[i])
 *     p = fma(1/7, r, -1/6); p = fma(p, r, 1/5); p = fma(p, r, -1/4); p = fma(p, r, 1/3); p = fma(p, r, -1/2)
 *     result = fma(r * r, p, lo) + hi
 * Every operation is an IEEE-754 double operation (fma = one rounding), so the sequence means the same on any machine. */
#include "orc_log_table.h"
static const struct { double logc_hi; float invc, logc_lo; } orc_log_tab[128] = {ORC_LOG_TABLE_ENTRIES};
static int orc_log_spec_v = 2;
void orc_set_log_spec(int v) { orc_log_spec_v = v == 1 ? 1 : 2; } /* 1: the fdlibm sequence for every argument (round 1-4's spec) */
int orc_get_log_spec(void) { return orc_log_spec_v; }
double orc_log(double x);
double orc_log_v2_f32(float xf) { /* xf positive, normal, finite */
  static const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  uint32_t bits, t, zb;
  float z;
  memcpy(&bits, &xf, 4);
  t = bits - 0x3f328000u;
  const int32_t k = (int32_t)t >> 23;
  const uint32_t i = (t >> 16) & 127u;
  zb = bits - (t & 0xff800000u);
  memcpy(&z, &zb, 4);
  const double r = fma((double)z, (double)orc_log_tab[i].invc, -1.0);
  const double kd = (double)k;
  const double w = fma(kd, Ln2hi, orc_log_tab[i].logc_hi);
  const double hi = w + r;
  double lo = (w - hi) + r;
  lo = fma(kd, Ln2lo, lo + (double)orc_log_tab[i].logc_lo);
  const double r2 = r * r;
  double p = fma(0x1.2492492492492p-3, r, -0x1.5555555555555p-3); /* 1/7, -1/6 */
  p = fma(p, r, 0x1.999999999999ap-3);                            /* 1/5 */
  p = fma(p, r, -0.25);
  p = fma(p, r, 0x1.5555555555555p-2);                            /* 1/3 */
  p = fma(p, r, -0.5);
  return fma(r2, p, lo) + hi;
}
/* The log the analysis uses: v2 where the argument is a positive normal float (always, on the path: speedy.c:716-717 forms
 * a float quotient of two floats >= 2.2e-16), v1 for anything else a caller of the unit-level hooks may feed it. */
double orc_log_spec(double x) {
  const float xf = (float)x;
  if (orc_log_spec_v == 2 && (double)xf == x && xf >= 0x1p-126f && xf <= 0x1.fffffep+127f) return orc_log_v2_f32(xf);
  return orc_log(x);
}

/* ------------------------------------------------------------------------------------------ */
/* First-order filter, speedy.c:50-88                                                          */
void orc_fof_design(orc_fof* f, float tc) {
  f->state = 0.0;
  if (tc > 0) {
    f->alpha = exp(-1.0 / tc); /* speedy.c:67: double exp, rounded to float on store */
  } else {
    f->alpha = 0.0;
  }
}
float orc_fof_iterate(orc_fof* f, float input) {
  /* speedy.c:74 — all-float expression: (1-alpha) float, two float products, float sum */
  f->state = (1 - f->alpha) * input + f->alpha * f->state;
  return f->state;
}
void orc_fof_reset(orc_fof* f) { f->state = 0; }
void orc_fof_set_state(orc_fof* f, float s) { f->state = s; }

/* ------------------------------------------------------------------------------------------ */
/* log spec: fdlibm's e_log.c algorithm, fixed operation order, no FMA.                        */
double orc_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                      two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                      Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                      Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                      Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int32_t k, hx, i, j;
  uint32_t lx;
  uint64_t bits;
  memcpy(&bits, &x, 8);
  hx = (int32_t)(bits >> 32);
  lx = (uint32_t)bits;
  k = 0;
  if (hx < 0x00100000) { /* x < 2**-1022 */
    if (((hx & 0x7fffffff) | lx) == 0) return -HUGE_VAL;
    if (hx < 0) return NAN;
    k -= 54;
    x *= two54;
    memcpy(&bits, &x, 8);
    hx = (int32_t)(bits >> 32);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  memcpy(&bits, &x, 8);
  bits = (bits & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
  memcpy(&x, &bits, 8);
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2**-20 */
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

/* ------------------------------------------------------------------------------------------ */
/* DFT spec.  Stockham autosort, decimation in frequency, mixed radix.             */
/* Radix order: all 4s, then a 2, then 3s, then 5s, then remaining primes ascending.           */
#define ORC_MAX_STAGES 32
typedef struct orc_plan_s {
  int n;
  int nstages;
  int radix[ORC_MAX_STAGES];
  double* tw;   /* tw[2t], tw[2t+1] = cos(2 pi t/n), -sin(2 pi t/n) */
  double* work; /* 2 * 2n doubles ping-pong, then 4n doubles of butterfly scratch (allocated once per plan) */
  /* Rader's algorithm for a large prime n whose n-1 is smooth (DESIGN.md "DFT spec"): the n-point transform as a cyclic
   * convolution of length n-1 done with the (n-1)-point plan.  NULL otherwise. */
  struct orc_plan_s* sub; /* (n-1)-point plan */
  int* perm;              /* perm[p]  = g^p  mod n, p < n-1, g the smallest primitive root */
  int* iperm;             /* iperm[q] = g^-q mod n */
  double* bfft;           /* forward transform of b[q] = tw[iperm[q]], 2(n-1) doubles */
  double* rwork;          /* 3 * 2(n-1) doubles */
} orc_plan;

static int orc_factor(int n, int* radix) {
  int ns = 0;
  while (n % 4 == 0) { radix[ns++] = 4; n /= 4; }
  while (n % 2 == 0) { radix[ns++] = 2; n /= 2; }
  while (n % 3 == 0) { radix[ns++] = 3; n /= 3; }
  while (n % 5 == 0) { radix[ns++] = 5; n /= 5; }
  for (int p = 7; n > 1; p += 2) {
    while (n % p == 0) { radix[ns++] = p; n /= p; }
  }
  return ns;
}

static void orc_plan_execute(orc_plan* P, const double* in, double* out);
static orc_plan* orc_plan_create(int n);
static void orc_plan_destroy(orc_plan* p);

/* Rader applies to a prime n > 64 whose n - 1 has no prime factor above 13 (44.1 kHz: W = 661, 660 = 4*3*5*11). */
static int orc_use_rader(int n) {
  if (n <= 64) return 0;
  for (int d = 2; (long)d * d <= n; d++) if (n % d == 0) return 0;
  int m = n - 1;
  for (int d = 2; d <= 13; d++) while (m % d == 0) m /= d;
  return m == 1;
}
static int orc_primitive_root(int n) { /* smallest g whose powers visit all of 1 .. n-1 */
  for (int g = 2; g < n; g++) {
    long v = 1;
    int k = 0;
    do { v = (v * g) % n; k++; } while (v != 1);
    if (k == n - 1) return g;
  }
  return 0;
}

/* A twiddle factor (cos, -sin)(2 pi k / n): orc_twiddle.h -- IEEE-754 double operations only, no libm call, the same bits on every
 * machine (DFT spec tables v3, round 6).  Rounds 1-5 called the box's libm (round 5: "one glibc sincos call"); orc_set_twiddle_spec(2)
 * gives those tables back for A/B (a plan reads the setting when it is created). */
#include <stdio.h>
#include "orc_twiddle.h"
#include "orc_twiddle_hashes.h"
static int orc_twiddle_spec_v = 3;
void orc_set_twiddle_spec(int v) { orc_twiddle_spec_v = v == 2 ? 2 : 3; }
int orc_get_twiddle_spec(void) { return orc_twiddle_spec_v; }
void orc_twiddle_entry(long k, long n, double* c, double* s) { orc_sincos_2pi(k, n, c, s); }
unsigned long long orc_twiddle_hash(long den, long count) { return (unsigned long long)orc_twiddle_table_hash(den, count); }
static inline void orc_twiddle(long k, long n, double* c, double* ms) {
  double sn, cs;
  if (orc_twiddle_spec_v == 2) sincos(2.0 * M_PI * (double)k / (double)n, &sn, &cs);
  else orc_sincos_2pi(k, n, &cs, &sn);
  *c = cs;
  *ms = 0.0 - sn;
}
/* the tables of the sizes the library compiles in are pinned (orc_twiddle_hashes.h, generated from a 60-digit evaluation):
 * a build whose arithmetic strays (fast-math, a contracted multiply-add) is refused here instead of producing other audio */
static void orc_check_twiddles(long den, long count, const double* tw) {
  if (orc_twiddle_spec_v != 3) return;
  for (unsigned i = 0; i < sizeof(orc_twiddle_pins) / sizeof(orc_twiddle_pins[0]); i++) {
    if (orc_twiddle_pins[i].den != den || orc_twiddle_pins[i].count != count) continue;
    uint64_t h = 0xcbf29ce484222325ull;
    const unsigned char* b = (const unsigned char*)tw;
    for (long j = 0; j < 16 * count; j++) { h ^= b[j]; h *= 0x100000001b3ull; }
    if (h != orc_twiddle_pins[i].hash) {
      fprintf(stderr, "oracle: twiddle table (2 pi t / %ld, %ld entries) does not hash to its pinned value -- built with fast-math or fp contraction?\n", den, count);
      abort();
    }
  }
}
static orc_plan* orc_plan_create(int n) {
  orc_plan* p = (orc_plan*)calloc(1, sizeof(orc_plan));
  p->n = n;
  p->nstages = (n > 1) ? orc_factor(n, p->radix) : 0;
  p->tw = (double*)malloc(sizeof(double) * 2 * n);
  p->work = (double*)malloc(sizeof(double) * 8 * n);
  for (int t = 0; t < n; t++) {
    orc_twiddle(t, n, &p->tw[2 * t], &p->tw[2 * t + 1]);
  }
  orc_check_twiddles(n, n, p->tw);
  if (orc_use_rader(n)) {
    int m = n - 1;
    p->sub = orc_plan_create(m);
    p->perm = (int*)malloc(sizeof(int) * m);
    p->iperm = (int*)malloc(sizeof(int) * m);
    p->bfft = (double*)malloc(sizeof(double) * 2 * m);
    p->rwork = (double*)malloc(sizeof(double) * 6 * m);
    int g = orc_primitive_root(n);
    long v = 1;
    for (int k = 0; k < m; k++) { p->perm[k] = (int)v; v = (v * g) % n; }
    for (int q = 0; q < m; q++) p->iperm[q] = p->perm[(m - q) % m]; /* g^-q = g^(m-q) */
    double* b = p->rwork;
    for (int q = 0; q < m; q++) { b[2 * q] = p->tw[2 * p->iperm[q]]; b[2 * q + 1] = p->tw[2 * p->iperm[q] + 1]; }
    orc_plan_execute(p->sub, b, p->bfft);
  }
  return p;
}
static void orc_plan_destroy(orc_plan* p) {
  if (!p) return;
  if (p->sub) { orc_plan_destroy(p->sub); free(p->perm); free(p->iperm); free(p->bfft); free(p->rwork); }
  free(p->tw);
  free(p->work);
  free(p);
}

#define C5_1 0.30901699437494742
#define C5_2 (-0.80901699437494742)
#define S5_1 0.95105651629515357
#define S5_2 0.58778525229247313
#define S3_1 0.86602540378443865

/* DFT spec v2 (round 5): the same transform with the multiply-add pairs of the twiddle products, of the radix-3 / 5 /
 * prime butterflies and of the untangle FUSED (fma = one rounding) -- what FFTW's own codelets do on a machine that has the
 * instruction, and a quarter fewer operations for the kernel.  orc_set_dft_spec(1): the unfused sequence of rounds 1-4. */
static int orc_dft_spec_v = 2;
void orc_set_dft_spec(int v) { orc_dft_spec_v = v == 1 ? 1 : 2; }
int orc_get_dft_spec(void) { return orc_dft_spec_v; }
/* y = b * w (a twiddle or pointwise product), v2: re = fma(br, wr, -(bi wi)), im = fma(br, wi, bi wr) */
static inline void orc_cmul(double br, double bi, double wr, double wi, double* yr, double* yi) {
  if (orc_dft_spec_v == 2) {
    *yr = fma(br, wr, -(bi * wi));
    *yi = fma(br, wi, bi * wr);
  } else {
    *yr = br * wr - bi * wi;
    *yi = br * wi + bi * wr;
  }
}

/* One radix-r butterfly: a[i] (re,im) for i<r -> b[j].  Fixed operation order. */
static void orc_butterfly_v2(const orc_plan* P, int r, const double* ar, const double* ai, double* br, double* bi);
static void orc_butterfly(const orc_plan* P, int r, const double* ar, const double* ai, double* br,
                          double* bi) {
  if (orc_dft_spec_v == 2 && r != 2 && r != 4) { orc_butterfly_v2(P, r, ar, ai, br, bi); return; }
  if (r == 2) {
    br[0] = ar[0] + ar[1]; bi[0] = ai[0] + ai[1];
    br[1] = ar[0] - ar[1]; bi[1] = ai[0] - ai[1];
  } else if (r == 4) {
    double t0r = ar[0] + ar[2], t0i = ai[0] + ai[2];
    double t1r = ar[0] - ar[2], t1i = ai[0] - ai[2];
    double t2r = ar[1] + ar[3], t2i = ai[1] + ai[3];
    double t3r = ar[1] - ar[3], t3i = ai[1] - ai[3];
    br[0] = t0r + t2r; bi[0] = t0i + t2i;
    br[2] = t0r - t2r; bi[2] = t0i - t2i;
    br[1] = t1r + t3i; bi[1] = t1i - t3r; /* t1 - i t3 */
    br[3] = t1r - t3i; bi[3] = t1i + t3r; /* t1 + i t3 */
  } else if (r == 3) {
    double t1r = ar[1] + ar[2], t1i = ai[1] + ai[2];
    double t2r = ar[0] - 0.5 * t1r, t2i = ai[0] - 0.5 * t1i;
    double t3r = S3_1 * (ar[1] - ar[2]), t3i = S3_1 * (ai[1] - ai[2]);
    br[0] = ar[0] + t1r; bi[0] = ai[0] + t1i;
    br[1] = t2r + t3i; bi[1] = t2i - t3r;
    br[2] = t2r - t3i; bi[2] = t2i + t3r;
  } else if (r == 5) {
    double t1r = ar[1] + ar[4], t1i = ai[1] + ai[4];
    double t2r = ar[2] + ar[3], t2i = ai[2] + ai[3];
    double t3r = ar[1] - ar[4], t3i = ai[1] - ai[4];
    double t4r = ar[2] - ar[3], t4i = ai[2] - ai[3];
    br[0] = (ar[0] + t1r) + t2r; bi[0] = (ai[0] + t1i) + t2i;
    double m1r = (ar[0] + C5_1 * t1r) + C5_2 * t2r, m1i = (ai[0] + C5_1 * t1i) + C5_2 * t2i;
    double m2r = (ar[0] + C5_2 * t1r) + C5_1 * t2r, m2i = (ai[0] + C5_2 * t1i) + C5_1 * t2i;
    double n1r = S5_1 * t3r + S5_2 * t4r, n1i = S5_1 * t3i + S5_2 * t4i;
    double n2r = S5_2 * t3r - S5_1 * t4r, n2i = S5_2 * t3i - S5_1 * t4i;
    br[1] = m1r + n1i; bi[1] = m1i - n1r; /* m1 - i n1 */
    br[4] = m1r - n1i; bi[4] = m1i + n1r;
    br[2] = m2r + n2i; bi[2] = m2i - n2r;
    br[3] = m2r - n2i; bi[3] = m2i + n2r;
  } else {
    /* odd prime radix r = 2h+1 (DESIGN.md "DFT spec"), conjugate-symmetric pairs: with u_i = a_i + a_{r-i},
     * v_i = a_i - a_{r-i} (i = 1..h) and w^k = c_k + i s_k the table entry of k = (i j) mod r,
     *   b_0     = ((a_0 + u_1) + u_2) + ... + u_h
     *   P_j     = ((a_0 + c u_1) + c u_2) + ...          (c = c_{(i j) mod r}, i ascending)
     *   Q_j     = (s v_1 + s v_2) + ...                  (s = s_{(i j) mod r})
     *   b_j     = P_j + i Q_j,   b_{r-j} = P_j - i Q_j    (j = 1..h)
     * -- a quarter of the multiplications of the plain sum over all r-1 inputs. */
    int step = P->n / r;
    int h = (r - 1) / 2;
    double ur[h + 1], ui[h + 1], vr[h + 1], vi[h + 1];
    double b0r = ar[0], b0i = ai[0];
    for (int i = 1; i <= h; i++) {
      ur[i] = ar[i] + ar[r - i]; ui[i] = ai[i] + ai[r - i];
      vr[i] = ar[i] - ar[r - i]; vi[i] = ai[i] - ai[r - i];
      b0r = b0r + ur[i]; b0i = b0i + ui[i];
    }
    br[0] = b0r; bi[0] = b0i;
    for (int j = 1; j <= h; j++) {
      double pr = ar[0], pi = ai[0], qr = 0.0, qi = 0.0;
      for (int i = 1; i <= h; i++) {
        int t = ((i * j) % r) * step;
        double c = P->tw[2 * t], sn = P->tw[2 * t + 1];
        pr = pr + c * ur[i]; pi = pi + c * ui[i];
        if (i == 1) { qr = sn * vr[i]; qi = sn * vi[i]; }
        else { qr = qr + sn * vr[i]; qi = qi + sn * vi[i]; }
      }
      br[j] = pr - qi; bi[j] = pi + qr;
      br[r - j] = pr + qi; bi[r - j] = pi - qr;
    }
  }
}

/* v2 of the butterflies that multiply (radix 2 and 4 only add):
 *   radix 3:  t2 = fma(-1/2, t1, a0)
 *   radix 5:  m1 = fma(C2, t2, fma(C1, t1, a0)), m2 = fma(C1, t2, fma(C2, t1, a0)), n1 = fma(S1, t3, S2 t4), n2 = fma(S2, t3, -(S1 t4))
 *   prime r:  P_j = fma(c, u_i, P_j), Q_j = s v_1 then fma(s, v_i, Q_j)                                                     */
static void orc_butterfly_v2(const orc_plan* P, int r, const double* ar, const double* ai, double* br, double* bi) {
  if (r == 3) {
    double t1r = ar[1] + ar[2], t1i = ai[1] + ai[2];
    double t2r = fma(-0.5, t1r, ar[0]), t2i = fma(-0.5, t1i, ai[0]);
    double t3r = S3_1 * (ar[1] - ar[2]), t3i = S3_1 * (ai[1] - ai[2]);
    br[0] = ar[0] + t1r; bi[0] = ai[0] + t1i;
    br[1] = t2r + t3i; bi[1] = t2i - t3r;
    br[2] = t2r - t3i; bi[2] = t2i + t3r;
  } else if (r == 5) {
    double t1r = ar[1] + ar[4], t1i = ai[1] + ai[4];
    double t2r = ar[2] + ar[3], t2i = ai[2] + ai[3];
    double t3r = ar[1] - ar[4], t3i = ai[1] - ai[4];
    double t4r = ar[2] - ar[3], t4i = ai[2] - ai[3];
    br[0] = (ar[0] + t1r) + t2r; bi[0] = (ai[0] + t1i) + t2i;
    double m1r = fma(C5_2, t2r, fma(C5_1, t1r, ar[0])), m1i = fma(C5_2, t2i, fma(C5_1, t1i, ai[0]));
    double m2r = fma(C5_1, t2r, fma(C5_2, t1r, ar[0])), m2i = fma(C5_1, t2i, fma(C5_2, t1i, ai[0]));
    double n1r = fma(S5_1, t3r, S5_2 * t4r), n1i = fma(S5_1, t3i, S5_2 * t4i);
    double n2r = fma(S5_2, t3r, -(S5_1 * t4r)), n2i = fma(S5_2, t3i, -(S5_1 * t4i));
    br[1] = m1r + n1i; bi[1] = m1i - n1r;
    br[4] = m1r - n1i; bi[4] = m1i + n1r;
    br[2] = m2r + n2i; bi[2] = m2i - n2r;
    br[3] = m2r - n2i; bi[3] = m2i + n2r;
  } else {
    int step = P->n / r;
    int h = (r - 1) / 2;
    double ur[h + 1], ui[h + 1], vr[h + 1], vi[h + 1];
    double b0r = ar[0], b0i = ai[0];
    for (int i = 1; i <= h; i++) {
      ur[i] = ar[i] + ar[r - i]; ui[i] = ai[i] + ai[r - i];
      vr[i] = ar[i] - ar[r - i]; vi[i] = ai[i] - ai[r - i];
      b0r = b0r + ur[i]; b0i = b0i + ui[i];
    }
    br[0] = b0r; bi[0] = b0i;
    for (int j = 1; j <= h; j++) {
      double pr = ar[0], pi = ai[0], qr = 0.0, qi = 0.0;
      for (int i = 1; i <= h; i++) {
        int t = ((i * j) % r) * step;
        double c = P->tw[2 * t], sn = P->tw[2 * t + 1];
        pr = fma(c, ur[i], pr); pi = fma(c, ui[i], pi);
        if (i == 1) { qr = sn * vr[i]; qi = sn * vi[i]; }
        else { qr = fma(sn, vr[i], qr); qi = fma(sn, vi[i], qi); }
      }
      br[j] = pr - qi; bi[j] = pi + qr;
      br[r - j] = pr + qi; bi[r - j] = pi - qr;
    }
  }
}

/* X[0] = x[0] + A[0];  X[g^-q] = x[0] + c[q],  c = a (*) b cyclically, a[p] = x[g^p], b[q] = w^(g^-q):
 * A = F(a), C = A .* F(b), c = conj(F(conj(C))) / (n-1), all with the (n-1)-point plan.  Fixed operation order. */
static void orc_plan_execute_rader(orc_plan* P, const double* in, double* out) {
  int n = P->n, m = n - 1;
  double* a = P->rwork;
  double* A = a + 2 * m;
  double* c = A + 2 * m;
  for (int p = 0; p < m; p++) { a[2 * p] = in[2 * P->perm[p]]; a[2 * p + 1] = in[2 * P->perm[p] + 1]; }
  orc_plan_execute(P->sub, a, A);
  double x0r = in[0], x0i = in[1];
  double A0r = A[0], A0i = A[1];
  for (int k = 0; k < m; k++) {
    double br = P->bfft[2 * k], bi = P->bfft[2 * k + 1];
    double cr, ci;
    orc_cmul(A[2 * k], A[2 * k + 1], br, bi, &cr, &ci);
    a[2 * k] = cr; a[2 * k + 1] = -ci; /* conj(C) */
  }
  orc_plan_execute(P->sub, a, c);
  double inv = 1.0 / (double)m;
  out[0] = x0r + A0r; out[1] = x0i + A0i;
  for (int q = 0; q < m; q++) {
    double cr = c[2 * q] * inv, ci = -c[2 * q + 1] * inv;
    out[2 * P->iperm[q]] = x0r + cr;
    out[2 * P->iperm[q] + 1] = x0i + ci;
  }
}

static void orc_plan_execute(orc_plan* P, const double* in, double* out) {
  int n = P->n;
  if (n == 1) { out[0] = in[0]; out[1] = in[1]; return; }
  if (P->sub) { orc_plan_execute_rader(P, in, out); return; }
  double* x = P->work;
  double* y = P->work + 2 * n;
  memcpy(x, in, sizeof(double) * 2 * n);
  int s = 1;   /* product of the radices of the stages already done */
  int cur = n; /* length of the sub-transforms still to do */
  double* ar = P->work + 4 * n;
  double* ai = ar + n; double* br = ai + n; double* bi = br + n;
  for (int st = 0; st < P->nstages; st++) {
    int r = P->radix[st];
    int m = cur / r;
    for (int p = 0; p < m; p++) {
      for (int q = 0; q < s; q++) {
        for (int i = 0; i < r; i++) {
          int idx = q + s * (p + i * m);
          ar[i] = x[2 * idx]; ai[i] = x[2 * idx + 1];
        }
        orc_butterfly(P, r, ar, ai, br, bi);
        for (int j = 0; j < r; j++) {
          int t = (int)(((long)s * p * j) % n);
          double wr = P->tw[2 * t], wi = P->tw[2 * t + 1];
          int o = q + s * (r * p + j);
          orc_cmul(br[j], bi[j], wr, wi, &y[2 * o], &y[2 * o + 1]);
        }
      }
    }
    double* tmp = x; x = y; y = tmp;
    s *= r;
    cur = m;
  }
  memcpy(out, x, sizeof(double) * 2 * n);
}

void orc_dft_forward(int n, const double* in, double* out) {
  orc_plan* P = orc_plan_create(n);
  orc_plan_execute(P, in, out);
  orc_plan_destroy(P);
}

void orc_dft_naive(int n, const double* in, double* out) {
  for (int k = 0; k < n; k++) {
    long double sr = 0, si = 0;
    for (int t = 0; t < n; t++) {
      long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)(((long)k * t) % n) / n;
      long double c = cosl(ang), s = sinl(ang);
      sr += in[2 * t] * c - in[2 * t + 1] * s;
      si += in[2 * t] * s + in[2 * t + 1] * c;
    }
    out[2 * k] = (double)sr; out[2 * k + 1] = (double)si;
  }
}

/* Spectrum of a real, zero-padded frame: x[W] floats -> |X[k]|, k < 2W (DESIGN.md "DFT spec"). */
typedef struct {
  int W;
  orc_plan* plan;  /* W-point */
  double* tw2;     /* e^{-2 pi i k / (2W)}, k < W */
  double* z;       /* 2W doubles */
  double* Z;       /* 2W doubles */
  void* fftw;      /* FFTW backend (orc_set_fft_backend(1), when the box has libfftw3): an N-point complex plan ... */
  double* fin;     /* ... its input, 2N doubles */
  double* fout;    /* ... and its output */
} orc_specplan;

/* ---- the reference's shipped FFT library, when the box has one (reference speedy.c:228-231, 458-473, Makefile:9) ----
 * libfftw3 is looked up at RUN time (dlopen): nothing links against it, nothing needs its header.  With the backend on, a frame's
 * spectrum is what the reference's FFTW build computes: the windowed frame (float products) zero-padded to N COMPLEX doubles,
 * fftw_plan_dft_1d(N, in, out, FFTW_FORWARD, FFTW_ESTIMATE), |X[i]| = cabs -> float.  It is a second CPU baseline of bench.py
 * ("cpu_baseline_fftw") and a cross-check of the port's spectra -- never the parity oracle (its last bits are FFTW's codelets'). */
#include <dlfcn.h>
#include <pthread.h>
static struct {
  int tried, ok;
  void* (*plan_dft_1d)(int, void*, void*, int, unsigned);
  void (*execute)(void*);
  void (*destroy_plan)(void*);
} orc_fftw;
static pthread_mutex_t orc_fftw_mu = PTHREAD_MUTEX_INITIALIZER;   /* FFTW's planner is not thread-safe (reference: one global planner) */
static int orc_fft_backend_v = 0;
int orc_fftw_available(void) {
  pthread_mutex_lock(&orc_fftw_mu);
  if (!orc_fftw.tried) {
    orc_fftw.tried = 1;
    void* h = dlopen("libfftw3.so.3", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libfftw3.so", RTLD_NOW | RTLD_LOCAL);
    if (h) {
      *(void**)&orc_fftw.plan_dft_1d = dlsym(h, "fftw_plan_dft_1d");
      *(void**)&orc_fftw.execute = dlsym(h, "fftw_execute");
      *(void**)&orc_fftw.destroy_plan = dlsym(h, "fftw_destroy_plan");
      orc_fftw.ok = orc_fftw.plan_dft_1d && orc_fftw.execute && orc_fftw.destroy_plan;
    }
  }
  const int ok = orc_fftw.ok;
  pthread_mutex_unlock(&orc_fftw_mu);
  return ok;
}
/* 0 (default): the port's own transform (the DFT spec: what the GPU equals bit for bit); 1: FFTW, if there.  Returns the backend in
 * force.  Read when a stream is created. */
int orc_set_fft_backend(int v) { orc_fft_backend_v = (v == 1 && orc_fftw_available()) ? 1 : 0; return orc_fft_backend_v; }

static orc_specplan* orc_specplan_create(int W) {
  orc_specplan* sp = (orc_specplan*)calloc(1, sizeof(orc_specplan));
  sp->W = W;
  sp->plan = orc_plan_create(W);
  sp->tw2 = (double*)malloc(sizeof(double) * 2 * W);
  sp->z = (double*)malloc(sizeof(double) * 2 * W);
  sp->Z = (double*)malloc(sizeof(double) * 2 * W);
  for (int k = 0; k < W; k++) {
    orc_twiddle(k, 2L * W, &sp->tw2[2 * k], &sp->tw2[2 * k + 1]);
  }
  orc_check_twiddles(2L * W, W, sp->tw2);
  if (orc_fft_backend_v == 1 && orc_fftw_available()) {
    const int N = 2 * W;
    sp->fin = (double*)calloc((size_t)2 * N, sizeof(double));   /* (the reference uses fftw_malloc: alignment only) */
    sp->fout = (double*)calloc((size_t)2 * N, sizeof(double));
    pthread_mutex_lock(&orc_fftw_mu);
    sp->fftw = orc_fftw.plan_dft_1d(N, sp->fin, sp->fout, -1 /* FFTW_FORWARD */, 1u << 6 /* FFTW_ESTIMATE */);   /* speedy.c:228-231 */
    pthread_mutex_unlock(&orc_fftw_mu);
  }
  return sp;
}
static void orc_specplan_destroy(orc_specplan* sp) {
  if (!sp) return;
  orc_plan_destroy(sp->plan);
  if (sp->fftw) { pthread_mutex_lock(&orc_fftw_mu); orc_fftw.destroy_plan(sp->fftw); pthread_mutex_unlock(&orc_fftw_mu); }
  free(sp->fin); free(sp->fout);
  free(sp->tw2); free(sp->z); free(sp->Z); free(sp);
}
static void orc_specplan_run(orc_specplan* sp, const float* x, float* mags) {
  int W = sp->W, N = 2 * W;
  if (sp->fftw) { /* speedy.c:458-473 */
    for (int i = 0; i < W; i++) { sp->fin[2 * i] = (double)x[i]; sp->fin[2 * i + 1] = 0.0; }
    for (int i = W; i < N; i++) { sp->fin[2 * i] = 0.0; sp->fin[2 * i + 1] = 0.0; }
    orc_fftw.execute(sp->fftw);
    for (int i = 0; i < N; i++) mags[i] = (float)hypot(sp->fout[2 * i], sp->fout[2 * i + 1]);   /* cabs */
    return;
  }
  for (int n = 0; n < W; n++) {
    sp->z[2 * n] = (2 * n < W) ? (double)x[2 * n] : 0.0;
    sp->z[2 * n + 1] = (2 * n + 1 < W) ? (double)x[2 * n + 1] : 0.0;
  }
  orc_plan_execute(sp->plan, sp->z, sp->Z);
  const double* Z = sp->Z;
  for (int k = 0; k < W; k++) {
    int k2 = (W - k) % W;
    double a_r = Z[2 * k], a_i = Z[2 * k + 1];
    double b_r = Z[2 * k2], b_i = -Z[2 * k2 + 1];
    double dr = a_r - b_r, di = a_i - b_i;
    double c = sp->tw2[2 * k], s = sp->tw2[2 * k + 1];
    if (orc_dft_spec_v == 2) {
      /* v2: 2 X[k] = (Z[k] + conj Z[W-k]) - i e^{-2 pi i k/N} (Z[k] - conj Z[W-k]) with the products fused, the halving after the root
       * (exact):  2 xr = fma(c, di, fma(s, dr, ar + br)),  2 xi = fma(s, di, fma(-c, dr, ai + bi)),  |X| = (float)sqrt(fma(2xr, 2xr, 2xi 2xi) / 4) */
      double xr2 = fma(c, di, fma(s, dr, a_r + b_r));
      double xi2 = fma(s, di, fma(-c, dr, a_i + b_i));
      mags[k] = (float)sqrt(0.25 * fma(xr2, xr2, xi2 * xi2));
    } else {
      double er = 0.5 * (a_r + b_r), ei = 0.5 * (a_i + b_i);
      double o_r = 0.5 * di, o_i = -0.5 * dr;
      double xr = er + (c * o_r - s * o_i);
      double xi = ei + (c * o_i + s * o_r);
      mags[k] = (float)sqrt(xr * xr + xi * xi);
    }
  }
  mags[W] = (float)fabs(Z[0] - Z[1]);
  for (int k = W + 1; k < N; k++) mags[k] = mags[N - k];
}
void orc_spectrum_magnitudes(int W, const float* x, float* mags) {
  orc_specplan* sp = orc_specplan_create(W);
  orc_specplan_run(sp, x, mags);
  orc_specplan_destroy(sp);
}

/* ------------------------------------------------------------------------------------------ */
/* speedyStreamStruct, speedy.c:130-176                                                        */
struct orc_speedy {
  int sample_rate, window_size, fft_size;
  int future, past;        /* speedy.h:136-146 made run-time */
  int hyst_size;           /* speedy.c:95: 2*(F+P+1) */
  int spec_hist;           /* speedy.c:97: F+P+1 */
  float* window;
  float* input;
  int64_t current_time;
  float* spectrogram;
  float** spectrogram_history;
  float* normalized_spectrogram;
  float* normalized_last_spectrogram;
  orc_specplan* plan;
  float* windowed;
  float* hysteresis_buffer;
  float preemph_state;
  float mean_spectrogram_energy, mean_emphasis_weighted_local_difference;
  float mean_emphasis_weighted_lpf, mean_relative_spectral_difference, max_energy_hysteresis;
  int skip_frame_count;
  orc_fof energy_filter, difference_filter;
  float current_duration, desired_duration;
  float features[ORC_FEATURE_COUNT];
};
/* feature slots, speedy.c:106-123 */
#define s_energy_lp (s->features[1])
#define s_energy_local (s->features[2])
#define s_energy_compressed (s->features[3])
#define s_time_energy (s->features[12])
#define s_energy_hysteresis (s->features[4])
#define s_spectrogram_energy (s->features[0])
#define s_low_energy_threshold (s->features[14])
#define s_low_energy_frame (s->features[5])
#define s_local_spectral_difference (s->features[6])
#define s_emphasis_weighted_local_difference (s->features[7])
#define s_emphasis_weighted_lpf (s->features[8])
#define s_relative_spectral_difference (s->features[9])
#define s_speech_changes (s->features[10])
#define s_time_spectral (s->features[13])
#define s_audio_tension (s->features[11])

static int orc_modulo(int64_t x, int N) { return (int)((x % N + N) % N); } /* speedy.c:198-200 */

orc_speedyStream orc_speedyCreateStream(int sample_rate, int match_matlab) { /* speedy.c:206-299 */
  orc_speedyStream s = (orc_speedyStream)calloc(1, sizeof(struct orc_speedy));
  if (!s) return NULL;
  s->future = match_matlab ? 8 : 12;
  s->past = match_matlab ? 12 : 8;
  s->hyst_size = 2 * (s->future + s->past + 1);
  s->spec_hist = s->future + s->past + 1;
  s->window_size = (int)(1.5 * sample_rate / (float)kFrameRateHz); /* speedy.c:213 */
  s->fft_size = 2 * s->window_size;
  s->sample_rate = sample_rate;
  s->input = (float*)malloc(sizeof(float) * s->window_size);
  s->windowed = (float*)malloc(sizeof(float) * s->window_size);
  s->hysteresis_buffer = (float*)calloc(s->hyst_size, sizeof(float));
  s->normalized_spectrogram = (float*)calloc(s->fft_size, sizeof(float));
  s->normalized_last_spectrogram = (float*)calloc(s->fft_size, sizeof(float));
  s->spectrogram = (float*)calloc(s->fft_size, sizeof(float));
  s->window = (float*)malloc(sizeof(float) * s->window_size);
  s->spectrogram_history = (float**)calloc(s->spec_hist, sizeof(float*));
  for (int i = 0; i < s->spec_hist; i++)
    s->spectrogram_history[i] = (float*)calloc(s->fft_size, sizeof(float));
  for (int i = 0; i < s->window_size; i++) { /* speedy.c:256-258: double expression -> float */
    /* the cosine from orc_twiddle.h (no libm call: the same window on every machine); spec 2: libm's, as rounds 1-5 */
    double c, sn;
    if (orc_twiddle_spec_v == 2 || s->window_size < 2) c = cos(2 * M_PI * i / (s->window_size - 1.0));
    else orc_sincos_2pi(i, s->window_size - 1, &c, &sn);
    s->window[i] = 0.54 - 0.46 * c;
  }
  s->mean_spectrogram_energy = 2.14204; /* speedy.c:263-267 */
  s->mean_emphasis_weighted_local_difference = 123.837;
  s->mean_emphasis_weighted_lpf = 123.979;
  s->mean_relative_spectral_difference = 0.971975;
  s->max_energy_hysteresis = 1.41421;
  s->plan = orc_specplan_create(s->window_size);
  orc_fof_design(&s->energy_filter, kFrameRateHz); /* speedy.c:287-292 */
  orc_fof_set_state(&s->energy_filter, s->mean_spectrogram_energy);
  orc_fof_design(&s->difference_filter, kFrameRateHz);
  orc_fof_set_state(&s->difference_filter, s->mean_emphasis_weighted_local_difference);
  s->skip_frame_count = 1; /* speedy.c:293 */
  return s;
}

void orc_speedyDestroyStream(orc_speedyStream s) {
  if (!s) return;
  free(s->input); free(s->windowed); free(s->hysteresis_buffer);
  free(s->normalized_spectrogram); free(s->normalized_last_spectrogram);
  free(s->spectrogram); free(s->window);
  for (int i = 0; i < s->spec_hist; i++) free(s->spectrogram_history[i]);
  free(s->spectrogram_history);
  orc_specplan_destroy(s->plan);
  free(s);
}

int orc_speedyInputFrameSize(orc_speedyStream s) { return s->window_size; }
int orc_speedyInputFrameStep(orc_speedyStream s) { return s->sample_rate / kFrameRateHz; } /* :335-338 */
int orc_speedyFFTSize(orc_speedyStream s) { return s->fft_size; }
int orc_speedyHysteresisFuture(orc_speedyStream s) { return s->future; }
int orc_speedyHysteresisPast(orc_speedyStream s) { return s->past; }
float orc_speedyBinToFreq(orc_speedyStream s, int bin) { /* speedy.c:345-348 */
  return bin * (s->sample_rate / (float)s->fft_size);
}
int orc_speedyFreqToBin(orc_speedyStream s, float freq) { /* speedy.c:350-353 */
  return round(freq * s->fft_size / s->sample_rate);
}
float* orc_speedyGetSpectrogram(orc_speedyStream s) { return s->spectrogram; }
float* orc_speedyGetNormalizedSpectrogram(orc_speedyStream s) { return s->normalized_spectrogram; }
float* orc_speedyGetInternalState(orc_speedyStream s) { return s->features; }
float orc_speedyGetEnergyCompressed(orc_speedyStream s) { return s_energy_compressed; }
float orc_speedyGetSpeechChanges(orc_speedyStream s) { return s_speech_changes; }
int64_t orc_speedyGetCurrentTime(orc_speedyStream s) { return s->current_time; }

void orc_speedyPreemphasisFilter(orc_speedyStream s, float* input, int length) { /* speedy.c:416-425 */
  for (int i = 0; i < length; i++) {
    float last_sample = input[i];
    input[i] = 1.0 * input[i] - 0.97 * s->preemph_state; /* double expression, float store */
    s->preemph_state = last_sample;
  }
}

float* orc_speedySpectrogram(orc_speedyStream s, float* input) { /* speedy.c:438-473 */
  for (int i = 0; i < s->window_size; i++) s->windowed[i] = input[i] * s->window[i]; /* float product */
  orc_specplan_run(s->plan, s->windowed, s->spectrogram);
  return s->spectrogram;
}

void orc_speedySaveSpectrogramData(orc_speedyStream s, float* spectrogram, int64_t at_time) { /* :476-483 */
  memcpy(s->spectrogram_history[orc_modulo(at_time, s->spec_hist)], spectrogram,
         sizeof(float) * s->fft_size);
}
float* orc_speedyGetSpectrogramAtTime(orc_speedyStream s, int64_t at_time) { /* :485-487 */
  return s->spectrogram_history[orc_modulo(at_time, s->spec_hist)];
}

void orc_speedyAddToHysteresisBuffer(orc_speedyStream s, float value, int64_t at_time) { /* :615-619 */
  s->hysteresis_buffer[orc_modulo(at_time, s->hyst_size)] = value;
}

void orc_speedyComputeLocalEnergy(orc_speedyStream s, float* spectrogram, int64_t at_time) { /* :510-523 */
  (void)spectrogram; /* the reference reads stream->spectrogram, not the argument (speedy.c:515) */
  float my_spectrogram_energy = 0.0;
  for (int i = 1; i < s->fft_size / 2; i++)
    my_spectrogram_energy += s->spectrogram[i] * s->spectrogram[i];
  s_energy_lp = orc_fof_iterate(&s->energy_filter, my_spectrogram_energy);
  s_energy_local = my_spectrogram_energy / s_energy_lp;
  s_energy_compressed = sqrt(s_energy_local > 2 ? 2.0 : s_energy_local); /* double sqrt -> float */
  orc_speedyAddToHysteresisBuffer(s, s_energy_compressed, at_time);
  s_time_energy = at_time;
}

void orc_speedyAddData(orc_speedyStream s, const float* input, int64_t at_time) { /* :540-551 */
  for (int i = 0; i < s->window_size; i++) s->input[i] = input[i];
  orc_speedyPreemphasisFilter(s, s->input, s->window_size);
  float* spectrogram = orc_speedySpectrogram(s, s->input);
  orc_speedySaveSpectrogramData(s, spectrogram, at_time);
  orc_speedyComputeLocalEnergy(s, spectrogram, at_time);
  s->current_time = at_time;
}
void orc_speedyAddDataShort(orc_speedyStream s, const int16_t* input, int64_t at_time) { /* :553-565 */
  for (int i = 0; i < s->window_size; i++) s->input[i] = input[i] / 32768.0;
  orc_speedyPreemphasisFilter(s, s->input, s->window_size);
  float* spectrogram = orc_speedySpectrogram(s, s->input);
  orc_speedySaveSpectrogramData(s, spectrogram, at_time);
  orc_speedyComputeLocalEnergy(s, spectrogram, at_time);
  s->current_time = at_time;
}

float orc_speedyEvaluateHysteresis(orc_speedyStream s, int64_t at_time) { /* speedy.c:590-610 */
  float past_max = 0.0, future_max = 0.0;
  const int F = s->future, P = s->past;
  for (int i = 0; i <= F; i++) {
    float value = s->hysteresis_buffer[orc_modulo(at_time + i, s->hyst_size)];
    value *= (F - i) / (float)F;
    if (value > future_max) future_max = value;
  }
  for (int i = 0; i <= P; i++) {
    float value = s->hysteresis_buffer[orc_modulo(at_time - i, s->hyst_size)];
    value *= (P - i) / (float)P;
    if (value > past_max) past_max = value;
  }
  return (past_max + future_max) / 2.0;
}

float orc_speedyNormalizeByEnergy(const float* spectrogram, float* normalized, int length) { /* :628-647 */
  float signal_energy = 0.0;
  float max_value = 0;
  for (int i = 1; i < length; i++) {
    signal_energy += spectrogram[i] * spectrogram[i];
    if (spectrogram[i] > max_value) max_value = spectrogram[i];
  }
  const float eps = 2.2204e-16;
  float inverse_norm = 1.0 / (sqrt(signal_energy) + eps);
  for (int i = 0; i < length; i++) normalized[i] = spectrogram[i] * inverse_norm;
  return signal_energy;
}

void orc_speedyComputeSpectralDifference(orc_speedyStream s, const float* spectrogram,
                                         const float* last_spectrogram, int64_t at_time) { /* :664-729 */
  s_energy_hysteresis = orc_speedyEvaluateHysteresis(s, at_time);
  s_spectrogram_energy = orc_speedyNormalizeByEnergy(spectrogram, s->normalized_spectrogram, s->fft_size / 2);
  orc_speedyNormalizeByEnergy(last_spectrogram, s->normalized_last_spectrogram, s->fft_size / 2);
  s_low_energy_threshold = 0.04 * s->max_energy_hysteresis;
  s_low_energy_frame = s_spectrogram_energy <= s_low_energy_threshold;
  s_time_spectral = at_time;
  if (s_low_energy_frame) s->skip_frame_count = 1;
  if (s->skip_frame_count-- > 0) {
    s_low_energy_frame = 1;
    s_local_spectral_difference = 0;
    s_emphasis_weighted_local_difference = 0;
    s_relative_spectral_difference = 0;
    s_speech_changes = 0;
    s_emphasis_weighted_lpf = orc_fof_iterate(&s->difference_filter, 0.0);
    return;
  } else {
    s->skip_frame_count = 0;
  }
  float bin_threshold = 0;
  for (int i = 1; i < s->fft_size / 2; i++) bin_threshold = fmax(bin_threshold, spectrogram[i]);
  bin_threshold /= 100.0; /* double division, float store */

  s_local_spectral_difference = 0.0;
  const float eps = 2.2204e-16;
  for (int i = 1; i < s->fft_size / 2; i++) {
    if (spectrogram[i] > bin_threshold && last_spectrogram[i] > bin_threshold) {
      /* float sum, float sum, float quotient; log and fabs in double; float accumulator */
      s_local_spectral_difference +=
          fabs(orc_log_spec((s->normalized_spectrogram[i] + eps) / (s->normalized_last_spectrogram[i] + eps)));
    }
  }
  s_emphasis_weighted_local_difference = s_local_spectral_difference * s_energy_hysteresis;
  s_emphasis_weighted_lpf = orc_fof_iterate(&s->difference_filter, s_emphasis_weighted_local_difference);
  s_relative_spectral_difference =
      s_emphasis_weighted_local_difference / (s_emphasis_weighted_lpf + 0.01 * s->mean_emphasis_weighted_lpf);
  s_speech_changes = fmin(s_relative_spectral_difference, 4 * s->mean_relative_spectral_difference);
}

int orc_speedyComputeTension(orc_speedyStream s, int64_t at_time, float* tension) { /* speedy.c:752-766 */
  float a = 1 / 2.0, b = 1 / 4.0, M_E_ = 0.7, M_S = 1.0;
  if (at_time + s->future <= s->current_time) {
    float* current_spectrogram = orc_speedyGetSpectrogramAtTime(s, at_time);
    float* previous_spectrogram = orc_speedyGetSpectrogramAtTime(s, at_time - 1);
    s_energy_hysteresis = orc_speedyEvaluateHysteresis(s, at_time);
    orc_speedyComputeSpectralDifference(s, current_spectrogram, previous_spectrogram, at_time);
    s_audio_tension = a * (s_energy_hysteresis - M_E_) + b * (s_speech_changes - M_S);
    *tension = s_audio_tension;
    return 1;
  }
  return 0;
}

float orc_speedyComputeSpeedFromTension(float tension, float R_g, float duration_feedback_strength,
                                        orc_speedyStream s) { /* speedy.c:768-788 */
  float requested_speed;
  if (R_g > 1.0) {
    requested_speed = fmax(1, R_g + (1 - R_g) * tension);
  } else {
    requested_speed = fmax(kMinimumSpeed, fmin(1, R_g - (1 - R_g) * tension));
  }
  if (duration_feedback_strength > 0) {
    float excess_duration = s->current_duration - s->desired_duration;
    requested_speed += fmax(kMinimumSpeed, duration_feedback_strength * excess_duration);
  }
  float frame_duration = 1.0 / kFrameRateHz;
  s->current_duration += frame_duration / requested_speed;
  s->desired_duration += frame_duration / R_g;
  return requested_speed;
}
