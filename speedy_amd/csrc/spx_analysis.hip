// Analysis kernel for gfx950: everything of the Speedy analysis that is local to a 10 ms frame (or to a
// frame and its predecessor).  One 256-thread workgroup owns a tile of SPX_TF consecutive frames of one
// stream plus one halo frame.
//
//   a2  mono mix            soniclib.c:262-287   (integer mean over channels, truncating)
//   a3  /32768, pre-emph    speedy.c:553-565,416-425  (carry = last sample of the previous window, which is
//                                                  sample W-B-1 of the current one)
//   a4  Hamming + |DFT_N|   speedy.c:256-258,438-473  (DESIGN.md "DFT spec": packed W-point fp64 Stockham
//                                                  transform in LDS + real-input untangle)
//   a6  frame energy        speedy.c:513-516 (== :633-640), float accumulation in index order
//   a8  normalise, 40 dB gate, sum |log ratio|   speedy.c:628-647,705-719
//
// Order-sensitive float reductions (energy, spectral difference) run one lane per frame in the
// reference's index order so that results are bit-identical to the CPU oracle; the DFT, the logs and the
// gates run one lane per bin.  Built with -ffp-contract=off.
#include <stdlib.h>

#include "spx_internal.h"
#include "spx_walk_common.h"   // wave_max_f (the DPP reduction of non-negative floats as unsigned integers)

#ifndef SPX_TF
#define SPX_TF 16  // frames per tile (plus one halo slot)
#endif

#ifdef SPX_STAMPS
__device__ unsigned long long g_spx_astamps[16];
#define ASTAMP_DECL unsigned long long as_last = __builtin_readcyclecounter(), as_acc[10] = {0,0,0,0,0,0,0,0,0,0};
#define ASTAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); as_acc[i] += t_ - as_last; as_last = t_; } while (0)
#define ASTAMP_FLUSH if (threadIdx.x == 0 && blockIdx.x == 7) for (int i_ = 0; i_ < 10; i_++) g_spx_astamps[i_] += as_acc[i_];
extern "C" void spx_debug_astamps(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spx_astamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spx_astamps), z, sizeof(z)); }
}
#else
#define ASTAMP_DECL
#define ASTAMP(i)
#define ASTAMP_FLUSH
#endif

// Two tile sizes are compiled: SPX_TF frames (default) and SPX_TF_SMALL, which the engine picks when the smaller LDS
// footprint is what lets two analysis workgroups sit beside a stream's walk and tension workgroups (concurrent mode).
#ifndef SPX_TF_SMALL
#define SPX_TF_SMALL 8
#endif
// ... and SPX_TF_TINY, with fewer than four waves doing transforms (SpxPlanDev::dft_waves), for the window sizes of sample
// rates above about 61 kHz, whose fp64 work areas and log-term rows would not fit one CU's LDS otherwise (plan-driven
// instantiation only).
#define SPX_TF_TINY 4
int spx_analysis_tile_frames() { return SPX_TF; }
int spx_analysis_small_tile_frames() { return SPX_TF_SMALL; }
int spx_analysis_tiny_tile_frames() { return SPX_TF_TINY; }

#define SPX_CB 24  // compiled-in windows: |log ratio| terms are handed from the waves that compute them to the wave that sums
                   // them in blocks of bins (two blocks in flight); row stride + 1 double (LDS banks).  SPX_CB bins per block
                   // for the 16-frame tile, twice that for the 8-frame tile: two terms per lane of waves 1..3 either way.
// row stride of the magnitudes in LDS (floats): 16-byte rows, so that the energy chain reads four bins per instruction
static __host__ __device__ constexpr int spx_mag_stride(int W) { return (W + 4) & ~3; }
static __host__ __device__ constexpr int spx_cb(int tf) { return tf <= 8 ? 2 * SPX_CB : SPX_CB; }
// The work area holds the transform buffers first and then, aliased, the |log ratio| terms with the table of log spec v2
// (spx_log.h: 128 entries of 16 bytes) behind them: terms_bytes is where the table starts.
#ifdef SPX_LOG_V1
#define SPX_LOG_LDS_BYTES 0
#else
#define SPX_LOG_LDS_BYTES 2048
#endif
static __host__ __device__ inline size_t terms_bytes(int W, int tf, bool ct) {
  const size_t b = ct ? (size_t)2 * tf * (spx_cb(tf) + 1) * sizeof(double)   // two blocks of log terms
                      : (size_t)tf * (W + 1) * sizeof(double);               // every term of the tile
  return (b + 15) & ~(size_t)15;
}
static __host__ __device__ inline size_t work_bytes(int W, int tf, bool ct = false, int dft_waves = 4) {
  if (dft_waves < 1 || dft_waves > 4) dft_waves = 4;
  const size_t a = ct ? (size_t)4 * 2 * W * sizeof(double)                   // 4 waves x W complex, stages in place
                      : (size_t)dft_waves * 2 * 2 * W * sizeof(double);      // transforming waves x ping-pong x W complex
  const size_t b = terms_bytes(W, tf, ct) + SPX_LOG_LDS_BYTES;
  return (a > b ? a : b);
}
// 16 kHz (W = 240 = 4*4*3*5) and 22.05 kHz (W = 330 = 2*3*5*11) have their own instantiations of the kernel; every
// other window size takes the plan-driven one.  Returns the compiled-in window size, or 0.
static inline int plan_ct_window(const SpxPlanDev& P) {
  static const bool generic_only = spx_tuning_env("SPX_ANALYSIS_GENERIC") != nullptr;  // A/B and tests of the plan-driven path
  if (generic_only) return 0;
  auto is = [](const int* r, int n, std::initializer_list<int> want) {
    if (n != (int)want.size()) return false;
    int k = 0;
    for (int w : want) if (r[k++] != w) return false;
    return true;
  };
  if (!P.rader && P.W == 240 && is(P.radix, P.nstages, {4, 4, 3, 5})) return 240;
  if (!P.rader && P.W == 330 && is(P.radix, P.nstages, {2, 3, 5, 11})) return 330;
  // round 4: 48 kHz (W = 720) and 44.1 kHz (W = 661, Rader over M = 660), built for the 8-frame tile only (two workgroups
  // per CU: their transform buffers are what fills the LDS)
  if (P.tile_frames == SPX_TF_SMALL && P.dft_waves == 4) {
    if (!P.rader && P.W == 720 && is(P.radix, P.nstages, {4, 4, 3, 3, 5})) return 720;
    if (P.rader && P.W == 661 && is(P.radixM, P.nstagesM, {4, 3, 5, 11})) return 661;
  }
  // 8, 12, 24 and 32 kHz: the same code as 48 kHz over their plans (built for the 16-frame tile; 8 kHz for both tiles)
  if (!P.rader && P.dft_waves == 4) {
    if (P.W == 120 && is(P.radix, P.nstages, {4, 2, 3, 5}) && (P.tile_frames == SPX_TF || P.tile_frames == SPX_TF_SMALL)) return 120;
    if (P.W == 180 && is(P.radix, P.nstages, {4, 3, 3, 5}) && P.tile_frames == SPX_TF) return 180;
    if (P.W == 360 && is(P.radix, P.nstages, {4, 2, 3, 3, 5}) && P.tile_frames == SPX_TF) return 360;
    if (P.W == 480 && is(P.radix, P.nstages, {4, 4, 2, 3, 5}) && P.tile_frames == SPX_TF) return 480;
  }
  return 0;
}
int spx_analysis_prefers_small_tile(const SpxPlanDev& P) {  // plan creation: window sizes whose compiled-in kernel exists for that tile only
  SpxPlanDev Q = P;
  Q.tile_frames = SPX_TF_SMALL;
  Q.dft_waves = 4;
  const int w = plan_ct_window(Q);
  return w == 720 || w == 661;
}
static __host__ __device__ inline size_t stage_samples(const SpxPlanDev& P, int tf) {
  return (size_t)(tf + 1) * P.B + (P.W - P.B) + 8;  // mono samples of frames j0-1 .. j0+TF-1
}
static size_t analysis_lds_bytes(const SpxPlanDev& P, bool ct) {  // for the tile size the plan copy carries (P.tile_frames)
  const int tf = P.tile_frames > 0 ? P.tile_frames : SPX_TF;
  size_t mags = (size_t)(tf + 1) * spx_mag_stride(P.W) * sizeof(float);
  size_t small = (size_t)3 * (tf + 1) * sizeof(float);
  size_t stage = (stage_samples(P, tf) * sizeof(short) + 15) & ~(size_t)15;
  // tuning only (fewer workgroups per CU); part of the size so that the co-residency rule sees it; read once per process
  static const size_t pad = [] { const char* e = spx_tuning_env("SPX_ANALYSIS_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();
  return work_bytes(P.W, tf, ct, P.dft_waves) + ((mags + 15) & ~(size_t)15) + ((small + 15) & ~(size_t)15) + stage + pad;
}
// what spx_launch_analysis (int16 input) will ask for: the engine's co-residency arithmetic uses this
size_t spx_analysis_lds_bytes(const SpxPlanDev& P) { return analysis_lds_bytes(P, plan_ct_window(P) != 0); }
int spx_analysis_ct_window(const SpxPlanDev& P) { return plan_ct_window(P); }   // which instantiation serves the plan (0 = plan-driven)

__device__ __forceinline__ void wave_sync() {
  // LDS traffic of one wave is serviced in issue order; only the compiler must not reorder across this.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#include "spx_log.h"  // spx_log: the fdlibm operation sequence of DESIGN.md "log spec"

// Untangle of the packed transform and magnitude of bin k:  X[k] = (Z[k] + conj Z[W-k]) / 2 - (i / 2) e^{-2 pi i k / N} (Z[k] - conj Z[W-k]);
// (ar, ai) = Z[k], (b_r, b_i) = conj Z[W-k], w = (cos, -sin)(2 pi k / N).
//   v2:  2 xr = fma(wx, di, fma(wy, dr, ar + br)),  2 xi = fma(wy, di, fma(-wx, dr, ai + bi)),  |X| = (float)sqrt(fma(2xr, 2xr, 2xi 2xi) / 4)
//   v1:  the halves first, the products and sums unfused (rounds 1-4)
__device__ __forceinline__ float spx_untangle_mag(double ar, double ai, double b_r, double b_i, double2 w) {
  const double dr = ar - b_r, di = ai - b_i;
#ifdef SPX_DFT_V1
  const double er = 0.5 * (ar + b_r), ei = 0.5 * (ai + b_i);
  const double o_r = 0.5 * di, o_i = -0.5 * dr;
  const double xr = er + (w.x * o_r - w.y * o_i);
  const double xi = ei + (w.x * o_i + w.y * o_r);
  return spx_sqrt64_to_f32(xr * xr + xi * xi);
#else
  const double xr2 = __builtin_fma(w.x, di, __builtin_fma(w.y, dr, ar + b_r));
  const double xi2 = __builtin_fma(w.y, di, __builtin_fma(-w.x, dr, ai + b_i));
  return spx_sqrt64_to_f32(0.25 * __builtin_fma(xr2, xr2, xi2 * xi2));
#endif
}

#define C5_1 0.30901699437494742
#define C5_2 (-0.80901699437494742)
#define S5_1 0.95105651629515357
#define S5_2 0.58778525229247313
#define S3_1 0.86602540378443865

struct cplx {
  double r, i;
};

// ---- DFT spec v2 (round 5): the multiply-add pairs of the twiddle products, of the radix-3 / 5 / prime butterflies
// and of the untangle are FUSED (fma = one rounding) -- oracle/orc_speedy.c orc_cmul / orc_butterfly_v2 / orc_specplan_run are the
// same sequences.  -DSPX_DFT_V1: the unfused transform of rounds 1-4 (A/B against orc_set_dft_spec(1)). ----
#ifdef SPX_DFT_V1
#define SPX_FMA_OR(a, b, c, unfused) (unfused)
#else
#define SPX_FMA_OR(a, b, c, unfused) __builtin_fma((a), (b), (c))
#endif
// b * w (a twiddle, or the pointwise product of Rader's convolution): re = fma(br, wx, -(bi wy)), im = fma(br, wy, bi wx)
__host__ __device__ __forceinline__ cplx spx_cmul(cplx b, double wx, double wy) {
  return {SPX_FMA_OR(b.r, wx, -(b.i * wy), b.r * wx - b.i * wy), SPX_FMA_OR(b.r, wy, b.i * wx, b.r * wy + b.i * wx)};
}
// radix 3: a0 - t1 / 2
__host__ __device__ __forceinline__ double spx_bf3h(double a0, double t1) { return SPX_FMA_OR(-0.5, t1, a0, a0 - 0.5 * t1); }
// radix 5: (a0 + c1 t1) + c2 t2;  s1 t3 + s2 t4;  s2 t3 - s1 t4
__host__ __device__ __forceinline__ double spx_bf5m(double a0, double c1, double t1, double c2, double t2) {
  return SPX_FMA_OR(c2, t2, __builtin_fma(c1, t1, a0), (a0 + c1 * t1) + c2 * t2);
}
__host__ __device__ __forceinline__ double spx_bf5n(double s1, double t3, double s2, double t4) { return SPX_FMA_OR(s1, t3, s2 * t4, s1 * t3 + s2 * t4); }
__host__ __device__ __forceinline__ double spx_bf5d(double s2, double t3, double s1, double t4) { return SPX_FMA_OR(s2, t3, -(s1 * t4), s2 * t3 - s1 * t4); }
// odd prime radix: P += c u,  Q += s v
__host__ __device__ __forceinline__ double spx_acc(double p, double w, double u) { return SPX_FMA_OR(w, u, p, p + w * u); }

__host__ __device__ __forceinline__ cplx ld(const double* buf, int idx) {
  const double2 v = *reinterpret_cast<const double2*>(buf + 2 * idx);
  return {v.x, v.y};
}
__host__ __device__ __forceinline__ void st_tw(double* buf, int idx, cplx b, const double* tw, int t) {
  const double2 w = *reinterpret_cast<const double2*>(tw + 2 * t);
  const cplx v = spx_cmul(b, w.x, w.y);
  *reinterpret_cast<double2*>(buf + 2 * idx) = make_double2(v.r, v.i);
}

// One Stockham stage of radix r over the W-point transform held in x (-> y).  s = product of earlier radices.
// b / s for b < 4096 by an exact multiply-shift (inv = ceil(2^20 / s), s | W).  The twiddle index j*s*p is
// always below W (s*p < W/r), so no reduction modulo W is needed.
// `lane` of `nl` cooperating lanes (a wavefront in the kernel; 0 of 1 on the host, where it builds the Rader tables).
__host__ __device__ void dft_stage(const int W, const double* tw, int r, int s, int cur, const double* x, double* y,
                                   int lane, const int nl) {
  const int m = cur / r;
  const int span = W / r;
  const unsigned inv_s = ((1u << 20) + (unsigned)s - 1u) / (unsigned)s;
  if (r == 4) {
    for (int b = lane; b < span; b += nl) {
      const int p = (int)(((unsigned)b * inv_s) >> 20), q = b - p * s;
      cplx a0 = ld(x, b), a1 = ld(x, b + span), a2 = ld(x, b + 2 * span), a3 = ld(x, b + 3 * span);
      cplx t0 = {a0.r + a2.r, a0.i + a2.i}, t1 = {a0.r - a2.r, a0.i - a2.i};
      cplx t2 = {a1.r + a3.r, a1.i + a3.i}, t3 = {a1.r - a3.r, a1.i - a3.i};
      cplx b0 = {t0.r + t2.r, t0.i + t2.i}, b2 = {t0.r - t2.r, t0.i - t2.i};
      cplx b1 = {t1.r + t3.i, t1.i - t3.r}, b3 = {t1.r - t3.i, t1.i + t3.r};
      const int o = q + s * 4 * p;
      const int tp = s * p;  // < W
      st_tw(y, o, b0, tw, 0);
      st_tw(y, o + s, b1, tw, tp);
      st_tw(y, o + 2 * s, b2, tw, 2 * tp);
      st_tw(y, o + 3 * s, b3, tw, 3 * tp);
    }
  } else if (r == 2) {
    for (int b = lane; b < span; b += nl) {
      const int p = (int)(((unsigned)b * inv_s) >> 20), q = b - p * s;
      cplx a0 = ld(x, b), a1 = ld(x, b + span);
      cplx b0 = {a0.r + a1.r, a0.i + a1.i}, b1 = {a0.r - a1.r, a0.i - a1.i};
      const int o = q + s * 2 * p;
      st_tw(y, o, b0, tw, 0);
      st_tw(y, o + s, b1, tw, s * p);
    }
  } else if (r == 3) {
    for (int b = lane; b < span; b += nl) {
      const int p = (int)(((unsigned)b * inv_s) >> 20), q = b - p * s;
      cplx a0 = ld(x, b), a1 = ld(x, b + span), a2 = ld(x, b + 2 * span);
      cplx t1 = {a1.r + a2.r, a1.i + a2.i};
      cplx t2 = {spx_bf3h(a0.r, t1.r), spx_bf3h(a0.i, t1.i)};
      cplx t3 = {S3_1 * (a1.r - a2.r), S3_1 * (a1.i - a2.i)};
      cplx b0 = {a0.r + t1.r, a0.i + t1.i};
      cplx b1 = {t2.r + t3.i, t2.i - t3.r}, b2 = {t2.r - t3.i, t2.i + t3.r};
      const int o = q + s * 3 * p;
      const int tp = s * p;
      st_tw(y, o, b0, tw, 0);
      st_tw(y, o + s, b1, tw, tp);
      st_tw(y, o + 2 * s, b2, tw, 2 * tp);
    }
  } else if (r == 5) {
    for (int b = lane; b < span; b += nl) {
      const int p = (int)(((unsigned)b * inv_s) >> 20), q = b - p * s;
      cplx a0 = ld(x, b), a1 = ld(x, b + span), a2 = ld(x, b + 2 * span), a3 = ld(x, b + 3 * span),
           a4 = ld(x, b + 4 * span);
      cplx t1 = {a1.r + a4.r, a1.i + a4.i}, t2 = {a2.r + a3.r, a2.i + a3.i};
      cplx t3 = {a1.r - a4.r, a1.i - a4.i}, t4 = {a2.r - a3.r, a2.i - a3.i};
      cplx b0 = {(a0.r + t1.r) + t2.r, (a0.i + t1.i) + t2.i};
      cplx m1 = {spx_bf5m(a0.r, C5_1, t1.r, C5_2, t2.r), spx_bf5m(a0.i, C5_1, t1.i, C5_2, t2.i)};
      cplx m2 = {spx_bf5m(a0.r, C5_2, t1.r, C5_1, t2.r), spx_bf5m(a0.i, C5_2, t1.i, C5_1, t2.i)};
      cplx n1 = {spx_bf5n(S5_1, t3.r, S5_2, t4.r), spx_bf5n(S5_1, t3.i, S5_2, t4.i)};
      cplx n2 = {spx_bf5d(S5_2, t3.r, S5_1, t4.r), spx_bf5d(S5_2, t3.i, S5_1, t4.i)};
      cplx b1 = {m1.r + n1.i, m1.i - n1.r}, b4 = {m1.r - n1.i, m1.i + n1.r};
      cplx b2 = {m2.r + n2.i, m2.i - n2.r}, b3 = {m2.r - n2.i, m2.i + n2.r};
      const int o = q + s * 5 * p;
      const int tp = s * p;
      st_tw(y, o, b0, tw, 0);
      st_tw(y, o + s, b1, tw, tp);
      st_tw(y, o + 2 * s, b2, tw, 2 * tp);
      st_tw(y, o + 3 * s, b3, tw, 3 * tp);
      st_tw(y, o + 4 * s, b4, tw, 4 * tp);
    }
  } else {
    // odd prime radix r = 2h+1, conjugate-symmetric pairs (DESIGN.md "DFT spec"; oracle/orc_speedy.c orc_butterfly):
    //   u_i = a_i + a_{r-i}, v_i = a_i - a_{r-i};  b_0 = ((a_0 + u_1) + u_2) + ...;
    //   P_j = ((a_0 + c u_1) + c u_2) + ..., Q_j = (s v_1 + s v_2) + ... with c + i s = w_r^{(i j) mod r};
    //   b_j = P_j + i Q_j, b_{r-j} = P_j - i Q_j.
    // One lane per (butterfly, j) with j = 1..h; the lane with j = 1 also produces b_0.  The table index (i j) mod r
    // advances by j per input (one add and one conditional subtract instead of a modulo).
    const int step = W / r;
    const int h = (r - 1) >> 1;
    for (int item = lane; item < span * h; item += nl) {
      const int b = item / h, j = item - b * h + 1;
      const int p = (int)(((unsigned)b * inv_s) >> 20), q = b - p * s;
      const cplx a0 = ld(x, b);
      cplx P = a0, Q = {0.0, 0.0}, B0 = a0;
      int t = 0;  // (i*j) mod r
      for (int i = 1; i <= h; i++) {
        t += j;
        t -= (t >= r) ? r : 0;
        const cplx ai = ld(x, b + i * span), ar = ld(x, b + (r - i) * span);
        const cplx u = {ai.r + ar.r, ai.i + ar.i}, v = {ai.r - ar.r, ai.i - ar.i};
        const double2 w = *reinterpret_cast<const double2*>(tw + 2 * (t * step));
        B0.r = B0.r + u.r; B0.i = B0.i + u.i;
        P.r = spx_acc(P.r, w.x, u.r); P.i = spx_acc(P.i, w.x, u.i);
        if (i == 1) { Q.r = w.y * v.r; Q.i = w.y * v.i; }
        else { Q.r = spx_acc(Q.r, w.y, v.r); Q.i = spx_acc(Q.i, w.y, v.i); }
      }
      const cplx bj = {P.r - Q.i, P.i + Q.r}, brj = {P.r + Q.i, P.i - Q.r};
      const int o = q + s * r * p;
      st_tw(y, o + s * j, bj, tw, s * p * j);
      st_tw(y, o + s * (r - j), brj, tw, s * p * (r - j));
      if (j == 1) st_tw(y, o, B0, tw, 0);
    }
  }
  (void)m;
}

void spx_host_dft(int n, const int* radix, int nstages, const double* tw, const double* in, double* out) {
  double* a = new double[4 * (size_t)n];
  double* x = a;
  double* y = a + 2 * (size_t)n;
  for (int i = 0; i < 2 * n; i++) x[i] = in[i];
  int s = 1, cur = n;
  for (int st = 0; st < nstages; st++) {
    dft_stage(n, tw, radix[st], s, cur, x, y, 0, 1);
    double* t = x; x = y; y = t;
    s *= radix[st];
    cur /= radix[st];
  }
  for (int i = 0; i < 2 * n; i++) out[i] = x[i];
  delete[] a;
}

__device__ __forceinline__ int mono_sample(const int16_t* __restrict__ in, int64_t a, int C) {
  if (C == 1) return in[a];
  int sum = 0;
  const int16_t* p = in + a * C;
  for (int c = 0; c < C; c++) sum += p[c];
  return sum / C;
}

// cplx helpers of the specialised transform (same operation order as dft_stage / st_tw)
__device__ __forceinline__ cplx cmul_tw(cplx b, double2 w) { return spx_cmul(b, w.x, w.y); }
__device__ __forceinline__ void st(double* buf, int idx, cplx v) {
  *reinterpret_cast<double2*>(buf + 2 * idx) = make_double2(v.r, v.i);
}

// ---------------- compiled-in transforms of further window sizes (round 4: W = 720 at 48 kHz, the M = 660 plan of
// Rader's algorithm at 44.1 kHz) ----------------
// The same butterflies and twiddle products as dft_stage, in the same order; a stage works IN PLACE on one buffer per
// wave: all the points a lane's butterflies take are loaded into registers before the stage's first store (a wave's
// LDS operations are served in issue order), indices are constants of the lane, and the last stage (every twiddle 1)
// stores its results as they are.
template <int R>
__device__ __forceinline__ void ct_bfly(const cplx (&a)[R], cplx (&o)[R]) {
  static_assert(R == 2 || R == 3 || R == 4 || R == 5, "fixed-formula radices");
  if constexpr (R == 2) {
    o[0] = {a[0].r + a[1].r, a[0].i + a[1].i};
    o[1] = {a[0].r - a[1].r, a[0].i - a[1].i};
  } else if constexpr (R == 3) {
    const cplx t1 = {a[1].r + a[2].r, a[1].i + a[2].i};
    const cplx t2 = {spx_bf3h(a[0].r, t1.r), spx_bf3h(a[0].i, t1.i)};
    const cplx t3 = {S3_1 * (a[1].r - a[2].r), S3_1 * (a[1].i - a[2].i)};
    o[0] = {a[0].r + t1.r, a[0].i + t1.i};
    o[1] = {t2.r + t3.i, t2.i - t3.r};
    o[2] = {t2.r - t3.i, t2.i + t3.r};
  } else if constexpr (R == 4) {
    const cplx t0 = {a[0].r + a[2].r, a[0].i + a[2].i}, t1 = {a[0].r - a[2].r, a[0].i - a[2].i};
    const cplx t2 = {a[1].r + a[3].r, a[1].i + a[3].i}, t3 = {a[1].r - a[3].r, a[1].i - a[3].i};
    o[0] = {t0.r + t2.r, t0.i + t2.i};
    o[2] = {t0.r - t2.r, t0.i - t2.i};
    o[1] = {t1.r + t3.i, t1.i - t3.r};
    o[3] = {t1.r - t3.i, t1.i + t3.r};
  } else {
    const cplx t1 = {a[1].r + a[4].r, a[1].i + a[4].i}, t2 = {a[2].r + a[3].r, a[2].i + a[3].i};
    const cplx t3 = {a[1].r - a[4].r, a[1].i - a[4].i}, t4 = {a[2].r - a[3].r, a[2].i - a[3].i};
    o[0] = {(a[0].r + t1.r) + t2.r, (a[0].i + t1.i) + t2.i};
    const cplx m1 = {spx_bf5m(a[0].r, C5_1, t1.r, C5_2, t2.r), spx_bf5m(a[0].i, C5_1, t1.i, C5_2, t2.i)};
    const cplx m2 = {spx_bf5m(a[0].r, C5_2, t1.r, C5_1, t2.r), spx_bf5m(a[0].i, C5_2, t1.i, C5_1, t2.i)};
    const cplx n1 = {spx_bf5n(S5_1, t3.r, S5_2, t4.r), spx_bf5n(S5_1, t3.i, S5_2, t4.i)};
    const cplx n2 = {spx_bf5d(S5_2, t3.r, S5_1, t4.r), spx_bf5d(S5_2, t3.i, S5_1, t4.i)};
    o[1] = {m1.r + n1.i, m1.i - n1.r};
    o[4] = {m1.r - n1.i, m1.i + n1.r};
    o[2] = {m2.r + n2.i, m2.i - n2.r};
    o[3] = {m2.r - n2.i, m2.i + n2.r};
  }
}

// The compiled-in plans (radices in the order of DESIGN.md "DFT spec": 4s, a 2, 3s, 5s).  240 and 330 have hand-written code
// of their own below; 661 is Rader's algorithm over the 660-point plan.
template <int W> struct ct_plan { static constexpr int n = 0; };
template <> struct ct_plan<120> { static constexpr int n = 4; static constexpr int r[4] = {4, 2, 3, 5}; };      //  8 kHz
template <> struct ct_plan<180> { static constexpr int n = 4; static constexpr int r[4] = {4, 3, 3, 5}; };      // 12 kHz
template <> struct ct_plan<360> { static constexpr int n = 5; static constexpr int r[5] = {4, 2, 3, 3, 5}; };   // 24 kHz
template <> struct ct_plan<480> { static constexpr int n = 5; static constexpr int r[5] = {4, 4, 2, 3, 5}; };   // 32 kHz
template <> struct ct_plan<720> { static constexpr int n = 5; static constexpr int r[5] = {4, 4, 3, 3, 5}; };   // 48 kHz

// One stage of radix R (S = product of the earlier radices) of an NPTS-point transform, in place.  load(u, i) hands
// over input i of the lane's u-th butterfly (point lane + 64 u + i NPTS/R): the buffer itself, or what the first stage
// of a transform is fed from.
template <int NPTS, int R, int S, class Load>
__device__ __forceinline__ void ct_stage(double* buf, const double2* __restrict__ twp, const int lane, Load load) {
  constexpr int SPAN = NPTS / R, NP = (SPAN + 63) / 64;
  constexpr bool LAST = (S * R == NPTS);
  cplx a[NP][R];
#pragma unroll
  for (int u = 0; u < NP; u++) {
#pragma unroll
    for (int i = 0; i < R; i++) a[u][i] = load(u, i);
  }
  wave_sync();  // compiler: no store of this stage above its loads
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const int b = lane + 64 * u;
    if (b < SPAN) {
      cplx o[R];
      ct_bfly<R>(a[u], o);
      const int p = b / S, q = b - p * S;
      const int base = q + S * R * p, tp = S * p;
      st(buf, base, o[0]);
#pragma unroll
      for (int j = 1; j < R; j++) st(buf, base + j * S, LAST ? o[j] : cmul_tw(o[j], twp[j * tp]));
    }
  }
  wave_sync();
}
// the usual source of a stage: the buffer (lanes beyond the last butterfly of a partial pass read point 0's group)
template <int NPTS, int R>
struct ct_from_buf {
  const double* buf;
  int lane;
  __device__ __forceinline__ cplx operator()(int u, int i) const {
    constexpr int SPAN = NPTS / R;
    const int b = (lane + 64 * u < SPAN) ? lane + 64 * u : 0;
    return ld(buf, b + i * SPAN);
  }
};

// stages ST .. of the plan of W, each from the buffer
template <int W, int ST, int S>
__device__ __forceinline__ void ct_stages_from(double* buf, const double2* __restrict__ twp, const int lane) {
  if constexpr (ST < ct_plan<W>::n) {
    constexpr int R = ct_plan<W>::r[ST];
    ct_stage<W, R, S>(buf, twp, lane, ct_from_buf<W, R>{buf, lane});
    ct_stages_from<W, ST + 1, S * R>(buf, twp, lane);
  }
}

// Last stage of an NPTS-point transform, odd prime radix R = 2H+1 by conjugate-symmetric pairs (DESIGN.md "DFT spec"),
// one lane per butterfly reading its R points and writing the same R; wc/ws = (cos, -sin)(2 pi k / R), k = 1..R-1.
template <int NPTS, int R>
__device__ __forceinline__ void ct_stage_prime_last(double* buf, const int lane, const double (&wc)[R - 1], const double (&ws)[R - 1]) {
  constexpr int SPAN = NPTS / R, H = (R - 1) / 2;
  static_assert(SPAN <= 64, "one pass");
  if (lane < SPAN) {
    cplx a[R];
#pragma unroll
    for (int i = 0; i < R; i++) a[i] = ld(buf, lane + SPAN * i);
    cplx uu[H], vv[H];
    cplx B0 = a[0];
#pragma unroll
    for (int i = 1; i <= H; i++) {
      uu[i - 1] = {a[i].r + a[R - i].r, a[i].i + a[R - i].i};
      vv[i - 1] = {a[i].r - a[R - i].r, a[i].i - a[R - i].i};
      B0.r = B0.r + uu[i - 1].r; B0.i = B0.i + uu[i - 1].i;
    }
    st(buf, lane, B0);
#pragma unroll
    for (int jj = 1; jj <= H; jj++) {
      cplx Pj = a[0], Qj = {0.0, 0.0};
#pragma unroll
      for (int i = 1; i <= H; i++) {
        const double wx = wc[(i * jj) % R - 1], wy = ws[(i * jj) % R - 1];
        Pj.r = spx_acc(Pj.r, wx, uu[i - 1].r); Pj.i = spx_acc(Pj.i, wx, uu[i - 1].i);
        if (i == 1) { Qj.r = wy * vv[0].r; Qj.i = wy * vv[0].i; }
        else { Qj.r = spx_acc(Qj.r, wy, vv[i - 1].r); Qj.i = spx_acc(Qj.i, wy, vv[i - 1].i); }
      }
      st(buf, lane + SPAN * jj, cplx{Pj.r - Qj.i, Pj.i + Qj.r});
      st(buf, lane + SPAN * (R - jj), cplx{Pj.r + Qj.i, Pj.i - Qj.r});
    }
  }
  wave_sync();
}

// a table pointer as this frame's own: keeps the compiler from hoisting the lane's loads from it out of the loop over the
// tile's frames (they are the same for every frame, and live in registers where that pays; where it does not, the
// kernel would spill)
template <class T>
__device__ __forceinline__ const T* per_frame(const T* p) {
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ double uniform_f64(double v) {  // a wave-uniform double, pinned to scalar registers
  return __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(v) >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(v)));
}

// packed point n of frame j: z[n] = v[2n] + i v[2n+1], v = pre-emphasised samples times the window (speedy.c:416-425,
// :442); fr = the frame's first mono sample in the staged span, w0 / w1 = window[2n], window[2n+1]
// One windowed, pre-emphasised sample as the transform takes it (speedy.c:553-565, 416-425, 442):
//   x = (float)(m / 32768.0), xp likewise;  y = (float)(1.0 (double)x - 0.97 (double)xp);  v = (double)(y * w)
// from the int16 samples m, mp THEMSELVES (round 5).  x = m 2^-15 exactly (sixteen bits in a float), and every operation behind it
// commutes with that power-of-two factor as long as nothing is subnormal -- RN(0.97 (mp 2^-15)) = RN(0.97 mp) 2^-15, the
// difference and its rounding to float likewise, RN32((yy 2^-15) w) = RN32(yy (w 2^-15)) -- and nothing is: a nonzero m - 0.97 mp
// is at least ~0.01 in magnitude, the Hamming window at least 0.08.  So the scale moves into the window value (w15 = w 2^-15, once
// per lane) and the three value-preserving conversions per sample (double -> float -> double of x, and the scaling itself) go:
// 44 -> 26 instructions for the four samples of a lane in the 16 kHz kernel.  Bit-identical, the spec (oracle) is untouched.
// md, mpd: (double)m, (double)mp.  -DSPX_PREEMPH_V1: the literal sequence.
#ifdef SPX_PREEMPH_V1
#define SPX_WIN_SCALE 1.0f
__device__ __forceinline__ float spx_preemph_win_f32(double md, double mpd, float w) {
  const float x = (float)(md / 32768.0), xp = (float)(mpd / 32768.0);
  const float y = (float)(1.0 * (double)x - 0.97 * (double)xp);
  return y * w;
}
#else
#define SPX_WIN_SCALE 0x1p-15f
__device__ __forceinline__ float spx_preemph_win_f32(double md, double mpd, float w15) {
  const float yy = (float)(md - 0.97 * mpd);
  return yy * w15;
}
#endif
__device__ __forceinline__ double spx_preemph_win(double md, double mpd, float w) { return (double)spx_preemph_win_f32(md, mpd, w); }
__device__ __forceinline__ cplx packed_point(const short* fr, int n, int j, int prev0, float w0, float w1) {
  const int i0 = 2 * n;
  const int m0 = fr[i0], m1 = fr[i0 + 1];
  const int mp0 = (i0 > 0) ? (int)fr[i0 - 1] : ((j > 0) ? (int)fr[prev0] : 0);
  const double d0 = (double)m0;
  return {spx_preemph_win(d0, (double)mp0, w0 * SPX_WIN_SCALE), spx_preemph_win((double)m1, d0, w1 * SPX_WIN_SCALE)};
}

// WCT != 0: the kernel is compiled for that window size (16 kHz: W = 240 = 4*4*3*5) -- see the phase-1 comment.
#ifndef SPX_AN_W_330
#define SPX_AN_W_330 3   // waves per SIMD the 22.05 kHz instantiation is compiled for (A/B: 4 = 128 registers)
#endif
#ifndef SPX_AN_W_BIG
#define SPX_AN_W_BIG 2   // waves per SIMD the 44.1 / 48 kHz instantiations are compiled for (A/B: 3 = 168 registers)
#endif
template <int TF, int WCT>
__global__ void __launch_bounds__(SPX_BLOCK, (WCT == 240 || WCT == 120 || WCT == 180) ? 4 : WCT == 330 ? SPX_AN_W_330 : (WCT == 661 || WCT == 720) ? SPX_AN_W_BIG : (WCT == 480 || WCT == 360) ? 2 : 1)  // (.., waves per SIMD the register count must allow: the concurrent mode's budgets)
spx_analysis_kernel(SpxPlanDev P, const SpxStreamDev* __restrict__ streams, int n_streams,
                    const int16_t* __restrict__ in_base, SpxFrameRec* __restrict__ rec, SpxTapsDev taps,
                    const int* __restrict__ tile_order, int* tile_flags, const float* __restrict__ frames,
                    int frame_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int W = WCT ? WCT : P.W, B = P.B, N = 2 * W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // tile -> (stream, first frame): binary search over streams[].first_tile
  // tile_order (optional): launch order -> tile id, earliest frames of every stream first, so that a walk kernel
  // running concurrently finds its next chunk ready
  const int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
  int lo = 0, hi = n_streams - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (streams[mid].first_tile <= tile) lo = mid; else hi = mid - 1;
  }
  const SpxStreamDev S = streams[lo];
  const int T = S.n_frames;
  const int j0 = S.frame_begin + (tile - S.first_tile) * TF;
  const int j1 = min(j0 + TF, T);
  const int C = S.channels;
  const int16_t* __restrict__ in = in_base + S.in_off;

  double* work = reinterpret_cast<double*>(lds);
  const size_t wb = work_bytes(W, TF, WCT != 0, (WCT != 0) ? 4 : P.dft_waves);
  float* mags = reinterpret_cast<float*>(lds + wb);
  const size_t mags_b = (((size_t)(TF + 1) * spx_mag_stride(W) * sizeof(float)) + 15) & ~(size_t)15;
  float* fE = reinterpret_cast<float*>(lds + wb + mags_b);
  float* fThr = fE + (TF + 1);
  float* fInv = fThr + (TF + 1);
  const int MS = spx_mag_stride(W);  // mags row stride (floats)

  const size_t small_b = (((size_t)3 * (TF + 1) * sizeof(float)) + 15) & ~(size_t)15;
  const double* ltw = P.tw;    // twiddles stay in global memory (L1-resident, 16 B per lane per use)
  const double* ltw2 = P.tw2;
  short* smono = reinterpret_cast<short*>(lds + wb + mags_b + small_b);    // mono mix of the tile's input span

  double* bufA = work + (size_t)wave * (WCT != 0 ? 2 : 4) * W;
  double* bufB = (WCT != 0) ? bufA : bufA + 2 * W;  // compiled-in sizes: one buffer per wave, every stage in place

  // ---------------- phase 0: the tile's input span into LDS (all loads in flight) -------------
  const int jfirst = (j0 > 0) ? j0 - 1 : 0;            // first frame whose samples are needed
  const int64_t sbase = (int64_t)jfirst * B;           // absolute index of smono[0]
  const int nstage = (j1 - jfirst) * B + (W - B);      // frames jfirst .. j1-1
  if (frames != nullptr) {
    // explicit float frames (unit-level API): nothing to stage
  } else if (C == 1) {
    const int16_t* __restrict__ src = in + sbase;
    for (int k0 = tid; k0 < nstage; k0 += 8 * SPX_BLOCK) {
      short v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int k = k0 + u * SPX_BLOCK; v[u] = (k < nstage) ? src[k] : (short)0; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int k = k0 + u * SPX_BLOCK; if (k < nstage) smono[k] = v[u]; }
    }
  } else {
    for (int k = tid; k < nstage; k += SPX_BLOCK) smono[k] = (short)mono_sample(in, sbase + k, C);
  }
  __syncthreads();

  ASTAMP_DECL
  // ---------------- phase 1: spectra of slots 0..TF (slot s = frame j0-1+s), one wave per slot ----------
  if constexpr (WCT == 240) {
    // W = 240 = 4*4*3*5, compiled in: every butterfly index, twiddle index and loop bound is a constant of the lane, the
    // twiddles, window values and untangle factors a lane needs are loaded ONCE (they are the same for every frame) and
    // live in registers; the window / pre-emphasis pass is fused into the first radix-4 stage, whose upper two inputs
    // are the zero padding (x + 0 and x - 0 are x); multiplications by the twiddle 1 (output 0 of every butterfly, all
    // of the last stage) are skipped (x*1 - y*(-0) is x).  Same operations in the same order otherwise -- the
    // magnitudes are bit-identical to the generic path and to the oracle (signs of exact zeros aside, which no
    // magnitude depends on).  The stages work IN PLACE on one buffer per wave: a wave's LDS operations are served in
    // issue order, so a stage that issues all its loads before its first store needs no second buffer (bufB == bufA).
    const double2* twp = reinterpret_cast<const double2*>(P.tw);
    const double2* tw2p = reinterpret_cast<const double2*>(P.tw2);
    const int b = lane;
    const bool on60 = b < 60, on48 = b < 48;
    const int bb = on60 ? b : 0;
    double2 w1[3], w2[3], w3[2][2], wu[4];
    float wn[4];
#pragma unroll
    for (int j = 1; j < 4; j++) { w1[j - 1] = twp[bb * j]; w2[j - 1] = twp[4 * (bb >> 2) * j]; }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int b3 = lane + 64 * u, p3 = (b3 < 80) ? (b3 >> 4) : 0;
      w3[u][0] = twp[16 * p3];
      w3[u][1] = twp[32 * p3];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = lane + 64 * u; wu[u] = tw2p[k < 240 ? k : 0]; }
    wn[0] = P.window[2 * bb]; wn[1] = P.window[2 * bb + 1]; wn[2] = P.window[2 * bb + 120]; wn[3] = P.window[2 * bb + 121];
#pragma unroll
    for (int k = 0; k < 4; k++) wn[k] *= SPX_WIN_SCALE;   // (spx_preemph_win: the samples' 2^-15 lives in the window value)
    for (int s = wave; s <= TF; s += 4) {
      const int j = j0 - 1 + s;
      float* mrow = mags + (size_t)s * MS;
      if (j < 0 || j >= j1) {  // outside the stream (or the tile's tail): zero spectrum
        for (int k = lane; k < 240; k += SPX_WAVE) mrow[k] = 0.0f;
        continue;
      }
      const short* fr = smono + (size_t)(j - jfirst) * B;  // this frame's 240 mono samples
      // stage 1 (radix 4, s = 1) on packed points z[n] = v[2n] + i v[2n+1]; v = windowed pre-emphasised samples,
      // z[n] = 0 for n >= 120: butterfly b takes z[b], z[b+60], 0, 0 and writes points 4b .. 4b+3
      if (on60) {
        const int i0 = 2 * b, i1 = 2 * b + 120;
        const int m0 = fr[i0], m1 = fr[i0 + 1], m2 = fr[i1], m3 = fr[i1 + 1];
        const int mp0 = (i0 > 0) ? (int)fr[i0 - 1] : ((j > 0) ? (int)fr[(240 - B) - 1] : 0);
        const int mp2 = fr[i1 - 1];
        const double d0 = (double)m0, d2 = (double)m2;
        const cplx a0 = {spx_preemph_win(d0, (double)mp0, wn[0]), spx_preemph_win((double)m1, d0, wn[1])},
                   a1 = {spx_preemph_win(d2, (double)mp2, wn[2]), spx_preemph_win((double)m3, d2, wn[3])};
        const cplx b0 = {a0.r + a1.r, a0.i + a1.i}, b2 = {a0.r - a1.r, a0.i - a1.i};
        const cplx b1 = {a0.r + a1.i, a0.i - a1.r}, b3 = {a0.r - a1.i, a0.i + a1.r};
        st(bufA, 4 * b, b0);
        st(bufA, 4 * b + 1, cmul_tw(b1, w1[0]));
        st(bufA, 4 * b + 2, cmul_tw(b2, w1[1]));
        st(bufA, 4 * b + 3, cmul_tw(b3, w1[2]));
      }
      wave_sync();
      ASTAMP(0);
      // stage 2 (radix 4, s = 4): bufA -> bufB
      if (on60) {
        const cplx a0 = ld(bufA, b), a1 = ld(bufA, b + 60), a2 = ld(bufA, b + 120), a3 = ld(bufA, b + 180);
        wave_sync();  // in place: every lane's loads are issued before any store
        const cplx t0 = {a0.r + a2.r, a0.i + a2.i}, t1 = {a0.r - a2.r, a0.i - a2.i};
        const cplx t2 = {a1.r + a3.r, a1.i + a3.i}, t3 = {a1.r - a3.r, a1.i - a3.i};
        const cplx b0 = {t0.r + t2.r, t0.i + t2.i}, b2 = {t0.r - t2.r, t0.i - t2.i};
        const cplx b1 = {t1.r + t3.i, t1.i - t3.r}, b3 = {t1.r - t3.i, t1.i + t3.r};
        const int o = (b & 3) + 16 * (b >> 2);
        st(bufB, o, b0);
        st(bufB, o + 4, cmul_tw(b1, w2[0]));
        st(bufB, o + 8, cmul_tw(b2, w2[1]));
        st(bufB, o + 12, cmul_tw(b3, w2[2]));
      }
      wave_sync();
      // stage 3 (radix 3, s = 16), 80 butterflies = two passes: both passes' loads first, then the stores (in place)
      {
        cplx q0[2], q1[2], q2[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int b3i = (lane + 64 * u < 80) ? lane + 64 * u : 0;
          q0[u] = ld(bufB, b3i); q1[u] = ld(bufB, b3i + 80); q2[u] = ld(bufB, b3i + 160);
        }
        wave_sync();  // compiler: no store of this stage above the loads
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int b3i = lane + 64 * u;
          if (b3i < 80) {
            const cplx a0 = q0[u], a1 = q1[u], a2 = q2[u];
            const cplx t1 = {a1.r + a2.r, a1.i + a2.i};
            const cplx t2 = {spx_bf3h(a0.r, t1.r), spx_bf3h(a0.i, t1.i)};
            const cplx t3 = {S3_1 * (a1.r - a2.r), S3_1 * (a1.i - a2.i)};
            const cplx b0 = {a0.r + t1.r, a0.i + t1.i};
            const cplx b1 = {t2.r + t3.i, t2.i - t3.r}, b2 = {t2.r - t3.i, t2.i + t3.r};
            const int o = (b3i & 15) + 48 * (b3i >> 4);
            st(bufA, o, b0);
            st(bufA, o + 16, cmul_tw(b1, w3[u][0]));
            st(bufA, o + 32, cmul_tw(b2, w3[u][1]));
          }
        }
      }
      wave_sync();
      // stage 4 (radix 5, s = 48, last: every twiddle is 1): bufA -> bufB
      if (on48) {
        const cplx a0 = ld(bufA, b), a1 = ld(bufA, b + 48), a2 = ld(bufA, b + 96), a3 = ld(bufA, b + 144),
                   a4 = ld(bufA, b + 192);
        wave_sync();
        const cplx t1 = {a1.r + a4.r, a1.i + a4.i}, t2 = {a2.r + a3.r, a2.i + a3.i};
        const cplx t3 = {a1.r - a4.r, a1.i - a4.i}, t4 = {a2.r - a3.r, a2.i - a3.i};
        const cplx b0 = {(a0.r + t1.r) + t2.r, (a0.i + t1.i) + t2.i};
        const cplx m1 = {spx_bf5m(a0.r, C5_1, t1.r, C5_2, t2.r), spx_bf5m(a0.i, C5_1, t1.i, C5_2, t2.i)};
        const cplx m2 = {spx_bf5m(a0.r, C5_2, t1.r, C5_1, t2.r), spx_bf5m(a0.i, C5_2, t1.i, C5_1, t2.i)};
        const cplx n1 = {spx_bf5n(S5_1, t3.r, S5_2, t4.r), spx_bf5n(S5_1, t3.i, S5_2, t4.i)};
        const cplx n2 = {spx_bf5d(S5_2, t3.r, S5_1, t4.r), spx_bf5d(S5_2, t3.i, S5_1, t4.i)};
        const cplx b1 = {m1.r + n1.i, m1.i - n1.r}, b4 = {m1.r - n1.i, m1.i + n1.r};
        const cplx b2 = {m2.r + n2.i, m2.i - n2.r}, b3 = {m2.r - n2.i, m2.i + n2.r};
        st(bufB, b, b0);
        st(bufB, b + 48, b1);
        st(bufB, b + 96, b2);
        st(bufB, b + 144, b3);
        st(bufB, b + 192, b4);
      }
      wave_sync();
      ASTAMP(1);
      // untangle the packed transform:  X[k] = E[k] + e^{-2 pi i k/N} O[k]
      float* spec_out = taps.spectrogram ? taps.spectrogram + (size_t)(S.frame_off + j) * 480 : nullptr;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = lane + 64 * u;
        if (k < 240) {
          const int k2 = (k == 0) ? 0 : 240 - k;
          const cplx a = ld(bufB, k), c = ld(bufB, k2);
          const double b_r = c.r, b_i = -c.i;
          const float mag = spx_untangle_mag(a.r, a.i, b_r, b_i, wu[u]);
          mrow[k] = mag;
          if (spec_out) {
            spec_out[k] = mag;
            if (k > 0) spec_out[480 - k] = mag;
            else spec_out[240] = (float)__builtin_fabs(a.r - a.i);
          }
        }
      }
      wave_sync();
      ASTAMP(2);
    }
  } else if constexpr (WCT == 330) {
    // W = 330 = 2*3*5*11 (22.05 kHz), compiled in like W = 240: constant butterfly indices, the first two stages'
    // twiddles and the window values in registers, the window pass fused into the first (radix-2) stage whose second
    // input is the zero padding, unit twiddles skipped, every stage in place on one buffer per wave -- the radix-11 stage
    // (conjugate-symmetric pairs, DESIGN.md "DFT spec") by one lane per butterfly, which reads its eleven points and
    // writes the same eleven.  Same operations in the same order as the plan-driven path otherwise.
    const double2* twp = reinterpret_cast<const double2*>(P.tw);
    const double2* tw2p = reinterpret_cast<const double2*>(P.tw2);
    double2 w1[3], w2[2][2];
    float wn[3][2];
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int b1 = (lane + 64 * u < 165) ? lane + 64 * u : 0;
      w1[u] = twp[b1];
      wn[u][0] = P.window[2 * b1] * SPX_WIN_SCALE; wn[u][1] = P.window[2 * b1 + 1] * SPX_WIN_SCALE;   // (spx_preemph_win)
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int p2 = (lane + 64 * u < 110) ? ((lane + 64 * u) >> 1) : 0;
      w2[u][0] = twp[2 * p2];
      w2[u][1] = twp[4 * p2];
    }
    // the ten eleventh roots of unity the last stage multiplies by: wave-uniform, pinned to scalar registers
    double wc[10], ws[10];
#pragma unroll
    for (int k = 1; k <= 10; k++) {
      const double2 w = twp[k * 30];
      wc[k - 1] = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(w.x) >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(w.x)));
      ws[k - 1] = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(w.y) >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(w.y)));
    }
    for (int s = wave; s <= TF; s += 4) {
      const int j = j0 - 1 + s;
      float* mrow = mags + (size_t)s * MS;
      if (j < 0 || j >= j1) {  // outside the stream (or the tile's tail): zero spectrum
        for (int k = lane; k < 330; k += SPX_WAVE) mrow[k] = 0.0f;
        continue;
      }
      const short* fr = smono + (size_t)(j - jfirst) * B;  // this frame's 330 mono samples
      // stage 1 (radix 2, s = 1) on packed points z[n] = v[2n] + i v[2n+1], z[n] = 0 for n >= 165: butterfly b takes
      // z[b] and 0 and writes points 2b (z[b]) and 2b+1 (z[b] times its twiddle)
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int b = lane + 64 * u;
        if (b < 165) {
          const int i0 = 2 * b;
          const int m0 = fr[i0], m1 = fr[i0 + 1];
          const int mp0 = (i0 > 0) ? (int)fr[i0 - 1] : ((j > 0) ? (int)fr[(330 - B) - 1] : 0);
          const double d0 = (double)m0;
          const cplx a0 = {spx_preemph_win(d0, (double)mp0, wn[u][0]), spx_preemph_win((double)m1, d0, wn[u][1])};
          st(bufA, 2 * b, a0);
          st(bufA, 2 * b + 1, cmul_tw(a0, w1[u]));
        }
      }
      wave_sync();
      ASTAMP(0);
      // stage 2 (radix 3, s = 2), 110 butterflies = two passes: both passes' loads first, then the stores (in place)
      {
        cplx q0[2], q1[2], q2[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int b = (lane + 64 * u < 110) ? lane + 64 * u : 0;
          q0[u] = ld(bufA, b); q1[u] = ld(bufA, b + 110); q2[u] = ld(bufA, b + 220);
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int b = lane + 64 * u;
          if (b < 110) {
            const cplx a0 = q0[u], a1 = q1[u], a2 = q2[u];
            const cplx t1 = {a1.r + a2.r, a1.i + a2.i};
            const cplx t2 = {spx_bf3h(a0.r, t1.r), spx_bf3h(a0.i, t1.i)};
            const cplx t3 = {S3_1 * (a1.r - a2.r), S3_1 * (a1.i - a2.i)};
            const cplx b0 = {a0.r + t1.r, a0.i + t1.i};
            const cplx b1 = {t2.r + t3.i, t2.i - t3.r}, b2 = {t2.r - t3.i, t2.i + t3.r};
            const int o = (b & 1) + 6 * (b >> 1);
            st(bufA, o, b0);
            st(bufA, o + 2, cmul_tw(b1, w2[u][0]));
            st(bufA, o + 4, cmul_tw(b2, w2[u][1]));
          }
        }
      }
      wave_sync();
      // stage 3 (radix 5, s = 6), 66 butterflies = two passes (the second with two lanes), loads first
      {
        cplx q[2][5];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int b = (lane + 64 * u < 66) ? lane + 64 * u : 0;
#pragma unroll
          for (int i = 0; i < 5; i++) q[u][i] = ld(bufA, b + 66 * i);
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int b = lane + 64 * u;
          if (b < 66) {
            const cplx a0 = q[u][0], a1 = q[u][1], a2 = q[u][2], a3 = q[u][3], a4 = q[u][4];
            const cplx t1 = {a1.r + a4.r, a1.i + a4.i}, t2 = {a2.r + a3.r, a2.i + a3.i};
            const cplx t3 = {a1.r - a4.r, a1.i - a4.i}, t4 = {a2.r - a3.r, a2.i - a3.i};
            const cplx b0 = {(a0.r + t1.r) + t2.r, (a0.i + t1.i) + t2.i};
            const cplx m1 = {spx_bf5m(a0.r, C5_1, t1.r, C5_2, t2.r), spx_bf5m(a0.i, C5_1, t1.i, C5_2, t2.i)};
            const cplx m2 = {spx_bf5m(a0.r, C5_2, t1.r, C5_1, t2.r), spx_bf5m(a0.i, C5_2, t1.i, C5_1, t2.i)};
            const cplx n1 = {spx_bf5n(S5_1, t3.r, S5_2, t4.r), spx_bf5n(S5_1, t3.i, S5_2, t4.i)};
            const cplx n2 = {spx_bf5d(S5_2, t3.r, S5_1, t4.r), spx_bf5d(S5_2, t3.i, S5_1, t4.i)};
            const cplx b1 = {m1.r + n1.i, m1.i - n1.r}, b4 = {m1.r - n1.i, m1.i + n1.r};
            const cplx b2 = {m2.r + n2.i, m2.i - n2.r}, b3 = {m2.r - n2.i, m2.i + n2.r};
            const int p = (int)(((unsigned)b * 10923u) >> 16);  // b / 6 for b < 66
            const int o = (b - 6 * p) + 30 * p, tp = 6 * p;
            st(bufA, o, b0);
            st(bufA, o + 6, cmul_tw(b1, twp[tp]));
            st(bufA, o + 12, cmul_tw(b2, twp[2 * tp]));
            st(bufA, o + 18, cmul_tw(b3, twp[3 * tp]));
            st(bufA, o + 24, cmul_tw(b4, twp[4 * tp]));
          }
        }
      }
      wave_sync();
      // stage 4 (radix 11, s = 30, last: every twiddle is 1): one lane per butterfly, its eleven points in place
      if (lane < 30) {
        cplx a[11];
#pragma unroll
        for (int i = 0; i < 11; i++) a[i] = ld(bufA, lane + 30 * i);
        cplx u5[5], v5[5];
        cplx B0 = a[0];
#pragma unroll
        for (int i = 1; i <= 5; i++) {
          u5[i - 1] = {a[i].r + a[11 - i].r, a[i].i + a[11 - i].i};
          v5[i - 1] = {a[i].r - a[11 - i].r, a[i].i - a[11 - i].i};
          B0.r = B0.r + u5[i - 1].r; B0.i = B0.i + u5[i - 1].i;
        }
        st(bufA, lane, B0);
#pragma unroll
        for (int jj = 1; jj <= 5; jj++) {
          cplx Pj = a[0], Qj = {0.0, 0.0};
#pragma unroll
          for (int i = 1; i <= 5; i++) {
            const double wx = wc[(i * jj) % 11 - 1], wy = ws[(i * jj) % 11 - 1];  // (cos, -sin)(2 pi k / 11), k = (i jj) mod 11
            // (spx_acc: fused in DFT spec v2, as orc_butterfly_v2's odd-prime branch and ct_stage_prime_last -- until the end of round 5
            // this hand-written stage had kept the unfused sums of spec v1: an fp64 last-bit difference from the oracle that a float
            // magnitude shows about once in 2^29 values)
#ifdef SPX_R11_V1   // (what the stage did until then: tools/r11_probe.py compares the two builds with the oracle)
            Pj.r = Pj.r + wx * u5[i - 1].r; Pj.i = Pj.i + wx * u5[i - 1].i;
            if (i == 1) { Qj.r = wy * v5[0].r; Qj.i = wy * v5[0].i; }
            else { Qj.r = Qj.r + wy * v5[i - 1].r; Qj.i = Qj.i + wy * v5[i - 1].i; }
#else
            Pj.r = spx_acc(Pj.r, wx, u5[i - 1].r); Pj.i = spx_acc(Pj.i, wx, u5[i - 1].i);
            if (i == 1) { Qj.r = wy * v5[0].r; Qj.i = wy * v5[0].i; }
            else { Qj.r = spx_acc(Qj.r, wy, v5[i - 1].r); Qj.i = spx_acc(Qj.i, wy, v5[i - 1].i); }
#endif
          }
          st(bufA, lane + 30 * jj, cplx{Pj.r - Qj.i, Pj.i + Qj.r});
          st(bufA, lane + 30 * (11 - jj), cplx{Pj.r + Qj.i, Pj.i - Qj.r});
        }
      }
      wave_sync();
      ASTAMP(1);
      // untangle the packed transform:  X[k] = E[k] + e^{-2 pi i k/N} O[k]
      float* spec_out = taps.spectrogram ? taps.spectrogram + (size_t)(S.frame_off + j) * 660 : nullptr;
#pragma unroll
      for (int u = 0; u < 6; u++) {
        const int k = lane + 64 * u;
        if (k < 330) {
          const int k2 = (k == 0) ? 0 : 330 - k;
          const cplx a = ld(bufA, k), c = ld(bufA, k2);
          const double b_r = c.r, b_i = -c.i;
          const float mag = spx_untangle_mag(a.r, a.i, b_r, b_i, tw2p[k]);
          mrow[k] = mag;
          if (spec_out) {
            spec_out[k] = mag;
            if (k > 0) spec_out[660 - k] = mag;
            else spec_out[330] = (float)__builtin_fabs(a.r - a.i);
          }
        }
      }
      wave_sync();
      ASTAMP(2);
    }
  } else if constexpr (ct_plan<WCT>::n != 0) {
    // A window size with a compiled-in plan (8 / 12 / 24 / 32 / 48 kHz): the stages of ct_stage, the first fed straight from the
    // staged samples (its upper inputs are the zero padding); the window values a lane needs are the same for every frame
    // and the compiler keeps them in registers.
    constexpr int R0 = ct_plan<WCT>::r[0], SPAN0 = WCT / R0;
    static_assert(WCT % 2 == 0 && (R0 == 4 || R0 == 2), "packed points 0 .. W/2 - 1 are inputs 0 .. R0/2 - 1 of the first stage");
    const double2* twp = reinterpret_cast<const double2*>(P.tw);
    const double2* tw2p = reinterpret_cast<const double2*>(P.tw2);
    const float* win = P.window;
    for (int s = wave; s <= TF; s += 4) {
      const int j = j0 - 1 + s;
      float* mrow = mags + (size_t)s * MS;
      if (j < 0 || j >= j1) {  // outside the stream (or the tile's tail): zero spectrum
        for (int k = lane; k < WCT; k += SPX_WAVE) mrow[k] = 0.0f;
        continue;
      }
      const short* fr = smono + (size_t)(j - jfirst) * B;  // this frame's W mono samples
      const int prev0 = (WCT - B) - 1;
      ct_stage<WCT, R0, 1>(bufA, twp, lane, [&](int u, int i) -> cplx {
        if (i * SPAN0 >= WCT / 2) return cplx{0.0, 0.0};
        const int n = ((lane + 64 * u < SPAN0) ? lane + 64 * u : 0) + SPAN0 * i;
        return packed_point(fr, n, j, prev0, win[2 * n], win[2 * n + 1]);
      });
      ASTAMP(0);
      ct_stages_from<WCT, 1, R0>(bufA, twp, lane);
      ASTAMP(1);
      // untangle the packed transform:  X[k] = E[k] + e^{-2 pi i k/N} O[k]
      float* spec_out = taps.spectrogram ? taps.spectrogram + (size_t)(S.frame_off + j) * (2 * WCT) : nullptr;
#pragma unroll 4
      for (int u = 0; u < (WCT + 63) / 64; u++) {
        const int k = lane + 64 * u;
        if (k < WCT) {
          const int k2 = (k == 0) ? 0 : WCT - k;
          const cplx a = ld(bufA, k), c = ld(bufA, k2);
          const double b_r = c.r, b_i = -c.i;
          const float mag = spx_untangle_mag(a.r, a.i, b_r, b_i, tw2p[k]);
          mrow[k] = mag;
          if (spec_out) {
            spec_out[k] = mag;
            if (k > 0) spec_out[2 * WCT - k] = mag;
            else spec_out[WCT] = (float)__builtin_fabs(a.r - a.i);
          }
        }
      }
      wave_sync();
      ASTAMP(2);
    }
  } else if constexpr (WCT == 661) {
    // W = 661, prime (44.1 kHz): Rader's algorithm (DESIGN.md "DFT spec") over the compiled-in M = 660 = 4*3*5*11 plan.
    // a[p] = z[g^p] is gathered straight from the staged samples by the first stage of the first transform (indices in
    // registers), conj(A .* F(b)) is formed by the loads of the second transform's first stage, and the
    // untangle pass gathers Z[k] = z[0] + c[q(k)] through the discrete-logarithm table instead of a scatter pass.
    constexpr int M = 660;
    const double2* twMp = reinterpret_cast<const double2*>(P.twM);
    const double2* bfp = reinterpret_cast<const double2*>(P.bfft);
    const double2* tw2p = reinterpret_cast<const double2*>(P.tw2);
    int pn[3][4];
#pragma unroll
    for (int u = 0; u < 3; u++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int pt = ((lane + 64 * u < 165) ? lane + 64 * u : 0) + 165 * i;
        const int n = P.perm[pt];          // 1 .. 660; points above 330 are the zero padding
        pn[u][i] = (n <= 330) ? n : -1;
      }
    }
    double wc[10], ws[10];
#pragma unroll
    for (int k = 1; k <= 10; k++) {
      const double2 w = twMp[k * 60];
      wc[k - 1] = uniform_f64(w.x);
      ws[k - 1] = uniform_f64(w.y);
    }
    unsigned qq[11];  // q(k) | q(661 - k) << 16 for the lane's bins k = lane + 64 u
#pragma unroll
    for (int u = 0; u < 11; u++) {
      const int k = lane + 64 * u;
      const int ka = (k >= 1 && k < 661) ? k : 1;
      qq[u] = (unsigned)P.qlog[ka] | ((unsigned)P.qlog[661 - ka] << 16);
    }
    const double inv = 1.0 / (double)M;
    for (int s = wave; s <= TF; s += 4) {
      const int j = j0 - 1 + s;
      float* mrow = mags + (size_t)s * MS;
      if (j < 0 || j >= j1) {  // outside the stream (or the tile's tail): zero spectrum
        for (int k = lane; k < 661; k += SPX_WAVE) mrow[k] = 0.0f;
        continue;
      }
      const short* fr = smono + (size_t)(j - jfirst) * B;  // this frame's 661 mono samples
      const int prev0 = (661 - B) - 1;
      // the frame's 661 pre-emphasised, windowed samples as floats into the head of the wave's buffer (one lane per sample,
      // every read in order), so that the gather below is one 8-byte read per point.  (Gathering the samples and the window
      // values themselves -- three 16-bit LDS reads and two scattered global loads per point -- was a fifth of the kernel.)
      float* vf = reinterpret_cast<float*>(bufA);
      const float* winq = per_frame(P.window);
#pragma unroll
      for (int u = 0; u < 11; u++) {
        const int i = lane + 64 * u;
        if (i < 661) {
          const int m = fr[i];
          const int mp = (i > 0) ? (int)fr[i - 1] : ((j > 0) ? (int)fr[prev0] : 0);
          vf[i] = spx_preemph_win_f32((double)m, (double)mp, winq[i] * SPX_WIN_SCALE);   // speedy.c:422, :442 / :462
        }
      }
      if (lane == 0) vf[661] = 0.0f;   // the imaginary half of point 330 is padding
      wave_sync();
      const double2* bfq = per_frame(bfp);
      const double2* tw2q = per_frame(tw2p);
      const double2* twMq = per_frame(twMp);
      int ln = lane;  // the lane index as this frame's own: the stages' addresses are computed per frame, not kept in ~150 registers
      asm volatile("" : "+v"(ln));
      const cplx x0 = [&] { const float2 v = *reinterpret_cast<const float2*>(vf); return cplx{(double)v.x, (double)v.y}; }();
      ct_stage<M, 4, 1>(bufA, twMp, ln, [&](int u, int i) -> cplx {
        const int n = pn[u][i];
        const float2 v = *reinterpret_cast<const float2*>(vf + 2 * (n < 0 ? 0 : n));
        return (n < 0) ? cplx{0.0, 0.0} : cplx{(double)v.x, (double)v.y};
      });
      ASTAMP(0);
      ct_stage<M, 3, 4>(bufA, twMq, ln, ct_from_buf<M, 3>{bufA, ln});
      ct_stage<M, 5, 12>(bufA, twMq, ln, ct_from_buf<M, 5>{bufA, ln});
      ct_stage_prime_last<M, 11>(bufA, ln, wc, ws);
      const cplx A0 = ld(bufA, 0);
      ct_stage<M, 4, 1>(bufA, twMp, ln, [&](int u, int i) -> cplx {
        const int k = ((ln + 64 * u < 165) ? ln + 64 * u : 0) + 165 * i;
        const cplx a = ld(bufA, k);
        const double2 b = bfq[k];
        const cplx cc = spx_cmul(a, b.x, b.y);
        return cplx{cc.r, -cc.i};
      });
      ct_stage<M, 3, 4>(bufA, twMq, ln, ct_from_buf<M, 3>{bufA, ln});
      ct_stage<M, 5, 12>(bufA, twMq, ln, ct_from_buf<M, 5>{bufA, ln});
      ct_stage_prime_last<M, 11>(bufA, ln, wc, ws);
      ASTAMP(1);
      // untangle the packed transform:  X[k] = E[k] + e^{-2 pi i k/N} O[k], Z[0] = z[0] + A[0], Z[k] = z[0] + conj(F[q(k)]) / M
      float* spec_out = taps.spectrogram ? taps.spectrogram + (size_t)(S.frame_off + j) * 1322 : nullptr;
      const cplx Z0 = {x0.r + A0.r, x0.i + A0.i};
#pragma unroll
      for (int u = 0; u < 11; u++) {
        const int k = lane + 64 * u;
        if (k < 661) {
          cplx a = Z0, c = Z0;
          if (k > 0) {
            const cplx f1 = ld(bufA, (int)(qq[u] & 0xffffu)), f2 = ld(bufA, (int)(qq[u] >> 16));
            const double c1r = f1.r * inv, c1i = -f1.i * inv, c2r = f2.r * inv, c2i = -f2.i * inv;
            a = {x0.r + c1r, x0.i + c1i};
            c = {x0.r + c2r, x0.i + c2i};
          }
          const double b_r = c.r, b_i = -c.i;
          const float mag = spx_untangle_mag(a.r, a.i, b_r, b_i, tw2q[k]);
          mrow[k] = mag;
          if (spec_out) {
            spec_out[k] = mag;
            if (k > 0) spec_out[1322 - k] = mag;
            else spec_out[661] = (float)__builtin_fabs(a.r - a.i);
          }
        }
      }
      wave_sync();
      ASTAMP(2);
    }
  } else
  for (int s = wave, ndw = (P.dft_waves >= 1 && P.dft_waves <= 4) ? P.dft_waves : 4; s <= TF && wave < ndw; s += ndw) {
    const int j = j0 - 1 + s;
    float* mrow = mags + (size_t)s * MS;
    if (j < 0 || j >= j1) {  // outside the stream (or the tile's tail): zero spectrum
      for (int k = lane; k < W; k += SPX_WAVE) mrow[k] = 0.0f;
      continue;
    }
    const short* fr = smono + (size_t)(j - jfirst) * B;  // this frame's W mono samples
    for (int i = lane; i < 2 * W; i += SPX_WAVE) {
      double v = 0.0;
      if (i < W) {
        float x, xp;
        if (frames != nullptr) {  // uniform: frame j is frames[j*W ..), the filter state the previous frame's last sample
          const float* ff = frames + (size_t)j * W;
          x = ff[i];
          if (i > 0) xp = ff[i - 1];
          else xp = (j > 0) ? ff[-1] : 0.0f;
        } else {
          const int m = fr[i];
          int mp;
          if (i > 0) mp = fr[i - 1];
          else mp = (j > 0) ? (int)fr[(W - B) - 1] : 0;
          x = (float)(m / 32768.0);
          xp = (float)(mp / 32768.0);
        }
        const float y = (frame_mode == 2) ? x : (float)(1.0 * (double)x - 0.97 * (double)xp);  // speedy.c:422
        v = (double)(y * P.window[i]);                                  // speedy.c:442 / :462
      }
      bufA[i] = v;
    }
    wave_sync();
    ASTAMP(0);
    double* x = bufA;
    double* y = bufB;
    if (P.rader) {
      // W prime: X[0] = x[0] + A[0], X[g^-q] = x[0] + c[q] with c = a (*) b cyclically over M = W-1 points,
      // a[p] = x[g^p], b[q] = w^(g^-q); A = F(a), C = A .* F(b), c = conj(F(conj(C))) / M (oracle: orc_plan_execute_rader)
      const int M = W - 1;
      const cplx x0 = ld(x, 0);
      for (int p = lane; p < M; p += SPX_WAVE) {
        const cplx v = ld(x, P.perm[p]);
        *reinterpret_cast<double2*>(y + 2 * p) = make_double2(v.r, v.i);
      }
      wave_sync();
      { double* t = x; x = y; y = t; }
      for (int pass = 0; pass < 2; pass++) {
        int sprod = 1, cur = M;
        for (int stg = 0; stg < P.nstagesM; stg++) {
          const int r = P.radixM[stg];
          dft_stage(M, P.twM, r, sprod, cur, x, y, lane, SPX_WAVE);
          wave_sync();
          double* t = x; x = y; y = t;
          sprod *= r;
          cur /= r;
        }
        if (pass == 0) {
          // x holds A: keep A[0] in slot M (free: the buffers have W slots), then conj(A .* F(b)) in place
          if (lane == 0) *reinterpret_cast<double2*>(x + 2 * M) = *reinterpret_cast<const double2*>(x);
          wave_sync();
          for (int k = lane; k < M; k += SPX_WAVE) {
            const cplx a = ld(x, k);
            const double2 b = *reinterpret_cast<const double2*>(P.bfft + 2 * k);
            const cplx cc = spx_cmul(a, b.x, b.y);
            *reinterpret_cast<double2*>(x + 2 * k) = make_double2(cc.r, -cc.i);
          }
          wave_sync();
        }
      }
      // x holds F(conj(C)), slot M of the buffer that held A holds A[0]: with an even number of stages that is x
      // itself, otherwise y -- both cases: the A buffer is the one the second transform started from
      const double inv = 1.0 / (double)M;
      const double* abuf = (P.nstagesM & 1) ? y : x;
      const cplx A0 = ld(abuf, M);
      wave_sync();
      // scatter into the other buffer (the transform result must stay readable while it is permuted)
      for (int q = lane; q < M; q += SPX_WAVE) {
        const cplx c = ld(x, q);
        const double cr = c.r * inv, ci = -c.i * inv;
        *reinterpret_cast<double2*>(y + 2 * P.iperm[q]) = make_double2(x0.r + cr, x0.i + ci);
      }
      if (lane == 0) *reinterpret_cast<double2*>(y) = make_double2(x0.r + A0.r, x0.i + A0.i);
      wave_sync();
      { double* t = x; x = y; y = t; }
    } else {
      int sprod = 1, cur = W;
      for (int stg = 0; stg < P.nstages; stg++) {
        const int r = P.radix[stg];
        dft_stage(W, ltw, r, sprod, cur, x, y, lane, SPX_WAVE);
        wave_sync();
        double* t = x; x = y; y = t;
        sprod *= r;
        cur /= r;
      }
    }
    ASTAMP(1);
    // untangle the packed transform:  X[k] = E[k] + e^{-2 pi i k/N} O[k]
    float* spec_out = taps.spectrogram ? taps.spectrogram + (size_t)(S.frame_off + j) * N : nullptr;
    for (int k = lane; k < W; k += SPX_WAVE) {
      const int k2 = (k == 0) ? 0 : W - k;
      cplx a = ld(x, k), bb = ld(x, k2);
      const double b_r = bb.r, b_i = -bb.i;
      const float mag = spx_untangle_mag(a.r, a.i, b_r, b_i, *reinterpret_cast<const double2*>(ltw2 + 2 * k));
      mrow[k] = mag;
      if (spec_out) {
        spec_out[k] = mag;
        if (k > 0) spec_out[N - k] = mag;
        else spec_out[W] = (float)__builtin_fabs(a.r - a.i);
      }
    }
    wave_sync();
    ASTAMP(2);
  }
  ASTAMP(3);
  __syncthreads();
  ASTAMP(4);

  // ---------------- phase 2: per-slot energy (float, index order), max, inverse norm ----------------
  // Wave 0, one lane per slot: the float sum in the reference's order (speedy.c:513-516), one dependent add per bin -- a
  // chain the whole tile waits for, so it carries nothing else: four bins per LDS read (16-byte rows), sixteen bins' reads
  // in flight ahead of the adds, the squares as packed multiplications, and the row maxima (speedy.c:709; a maximum does
  // not depend on the order) by the other three waves meanwhile.  (Rounds 1-3: load, multiply, add, maximum and a register
  // copy per bin on the chain's wave: 23-40 cycles per bin, 10-27 % of a tile's time.)
  if (wave > 0) {
#ifdef SPX_ROWMAX_V1
    for (int s = wave - 1; s <= TF; s += 3) {
      const float* mrow = mags + (size_t)s * MS;
      float mx = 0.0f;
      for (int i = 1 + lane; i < W; i += SPX_WAVE) mx = fmaxf(mx, mrow[i]);
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
      if (lane == 0) fThr[s] = (float)((double)mx / 100.0);                      // speedy.c:709
    }
#else
    // Round 5.  The magnitudes are non-negative floats, so their maximum is the maximum of their bit patterns as unsigned integers:
    // four bins per 16-byte LDS read, v_max_u32 without the canonicalising copies fmaxf needs, the wave's maximum by a DPP
    // reduction (six instructions; __shfl_xor was six LDS round trips), and the division by 100 -- an fp64 IEEE sequence -- ONCE for
    // all the rows of the wave (lane j keeps the maximum of the wave's j-th row) instead of once per row on lane 0.  These three
    // waves, not the energy chain on wave 0, were what phase 2 waited for.
    unsigned keep = 0u;   // lane j: bit pattern of the maximum of row wave - 1 + 3 j
    int nrows = 0;
    for (int s = wave - 1; s <= TF; s += 3, nrows++) {
      const unsigned* mrow = reinterpret_cast<const unsigned*>(mags + (size_t)s * MS);
      unsigned m = 0u;
      for (int q = lane; 4 * q < W; q += SPX_WAVE) {
        const uint4 v = *reinterpret_cast<const uint4*>(mrow + 4 * q);
        const unsigned a = (q == 0) ? 0u : v.x;                                  // bin 0 is not part of the maximum (speedy.c:709)
        const unsigned b = (4 * q + 1 < W) ? v.y : 0u, c = (4 * q + 2 < W) ? v.z : 0u, d = (4 * q + 3 < W) ? v.w : 0u;
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
        const unsigned abcd = ab > cd ? ab : cd;
        m = m > abcd ? m : abcd;
      }
      const float mx = wave_max_f(__builtin_bit_cast(float, m));
      if (lane == nrows) keep = __builtin_bit_cast(unsigned, mx);
    }
    if (lane < nrows) fThr[wave - 1 + 3 * lane] = (float)((double)__builtin_bit_cast(float, keep) / 100.0);   // speedy.c:709
#endif
#ifndef SPX_LOG_V1
    // ... and bring the table of log spec v2 into the work area behind the terms (the transform buffers are free by now; the
    // barrier at the end of this phase publishes it): 128 entries of 16 bytes, one per lane of waves 1 and 2
    if (tid < SPX_WAVE + 128)
      reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(work) + terms_bytes(W, TF, WCT != 0))[tid - SPX_WAVE] =
          reinterpret_cast<const uint4*>(spx_log_table_dev)[tid - SPX_WAVE];
#endif
  } else if (tid <= TF) {
    const float4* r4 = reinterpret_cast<const float4*>(mags + (size_t)tid * MS);
    float e = 0.0f;
    __builtin_amdgcn_s_setprio(3);
    auto acc8 = [&](const float4& p, const float4& q) {
      const float4 pp = {p.x * p.x, p.y * p.y, p.z * p.z, p.w * p.w}, qq = {q.x * q.x, q.y * q.y, q.z * q.z, q.w * q.w};
      e += pp.x; e += pp.y; e += pp.z; e += pp.w;
      e += qq.x; e += qq.y; e += qq.z; e += qq.w;
    };
    {
      const float4 h = r4[0];   // bins 0 (not part of the sum) .. 3
      if (1 < W) e += h.y * h.y;
      if (2 < W) e += h.z * h.z;
      if (3 < W) e += h.w * h.w;
    }
    // sixteen bins per round, the next round's four reads issued before this round's adds (they run up to 64 bytes past
    // a row's end: the next row, or the arrays behind the magnitudes)
    int i = 4;
    float4 c0 = r4[1], c1 = r4[2], c2 = r4[3], c3 = r4[4];
#pragma unroll 1
    for (; i + 16 <= W; i += 16) {
      const float4 n0 = r4[(i >> 2) + 4], n1 = r4[(i >> 2) + 5], n2 = r4[(i >> 2) + 6], n3 = r4[(i >> 2) + 7];
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks some of the reads below the adds and the chain
                                           // pays their round trip every round)
      acc8(c0, c1);
      acc8(c2, c3);
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    {
      const float t[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (i + u < W) e += t[u] * t[u];
    }
    __builtin_amdgcn_s_setprio(0);
    const float eps = 2.2204e-16f;
    fE[tid] = e;
    fInv[tid] = (float)(1.0 / (__builtin_sqrt((double)e) + (double)eps));      // speedy.c:642
  }
  __syncthreads();
  ASTAMP(5);

  const int nfr = j1 - j0;
#ifndef SPX_LOG_V1
  const SpxLogEntry* ltab = reinterpret_cast<const SpxLogEntry*>(reinterpret_cast<const unsigned char*>(work) + terms_bytes(W, TF, WCT != 0));
  // log spec v2 for a float quotient; anything but a positive normal float (only explicit float frames of the unit-level API can
  // produce one) takes v1, as in the oracle
  auto log_of_ratio = [&](float ratio) -> double {
    double v = spx_log_v2_f32(ratio, ltab);
    if (__builtin_expect(!spx_log_v2_domain(ratio), 0)) v = spx_log_v1((double)ratio);
    return v;
  };
#else
  auto log_of_ratio = [&](float ratio) -> double { return spx_log_v1((double)ratio); };
#endif
  // gated |log ratio| term of (frame f of the tile, bin i): speedy.c:705-717
  auto log_term = [&](int f, int i) -> double {
    const int sl = f + 1;
    const float cur = mags[(size_t)sl * MS + i], last = mags[(size_t)(sl - 1) * MS + i];
    const float thr = fThr[sl];
    double term = 0.0;
    if (cur > thr && last > thr) {
      const float eps = 2.2204e-16f;
      const float nc = cur * fInv[sl], nl = last * fInv[sl - 1];
      const float ratio = spx_fdiv32(nc + eps, nl + eps);
      term = __builtin_fabs(log_of_ratio(ratio));                              // speedy.c:715-717
    }
    return term;
  };
  auto write_rec = [&](int f, float lsd) {
    const float lowthr = (float)(0.04 * (double)1.41421f);                     // speedy.c:682
    const float e = fE[f + 1];
    SpxFrameRec r;
    r.energy = e;
    r.lsd = (e <= lowthr) ? 0.0f : lsd;
    rec[S.frame_off + j0 + f] = r;
  };
  if (taps.normalized) {
    // normalised spectrum of frame j is the one used for tension k = j+1 (speedy.c:673-675)
    for (int idx = tid; idx < nfr * W; idx += SPX_BLOCK) {
      const int f = idx / W, i = idx - f * W;
      const int j = j0 + f;
      const int row = j + (S.unit_time0 ? 0 : 1);  // the tension frame whose `cur` spectrum frame j is
      if (row < T + (S.unit_time0 ? 1 : 0))
        taps.normalized[(size_t)(S.frame_off + row) * W + i] = mags[(size_t)(f + 1) * MS + i] * fInv[f + 1];
    }
    if (S.unit_time0) {
      // unit-level time base: tension t uses frame t itself, there is no all-zero row
    } else if (j0 == 0) {
      for (int i = tid; i < W; i += SPX_BLOCK) taps.normalized[(size_t)S.frame_off * W + i] = 0.0f;
    } else if (j0 == S.frame_begin) {
      // a resumed stream: the launch that analysed frame j0-1 stopped before row j0; the halo slot holds it
      for (int i = tid; i < W; i += SPX_BLOCK)
        taps.normalized[(size_t)(S.frame_off + j0) * W + i] = mags[i] * fInv[0];
    }
  }
  if constexpr (WCT != 0) {
    // ---------------- phases 3 + 4, pipelined: waves 1..3 compute the terms of a block of SPX_CB bins (all frames)
    // while wave 0 -- one lane per frame, bin order, the float accumulation of speedy.c:715 -- sums the previous block.
    // Two blocks in flight in the (now free) transform buffers; one workgroup barrier per block. ----------------
    constexpr int CB = spx_cb(TF);
    constexpr int CBS = CB + 1;
    static_assert(TF * CB == 2 * (SPX_BLOCK - SPX_WAVE), "two terms per lane of waves 1..3 and block");
    double* tblk = work;  // [2][TF][CBS]
    constexpr int NBLK = (WCT - 1 + CB - 1) / CB;
    float lsd = 0.0f;
    // What a term lane needs of its two frames is the same for every block: the two rows of magnitudes, the frame's threshold and
    // the two inverse norms (round 5: they were re-derived, and the three scalars re-read from LDS, for every term -- a fifth of
    // the term's instructions; -DSPX_TERMS_V1: that code).
    int fi[2], ci[2];
    bool ok[2];
#ifndef SPX_TERMS_V1
    const float* rowc[2];
    const float* rowl[2];
    float thrv[2], invc[2], invl[2];
#endif
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int it = tid - SPX_WAVE + t * (SPX_BLOCK - SPX_WAVE);
      fi[t] = wave > 0 ? it / CB : 0;
      ci[t] = wave > 0 ? it - fi[t] * CB : 0;
      ok[t] = wave > 0 && fi[t] < nfr;                                         // (bins past the window: a term of 0, see the chain)
#ifndef SPX_TERMS_V1
      const int fl = ok[t] ? fi[t] : 0;
      rowc[t] = mags + (size_t)(fl + 1) * MS;
      rowl[t] = mags + (size_t)fl * MS;
      thrv[t] = fThr[fl + 1];
      invc[t] = fInv[fl + 1];
      invl[t] = fInv[fl];
#endif
    }
    for (int k = 0; k <= NBLK; k++) {
      if (wave > 0) {
        if (k < NBLK) {
          double* tb = tblk + (size_t)(k & 1) * TF * CBS;
          // the lane's two terms, their straight-line halves first (spx_log.h) so that the two interleave
          bool gate[2];
          float xr[2];
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int i = 1 + k * CB + ci[t];
            const bool in = ok[t] && i < WCT;
#ifdef SPX_TERMS_V1
            const int fl = in ? fi[t] : 0, il = in ? i : 1;
            const float cur = mags[(size_t)(fl + 1) * MS + il], last = mags[(size_t)fl * MS + il];
            const float thr = fThr[fl + 1];
            const float fic = fInv[fl + 1], fil = fInv[fl];
#else
            const int il = (i < WCT) ? i : 1;
            const float cur = rowc[t][il], last = rowl[t][il];
            const float thr = thrv[t], fic = invc[t], fil = invl[t];
#endif
            gate[t] = in && cur > thr && last > thr;                           // speedy.c:705-717
            const float eps = 2.2204e-16f;
            const float nc = cur * fic, nl = last * fil;
            // (both operands in [2.2e-16, ~1e3] whenever the gate is open: spx_fdiv32's range; a closed gate's quotient is not used)
            const float ratio = spx_fdiv32(nc + eps, nl + eps);
            xr[t] = gate[t] ? ratio : 2.0f;
          }
#ifdef SPX_LOG_V1
          const spx_log_parts p0 = spx_log_main((double)xr[0]), p1 = spx_log_main((double)xr[1]);
          const double t0 = gate[0] ? __builtin_fabs(spx_log_finish(p0, (double)xr[0])) : 0.0;
          const double t1 = gate[1] ? __builtin_fabs(spx_log_finish(p1, (double)xr[1])) : 0.0;
#else
          // (a closed gate's argument is 2.0f: inside the domain, its value unused)
          const double l0 = log_of_ratio(xr[0]), l1 = log_of_ratio(xr[1]);
          const double t0 = gate[0] ? __builtin_fabs(l0) : 0.0;
          const double t1 = gate[1] ? __builtin_fabs(l1) : 0.0;
#endif
          if (ok[0]) tb[fi[0] * CBS + ci[0]] = t0;
          if (ok[1]) tb[fi[1] * CBS + ci[1]] = t1;
        }
      } else if (k > 0 && tid < nfr) {
        const double* tb = tblk + (size_t)((k - 1) & 1) * TF * CBS + tid * CBS;
        const int i0 = 1 + (k - 1) * CB;
        // the order-bound chain (three dependent conversions / additions per term) is what a tile waits for: its wave goes
        // first whenever it can issue (the SIMD is shared with other workgroups' transform and log waves)
        __builtin_amdgcn_s_setprio(3);
        // eight terms' loads in flight, then their eight chain steps: a term's LDS round trip is not on the chain.  The last
        // block's bins past the window hold 0 -- (float)((double)lsd + 0.0) is lsd -- so no step is conditional.
        (void)i0;
        static_assert(CB % 8 == 0, "batches of eight");
        double tv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) tv[u] = tb[u];
#pragma unroll
        for (int c = 0; c < CB; c += 8) {
          double nx[8];
          if (c + 8 < CB) {
#pragma unroll
            for (int u = 0; u < 8; u++) nx[u] = tb[c + 8 + u];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) lsd = (float)((double)lsd + tv[u]);        // speedy.c:715 (float +=)
          if (c + 8 < CB) {
#pragma unroll
            for (int u = 0; u < 8; u++) tv[u] = nx[u];
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
      ASTAMP(6);   // wave 0: its chain; (7): the wait for the waves that compute the next block's terms
      __syncthreads();
      ASTAMP(7);
    }
    if (tid < nfr) write_rec(tid, lsd);
  } else {
  // ---------------- phase 3: gated |log ratio| terms, one lane per (slot, bin) ----------------
  double* terms = work;  // aliases the DFT buffers, [TF][W+1]
  for (int idx = tid; idx < nfr * (W - 1); idx += SPX_BLOCK) {
    const int f = idx / (W - 1);
    const int i = 1 + (idx - f * (W - 1));
    terms[(size_t)f * (W + 1) + i] = log_term(f, i);
  }
  __syncthreads();
  ASTAMP(6);

  // ---------------- phase 4: float accumulation of the terms in bin order, one lane per frame --------
  if (tid < nfr) {
    const double* trow = terms + (size_t)tid * (W + 1);
    float lsd = 0.0f;
    int i = 1;
    for (; i + 8 <= W; i += 8) {   // eight loads in flight per eight chain steps
      double tv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) tv[u] = trow[i + u];
#pragma unroll
      for (int u = 0; u < 8; u++) lsd = (float)((double)lsd + tv[u]);          // speedy.c:715 (float +=)
    }
    for (; i < W; i++) lsd = (float)((double)lsd + trow[i]);
    write_rec(tid, lsd);
  }
  }
  if (tile_flags) {
    // publish the tile's records to the concurrently running walk kernel (cdna_hip_programming.md Guideline 16):
    // every storing wave drains its stores, workgroup barrier, one agent-scope release, then the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&tile_flags[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  ASTAMP(7);
  ASTAMP_FLUSH
}

// Diagnostic (tests/test_gpu_parity.py): the scale-free division and square-root sequences of spx_log.h against the
// compiler's IEEE sequences, bit for bit, on pseudo-random operands of the ranges the analysis kernel feeds them (and, for
// the square root, on zero / tiny / huge arguments, which must take the library sequence).  Returns the number of mismatches.
__global__ void spx_arith_check_kernel(unsigned seed, unsigned per_thread, unsigned long long* bad) {
  unsigned long long st = (unsigned long long)seed * 0x9E3779B97F4A7C15ull + (blockIdx.x * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull + 1;
  auto next = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 0x2545F4914F6CDD1Dull; };
  unsigned long long miss = 0;
  for (unsigned i = 0; i < per_thread; i++) {
    const unsigned long long a = next(), b = next(), c = next();
    // float operands 2^u (1 + m), u in [-53, 3]
    const float fn = __uint_as_float((unsigned)((127 - 53 + (a % 57)) << 23) | (unsigned)((a >> 8) & 0x7fffff));
    const float fd = __uint_as_float((unsigned)((127 - 53 + (b % 57)) << 23) | (unsigned)((b >> 8) & 0x7fffff));
    if (__float_as_uint(spx_fdiv32(fn, fd)) != __float_as_uint(fn / fd)) miss++;
    // f / (2 + f), |f| = 2^u (1 + m), u in [-21, -2]: what spx_log_main divides
    const long long fb = ((long long)(1023 - 21 + (c % 20)) << 52) | (long long)((c >> 8) & 0xfffffffffffffll) | ((c >> 63) ? (1ll << 63) : 0);
    const double f = __longlong_as_double(fb);
    if (__double_as_longlong(spx_fdiv64(f, 2.0 + f)) != __double_as_longlong(f / (2.0 + f))) miss++;
    // square roots: 2^u (1 + m), u in [-800, 1010] (both ends beyond the fast sequence's range), and zero now and then
    const long long xb = ((long long)(1023 - 800 + (a >> 33) % 1811) << 52) | (long long)((b >> 10) & 0xfffffffffffffll);
    const double x = ((c & 1023) == 0) ? 0.0 : __longlong_as_double(xb);
    if (__float_as_uint(spx_sqrt64_to_f32(x)) != __float_as_uint((float)__builtin_sqrt(x))) miss++;
  }
  if (miss) atomicAdd(bad, miss);
}
extern "C" long long spx_debug_arith_check(unsigned seed, unsigned threads, unsigned per_thread) {
  unsigned long long* d = nullptr;
  unsigned long long h = 0;
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return -1;
  if (hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return -1; }
  const unsigned blocks = (threads + 255) / 256;
  hipLaunchKernelGGL(spx_arith_check_kernel, dim3(blocks), dim3(256), 0, nullptr, seed, per_thread, d);
  const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  return ok ? (long long)h : -1;
}

// Diagnostic (tests/test_gpu_parity.py): log spec v2 over EVERY positive normal float.  Workgroup b takes the 2^20 float patterns
// b << 20 .. and leaves the sum of the results' bit patterns (mod 2^64) in sums[b]; oracle/orc_logcheck.c computes the same sums
// with the oracle's orc_log_v2_f32 -- equal sums block for block = the GPU's log is the oracle's on the whole domain.  The table
// is read from LDS, as the analysis kernel reads it.
__global__ void __launch_bounds__(256) spx_log_check_kernel(unsigned first_block, unsigned long long* __restrict__ sums) {
#ifndef SPX_LOG_V1
  __shared__ SpxLogEntry tab[128];
  __shared__ unsigned long long acc;
  if (threadIdx.x < 128) reinterpret_cast<uint4*>(tab)[threadIdx.x] = reinterpret_cast<const uint4*>(spx_log_table_dev)[threadIdx.x];
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  const unsigned b = first_block + blockIdx.x;
  unsigned long long sum = 0;
  for (unsigned i = threadIdx.x; i < (1u << 20); i += 256) {
    const float x = __uint_as_float((b << 20) | i);
    sum += (unsigned long long)__double_as_longlong(spx_log_v2_f32(x, tab));
  }
  atomicAdd(&acc, sum);
  __syncthreads();
  if (threadIdx.x == 0) sums[b] = acc;
#endif
}
// sums: HOST uint64[2040] indexed by block (blocks 8 .. 2039 = the positive normal floats); returns 0, -1 on a runtime error,
// -3 in a build with log spec v1 (-DSPX_LOG_V1)
extern "C" int spx_debug_log_check(unsigned first_block, unsigned end_block, unsigned long long* sums) {
#ifdef SPX_LOG_V1
  (void)first_block; (void)end_block; (void)sums;
  return -3;
#else
  if (first_block < 8) first_block = 8;
  if (end_block > 2040) end_block = 2040;
  if (!sums || first_block >= end_block) return -1;
  unsigned long long* d = nullptr;
  if (hipMalloc(&d, 2040 * sizeof(unsigned long long)) != hipSuccess) return -1;
  bool ok = hipMemset(d, 0, 2040 * sizeof(unsigned long long)) == hipSuccess;
  if (ok) hipLaunchKernelGGL(spx_log_check_kernel, dim3(end_block - first_block), dim3(256), 0, nullptr, first_block, d);
  ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(sums, d, 2040 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  return ok ? 0 : -1;
#endif
}

void spx_launch_analysis(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, int n_tiles,
                         const int16_t* in, SpxFrameRec* rec, SpxTapsDev taps, const int* tile_order, int* tile_flags,
                         hipStream_t st) {
  if (n_tiles <= 0) return;
  const size_t lds = spx_analysis_lds_bytes(P);
  const int ctw = plan_ct_window(P);
#define SPX_LAUNCH_ANALYSIS(TFV, WV)                                                                                   \
  hipLaunchKernelGGL((spx_analysis_kernel<TFV, WV>), dim3(n_tiles), dim3(SPX_BLOCK), lds, st, P, streams, n_streams, in, \
                     rec, taps, tile_order, tile_flags, (const float*)nullptr, 0)
  if (P.tile_frames == SPX_TF_TINY) {
    SPX_LAUNCH_ANALYSIS(SPX_TF_TINY, 0);
  } else if (P.tile_frames == SPX_TF_SMALL) {
    if (ctw == 240) SPX_LAUNCH_ANALYSIS(SPX_TF_SMALL, 240); else if (ctw == 330) SPX_LAUNCH_ANALYSIS(SPX_TF_SMALL, 330);
    else if (ctw == 720) SPX_LAUNCH_ANALYSIS(SPX_TF_SMALL, 720); else if (ctw == 661) SPX_LAUNCH_ANALYSIS(SPX_TF_SMALL, 661);
    else if (ctw == 120) SPX_LAUNCH_ANALYSIS(SPX_TF_SMALL, 120);
    else SPX_LAUNCH_ANALYSIS(SPX_TF_SMALL, 0);
  } else {
    if (ctw == 240) SPX_LAUNCH_ANALYSIS(SPX_TF, 240); else if (ctw == 330) SPX_LAUNCH_ANALYSIS(SPX_TF, 330);
    else if (ctw == 120) SPX_LAUNCH_ANALYSIS(SPX_TF, 120); else if (ctw == 360) SPX_LAUNCH_ANALYSIS(SPX_TF, 360);
    else if (ctw == 180) SPX_LAUNCH_ANALYSIS(SPX_TF, 180);
    else if (ctw == 480) SPX_LAUNCH_ANALYSIS(SPX_TF, 480);
    else SPX_LAUNCH_ANALYSIS(SPX_TF, 0);
  }
#undef SPX_LAUNCH_ANALYSIS
}

int spx_analysis_vgprs(const SpxPlanDev& P, int* scratch_bytes) {
  const int ctw = plan_ct_window(P);
  const bool small = P.tile_frames == SPX_TF_SMALL;
  const void* fn;
#define SPX_AN_FN(TFV, WV) reinterpret_cast<const void*>(spx_analysis_kernel<TFV, WV>)
  if (P.tile_frames == SPX_TF_TINY) fn = SPX_AN_FN(SPX_TF_TINY, 0);
  else if (small) fn = ctw == 240 ? SPX_AN_FN(SPX_TF_SMALL, 240) : ctw == 330 ? SPX_AN_FN(SPX_TF_SMALL, 330)
                     : ctw == 720 ? SPX_AN_FN(SPX_TF_SMALL, 720) : ctw == 661 ? SPX_AN_FN(SPX_TF_SMALL, 661)
                     : ctw == 120 ? SPX_AN_FN(SPX_TF_SMALL, 120) : SPX_AN_FN(SPX_TF_SMALL, 0);
  else fn = ctw == 240 ? SPX_AN_FN(SPX_TF, 240) : ctw == 330 ? SPX_AN_FN(SPX_TF, 330) : ctw == 120 ? SPX_AN_FN(SPX_TF, 120)
          : ctw == 360 ? SPX_AN_FN(SPX_TF, 360) : ctw == 480 ? SPX_AN_FN(SPX_TF, 480) : ctw == 180 ? SPX_AN_FN(SPX_TF, 180)
          : SPX_AN_FN(SPX_TF, 0);
#undef SPX_AN_FN
  return spx_kernel_vgprs(fn, scratch_bytes);
}

void spx_launch_analysis_frames(const SpxPlanDev& P, const SpxStreamDev* streams, int n_tiles, const float* frames,
                                bool preemph, SpxFrameRec* rec, SpxTapsDev taps, hipStream_t st) {
  if (n_tiles <= 0) return;
  SpxPlanDev Q = P;
  if (Q.tile_frames != SPX_TF_SMALL && Q.tile_frames != SPX_TF_TINY) Q.tile_frames = SPX_TF;   // (smaller tiles above about 49 / 61 kHz)
  const size_t lds = analysis_lds_bytes(Q, false);  // the plan-driven instantiation
  if (Q.tile_frames == SPX_TF_TINY)
    hipLaunchKernelGGL((spx_analysis_kernel<SPX_TF_TINY, 0>), dim3(n_tiles), dim3(SPX_BLOCK), lds, st, Q, streams, 1,
                       (const int16_t*)nullptr, rec, taps, (const int*)nullptr, (int*)nullptr, frames, preemph ? 1 : 2);
  else if (Q.tile_frames == SPX_TF_SMALL)
    hipLaunchKernelGGL((spx_analysis_kernel<SPX_TF_SMALL, 0>), dim3(n_tiles), dim3(SPX_BLOCK), lds, st, Q, streams, 1,
                       (const int16_t*)nullptr, rec, taps, (const int*)nullptr, (int*)nullptr, frames, preemph ? 1 : 2);
  else
    hipLaunchKernelGGL((spx_analysis_kernel<SPX_TF, 0>), dim3(n_tiles), dim3(SPX_BLOCK), lds, st, Q, streams, 1,
                       (const int16_t*)nullptr, rec, taps, (const int*)nullptr, (int*)nullptr, frames, preemph ? 1 : 2);
}
