// Walk kernel for gfx950: one workgroup of NW wavefronts (NW = 1, 2, 4, 8 or 16) per stream.
//
// The sample-rate stage, inherently sequential per stream (each step's position depends on the last period):
//   a10 AMDF pitch search on the decimated then the full-rate signal   (libsonic)
//   a11 skip / insert pitch periods with a linear cross-fade, FIFO bookkeeping, flush padding
//       driven exactly as the shim drives it: one (setSpeed, write frameStep samples) pair per tension
//       frame (soniclib.c:354,369), the un-analysed tail at the last speed (soniclib.c:538-550), then
//       sonicIntFlushStream (soniclib.c:551).
// The speed of every tension frame comes from the tension kernel (spx_tension.hip) through scratch[4k + 3]; in
// concurrent mode that kernel runs beside this one and publishes how many speeds are final (`speed_ready`).
// Two variants: FAST (all streams mono and speeding up, rates below 32 kHz; the bench and the usual case) and the
// general one (any channel count, slow-down, any rate).
//
// How a pitch step maps to the hardware:
//   * the input lives in an LDS sliding window (int16, refilled with coalesced loads every ~25 steps), so a
//     step touches HBM only to store its 2n output bytes; the stores are never waited for (the workgroup
//     barrier used here drains lgkmcnt only);
//   * the search signals are kept in LDS as u16 biased by 32768, twice (once shifted by a sample), so that
//     any lag reads two aligned sample PAIRS per ds_read_b32 and one v_sad_u16 adds two |a-b| terms;
//   * the lags x sample-pairs space is spread over all lanes; partial sums meet in LDS with ds_add_u32;
//   * arg-min / arg-max of diff/lag: float ratios, DPP wave reduction, then an exact integer resolve of
//     the few lanes within 2^-16 of the extremum on the scalar unit (ties -> smallest lag, as a sequential
//     scan would);  every wave does this redundantly, so no broadcast barrier is needed.
// All sample arithmetic is integer; results are bit-exact against oracle/orc_sonic.c.
#include <atomic>
#include <stdlib.h>

#include "spx_internal.h"

#define SPX_CH 1024  // frames per prologue chunk held in LDS
#define SPX_WCH 64   // frames in the first walk chunk when the analysis kernel runs concurrently (multiple of the tile)

#include "spx_walk_common.h"

// Diagnostic build only (-DSPX_STAMPS): per-phase shader-cycle sums of workgroup 0, lane 0.  Never in the product.
#ifdef SPX_STAMPS
__device__ unsigned long long g_spx_stamps[32];
#ifndef SPX_STAMP_SEL
#define SPX_STAMP_SEL 0
#endif
// One region per build (SPX_STAMP_SEL): a single accumulator keeps the register pressure of the measured kernel
// close to the product's.  Slot 30 = whole kernel, slot 31 = pitch steps.
#define STAMP_DECL                                         \
  X.stamp_acc = 0; X.stamp_steps = 0;                      \
  X.stamp_last = __builtin_readcyclecounter();             \
  X.stamp_t0 = X.stamp_last;
#define STAMP(i)                                                          \
  do {                                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter();           \
    if ((i) == SPX_STAMP_SEL) X.stamp_acc += t_ - X.stamp_last;           \
    if ((i) == 1) X.stamp_steps++;                                        \
    X.stamp_last = t_;                                                    \
  } while (0)
#define STAMP_FLUSH                                                                              \
  if (threadIdx.x == 0 && blockIdx.x == 0) {                                                     \
    g_spx_stamps[SPX_STAMP_SEL] += X.stamp_acc;                                                  \
    g_spx_stamps[30] += __builtin_readcyclecounter() - X.stamp_t0;                               \
    g_spx_stamps[31] += X.stamp_steps;                                                           \
  }
extern "C" void spx_debug_stamps(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spx_stamps), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spx_stamps), z, sizeof(z));
  }
}
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

struct WalkCtx {
  int* flush_rem;     // SPX_F_NO_TRUNC: where the flush leaves {frames the stage held, frames produced so far} (nullptr: truncate as usual)
  const int16_t* in;  // stream input (interleaved)
  int16_t* out;       // stream output
  pos_t out_cap;
  pos_t limit;      // absolute frame index from which reads return 0 (end of input / flush padding)
  int C;
  // LDS sliding window over frames [wbase, wbase + wcap)
  unsigned short* monoH;   // mono mix biased by 32768: monoH[k] = mono(wbase + k) + 32768
  unsigned short* monoHB;  // the same shifted by one frame: monoHB[k] = monoH[k + 1]
  short* raw;              // interleaved raw samples, only kept when C > 1 (the cross-fade is per channel)
  pos_t wbase;           // -1 = window invalid
  int wcap;
  unsigned short* dnH;     // biased decimated signal of the current step (and its shifted copy)
  unsigned short* dnHB;
  // FAST kernels: the decimated signal of EVERY window position, built once per refill.  Plane r (r < skip) holds
  // S[m*skip + r], S[i] = mean of window samples i .. i+skip-1, so the decimated signal of a step at window offset o
  // is plane (o % skip) from element o / skip on, contiguous.  plB is the copy shifted by one element.
  unsigned short* pl;
  unsigned short* plB;
  int plStride;            // elements per plane (even)
  int skip, skipM;         // skipM = ceil(2^16 / skip): i / skip == (i * skipM) >> 16 for i < 8192
  unsigned skipM32;        // ceil(2^32 / skip): x / skip == mulhi(x, skipM32) for x < 2^18
  unsigned* diffC;    // per-lag AMDF sums, coarse search
  unsigned* diffR;    // per-lag AMDF sums, refine search
#ifdef SPX_STAMPS
  unsigned long long stamp_last, stamp_acc, stamp_t0;
  unsigned stamp_steps;
#endif
};

__device__ __forceinline__ int global_sample(const WalkCtx& X, pos_t a, int c) {
  return (a < X.limit) ? (int)X.in[(size_t)a * X.C + c] : 0;
}
// sample through the LDS window when it covers frame a, else from HBM
__device__ __forceinline__ int any_sample(const WalkCtx& X, pos_t a, int c) {
  const pos_t o = a - X.wbase;
  if (X.wbase >= 0 && o >= 0 && o < X.wcap) {
    if (X.C == 1) return (int)X.monoH[o] - 32768;
    return (int)X.raw[o * X.C + c];
  }
  return global_sample(X, a, c);
}

// Make the window cover [pos, pos + need).  Uniform across the workgroup.  The biased mono signal and its
// shifted copy are built here, once per refill, so a pitch step never touches HBM for its input.
template <int NW, int FAST>
__device__ __forceinline__ void ensure_window(WalkCtx& X, pos_t pos, int need) {
  constexpr int NT = 64 * NW;
  if (X.wbase >= 0 && pos >= X.wbase && pos + need <= X.wbase + X.wcap) return;
  lds_sync<NW>();  // everyone is done reading the old window
  const pos_t nb = pos & ~7;
  const int C = (FAST == 1) ? 1 : X.C;
  if (C == 1) {
    const int16_t* __restrict__ src = X.in + nb;
    const pos_t room = X.limit - nb;  // frames of real input from nb on
    for (int k0 = threadIdx.x; k0 < X.wcap + 1; k0 += 8 * NT) {
      int v[8];
      const int last = (int)(room < X.wcap + 1 ? room : X.wcap + 1) - 1;  // last index holding real input
      if (last >= 0) {  // uniform
#pragma unroll
        for (int u = 0; u < 8; u++) {  // eight coalesced loads in flight before the first LDS write:
          const int k = k0 + u * NT;   // clamped address, unconditional load, so nothing serialises them
          v[u] = (int)src[k < last ? k : last];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int k = k0 + u * NT;
          if (k > last) v[u] = 0;
        }
      } else {  // the whole window lies in the zero padding: no address there may be touched
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = 0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 + u * NT;
        const unsigned short w = (unsigned short)(v[u] + 32768);
        if (k < X.wcap) X.monoH[k] = w;
        if (k > 0 && k < X.wcap + 1) X.monoHB[k - 1] = w;
      }
    }
  } else {
    for (int k = threadIdx.x; k < X.wcap + 1; k += NT) {
      const pos_t g = nb + k;
      int sum = 0;
      for (int c = 0; c < C; c++) {
        const int v = (g < X.limit) ? (int)X.in[(size_t)g * C + c] : 0;
        if (k < X.wcap) X.raw[(size_t)k * C + c] = (short)v;
        sum += v;
      }
      const unsigned short u = (unsigned short)(sum / C + 32768);
      if (k < X.wcap) X.monoH[k] = u;
      if (k > 0) X.monoHB[k - 1] = u;
    }
  }
  X.wbase = nb;
  lds_sync<NW>();
  if (FAST == 1 || (FAST == 2 && C == 1)) {
    // One thread per decimated index m: it reads the 2*skip-1 window samples m*skip .. m*skip+2*skip-2 once and slides
    // the sum over them, giving element m of every plane.  |sum| < 2^18 and skip <= 7, so the truncating division is
    // exactly mulhi(|sum|, ceil(2^32 / skip)).
    const int skip = X.skip;
    const unsigned M = X.skipM32;
    const int bias = 32768 * skip;
    for (int m = threadIdx.x; (m + 1) * skip <= X.wcap; m += NT) {
      const unsigned short* w = X.monoH + m * skip;
      int sum = 0;
      for (int j = 0; j < skip; j++) sum += (int)w[j];
      for (int r = 0; r < skip; r++) {
        if ((m + 1) * skip + r > X.wcap) break;  // the last element of the higher planes needs samples past the window
        const int v = sum - bias;
        const unsigned mag = (unsigned)(v < 0 ? -v : v);
        const int qm = (int)__umulhi(mag, M);
        const unsigned short u = (unsigned short)((v < 0 ? -qm : qm) + 32768);
        X.pl[r * X.plStride + m] = u;
        if (m > 0) X.plB[r * X.plStride + m - 1] = u;
        sum += (int)w[skip + r] - (int)w[r];
      }
    }
  } else if (FAST == 2) {
    // Several channels: the decimated sample is the sum of skip*C RAW samples divided ONCE by skip*C (the dependency's
    // downSampleInput), not the mean of the per-frame channel means.  |sum| < 2^21, skip*C <= 56: mulhi is still exact.
    const int skip = X.skip, div = skip * C;
    const unsigned M = (unsigned)((0x100000000ull + (unsigned)div - 1) / (unsigned)div);
    for (int m = threadIdx.x; (m + 1) * skip <= X.wcap; m += NT) {
      const short* w = X.raw + (size_t)m * div;
      int sum = 0;
      for (int j = 0; j < div; j++) sum += (int)w[j];
      for (int r = 0; r < skip; r++) {
        if ((m + 1) * skip + r > X.wcap) break;
        const unsigned mag = (unsigned)(sum < 0 ? -sum : sum);
        const int qm = (int)__umulhi(mag, M);
        const unsigned short u = (unsigned short)((sum < 0 ? -qm : qm) + 32768);
        X.pl[r * X.plStride + m] = u;
        if (m > 0) X.plB[r * X.plStride + m - 1] = u;
        // slide by one frame: drop frame r, take frame skip + r (its samples may lie past the window: unused then)
        for (int c = 0; c < C; c++) sum += (int)w[(skip + r) * C + c] - (int)w[r * C + c];
      }
    }
  }
  if (FAST) {
    lds_sync<NW>();
  }
}

// Running result of the dependency's sequential arg-min / arg-max scan over lags.
struct Sel {
  unsigned bestD, worstD;
  int bestP, worstP;
};

// Fold the 64 lanes' (d, p) candidates (lane order = ascending lag) into the running result, exactly as a
// sequential scan would: float ratios, DPP wave min/max, then an exact integer resolve on the scalar unit of the
// lanes within 2^-16 of the extremum (strict compare, so the FIRST lag wins ties).  Wave-uniform result.
template <bool WANT_MAX, bool FRESH = false>
__device__ __forceinline__ void select_fold(Sel& S, unsigned d, int p0, bool valid) {
  const int lane = threadIdx.x & 63;
  const float r = (float)d * __builtin_amdgcn_rcpf((float)(p0 + lane));  // within 2^-21 of d/p
  const float rmin = wave_min_f(valid ? r : __builtin_huge_valf());
  unsigned long long mmin = __builtin_amdgcn_ballot_w64(valid && r <= rmin * 1.0000153f);  // 1 + 2^-16
  if (FRESH) {  // S is empty and some lane is valid: the first candidate is taken without a comparison
    const int i = __builtin_ctzll(mmin);
    mmin &= mmin - 1;
    S.bestD = (unsigned)__builtin_amdgcn_readlane((int)d, i);
    S.bestP = p0 + i;
  }
  while (mmin) {
    const int i = __builtin_ctzll(mmin);
    mmin &= mmin - 1;
    const unsigned di = (unsigned)__builtin_amdgcn_readlane((int)d, i);
    const int pi = p0 + i;
    if (S.bestP == 0 || (unsigned long long)di * (unsigned)S.bestP < (unsigned long long)S.bestD * (unsigned)pi) {
      S.bestD = di;
      S.bestP = pi;
    }
  }
  if (WANT_MAX) {
    const float rmax = wave_max_f(valid ? r : 0.0f);
    unsigned long long mmax = __builtin_amdgcn_ballot_w64(valid && r >= rmax * 0.9999847f);
    while (mmax) {
      const int i = __builtin_ctzll(mmax);
      mmax &= mmax - 1;
      const unsigned di = (unsigned)__builtin_amdgcn_readlane((int)d, i);
      const int pi = p0 + i;
      if (S.worstP == 0 ||
          (unsigned long long)di * (unsigned)S.worstP > (unsigned long long)S.worstD * (unsigned)pi) {
        S.worstD = di;
        S.worstP = pi;
      }
    }
  }
}
__device__ __forceinline__ void select_finish(const Sel& S, int* retBest, int* retMin, int* retMax) {
  // the scan's initial (maxDiff = 0, worstPeriod = 255) survives unless some lag has diff > 0
  const unsigned worstD = S.worstD;
  const int worstP = (worstD == 0) ? 255 : S.worstP;
  *retBest = uni(S.bestP);
  *retMin = uni((int)udiv_small(S.bestD, (unsigned)S.bestP));
  *retMax = uni((int)udiv_small(worstD, (unsigned)worstP));
}

// One lane per lag, the whole sum in the lane (no cross-lane traffic): used for the coarse search, which every
// wave runs redundantly.  A0/A1: dword views of the signal array and of its copy shifted by one sample; o = offset
// of the search position inside them.
template <bool WANT_MAX, int FAST>
__device__ __forceinline__ void search_lane_per_lag(const unsigned* A0, const unsigned* A1, int o, int minP, int nl,
                                                    Sel& S, unsigned* dlane = nullptr) {
  const int lane = threadIdx.x & 63;
  const unsigned* ap = (o & 1) ? A1 + ((o - 1) >> 1) : A0 + (o >> 1);
  for (int base = 0; base < (FAST ? 1 : nl); base += 64) {  // FAST: at most 64 lags per search
    const bool valid = base + lane < nl;
    const int p = minP + base + lane;
    const int ob = o + p;
    const unsigned* bp = (ob & 1) ? A1 + ((ob - 1) >> 1) : A0 + (ob >> 1);
    const int nfull = valid ? (p >> 1) : 0;
    unsigned d = sad_run(ap, bp, 0, nfull);
    if (valid && (p & 1)) d = __builtin_amdgcn_sad_u16(ap[nfull] & 0xffffu, bp[nfull] & 0xffffu, d);  // term i = p-1
    if (dlane && base == 0) *dlane = d;
    select_fold<WANT_MAX, (FAST != 0)>(S, d, minP + base, valid);
  }
}

// The same search spread over the NW waves of the workgroup: wave w takes the pairs [w*CH, (w+1)*CH) of every lag
// (one lane per lag), the partial sums meet in LDS (ds_add_u32 into `buf`, which is all zero on entry), one
// LDS barrier, then every wave folds the totals itself.  `other` is the buffer the PREVIOUS step used; every wave
// is past reading it once this step's barrier is crossed, so it is cleared here for the next step.
template <int NW, bool WANT_MAX, int FAST>
__device__ __forceinline__ void search_split(const unsigned* A0, const unsigned* A1, int o, int minP, int nl,
                                             unsigned* buf, unsigned* other, Sel& S, WalkCtx& X, int sb, unsigned* dlane = nullptr) {
  (void)X; (void)sb;
  if (NW == 1) {
    search_lane_per_lag<WANT_MAX, FAST>(A0, A1, o, minP, nl, S, dlane);
    return;
  }
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int maxP = minP + nl - 1;
  const int CH = (((maxP >> 1) + NW) / NW + 3) & ~3;
  const unsigned* ap = (o & 1) ? A1 + ((o - 1) >> 1) : A0 + (o >> 1);
  for (int base = 0; base < (FAST ? 1 : nl); base += 64) {  // FAST: at most 64 lags per search
    const bool valid = base + lane < nl;
    const int p = minP + base + lane;
    const int ob = o + p;
    const unsigned* bp = (ob & 1) ? A1 + ((ob - 1) >> 1) : A0 + (ob >> 1);
    const int nfull = valid ? (p >> 1) : 0;
    // Chunk c of the pairs goes to wave c for c < 4 and to wave 11 - c above (8 waves): low chunks are the busy ones
    // (every lag has those pairs), and waves w and w + 4 share a SIMD, so each SIMD gets a busy and a light chunk.
    const int cw = (NW == 8) ? (wave < 4 ? wave : 11 - wave) : wave;
    int j1 = (cw + 1) * CH;
    if (j1 > nfull) j1 = nfull;
    unsigned d = sad_run(ap, bp, cw * CH, j1);
    if (cw == NW - 1 && valid && (p & 1))
      d = __builtin_amdgcn_sad_u16(ap[nfull] & 0xffffu, bp[nfull] & 0xffffu, d);  // the lone term i = p-1
    if (valid) atomicAdd(&buf[base + lane], d);
  }
  STAMP(sb);
  lds_sync<NW>();
  STAMP(sb + 1);
  for (int t = tid; t < 256; t += NT) other[t] = 0;
  for (int base = 0; base < (FAST ? 1 : nl); base += 64) {  // FAST: at most 64 lags per search
    const bool valid = base + lane < nl;
    const unsigned d = valid ? buf[base + lane] : 0u;
    if (dlane && base == 0) *dlane = d;
    select_fold<WANT_MAX, (FAST != 0)>(S, d, minP + base, valid);
  }
}

// findPitchPeriod at absolute position pos (all threads return the same value).
template <int NW, int FAST>
__device__ __forceinline__ int find_pitch_period(const SpxPlanDev& P, WalkCtx& X, WalkState& st, pos_t pos) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  const int C = FAST ? 1 : X.C, skip = P.skip, maxRequired = P.maxRequired;
  STAMP(1);
  ensure_window<NW, FAST>(X, pos, maxRequired + 2 * skip + 2);
  STAMP(2);
  const int o = (int)(pos - X.wbase);
  const bool direct = FAST ? false : (C == 1 && skip == 1);
  // ---- phase B: the decimated, biased search signal of this step (earlier readers are past a barrier) ----
  const int cnt = maxRequired / skip;
  if (!FAST && !direct) {
    const int div = skip * C;
    const double inv = xfade_rcp(div);
    for (int t = tid; t < cnt + 2; t += NT) {
      int sum = 0;
      if (C == 1) {
        const unsigned short* w = X.monoH + o + t * skip;
        for (int j = 0; j < skip; j++) sum += (int)w[j];
        sum -= 32768 * skip;
      } else {
        const short* w = X.raw + (size_t)(o + t * skip) * C;
        for (int j = 0; j < div; j++) sum += w[j];
      }
      // truncating sum / div via the exact double-reciprocal form (xfade_quot: |sum| <= 32768 div)
      const unsigned short u = (unsigned short)(xfade_quot(sum, inv) + 32768);
      X.dnH[t] = u;
      if (t > 0) X.dnHB[t - 1] = u;
    }
  }
  STAMP(3);
  // FAST: nothing was built, and the sum buffers this step adds into were cleared before the previous step's last
  // barrier, so no barrier is needed here
  if (!FAST) lds_sync<NW>();  // decimated signal visible; last step's buffer clearing finished
  STAMP(4);
  // ---- first search ----
  const unsigned* M0 = reinterpret_cast<const unsigned*>(X.monoH);
  const unsigned* M1 = reinterpret_cast<const unsigned*>(X.monoHB);
  const unsigned* D0 = reinterpret_cast<const unsigned*>(X.dnH);
  const unsigned* D1 = reinterpret_cast<const unsigned*>(X.dnHB);
  int oD = 0;  // where the decimated signal of this step starts in D0 / D1
  if (FAST) {
    oD = (o * X.skipM) >> 16;
    const int r = o - oD * skip;
    D0 = reinterpret_cast<const unsigned*>(X.pl + r * X.plStride);
    D1 = reinterpret_cast<const unsigned*>(X.plB + r * X.plStride);
  }
  int period, minDiff, maxDiff;
  const int minC = direct ? P.minPeriod : P.minPeriod / skip;
  const int maxC = direct ? P.maxPeriod : P.maxPeriod / skip;
  const int tg = st.prevPeriod_toggle & 1;
  st.prevPeriod_toggle ^= 1;
  st.steps++;
  Sel S1 = {0u, 0u, 0, 0};
  if (!FAST && (direct || skip == 1)) {  // this search is the final one: it also needs the worst lag
    search_split<NW, true, FAST>(direct ? M0 : D0, direct ? M1 : D1, direct ? o : 0, minC, maxC - minC + 1,
                           X.diffC + 256 * tg, X.diffC + 256 * (1 - tg), S1, X, 5);
    select_finish(S1, &period, &minDiff, &maxDiff);
    STAMP(7);
  } else {
    search_split<NW, false, FAST>(D0, D1, oD, minC, maxC - minC + 1, X.diffC + 256 * tg, X.diffC + 256 * (1 - tg), S1, X, 5);
    STAMP(7);
    period = S1.bestP * skip;
    int lo = period - (skip << 2), hi = period + (skip << 2);
    if (lo < P.minPeriod) lo = P.minPeriod;
    if (hi > P.maxPeriod) hi = P.maxPeriod;
    // ---- refine at full rate ----
    Sel S2 = {0u, 0u, 0, 0};
    if (FAST) {
      // Only "maxDiff > 3*minDiff" is ever asked of the worst lag, and max_p floor(d_p/p) = floor(max_p d_p/p), so
      // the test is "some lag has d_p >= (3*minDiff+1)*p": no arg-max, and only on the steps where the
      // previous-period rule can fire at all.
      unsigned dl = 0;
      search_split<NW, false, FAST>(M0, M1, o, lo, hi - lo + 1, X.diffR + 256 * tg, X.diffR + 256 * (1 - tg), S2, X, 8,
                                    &dl);
      period = uni(S2.bestP);
      minDiff = uni((int)udiv_small(S2.bestD, (unsigned)S2.bestP));
      maxDiff = 0x7fffffff;  // "a clear match" unless shown otherwise
      if (minDiff != 0 && st.prevPeriod != 0 && minDiff * 2 > st.prevMinDiff * 3) {
        const int lane = tid & 63;
        const bool valid = lane < hi - lo + 1;
        const unsigned long long need = (unsigned long long)(3u * (unsigned)minDiff + 1u) * (unsigned)(lo + lane);
        if (__builtin_amdgcn_ballot_w64(valid && (unsigned long long)dl >= need) == 0) maxDiff = 0;  // no lag that bad: keep the old period
      }
    } else {
      search_split<NW, true, FAST>(M0, M1, o, lo, hi - lo + 1, X.diffR + 256 * tg, X.diffR + 256 * (1 - tg), S2, X, 8);
      select_finish(S2, &period, &minDiff, &maxDiff);
    }
    STAMP(10);
  }
  int ret = period;
  if (!(minDiff == 0 || st.prevPeriod == 0) && !(maxDiff > minDiff * 3) && !(minDiff * 2 <= st.prevMinDiff * 3))
    ret = st.prevPeriod;
  st.prevMinDiff = minDiff;
  st.prevPeriod = period;
  return ret;
}

// Append n frames copied from absolute input position a.
template <int NW, int FAST>
__device__ __forceinline__ void emit_copy(const WalkCtx& X, WalkState& st, pos_t a, pos_t n) {
  constexpr int NT = 64 * NW;
  const int C = (FAST == 1) ? 1 : X.C;
  if (st.out_n + n > X.out_cap) st.overflow = 1;
  pos_t nv = X.out_cap - st.out_n;  // frames that still fit
  if (nv > n) nv = n;
  const pos_t o = a - X.wbase;
  int16_t* __restrict__ dst = X.out + (size_t)st.out_n * C;
  if (C == 1 && X.wbase >= 0 && o >= 0 && o + n <= X.wcap) {  // whole run inside the LDS window
    const unsigned short* w = X.monoH + o;
    for (int t = threadIdx.x; t < (int)nv; t += NT) dst[t] = (int16_t)((int)w[t] - 32768);
  } else if (FAST == 2 && X.wbase >= 0 && o >= 0 && o + n <= X.wcap) {  // multi-channel run inside the window
    const short* w = X.raw + (size_t)o * C;
    const int total = (int)nv * C;
    for (int e = threadIdx.x; e < total; e += NT) dst[e] = w[e];
  } else {
    const pos_t total = nv * C;
    for (pos_t e = threadIdx.x; e < total; e += NT) {
      pos_t f = e;
      int c = 0;
      if (C != 1) { f = e / C; c = (int)(e - f * C); }
      dst[e] = (int16_t)any_sample(X, a + f, c);
    }
  }
  st.out_n += n;
}

// Append n frames of cross-fade: out[t] = (down[t]*(n-t) + up[t]*t)/n, integer, truncating toward zero.
// |numerator| <= 32768*n < 2^31; the quotient: xfade_rcp / xfade_quot (spx_walk_common.h).
template <int NW, int FAST>
__device__ __forceinline__ void emit_overlap_add(const WalkCtx& X, pos_t a_down, pos_t a_up, int n,
                                                 pos_t out_at) {
  constexpr int NT = 64 * NW;
  const int C = (FAST == 1) ? 1 : X.C;
  const double inv = xfade_rcp(n);
  pos_t nv64 = X.out_cap - out_at;
  const int nv = nv64 > n ? n : (nv64 < 0 ? 0 : (int)nv64);
  int16_t* __restrict__ dst = X.out + (size_t)out_at * C;
  const pos_t od = a_down - X.wbase, ou = a_up - X.wbase;
  const bool inwin = X.wbase >= 0 && od >= 0 && ou >= 0 && od + n <= X.wcap && ou + n <= X.wcap;
  if (FAST == 1 || (C == 1 && inwin)) {  // FAST: ensure_window(pos, maxRequired + ...) of the search covers both ramps
    const unsigned short* wd = X.monoH + od;
    const unsigned short* wu = X.monoH + ou;
    // FAST: n <= maxPeriod < 512 (rates below 32 kHz), so with eight waves this is one predicated pass, no loop
    for (int t = threadIdx.x; t < nv; t += NT) {
      const int d = (int)wd[t] - 32768, u = (int)wu[t] - 32768;
      dst[t] = (int16_t)xfade_quot(d * (n - t) + u * t, inv);
      if (FAST && NT >= 512) break;
    }
  } else if (FAST == 2) {
    // multi-channel speed-up kernel: both ramps lie in the window; element e = t*C + c of each is contiguous in `raw`
    const short* rd = X.raw + (size_t)od * C;
    const short* ru = X.raw + (size_t)ou * C;
    const int total = nv * C;
    const unsigned invC = (0x10000u + (unsigned)C - 1u) / (unsigned)C;  // e / C for e < 8192, C <= 8
    for (int e = threadIdx.x; e < total; e += NT) {
      const int t = (C == 2) ? (e >> 1) : (int)(((unsigned)e * invC) >> 16);
      const int d = rd[e], u = ru[e];
      dst[e] = (int16_t)xfade_quot(d * (n - t) + u * t, inv);
    }
  } else {
    const int total = nv * C;
    for (int e = threadIdx.x; e < total; e += NT) {
      int t = e, c = 0;
      if (C != 1) { t = e / C; c = e - t * C; }
      const int d = any_sample(X, a_down + t, c), u = any_sample(X, a_up + t, c);
      dst[e] = (int16_t)xfade_quot(d * (n - t) + u * t, inv);
    }
  }
}

// processStreamInput with `avail` frames handed over so far (absolute count).
template <int NW, int FAST>
__device__ __forceinline__ void tsm_process(const SpxPlanDev& P, WalkCtx& X, WalkState& st, float speed, pos_t avail) {
  const int maxRequired = P.maxRequired;
  if ((double)speed > 1.00001 || (double)speed < 0.99999) {
    const pos_t numSamples = avail - st.base;
    if (numSamples < maxRequired) return;
    pos_t position = 0;
    do {
      if (st.remaining > 0) {
        int n = st.remaining;
        if (n > maxRequired) n = maxRequired;
        emit_copy<NW, FAST>(X, st, st.base + position, n);
        st.remaining -= n;
        position += n;
      } else {
        const pos_t pos = st.base + position;
        const int period = find_pitch_period<NW, FAST>(P, X, st, pos);
        if ((double)speed > 1.0) {
          int n;  // the dependency converts to long; every value here fits an int
          if (speed >= 2.0f) {
            n = uni((int)((float)period / (speed - 1.0f)));
          } else {
            n = period;
            st.remaining = uni((int)((float)period * (2.0f - speed) / (speed - 1.0f)));
          }
          if (st.out_n + n > X.out_cap) st.overflow = 1;
          if (n == 0) return;  // the dependency treats this as failure and leaves the input untouched
          STAMP(11);
          emit_overlap_add<NW, FAST>(X, pos, pos + period, n, st.out_n);
          STAMP(13);
          st.out_n += n;
          position += period + n;
        } else {
          int n;
          if (speed < 0.5f) {
            n = uni((int)((float)period * speed / (1.0f - speed)));
          } else {
            n = period;
            st.remaining = uni((int)((float)period * (2.0f * speed - 1.0f) / (1.0f - speed)));
          }
          emit_copy<NW, FAST>(X, st, pos, period);
          if (st.out_n + n > X.out_cap) st.overflow = 1;
          if (n == 0) return;
          emit_overlap_add<NW, FAST>(X, pos + period, pos, n, st.out_n);
          st.out_n += n;
          position += n;
        }
      }
    } while (position + maxRequired <= numSamples);
    st.base += position;
  } else {
    emit_copy<NW, FAST>(X, st, st.base, avail - st.base);
    st.base = avail;
  }
}

// ---- FAST kernels (mono, every job's speed > 1 and 0 <= nonlinear <= 1, so every speed the stage sees is >= 1) ----
// The pitch steps one event can run = the loop of processStreamInput for speed > 1 with `avail` frames handed over.
// The caller guarantees avail - st.base >= maxRequired.  Returns false when a step fails (n == 0): the dependency
// then returns without removing the input it has consumed in this call, so st.base keeps its value.
template <int NW, int FAST>
__device__ __forceinline__ bool fast_steps(const SpxPlanDev& P, WalkCtx& X, WalkState& st, float speed, pos_t avail) {
  const int maxRequired = P.maxRequired;
  const bool ge2 = speed >= 2.0f;
  const float sm1 = speed - 1.0f, twom = 2.0f - speed;
  pos_t pos = st.base;
  do {
    if (st.remaining > 0) {
      int n = st.remaining;
      if (n > maxRequired) n = maxRequired;
      ensure_window<NW, FAST>(X, pos, n);
      emit_copy<NW, FAST>(X, st, pos, n);
      st.remaining -= n;
      pos += n;
    } else {
      const int period = find_pitch_period<NW, FAST>(P, X, st, pos);
      int n;
      if (ge2) {
        n = uni((int)((float)period / sm1));
      } else {
        n = period;
        st.remaining = uni((int)((float)period * twom / sm1));
      }
      if (st.out_n + n > X.out_cap) st.overflow = 1;
      if (n == 0) return false;
      STAMP(11);
      emit_overlap_add<NW, FAST>(X, pos, pos + period, n, st.out_n);
      STAMP(13);
      st.out_n += n;
      pos += period + n;
    }
  } while (pos + maxRequired <= avail);
  st.base = pos;
  return true;
}


// All events of a chunk through ONE call site of the step code (the kernel stays small):
//   ordinary events [ev0, ev1): nonlinear -- event e sets the speed of tension frame e (e < K; later events keep the
//     last speed) and hands over B more frames (soniclib.c:354,369,538-550); linear -- one event handing over
//     everything new at the stream's speed (soniclib.c:397-399);
//   then, if `fin`, sonicIntFlushStream (soniclib.c:551): pad 2*maxRequired zeros, process, truncate.
// Most nonlinear events cannot run a step (a step needs maxRequired frames, an event brings B): those cost a few
// scalar instructions.  The speeds of 64 consecutive events sit in one VGPR (lane = event), so picking one is a
// v_readlane, not a memory access.
template <int NW, int FAST>
__device__ __forceinline__ void fast_events(const SpxPlanDev& P, WalkCtx& X, WalkState& st, const float* scr, pos_t ev0,
                                            pos_t ev1, pos_t K, pos_t& avail, int B, bool linear, pos_t n_in, bool fin,
                                            float& curSpeed) {
  const int lane = threadIdx.x & 63;
  const int maxRequired = P.maxRequired;
  const pos_t Kc = ev1 < K ? ev1 : K;  // tension events of this chunk: [ev0, Kc)
  float tailSpeed = curSpeed;          // what later events run at, and what the stream carries on
  if (Kc > ev0) tailSpeed = unif(scr[4 * (size_t)(Kc - 1) + 3]);
  const bool tailUnity = speed_is_unity(tailSpeed);
  pos_t blk0 = ev0 - 64;
  float spv = 0.0f;
  unsigned long long unityMask = 0;
  const pos_t ev_end = ev1 + (fin ? 1 : 0);
  for (pos_t e = ev0; e < ev_end; e++) {
    float speed = tailSpeed;
    bool unity = tailUnity;
    pos_t expected = 0;
    const bool flush = e >= ev1;
    if (!flush) {
      avail = linear ? n_in : avail + B;
      if (e < K) {
        int i = e - blk0;
        if (i >= 64) {  // next 64 events' speeds into the lanes
          blk0 = e;
          i = 0;
          const pos_t idx = blk0 + lane;
          spv = (idx < K) ? scr[4 * (size_t)idx + 3] : 2.0f;
          unityMask = __builtin_amdgcn_ballot_w64(speed_is_unity(spv));
        }
        unity = (unityMask >> i) & 1;
        if (!unity && avail - st.base < maxRequired) continue;
        speed = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, spv), i));
      }
    } else {
      const pos_t remainingS = avail - st.base;
      expected = st.out_n + uni((int)(((float)remainingS / tailSpeed + 0) / 1.0f + 0.5f));
      if (X.flush_rem) {  // a rate stage follows: it truncates its own output (and needs the count)
        expected = 0x7fffffff;
        if (threadIdx.x == 0) { X.flush_rem[0] = (int)remainingS; X.flush_rem[1] = (int)st.out_n; }
      }
      X.limit = avail;  // everything from here on reads as the flush's zero padding
      lds_sync<NW>();
      X.wbase = -1;     // the window may hold samples past the new limit
      avail += 2 * maxRequired;
    }
    STAMP(0);
    if (unity) {
      emit_copy<NW, FAST>(X, st, st.base, avail - st.base);
      st.base = avail;
    } else if (avail - st.base >= maxRequired) {
      (void)fast_steps<NW, FAST>(P, X, st, speed, avail);
    }
    STAMP(12);
    if (flush) {
      if (st.out_n > expected) st.out_n = expected;
      st.base = avail;  // the dependency empties its input after a flush
      st.remaining = 0;
    }
  }
  curSpeed = tailSpeed;
}

// LDS layout (bytes), shared by host and device
struct WalkLds {
  int off_sA, off_mono, off_monoB, off_raw, off_dn, off_dnB, off_pl, off_plB, plStride, off_diffC, off_diffR,
      off_wait, total, wcap;
};
// fast != 0: the speed-up kernels, which need neither the speed staging array nor the per-step decimated signal.
static __host__ __device__ inline WalkLds walk_lds_layout(const SpxPlanDev& P, int maxC, int fast = 0) {
  WalkLds L;
  const int need = P.maxRequired + 2 * P.skip + 2;
  int wcap = 4096;
  if (wcap < 4 * need) wcap = 4 * need;
  // many channels: a shorter window (every channel of it sits in LDS), but never shorter than one search needs
  while ((size_t)wcap * (maxC > 1 ? maxC + 2 : 2) * 2 > 48 * 1024 && wcap > need + 64) wcap = wcap / 2 > need + 64 ? wcap / 2 : need + 64;
  wcap = (wcap + 7) & ~7;
  L.wcap = wcap;
  int o = 0;
  L.off_sA = o; if (!fast) o += SPX_CH * 4;
  L.off_wait = o; o += 16;
  L.off_mono = o;
  const int mb = ((wcap + 8) * 2 + 15) & ~15;
  o += mb;
  L.off_monoB = o; o += mb;
  L.off_raw = o;
  if (maxC > 1) o += ((wcap + 1) * maxC * 2 + 15) & ~15;
  const int dnb = fast ? 0 : ((P.maxRequired / P.skip + 8) * 2 + 15) & ~15;
  L.off_dn = o; o += dnb;
  L.off_dnB = o; o += dnb;
  // decimated planes of the FAST kernels (allocated always: the layout does not depend on the kernel variant)
  L.plStride = ((wcap / (P.skip > 0 ? P.skip : 1) + 4) + 1) & ~1;
  const int plb = (L.plStride * (P.skip > 0 ? P.skip : 1) * 2 + 15) & ~15;
  L.off_pl = o; o += plb;
  L.off_plB = o; o += plb;
  L.off_diffC = o; o += 2 * 256 * 4;  // double-buffered per-lag sums, first search
  L.off_diffR = o; o += 2 * 256 * 4;  // double-buffered per-lag sums, refine search
  L.total = o;
  return L;
}

template <int NW, int FAST>
__global__ void __launch_bounds__(64 * NW)
spx_walk_kernel(SpxPlanDev P, const SpxStreamDev* __restrict__ streams, const int16_t* __restrict__ in_base,
                int16_t* __restrict__ out_base, int64_t* __restrict__ n_out, SpxStreamState* __restrict__ states,
                const float* scratch_base, int maxC, const int* speed_ready) {
  constexpr int NT = 64 * NW;
  // the walk is the latency-critical chain: where the analysis kernel shares a SIMD, these waves issue first
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const SpxStreamDev S = streams[blockIdx.x];
  const int Ttot = S.n_frames, F = P.F, B = P.B;
  const float Rg = S.speed, nl = S.nonlinear;

  const WalkLds LY = walk_lds_layout(P, maxC, FAST);
  float* sA = reinterpret_cast<float*>(lds + LY.off_sA);  // [SPX_CH]

  // ---- the part of the stream state this stage owns (the tension kernel owns the filter states) ----
  SpxStreamState Z;
  if (S.flags & SPX_F_INIT) {
    Z.w.base = 0; Z.w.out_n = 0; Z.w.avail = 0; Z.w.remaining = 0; Z.w.prevPeriod = 0; Z.w.prevMinDiff = 0;
    Z.w.overflow = 0; Z.w.prevPeriod_toggle = 0; Z.w.steps = 0;
    Z.curSpeed = Rg;    // sonicSetSpeed -> sonicIntSetSpeed, soniclib.c:182
    Z.handed = 0;
  } else {
    Z.w = states[blockIdx.x].w;
    Z.curSpeed = states[blockIdx.x].curSpeed;
    Z.handed = states[blockIdx.x].handed;
    // sonicSetSpeed between writes reaches the TSM stage at once (soniclib.c:182); in nonlinear mode the
    // next tension frame overrides it (soniclib.c:354)
    if ((nl == 0.0f && !(S.flags & SPX_F_KEEP_SPEED)) || (S.flags & SPX_F_SPEED_SET)) Z.curSpeed = Rg;
  }

  const float* scr = scratch_base + (size_t)S.frame_off * 4;  // per frame: ..., speed (written by the tension kernel)
  int* sWait = reinterpret_cast<int*>(lds + LY.off_wait);
  // ---------------------------------- TSM stage context ----------------------------------
  WalkCtx X;
  X.flush_rem = (S.flags & SPX_F_NO_TRUNC) ? &states[blockIdx.x].flush_remaining : nullptr;
  X.in = in_base + S.in_off - S.tsm_shift * S.channels;  // indexed by TSM position (= input frame + flush padding so far)
  X.out = out_base + S.out_off;
  X.out_cap = (pos_t)(S.out_cap > 0x7fffffff ? 0x7fffffff : S.out_cap);
  X.limit = (pos_t)(S.n_in + S.tsm_shift);
  X.C = S.channels;
  X.monoH = reinterpret_cast<unsigned short*>(lds + LY.off_mono);
  X.monoHB = reinterpret_cast<unsigned short*>(lds + LY.off_monoB);
  X.raw = reinterpret_cast<short*>(lds + LY.off_raw);
  X.wbase = -1;
  X.wcap = LY.wcap;
  X.dnH = reinterpret_cast<unsigned short*>(lds + LY.off_dn);
  X.dnHB = reinterpret_cast<unsigned short*>(lds + LY.off_dnB);
  X.pl = reinterpret_cast<unsigned short*>(lds + LY.off_pl);
  X.plB = reinterpret_cast<unsigned short*>(lds + LY.off_plB);
  X.plStride = LY.plStride;
  X.skip = P.skip;
  X.skipM = (65536 + P.skip - 1) / P.skip;
  X.skipM32 = (unsigned)((0x100000000ull + (unsigned)P.skip - 1) / (unsigned)P.skip);
  X.diffC = reinterpret_cast<unsigned*>(lds + LY.off_diffC);
  X.diffR = reinterpret_cast<unsigned*>(lds + LY.off_diffR);
  for (int t = tid; t < 512; t += NT) { X.diffR[t] = 0; X.diffC[t] = 0; }
  __syncthreads();
  STAMP_DECL
  WalkState st;
  st.base = uni((pos_t)Z.w.base); st.out_n = uni((pos_t)Z.w.out_n); st.avail = uni((pos_t)Z.w.avail);
  st.remaining = uni(Z.w.remaining); st.prevPeriod = uni(Z.w.prevPeriod); st.prevMinDiff = uni(Z.w.prevMinDiff);
  st.overflow = uni(Z.w.overflow); st.prevPeriod_toggle = uni(Z.w.prevPeriod_toggle); st.steps = 0;
  float curSpeed = unif(Z.curSpeed);
  pos_t avail = st.avail;
  pos_t handed = (nl != 0.0f) ? ((S.flags & SPX_F_HANDED_IN) ? S.handed_in : Z.handed) : 0;
  const bool do_flush = (S.flags & SPX_F_FLUSH) != 0;
  // The speeds come from the tension kernel.  Sequential launches (speed_ready == nullptr): all of them are there.
  // Concurrent launches: the tension kernel runs beside this one and publishes the number of tension frames whose
  // speeds are final (agent-scope release there; one relaxed poll + agent-scope acquire here, Guideline 16); the walk
  // takes whatever is ready, and only waits when it has caught up -- after the first frames it never does.
  const int K_total = (nl != 0.0f && Ttot >= F) ? Ttot - F + 1 : 0;   // soniclib.c:317
  // concurrent mode: this workgroup has been placed (the engine's idle-start gate counts the arrivals, spx_engine.hip)
  if (speed_ready != nullptr && tid == 0)
    __hip_atomic_fetch_add(const_cast<int*>(speed_ready) + gridDim.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    int K = K_total;
    if (speed_ready != nullptr && K_total > 0) {
      if (tid == 0) {
        int got;
        unsigned spins = 0;
        for (;;) {
          got = __hip_atomic_load(&speed_ready[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (got < 0 || got > (int)handed || got >= K_total) break;
          __builtin_amdgcn_s_sleep(32);
          if (++spins > (1u << 22)) { got = -2; break; }  // ~seconds: never hang the GPU on a lost producer
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *sWait = got;
      }
      __syncthreads();
      K = uni(*sWait);
      __syncthreads();  // sWait may be rewritten by the next round
      if (K < 0) { st.overflow = 2; break; }  // the producer was lost: its own status, not an overflow
      if (K > K_total) K = K_total;
    }
    const bool last = K >= K_total;
    // Events, in the order the shim issues them:
    //   nonlinear: one (setSpeed, write B) per tension frame           soniclib.c:354,369
    //              at flush, the remaining complete ring buffers at the last speed   soniclib.c:538-550
    //   linear:    one write of everything new (soniclib.c:397-399; chunking is irrelevant at constant speed)
    //   at flush:  sonicIntFlushStream (soniclib.c:551): pad 2*maxRequired zeros, process, truncate
    const bool fin = last && do_flush;
    const pos_t ev0 = handed;
    pos_t ev1;  // one past the last ordinary event of this chunk
    if (nl != 0.0f) ev1 = fin ? ((S.flags & SPX_F_HANDED_IN) ? (pos_t)S.ring_bufs : (pos_t)(S.n_in / B)) : K;  // complete ring buffers written: soniclib.c:446-449
    else ev1 = (last && (pos_t)(S.n_in + S.tsm_shift) > avail) ? 1 : 0;
    if (ev1 < ev0) ev1 = ev0;
    const pos_t ev_end = ev1 + (fin ? 1 : 0);
    (void)ev_end;
    if constexpr (FAST != 0) {
      fast_events<NW, FAST>(P, X, st, scr, ev0, ev1, (pos_t)K, avail, B, nl == 0.0f, (pos_t)(S.n_in + S.tsm_shift), fin,
                            curSpeed);
    } else {
      for (pos_t ev = ev0; ev < ev_end; ev++) {
        pos_t expected = 0;
        if (ev < ev1) {
          if (nl != 0.0f) {
            if (ev < K) {
              const int i = (int)((ev - ev0) % SPX_CH);
              if (i == 0) {  // stage the next chunk of speeds in LDS
                __syncthreads();
                const int n = (int)min((pos_t)SPX_CH, (pos_t)K - ev);
                for (int t = tid; t < n; t += NT) sA[t] = scr[4 * (ev + t) + 3];
                __syncthreads();
              }
              curSpeed = unif(sA[i]);
            }
            avail += B;
          } else {
            avail = (pos_t)(S.n_in + S.tsm_shift);
          }
        } else {
          const pos_t remainingS = avail - st.base;
          expected = st.out_n + uni((int)(((float)remainingS / curSpeed + 0) / 1.0f + 0.5f));
          if (X.flush_rem) {  // a rate stage follows: it truncates its own output (and needs the count)
            expected = 0x7fffffff;
            if (threadIdx.x == 0) { X.flush_rem[0] = (int)remainingS; X.flush_rem[1] = (int)st.out_n; }
          }
          X.limit = avail;  // everything from here on reads as the flush's zero padding
          lds_sync<NW>();
          X.wbase = -1;     // the window may hold samples past the new limit
          avail += 2 * P.maxRequired;
        }
        STAMP(0);
        tsm_process<NW, FAST>(P, X, st, curSpeed, avail);
        STAMP(12);
        if (ev >= ev1) {
          if (st.out_n > expected) st.out_n = expected;
          st.base = avail;  // the dependency empties its input after a flush
          st.remaining = 0;
        }
      }
    }
    if (nl != 0.0f) handed = ev1;
    if (last) break;
  }
  STAMP_FLUSH
  if (tid == 0) {
    Z.w.base = st.base; Z.w.out_n = st.out_n; Z.w.avail = avail; Z.w.remaining = st.remaining;
    Z.w.prevPeriod = st.prevPeriod; Z.w.prevMinDiff = st.prevMinDiff; Z.w.overflow = st.overflow;
    Z.w.prevPeriod_toggle = st.prevPeriod_toggle; Z.w.steps += st.steps;   // (Z.w.steps: 0 on a fresh stream, else what earlier jobs ran)
    states[blockIdx.x].w = Z.w;  // field-wise: the tension kernel may be writing its own fields of this record
    states[blockIdx.x].curSpeed = curSpeed;
    if (nl != 0.0f) states[blockIdx.x].handed = (int)handed;
    if (n_out) n_out[blockIdx.x] = st.overflow == 2 ? INT64_MIN : (st.overflow ? -(int64_t)st.out_n : (int64_t)st.out_n);
  }
}

// Tuning knobs from the environment, read ONCE per process (never on the launch path).
struct WalkTuning {
  bool generic, old_fast;
  int nw, nwm, nwc, wcap;
};
static const WalkTuning& walk_tuning() {
  static const WalkTuning T = [] {
    WalkTuning t;
    auto geti = [](const char* k) { const char* e = spx_tuning_env(k); return e ? atoi(e) : -1; };
    t.generic = spx_tuning_env("SPX_WALK_GENERIC") != nullptr;   // force the general kernel
    t.old_fast = spx_tuning_env("SPX_WALK_OLD") != nullptr;      // mono speed-up batches on spx_walk_kernel<NW, 1> (A/B only)
    t.nw = geti("SPX_WALK_NW");
    t.nwm = geti("SPX_WALK_NWM");
    t.nwc = geti("SPX_WALK_NWC");
    t.wcap = geti("SPX_WALK_WCAP");
    return t;
  }();
  return T;
}

// Which kernel variant a batch gets: 0 general, 1 speed-up mono, 2 speed-up multi-channel.  FAST: all streams speeding
// up (speed > 1, 0 <= nonlinear <= 1: the stage never sees a speed below 1), decimated search, at most 128 lags in the coarse
// search (more than 64: the wide-coarse instantiations, round 5 -- 11.025 kHz has 72) and 121 in the refine search (rates below 64 kHz; spx_walk_fast_supports has the last word per wave count), and
// skip x channels <= 56 (the refill's exact division of the decimated planes).
static int walk_mode(const SpxPlanDev& P, int maxC, bool speedup_only) {
  if (speedup_only && P.skip >= 2 && (P.maxPeriod / P.skip - P.minPeriod / P.skip + 1) <= 128 && (8 * P.skip + 1) <= 121 &&
      P.skip * maxC <= 56 && maxC <= 8 && !walk_tuning().generic)
    return (maxC == 1) ? 1 : 2;
  return 0;
}

// Kernel variant, waves per stream and LDS per stream for a batch: the one place the launcher and the engine's
// co-residency arithmetic both ask.
SpxWalkConfig spx_walk_config(const SpxPlanDev& P, int n_streams, int maxC, bool speedup_only, bool short_jobs, bool lean, bool any_speed,
                              bool short_window) {
  if (maxC < 1) maxC = 1;
  const WalkTuning& T = walk_tuning();
  SpxWalkConfig c;
  static const bool no_slow_fast = spx_tuning_env("SPX_NO_SLOW_FAST") != nullptr;   // A/B: slow-down batches on the general kernel
  const bool slow = !speedup_only && any_speed && !no_slow_fast;
  c.mode = walk_mode(P, maxC, speedup_only || slow);
  c.fast_kernel = ((c.mode == 1 || c.mode == 2) && !T.old_fast);  // spx_walk_fast_kernel: mono and (round 2) multi-channel
  // Waves per stream of spx_walk_kernel.  Measured on MI355X, 10 s streams (ms per call; 2 / 4 / 8 waves): 256 streams
  // 3.67 / 3.19 / 2.91 (walk kernel alone), 512: 5.40 / 4.78 / 5.63, 1024: 10.3 / 9.3 / 10.5, 2048: 19.7 / 17.7 / -.
  c.nw = (n_streams <= 256) ? 8 : 4;
  if (T.nw > 0) c.nw = T.nw;
  // spx_walk_fast_kernel: search waves + output waves, window frames.  Three regimes (MI355X, 16 kHz mono x 10 s, ms per call;
  // profiles/r03/r03l_tp_variants.txt):
  //   up to two streams per CU: 4 search + 4 output waves, 4096-frame window -- a stream's chain is the run time, the
  //     output work is off it (512 streams: 3.63 against 3.89 without output waves)
  //   beyond: THROUGHPUT form -- 2 search waves, no output waves, 1536-frame window: 15 KB of LDS and two waves per stream,
  //     eight streams per CU at four waves per SIMD; the chains hide each other's latencies and the control flow is
  //     executed twice per stream, not four times (1024 streams: 5.5 against 6.3; 2048: 9.6 against 12.4; 4096: 19.0)
  // (per device: a process may hold plans on several GPUs; the count is cached, the query is not free)
  const int cus = [] {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    (void)hipGetLastError();
    cache[dev].store(v, std::memory_order_relaxed);
    return v;
  }();
  //   (the register file holds five waves of the 4 + 4 form per SIMD, i.e. TWO such workgroups per CU: a third waits for a
  //   slot -- 576 streams 5.8 ms -- so the throughput form starts right above two streams per CU)
  //   short jobs (coalesced sonic2.h writes: a few pitch steps per stream and launch): nothing to amortise a 2-wave step's
  //   longer latency over; 4 search waves, output waves while a stream has a CU to itself (1024 handles x 1000 frames:
  //   511 us per round against 714 in the throughput form)
  const bool throughput = !short_jobs && n_streams > 2 * cus;
  c.nwm = throughput ? 2 : 4;
  c.nwc = throughput ? 0 : ((short_jobs && n_streams > cus) ? 0 : 4);
  c.wcap = throughput ? 1536 : 4096;
  // long jobs with at most two streams per CU on the rate-specialised kernels with output waves: the 8192-frame window (half
  // as many refills; spx_walk_fast.hip, SPEC = 1)
  // (not for the instantiations that serve slow-down: those exist in their plan-driven form only)
  if (!throughput && !short_jobs && c.fast_kernel && c.nwm == 4 && c.nwc == 4 && (P.rate == 16000 || P.rate == 22050) && T.nwm <= 0 &&
      T.nwc < 0 && !slow) c.wcap = 8192;
  // rates from 24 kHz (the eight-search-wave form): a step needs up to 2 x maxRequired frames of window, so the 4096-frame window
  // is refilled every ~2 700 frames at 44.1 kHz -- 160 times per 10 s stream; twice the window, a third of the refills
  if (!throughput && !short_jobs && c.fast_kernel && P.skip >= 6 && T.wcap <= 0) c.wcap = 8192;
  if (lean && !throughput) { c.nwc = 0; c.wcap = 4096; }
  if (short_window && !throughput && P.skip < 6 && c.wcap > 4096) c.wcap = 4096;
  if (T.nwm > 0) c.nwm = T.nwm;
  if (T.nwc >= 0) c.nwc = T.nwc;
  if (T.wcap > 0) c.wcap = T.wcap;
  const int need = P.maxRequired + 2 * P.skip + 2;
  if (c.wcap < 2 * need) c.wcap = 2 * need;
  c.wcap = (c.wcap + 7) & ~7;
  if (c.nwm != 2 && c.nwm != 4 && c.nwm != 8) c.nwm = 4;
#ifndef SPX_TUNING
  c.nwc = c.nwc >= 4 ? 4 : 0;          // the shipped library's forms (spx_walk_fast.hip SPX_FAST_FORMS)
  if (c.nwm == 2) c.nwc = 0;
#endif
  if (c.nwm == 8 && P.skip < 6) c.nwm = 4;   // the eight-search-wave form's sum buffers exist in the LDS layout from skip 6 on
  while (c.fast_kernel && !spx_walk_fast_supports(P, c.nwm)) {  // the coarse triangle must fit the search lanes
    if (c.nwm < 4) c.nwm = 4; else if (c.nwm < 8) c.nwm = 8; else c.fast_kernel = false;
  }
  // without spx_walk_fast_kernel: modes 1 / 2 of spx_walk_kernel hold one lag per lane and the 4096-frame window
  if (!c.fast_kernel && c.mode != 0 && !((8 * P.skip + 1) <= 64 && (P.maxPeriod / P.skip - P.minPeriod / P.skip + 1) <= 64 &&
                                           walk_lds_layout(P, maxC).wcap == 4096)) c.mode = 0;
  c.slow = slow && c.fast_kernel;
  if (c.fast_kernel) {
    c.mode = 1;
    if (c.nwm == 8) c.nwc = 4;           // the eight-search-wave form always has its four output waves
    c.waves = c.nwm + c.nwc;
    c.lds = spx_walk_fast_lds_bytes(P, c.wcap);
  } else {
    c.waves = c.nw;
    c.lds = (size_t)walk_lds_layout(P, maxC, c.mode).total;
  }
  return c;
}
int spx_walk_fast_vgprs(const SpxPlanDev& P, int nwm, int nwc, int wcap, int maxC, int* scratch_bytes = nullptr, bool slow = false);
int spx_walk_kernel_regs(const SpxPlanDev& P, int n_streams, int maxC, bool speedup_only, bool short_jobs, bool lean, int* scratch_bytes,
                         bool any_speed, bool short_window) {
  const SpxWalkConfig cfg = spx_walk_config(P, n_streams, maxC < 1 ? 1 : maxC, speedup_only, short_jobs, lean, any_speed, short_window);
  if (cfg.fast_kernel) return spx_walk_fast_vgprs(P, cfg.nwm, cfg.nwc, cfg.wcap, maxC, scratch_bytes, cfg.slow);
  const void* fn;
#define SPX_FN_W(NWV) (cfg.mode == 1 ? reinterpret_cast<const void*>(spx_walk_kernel<NWV, 1>)   \
                       : cfg.mode == 2 ? reinterpret_cast<const void*>(spx_walk_kernel<NWV, 2>) \
                                       : reinterpret_cast<const void*>(spx_walk_kernel<NWV, 0>))
  switch (cfg.nw) {
    case 1: fn = SPX_FN_W(1); break;
    case 2: fn = SPX_FN_W(2); break;
    case 8: fn = SPX_FN_W(8); break;
    case 16: fn = SPX_FN_W(16); break;
    default: fn = SPX_FN_W(4); break;
  }
#undef SPX_FN_W
  return spx_kernel_vgprs(fn, scratch_bytes);
}
int spx_walk_vgprs(const SpxPlanDev& P, int n_streams, int maxC, bool speedup_only, bool lean, bool any_speed, bool short_window) {
  return spx_walk_kernel_regs(P, n_streams, maxC, speedup_only, false, lean, nullptr, any_speed, short_window);
}
size_t spx_walk_lds_bytes(const SpxPlanDev& P, int maxC, bool speedup_only) {
  return spx_walk_config(P, 256, maxC, speedup_only).lds;
}

static std::atomic<int> g_last_walk_form{0};
extern "C" int spx_debug_last_walk_form(void) { return g_last_walk_form.load(std::memory_order_relaxed); }

void spx_launch_walk(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, int maxC, const int16_t* in,
                     int16_t* out, int64_t* n_out, SpxStreamState* states, const float* scratch,
                     const int* speed_ready, bool speedup_only, hipStream_t st, bool short_jobs, size_t lds_min, bool lean, bool any_speed,
                     bool short_window) {
  if (n_streams <= 0) return;
  if (maxC < 1) maxC = 1;
  const SpxWalkConfig cfg = spx_walk_config(P, n_streams, maxC, speedup_only, short_jobs, lean, any_speed, short_window);
  g_last_walk_form.store(cfg.fast_kernel ? 16 * cfg.nwm + cfg.nwc : 0, std::memory_order_relaxed);
  static const bool dbg_mode = getenv("SPX_DEBUG_MODE") != nullptr;
  if (dbg_mode) fprintf(stderr, "[spx walk] rate %d n %d maxC %d: %s %d + %d waves, window %d frames, %zu B LDS per stream (at least %zu asked)\n", P.rate, n_streams, maxC,
                        cfg.fast_kernel ? "fast kernel" : "general kernel", cfg.fast_kernel ? cfg.nwm : cfg.nw, cfg.fast_kernel ? cfg.nwc : 0, cfg.wcap, cfg.lds, lds_min);
  if (cfg.fast_kernel) {
    spx_launch_walk_fast(P, streams, n_streams, in, out, n_out, states, scratch, speed_ready, cfg.nwm, cfg.nwc, cfg.wcap,
                         maxC, st, lds_min, cfg.slow);
    return;
  }
  const int fast = cfg.mode;
  const int nw = cfg.nw;
  const WalkLds LY = walk_lds_layout(P, maxC, fast);
#define SPX_LAUNCH_WALK(NWV)                                                                                     \
  do {                                                                                                           \
    if (fast == 1)                                                                                               \
      hipLaunchKernelGGL((spx_walk_kernel<NWV, 1>), dim3(n_streams), dim3(64 * NWV), LY.total, st, P, streams,   \
                         in, out, n_out, states, scratch, maxC, speed_ready);                                    \
    else if (fast == 2)                                                                                          \
      hipLaunchKernelGGL((spx_walk_kernel<NWV, 2>), dim3(n_streams), dim3(64 * NWV), LY.total, st, P, streams,   \
                         in, out, n_out, states, scratch, maxC, speed_ready);                                    \
    else                                                                                                         \
      hipLaunchKernelGGL((spx_walk_kernel<NWV, 0>), dim3(n_streams), dim3(64 * NWV), LY.total, st, P, streams,   \
                         in, out, n_out, states, scratch, maxC, speed_ready);                                    \
  } while (0)
#ifdef SPX_STAMPS
  if (nw == 4) SPX_LAUNCH_WALK(4); else SPX_LAUNCH_WALK(8);  // the diagnostic build carries two kernels only
  return;
#endif
  switch (nw) {
    case 1: SPX_LAUNCH_WALK(1); break;
    case 2: SPX_LAUNCH_WALK(2); break;
    case 8: SPX_LAUNCH_WALK(8); break;
    case 16: SPX_LAUNCH_WALK(16); break;
    default: SPX_LAUNCH_WALK(4); break;
  }
#undef SPX_LAUNCH_WALK
}
