// Walk kernel for the common case -- mono streams that only ever speed up (every job speed > 1, 0 <= nonlinear <= 1),
// rates below 32 kHz (at most 64 lags in either pitch search): BASELINE configs[1..3] and the mono half of configs[4].
// Same stage as spx_walk.hip (a10 AMDF pitch search, a11 skip + cross-fade, FIFO bookkeeping, flush; driven by the
// shim's event sequence soniclib.c:354,369,538-551), same results bit for bit; what differs is how a pitch step is laid
// on the hardware, because a stream is a chain of ~130 dependent pitch steps per second of audio and at 256 streams per
// GPU (one workgroup per CU) the length of that chain is the run time:
//
//   * two kinds of wavefronts in a workgroup.  NWM "search" waves run the chain: the two AMDF searches, the decision,
//     the event bookkeeping -- all control flow, redundantly and uniformly.  NWC "output" waves never take part in a
//     search: they wait at the workgroup barrier (a waiting wave uses no issue slots, so each SIMD's search wave issues
//     alone), receive {cross-fade, copy, refill, poll, exit} commands through two LDS slots and produce every output
//     sample.  Output work is thereby off the chain.
//   * arg-min of diff/lag without floats or a resolve loop: key = floor(diff * 2^16 / lag) as an exact integer
//     (one fp64 fma against a 65536/lag table + truncation); keys order exactly like the rationals whenever
//     lag1*lag2 < 2^16 (16 kHz: 246^2) and never invert the order otherwise (ties are then resolved exactly, rarely);
//     one v_min_u32 DPP reduction, one ballot, first set bit = the lag a sequential scan would have chosen;
//     minDiff = key >> 16 comes for free.
//   * the rest as in spx_walk.hip: LDS sliding window of biased u16 samples kept twice (shifted by one) so that any lag
//     reads aligned pairs for v_sad_u16, decimated planes built at refill time, partial sums of the waves met with
//     ds_add_u32, LDS-only barriers (output stores are never waited for).
#include <stdlib.h>

#include "spx_walk_common.h"

enum { FCMD_STEP = 1, FCMD_COPY = 2, FCMD_REFILL = 3, FCMD_POLL = 4, FCMD_EXIT = 5 };
#define FCMD_INTS 16  // ints per command slot

// Diagnostic build only (-DSPX_STAMPS): per-phase shader-cycle sums of workgroup 0, wave 0.  Never in the product.
#ifdef SPX_STAMPS
__device__ unsigned long long g_spx_fstamps[32];
extern "C" void spx_debug_fstamps(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spx_fstamps), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spx_fstamps), z, sizeof(z));
  }
}
#ifndef SPX_STAMP_SEL
#define SPX_STAMP_SEL 0
#endif
#define FSTAMP_DECL X.stamp_acc = 0; X.stamp_steps = 0; X.stamp_last = __builtin_readcyclecounter(); X.stamp_t0 = X.stamp_last;
#define FSTAMP(i)                                                          \
  do {                                                                     \
    const unsigned long long t_ = __builtin_readcyclecounter();            \
    if ((i) == SPX_STAMP_SEL) X.stamp_acc += t_ - X.stamp_last;            \
    if ((i) == 1) X.stamp_steps++;                                         \
    X.stamp_last = t_;                                                     \
  } while (0)
#define FSTAMP_FLUSH                                                       \
  if (threadIdx.x == 0 && blockIdx.x == 0) {                               \
    g_spx_fstamps[SPX_STAMP_SEL] += X.stamp_acc;                            \
    g_spx_fstamps[30] += __builtin_readcyclecounter() - X.stamp_t0;         \
    g_spx_fstamps[31] += X.stamp_steps;                                     \
  }
#else
#define FSTAMP_DECL
#define FSTAMP(i)
#define FSTAMP_FLUSH
#endif

// LDS layout (bytes), shared by host and device
struct FastLds {
  int off_cmd, off_wait, off_sumC, off_sumR, off_inv, off_mono, off_monoB, off_pl, off_plB, plStrideB, total, wcap;
};
static __host__ __device__ inline FastLds fast_lds_layout(const SpxPlanDev& P, int wcap) {
  FastLds L;
  L.wcap = wcap;
  int o = 0;
  L.off_cmd = o; o += 2 * FCMD_INTS * 4;
  L.off_wait = o; o += 16;
  L.off_sumC = o; o += 2 * 64 * 4;
  L.off_sumR = o; o += 2 * 64 * 4;
  L.off_inv = o; o += ((P.maxPeriod + 2) * 8 + 15) & ~15;
  const int mb = ((wcap + 8) * 2 + 15) & ~15;
  L.off_mono = o; o += mb;
  L.off_monoB = o; o += mb;
  const int skip = P.skip > 0 ? P.skip : 1;
  const int plStride = ((wcap / skip + 4) + 1) & ~1;  // elements per plane (even)
  L.plStrideB = plStride * 2;
  const int plb = (plStride * skip * 2 + 15) & ~15;
  L.off_pl = o; o += plb;
  L.off_plB = o; o += plb;
  L.total = o;
  return L;
}

struct FastCtx {
  const int16_t* in;
  int16_t* out;
  unsigned char* lds;
  pos_t out_cap, limit, wbase;
  int wcap, offA0, offA1, offPl, offPlB, plStrideB;
  int skip, skipM;
  unsigned skipM32;
  unsigned* sumC;
  unsigned* sumR;
  const double* inv;
  int* cmd;
  int seq;                              // commands published (search waves) / consumed (output waves)
  int xf_n, xf_down, xf_period, xf_out;  // cross-fade decided but not yet handed to the output waves
#ifdef SPX_STAMPS
  unsigned long long stamp_last, stamp_acc, stamp_t0;
  unsigned stamp_steps;
#endif
};

__device__ __forceinline__ void fast_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  SPX_WAVE_REDUCE("v_min_u32_dpp", v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Every output sample of the stream is produced here: by the NWC output waves on command, or -- NWC == 0 -- by the search
// waves themselves.  `t0` = index of this thread among the NTO threads doing output work.
//   cross-fade (libsonic overlapAdd): out[t] = (down[t]*(n-t) + up[t]*t)/n, integer, truncating toward zero; both runs
//     lie in the LDS window.  |numerator| <= 32768*n < 2^31; the quotient is trunc(|num| * (1/n) + 2^-20) in double,
//     exact: non-integer quotients are at least 1/n >= 2^-11 below the next integer, integer ones land 2^-20 above.
//   copy: n frames from absolute input position src (straight from HBM, coalesced; beyond `limit` = flush padding = 0).
template <int NTO>
__device__ __forceinline__ void fast_outputs(const FastCtx& X, int t0, int xf_n, int xf_down, int xf_period, pos_t xf_out,
                                             int cp_n, pos_t cp_src, pos_t cp_out, pos_t limit) {
  if (xf_n > 0) {
    const double inv = 1.0 / (double)xf_n;
    pos_t room = X.out_cap - xf_out;
    const int nv = room > xf_n ? xf_n : (room < 0 ? 0 : (int)room);
    int16_t* __restrict__ dst = X.out + (size_t)xf_out;
    const unsigned short* wd = reinterpret_cast<const unsigned short*>(X.lds + X.offA0) + xf_down;
    const unsigned short* wu = wd + xf_period;
    for (int t = t0; t < nv; t += NTO) {
      const int d = (int)wd[t] - 32768, u = (int)wu[t] - 32768;
      const int num = d * (xf_n - t) + u * t;
      const int mag = num < 0 ? -num : num;
      const int qm = (int)((double)mag * inv + 9.5367431640625e-07);
      dst[t] = (int16_t)(num < 0 ? -qm : qm);
    }
  }
  if (cp_n > 0) {
    pos_t room = X.out_cap - cp_out;
    const int nv = room > cp_n ? cp_n : (room < 0 ? 0 : (int)room);
    int16_t* __restrict__ dst = X.out + (size_t)cp_out;
    const int16_t* __restrict__ src = X.in + (size_t)cp_src;
    const pos_t real = limit - cp_src;  // frames of real input from cp_src on (may be <= 0: all padding)
    for (int t = t0; t < nv; t += NTO) dst[t] = (t < real) ? src[t] : (int16_t)0;
  }
}

// Load the window [nb, nb + wcap] (biased u16, plus the copy shifted by one sample) and build the decimated planes:
// plane r (r < skip) holds S[m*skip + r], S[i] = truncated mean of window samples i .. i+skip-1, so the decimated
// search signal of a step at window offset o is plane (o % skip) from element o / skip on, contiguous.  All NT threads
// of the workgroup; three LDS barriers.
template <int NT>
__device__ __forceinline__ void fast_refill(FastCtx& X, pos_t nb, pos_t limit) {
  fast_sync();  // everyone is done reading the old window
  unsigned short* monoH = reinterpret_cast<unsigned short*>(X.lds + X.offA0);
  unsigned short* monoHB = reinterpret_cast<unsigned short*>(X.lds + X.offA1);
  const int16_t* __restrict__ src = X.in + nb;
  const pos_t room = limit - nb;  // frames of real input from nb on
  const int last = (int)(room < X.wcap + 1 ? room : X.wcap + 1) - 1;  // last index holding real input
  for (int k0 = threadIdx.x; k0 < X.wcap + 1; k0 += 8 * NT) {
    int v[8];
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
      if (k < X.wcap) monoH[k] = w;
      if (k > 0 && k < X.wcap + 1) monoHB[k - 1] = w;
    }
  }
  X.wbase = nb;
  fast_sync();
  // One thread per decimated index m: it reads the 2*skip-1 window samples m*skip .. m*skip+2*skip-2 once and slides
  // the sum over them, giving element m of every plane.  |sum| < 2^18 and skip <= 7, so the truncating division is
  // exactly mulhi(|sum|, ceil(2^32 / skip)).
  const int skip = X.skip;
  const unsigned M = X.skipM32;
  const int bias = 32768 * skip;
  const int plStride = X.plStrideB >> 1;
  unsigned short* pl = reinterpret_cast<unsigned short*>(X.lds + X.offPl);
  unsigned short* plB = reinterpret_cast<unsigned short*>(X.lds + X.offPlB);
  for (int m = threadIdx.x; (m + 1) * skip <= X.wcap; m += NT) {
    const unsigned short* w = monoH + m * skip;
    int sum = 0;
    for (int j = 0; j < skip; j++) sum += (int)w[j];
    for (int r = 0; r < skip; r++) {
      if ((m + 1) * skip + r > X.wcap) break;  // the last element of the higher planes needs samples past the window
      const int v = sum - bias;
      const unsigned mag = (unsigned)(v < 0 ? -v : v);
      const int qm = (int)__umulhi(mag, M);
      const unsigned short u = (unsigned short)((v < 0 ? -qm : qm) + 32768);
      pl[r * plStride + m] = u;
      if (m > 0) plB[r * plStride + m - 1] = u;
      sum += (int)w[skip + r] - (int)w[r];
    }
  }
  fast_sync();
}

// Search waves: hand the pending cross-fade (and, for FCMD_COPY, a plain copy) to the output waves.  Every command is
// followed by exactly one workgroup barrier before the next command is published, and the two slots alternate, so a
// slot is rewritten only after the output waves have consumed it.  NWC == 0: the search waves do the work themselves.
template <int NWM, int NWC>
__device__ __forceinline__ void fast_publish(FastCtx& X, int type, int cp_n, pos_t cp_src, pos_t cp_out, pos_t nb) {
  if constexpr (NWC > 0) {
    if (threadIdx.x == 64 * (NWM - 1)) {
      int* c = X.cmd + (X.seq & 1) * FCMD_INTS;
      c[0] = type; c[1] = X.xf_n; c[2] = X.xf_down; c[3] = X.xf_period;
      c[4] = X.xf_out; c[5] = cp_n; c[6] = cp_src; c[7] = cp_out;
      c[8] = X.limit; c[9] = nb;
    }
    X.seq++;
  } else {
    fast_outputs<64 * NWM>(X, threadIdx.x, X.xf_n, X.xf_down, X.xf_period, X.xf_out, cp_n, cp_src, cp_out, X.limit);
  }
  X.xf_n = 0;
}

// Per-lane constants of the search waves.
struct FastLane {
  int lane, cw;            // cw: which chunk of the sample pairs this wave sums
  int pC;                  // coarse lag of this lane
  bool validC, loneC;
  int nfullC;              // coarse: whole pairs of this lag
  int j0C, nGC, cntC;      // this wave's share: pairs [j0C, j0C + 4*nGC), of which the first cntC belong to this lag
  double scaleC;           // 65536 / pC
};

// arg-min over the lanes of diff/lag (first lag wins ties, as the dependency's sequential scan).  Returns the lane.
__device__ __forceinline__ int fast_select(unsigned dsum, double scale, bool valid, bool needResolve, int p0,
                                           unsigned& kmin) {
  const double q = __builtin_fma((double)dsum, scale, 0x1p-12);  // floor(dsum * 65536 / p) + frac; exact (header comment)
  const unsigned key = valid ? (unsigned)q : 0xffffffffu;
  kmin = wave_min_u32(key);
  unsigned long long m = __builtin_amdgcn_ballot_w64(key == kmin);
  int idx = __builtin_ctzll(m);
  m &= m - 1;
  if (needResolve && m) {  // lag products can exceed 2^16: equal keys need not be equal ratios -- exact scan of the ties
    unsigned bd = (unsigned)__builtin_amdgcn_readlane((int)dsum, idx);
    int bp = p0 + idx;
    while (m) {
      const int i = __builtin_ctzll(m);
      m &= m - 1;
      const unsigned di = (unsigned)__builtin_amdgcn_readlane((int)dsum, i);
      const int pi = p0 + i;
      if ((unsigned long long)di * (unsigned)bp < (unsigned long long)bd * (unsigned)pi) { bd = di; bp = pi; idx = i; }
    }
  }
  return idx;
}

// What a step does once its period is chosen, for every speed > 1 (libsonic skipPitchPeriod): n frames of cross-fade,
// and for 1 < speed < 2 `rem` frames copied through afterwards.  Both are exact IEEE float divisions followed by a
// truncation; they are evaluated for EVERY candidate period of the refine search (lane = candidate, lane 63 = the
// previous period) while the sums are still on their way, so that the step's chain only pays a v_readlane for them.
struct FastSpeed {
  bool ge2;
  float sm1, twom;  // speed - 1, 2 - speed
};

// findPitchPeriod at absolute position pos (search waves; every wave returns the same values).
template <int NWM, int NWC>
__device__ __forceinline__ int fast_find_period(const SpxPlanDev& P, FastCtx& X, WalkState& st, const FastLane& LN,
                                                pos_t pos, bool needResolve, const FastSpeed& SP, int& n_out,
                                                int& rem_out) {
  constexpr int NT = 64 * (NWM + NWC);
  constexpr int MAXGC = NWM >= 8 ? 2 : (NWM == 4 ? 3 : 5);   // coarse share: groups of four pairs in one flight
  constexpr int MAXGR = NWM >= 8 ? 4 : 8;                     // refine share
  const int skip = P.skip;
  FSTAMP(1);
  const int need = P.maxRequired + 2 * skip + 2;
  if (!(X.wbase >= 0 && pos >= X.wbase && pos + need <= X.wbase + X.wcap)) {
    const pos_t nb = pos & ~7;
    fast_publish<NWM, NWC>(X, FCMD_REFILL, 0, 0, 0, nb);
    if (NWC > 0) fast_sync();
    fast_refill<NT>(X, nb, X.limit);
  }
  FSTAMP(2);
  fast_publish<NWM, NWC>(X, FCMD_STEP, 0, 0, 0, 0);  // the previous step's cross-fade rides on this step's first barrier
  const int o = (int)(pos - X.wbase);
  const int lane = LN.lane;
  const int tg = st.prevPeriod_toggle & 1;
  st.prevPeriod_toggle ^= 1;
  // ---- coarse search on the decimated signal: lane = lag, this wave's share of the sample pairs ----
  int bestC;
  {
    const int oD = (o * X.skipM) >> 16;
    const int r = o - oD * skip;
    const int plr = X.offPl + r * X.plStrideB, plBr = X.offPlB + r * X.plStrideB;
    const int aoff = (oD & 1) ? plBr + 2 * (oD - 1) : plr + 2 * oD;
    const int e = oD + LN.pC;
    const int boff = (e & 1) ? plBr + 2 * (e - 1) : plr + 2 * e;
    const unsigned* ap = reinterpret_cast<const unsigned*>(X.lds + aoff);
    const unsigned* bp = reinterpret_cast<const unsigned*>(X.lds + boff);
    unsigned ah = 0, bh = 0;
    if (LN.cw == NWM - 1) { ah = ap[LN.nfullC]; bh = bp[LN.nfullC]; }  // the lone term i = p-1 of the odd lags
    unsigned d = sad_share<MAXGC, true>(ap + LN.j0C, bp + LN.j0C, LN.nGC, LN.cntC, 0u);
    if (LN.cw == NWM - 1) d = __builtin_amdgcn_sad_u16(ah & 0xffffu, (LN.loneC ? bh : ah) & 0xffffu, d);
    if (LN.validC) atomicAdd(&X.sumC[tg * 64 + lane], d);
    FSTAMP(3);
    fast_sync();
    FSTAMP(4);
    if (threadIdx.x < 64) X.sumC[(1 - tg) * 64 + lane] = 0;  // the buffer the previous step used: everyone is past it
    const unsigned dsum = X.sumC[tg * 64 + lane];
    unsigned kmin;
    bestC = fast_select(dsum, LN.scaleC, LN.validC, false, P.minPeriod / skip, kmin);
  }
  FSTAMP(5);
  // ---- refine at full rate around the coarse winner ----
  int period = (P.minPeriod / skip + bestC) * skip;
  int lo = period - (skip << 2), hi = period + (skip << 2);
  if (lo < P.minPeriod) lo = P.minPeriod;
  if (hi > P.maxPeriod) hi = P.maxPeriod;
  const int nl = hi - lo + 1;
  const bool valid = lane < nl;
  const int p = lo + lane;
  const double scale = X.inv[valid ? p : lo];
  unsigned dsum;
  int nLane, remLane;
  {
    const int CH = (((hi >> 1) + NWM) / NWM + 3) & ~3;
    const int aoff = (o & 1) ? X.offA1 + 2 * (o - 1) : X.offA0 + 2 * o;
    const int e = o + p;
    const int boff = (e & 1) ? X.offA1 + 2 * (e - 1) : X.offA0 + 2 * e;
    const unsigned* ap = reinterpret_cast<const unsigned*>(X.lds + aoff);
    const unsigned* bp = reinterpret_cast<const unsigned*>(X.lds + boff);
    const int nfull = p >> 1;
    const int j0 = LN.cw * CH;
    unsigned ah = 0, bh = 0;
    if (LN.cw == NWM - 1) { ah = ap[nfull]; bh = bp[nfull]; }
    unsigned d;
    // every valid lag has at least (lo >> 1) whole pairs: shares that end below that need no per-pair masks
    if ((lo >> 1) >= j0 + CH) d = sad_share<MAXGR, false>(ap + j0, bp + j0, CH >> 2, 0, 0u);
    else d = sad_share<MAXGR, true>(ap + j0, bp + j0, CH >> 2, nfull - j0, 0u);
    if (LN.cw == NWM - 1) d = __builtin_amdgcn_sad_u16(ah & 0xffffu, ((p & 1) ? bh : ah) & 0xffffu, d);
    if (valid) atomicAdd(&X.sumR[tg * 64 + lane], d);
    // what the step will do for each candidate period (see FastSpeed); lane 63 holds the previous period
    {
      const int pc = (lane == 63) ? st.prevPeriod : p;
      const float fp = (float)pc;
      nLane = SP.ge2 ? (int)(fp / SP.sm1) : pc;
      remLane = SP.ge2 ? 0 : (int)(fp * SP.twom / SP.sm1);
    }
    FSTAMP(6);
    fast_sync();
    FSTAMP(7);
    if (threadIdx.x < 64) X.sumR[(1 - tg) * 64 + lane] = 0;
    dsum = X.sumR[tg * 64 + lane];
  }
  unsigned kmin;
  const int best = fast_select(dsum, scale, valid, needResolve, lo, kmin);
  period = lo + best;
  const int minDiff = (int)(kmin >> 16);  // floor(diff / lag) of the winner
  FSTAMP(8);
  // Previous-period rule (libsonic prevPeriodBetter, preferNewPeriod = 1).  Only "maxDiff > 3*minDiff" is ever asked of
  // the worst lag, and max_p floor(d_p/p) = floor(max_p d_p/p), so the test is "some lag has d_p >= (3*minDiff+1)*p".
  int ret = period, sel = best;
  if (minDiff != 0 && st.prevPeriod != 0 && minDiff * 2 > st.prevMinDiff * 3) {
    const unsigned need3 = 3u * (unsigned)minDiff + 1u;
    if (__builtin_amdgcn_ballot_w64(valid && dsum >= need3 * (unsigned)p) == 0) { ret = st.prevPeriod; sel = 63; }
  }
  st.prevMinDiff = minDiff;
  st.prevPeriod = period;
  n_out = __builtin_amdgcn_readlane(nLane, sel);
  rem_out = __builtin_amdgcn_readlane(remLane, sel);
  FSTAMP(9);
  return ret;
}

// The pitch steps one event can run = the loop of libsonic's processStreamInput for speed > 1 with `avail` frames handed
// over.  The caller guarantees avail - st.base >= maxRequired.  A step that fails (n == 0) ends the event without
// removing the input consumed in this call (the dependency returns 0 there), so st.base keeps its value.
template <int NWM, int NWC>
__device__ __forceinline__ void fast_run_steps(const SpxPlanDev& P, FastCtx& X, WalkState& st, const FastLane& LN,
                                               float speed, pos_t avail, bool needResolve) {
  const int maxRequired = P.maxRequired;
  FastSpeed SP;
  SP.ge2 = speed >= 2.0f;
  SP.sm1 = speed - 1.0f;
  SP.twom = 2.0f - speed;
  pos_t pos = st.base;
  do {
    if (st.remaining > 0) {
      int n = st.remaining;
      if (n > maxRequired) n = maxRequired;
      if (st.out_n + n > X.out_cap) st.overflow = 1;
      fast_publish<NWM, NWC>(X, FCMD_COPY, n, pos, st.out_n, 0);
      if (NWC > 0) fast_sync();
      st.out_n += n;
      st.remaining -= n;
      pos += n;
    } else {
      int n, rem;
      const int period = fast_find_period<NWM, NWC>(P, X, st, LN, pos, needResolve, SP, n, rem);
      if (!SP.ge2) st.remaining = rem;
      if (st.out_n + n > X.out_cap) st.overflow = 1;
      if (n == 0) return;
      X.xf_n = n; X.xf_down = (int)(pos - X.wbase); X.xf_period = period; X.xf_out = st.out_n;
      st.out_n += n;
      pos += period + n;
      FSTAMP(10);
    }
  } while (pos + maxRequired <= avail);
  st.base = pos;
}

template <int NWM, int NWC>
__global__ void __launch_bounds__(64 * (NWM + NWC))
spx_walk_fast_kernel(SpxPlanDev P, const SpxStreamDev* __restrict__ streams, const int16_t* __restrict__ in_base,
                     int16_t* __restrict__ out_base, int64_t* __restrict__ n_out, SpxStreamState* __restrict__ states,
                     const float* scratch_base, const int* speed_ready, int wcap) {
  constexpr int NT = 64 * (NWM + NWC);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_search = wave < NWM;
  const SpxStreamDev S = streams[blockIdx.x];
  const int Ttot = S.n_frames, F = P.F, B = P.B;
  const float Rg = S.speed, nl = S.nonlinear;
  const FastLds LY = fast_lds_layout(P, wcap);

  FastCtx X;
  X.in = in_base + S.in_off;
  X.out = out_base + S.out_off;
  X.lds = lds;
  X.out_cap = (pos_t)(S.out_cap > 0x7fffffff ? 0x7fffffff : S.out_cap);
  X.limit = (pos_t)S.n_in;
  X.wbase = -1;
  X.wcap = LY.wcap;
  X.offA0 = LY.off_mono; X.offA1 = LY.off_monoB; X.offPl = LY.off_pl; X.offPlB = LY.off_plB; X.plStrideB = LY.plStrideB;
  X.skip = P.skip;
  X.skipM = (65536 + P.skip - 1) / P.skip;
  X.skipM32 = (unsigned)((0x100000000ull + (unsigned)P.skip - 1) / (unsigned)P.skip);
  X.sumC = reinterpret_cast<unsigned*>(lds + LY.off_sumC);
  X.sumR = reinterpret_cast<unsigned*>(lds + LY.off_sumR);
  X.inv = reinterpret_cast<const double*>(lds + LY.off_inv);
  X.cmd = reinterpret_cast<int*>(lds + LY.off_cmd);
  X.seq = 0;
  X.xf_n = 0; X.xf_down = 0; X.xf_period = 0; X.xf_out = 0;
  int* sWait = reinterpret_cast<int*>(lds + LY.off_wait);
  {
    double* invw = reinterpret_cast<double*>(lds + LY.off_inv);
    for (int t = tid; t <= P.maxPeriod; t += NT) invw[t] = t > 0 ? 65536.0 / (double)t : 0.0;
    for (int t = tid; t < 128; t += NT) { X.sumC[t] = 0; X.sumR[t] = 0; }
  }
  __syncthreads();

  if (!is_search) {
    // ------------------------------ output waves: obey commands until FCMD_EXIT ------------------------------
    if constexpr (NWC > 0) {
      for (;;) {
        fast_sync();  // the barrier that follows every published command
        const int* c = X.cmd + (X.seq & 1) * FCMD_INTS;
        X.seq++;
        const int type = uni(c[0]);
        const int xf_n = uni(c[1]), xf_down = uni(c[2]), xf_period = uni(c[3]), xf_out = uni(c[4]);
        const int cp_n = uni(c[5]), cp_src = uni(c[6]), cp_out = uni(c[7]);
        const pos_t limit = uni(c[8]), nb = uni(c[9]);
        fast_outputs<64 * NWC>(X, tid - 64 * NWM, xf_n, xf_down, xf_period, xf_out, cp_n, cp_src, cp_out, limit);
        if (type == FCMD_STEP) {
          fast_sync();            // the step's second barrier (refine sums complete)
        } else if (type == FCMD_REFILL) {
          fast_refill<NT>(X, nb, limit);
        } else if (type == FCMD_POLL) {
          fast_sync();            // the polled count is in LDS
        } else if (type == FCMD_EXIT) {
          break;
        }
      }
    }
    return;
  }

  // ---------------------------------------- search waves: the chain ----------------------------------------
  // the walk is the latency-critical chain: where another kernel shares a SIMD, these waves issue first
  __builtin_amdgcn_s_setprio(3);
  // the part of the stream state this stage owns (the tension kernel owns the filter states)
  SpxStreamState Z;
  if (S.flags & SPX_F_INIT) {
    Z.w.base = 0; Z.w.out_n = 0; Z.w.avail = 0; Z.w.remaining = 0; Z.w.prevPeriod = 0; Z.w.prevMinDiff = 0;
    Z.w.overflow = 0; Z.w.prevPeriod_toggle = 0; Z.w.pad_ = 0;
    Z.curSpeed = Rg;  // sonicSetSpeed -> sonicIntSetSpeed, soniclib.c:182
    Z.handed = 0;
  } else {
    Z.w = states[blockIdx.x].w;
    Z.curSpeed = states[blockIdx.x].curSpeed;
    Z.handed = states[blockIdx.x].handed;
    if (nl == 0.0f) Z.curSpeed = Rg;  // sonicSetSpeed between writes reaches the TSM stage at once (soniclib.c:182)
  }
  const float* scr = scratch_base + (size_t)S.frame_off * 4;  // per frame: ..., speed (written by the tension kernel)
  WalkState st;
  st.base = uni((pos_t)Z.w.base); st.out_n = uni((pos_t)Z.w.out_n); st.avail = uni((pos_t)Z.w.avail);
  st.remaining = uni(Z.w.remaining); st.prevPeriod = uni(Z.w.prevPeriod); st.prevMinDiff = uni(Z.w.prevMinDiff);
  st.overflow = uni(Z.w.overflow); st.prevPeriod_toggle = 0;  // both lag-sum buffers are clear at kernel start
  float tailSpeed = unif(Z.curSpeed);  // the speed in force in the TSM stage
  pos_t avail = st.avail;
  pos_t handed = (nl != 0.0f) ? uni(Z.handed) : 0;
  const bool do_flush = (S.flags & SPX_F_FLUSH) != 0;
  const bool linear = nl == 0.0f;
  const int maxRequired = P.maxRequired;
  const bool needResolve = (long)P.maxPeriod * P.maxPeriod >= 65536;

  FastLane LN;
  LN.lane = lane;
  LN.cw = (NWM == 8) ? (wave < 4 ? wave : 11 - wave) : wave;  // 8 waves: a busy and a light chunk on every SIMD
  {
    const int minC = P.minPeriod / P.skip, maxC = P.maxPeriod / P.skip;
    LN.pC = minC + lane;
    LN.validC = lane < maxC - minC + 1;
    const int CH = (((maxC >> 1) + NWM) / NWM + 3) & ~3;
    LN.nfullC = LN.validC ? (LN.pC >> 1) : 0;
    LN.j0C = LN.cw * CH;
    LN.nGC = CH >> 2;
    LN.cntC = LN.nfullC - LN.j0C;   // may be negative or beyond the share: sad_flight compares pair indices with it
    LN.loneC = LN.validC && (LN.pC & 1);
    LN.scaleC = 65536.0 / (double)LN.pC;
  }
  FSTAMP_DECL

  // The speeds come from the tension kernel.  Sequential launches (speed_ready == nullptr): all of them are there.
  // Concurrent launches: that kernel runs beside this one and publishes the number of tension frames whose speeds are
  // final (agent-scope release there; one relaxed poll + agent-scope acquire here, cdna_hip_programming.md G16).
  const int K_total = (!linear && Ttot >= F) ? Ttot - F + 1 : 0;  // soniclib.c:317
  for (;;) {
    int K = K_total;
    if (speed_ready != nullptr && K_total > 0) {
      fast_publish<NWM, NWC>(X, FCMD_POLL, 0, 0, 0, 0);
      fast_sync();
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
      fast_sync();
      K = uni(*sWait);
      if (K < 0) { st.overflow = 2; break; }  // the producer was lost (or failed): reported as its own status
      if (K > K_total) K = K_total;
    }
    const bool last = K >= K_total;
    // Events, in the order the shim issues them:
    //   nonlinear: one (setSpeed, write B) per tension frame                              soniclib.c:354,369
    //              at flush, the remaining complete ring buffers at the last speed          soniclib.c:538-550
    //   linear:    one write of everything new (soniclib.c:397-399; chunking is irrelevant at constant speed)
    //   at flush:  sonicIntFlushStream (soniclib.c:551): pad 2*maxRequired zeros, process, truncate
    const bool fin = last && do_flush;
    const pos_t ev0 = handed;
    pos_t ev1;  // one past the last ordinary event of this round
    if (!linear) ev1 = fin ? (pos_t)(S.n_in / B) : (pos_t)K;  // complete ring buffers written: soniclib.c:446-449
    else ev1 = (last && (pos_t)S.n_in > avail) ? 1 : 0;
    if (ev1 < ev0) ev1 = ev0;
    {
      // Most nonlinear events cannot run a step (a step needs maxRequired frames, an event brings B).  The speeds of 64
      // consecutive events sit in one VGPR (lane = event, tail events at the last speed), and the next event that does
      // anything -- a pass-through at unity speed, or one with enough input for a step -- is found with one ballot.
      const pos_t Kc = ev1 < (pos_t)K ? ev1 : (pos_t)K;  // tension events of this round: [ev0, Kc)
      if (Kc > ev0) tailSpeed = unif(scr[4 * (size_t)(Kc - 1) + 3]);  // what later events run at / the stream carries on
      for (pos_t blk0 = ev0; blk0 < ev1; blk0 += 64) {
        const int nIn = (int)(ev1 - blk0 < 64 ? ev1 - blk0 : 64);
        const pos_t idx = blk0 + lane;
        float spv = tailSpeed;
        if (lane < nIn && idx < (pos_t)K) spv = scr[4 * (size_t)idx + 3];
        const bool unityLane = lane < nIn && speed_is_unity(spv);
        const unsigned long long unityMask = __builtin_amdgcn_ballot_w64(unityLane);
        const pos_t availBlk = avail;
        const pos_t availLane = linear ? (pos_t)S.n_in : availBlk + (lane + 1) * B;  // frames handed over after event `lane`
        int i = 0;
        while (i < nIn) {
          const unsigned long long runnable = __builtin_amdgcn_ballot_w64(
              lane >= i && lane < nIn && (unityLane || availLane - st.base >= maxRequired));
          if (runnable == 0) break;
          i = __builtin_ctzll(runnable);
          const pos_t availE = linear ? (pos_t)S.n_in : availBlk + (i + 1) * B;
          FSTAMP(0);
          if ((unityMask >> i) & 1) {
            const pos_t n = availE - st.base;
            if (n > 0) {
              if (st.out_n + n > X.out_cap) st.overflow = 1;
              fast_publish<NWM, NWC>(X, FCMD_COPY, (int)n, st.base, st.out_n, 0);
              if (NWC > 0) fast_sync();
              st.out_n += n;
            }
            st.base = availE;
          } else {
            const float speed = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, spv), i));
            fast_run_steps<NWM, NWC>(P, X, st, LN, speed, availE, needResolve);
          }
          FSTAMP(11);
          i++;
        }
        avail = linear ? (pos_t)S.n_in : availBlk + nIn * B;
      }
      if (fin) {  // sonicIntFlushStream
        const pos_t remainingS = avail - st.base;
        const pos_t expected = st.out_n + uni((int)(((float)remainingS / tailSpeed + 0) / 1.0f + 0.5f));
        X.limit = avail;  // everything from here on reads as the flush's zero padding
        X.wbase = -1;     // the window may hold samples past the new limit: the next step refills it
        avail += 2 * maxRequired;
        if (speed_is_unity(tailSpeed)) {
          const pos_t n = avail - st.base;
          if (st.out_n + n > X.out_cap) st.overflow = 1;
          fast_publish<NWM, NWC>(X, FCMD_COPY, (int)n, st.base, st.out_n, 0);
          if (NWC > 0) fast_sync();
          st.out_n += n;
        } else if (avail - st.base >= maxRequired) {
          fast_run_steps<NWM, NWC>(P, X, st, LN, tailSpeed, avail, needResolve);
        }
        if (st.out_n > expected) st.out_n = expected;
        st.base = avail;  // the dependency empties its input after a flush
        st.remaining = 0;
      }
    }
    if (!linear) handed = ev1;
    if (last) break;
  }
  fast_publish<NWM, NWC>(X, FCMD_EXIT, 0, 0, 0, 0);
  if (NWC > 0) fast_sync();
  FSTAMP_FLUSH
  if (tid == 0) {
    Z.w.base = st.base; Z.w.out_n = st.out_n; Z.w.avail = avail; Z.w.remaining = st.remaining;
    Z.w.prevPeriod = st.prevPeriod; Z.w.prevMinDiff = st.prevMinDiff; Z.w.overflow = st.overflow;
    Z.w.prevPeriod_toggle = 0; Z.w.pad_ = 0;
    states[blockIdx.x].w = Z.w;  // field-wise: the tension kernel may be writing its own fields of this record
    states[blockIdx.x].curSpeed = tailSpeed;
    if (!linear) states[blockIdx.x].handed = (int)handed;
    // a truncated output is reported as a negative count; a lost producer as INT64_MIN (SPX_NOUT_LOST_PRODUCER)
    if (n_out) n_out[blockIdx.x] = st.overflow == 2 ? INT64_MIN : (st.overflow ? -(int64_t)st.out_n : (int64_t)st.out_n);
  }
}

size_t spx_walk_fast_lds_bytes(const SpxPlanDev& P, int wcap) { return (size_t)fast_lds_layout(P, wcap).total; }

void spx_launch_walk_fast(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, const int16_t* in,
                          int16_t* out, int64_t* n_out, SpxStreamState* states, const float* scratch,
                          const int* speed_ready, int nwm, int nwc, int wcap, hipStream_t st) {
  if (n_streams <= 0) return;
  const FastLds LY = fast_lds_layout(P, wcap);
#define SPX_LAUNCH_FAST(M, C)                                                                                         \
  hipLaunchKernelGGL((spx_walk_fast_kernel<M, C>), dim3(n_streams), dim3(64 * (M + C)), LY.total, st, P, streams, in,  \
                     out, n_out, states, scratch, speed_ready, wcap)
#ifdef SPX_STAMPS
  SPX_LAUNCH_FAST(4, 4);
  return;
#endif
  if (nwm == 8) {
    if (nwc >= 4) SPX_LAUNCH_FAST(8, 4);
    else if (nwc >= 2) SPX_LAUNCH_FAST(8, 2);
    else SPX_LAUNCH_FAST(8, 0);
  } else if (nwm == 2) {
    if (nwc >= 2) SPX_LAUNCH_FAST(2, 2);
    else if (nwc >= 1) SPX_LAUNCH_FAST(2, 1);
    else SPX_LAUNCH_FAST(2, 0);
  } else if (nwm == 1) {
    if (nwc >= 1) SPX_LAUNCH_FAST(1, 1);
    else SPX_LAUNCH_FAST(1, 0);
  } else {
    if (nwc >= 4) SPX_LAUNCH_FAST(4, 4);
    else if (nwc >= 2) SPX_LAUNCH_FAST(4, 2);
    else if (nwc >= 1) SPX_LAUNCH_FAST(4, 1);
    else SPX_LAUNCH_FAST(4, 0);
  }
#undef SPX_LAUNCH_FAST
}
