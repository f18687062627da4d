// Helpers shared by the two walk kernels (spx_walk.hip: general + multi-channel; spx_walk_fast.hip: mono speed-up).
// gfx950 only.
#pragma once
#include "spx_internal.h"

// In-kernel positions are 32-bit (the host rejects streams of 2^30 frames or more): half the SGPRs and none of the
// 64-bit add/compare sequences in the per-step bookkeeping.  The carried state record stays 64-bit.
typedef int pos_t;
struct WalkState {
  pos_t base, out_n, avail;
  int remaining, prevPeriod, prevMinDiff, overflow, prevPeriod_toggle;
  int steps;   // pitch searches of this job (diagnostic: SpxWalkState::steps)
};

// Values that are the same in every lane but that the compiler cannot prove uniform (they come from LDS or from
// lane-indexed loads): pin them to SGPRs so the bookkeeping around a pitch step runs on the scalar unit.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}

template <int NW>
__device__ __forceinline__ void lds_sync() {
  if (NW > 1) {
    // LDS-only workgroup barrier: outstanding global stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---- wave-level min / max over 64 lanes of NON-NEGATIVE floats, done on their bit patterns (same order as
// unsigned integers, and v_min_u32 / v_max_u32 take the DPP operand directly: one instruction per stage, no float
// canonicalisation).  Rows of 16 by quad_perm / mirror, then row_bcast:15 and row_bcast:31 carry the row results
// to lane 63.  Every lane gets the result (readlane 63 -> SGPR).
// Written as inline assembly because the compiler keeps a v_mov_b32_dpp + v_min pair (and a copy) per stage; the
// s_nop 1 in front of every stage is the VALU-write -> DPP-read hazard distance the assembler does not insert here.
#define SPX_WAVE_REDUCE(OP, v)                                                            \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
               "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
               "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
               "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
               "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
               "s_nop 1"                                                                  \
               : "+v"(v))
__device__ __forceinline__ float wave_min_f(float f) {
  unsigned v = __builtin_bit_cast(unsigned, f);
  SPX_WAVE_REDUCE("v_min_u32_dpp", v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ float wave_max_f(float f) {
  unsigned v = __builtin_bit_cast(unsigned, f);
  SPX_WAVE_REDUCE("v_max_u32_dpp", v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)v, 63));
}

// Sum over pairs j in [j0, j1) of |s[2j] - s[2j+p]| + |s[2j+1] - s[2j+p+1]| on biased u16 data: ap is the
// (wave-uniform) dword view of the signal at the search position, bp the lane's own dword view at lag p.
// j1 differs per lane: the EXEC mask does the predication, no per-term compare/select instructions.
__device__ __forceinline__ unsigned sad_run(const unsigned* ap, const unsigned* bp, int j0, int j1) {
  unsigned acc = 0;
  int j = j0;
  if (j + 4 <= j1) {
    // software-pipelined: the loads of group g+1 are issued before the SADs of group g, so the LDS latency of all
    // but the first group hides behind arithmetic
    unsigned a0 = ap[j], a1 = ap[j + 1], a2 = ap[j + 2], a3 = ap[j + 3];
    unsigned b0 = bp[j], b1 = bp[j + 1], b2 = bp[j + 2], b3 = bp[j + 3];
    j += 4;
    while (j + 4 <= j1) {
      const unsigned c0 = ap[j], c1 = ap[j + 1], c2 = ap[j + 2], c3 = ap[j + 3];
      const unsigned e0 = bp[j], e1 = bp[j + 1], e2 = bp[j + 2], e3 = bp[j + 3];
      acc = __builtin_amdgcn_sad_u16(a0, b0, acc);
      acc = __builtin_amdgcn_sad_u16(a1, b1, acc);
      acc = __builtin_amdgcn_sad_u16(a2, b2, acc);
      acc = __builtin_amdgcn_sad_u16(a3, b3, acc);
      a0 = c0; a1 = c1; a2 = c2; a3 = c3;
      b0 = e0; b1 = e1; b2 = e2; b3 = e3;
      j += 4;
    }
    acc = __builtin_amdgcn_sad_u16(a0, b0, acc);
    acc = __builtin_amdgcn_sad_u16(a1, b1, acc);
    acc = __builtin_amdgcn_sad_u16(a2, b2, acc);
    acc = __builtin_amdgcn_sad_u16(a3, b3, acc);
  }
  // up to three pairs are left: all loaded at once, a pair beyond the bound contributes |a - a| = 0
  if (j < j1) {
    const unsigned a0 = ap[j], a1 = ap[j + 1], a2 = ap[j + 2];
    unsigned b0 = bp[j], b1 = bp[j + 1], b2 = bp[j + 2];
    b1 = (j + 1 < j1) ? b1 : a1;
    b2 = (j + 2 < j1) ? b2 : a2;
    acc = __builtin_amdgcn_sad_u16(a0, b0, acc);
    acc = __builtin_amdgcn_sad_u16(a1, b1, acc);
    acc = __builtin_amdgcn_sad_u16(a2, b2, acc);
  }
  return acc;
}

// Exact division of small unsigned numbers (x < 2^31, 1 <= d < 2^12) without the 30-instruction integer
// division expansion: float estimate, then at most one correction each way.
__device__ __forceinline__ unsigned udiv_small(unsigned x, unsigned d) {
  // x < 2^27 (an AMDF sum) and d >= 10 (a lag), q < 2^19, so the estimate is off by at most 2: two branch-free fix-ups each
  // way in wrapping 32-bit arithmetic (the true remainder lies in (-2d, 3d), far from the wrap)
  unsigned q = (unsigned)((float)x * __builtin_amdgcn_rcpf((float)d));
  int r = (int)(x - q * d);
  const int di = (int)d;
  q -= r < 0; r += r < 0 ? di : 0;
  q -= r < 0; r += r < 0 ? di : 0;
  q += r >= di; r -= r >= di ? di : 0;
  q += r >= di; r -= r >= di ? di : 0;
  return q;
}


__device__ __forceinline__ bool speed_is_unity(float speed) {  // the dependency's pass-through test
  return !((double)speed > 1.00001 || (double)speed < 0.99999);
}

// The same sums with every operand load of the share issued before the first SAD: `nG` (wave-uniform) groups of four
// pairs starting at ap / bp; the hardware returns LDS data in issue order, so the arithmetic starts as the first
// operands arrive and the whole share costs ONE LDS round trip instead of one per group (MI355X_MICROARCH.md, LDS:
// "at low occupancy keep LDS reads in flight").  MASKED: pairs from index `cnt` (per lane) on contribute |a - a| = 0.
// Reads past a lane's own range stay inside the workgroup's LDS allocation (the arrays are followed by others).
template <int MAXG, bool MASKED>
__device__ __forceinline__ unsigned sad_flight(const unsigned* ap, const unsigned* bp, int nG, int cnt, unsigned acc) {
  unsigned a[MAXG][4], b[MAXG][4];
#pragma unroll
  for (int g = 0; g < MAXG; g++) {
    if (g < nG) {
#pragma unroll
      for (int k = 0; k < 4; k++) { a[g][k] = ap[4 * g + k]; b[g][k] = bp[4 * g + k]; }
    }
  }
#pragma unroll
  for (int g = 0; g < MAXG; g++) {
    if (g < nG) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        unsigned bb = b[g][k];
        if (MASKED) bb = (4 * g + k < cnt) ? bb : a[g][k];
        acc = __builtin_amdgcn_sad_u16(a[g][k], bb, acc);
      }
    }
  }
  return acc;
}
template <int MAXG, bool MASKED>
__device__ __forceinline__ unsigned sad_share(const unsigned* ap, const unsigned* bp, int nG, int cnt, unsigned acc) {
  for (int g0 = 0; g0 < nG; g0 += MAXG)
    acc = sad_flight<MAXG, MASKED>(ap + 4 * g0, bp + 4 * g0, nG - g0, cnt - 4 * g0, acc);
  return acc;
}

// Exact-count flights: NG groups of four pairs, all loads first, then all SADs; dispatched on the (wave-uniform) group
// count so that no flight computes groups it does not have.
template <int NG, bool MASKED>
__device__ __forceinline__ unsigned sad_flight_n(const unsigned* ap, const unsigned* bp, int cnt, unsigned acc) {
  unsigned a[NG > 0 ? NG : 1][4], b[NG > 0 ? NG : 1][4];
#pragma unroll
  for (int g = 0; g < NG; g++) {
#pragma unroll
    for (int k = 0; k < 4; k++) { a[g][k] = ap[4 * g + k]; b[g][k] = bp[4 * g + k]; }
  }
#pragma unroll
  for (int g = 0; g < NG; g++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      unsigned bb = b[g][k];
      if (MASKED) bb = (4 * g + k < cnt) ? bb : a[g][k];
      acc = __builtin_amdgcn_sad_u16(a[g][k], bb, acc);
    }
  }
  return acc;
}
template <bool MASKED>
__device__ __forceinline__ unsigned sad_groups(const unsigned* ap, const unsigned* bp, int nG, int cnt, unsigned acc) {
  while (nG > 8) {  // longer shares (high sample rates, few search waves): flights of eight groups
    acc = sad_flight_n<8, MASKED>(ap, bp, cnt, acc);
    ap += 32; bp += 32; cnt -= 32; nG -= 8;
  }
  switch (nG) {
    case 1: return sad_flight_n<1, MASKED>(ap, bp, cnt, acc);
    case 2: return sad_flight_n<2, MASKED>(ap, bp, cnt, acc);
    case 3: return sad_flight_n<3, MASKED>(ap, bp, cnt, acc);
    case 4: return sad_flight_n<4, MASKED>(ap, bp, cnt, acc);
    case 5: return sad_flight_n<5, MASKED>(ap, bp, cnt, acc);
    case 6: return sad_flight_n<6, MASKED>(ap, bp, cnt, acc);
    case 7: return sad_flight_n<7, MASKED>(ap, bp, cnt, acc);
    case 8: return sad_flight_n<8, MASKED>(ap, bp, cnt, acc);
    default: return acc;
  }
}

// NG whole groups at ap / bp, one more group at apx / bpx and one more pair at app / bpp (per-lane pointers; a lane
// without the extra group or pair passes the a pointer twice): every load first, then all SADs -- one LDS round trip.
template <int NG>
__device__ __forceinline__ unsigned sad_rect(const unsigned* ap, const unsigned* bp, const unsigned* apx, const unsigned* bpx,
                                             const unsigned* app, const unsigned* bpp, unsigned acc) {
  unsigned a[NG + 1][4], b[NG + 1][4];
#pragma unroll
  for (int g = 0; g < NG; g++) {
#pragma unroll
    for (int k = 0; k < 4; k++) { a[g][k] = ap[4 * g + k]; b[g][k] = bp[4 * g + k]; }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { a[NG][k] = apx[k]; b[NG][k] = bpx[k]; }
  const unsigned pa = *app, pb = *bpp;
#pragma unroll
  for (int g = 0; g <= NG; g++) {
#pragma unroll
    for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_sad_u16(a[g][k], b[g][k], acc);
  }
  return __builtin_amdgcn_sad_u16(pa, pb, acc);
}

// A wave-uniform range of pairs [0, n) at ap / bp, n <= 4 * NG: every load first (groups the range does not reach read
// the `a` operand twice, so they add |a - a| = 0 -- one address select per group instead of a branch), then all SADs;
// the pairs of the last, partial group are switched off by wave-uniform selects.  One LDS round trip, no dispatch.
template <int NG>
__device__ __forceinline__ unsigned sad_uniform(const unsigned* ap, const unsigned* bp, int n, unsigned acc) {
  unsigned a[NG][4], b[NG][4];
#pragma unroll
  for (int g = 0; g < NG; g++) {
    const unsigned* bg = (4 * g < n) ? bp : ap;   // uniform condition
#pragma unroll
    for (int k = 0; k < 4; k++) { a[g][k] = ap[4 * g + k]; b[g][k] = bg[4 * g + k]; }
  }
  const int tail = n & ~3;  // first pair of the partial group (if any)
#pragma unroll
  for (int g = 0; g < NG; g++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      unsigned bb = b[g][k];
      if (k > 0) bb = (4 * g == tail && 4 * g + k >= n) ? a[g][k] : bb;   // uniform: pairs past n in the partial group
      acc = __builtin_amdgcn_sad_u16(a[g][k], bb, acc);
    }
  }
  return acc;
}

// The cross-fade's quotient num / n, truncated toward zero (libsonic overlapAdd), for |num| <= 32768 n < 2^31 (the argument below
// is written for the speed-up kernels' n <= 4096; it holds as long as 2^-20 < 1 / n and the product's error stays below 2^-21):
//   q = (int)fma((double)num, inv, copysign(2^-20, num)),   inv = 1 / n to within 2^-40 relative
// Exact: a non-integer quotient is at least 1/n >= 2^-12 away from the next integer towards which the 2^-20 pushes, an integer one
// lands 2^-20 on the far side of itself, and the error of the product is below 2^27 2^-12 2^-40 = 2^-25.  So the reciprocal needs
// no IEEE division (round 5: v_rcp_f64 and two Newton steps -- five instructions for the compiler's twelve; in the lean form they
// are on the chain, once per step), the sign needs no absolute value / negate pair (the conversion truncates toward zero by
// itself), and the numerator d (n - t) + u t is two 24-bit multiplications (|d|, |u| <= 2^15, n <= 2^12).  Checked against the
// integer division for every n and every num = k n + {-1, 0, 1}: spx_debug_xfade_check (tests/test_gpu_parity.py).
// -DSPX_XFADE_V1: the round-1 sequence.
__device__ __forceinline__ double xfade_rcp(int n) {
#ifdef SPX_XFADE_V1
  return 1.0 / (double)n;
#else
  const double nd = (double)n;
  double r = __builtin_amdgcn_rcp(nd);
  r = __builtin_fma(__builtin_fma(-nd, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-nd, r, 1.0), r, r);
  return r;
#endif
}
__device__ __forceinline__ int xfade_quot(int num, double inv) {
#ifdef SPX_XFADE_V1
  const int mag = num < 0 ? -num : num;
  const int qm = (int)((double)mag * inv + 9.5367431640625e-07);
  return num < 0 ? -qm : qm;
#else
  const double x = (double)num;
  return (int)__builtin_fma(x, inv, __builtin_copysign(9.5367431640625e-07, x));
#endif
}
