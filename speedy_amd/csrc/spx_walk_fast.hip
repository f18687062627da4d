// Walk kernel for the common case -- mono streams that only ever speed up (every job speed > 1, 0 <= nonlinear <= 1),
// rates below 64 kHz (at most 64 lags in the coarse pitch search, 121 in the refine search -- more than 63 of those in the
// eight-search-wave form only): BASELINE configs[1..3] and configs[4].
// Same stage as spx_walk.hip (a10 AMDF pitch search, a11 skip + cross-fade, FIFO bookkeeping, flush; driven by the
// shim's event sequence soniclib.c:354,369,538-551), same results bit for bit.  What differs is how a pitch step is laid
// on the hardware: a stream is a chain of ~130 dependent pitch steps per second of audio, at 256 streams per GPU (one
// workgroup per CU) the length of that chain is the run time, and on gfx950 a wave issues one instruction every 4-5
// cycles whatever the instruction is (tools/ubench/issue_costs.hip: s_add 4.1, v_add 5.1, a taken branch 20, a
// v_readlane feeding the scalar unit 12+, an LDS round trip 67) -- so the chain is priced in INSTRUCTIONS per step:
//
//   * two kinds of wavefronts in a workgroup.  NWM "search" waves run the chain: the two AMDF searches, the decision,
//     the event bookkeeping -- all control flow, redundantly and uniformly.  NWC "output" waves never take part in a
//     search: they wait at the workgroup barrier (a waiting wave uses no issue slots), receive {cross-fade, copy,
//     refill, poll, exit} commands through two LDS slots and produce every output sample.  Output work is off the chain.
//   * coarse search: the lags x sample-pairs triangle is cut into groups of four pairs and dealt to the lanes of the
//     search waves ONCE, at kernel start (16 kHz: 254 groups for 256 lanes): a lane's operand offsets and byte masks are
//     constants, a step is two address computations, four ds_read2_b32, four masked v_sad_u16 and one ds_add_u32.
//   * refine search: the pairs EVERY lag of the search has form a rectangle (lags x pairs) that is dealt to all search
//     lanes as (lag, chunk of consecutive pairs) -- equal chunks, so no masks, and every operand load of a lane in flight
//     before its first SAD (one LDS round trip); the few pairs only the longer lags have are a constant ragged
//     triangle dealt once at kernel start.  Lag sums meet in LDS with ds_add_u32; the arg-min runs lane = lag.
//   * arg-min of diff/lag without floats or a resolve loop: key = floor(diff * 2^16 / lag) as an exact integer
//     (one fp64 fma against a 65536/lag table + truncation); keys order exactly like the rationals whenever
//     lag1*lag2 < 2^16 (16 kHz: 246^2) and never invert the order otherwise (ties are then resolved exactly, rarely);
//     one v_min_u32 DPP reduction, one ballot, first set bit = the lag a sequential scan would have chosen;
//     minDiff = key >> 16 comes for free.
//   * what the step does with the chosen period (n = (int)(period / (speed - 1)), an exact IEEE division) is evaluated
//     for EVERY candidate period (lane = candidate) while the refine sums are still on their way; the chain pays a
//     v_readlane for it.
//   * the next event that can do anything (most bring too little input for a step) is found with one ballot over the
//     64 events whose speeds sit in a VGPR.
//   * as in spx_walk.hip: LDS sliding window of biased u16 samples kept twice (shifted by one) so that any lag reads
//     aligned pairs for v_sad_u16, decimated planes built at refill time, partial sums met with ds_add_u32, LDS-only
//     barriers (output stores are never waited for).
//   * round 3: the loop around the step is laid out for the common case (DESIGN.md 5.3 "the step loop re-cut") -- events at
//     speed >= 2 run in a loop of their own with `ge2` a constant (hotStream), every rare path carries a branch hint, a
//     failed step is a flag in the loop condition; the window is refilled in one pass from registers (fast_refill_onepass:
//     mono / stereo, skip 4 / 5), and long jobs with a CU (or half of one) to themselves get an 8192-frame window
//     (template parameter SPEC = 1).  The same instruction stream per step otherwise; the walk kernel of the bench batch
//     2.29 -> 2.03 ms.
#include <type_traits>
#include <stdlib.h>

#include "spx_walk_common.h"

#ifndef SPX_WALK_PRIO
#define SPX_WALK_PRIO 3  // the search waves are the latency-critical chain: they issue first where another kernel shares a SIMD
#endif
// LDS bank placement of the shifted copies relative to the originals (bytes added between the two): lanes of adjacent lags
// read alternately from a signal and from its shifted copy, so the copies' bank offset decides the conflicts.
#ifndef SPX_PAD_MONO
#define SPX_PAD_MONO 48  // 16 kHz: the refine rectangle's reads become conflict-free (tools/lds_conflict_model.py: 267 -> 202 LDS cycles per step)
#endif
// Branch hints for the step loop: the compiler lays the expected side out as the fall-through (a taken branch costs a wave
// ~20 cycles, tools/ubench/issue_costs.hip).  -DSPX_NO_HINTS builds the loop as the compiler would place it by itself.
#ifdef SPX_NO_HINTS
#define SPX_LIKELY(x) (x)
#define SPX_UNLIKELY(x) (x)
#else
#define SPX_LIKELY(x) __builtin_expect(!!(x), 1)
#define SPX_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif
#ifndef SPX_PAD_PL
#define SPX_PAD_PL 0
#endif
#ifndef SPX_CT_WCAP
#define SPX_CT_WCAP 4096  // window frames of the rate-specialised kernels
#endif
#ifndef SPX_CT_WCAP_TP
#define SPX_CT_WCAP_TP 1536  // ... of their throughput instantiations (two search waves, no output waves, eight streams per CU)
#endif
#ifndef SPX_CT_WCAP_LONG
#define SPX_CT_WCAP_LONG 8192  // ... of their long-window instantiations (SPEC = 1; spx_walk_config picks the window)
#endif
#define SPX_CT_WCAP_OF(NWMV, NWCV) (((NWCV) == 0 && (NWMV) <= 2) ? SPX_CT_WCAP_TP : SPX_CT_WCAP)
enum { FCMD_STEP = 1, FCMD_COPY = 2, FCMD_REFILL = 3, FCMD_POLL = 4, FCMD_EXIT = 5 };
// Diagnostic builds only (-DSPX_PROBE_SITE=k through tools/build_variant.sh + tools/variant_times.sh; rounds 4-5: tools/slack_probe.sh, in the history): about 100 cycles of s_nop at ONE place of the step.  What
// the walk kernel's time grows by tells whether that place is on the chain (all of it shows) or in the shadow of a wait (none
// does) -- in-kernel time stamps cannot tell since the waits went: reading s_memtime drains the LDS counter and serialises
// exactly the overlap that is to be measured.  Never in the product.
#ifdef SPX_PROBE_SITE
#define SPX_PROBE(k) do { if ((k) == SPX_PROBE_SITE) asm volatile(".rept 7\n\ts_nop 15\n\t.endr" ::: "memory"); } while (0)
#else
#define SPX_PROBE(k)
#endif
#define FCMD_INTS 64  // ints per command slot: field k is written by lane k of the publishing wave
// at most this many coarse groups / ragged refine tasks per lane (22.05 kHz: 303 groups and 441 tasks over the search lanes):
// constants of the instantiation -- fewer search waves, more tasks per lane
static __host__ __device__ constexpr int fcg_of(int nwm) { return nwm == 2 ? 3 : 2; }
static __host__ __device__ constexpr int frg_of(int nwm) { return nwm == 2 ? 4 : nwm == 8 ? 5 : 2; }   // (8 search waves: up to 121 refine lags)

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
#define FSTAMP_VARS unsigned long long stamp_acc = 0, stamp_last = __builtin_readcyclecounter(), stamp_t0 = stamp_last; unsigned stamp_steps = 0;
#define FSTAMP(i)                                                        \
  do {                                                                   \
    const unsigned long long t_ = __builtin_readcyclecounter();          \
    if ((i) == SPX_STAMP_SEL) stamp_acc += t_ - stamp_last;              \
    if ((i) == 1) stamp_steps++;                                         \
    stamp_last = t_;                                                     \
  } while (0)
#define FSTAMP_FLUSH                                                     \
  if (threadIdx.x == 0 && blockIdx.x == 0) {                             \
    g_spx_fstamps[SPX_STAMP_SEL] += stamp_acc;                           \
    g_spx_fstamps[30] += __builtin_readcyclecounter() - stamp_t0;        \
    g_spx_fstamps[31] += stamp_steps;                                    \
  }
#else
#define FSTAMP_VARS
#define FSTAMP(i)
#define FSTAMP_FLUSH
#endif

// LDS layout (bytes), shared by host and device
struct FastLds {
  int off_cmd, off_wait, off_sumC, off_sumR, off_sumS, off_inv, off_mono, off_monoB, off_pl, off_plB, plStrideB, total, wcap;
  int off_sumW;   // refine searches of more than 64 lags (rates from 32 kHz: 8 skip + 1 lags): two buffers of 128 sums + spare words;
                  // or (round 5) COARSE searches of more than 64 lags (rates of about 9.8 - 12 and 14.7 - 16 kHz: 11.025 kHz has 72)
};
// more than one coarse lag per lane of the coarse select (then two: fast_wide_coarse instantiations, SPEC = 2)
static __host__ __device__ inline bool fast_wide_coarse(int minPeriod, int maxPeriod, int skip) {
  return skip > 0 && (maxPeriod / skip - minPeriod / skip + 1) > 64;
}
static __host__ __device__ inline FastLds fast_lds_layout_i(int minPeriod, int maxPeriod, int skip_, int wcap) {
  FastLds L;
  L.wcap = wcap;
  int o = 0;
  L.off_cmd = o; o += 2 * FCMD_INTS * 4;
  L.off_wait = o; o += 16;
  L.off_sumC = o; o += 2 * 64 * 4;
  L.off_sumR = o; o += 2 * 64 * 4;
  L.off_sumS = o; o += 2 * 64 * 4;   // spare (was: speculative refine sums; kept so that no LDS offset moves)
  L.off_inv = o; o += ((maxPeriod + 2) * 8 + 15) & ~15;
  const int mb = ((wcap + 8) * 2 + 15) & ~15;
  L.off_mono = o; o += mb + SPX_PAD_MONO;
  L.off_monoB = o; o += mb;
  const int skip = skip_ > 0 ? skip_ : 1;
  const int plStride = ((wcap / skip + 4) + 1) & ~1;  // elements per plane (even)
  L.plStrideB = plStride * 2;
  const int plb = (plStride * skip * 2 + 15) & ~15;
  L.off_pl = o; o += plb + SPX_PAD_PL;
  L.off_plB = o; o += plb;
  L.off_sumW = o;
  if (skip >= 6 || fast_wide_coarse(minPeriod, maxPeriod, skip)) o += 512 * 4;   // (the rates the eight-search-wave form serves, or the wide coarse select) behind everything else: no other offset moves
  L.total = o;
  return L;
}
static __host__ __device__ inline FastLds fast_lds_layout(const SpxPlanDev& P, int wcap) {
  return fast_lds_layout_i(P.minPeriod, P.maxPeriod, P.skip, wcap);
}

// What the output side needs to know about the stream.
struct FastOut {
  const int16_t* in;   // interleaved, indexed by (TSM position) * C + channel
  int16_t* out;
  unsigned char* lds;
  pos_t out_cap;
  int offA0;
  int C;               // channels of this stream (1 .. 8)
};

// One LDS word, read again on every call (a spin loop's probe): ds_read_b32 on the word's LDS offset -- the low half of
// its generic address.  (A volatile C++ load through the generic pointer compiles to a system-coherent FLAT load.)
__device__ __forceinline__ int lds_probe(const int* p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ void fast_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  SPX_WAVE_REDUCE("v_min_u32_dpp", v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// (The cross-fade's quotient -- xfade_rcp, xfade_quot -- lives in spx_walk_common.h: the general walk kernel uses it too.)
// Its numerator d (n - t) + u t as two 24-bit multiplications (|d|, |u| <= 2^15, n <= 2^12):
__device__ __forceinline__ int xfade_num(int d, int nt, int u, int t) {
#ifdef SPX_XFADE_V1
  return d * nt + u * t;
#else
  return __mul24(d, nt) + __mul24(u, t);
#endif
}
// ... from the window's BIASED samples D = d + 32768, U = u + 32768 (what the LDS window holds): d (n - t) + u t =
// D (n - t) + U t - 32768 n -- two unsigned 24-bit multiplications on the values as they are read and one subtraction of a
// wave-uniform constant, instead of two bias subtractions and two signed multiplications
__device__ __forceinline__ int xfade_num_biased(unsigned D, int nt, unsigned U, int t, int n) {
#ifdef SPX_XFADE_V1
  return ((int)D - 32768) * nt + ((int)U - 32768) * t;
#else
  return (int)(__umul24(D, (unsigned)nt) + __umul24(U, (unsigned)t)) - (n << 15);
#endif
}
__global__ void spx_xfade_check_kernel(int n_lo, int n_hi, unsigned* mismatches) {
  // block = one n; threads stride over k = -32768 .. 32768, num = k n + {-1, 0, 1} clipped to |num| <= 32768 n
  const int n = n_lo + (int)blockIdx.x;
  if (n > n_hi) return;
  const double inv = xfade_rcp(n);
  const long long lim = 32768ll * n;
  unsigned bad = 0;
  for (int k = -32768 + (int)threadIdx.x; k <= 32768; k += (int)blockDim.x)
    for (int e = -1; e <= 1; e++) {
      const long long v = (long long)k * n + e;
      if (v < -lim || v > lim) continue;
      const int num = (int)v;
      bad += (xfade_quot(num, inv) != num / n);
    }
  // the numerator's two forms on a few operands per thread
  for (int t = (int)threadIdx.x; t <= n; t += (int)blockDim.x) {
    const int d = 32767 - ((17 * t) & 0xffff), u = -32768 + ((23 * t + 7 * n) & 0xffff);   // both over the whole int16 range
    bad += (xfade_num(d, n - t, u, t) != d * (n - t) + u * t);
    bad += (xfade_num_biased((unsigned)(d + 32768), n - t, (unsigned)(u + 32768), t, n) != d * (n - t) + u * t);
  }
  if (bad) atomicAdd(mismatches, bad);
}
extern "C" long long spx_debug_xfade_check(int n_lo, int n_hi) {
  if (n_lo < 1 || n_hi > 4096 || n_hi < n_lo) return -1;
  unsigned* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned)) != hipSuccess) return -1;
  (void)hipMemset(d, 0, sizeof(unsigned));
  hipLaunchKernelGGL(spx_xfade_check_kernel, dim3(n_hi - n_lo + 1), dim3(256), 0, nullptr, n_lo, n_hi, d);
  unsigned h = 0;
  const hipError_t e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? (long long)h : -1;
}

// (Round 5, tried and dropped: the forms without output waves preparing a cross-fade when it is DECIDED -- the reciprocal, and the
// thread's two window samples loaded from LDS, at the end of the step before -- so that the step which performs it on the chain
// finds them ready.  Bit-equal, 1 % SLOWER in the pipelined loop (0.983 against 0.974 ms per step): four more live registers per
// lane across the whole step and the reciprocal in every wave cost more than the two round trips they hide.)
// Every output sample of the stream is produced here: by the NWC output waves on command, or -- NWC == 0 -- by the search
// waves themselves.  `t0` = index of this thread among the NTO threads doing output work.
//   cross-fade (libsonic overlapAdd): out[t] = (down[t]*(n-t) + up[t]*t)/n, integer, truncating toward zero; both runs
//     lie in the LDS window.  The quotient: xfade_quot above.
//   copy: n frames from absolute input position src (straight from HBM, coalesced; beyond `limit` = flush padding = 0).
template <int NTO, bool MC>
__device__ __forceinline__ void fast_outputs(const FastOut& X, int t0, int xf_n, int xf_down, int xf_period, pos_t xf_out,
                                             int cp_n, pos_t cp_src, pos_t cp_out, pos_t limit, pos_t wbase) {
  const int C = MC ? X.C : 1;  // MC = false: the mono-only instantiation (no channel arithmetic at all)
  if (xf_n > 0) {
    const double inv = xfade_rcp(xf_n);
    pos_t room = X.out_cap - xf_out;
    const int nv = room > xf_n ? xf_n : (room < 0 ? 0 : (int)room);
    if (C == 1) {
      int16_t* __restrict__ dst = X.out + (size_t)xf_out;
      const unsigned short* wd = reinterpret_cast<const unsigned short*>(X.lds + X.offA0) + xf_down;
      const unsigned short* wu = wd + xf_period;
      // (a handful of iterations at most -- n <= maxPeriod < 1024 -- so no interleaved copy of the loop: it costs the forms that do
      // their own output work registers, and every form code)
#ifndef SPX_XFADE_V1
#pragma clang loop unroll(disable) interleave(disable) vectorize(disable)
#endif
      for (int t = t0; t < nv; t += NTO) {
        dst[t] = (int16_t)xfade_quot(xfade_num_biased(wd[t], xf_n - t, wu[t], t, xf_n), inv);
      }
    } else {
      // several channels: the window holds the channel mean only; the two ramps come from the input itself (these waves
      // are off the chain, the samples were read moments ago), element e = t*C + c, zeros past `limit`
      int16_t* __restrict__ dst = X.out + (size_t)xf_out * C;
      const pos_t ad = wbase + xf_down, au = ad + xf_period;  // absolute first frames of the two ramps
      const int16_t* __restrict__ rd = X.in + (size_t)ad * C;
      const int16_t* __restrict__ ru = X.in + (size_t)au * C;
      const int total = nv * C;
      const unsigned invC = (0x10000u + (unsigned)C - 1u) / (unsigned)C;  // e / C for e < 8192, C <= 8
      const int realD = (int)((limit - ad) * C), realU = (int)((limit - au) * C);  // elements of real input (may be <= 0)
#ifndef SPX_XFADE_V1
#pragma clang loop unroll(disable) interleave(disable) vectorize(disable)
#endif
      for (int e = t0; e < total; e += NTO) {
        const int t = (C == 2) ? (e >> 1) : (int)(((unsigned)e * invC) >> 16);
        const int d = (e < realD) ? (int)rd[e] : 0, u = (e < realU) ? (int)ru[e] : 0;
        dst[e] = (int16_t)xfade_quot(xfade_num(d, xf_n - t, u, t), inv);
      }
    }
  }
  if (cp_n > 0) {
    pos_t room = X.out_cap - cp_out;
    const int nv = room > cp_n ? cp_n : (room < 0 ? 0 : (int)room);
    int16_t* __restrict__ dst = X.out + (size_t)cp_out * C;
    const int16_t* __restrict__ src = X.in + (size_t)cp_src * C;
    const pos_t real = (limit - cp_src) * C;  // elements of real input from cp_src on (may be <= 0: all padding)
    const int total = nv * C;
    for (int e = t0; e < total; e += NTO) dst[e] = (e < real) ? src[e] : (int16_t)0;
  }
}

// Load the window [nb, nb + wcap] (biased u16, plus the copy shifted by one sample) and build the decimated planes:
// plane r (r < skip) holds S[m*skip + r], S[i] = truncated mean of window samples i .. i+skip-1, so the decimated
// search signal of a step at window offset o is plane (o % skip) from element o / skip on, contiguous.  All NT threads
// of the workgroup; three LDS barriers.
// The refill of a mono or stereo window at skip 4 or 5 (16 / 22.05 kHz) whose frames are all real input, 4-byte aligned in
// HBM, in ONE pass between two barriers: thread t takes the 2*skip window frames from 2*skip*t on and the 2*skip - 1 behind
// them as dword loads (all in flight at once) and writes, from registers, its share of all four images -- the window and its
// copy shifted by one frame (skip pairs each: one ds_write_b128 each at skip 4), and elements m = 2t, 2t+1 (shifted copies:
// 2t+1, 2t+2) of every decimated plane as one ds_write_b32 each.  The arithmetic of the general code below (window = the
// frame, stereo: trunc((L + R) / 2); plane element = trunc(sum of skip*C raw samples / (skip*C))), and the same LDS contents
// wherever a search may read; the few entries past the window that the general code leaves unwritten get their true values.
// (General code, mono: sixteen 2-byte LDS writes per thread, a barrier, then the planes from fourteen 2-byte LDS reads and
// sixteen 2-byte writes per thread -- 5 700 cycles per refill, 5 % of a 16 kHz chain; the walk kernel of the bench batch
// 2.10 -> 2.06 ms with this pass, profiles/r03/r03z_refill.txt.)
// (Round 5, tried and dropped: both rounds of the threads' loads issued before the first round is used -- 256 threads cover a
// 4096-frame window in two rounds -- 1 % SLOWER in the pipelined loop and 3 % on the stereo forms: eight more registers in the
// refill, spilled in the capped forms, for a latency the step loop does not wait on as long as assumed.)
template <int NT, int SKIP, bool STEREO>
__device__ __forceinline__ void fast_refill_onepass(const FastOut& X, const FastLds& LY, pos_t nb) {
  constexpr int NF = 4 * SKIP - 1;                    // frames a thread needs
  constexpr int ND = STEREO ? NF : (NF + 1) / 2;      // dwords holding them
  constexpr int DIV = SKIP * (STEREO ? 2 : 1);
  constexpr unsigned M = (unsigned)((0x100000000ull + DIV - 1) / DIV);
  const int wcap = LY.wcap;
  const unsigned* __restrict__ src = reinterpret_cast<const unsigned*>(X.in + (size_t)nb * (STEREO ? 2 : 1));
  const int plStrideB = LY.plStrideB;
  for (int t = threadIdx.x; 2 * SKIP * t < wcap; t += NT) {
    unsigned w[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) w[i] = src[(STEREO ? 2 * SKIP : SKIP) * t + i];
    int fs[NF];      // raw channel sums per frame
#pragma unroll
    for (int i = 0; i < NF; i++) {
      if (STEREO) fs[i] = (int)(short)(w[i] & 0xffffu) + ((int)w[i] >> 16);
      else fs[i] = (i & 1) ? ((int)w[i >> 1] >> 16) : (int)(short)(w[i >> 1] & 0xffffu);
    }
    // the window (u16 biased by 32768) and its copy shifted by one frame: SKIP pairs each
    unsigned m0[SKIP], m1[SKIP];
#pragma unroll
    for (int k = 0; k < SKIP; k++) {
      if (STEREO) {
        const unsigned a = (unsigned)(fs[2 * k] / 2 + 32768) & 0xffffu, b = (unsigned)(fs[2 * k + 1] / 2 + 32768) & 0xffffu,
                       c = (unsigned)(fs[2 * k + 2] / 2 + 32768) & 0xffffu;
        m0[k] = a | (b << 16);
        m1[k] = b | (c << 16);
      } else {
        m0[k] = w[k] ^ 0x80008000u;
        m1[k] = __builtin_amdgcn_alignbit(w[k + 1], w[k], 16) ^ 0x80008000u;
      }
    }
    unsigned* d0 = reinterpret_cast<unsigned*>(X.lds + LY.off_mono + 4 * SKIP * t);
    unsigned* d1 = reinterpret_cast<unsigned*>(X.lds + LY.off_monoB + 4 * SKIP * t);
    if (SKIP == 4) {
      *reinterpret_cast<uint4*>(d0) = make_uint4(m0[0], m0[1], m0[2], m0[3]);
      *reinterpret_cast<uint4*>(d1) = make_uint4(m1[0], m1[1], m1[2], m1[3]);
    } else {
#pragma unroll
      for (int k = 0; k < SKIP; k++) { d0[k] = m0[k]; d1[k] = m1[k]; }
    }
    // the 3*SKIP sums of SKIP consecutive frames: element m = 2t + a of plane r is P[a*SKIP + r]
    int P[3 * SKIP];
    P[0] = 0;
#pragma unroll
    for (int j = 0; j < SKIP; j++) P[0] += fs[j];
#pragma unroll
    for (int i = 1; i < 3 * SKIP; i++) P[i] = P[i - 1] + fs[i + SKIP - 1] - fs[i - 1];
    unsigned u[3 * SKIP];
#pragma unroll
    for (int i = 0; i < 3 * SKIP; i++) {
      const int v = P[i];
      const unsigned mag = (unsigned)(v < 0 ? -v : v);
      const int qm = (int)__umulhi(mag, M);              // the truncating division by skip * C (exact: |v| < 2^21)
      u[i] = (unsigned)((v < 0 ? -qm : qm) + 32768) & 0xffffu;
    }
#pragma unroll
    for (int r = 0; r < SKIP; r++) {
      *reinterpret_cast<unsigned*>(X.lds + LY.off_pl + r * plStrideB + 4 * t) = u[r] | (u[SKIP + r] << 16);
      *reinterpret_cast<unsigned*>(X.lds + LY.off_plB + r * plStrideB + 4 * t) = u[SKIP + r] | (u[2 * SKIP + r] << 16);
    }
  }
  fast_sync();
}

template <int NT, bool MC>
__device__ __forceinline__ void fast_refill(const FastOut& X, const FastLds& LY, int skip, pos_t nb, pos_t limit) {
  fast_sync();  // everyone is done reading the old window
#ifndef SPX_NO_FAST_REFILL
  {
    const int C = MC ? X.C : 1;
    if (C <= 2 && (skip == 4 || skip == 5) && limit - nb >= LY.wcap + 4 * skip && ((size_t)(X.in + (size_t)nb * C) & 3) == 0) {
      if (C == 1) { if (skip == 4) fast_refill_onepass<NT, 4, false>(X, LY, nb); else fast_refill_onepass<NT, 5, false>(X, LY, nb); }
      else if constexpr (MC) { if (skip == 4) fast_refill_onepass<NT, 4, true>(X, LY, nb); else fast_refill_onepass<NT, 5, true>(X, LY, nb); }
      return;
    }
  }
#endif
  const int wcap = LY.wcap;
  const int C = MC ? X.C : 1;
  unsigned short* monoH = reinterpret_cast<unsigned short*>(X.lds + LY.off_mono);
  unsigned short* monoHB = reinterpret_cast<unsigned short*>(X.lds + LY.off_monoB);
  const pos_t room = limit - nb;  // frames of real input from nb on
  const int last = (int)(room < wcap + 1 ? room : wcap + 1) - 1;  // last index holding real input
  const int plStride = LY.plStrideB >> 1;
  unsigned short* pl = reinterpret_cast<unsigned short*>(X.lds + LY.off_pl);
  unsigned short* plB = reinterpret_cast<unsigned short*>(X.lds + LY.off_plB);
  if (C == 1) {
    const int16_t* __restrict__ src = X.in + nb;
    for (int k0 = threadIdx.x; k0 < wcap + 1; k0 += 8 * NT) {
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
        if (k < wcap) monoH[k] = w;
        if (k > 0 && k < wcap + 1) monoHB[k - 1] = w;
      }
    }
    fast_sync();
    // One thread per decimated index m: it reads the 2*skip-1 window samples m*skip .. m*skip+2*skip-2 once and slides
    // the sum over them, giving element m of every plane.  |sum| < 2^18 and skip <= 7, so the truncating division is
    // exactly mulhi(|sum|, ceil(2^32 / skip)).
    const unsigned M = (unsigned)((0x100000000ull + (unsigned)skip - 1) / (unsigned)skip);
    const int bias = 32768 * skip;
    for (int m = threadIdx.x; (m + 1) * skip <= wcap; m += NT) {
      const unsigned short* w = monoH + m * skip;
      int sum = 0;
      for (int j = 0; j < skip; j++) sum += (int)w[j];
      for (int r = 0; r < skip; r++) {
        if ((m + 1) * skip + r > wcap) break;  // the last element of the higher planes needs samples past the window
        const int v = sum - bias;
        const unsigned mag = (unsigned)(v < 0 ? -v : v);
        const int qm = (int)__umulhi(mag, M);
        const unsigned short u = (unsigned short)((v < 0 ? -qm : qm) + 32768);
        pl[r * plStride + m] = u;
        if (m > 0) plB[r * plStride + m - 1] = u;
        sum += (int)w[skip + r] - (int)w[r];
      }
    }
  } else if (C == 2 && skip <= 6) {
    // Stereo, the common multi-channel case: the arithmetic of the general branch below with every global load of a thread
    // in flight before its first use (the general branch walks frame by frame and channel by channel, one dependent load
    // after the other: 17 us per refill at 16 kHz stereo against 3 us for mono -- a sixth of such a stream's chain).
    const int16_t* __restrict__ src = X.in + (size_t)nb * 2;
    for (int k0 = threadIdx.x; k0 < wcap + 1; k0 += 4 * NT) {
      int a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * NT;
        const int kk = k < last ? k : last;   // clamped address, unconditional load (last >= 0 checked below)
        if (last >= 0) { a[u] = (int)src[2 * (size_t)kk]; b[u] = (int)src[2 * (size_t)kk + 1]; } else { a[u] = 0; b[u] = 0; }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * NT;
        const int sum = (k <= last) ? a[u] + b[u] : 0;
        const unsigned short w = (unsigned short)(sum / 2 + 32768);
        if (k < wcap) monoH[k] = w;
        if (k > 0 && k < wcap + 1) monoHB[k - 1] = w;
      }
    }
    const int div = skip * 2;
    const unsigned M = (unsigned)((0x100000000ull + (unsigned)div - 1) / (unsigned)div);
    for (int m = threadIdx.x; (m + 1) * skip <= wcap; m += NT) {
      const int f0 = m * skip;
      int fs[11];   // raw channel sums of frames f0 .. f0 + 2*skip - 2 (zero past the input)
#pragma unroll
      for (int j = 0; j < 11; j++) {
        fs[j] = 0;
        if (j < 2 * skip - 1) {
          const int f = f0 + j;
          const int ff = f < last ? f : last;
          if (last >= 0) {
            const int v = (int)src[2 * (size_t)ff] + (int)src[2 * (size_t)ff + 1];
            fs[j] = (f <= last) ? v : 0;
          }
        }
      }
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 6; j++) if (j < skip) sum += fs[j];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        if (r < skip && (m + 1) * skip + r <= wcap) {
          const unsigned mag = (unsigned)(sum < 0 ? -sum : sum);
          const int qm = (int)__umulhi(mag, M);
          const unsigned short u = (unsigned short)((sum < 0 ? -qm : qm) + 32768);
          pl[r * plStride + m] = u;
          if (m > 0) plB[r * plStride + m - 1] = u;
          int add = 0;   // fs[skip + r] - fs[r] with constant indices only (the array stays in registers)
#pragma unroll
          for (int j = 0; j < 11; j++) add += (j == skip + r ? fs[j] : 0) - (j == r ? fs[j] : 0);
          sum += add;
        }
      }
    }
  } else {
    // Several channels.  Search signal at full rate: the channel mean, truncated (the dependency's downSampleInput with
    // skip 1).  Decimated planes: skip*C RAW samples summed and divided ONCE by skip*C -- not the mean of the per-frame
    // channel means (|sum| < 2^21, skip*C <= 56: mulhi is still exact).  The raw samples are not kept in LDS: both
    // passes read the input (the second one out of the cache), the cross-fades later too (fast_outputs).
    const int16_t* __restrict__ src = X.in + (size_t)nb * C;
    for (int k = threadIdx.x; k < wcap + 1; k += NT) {
      int sum = 0;
      if (k <= last)
        for (int c = 0; c < C; c++) sum += (int)src[(size_t)k * C + c];
      const unsigned short u = (unsigned short)(sum / C + 32768);
      if (k < wcap) monoH[k] = u;
      if (k > 0) monoHB[k - 1] = u;
    }
    const int div = skip * C;
    const unsigned M = (unsigned)((0x100000000ull + (unsigned)div - 1) / (unsigned)div);
    for (int m = threadIdx.x; (m + 1) * skip <= wcap; m += NT) {
      const int f0 = m * skip;  // first frame of the decimated sample (window-relative)
      int sum = 0;
      for (int j = 0; j < skip; j++)
        if (f0 + j <= last)
          for (int c = 0; c < C; c++) sum += (int)src[(size_t)(f0 + j) * C + c];
      for (int r = 0; r < skip; r++) {
        if ((m + 1) * skip + r > wcap) break;
        const unsigned mag = (unsigned)(sum < 0 ? -sum : sum);
        const int qm = (int)__umulhi(mag, M);
        const unsigned short u = (unsigned short)((sum < 0 ? -qm : qm) + 32768);
        pl[r * plStride + m] = u;
        if (m > 0) plB[r * plStride + m - 1] = u;
        // slide by one frame: drop frame f0 + r, take frame f0 + skip + r (zeros past the input)
        for (int c = 0; c < C; c++) {
          const int fa = f0 + skip + r, fd = f0 + r;
          sum += ((fa <= last) ? (int)src[(size_t)fa * C + c] : 0) - ((fd <= last) ? (int)src[(size_t)fd * C + c] : 0);
        }
      }
    }
  }
  fast_sync();
}

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

// The same over two lags per lane (lags lane and 64 + lane; refine searches of more than 64 lags).  Returns the lag index.
__device__ __forceinline__ int fast_select2(unsigned dsum, unsigned dsum2, double scale, double scale2, bool valid, bool valid2,
                                            bool needResolve, int p0, unsigned& kmin) {
  const double q = __builtin_fma((double)dsum, scale, 0x1p-12), q2 = __builtin_fma((double)dsum2, scale2, 0x1p-12);
  const unsigned key = valid ? (unsigned)q : 0xffffffffu, key2 = valid2 ? (unsigned)q2 : 0xffffffffu;
  kmin = wave_min_u32(key < key2 ? key : key2);
  unsigned long long m = __builtin_amdgcn_ballot_w64(key == kmin), m2 = __builtin_amdgcn_ballot_w64(key2 == kmin);
  int idx = m ? __builtin_ctzll(m) : 64 + __builtin_ctzll(m2);
  if (m) m &= m - 1; else m2 &= m2 - 1;
  if (needResolve && (m | m2)) {  // exact scan of the ties, in lag order
    unsigned bd = idx < 64 ? (unsigned)__builtin_amdgcn_readlane((int)dsum, idx) : (unsigned)__builtin_amdgcn_readlane((int)dsum2, idx - 64);
    int bp = p0 + idx;
    while (m) {
      const int i = __builtin_ctzll(m);
      m &= m - 1;
      const unsigned di = (unsigned)__builtin_amdgcn_readlane((int)dsum, i);
      const int pi = p0 + i;
      if ((unsigned long long)di * (unsigned)bp < (unsigned long long)bd * (unsigned)pi) { bd = di; bp = pi; idx = i; }
    }
    while (m2) {
      const int i = __builtin_ctzll(m2);
      m2 &= m2 - 1;
      const unsigned di = (unsigned)__builtin_amdgcn_readlane((int)dsum2, i);
      const int pi = p0 + 64 + i;
      if ((unsigned long long)di * (unsigned)bp < (unsigned long long)bd * (unsigned)pi) { bd = di; bp = pi; idx = 64 + i; }
    }
  }
  return idx;
}

// lane `LANE` of `old` replaced by the wave-uniform value v (v_writelane_b32: no EXEC change)
template <typename T>
__device__ __forceinline__ int fast_writelane_impl(int v, int old, T) { return old; }
#define fast_writelane(V, LANE, OLD) ([&] { int o_ = (OLD); const int v_ = uni((int)(V)); \
  asm volatile("v_writelane_b32 %0, %1, " #LANE : "+v"(o_) : "s"(v_)); return o_; }())

// A masked pair of a SAD: both operands ANDed with the mask.  (Round 5 tried ONE v_bfi_b32 in front of the v_sad_u16 instead -- the
// halves the mask switches off read `a` on both sides, four instructions fewer per coarse group and one per ragged task: bit-equal
// and 0.2 - 1 % SLOWER in both launch orders, profiles/r05/r5u_bfi_ab.txt: the VOP3 encoding is eight bytes where v_and is four, and
// the step loop's speed follows its fetch lines more than its instruction count.  -DSPX_SAD_BFI builds it.)
#ifdef SPX_SAD_BFI
__device__ __forceinline__ unsigned sad_masked_b(unsigned m, unsigned b, unsigned a) {
  unsigned r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(b), "v"(a));
  return r;
}
#define SPX_SAD_MASKED(M, A, B, ACC) __builtin_amdgcn_sad_u16((A), sad_masked_b((M), (B), (A)), (ACC))
#else
#define SPX_SAD_MASKED(M, A, B, ACC) __builtin_amdgcn_sad_u16((A) & (M), (B) & (M), (ACC))
#endif

// byte address of the aligned dword holding elements (e, e+1) of a u16 array kept twice, the second copy shifted by one
// element: base + 2e for even e, (base of the shifted copy - 2) + 2e for odd e;  d2 = shiftedBase - 2 - base
__device__ __forceinline__ int pair_addr(int base, int d2, int e) { return base + 2 * e + (e & 1) * d2; }

// n / d for the candidate step lengths of a pitch step, WITHOUT the scaling and fix-up halves of the IEEE division sequence.
// The compiler's expansion of a float division is: v_div_scale x 2, v_rcp, two fma for the reciprocal, a multiplication, two
// rounds of (residual, correction) and v_div_fmas / v_div_fixup -- twelve instructions per step on the chain.  For the operands
// that occur here -- n a sample count up to a few thousand (or that times 2 - speed), d = speed - 1 between 1e-5 and 1e18
// (SPX_FAST_MAX_SPEED), nowhere near the exponent range's ends -- v_div_scale returns its operands unchanged, v_div_fmas is a
// plain fma and v_div_fixup returns the quotient as it is: what remains is the SAME arithmetic, and the reciprocal half of it
// depends on the event's speed only.  fast_rcp_refined once per event, fast_div per step: five instructions (walk kernel of the
// bench batch 1.687 -> 1.671 ms).  Bit-identical to the IEEE quotient for d in [2^-60, 2^80)
// (tests/test_gpu_parity.py::test_fast_division_equals_the_ieee_quotient, through spx_debug_fdiv_check below).
__device__ __forceinline__ float fast_rcp_refined(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}
__device__ __forceinline__ float fast_div(float n, float d, float r) {
  float q = n * r;
  float e = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
__global__ void spx_fdiv_check_kernel(unsigned seed, unsigned trials, int exp_lo, int exp_hi, unsigned* mismatches) {
  // thread = one denominator per trial; every count 1 .. 4096 against it, both numerator forms of a step
  unsigned x = seed * 2654435761u + (blockIdx.x * blockDim.x + threadIdx.x) * 40503u + 12345u;
  unsigned bad = 0;
  for (unsigned t = 0; t < trials; t++) {
    x = x * 1664525u + 1013904223u;
    // speed - 1: a random float in [2^exp_lo, 2^exp_hi), log-uniform exponent, random mantissa
    const unsigned ex = (unsigned)(127 + exp_lo) + (x >> 24) % (unsigned)(exp_hi - exp_lo);
    const float d = __builtin_bit_cast(float, (ex << 23) | (x & 0x7fffffu));
    const float twom = 1.0f - d;      // 2 - speed
    const float r = fast_rcp_refined(d);
    for (int n = 1; n <= 4096; n++) {
      const float fn = (float)n;
      bad += (fn / d != fast_div(fn, d, r));
      const float m = fn * twom;
      bad += (m / d != fast_div(m, d, r));
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}
extern "C" long long spx_debug_fdiv_check(unsigned seed, unsigned denominators, int exp_lo, int exp_hi) {
  if (exp_lo < -126 || exp_hi > 127 || exp_hi <= exp_lo) return -1;
  unsigned* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned)) != hipSuccess) return -1;
  (void)hipMemset(d, 0, sizeof(unsigned));
  const unsigned threads = 256, blocks = 1024;
  const unsigned trials = (denominators + threads * blocks - 1) / (threads * blocks);
  hipLaunchKernelGGL(spx_fdiv_check_kernel, dim3(blocks), dim3(threads), 0, nullptr, seed, trials, exp_lo, exp_hi, d);
  unsigned h = 0;
  const hipError_t e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? (long long)h : -1;
}

// RATE != 0: the kernel is compiled for that sample rate and a 4096-frame window -- every LDS offset, period limit and
// divisor an immediate, which frees some thirty scalar registers in the step loop; RATE == 0 takes them from the plan.
// At most 96 VGPRs: in concurrent mode a SIMD holds two waves of this kernel (a search and an output wave), one of the
// tension kernel (56 registers) and analysis waves of 128 -- with 96 here two of those fit in the 512-register file,
// with the 97 the compiler would take by itself only one (and the analysis then runs at a third of its speed: measured).
// SPEC: 0 = the usual window (SPX_CT_WCAP_OF), 1 = the LONG window of SPX_CT_WCAP_LONG frames for long jobs that have a CU
// to themselves or share it with one other stream (rate-specialised instantiations with output waves only): half as many
// refills, 68 KB of LDS instead of 36 -- two analysis workgroups still fit beside it at 16 kHz.  Walk kernel of the bench
// batch 2.058 -> 2.035 ms (profiles/r03/r03ab_w8k.txt).
// (The parameter's name is history: round 2 tried speculative refine searches on the output waves -- the previous step's coarse winner
// predicts this step's refine window in 30-60 % of steps -- bit-exact and SLOWER, 2.44 against 2.06 ms: DESIGN.md 5.3; the
// protocol lived here behind this parameter until round 3 and is in the history, commit dd0437d and before.)
// MC: 0 = mono streams only (the instantiation of the bench), 1 = any channel count up to 8 per stream; + 2 (round 5): the
// instantiation also serves events at speeds BELOW 1 (libsonic's insertPitchPeriod) -- a batch with slow-down
// jobs no longer falls back to the general kernel.  The search is the same; what differs is what a step does with its period
// (run_event_slow below).  Instantiations without the bit compile to the code they always had.
template <int NWM, int NWC, int RATE, int SPEC, int MC>
#ifndef SPX_TP_WAVES
#define SPX_TP_WAVES 4   // minimum waves per SIMD the throughput instantiations (NWC == 0) are compiled for
#endif
__global__ void __launch_bounds__(64 * (NWM + NWC)) __attribute__((amdgpu_waves_per_eu((RATE == 16000 && NWC > 0) ? 5 : (NWC == 0 ? SPX_TP_WAVES : (NWM == 8 ? 3 : 4)), (RATE == 16000 && NWC > 0) ? 5 : 8)))
spx_walk_fast_kernel(SpxPlanDev P, const SpxStreamDev* __restrict__ streams, const int16_t* __restrict__ in_base,
                     int16_t* __restrict__ out_base, int64_t* __restrict__ n_out, SpxStreamState* __restrict__ states,
                     const float* scratch_base, const int* speed_ready, int wcap) {
  constexpr int NT = 64 * (NWM + NWC);
  constexpr int FCG = fcg_of(NWM), FRG = frg_of(NWM);
  static_assert(SPEC == 0 || (SPEC == 1 && RATE != 0 && NWC > 0) || (SPEC == 2 && RATE == 0 && NWM == 4),
                "SPEC = 1: the long window of the rate-specialised kernels with output waves; SPEC = 2: the wide coarse select, plan-driven, four search waves");
  // WIDEC (SPEC = 2, round 5): coarse searches of 65 .. 128 lags -- two lags per lane in the coarse select (lags lane and 64 + lane),
  // sum buffers of 128 words in the block behind the layout (off_sumW).  11.025 kHz (72 coarse lags at skip 2) leaves the
  // general kernel with this.
  constexpr bool WIDEC = SPEC == 2;
  constexpr int CS = WIDEC ? 128 : 64;       // words per coarse sum buffer
  constexpr bool MCH = (MC & 1) != 0;
  constexpr bool SLOWK = (MC & 2) != 0;
  // WIDE: refine searches of up to 121 lags (8 skip + 1; 44.1 kHz: 89, 48 kHz: 97) -- two lags per lane in the refine select,
  // sum buffers of 128 words.  The eight-search-wave form only (its rectangle has the lanes for that many lags).
  constexpr bool WIDE = NWM == 8;
  constexpr int RS = WIDE ? 128 : 64;        // words per refine sum buffer
  constexpr int RIDLE = WIDE ? 256 : 128;    // sums[RIDLE + lane]: an idle lane's own word behind the two buffers
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);  // wave-uniform, and the compiler must know it: everything keyed on it stays scalar
  const SpxStreamDev S = streams[blockIdx.x];
  constexpr bool CT = RATE != 0;
  const int skip = CT ? (RATE > 4000 ? RATE / 4000 : 1) : P.skip;
  const int minP = CT ? RATE / 400 : P.minPeriod, maxP = CT ? RATE / 65 : P.maxPeriod;
  const int maxRequired = 2 * maxP;
  const int B = CT ? (int)(RATE / 100.0) : P.B;
  if (CT) wcap = SPEC == 1 ? SPX_CT_WCAP_LONG : SPX_CT_WCAP_OF(NWM, NWC);
  const FastLds LY = fast_lds_layout_i(minP, maxP, skip, wcap);

  FastOut X;
  X.C = MCH ? S.channels : 1;
  X.in = in_base + S.in_off - S.tsm_shift * X.C;  // indexed by (TSM position = input frame + flush padding so far) * C
  X.out = out_base + S.out_off;
  X.lds = lds;
  X.out_cap = (pos_t)(S.out_cap > 0x7fffffff ? 0x7fffffff : S.out_cap);
  X.offA0 = LY.off_mono;
  unsigned* sumC = reinterpret_cast<unsigned*>(lds + (WIDEC ? LY.off_sumW : LY.off_sumC));
  unsigned* sumR = reinterpret_cast<unsigned*>(lds + (WIDE ? LY.off_sumW : LY.off_sumR));
  int* cmd = reinterpret_cast<int*>(lds + LY.off_cmd);
  int* sWait = reinterpret_cast<int*>(lds + LY.off_wait);  // [0] polled count, [1] search-wave arrivals, [2] speculation done
  {
    double* invw = reinterpret_cast<double*>(lds + LY.off_inv);
    for (int t = tid; t <= maxP; t += NT) invw[t] = t > 0 ? 65536.0 / (double)t : 0.0;
    for (int t = tid; t < 128; t += NT) { sumC[t] = 0; sumR[t] = 0; }
    if constexpr (WIDE) { for (int t = 128 + tid; t < 512; t += NT) sumR[t] = 0; }
    if constexpr (WIDEC) { for (int t = 128 + tid; t < 512; t += NT) sumC[t] = 0; }
    if (tid < 4) sWait[tid] = 0;
    for (int t = tid; t < 2 * FCMD_INTS; t += NT) cmd[t] = 0;
  }
  __syncthreads();
#ifndef SPX_WALK_PAD
#define SPX_WALK_PAD 2
#endif
#if SPX_WALK_PAD > 0
  // Code placement: the step loop's speed depends on where it lies relative to the 64-byte instruction fetch lines (round 3:
  // removing ONE 8-byte prologue instruction, the loop's ISA unchanged, cost 2 %).  SPX_WALK_PAD s_nop's here, executed once,
  // shift everything behind them by four bytes each.  Swept 0 .. 64 bytes (variants -DSPX_WALK_PAD=n through tools/build_variant.sh + tools/variant_times.sh; rounds 3-5: tools/walk_pad_sweep.sh,
  // profiles/r03/r03x_walk_pad.txt): the walk kernel of the bench batch read 2.29 .. 2.36 ms, periodic in 64 bytes; 8 bytes
  // was the best.  With the step loop of the round's second half the sweep reads 2.032 .. 2.056 ms (r03af_walk_pad.txt): 8
  // bytes is within 0.2 % of the best and stays.  Re-run the sweep after any change to this kernel.
  asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(SPX_WALK_PAD));
#endif
  const int dA = LY.off_monoB - 2 - LY.off_mono;         // see pair_addr

  // ---- refine search: the dealing of its tasks to lanes (constants of the lane).  `sid` = index of the thread among
  // the 64 * NWM threads that run the refine SADs ----
  const int sid = (wave >= NWM) ? tid - 64 * NWM : tid;
  int nRG;  // ragged refine tasks per lane (uniform)
  {
    const int nlMax = 8 * skip + 1;
    int total = 0;
    for (int t = 0; t < nlMax; t++) total += (t + 2) >> 1;  // the larger of the two parities
    nRG = (total + 64 * NWM - 1) / (64 * NWM);
    if (nRG > FRG) nRG = FRG;
  }
  // ---- refine search, the ragged part dealt once.  Lag t of a search (p = lo + t) shares its first lo >> 1 pairs with
  // every other lag (summed lane = lag, wave = share); what is left is floor((t + (lo & 1)) / 2) whole pairs and, for an
  // odd p, the lone sample i = p - 1.  That triangle depends on lo only through its parity, so its enumeration over the
  // lanes of the search waves is a constant: task FRG * par + k of a lane = (lag t, pair r beyond lo >> 1, half?).
  // rT < 0: no task. ----
  // One packed word per task and parity: lag t (bits 0-7), pair r (8-15), lone-sample flag (16), task present (17) --
  // two plain arrays indexed by unrolled constants only, so they stay in registers in both copies of refine_sads.
  int rP0[FRG], rP1[FRG];
#pragma unroll
  for (int par = 0; par < 2; par++) {
#pragma unroll
    for (int k = 0; k < FRG; k++) {
      const int T = k * 64 * NWM + sid;
      const int nlMax = 8 * skip + 1;
      int first = 0, ft = -1, fr = 0, half = 0;
      // Round 4: PAIR-major order -- pair r of every lag that has one, then pair r + 1 ... -- so that neighbouring lanes hold
      // DIFFERENT lags.  A lane ends its task with a ds_add_u32 into its lag's sum, and lanes of a wave that add into one word
      // are served one after the other: in lag-major order up to 17 neighbouring lanes shared a lag (and every idle lane of the
      // second round added its zero into the sum of lag 0): ~330 LDS cycles of atomics per step, right in front of the step's
      // second barrier, against ~30 in this order (tools/lds_conflict_model.py) -- the bank conflicts the round-3 counters showed
      // and the model of the reads could not explain.  Lag t has pair r iff r < ceil((t + par) / 2), i.e. t >= 2 r + 1 - par.
      for (int r = 0; 2 * r + 1 - par < nlMax; r++) {
        const int tmin = (2 * r + 1 - par) > 0 ? (2 * r + 1 - par) : 0;
        const int cnt = nlMax - tmin;                // lags that have a pair (or the lone sample) number r
        if (ft < 0 && T < first + cnt) {
          ft = tmin + (T - first); fr = r;
          half = (r < ((ft + par) >> 1)) ? 0 : 1;   // beyond the lag's whole pairs: the lone sample of an odd lag
        }
        first += cnt;
      }
      const int w = (ft < 0) ? 0 : ((ft & 0xff) | ((fr & 0xff) << 8) | (half << 16) | (1 << 17));
      if (par == 0) rP0[k] = w; else rP1[k] = w;
    }
  }
  // ---- refine search, the common rectangle: lane -> (lag myT, chunk myC), constant.  NLAG = 8*skip + 1 lags at most,
  // NCH = (search lanes) / NLAG chunks (16 kHz: 33 lags x 7 chunks = 231 of 256 lanes; 22.05 kHz: 41 x 6 = 246). ----
  const int NLAG = 8 * skip + 1;
  const int NCH = (64 * NWM) / NLAG;
  const int myT = sid % NLAG, myC = sid / NLAG;
  const bool myOn = myC < NCH;
  const int chM = (65536 + NCH - 1) / NCH;  // g / NCH == (g * chM) >> 16 for every group count (checked by spx_walk_fast_supports)
  // the lane's constant parts of its rectangle operand addresses (round 4, see the coarse dealing below for the algebra):
  // byte address = (uniform part of the step) + (constant of the lane) [+ one uniform-selected term for the `b` parity]
  const int rC16 = 16 * myC, rC4 = 4 * myC;
  const int rB0 = 2 * myT + ((myT & 1) ? dA : 0), rFlip = (myT & 1) ? -dA : dA;
  FSTAMP_VARS
  // The SAD phase of a refine search at window offset o over the lags lo..hi: the ragged tasks and the rectangle of the
  // calling lane, added into sums[lag - lo].
  auto refine_sads = [&](int o, int lo, int hi, unsigned* sums) __attribute__((always_inline)) {
    const int c0 = lo >> 1;  // pairs every lag of this search has
    const int par = lo & 1;
    const int nl = hi - lo + 1;
    // the ragged tasks first: their operands are in flight while the common share is summed
    unsigned ra[FRG], rb[FRG];
    int rt[FRG];
    unsigned rm[FRG];
#pragma unroll
    for (int k = 0; k < FRG; k++) {
      if (k >= nRG) { rt[k] = 0; rm[k] = 0u; ra[k] = 0u; rb[k] = 0u; continue; }   // (no atomic is issued for k >= nRG)
      const int w = par ? rP1[k] : rP0[k];
      rt[k] = w & 0xff;
      const int rr = (w >> 8) & 0xff;
      rm[k] = (w & (1 << 17)) ? ((w & (1 << 16)) ? 0xffffu : 0xffffffffu) : 0u;
      // no task, or a lag the clamped search does not have
      if (!(w & (1 << 17)) || rt[k] >= nl) { rt[k] = 0; rm[k] = 0u; }   // (rm == 0 marks the lane idle for the add below)
      const int ea = o + 2 * (c0 + rr);
      ra[k] = *reinterpret_cast<const unsigned*>(lds + pair_addr(LY.off_mono, dA, ea));
      rb[k] = *reinterpret_cast<const unsigned*>(lds + pair_addr(LY.off_mono, dA, ea + lo + rt[k]));
    }
    FSTAMP(12);
    SPX_PROBE(6);   // behind the ragged loads, in front of the rectangle set-up
    // common share: the c0 pairs every lag of the search has form a rectangle of lags x pairs, cut into groups of four
    // pairs and dealt to ALL search lanes: lane = (lag myT, chunk myC) takes NGL = (c0 / 4) / NCH consecutive groups --
    // the same count for every lane, so no masks -- plus, for the first chunks, one of the left-over groups and one of
    // the c0 % 4 left-over pairs (switched off by reading the a operand twice: |a - a| = 0).  One flight of loads.
    const int G = c0 >> 2, rho = c0 & 3;
    int NGL = CT ? G / NCH : (G * chM) >> 16;
    const int LG = G - NCH * NGL;
    asm volatile("" : "+s"(NGL));  // opaque: keeps the branch conditions below scalar compares of this value
    const bool tOk = myOn && myT < nl;
    // The lane's groups start at sample ea = o + 8 myC NGL (same parity as o), its `b` operand at ea + lo + myT; its left-over
    // group is group NCH NGL + myC and its left-over pair is pair 4 G + myC of the rectangle.  As byte addresses:
    //   a      = [M + 2 o + (o & 1) dA]  + 16 myC NGL
    //   b      = [M + 2 (o + lo)]        + 16 myC NGL + 2 myT + (parity of o + lo + myT ? dA : 0)
    //   a left-over group = a's uniform part + 16 NCH NGL + 16 myC      (the 16 myC NGL cancel),   pair: + 16 G + 4 myC
    // -- uniform parts on the scalar unit, lane constants in registers since kernel start, one multiplication.
    const int aU = LY.off_mono + 2 * o + (o & 1) * dA;
    const int bU = LY.off_mono + 2 * (o + lo);
    const int aL = (int)__umul24((unsigned)rC16, (unsigned)NGL);
    const int bSel = rB0 + ((o + lo) & 1) * rFlip;
    const int XU = 16 * NCH * NGL, GU = 16 * G;
    const unsigned* ap = reinterpret_cast<const unsigned*>(lds + (aU + aL));
    const unsigned* bp = reinterpret_cast<const unsigned*>(lds + (bU + aL + bSel));
    const int apxO = aU + XU + rC16, appO = aU + GU + rC4;
    const unsigned* apx = reinterpret_cast<const unsigned*>(lds + apxO);
    const unsigned* bpx = reinterpret_cast<const unsigned*>(lds + ((myC < LG) ? bU + XU + rC16 + bSel : apxO));
    const unsigned* app = reinterpret_cast<const unsigned*>(lds + appO);
    const unsigned* bpp = reinterpret_cast<const unsigned*>(lds + ((myC < rho) ? bU + GU + rC4 + bSel : appO));
    unsigned d = 0u;
    // the left-over group and pair are the same code whatever the group count: their loads go first, the dispatch on the
    // group count holds whole groups only, their SADs come last (one flight of loads all the same: nothing waits in between)
    unsigned xa[4], xb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { xa[k] = apx[k]; xb[k] = bpx[k]; }
    const unsigned pa = *app, pb = *bpp;
    while (SPX_UNLIKELY(NGL > 3)) {  // long periods at the higher rates only
      d = sad_flight_n<4, false>(ap, bp, 0, d);
      ap += 16; bp += 16; NGL -= 4;
    }
    switch (NGL) {
      case 1: d = sad_flight_n<1, false>(ap, bp, 0, d); break;
      case 2: d = sad_flight_n<2, false>(ap, bp, 0, d); break;
      case 3: d = sad_flight_n<3, false>(ap, bp, 0, d); break;
      default: break;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) d = __builtin_amdgcn_sad_u16(xa[k], xb[k], d);
    d = __builtin_amdgcn_sad_u16(pa, pb, d);
    atomicAdd(&sums[tOk ? myT : RIDLE + lane], tOk ? d : 0u);
    FSTAMP(13);
#pragma unroll
    for (int k = 0; k < FRG; k++) {
      if (k < nRG) {
        const unsigned dr = SPX_SAD_MASKED(rm[k], ra[k], rb[k], 0u);
        // an idle lane's zero goes to a word of its own in the spare block behind the two sum buffers (sums + 128 + lane stays
        // inside it from either buffer), not into a sum other lanes add into: same-address atomics of a wave are served in turn
        atomicAdd(&sums[rm[k] ? rt[k] : RIDLE + lane], dr);
      }
    }
    FSTAMP(14);
  };

  if (wave >= NWM) {
    // ------------------------------ output waves: obey commands until FCMD_EXIT ------------------------------
    if constexpr (NWC > 0) {
      int seq = 0;
      int nsteps = 0;  // step commands seen = pitch searches of this job: counted HERE, off the chain (SpxWalkState::steps)
      // this thread's index among the output threads, formed HERE: hoisted into the kernel's prologue the compiler once kept it in
      // scratch (the 96-register forms), reloaded it in front of this loop, and the reload's s_waitcnt vmcnt(0) -- placed at the first
      // use, inside the loop -- made every command wait for the previous command's STORES as well: the plain call's walk kernel
      // 1.70 -> 1.80 ms (round 5, profiles/r05/r5ac_variant_times.txt)
      int t0o = tid - 64 * NWM;
      asm volatile("" : "+v"(t0o));
      pos_t wb = -1, lim = (pos_t)(S.n_in + S.tsm_shift);  // window base / input limit as the commands have announced them
      for (;;) {
        fast_sync();  // the barrier that follows every published command
        const int* c = cmd + (seq & 1) * FCMD_INTS;
        seq++;
        const int type = uni(c[0]);
        const int xf_n = uni(c[1]), xf_down = uni(c[2]), xf_period = uni(c[3]), xf_out = uni(c[4]);
        const int cp_n = uni(c[5]), cp_src = uni(c[6]), cp_out = uni(c[7]);
        const pos_t nb = uni(c[9]);
        if (type != FCMD_STEP) lim = uni(c[8]);  // a step command carries fields 0..4 only
#ifndef SPX_EXP_NO_OUTPUT   // (diagnostic builds: the output waves only take part in the barriers -- WRONG audio, same chain)
        fast_outputs<64 * NWC, MCH>(X, t0o, xf_n, xf_down, xf_period, xf_out, type == FCMD_STEP ? 0 : cp_n, cp_src,
                               cp_out, lim, wb);
#endif
        if (type == FCMD_STEP) {
          nsteps++;
          fast_sync();            // the step's second barrier (refine sums complete)
        } else if (type == FCMD_REFILL) {
          fast_refill<NT, MCH>(X, LY, skip, nb, lim);
          wb = nb;
          // (Touching the lines the NEXT refill will load from here -- one element per 128-byte line, so that they sit in this
          // XCD's L2 by then -- changed nothing: 2.052 against 2.049 ms, profiles/r03/r03z_l2pf.txt.)
        } else if (type == FCMD_POLL) {
          fast_sync();            // the polled count is in LDS
        } else if (type == FCMD_EXIT) {
          // (the search waves write the rest of the record, field by field, at about the same time)
          if (tid == 64 * NWM) states[blockIdx.x].w.steps = ((S.flags & SPX_F_INIT) ? 0 : states[blockIdx.x].w.steps) + nsteps;
          break;
        }
      }
    }
    return;
  }

  // ---------------------------------------- search waves: the chain ----------------------------------------
  // the walk is the latency-critical chain: where another kernel shares a SIMD, these waves issue first
  __builtin_amdgcn_s_setprio(SPX_WALK_PRIO);
  const int Ttot = S.n_frames, F = P.F;
  const float Rg = S.speed, nl = S.nonlinear;
  const bool linear = nl == 0.0f;
  const bool hotStream = Rg >= 2.0f;  // most of its events run at speed >= 2: see the hot loop
  const bool do_flush = (S.flags & SPX_F_FLUSH) != 0;
  const int minC = minP / skip, nC = maxP / skip - minC + 1;
  const bool needResolve = (long)maxP * maxP >= 65536;
  const int need = maxRequired + 2 * skip + 2;           // window frames a step needs from its position on
  const int skipM = (65536 + skip - 1) / skip;           // i / skip == (i * skipM) >> 16 for i < 8192
  const int dPl = LY.off_plB - 2 - LY.off_pl;
  const double* invTab = reinterpret_cast<const double*>(lds + LY.off_inv);
  const float* scr = scratch_base + (size_t)S.frame_off * 4;  // per frame: ..., speed (written by the tension kernel)

  // the part of the stream state this stage owns (the tension kernel owns the filter states)
  pos_t base, out_n, avail, handed;
  int remaining, prevPeriod, prevMinDiff, overflow;
  float tailSpeed;  // the speed in force in the TSM stage
  if (S.flags & SPX_F_INIT) {
    base = 0; out_n = 0; avail = 0; handed = 0; remaining = 0; prevPeriod = 0; prevMinDiff = 0; overflow = 0;
    tailSpeed = Rg;  // sonicSetSpeed -> sonicIntSetSpeed, soniclib.c:182
  } else {
    const SpxStreamState& Z = states[blockIdx.x];
    base = uni((pos_t)Z.w.base); out_n = uni((pos_t)Z.w.out_n); avail = uni((pos_t)Z.w.avail);
    remaining = uni(Z.w.remaining); prevPeriod = uni(Z.w.prevPeriod); prevMinDiff = uni(Z.w.prevMinDiff);
    overflow = uni(Z.w.overflow);
    handed = linear ? 0 : uni(Z.handed);
    tailSpeed = (linear || (S.flags & SPX_F_SPEED_SET)) ? Rg : unif(Z.curSpeed);  // sonicSetSpeed between writes reaches the TSM stage at once (soniclib.c:182)
  }
  const pos_t n_tsm = (pos_t)(S.n_in + S.tsm_shift);  // the written input in TSM positions
  pos_t limit = n_tsm;           // frames of real input; reads beyond are the flush's zero padding
  pos_t wbase = -1;              // window covers [wbase, wbase + wcap); -1 = invalid
  int tg = 0;                    // which of the two lag-sum buffers this step adds into (both clear at kernel start)
  int seq = 0;                   // commands published
  int nsteps = 0;                // pitch searches of this job (counted here only when there are no output waves)
  (void)nsteps;
  int xf_n = 0, xf_down = 0, xf_period = 0;  // cross-fade decided but not yet handed to the output waves
  pos_t xf_out = 0;

  // ---- coarse search, dealt once: task T = g * 64 * NWM + tid is group (T - first group of its lag) of the lag whose
  // groups contain T; a group is four consecutive pair slots of that lag (slot j = samples 2j, 2j+1; an odd lag's last
  // slot holds one sample).  cOffA / cOffB: element offsets of the two operands from the step's decimated position;
  // cMask: per-slot byte masks (all / low half / none); cLag: lag index (= lane of the sum it adds into). ----
  int cOffA[FCG], cOffB[FCG], cLag[FCG], cByteB[FCG], cFlipB[FCG];
  unsigned cMask[FCG][4];
  int nGC;  // groups per lane (uniform)
  {
    int total = 0;
    for (int q = 0; q < nC; q++) { const int p = minC + q; total += ((p >> 1) + (p & 1) + 3) >> 2; }
    nGC = (total + 64 * NWM - 1) / (64 * NWM);
    if (nGC > FCG) nGC = FCG;  // spx_walk_config keeps such plans off this kernel
#pragma unroll
    for (int g = 0; g < FCG; g++) {
      // (round 4) consecutive groups -- the groups of one lag -- on DIFFERENT waves: a wave's ds_add_u32 then finds at most two
      // or three of its lanes on one sum instead of up to eight neighbours (model: 47 -> 20 LDS cycles of atomics per step in
      // front of the first barrier, for 12 more cycles of bank conflicts in the operand reads behind it)
      const int T = g * 64 * NWM + lane * NWM + wave;
      int first = 0, found = 0, fq = 0, ff = 0;
      for (int q = 0; q < nC; q++) {
        const int p = minC + q;
        const int ng = ((p >> 1) + (p & 1) + 3) >> 2;
        if (!found && T < first + ng) { found = 1; fq = q; ff = first; }
        first += ng;
      }
      const int p = minC + fq, gi = T - ff;
      cOffA[g] = 8 * gi;
      cOffB[g] = p + 8 * gi;
      // (a lane without a group adds its zero into a word of its own in the spare block behind the sum buffers -- index 256 +
      // lane from either coarse buffer -- not into the sum of lag 0: same-address atomics of a wave are served one after the
      // other, and with two search waves or at 22.05 kHz a third of the lanes of a round have no group)
      cLag[g] = found ? fq : 256 + lane;
      if (!found) { cOffA[g] = 0; cOffB[g] = 0; }
      // Round 4: the operand addresses as (wave-uniform part of the step) + (constant of the lane).  pair_addr(base, d, e) =
      // base + 2 e + (e & 1) d; the step's decimated position oD is uniform, so the parity of e = oD + offset is the uniform
      // parity of oD XOR the lane's constant parity of its offset:  address = base + 2 oD + [2 off + (off odd ? d : 0)] +
      // (oD & 1) * (off odd ? -d : d).  A step then forms its two addresses with one v_add and one v_mad instead of the dozen
      // VALU instructions of two generic pair_addr evaluations -- and every instruction a search wave issues is on the chain.
      cOffA[g] = 2 * cOffA[g];                                                   // (the `a` offset is even: its parity is oD's)
      cByteB[g] = 2 * cOffB[g] + ((cOffB[g] & 1) ? dPl : 0);
      cFlipB[g] = (cOffB[g] & 1) ? -dPl : dPl;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int j = 4 * gi + k;
        cMask[g][k] = !found ? 0u : (j < (p >> 1) ? 0xffffffffu : ((j == (p >> 1) && (p & 1)) ? 0xffffu : 0u));
      }
    }
  }
  const double scaleC = 65536.0 / (double)(minC + lane);
  const bool validC = lane < nC;
  const double scaleC2 = WIDEC ? 65536.0 / (double)(minC + 64 + lane) : 0.0;   // the lane's second coarse lag, 64 + lane
  const bool validC2 = WIDEC && lane + 64 < nC;

  // Hand the pending cross-fade (and, for FCMD_COPY, a plain copy) to the output waves.  Every command is followed by
  // exactly one workgroup barrier before the next command is published, and the two slots alternate, so a slot is
  // rewritten only after the output waves have consumed it.  Field k of a command is written by lane k of the last
  // search wave (v_writelane: no EXEC juggling).  NWC == 0: the search waves do the output work themselves.
#ifdef SPX_EXP_NO_OUTPUT_LEAN   // (diagnostic builds: the forms without output waves produce NO audio -- the chain's length without its output work)
#define SPX_LEAN_OUTPUTS(CP_N, CP_SRC, CP_OUT) do { } while (0)
#else
#define SPX_LEAN_OUTPUTS(CP_N, CP_SRC, CP_OUT) \
  fast_outputs<64 * NWM, MCH>(X, tid, xf_n, xf_down, xf_period, xf_out, (int)(CP_N), (pos_t)(CP_SRC), (pos_t)(CP_OUT), limit, wbase)
#endif
#define FAST_PUBLISH(TYPE, CP_N, CP_SRC, CP_OUT, NB)                                                                   \
  do {                                                                                                                 \
    if constexpr (NWC > 0) {                                                                                           \
      if (wave == NWM - 1) {                                                                                           \
        int rec_ = 0;                                                                                                  \
        rec_ = fast_writelane((TYPE), 0, rec_);                                                            \
        rec_ = fast_writelane(xf_n, 1, rec_);                                                              \
        rec_ = fast_writelane(xf_down, 2, rec_);                                                           \
        rec_ = fast_writelane(xf_period, 3, rec_);                                                         \
        rec_ = fast_writelane((int)xf_out, 4, rec_);                                                       \
        if ((TYPE) != FCMD_STEP) {                                                                                     \
          rec_ = fast_writelane((int)(CP_N), 5, rec_);                                                     \
          rec_ = fast_writelane((int)(CP_SRC), 6, rec_);                                                   \
          rec_ = fast_writelane((int)(CP_OUT), 7, rec_);                                                   \
          rec_ = fast_writelane((int)limit, 8, rec_);                                                      \
          rec_ = fast_writelane((int)(NB), 9, rec_);                                                       \
        }                                                                                                              \
        cmd[(seq & 1) * FCMD_INTS + lane] = rec_;                                                                      \
      }                                                                                                                \
      seq++;                                                                                                           \
    } else {                                                                                                           \
      SPX_LEAN_OUTPUTS(CP_N, CP_SRC, CP_OUT);                                                                          \
    }                                                                                                                  \
    xf_n = 0;                                                                                                          \
  } while (0)

  // findPitchPeriod at absolute position pos; every search wave computes the same result.  Also returns what the step
  // then does: n frames of cross-fade and, for 1 < speed < 2, `rem` frames copied through afterwards.
  // The operands of the coarse search at window offset o (each lane its constant groups of pair slots): loads only.
  auto coarse_loads = [&](int o, unsigned (&a)[FCG][4], unsigned (&b)[FCG][4]) __attribute__((always_inline)) {
    const int oD = (o * skipM) >> 16;
    const int r = o - oD * skip;
    const int plr = LY.off_pl + r * LY.plStrideB;
    const int parD = oD & 1;
    const int baseU = plr + 2 * oD;            // uniform
    const int aU = baseU + parD * dPl;         // uniform
#pragma unroll
    for (int g = 0; g < FCG; g++) {
      if (g < nGC) {
        const unsigned* ap = reinterpret_cast<const unsigned*>(lds + (aU + cOffA[g]));
        const unsigned* bp = reinterpret_cast<const unsigned*>(lds + (baseU + cByteB[g] + parD * cFlipB[g]));
#pragma unroll
        for (int k = 0; k < 4; k++) { a[g][k] = ap[k]; b[g][k] = bp[k]; }
      }
    }
  };
  // (Round 3 tried issuing the NEXT step's coarse loads from the end of a step, in front of the bookkeeping between two steps:
  // 3 % slower at every code placement, profiles/r03/r03y_pf_pads.txt -- the round trip was not what the chain waited for.)

  // kind (the general call site of an instantiation that serves slow-down only): which candidate formula the step takes --
  //   0  speed >= 2:       n = (int)(p / (speed - 1))                     1  1 < speed < 2:     n = p, rem = (int)(p (2 - speed) / (speed - 1))
  //   2  speed < 0.5:      n = (int)(p speed / (1 - speed))               3  0.5 <= speed < 1:  n = p, rem = (int)(p (2 speed - 1) / (1 - speed))
  // with `sm1` the divisor and `twom` the numerator's factor as the caller prepared them; IEEE divisions there.
  auto find_period = [&](pos_t pos, bool ge2, float sm1, float twom, float rinv, int& n_ret, int& rem_ret, auto hot, int kind = 0)
                         __attribute__((always_inline)) -> int {
    (void)hot;  // std::true_type from the hot loop: its copy of this code has ge2 a constant
    (void)kind;
    constexpr bool ANYK = SLOWK && !decltype(hot)::value;
    if constexpr (NWC == 0) nsteps++;   // (with output waves THEY count the step commands, off the chain)
    FSTAMP(1);
    if (SPX_UNLIKELY(!(wbase >= 0 && pos >= wbase && pos + need <= wbase + LY.wcap))) {
      FSTAMP(16);
      const pos_t nb = pos & ~7;
      FAST_PUBLISH(FCMD_REFILL, 0, 0, 0, nb);
      if (NWC > 0) fast_sync();
      fast_refill<NT, MCH>(X, LY, skip, nb, limit);
      wbase = nb;
      FSTAMP(15);
    }
    FSTAMP(2);
    const int o = (int)(pos - wbase);
    SPX_PROBE(1);   // in front of the coarse operand addresses
    // ---- coarse search on the decimated signal: each lane its constant group(s) of pair slots ----
    int bestC;
    {
      unsigned a[FCG][4], b[FCG][4];
      coarse_loads(o, a, b);
      SPX_PROBE(2);   // behind the coarse loads (their shadow)
      FAST_PUBLISH(FCMD_STEP, 0, 0, 0, 0);  // the previous step's cross-fade rides on this step's command; behind the loads
#pragma unroll
      for (int g = 0; g < FCG; g++) {
        if (g < nGC) {
          unsigned d = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) d = SPX_SAD_MASKED(cMask[g][k], a[g][k], b[g][k], d);
          atomicAdd(&sumC[tg * CS + cLag[g]], d);  // (idle lanes: a word of their own, see the dealing)
        }
      }
      FSTAMP(3);
      SPX_PROBE(3);   // behind the coarse atomics, in front of the first barrier
      fast_sync();
      FSTAMP(4);
      SPX_PROBE(4);   // behind the first barrier, in front of the coarse select
      if (wave == 0) {                                 // the buffer the previous step used: everyone is past it
        sumC[(1 - tg) * CS + lane] = 0;
        if constexpr (WIDEC) sumC[(1 - tg) * CS + 64 + lane] = 0;
      }
      const unsigned dsum = sumC[tg * CS + lane];
      unsigned kmin;
      if constexpr (WIDEC) bestC = fast_select2(dsum, sumC[tg * CS + 64 + lane], scaleC, scaleC2, validC, validC2, false, minC, kmin);
      else bestC = fast_select(dsum, scaleC, validC, false, minC, kmin);
    }
    FSTAMP(5);
    SPX_PROBE(5);   // behind the coarse select, in front of the refine set-up
    // ---- refine at full rate around the coarse winner ----
    int period = (minC + bestC) * skip;
    int lo = period - (skip << 2), hi = period + (skip << 2);
    if (lo < minP) lo = minP;
    if (hi > maxP) hi = maxP;
    const bool valid = lane <= hi - lo;
    const int p = lo + lane;
    const double scale = invTab[valid ? p : lo];
    // WIDE: the lane's second lag, 64 + lane (registers `...2`; the previous period's slot is lane 63 of THAT set)
    const bool valid2 = WIDE && lane + 64 <= hi - lo;
    const int p2 = lo + 64 + lane;
    const double scale2 = WIDE ? invTab[valid2 ? p2 : lo] : 0.0;
    unsigned dsum, dsum2 = 0u;
    int nLane, remLane, nLane2 = 0, remLane2 = 0;
    {
      refine_sads(o, lo, hi, sumR + tg * RS);
      // what the step will do for each candidate period: exact IEEE divisions, off the chain; lane 63 = previous period
      {
        const int pc = (!WIDE && lane == 63) ? prevPeriod : p;
        const float fp = (float)pc;
#ifdef SPX_IEEE_DIV
        (void)rinv;
        nLane = ge2 ? (int)(fp / sm1) : pc;
        remLane = ge2 ? 0 : (int)(fp * twom / sm1);
#else
        if constexpr (ANYK) {
          const int q = (int)((kind == 0 ? fp : fp * twom) / sm1);
          nLane = (kind & 1) ? pc : q;
          remLane = (kind & 1) ? q : 0;
        } else {
          nLane = ge2 ? (int)fast_div(fp, sm1, rinv) : pc;
          remLane = ge2 ? 0 : (int)fast_div(fp * twom, sm1, rinv);
        }
#endif
        if constexpr (WIDE) {
          const int pc2 = (lane == 63) ? prevPeriod : p2;
          const float fp2 = (float)pc2;
#ifdef SPX_IEEE_DIV
          nLane2 = ge2 ? (int)(fp2 / sm1) : pc2;
          remLane2 = ge2 ? 0 : (int)(fp2 * twom / sm1);
#else
          if constexpr (ANYK) {
            const int q2 = (int)((kind == 0 ? fp2 : fp2 * twom) / sm1);
            nLane2 = (kind & 1) ? pc2 : q2;
            remLane2 = (kind & 1) ? q2 : 0;
          } else {
            nLane2 = ge2 ? (int)fast_div(fp2, sm1, rinv) : pc2;
            remLane2 = ge2 ? 0 : (int)fast_div(fp2 * twom, sm1, rinv);
          }
#endif
        }
      }
      asm volatile("" ::"v"(nLane), "v"(remLane));  // here, while the sums are on their way -- not behind the barrier
      if constexpr (WIDE) asm volatile("" ::"v"(nLane2), "v"(remLane2));
      FSTAMP(6);
      SPX_PROBE(7);   // behind the refine atomics and the candidate divisions, in front of the second barrier
      fast_sync();  // the step's one workgroup barrier: refine sums complete, the output waves done with the command
      FSTAMP(7);
      SPX_PROBE(8);   // behind the second barrier, in front of the refine select
      if (wave == 0) {
        sumR[(1 - tg) * RS + lane] = 0;
        if constexpr (WIDE) sumR[(1 - tg) * RS + 64 + lane] = 0;
      }
      dsum = sumR[tg * RS + lane];
      if constexpr (WIDE) dsum2 = sumR[tg * RS + 64 + lane];
    }
    tg ^= 1;
    unsigned kmin;
    int best;
    if constexpr (WIDE) best = fast_select2(dsum, dsum2, scale, scale2, valid, valid2, needResolve, lo, kmin);
    else best = fast_select(dsum, scale, valid, needResolve, lo, kmin);
    period = lo + best;
    const int minDiff = (int)(kmin >> 16);  // floor(diff / lag) of the winner
    FSTAMP(8);
    SPX_PROBE(9);   // behind the refine select (decision, bookkeeping)
    // Previous-period rule (libsonic prevPeriodBetter, preferNewPeriod = 1).  Only "maxDiff > 3*minDiff" is ever asked
    // of the worst lag, and max_p floor(d_p/p) = floor(max_p d_p/p), so the test is "some lag has d_p >= (3*minDiff+1)*p".
    int ret = period, sel = best;
    // (unlikely: the rule's body out of line, the common case falls through -- walk kernel 2.035 -> 2.02 ms, r03aw_micro.txt)
    if (SPX_UNLIKELY(minDiff != 0 && prevPeriod != 0 && minDiff * 2 > prevMinDiff * 3)) {
      const unsigned need3 = 3u * (unsigned)minDiff + 1u;
      if constexpr (WIDE) {
        if (__builtin_amdgcn_ballot_w64((valid && dsum >= need3 * (unsigned)p) || (valid2 && dsum2 >= need3 * (unsigned)p2)) == 0) { ret = prevPeriod; sel = 127; }
      } else {
        if (__builtin_amdgcn_ballot_w64(valid && dsum >= need3 * (unsigned)p) == 0) { ret = prevPeriod; sel = 63; }
      }
    }
    prevMinDiff = minDiff;
    prevPeriod = period;
    if (WIDE && sel >= 64) {
      n_ret = __builtin_amdgcn_readlane(nLane2, sel - 64);
      rem_ret = __builtin_amdgcn_readlane(remLane2, sel - 64);
    } else {
      n_ret = __builtin_amdgcn_readlane(nLane, sel);
      rem_ret = __builtin_amdgcn_readlane(remLane, sel);
    }
    FSTAMP(9);
    return ret;
  };

  // The pitch steps one event can run = the loop of libsonic's processStreamInput for speed > 1 with `availE` frames
  // handed over (the caller guarantees availE - base >= maxRequired).  A step that fails (n == 0) ends the event without
  // removing the input consumed in this call (the dependency returns 0 there), so `base` keeps its value.
  auto run_event = [&](float speed, pos_t availE) __attribute__((always_inline)) {
    const bool ge2 = speed >= 2.0f;
    const float sm1 = speed - 1.0f, twom = 2.0f - speed;
    const float rinv = fast_rcp_refined(sm1);
    pos_t pos = base;
    bool failed = false;
    do {
      if (SPX_UNLIKELY(remaining > 0)) {
        int n = remaining;
        if (n > maxRequired) n = maxRequired;
        if (out_n + n > X.out_cap) overflow = 1;
        FAST_PUBLISH(FCMD_COPY, n, pos, out_n, 0);
        if (NWC > 0) fast_sync();
        out_n += n;
        remaining -= n;
        pos += n;
      } else {
        int n, rem;
        const int period = find_period(pos, ge2, sm1, twom, rinv, n, rem, std::false_type(), ge2 ? 0 : 1);
        if (!ge2) remaining = rem;
        if (out_n + n > X.out_cap) overflow = 1;
        // a failed step (n == 0) is no branch of its own: it hands over no cross-fade (xf_n = 0), leaves pos where it is and ends
        // the loop through its condition
        failed = n == 0;
        xf_n = n; xf_down = (int)(pos - wbase); xf_period = period; xf_out = out_n;
        out_n += n;
        pos += failed ? 0 : period + n;
        FSTAMP(10);
      }
    } while (!failed && pos + maxRequired <= availE);
    if (!failed) base = pos;
  };

  // The same for an event at a speed BELOW 1 (SLOWK instantiations): libsonic's insertPitchPeriod.  A step copies the period at the
  // position through (`period` frames, a plain copy command), cross-fades n frames from the period BEHIND the position down into
  // the one AT it -- the usual cross-fade with the ramps swapped: down = pos + period, up = down - period -- and consumes n frames:
  // n = (int)(period speed / (1 - speed)) below 0.5, else n = period with (int)(period (2 speed - 1) / (1 - speed)) frames copied
  // through afterwards.  A step with n == 0 (a very low speed and a short period) has ALREADY emitted its period when it fails --
  // the dependency appends before it reports -- and leaves the position where it was.
  auto run_event_slow = [&](float speed, pos_t availE) __attribute__((always_inline)) {
    const bool lt05 = speed < 0.5f;
    const float den = 1.0f - speed, fac = lt05 ? speed : 2.0f * speed - 1.0f;
    pos_t pos = base;
    bool failed = false;
    do {
      if (remaining > 0) {
        int n = remaining;
        if (n > maxRequired) n = maxRequired;
        if (out_n + n > X.out_cap) overflow = 1;
        FAST_PUBLISH(FCMD_COPY, n, pos, out_n, 0);
        if (NWC > 0) fast_sync();
        out_n += n;
        remaining -= n;
        pos += n;
      } else {
        int n, rem;
        const int period = find_period(pos, false, den, fac, 0.0f, n, rem, std::false_type(), lt05 ? 2 : 3);
        if (!lt05) remaining = rem;
        if (out_n + period + n > X.out_cap) overflow = 1;
        failed = n == 0;
        FAST_PUBLISH(FCMD_COPY, period, pos, out_n, 0);   // (with the previous step's cross-fade, as every command)
        if (NWC > 0) fast_sync();
        xf_n = n; xf_down = (int)(pos - wbase) + period; xf_period = -period; xf_out = out_n + period;
        out_n += period + n;
        pos += n;
      }
    } while (!failed && pos + maxRequired <= availE);
    if (!failed) base = pos;
  };

  // The speeds come from the tension kernel.  Sequential launches (speed_ready == nullptr): all of them are there.
  // Concurrent launches: that kernel runs beside this one and publishes the number of tension frames whose speeds are
  // final (agent-scope release there; one relaxed poll + agent-scope acquire here, cdna_hip_programming.md G16).
  const int K_total = (!linear && Ttot >= F) ? Ttot - F + 1 : 0;  // soniclib.c:317
  // concurrent mode: this workgroup has been placed (the engine's idle-start gate counts the arrivals, spx_engine.hip)
  if (speed_ready != nullptr && tid == 0)
    __hip_atomic_fetch_add(const_cast<int*>(speed_ready) + gridDim.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    int K = K_total;
    if (speed_ready != nullptr && K_total > 0) {
      FAST_PUBLISH(FCMD_POLL, 0, 0, 0, 0);
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
      if (K < 0) { overflow = 2; break; }  // the producer was lost (or failed): reported as its own status
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
    else ev1 = (last && n_tsm > avail) ? 1 : 0;
    if (ev1 < ev0) ev1 = ev0;
    // Most nonlinear events cannot run a step (a step needs maxRequired frames, an event brings B).  The speeds of 64
    // consecutive events sit in one VGPR (lane = event, tail events at the last speed), and the next event that does
    // anything -- a pass-through at unity speed, or one with enough input for a step -- is found with one ballot.
    const pos_t Kc = ev1 < (pos_t)K ? ev1 : (pos_t)K;  // tension events of this round: [ev0, Kc)
    if (Kc > ev0) tailSpeed = unif(scr[4 * (size_t)(Kc - 1) + 3]);  // what later events run at / the stream carries on
    // The flush is one more block of a single event: sonicIntFlushStream pads 2*maxRequired zeros, processes at the
    // last speed, truncates to the expected length and empties its input.
    const int nBlk = (int)((ev1 - ev0 + 63) >> 6);
    for (int bI = 0; bI < nBlk + (fin ? 1 : 0); bI++) {
      const bool flushBlk = bI == nBlk;
      const pos_t blk0 = ev0 + 64 * (pos_t)bI;
      int nIn = (int)(ev1 - blk0 < 64 ? ev1 - blk0 : 64);
      pos_t expected = 0;
      if (flushBlk) {
        nIn = 1;
        const pos_t remainingS = avail - base;
        expected = out_n + uni((int)(((float)remainingS / tailSpeed + 0) / 1.0f + 0.5f));
        if constexpr (NWC == 0 && MCH) {
          // a cross-fade still waiting to be written takes its ramps from the input at wbase + offset (several channels:
          // the window holds the mean only): write it while wbase is still the window it was decided in
          if (xf_n > 0) {
            fast_outputs<64 * NWM, MCH>(X, tid, xf_n, xf_down, xf_period, xf_out, 0, 0, 0, limit, wbase);
            xf_n = 0;
          }
        }
        limit = avail;  // everything from here on reads as the flush's zero padding
        wbase = -1;     // the window may hold samples past the new limit: the next step refills it
      }
      const pos_t idx = blk0 + lane;
      float spv = tailSpeed;
      if (!flushBlk && lane < nIn && idx < (pos_t)K) spv = scr[4 * (size_t)idx + 3];
      const bool unityLane = lane < nIn && speed_is_unity(spv);
      const unsigned long long unityMask = __builtin_amdgcn_ballot_w64(unityLane);
      const pos_t availBlk = avail;
      const int perEvent = flushBlk ? 2 * maxRequired : B;
      const pos_t availLane = (linear && !flushBlk) ? n_tsm : availBlk + (lane + 1) * perEvent;  // frames handed over after event `lane`
      int i = 0;
      for (;;) {
        // The hot loop: events at speed >= 2 with nothing left to copy through -- nearly every event of a stream that is sped
        // up by 2 or more -- run here, in a loop of their own whose only loop-carried state is what such a step changes;
        // anything else (a pass-through at unity speed, 1 < speed < 2, frames still to be copied) leaves it for ONE pass of
        // the general code below.  Same arithmetic: run_event with ge2 a constant.
        if (hotStream) {
          for (;;) {
            const unsigned long long runnable = __builtin_amdgcn_ballot_w64(
                lane >= i && lane < nIn && (unityLane || availLane - base >= maxRequired));
            if (SPX_UNLIKELY(runnable == 0)) break;
            const int e = __builtin_ctzll(runnable);
            const float speed = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, spv), e));
            if (SPX_UNLIKELY(((unityMask >> e) & 1) != 0 || !(speed >= 2.0f) || remaining > 0)) break;
            i = e;
            const pos_t availE = (linear && !flushBlk) ? n_tsm : availBlk + (i + 1) * perEvent;
            FSTAMP(0);
            SPX_PROBE(10);  // once per EVENT of the hot loop (not per step)
            const float sm1 = speed - 1.0f;
            const float rinv = fast_rcp_refined(sm1);
            pos_t pos = base;
            bool failed;
            do {
              int n, rem;
              const int period = find_period(pos, true, sm1, 0.0f, rinv, n, rem, std::true_type());
              if (out_n + n > X.out_cap) overflow = 1;
              failed = n == 0;
              xf_n = n; xf_down = (int)(pos - wbase); xf_period = period; xf_out = out_n;
              out_n += n;
              pos += failed ? 0 : period + n;
              FSTAMP(10);
            } while (!failed && pos + maxRequired <= availE);
            if (!failed) base = pos;
            FSTAMP(11);
            i++;
          }
        }
        // ... and the same for a stream that runs at 1 < speed < 2: steps with ge2 = false a constant, each followed by the pass
        // that copies what it left to copy; events at speed >= 2 or unity leave for the general code (16 kHz mono 1.5x: walk
        // 1.50 -> 1.46 ms per 256 x 10 s, 22.05 kHz stereo 1.5x 1.25 -> 1.18; profiles/r03/r03at_hot_lt2.txt)
#ifndef SPX_NO_HOT_LT2
        else {
          for (;;) {
            const unsigned long long runnable = __builtin_amdgcn_ballot_w64(
                lane >= i && lane < nIn && (unityLane || availLane - base >= maxRequired));
            if (SPX_UNLIKELY(runnable == 0)) break;
            const int e = __builtin_ctzll(runnable);
            const float speed = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, spv), e));
            if (SPX_UNLIKELY(((unityMask >> e) & 1) != 0 || !(speed < 2.0f) || (SLOWK && !(speed > 1.0f)))) break;
            i = e;
            const pos_t availE = (linear && !flushBlk) ? n_tsm : availBlk + (i + 1) * perEvent;
            const float sm1 = speed - 1.0f, twom = 2.0f - speed;
            const float rinv = fast_rcp_refined(sm1);
            pos_t pos = base;
            bool failed = false;
            do {
              if (remaining > 0) {
                int n = remaining;
                if (n > maxRequired) n = maxRequired;
                if (out_n + n > X.out_cap) overflow = 1;
                FAST_PUBLISH(FCMD_COPY, n, pos, out_n, 0);
                if (NWC > 0) fast_sync();
                out_n += n;
                remaining -= n;
                pos += n;
              } else {
                int n, rem;
                const int period = find_period(pos, false, sm1, twom, rinv, n, rem, std::true_type());
                remaining = rem;
                if (out_n + n > X.out_cap) overflow = 1;
                failed = n == 0;
                xf_n = n; xf_down = (int)(pos - wbase); xf_period = period; xf_out = out_n;
                out_n += n;
                pos += failed ? 0 : period + n;
              }
            } while (!failed && pos + maxRequired <= availE);
            if (!failed) base = pos;
            i++;
          }
        }
#endif
        // the general code: one event
        if (i >= nIn) break;
        const unsigned long long runnable = __builtin_amdgcn_ballot_w64(
            lane >= i && lane < nIn && (unityLane || availLane - base >= maxRequired));
        if (runnable == 0) break;
        i = __builtin_ctzll(runnable);
        const pos_t availE = (linear && !flushBlk) ? n_tsm : availBlk + (i + 1) * perEvent;
        FSTAMP(0);
        if ((unityMask >> i) & 1) {
          const pos_t n = availE - base;
          if (n > 0) {
            if (out_n + n > X.out_cap) overflow = 1;
            FAST_PUBLISH(FCMD_COPY, n, base, out_n, 0);
            if (NWC > 0) fast_sync();
            out_n += n;
          }
          base = availE;
        } else {
          const float speed = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, spv), i));
          if (SLOWK && speed < 1.0f) run_event_slow(speed, availE);
          else run_event(speed, availE);
        }
        FSTAMP(11);
        i++;
      }
      avail = (linear && !flushBlk) ? n_tsm : availBlk + nIn * perEvent;
      if (flushBlk) {
        if (out_n > expected) out_n = expected;
        base = avail;  // the dependency empties its input after a flush
        remaining = 0;
      }
    }
    if (!linear) handed = ev1;
    if (last) break;
  }
  FAST_PUBLISH(FCMD_EXIT, 0, 0, 0, 0);
  if (NWC > 0) fast_sync();
  FSTAMP_FLUSH
  if (tid == 0) {
    SpxWalkState W;
    W.base = base; W.out_n = out_n; W.avail = avail; W.remaining = remaining;
    W.prevPeriod = prevPeriod; W.prevMinDiff = prevMinDiff; W.overflow = overflow;
    W.prevPeriod_toggle = 0;
    // field-wise: the tension kernel may be writing its own fields of this record (and, with output waves, `steps` is theirs)
    SpxWalkState& D = states[blockIdx.x].w;
    D.base = W.base; D.out_n = W.out_n; D.avail = W.avail; D.remaining = W.remaining; D.prevPeriod = W.prevPeriod;
    D.prevMinDiff = W.prevMinDiff; D.overflow = W.overflow; D.prevPeriod_toggle = 0;
    if constexpr (NWC == 0) D.steps = ((S.flags & SPX_F_INIT) ? 0 : D.steps) + nsteps;
    states[blockIdx.x].curSpeed = tailSpeed;
    if (!linear) states[blockIdx.x].handed = (int)handed;
    // a truncated output is reported as a negative count; a lost producer as INT64_MIN (SPX_NOUT_LOST_PRODUCER)
    if (n_out) n_out[blockIdx.x] = overflow == 2 ? INT64_MIN : (overflow ? -(int64_t)out_n : (int64_t)out_n);
  }
#undef FAST_PUBLISH
#undef SPX_LEAN_OUTPUTS
}

size_t spx_walk_fast_lds_bytes(const SpxPlanDev& P, int wcap) { return (size_t)fast_lds_layout(P, wcap).total; }

// Can this kernel serve the plan with `nwm` search waves?  (The coarse triangle must fit FCG groups per lane.)
bool spx_walk_fast_supports(const SpxPlanDev& P, int nwm) {
  const int skip = P.skip > 0 ? P.skip : 1;
  const int minC = P.minPeriod / skip, maxC = P.maxPeriod / skip;
  int total = 0;
  for (int p = minC; p <= maxC; p++) total += ((p >> 1) + (p & 1) + 3) >> 2;
  int ragged = 0;  // the refine search's ragged triangle (the larger parity) must fit FRG tasks per lane
  for (int t = 0; t < 8 * skip + 1; t++) ragged += (t + 2) >> 1;
  // the refine search's common rectangle: at least one chunk per lag, and the multiply-shift division by NCH exact
  const int nlag = 8 * skip + 1, nch = (64 * nwm) / nlag;
  if (nlag > (nwm == 8 ? 121 : 63)) return false;   // one lag per lane of the refine select (two in the eight-wave form); lane 63 / 127 = the previous period
  if (nwm == 8 && skip < 6) return false;            // (that form's sum buffers exist in the layout from skip 6 on)
  if (nwm < 2) return false;   // (one search wave per stream was tried in round 3: no faster than two; not instantiated)
  const int nC = maxC - minC + 1;
  if (nC > 128 || (nC > 64 && nwm != 4)) return false;   // the coarse select: one lag per lane, two in the wide-coarse instantiations (four search waves)
  if (nch < 3) return false;   // up to three left-over pairs of the rectangle, one per chunk
  const int chM = (65536 + nch - 1) / nch;
  for (int c0 = 0; c0 <= P.maxPeriod / 2 + 1; c0++)
    if (((c0 * chM) >> 16) != c0 / nch) return false;
  return total <= fcg_of(nwm) * 64 * nwm && ragged <= frg_of(nwm) * 64 * nwm;
}

// numRegs of the instantiation spx_launch_walk_fast picks for (nwm, nwc) at this plan's rate
int spx_walk_fast_vgprs(const SpxPlanDev& P, int nwm, int nwc, int wcap, int maxC, int* scratch_bytes, bool slow) {
  const void* fn = nullptr;
  if (fast_wide_coarse(P.minPeriod, P.maxPeriod, P.skip)) {   // the wide-coarse instantiations (SPEC = 2): plan-driven, 4 + 4 or 4 + 0 waves
#define SPX_FN_WC(C) (slow ? (maxC > 1 ? reinterpret_cast<const void*>(spx_walk_fast_kernel<4, C, 0, 2, 3>) : reinterpret_cast<const void*>(spx_walk_fast_kernel<4, C, 0, 2, 2>)) \
                           : (maxC > 1 ? reinterpret_cast<const void*>(spx_walk_fast_kernel<4, C, 0, 2, 1>) : reinterpret_cast<const void*>(spx_walk_fast_kernel<4, C, 0, 2, 0>)))
    fn = nwc >= 4 ? SPX_FN_WC(4) : SPX_FN_WC(0);
#undef SPX_FN_WC
    return spx_kernel_vgprs(fn, scratch_bytes);
  }
  if (slow) {   // the plan-driven instantiations that also serve speeds below 1 (MC + 2)
#define SPX_FN_SLOW(M, C) (maxC > 1 ? reinterpret_cast<const void*>(spx_walk_fast_kernel<M, C, 0, 0, 3>) : reinterpret_cast<const void*>(spx_walk_fast_kernel<M, C, 0, 0, 2>))
    fn = nwm == 8 ? SPX_FN_SLOW(8, 4) : nwm == 2 ? SPX_FN_SLOW(2, 0) : (nwc >= 4 ? SPX_FN_SLOW(4, 4) : SPX_FN_SLOW(4, 0));
#undef SPX_FN_SLOW
    return spx_kernel_vgprs(fn, scratch_bytes);
  }
#define SPX_FN_RM(M, C, MCV) (P.rate == 16000 && wcap == SPX_CT_WCAP_OF(M, C) ? reinterpret_cast<const void*>(spx_walk_fast_kernel<M, C, 16000, 0, MCV>) \
                        : P.rate == 22050 && wcap == SPX_CT_WCAP_OF(M, C) ? reinterpret_cast<const void*>(spx_walk_fast_kernel<M, C, 22050, 0, MCV>) \
                        : reinterpret_cast<const void*>(spx_walk_fast_kernel<M, C, 0, 0, MCV>))
#define SPX_FN_R(M, C) (maxC > 1 ? SPX_FN_RM(M, C, 1) : SPX_FN_RM(M, C, 0))
  if (nwm == 4 && nwc >= 4 && wcap == SPX_CT_WCAP_LONG && (P.rate == 16000 || P.rate == 22050))
    fn = P.rate == 16000 ? (maxC > 1 ? reinterpret_cast<const void*>(spx_walk_fast_kernel<4, 4, 16000, 1, 1>) : reinterpret_cast<const void*>(spx_walk_fast_kernel<4, 4, 16000, 1, 0>))
                         : (maxC > 1 ? reinterpret_cast<const void*>(spx_walk_fast_kernel<4, 4, 22050, 1, 1>) : reinterpret_cast<const void*>(spx_walk_fast_kernel<4, 4, 22050, 1, 0>));
#ifdef SPX_TUNING
  else if (nwm == 8) fn = SPX_FN_R(8, 4);
  else if (nwm == 2) fn = nwc >= 1 ? SPX_FN_R(2, 1) : SPX_FN_R(2, 0);
  else fn = nwc >= 4 ? SPX_FN_R(4, 4) : nwc >= 2 ? SPX_FN_R(4, 2) : nwc >= 1 ? SPX_FN_R(4, 1) : SPX_FN_R(4, 0);
#else
  // the shipped library carries the forms spx_walk_config selects by itself (SPX_FAST_FORMS below)
  else if (nwm == 8) fn = maxC > 1 ? reinterpret_cast<const void*>(spx_walk_fast_kernel<8, 4, 0, 0, 1>) : reinterpret_cast<const void*>(spx_walk_fast_kernel<8, 4, 0, 0, 0>);
  else if (nwm == 2) fn = SPX_FN_R(2, 0);
  else fn = nwc >= 4 ? SPX_FN_R(4, 4) : SPX_FN_R(4, 0);
#endif
#undef SPX_FN_R
#undef SPX_FN_RM
  return spx_kernel_vgprs(fn, scratch_bytes);
}

void spx_launch_walk_fast(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, const int16_t* in,
                          int16_t* out, int64_t* n_out, SpxStreamState* states, const float* scratch,
                          const int* speed_ready, int nwm, int nwc, int wcap, int maxC, hipStream_t st, size_t lds_min, bool slow) {
  if (n_streams <= 0) return;
  const FastLds LY = fast_lds_layout(P, wcap);
  // lds_min: the caller wants these workgroups ONE to a CU (it asks for more than half a CU's LDS): walk kernels of several
  // groups launched side by side otherwise land two to a CU here and there, and those chains end the call (spx_engine.hip)
  const size_t lds_req = (size_t)LY.total > lds_min ? (size_t)LY.total : lds_min;
#define SPX_LAUNCH_FAST_RSM(M, C, R, SPECV, MCV)                                                                               \
  hipLaunchKernelGGL((spx_walk_fast_kernel<M, C, R, SPECV, MCV>), dim3(n_streams), dim3(64 * (M + C)), lds_req, st, P, streams, \
                     in, out, n_out, states, scratch, speed_ready, wcap)
#define SPX_LAUNCH_FAST_RS(M, C, R, SPECV)                                                                                \
  do { if (maxC > 1) SPX_LAUNCH_FAST_RSM(M, C, R, SPECV, 1); else SPX_LAUNCH_FAST_RSM(M, C, R, SPECV, 0); } while (0)
#define SPX_LAUNCH_FAST_R(M, C, R) SPX_LAUNCH_FAST_RS(M, C, R, 0)
  // the two rates of the BASELINE configs get their own specialisation (with the default 4096-frame window)
#define SPX_LAUNCH_FAST(M, C)                                              \
  do {                                                                     \
    if (P.rate == 16000 && wcap == SPX_CT_WCAP_OF(M, C)) SPX_LAUNCH_FAST_R(M, C, 16000);   \
    else if (P.rate == 22050 && wcap == SPX_CT_WCAP_OF(M, C)) SPX_LAUNCH_FAST_R(M, C, 22050); \
    else SPX_LAUNCH_FAST_R(M, C, 0);                                       \
  } while (0)
#ifdef SPX_STAMPS
  SPX_LAUNCH_FAST(4, 4);
  return;
#endif
  if (fast_wide_coarse(P.minPeriod, P.maxPeriod, P.skip)) {
    // more than 64 coarse lags (11.025 kHz): the wide-coarse instantiations (SPEC = 2), plan-driven, 4 + 4 or 4 + 0 waves
    // (spx_walk_fast_supports admits four search waves only there)
#define SPX_LAUNCH_WC(C)                                                                                                  \
  do {                                                                                                                    \
    if (slow) { if (maxC > 1) SPX_LAUNCH_FAST_RSM(4, C, 0, 2, 3); else SPX_LAUNCH_FAST_RSM(4, C, 0, 2, 2); }              \
    else { if (maxC > 1) SPX_LAUNCH_FAST_RSM(4, C, 0, 2, 1); else SPX_LAUNCH_FAST_RSM(4, C, 0, 2, 0); }                   \
  } while (0)
    if (nwc >= 4) SPX_LAUNCH_WC(4); else SPX_LAUNCH_WC(0);
#undef SPX_LAUNCH_WC
    return;
  }
  if (slow) {
    // batches with slow-down jobs: the plan-driven instantiations with the insertPitchPeriod event (MC + 2), the forms
    // spx_walk_config picks (4 + 4, 4 + 0, 2 + 0, 8 + 4)
#define SPX_LAUNCH_SLOW(M, C) do { if (maxC > 1) SPX_LAUNCH_FAST_RSM(M, C, 0, 0, 3); else SPX_LAUNCH_FAST_RSM(M, C, 0, 0, 2); } while (0)
    if (nwm == 8) SPX_LAUNCH_SLOW(8, 4);
    else if (nwm == 2) SPX_LAUNCH_SLOW(2, 0);
    else if (nwc >= 4) SPX_LAUNCH_SLOW(4, 4);
    else SPX_LAUNCH_SLOW(4, 0);
#undef SPX_LAUNCH_SLOW
    return;
  }
  // SPX_FAST_FORMS -- the forms the SHIPPED library carries are the ones spx_walk_config selects by itself: 4 + 4 waves (one or two
  // streams per CU; with the long window at the two BASELINE rates), 4 + 0 (lean form, short jobs beyond one stream per CU),
  // 2 + 0 (throughput form), each rate-specialised for 16 / 22.05 kHz and generic, mono-only and multi-channel; 8 + 4 generic
  // for the rates whose ragged refine tasks do not fit four search waves (24 .. 32 kHz).  The other combinations (1 or 2
  // output waves, two search waves with an output wave, eight search waves at a BASELINE rate) were A/B points of rounds 2
  // and 3; they exist in builds with -DSPX_TUNING, where SPX_WALK_NWM / NWC select them (`make tuning`).  52 -> 24 kernels.
  if (nwm == 4 && nwc >= 4 && wcap == SPX_CT_WCAP_LONG && (P.rate == 16000 || P.rate == 22050)) {
    if (P.rate == 16000) SPX_LAUNCH_FAST_RS(4, 4, 16000, 1); else SPX_LAUNCH_FAST_RS(4, 4, 22050, 1);
#ifdef SPX_TUNING
  } else if (nwm == 8) {
    SPX_LAUNCH_FAST(8, 4);
  } else if (nwm == 2) {
    if (nwc >= 1) SPX_LAUNCH_FAST(2, 1);
    else SPX_LAUNCH_FAST(2, 0);
  } else {
    if (nwc >= 4) SPX_LAUNCH_FAST(4, 4);
    else if (nwc >= 2) SPX_LAUNCH_FAST(4, 2);
    else if (nwc >= 1) SPX_LAUNCH_FAST(4, 1);
    else SPX_LAUNCH_FAST(4, 0);
  }
#else
  } else if (nwm == 8) {
    SPX_LAUNCH_FAST_R(8, 4, 0);
  } else if (nwm == 2) {
    SPX_LAUNCH_FAST(2, 0);
  } else {
    if (nwc >= 4) SPX_LAUNCH_FAST(4, 4);
    else SPX_LAUNCH_FAST(4, 0);
  }
#endif
#undef SPX_LAUNCH_FAST
#undef SPX_LAUNCH_FAST_RS
#undef SPX_LAUNCH_FAST_RSM
#undef SPX_LAUNCH_FAST_R
}
