// Walk kernel for gfx950: one 256-thread workgroup per stream.
//
// Prologue (frame rate, O(1) work per 10 ms frame; order-sensitive recurrences run on one lane in the
// reference's order, everything else one lane per frame):
//   a6  energy low-pass, local energy, sqrt compression   speedy.c:517-521,73-76
//   a7  tapered-max temporal hysteresis                   speedy.c:590-610
//   a8  low-energy gate, emphasis weighting, difference low-pass, relative difference, clamp
//                                                         speedy.c:682-700,720-728
//   a7  tension                                           speedy.c:752-766
//   a9  speed from tension (+ duration feedback), blend   speedy.c:768-788, soniclib.c:339-345
// Walk (sample rate, inherently sequential per stream: each step's position depends on the last period):
//   a10 AMDF pitch search on the decimated then the full-rate signal   (libsonic)
//   a11 skip / insert pitch periods with a linear cross-fade, FIFO bookkeeping, flush padding
//       driven exactly as the shim drives it: one (setSpeed, write frameStep samples) pair per tension
//       frame (soniclib.c:354,369), the un-analysed tail at the last speed (soniclib.c:538-550), then
//       sonicIntFlushStream (soniclib.c:551).
// All sample arithmetic is integer; results are bit-exact against oracle/orc_sonic.c.
#include "spx_internal.h"

#define SPX_CH 1024  // frames per prologue chunk held in LDS

typedef SpxWalkState WalkState;

struct Cand {  // AMDF candidate: diff over `p` terms
  unsigned diff;
  int p;  // 0 = empty
};

__device__ __forceinline__ Cand cand_min(Cand a, Cand b) {
  if (b.p == 0) return a;
  if (a.p == 0) return b;
  const unsigned long long l = (unsigned long long)a.diff * (unsigned)b.p;
  const unsigned long long r = (unsigned long long)b.diff * (unsigned)a.p;
  if (l < r) return a;
  if (r < l) return b;
  return a.p < b.p ? a : b;
}
__device__ __forceinline__ Cand cand_max(Cand a, Cand b) {
  if (b.p == 0) return a;
  if (a.p == 0) return b;
  const unsigned long long l = (unsigned long long)a.diff * (unsigned)b.p;
  const unsigned long long r = (unsigned long long)b.diff * (unsigned)a.p;
  if (l > r) return a;
  if (r > l) return b;
  return a.p < b.p ? a : b;
}
__device__ __forceinline__ Cand cand_shfl_xor(Cand c, int m) {
  Cand o;
  o.diff = (unsigned)__shfl_xor((int)c.diff, m);
  o.p = __shfl_xor(c.p, m);
  return o;
}

struct WalkCtx {
  const int16_t* in;  // stream input (interleaved)
  int16_t* out;       // stream output
  int64_t out_cap;
  int64_t zero_from;  // absolute frame index from which reads return 0 (flush padding)
  int C;
  short* sMono;       // [maxRequired] mono mix of the current window
  short* sDown;       // [maxRequired/skip] decimated window
  Cand* sCand;        // [8] per-wave partial results
};

__device__ __forceinline__ int raw_sample(const WalkCtx& X, int64_t a, int c) {
  return (a < X.zero_from) ? (int)X.in[a * X.C + c] : 0;
}

// AMDF over lags [minP, maxP] on x (LDS).  Returns best/worst exactly as a sequential scan that keeps the
// FIRST lag with the smallest (largest) diff/lag would.  4 lanes per lag, 64 lags per pass.
__device__ __forceinline__ void amdf_search(const WalkCtx& X, const short* x, int minP, int maxP, int* retBest, int* retMin,
                            int* retMax) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = tid & 3;
  Cand bmin = {0u, 0}, bmax = {0u, 0};
  for (int l0 = 0; minP + l0 <= maxP; l0 += 64) {
    const int p = minP + l0 + (tid >> 2);
    unsigned acc = 0;
    if (p <= maxP) {
      for (int i = q; i < p; i += 4) {
        const int d = (int)x[i] - (int)x[i + p];
        acc += (unsigned)(d < 0 ? -d : d);
      }
    }
    acc += (unsigned)__shfl_xor((int)acc, 1);
    acc += (unsigned)__shfl_xor((int)acc, 2);
    Cand c = {acc, (p <= maxP && q == 0) ? p : 0};
    Cand cmn = c, cmx = c;
    for (int m = 4; m < 64; m <<= 1) {
      cmn = cand_min(cmn, cand_shfl_xor(cmn, m));
      cmx = cand_max(cmx, cand_shfl_xor(cmx, m));
    }
    __syncthreads();  // previous users of sCand are done
    if (lane == 0) {
      X.sCand[wave] = cmn;
      X.sCand[4 + wave] = cmx;
    }
    __syncthreads();
    for (int w = 0; w < 4; w++) {
      bmin = cand_min(bmin, X.sCand[w]);
      bmax = cand_max(bmax, X.sCand[4 + w]);
    }
  }
  // sequential-scan initial state for the maximum is (maxDiff = 0, worstPeriod = 255): it is only replaced
  // by a lag with diff > 0.
  int worst = 255;
  unsigned maxDiff = 0;
  if (bmax.p != 0 && bmax.diff > 0) {
    worst = bmax.p;
    maxDiff = bmax.diff;
  }
  *retBest = bmin.p;
  *retMin = (int)(bmin.diff / (unsigned)bmin.p);
  *retMax = (int)(maxDiff / (unsigned)worst);
}

// findPitchPeriod at absolute position pos (all threads return the same value).
__device__ __forceinline__ int find_pitch_period(const SpxPlanDev& P, const WalkCtx& X, WalkState& st, int64_t pos) {
  const int tid = threadIdx.x;
  const int C = X.C, skip = P.skip, maxRequired = P.maxRequired;
  __syncthreads();  // earlier readers of sMono/sDown are done
  for (int t = tid; t < maxRequired; t += SPX_BLOCK) {
    int v;
    if (C == 1) {
      v = raw_sample(X, pos + t, 0);
    } else {
      int sum = 0;
      for (int c = 0; c < C; c++) sum += raw_sample(X, pos + t, c);
      v = sum / C;
    }
    X.sMono[t] = (short)v;
  }
  if (skip > 1 || C > 1) {
    const int cnt = maxRequired / skip;
    for (int t = tid; t < cnt; t += SPX_BLOCK) {
      int sum = 0;
      for (int j = 0; j < skip; j++)
        for (int c = 0; c < C; c++) sum += raw_sample(X, pos + (int64_t)t * skip + j, c);
      X.sDown[t] = (short)(sum / (skip * C));
    }
  }
  __syncthreads();
  int period, minDiff, maxDiff;
  if (C == 1 && skip == 1) {
    amdf_search(X, X.sMono, P.minPeriod, P.maxPeriod, &period, &minDiff, &maxDiff);
  } else {
    amdf_search(X, X.sDown, P.minPeriod / skip, P.maxPeriod / skip, &period, &minDiff, &maxDiff);
    if (skip != 1) {
      period *= skip;
      int lo = period - (skip << 2), hi = period + (skip << 2);
      if (lo < P.minPeriod) lo = P.minPeriod;
      if (hi > P.maxPeriod) hi = P.maxPeriod;
      amdf_search(X, X.sMono, lo, hi, &period, &minDiff, &maxDiff);
    }
  }
  int ret = period;
  if (!(minDiff == 0 || st.prevPeriod == 0) && !(maxDiff > minDiff * 3) && !(minDiff * 2 <= st.prevMinDiff * 3))
    ret = st.prevPeriod;
  st.prevMinDiff = minDiff;
  st.prevPeriod = period;
  return ret;
}

// Append n frames copied from absolute input position a.
__device__ __forceinline__ void emit_copy(const WalkCtx& X, WalkState& st, int64_t a, int64_t n) {
  const int C = X.C;
  if (st.out_n + n > X.out_cap) st.overflow = 1;
  const int64_t total = n * C;
  for (int64_t e = threadIdx.x; e < total; e += SPX_BLOCK) {
    const int64_t f = e / C;
    const int c = (int)(e - f * C);
    if (st.out_n + f < X.out_cap) X.out[(st.out_n + f) * C + c] = (int16_t)raw_sample(X, a + f, c);
  }
  st.out_n += n;
}

// Append n frames of cross-fade: out[t] = (down[t]*(n-t) + up[t]*t)/n, integer, truncating.
__device__ __forceinline__ void emit_overlap_add(const WalkCtx& X, WalkState& st, int64_t a_down, int64_t a_up, int n,
                                 int64_t out_at) {
  const int C = X.C;
  const int total = n * C;
  for (int e = threadIdx.x; e < total; e += SPX_BLOCK) {
    const int t = e / C, c = e - t * C;
    const int d = raw_sample(X, a_down + t, c), u = raw_sample(X, a_up + t, c);
    if (out_at + t < X.out_cap) X.out[(out_at + t) * C + c] = (int16_t)((d * (n - t) + u * t) / n);
  }
}

// processStreamInput with `avail` frames handed over so far (absolute count).
__device__ __forceinline__ void tsm_process(const SpxPlanDev& P, const WalkCtx& X, WalkState& st, float speed, int64_t avail) {
  const int maxRequired = P.maxRequired;
  if ((double)speed > 1.00001 || (double)speed < 0.99999) {
    const int64_t numSamples = avail - st.base;
    if (numSamples < maxRequired) return;
    int64_t position = 0;
    do {
      if (st.remaining > 0) {
        int n = st.remaining;
        if (n > maxRequired) n = maxRequired;
        emit_copy(X, st, st.base + position, n);
        st.remaining -= n;
        position += n;
      } else {
        const int64_t pos = st.base + position;
        const int period = find_pitch_period(P, X, st, pos);
        if ((double)speed > 1.0) {
          long n;
          if (speed >= 2.0f) {
            n = (long)((float)period / (speed - 1.0f));
          } else {
            n = period;
            st.remaining = (int)((float)period * (2.0f - speed) / (speed - 1.0f));
          }
          if (st.out_n + n > X.out_cap) st.overflow = 1;
          emit_overlap_add(X, st, pos, pos + period, (int)n, st.out_n);
          st.out_n += n;
          if (n == 0) return;  // the dependency treats this as failure and leaves the input untouched
          position += period + n;
        } else {
          long n;
          if (speed < 0.5f) {
            n = (long)((float)period * speed / (1.0f - speed));
          } else {
            n = period;
            st.remaining = (int)((float)period * (2.0f * speed - 1.0f) / (1.0f - speed));
          }
          emit_copy(X, st, pos, period);
          if (st.out_n + n > X.out_cap) st.overflow = 1;
          emit_overlap_add(X, st, pos + period, pos, (int)n, st.out_n);
          st.out_n += n;
          if (n == 0) return;
          position += n;
        }
      }
    } while (position + maxRequired <= numSamples);
    st.base += position;
  } else {
    emit_copy(X, st, st.base, avail - st.base);
    st.base = avail;
  }
}

__global__ void __launch_bounds__(SPX_BLOCK)
spx_walk_kernel(SpxPlanDev P, const SpxStreamDev* __restrict__ streams, const int16_t* __restrict__ in_base,
                int16_t* __restrict__ out_base, int64_t* __restrict__ n_out, SpxStreamState* __restrict__ states,
                const SpxFrameRec* __restrict__ rec_base, float* __restrict__ scratch_base, SpxTapsDev taps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const SpxStreamDev S = streams[blockIdx.x];
  const int T = S.n_frames, F = P.F, Pp = P.Pp, B = P.B;
  const float Rg = S.speed, nl = S.nonlinear, fb = S.feedback;

  float* sA = reinterpret_cast<float*>(lds);  // [SPX_CH]
  float* sB = sA + SPX_CH;                    // [SPX_CH]
  short* sMono = reinterpret_cast<short*>(sB + SPX_CH);
  short* sDown = sMono + ((P.maxRequired + 7) & ~7);
  Cand* sCand = reinterpret_cast<Cand*>(sDown + ((P.maxRequired + 7) & ~7));

  // ---- state carried between jobs of one stream ----
  SpxStreamState Z;
  if (S.flags & SPX_F_INIT) {
    Z.w.base = 0; Z.w.out_n = 0; Z.w.avail = 0; Z.w.remaining = 0; Z.w.prevPeriod = 0; Z.w.prevMinDiff = 0;
    Z.w.overflow = 0;
    Z.lp = 2.14204f;    // speedy.c:263,288
    Z.lpf = 123.837f;   // speedy.c:264,291
    Z.cur_dur = 0.0f; Z.des_dur = 0.0f;
    Z.curSpeed = Rg;    // sonicSetSpeed -> sonicIntSetSpeed, soniclib.c:182
    Z.handed = 0;
  } else {
    Z = states[blockIdx.x];
    // sonicSetSpeed between writes reaches the TSM stage at once (soniclib.c:182); in nonlinear mode the
    // next tension frame overrides it (soniclib.c:354)
    if (nl == 0.0f) Z.curSpeed = Rg;
  }

  const SpxFrameRec* rec = rec_base + S.frame_off;
  float* scr = scratch_base + (size_t)S.frame_off * 4;  // per frame: comp, hyst, ewld->tension, speed
  const int fa = S.frame_begin;                                   // frames already folded into the state
  const int K0 = (nl != 0.0f && fa >= F) ? fa - F + 1 : 0;        // tension frames already done
  const int K = (nl != 0.0f && T >= F) ? T - F + 1 : 0;           // tension frames available (soniclib.c:317)
  const float lowthr = (float)(0.04 * (double)1.41421f);          // speedy.c:682
  float* tfeat = taps.features ? taps.features + (size_t)S.frame_off * SPX_FEATURE_COUNT : nullptr;

  if (nl != 0.0f && T > fa) {
    // ---- pass 1: energy low-pass (sequential) -> local -> compressed ----
    float lp = Z.lp;
    for (int c0 = fa; c0 < T; c0 += SPX_CH) {
      const int n = min(SPX_CH, T - c0);
      for (int i = tid; i < n; i += SPX_BLOCK) sA[i] = rec[c0 + i].energy;
      __syncthreads();
      if (tid == 0) {
        for (int i = 0; i < n; i++) {
          lp = P.one_minus_alpha * sA[i] + P.alpha * lp;  // speedy.c:74
          sB[i] = lp;
        }
      }
      __syncthreads();
      for (int i = tid; i < n; i += SPX_BLOCK) {
        const float e = sA[i], l = sB[i];
        const float local = e / l;                                               // speedy.c:519
        const float comp = (float)__builtin_sqrt(local > 2 ? 2.0 : (double)local);  // speedy.c:520
        const int j = c0 + i;
        scr[4 * j + 0] = comp;
        const int k = j - F + 1;  // the tension frame whose callback sees these AddData-time values
        if (tfeat && k >= 0) {
          float* f = tfeat + (size_t)k * SPX_FEATURE_COUNT;
          f[1] = l; f[2] = local; f[3] = comp; f[12] = (float)(j + 1);
        }
      }
      if (n > 0) lp = sB[n - 1];  // every lane keeps the carried state
      __syncthreads();
    }
    Z.lp = lp;
    // ---- pass 2: hysteresis and emphasis-weighted difference, one lane per tension frame ----
    for (int k = K0 + tid; k < K; k += SPX_BLOCK) {
      float future_max = 0.0f, past_max = 0.0f;
      for (int i = 0; i <= F; i++) {
        const int tau = k + i;  // hysteresis slot `tau` holds frame tau-1; slots <= 0 are the zero init
        float v = (tau >= 1) ? scr[4 * (tau - 1) + 0] : 0.0f;
        v *= P.taperF[i];
        if (v > future_max) future_max = v;
      }
      for (int i = 0; i <= Pp; i++) {
        const int tau = k - i;
        float v = (tau >= 1) ? scr[4 * (tau - 1) + 0] : 0.0f;
        v *= P.taperP[i];
        if (v > past_max) past_max = v;
      }
      const float hyst = (float)((double)(past_max + future_max) / 2.0);  // speedy.c:609
      const float e_cur = (k == 0) ? 0.0f : rec[k - 1].energy;           // history slot k holds frame k-1
      const bool low = e_cur <= lowthr;
      const float lsd = (k == 0 || low) ? 0.0f : rec[k - 1].lsd;
      const float ewld = low ? 0.0f : lsd * hyst;                          // speedy.c:720
      scr[4 * k + 1] = hyst;
      scr[4 * k + 2] = ewld;
      if (tfeat) {
        float* f = tfeat + (size_t)k * SPX_FEATURE_COUNT;
        f[0] = e_cur; f[4] = hyst; f[5] = low ? 1.0f : 0.0f; f[6] = lsd; f[7] = ewld;
        f[13] = (float)k; f[14] = lowthr;
      }
    }
    __syncthreads();
    // ---- pass 3: difference low-pass (sequential) -> relative difference -> tension -> raw speed ----
    float lpf = Z.lpf;
    for (int c0 = K0; c0 < K; c0 += SPX_CH) {
      const int n = min(SPX_CH, K - c0);
      for (int i = tid; i < n; i += SPX_BLOCK) sA[i] = scr[4 * (c0 + i) + 2];
      __syncthreads();
      if (tid == 0) {
        for (int i = 0; i < n; i++) {
          lpf = P.one_minus_alpha * sA[i] + P.alpha * lpf;
          sB[i] = lpf;
        }
      }
      __syncthreads();
      for (int i = tid; i < n; i += SPX_BLOCK) {
        const int k = c0 + i;
        const float ewld = sA[i], l = sB[i];
        const float hyst = scr[4 * k + 1];
        const float e_cur = (k == 0) ? 0.0f : rec[k - 1].energy;
        const bool low = e_cur <= lowthr;
        float rel = 0.0f, sc = 0.0f;
        if (!low) {
          rel = (float)((double)ewld / ((double)l + 0.01 * (double)123.979f));       // speedy.c:725-726
          sc = (float)fmin((double)rel, (double)(4 * 0.971975f));                    // speedy.c:727-728
        }
        const float a = 0.5f, b = 0.25f, M_E_ = 0.7f, M_S = 1.0f;
        const float tension = a * (hyst - M_E_) + b * (sc - M_S);                    // speedy.c:761
        float v;
        if ((double)Rg > 1.0) {
          v = (float)fmax(1.0, (double)(Rg + (1 - Rg) * tension));                   // speedy.c:774
        } else {
          v = (float)fmax(0.01, fmin(1.0, (double)(Rg - (1 - Rg) * tension)));       // speedy.c:776
        }
        scr[4 * k + 2] = tension;
        scr[4 * k + 3] = v;
        if (tfeat) {
          float* f = tfeat + (size_t)k * SPX_FEATURE_COUNT;
          f[8] = l; f[9] = rel; f[10] = sc; f[11] = tension;
        }
        if (taps.tension) taps.tension[S.frame_off + k] = tension;
      }
      if (n > 0) lpf = sB[n - 1];
      __syncthreads();
    }
    Z.lpf = lpf;
    // ---- pass 4: duration feedback (sequential) and blend with the global speed ----
    float cur_dur = Z.cur_dur, des_dur = Z.des_dur;
    const float fd = (float)(1.0 / 100.0);  // speedy.c:783
    for (int c0 = K0; c0 < K; c0 += SPX_CH) {
      const int n = min(SPX_CH, K - c0);
      for (int i = tid; i < n; i += SPX_BLOCK) sA[i] = scr[4 * (c0 + i) + 3];
      __syncthreads();
      if (tid == 0) {
        for (int i = 0; i < n; i++) {
          float req = sA[i];
          if (fb > 0) {
            const float excess = cur_dur - des_dur;
            req = (float)((double)req + fmax(0.01, (double)(fb * excess)));          // speedy.c:780-781
          }
          cur_dur += fd / req;
          des_dur += fd / Rg;
          sB[i] = req * nl + Rg * (1 - nl);                                          // soniclib.c:344-345
        }
        sA[0] = cur_dur;  // broadcast the carried sums (sA is re-read only by the next chunk's load)
        sA[1] = des_dur;
      }
      __syncthreads();
      cur_dur = sA[0];
      des_dur = sA[1];
      for (int i = tid; i < n; i += SPX_BLOCK) {
        scr[4 * (c0 + i) + 3] = sB[i];
        if (taps.speed) taps.speed[S.frame_off + c0 + i] = sB[i];
      }
      __syncthreads();
    }
    Z.cur_dur = cur_dur;
    Z.des_dur = des_dur;
  }
  __syncthreads();

  // ---------------------------------- the TSM walk ----------------------------------
  WalkCtx X;
  X.in = in_base + S.in_off;
  X.out = out_base + S.out_off;
  X.out_cap = S.out_cap;
  X.zero_from = INT64_MAX;
  X.C = S.channels;
  X.sMono = sMono;
  X.sDown = sDown;
  X.sCand = sCand;
  WalkState st = Z.w;
  float curSpeed = Z.curSpeed;
  int64_t avail = st.avail;
  // Events, in the order the shim issues them:
  //   nonlinear: one (setSpeed, write B) per tension frame           soniclib.c:354,369
  //              at flush, the remaining complete ring buffers at the last speed   soniclib.c:538-550
  //   linear:    one write of everything new (soniclib.c:397-399; chunking is irrelevant at constant speed)
  //   at flush:  sonicIntFlushStream (soniclib.c:551): pad 2*maxRequired zeros, process, truncate
  const bool do_flush = (S.flags & SPX_F_FLUSH) != 0;
  const int64_t ev0 = (nl != 0.0f) ? Z.handed : 0;
  int64_t ev1;  // one past the last ordinary event
  if (nl != 0.0f) ev1 = do_flush ? S.n_in / B : K;  // complete ring buffers written: soniclib.c:446-449
  else ev1 = (S.n_in > avail) ? 1 : 0;
  if (ev1 < ev0) ev1 = ev0;
  const int64_t ev_end = ev1 + (do_flush ? 1 : 0);
  for (int64_t ev = ev0; ev < ev_end; ev++) {
    int64_t expected = 0;
    if (ev < ev1) {
      if (nl != 0.0f) {
        if (ev < K) {
          const int i = (int)((ev - ev0) % SPX_CH);
          if (i == 0) {  // stage the next chunk of speeds in LDS
            __syncthreads();
            const int n = (int)min((int64_t)SPX_CH, (int64_t)K - ev);
            for (int t = tid; t < n; t += SPX_BLOCK) sA[t] = scr[4 * (ev + t) + 3];
            __syncthreads();
          }
          curSpeed = sA[i];
        }
        avail += B;
      } else {
        avail = S.n_in;
      }
    } else {
      const int64_t remainingS = avail - st.base;
      expected = st.out_n + (int)(((float)remainingS / curSpeed + 0) / 1.0f + 0.5f);
      X.zero_from = avail;
      avail += 2 * P.maxRequired;
    }
    tsm_process(P, X, st, curSpeed, avail);
    if (ev >= ev1) {
      if (st.out_n > expected) st.out_n = expected;
      st.base = avail;  // the dependency empties its input after a flush
      st.remaining = 0;
    }
  }
  if (tid == 0) {
    st.avail = avail;
    Z.w = st;
    Z.curSpeed = curSpeed;
    if (nl != 0.0f) Z.handed = (int)ev1;
    states[blockIdx.x] = Z;
    if (n_out) n_out[blockIdx.x] = st.overflow ? -st.out_n : st.out_n;
  }
}

static size_t walk_lds_bytes(const SpxPlanDev& P) {
  size_t mr = (size_t)((P.maxRequired + 7) & ~7);
  return 2 * SPX_CH * sizeof(float) + 2 * mr * sizeof(short) + 8 * sizeof(Cand) + 16;
}

void spx_launch_walk(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, const int16_t* in,
                     int16_t* out, int64_t* n_out, SpxStreamState* states, const SpxFrameRec* rec,
                     float* scratch, SpxTapsDev taps, hipStream_t st) {
  if (n_streams <= 0) return;
  hipLaunchKernelGGL(spx_walk_kernel, dim3(n_streams), dim3(SPX_BLOCK), walk_lds_bytes(P), st, P, streams,
                     in, out, n_out, states, rec, scratch, taps);
}
