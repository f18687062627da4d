// Tension kernel for gfx950: the frame-rate stage between the analysis kernel and the walk kernel.
//
// One workgroup per stream.  O(1) work per 10 ms frame; the order-sensitive recurrences run on one lane in the
// reference's order, everything else one lane per frame:
//   a6  energy low-pass, local energy, sqrt compression   speedy.c:517-521,73-76
//   a7  tapered-max temporal hysteresis                   speedy.c:590-610
//   a8  low-energy gate, emphasis weighting, difference low-pass, relative difference, clamp
//                                                         speedy.c:682-700,720-728
//   a7  tension                                           speedy.c:752-766
//   a9  speed from tension (+ duration feedback), blend   speedy.c:768-788, soniclib.c:339-345
// Output: scratch[4*k + 3] = the speed the shim sets before handing ring buffer k to the time-scale stage
// (soniclib.c:354), plus the tension / speed / features taps.
//
// It is its own kernel so that, in concurrent mode, it runs on a third HIP stream beside the other two: it takes
// the frames in growing chunks as the analysis tiles that cover them publish their flags (agent-scope release there,
// relaxed poll + agent-scope acquire here: cdna_hip_programming.md Guideline 16), and publishes in turn the number of
// tension frames whose speeds are final (`speed_ready`), which the walk kernel polls.  The walk workgroups -- the
// latency-critical chain of the whole job -- then never execute a frame-rate pass themselves.
#include "spx_internal.h"

#define SPX_CH 512   // frames per pass chunk held in LDS
// Frames per chunk when the analysis kernel runs concurrently (a multiple of the 16-frame tile).  Measured on the
// bench batch (ms per step): 16 constant 2.87 | 16,32,48.. 2.95 | 64,128,192.. 3.10 | 16,32,64.. 3.26 -- the finer the
// hand-off, the less the walk kernel ever waits for speeds; this kernel has the slack for the extra passes.
#ifndef SPX_TCH
#define SPX_TCH 16
#endif
#define SPX_TENSION_THREADS 256

// y[i] = a*x[i] + b*y[i-1] for i < n in the reference's rounding order (float product, float product, float sum --
// the recurrence cannot be re-associated), x in sA, y to sB.  Run by the 64 lanes of wave 0: 64 products a*x[i] are
// formed in parallel, then the chain runs on uniform values with a v_readlane per element -- two dependent VALU
// operations per element instead of an LDS round trip.  Returns the last y.
__device__ __forceinline__ float iir_wave(const float* sA, float* sB, int n, float a, float b, float y) {
  const int lane = threadIdx.x;
  for (int b0 = 0; b0 < n; b0 += 64) {
    const int m = n - b0 < 64 ? n - b0 : 64;
    const float ax = a * ((lane < m) ? sA[b0 + lane] : 0.0f);
#ifndef SPX_TENSION_OLD_CHAIN
    if (m == 64) {
      // A full block: the chain runs on lane 0 alone -- per element one v_readlane (it ignores EXEC), the two dependent
      // operations and one LDS store at a constant offset; no per-element bound check, no lane masks.  (Before: ten
      // instructions per element, among them two v_readlane of a spilled lane mask: 63 of the kernel's 113 us per 1 000 frames.)
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 64; j++) {
          const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ax), j));
          y = s + b * y;
          sB[b0 + j] = y;
        }
      }
      y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y)));
      continue;
    }
#endif
#ifndef SPX_TENSION_OLD_CHAIN
    // a partial block (the last one; every block of a concurrent-mode chunk or of a streamed write): the same on lane 0,
    // with the bound check per element
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 64; j++) {
        if (j < m) {  // uniform
          const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ax), j));
          y = s + b * y;
          sB[b0 + j] = y;
        }
      }
    }
    y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y)));
#else
    float out = 0.0f;
#pragma unroll
    for (int j = 0; j < 64; j++) {
      if (j < m) {  // uniform
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ax), j));
        y = s + b * y;
        out = (lane == j) ? y : out;
      }
    }
    if (lane < m) sB[b0 + lane] = out;
#endif
  }
  return y;
}

// REGISTER BUDGET: at most 48 VGPRs.  In the concurrent mode a SIMD holds one wave of this kernel beside the walk and the
// analysis waves, and at 22.05 kHz mono the sum is tight to the register: lean walk 128 + 48 + 2 x 168 (analysis) = 512
// (spx_engine.hip).  A version of pass 2 that staged its inputs in LDS (6 us faster per 1 000 frames) took 49, was
// allocated 56, and silently cost that mode; tests/test_gpu_parity.py::test_register_budgets_of_the_concurrent_mode watches it.
__global__ void __launch_bounds__(SPX_TENSION_THREADS)
spx_tension_kernel(SpxPlanDev P, const SpxStreamDev* __restrict__ streams, SpxStreamState* __restrict__ states,
                   const SpxFrameRec* __restrict__ rec_base, float* __restrict__ scratch_base, SpxTapsDev taps,
                   const int* tile_flags, int* speed_ready) {
  constexpr int NT = SPX_TENSION_THREADS;
  __shared__ float sA[SPX_CH];
  __shared__ float sB[SPX_CH];
  __shared__ int sWait;
  const int tid = threadIdx.x;
  const SpxStreamDev S = streams[blockIdx.x];
  const int Ttot = S.n_frames, F = P.F, Pp = P.Pp;
  const int t0 = S.unit_time0 ? 0 : 1;  // frame j is added at time j + t0; history / hysteresis slot tau holds frame tau - t0
  const float Rg = S.speed, nl = S.nonlinear, fb = S.feedback;
  if (nl == 0.0f) return;  // a linear stream has no frames (uniform per workgroup)

  // the part of the stream state this stage owns
  struct { float lp, lpf, cur_dur, des_dur; int first_k; } Z;
  if (S.flags & SPX_F_INIT) {
    Z.lp = 2.14204f;    // speedy.c:263,288
    Z.lpf = 123.837f;   // speedy.c:264,291
    Z.cur_dur = 0.0f; Z.des_dur = 0.0f;
    Z.first_k = -1;
  } else {
    const SpxStreamState& in = states[blockIdx.x];
    Z.lp = in.lp; Z.lpf = in.lpf; Z.cur_dur = in.cur_dur; Z.des_dur = in.des_dur;
    Z.first_k = in.tension_first;
  }
  const SpxFrameRec* rec = rec_base + S.frame_off;
  float* scr = scratch_base + (size_t)S.frame_off * 4;  // per frame: comp, hyst, ewld->tension, speed
  const float lowthr = (float)(0.04 * (double)1.41421f);          // speedy.c:682
  float* tfeat = taps.features ? taps.features + (size_t)S.frame_off * SPX_FEATURE_COUNT : nullptr;

  // Sequential launches (tile_flags == nullptr): one chunk with every new frame.  Concurrent: chunks of SPX_TCH
  // frames, each started once its analysis tile is published.
  int fa_c = S.frame_begin;
  int wch = SPX_TCH;
  bool ok_all = true;
  for (;;) {
    int T_c = Ttot;
    if (tile_flags != nullptr && Ttot - fa_c > wch) T_c = fa_c + wch;
    const bool last = T_c >= Ttot;
    if (tile_flags != nullptr && T_c > fa_c) {
      const int TF = P.tile_frames;
      const int i0 = (fa_c - S.frame_begin) / TF, i1 = (T_c - S.frame_begin + TF - 1) / TF;
      if (tid == 0) {
        int ok = 1;
        for (int i = i0; i < i1 && ok; i++) {
          unsigned spins = 0;
          while (__hip_atomic_load(&tile_flags[S.first_tile + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1u << 22)) { ok = 0; break; }  // ~seconds: never hang the GPU on a lost producer
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sWait = ok;
      }
      __syncthreads();
      if (sWait == 0) { ok_all = false; break; }
    }
    const int fa = fa_c, T = T_c;
    int K0 = (fa + t0 >= F) ? fa + t0 - F : 0;        // tension frames already done
    if (K0 < S.tension_skip) K0 = S.tension_skip;     // ... or skipped for good by a flush (soniclib.c:538-550)
    int K = (T + t0 >= F) ? T + t0 - F : 0;           // tension frames available: t + F <= current time (speedy.c:756)
    if (S.flags & SPX_F_TENSION_RANGE) {              // the unit-level API asks for its tension frames explicitly
      K0 = S.tension_skip;
      if (K > S.tension_to) K = S.tension_to;
    }
    // the first speedyComputeTension call that succeeds finds skip_frame_count == 1 (speedy.c:293) and is treated as a
    // low-energy frame whatever its time index: frame 0, or the first frame behind a flush that came before any tension
    if (Z.first_k < 0 && K > K0) Z.first_k = K0;
    const int first_k = Z.first_k;
    if (T > fa || (S.flags & SPX_F_TENSION_RANGE)) {  // (a unit-level tension request brings no new frame)
      // ---- pass 1: energy low-pass (sequential) -> local -> compressed ----
      float lp = Z.lp;
      for (int c0 = fa; c0 < T; c0 += SPX_CH) {
        const int n = min(SPX_CH, T - c0);
        for (int i = tid; i < n; i += NT) sA[i] = rec[c0 + i].energy;
        __syncthreads();
        if (tid < 64) lp = iir_wave(sA, sB, n, P.one_minus_alpha, P.alpha, lp);  // speedy.c:74
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
          const float e = sA[i], l = sB[i];
          const float local = e / l;                                               // speedy.c:519
          const float comp = (float)__builtin_sqrt(local > 2 ? 2.0 : (double)local);  // speedy.c:520
          const int j = c0 + i;
          scr[4 * j + 0] = comp;
          const int k = j + t0 - F;  // the tension frame whose callback sees these AddData-time values
          if (tfeat && k >= 0) {
            float* f = tfeat + (size_t)k * SPX_FEATURE_COUNT;
            f[1] = l; f[2] = local; f[3] = comp; f[12] = (float)(j + t0);
          }
        }
        if (n > 0) lp = sB[n - 1];  // every lane keeps the carried state
        __syncthreads();
      }
      Z.lp = lp;
      // ---- pass 2: hysteresis and emphasis-weighted difference, one lane per tension frame ----
      for (int k = K0 + tid; k < K; k += NT) {
        float future_max = 0.0f, past_max = 0.0f;
        for (int i = 0; i <= F; i++) {
          const int tau = k + i;  // hysteresis slot `tau` holds frame tau - t0; earlier slots are the zero init
          float v = (tau >= t0) ? scr[4 * (tau - t0) + 0] : 0.0f;
          v *= P.taperF[i];
          if (v > future_max) future_max = v;
        }
        for (int i = 0; i <= Pp; i++) {
          const int tau = k - i;
          float v = (tau >= t0) ? scr[4 * (tau - t0) + 0] : 0.0f;
          v *= P.taperP[i];
          if (v > past_max) past_max = v;
        }
        const float hyst = (float)((double)(past_max + future_max) / 2.0);  // speedy.c:609
        const float e_cur = (k < t0) ? 0.0f : rec[k - t0].energy;          // history slot k holds frame k - t0
        const bool low = e_cur <= lowthr || k == first_k;                  // the very first call is skipped (speedy.c:692)
        const float lsd = low ? 0.0f : rec[k - t0].lsd;
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
        for (int i = tid; i < n; i += NT) sA[i] = scr[4 * (c0 + i) + 2];
        __syncthreads();
        if (tid < 64) lpf = iir_wave(sA, sB, n, P.one_minus_alpha, P.alpha, lpf);
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
          const int k = c0 + i;
          const float ewld = sA[i], l = sB[i];
          const float hyst = scr[4 * k + 1];
          const float e_cur = (k < t0) ? 0.0f : rec[k - t0].energy;
          const bool low = e_cur <= lowthr || k == first_k;
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
      for (int c0 = K0; c0 < ((S.flags & SPX_F_NO_SPEED) ? K0 : K); c0 += SPX_CH) {
        const int n = min(SPX_CH, K - c0);
        for (int i = tid; i < n; i += NT) sA[i] = scr[4 * (c0 + i) + 3];
        __syncthreads();
        if (!(fb > 0)) {
          // open loop (speedy.c:779 not taken): the requested speeds do not depend on the running sums, so only the
          // two float sums are sequential; quotients and blended speeds are formed 64 at a time
          if (tid < 64) {
            const float cstep = fd / Rg;
            for (int b0 = 0; b0 < n; b0 += 64) {
              const int m = n - b0 < 64 ? n - b0 : 64;
              const float req = (tid < m) ? sA[b0 + tid] : 1.0f;
              const float q = fd / req;
              if (m == 64) {   // a full block: no per-element bound check
#pragma unroll
                for (int j = 0; j < 64; j++) {
                  cur_dur += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), j));
                  des_dur += cstep;
                }
              } else {
#pragma unroll
                for (int j = 0; j < 64; j++) {
                  if (j < m) {  // uniform
                    cur_dur += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), j));
                    des_dur += cstep;
                  }
                }
              }
              if (tid < m) sB[b0 + tid] = req * nl + Rg * (1 - nl);                      // soniclib.c:344-345
            }
            if (tid == 0) { sA[0] = cur_dur; sA[1] = des_dur; }
          }
        } else if (tid == 0) {
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
        for (int i = tid; i < n; i += NT) {
          scr[4 * (c0 + i) + 3] = sB[i];
          if (taps.speed) taps.speed[S.frame_off + c0 + i] = sB[i];
        }
        __syncthreads();
      }
      Z.cur_dur = cur_dur;
      Z.des_dur = des_dur;
    }
    if (speed_ready != nullptr) {
      // publish: every store of this chunk (speeds, taps) is complete and visible device-wide before the count moves
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&speed_ready[blockIdx.x], K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      __syncthreads();
    }
    fa_c = T_c;
    if (last) break;
  }
  if (!ok_all && speed_ready != nullptr && tid == 0)  // lost producer: tell the consumer to give up as well
    __hip_atomic_store(&speed_ready[blockIdx.x], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    SpxStreamState& o = states[blockIdx.x];
    o.lp = Z.lp; o.lpf = Z.lpf; o.cur_dur = Z.cur_dur; o.des_dur = Z.des_dur;
    o.tension_first = Z.first_k;
  }
}

size_t spx_tension_lds_bytes() { return sizeof(float) * 2 * SPX_CH + 64; }
int spx_tension_vgprs() {
  return spx_kernel_vgprs(reinterpret_cast<const void*>(spx_tension_kernel));
}

// speedyComputeSpeedFromTension (speedy.c:768-788), one lane, the arithmetic of passes 3 and 4 above for one frame.
__global__ void spx_speed_kernel(SpxStreamState* state, float tension, float Rg, float fb, float* speed_out) {
  if (threadIdx.x != 0) return;
  float v;
  if ((double)Rg > 1.0) {
    v = (float)fmax(1.0, (double)(Rg + (1 - Rg) * tension));                   // speedy.c:774
  } else {
    v = (float)fmax(0.01, fmin(1.0, (double)(Rg - (1 - Rg) * tension)));       // speedy.c:776
  }
  float cur_dur = state->cur_dur, des_dur = state->des_dur;
  if (fb > 0) {
    const float excess = cur_dur - des_dur;
    v = (float)((double)v + fmax(0.01, (double)(fb * excess)));                // speedy.c:780-781
  }
  const float fd = (float)(1.0 / 100.0);                                       // speedy.c:783
  cur_dur += fd / v;
  des_dur += fd / Rg;
  state->cur_dur = cur_dur;
  state->des_dur = des_dur;
  *speed_out = v;
}
void spx_launch_speed_from_tension(SpxStreamState* state, float tension, float Rg, float feedback, float* speed_out,
                                   hipStream_t st) {
  hipLaunchKernelGGL(spx_speed_kernel, dim3(1), dim3(64), 0, st, state, tension, Rg, feedback, speed_out);
}

void spx_launch_tension(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, SpxStreamState* states,
                        const SpxFrameRec* rec, float* scratch, SpxTapsDev taps, const int* tile_flags, int* speed_ready,
                        hipStream_t st) {
  if (n_streams <= 0) return;
  hipLaunchKernelGGL(spx_tension_kernel, dim3(n_streams), dim3(SPX_TENSION_THREADS), 0, st, P, streams, states, rec,
                     scratch, taps, tile_flags, speed_ready);
}
