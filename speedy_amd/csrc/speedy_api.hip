// The unit-level API of include/speedy.h (reference speedy.h:61-100) on the HIP kernels: one analysis frame per
// speedyAddData, one tension frame per speedyComputeTension, each a launch of the SAME kernels the batch path uses
// (spx_analysis_kernel in its float-frame mode, spx_tension_kernel with an explicit tension range).  Nothing is
// computed on the host; results come back through the tap arrays.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/speedy.h"
#include "spx_internal.h"
#include "spx_log.h"

extern "C" const char* speedyHipLastError(void);
void spx_internal_set_api_error(const std::string& msg);   // sonic2_api.hip
int spx_internal_match_matlab();                            // sonic2_api.hip

#define SPD_KEEP 24  // frames of history kept on the device (the reference keeps 21 spectra, speedy.c:97)

struct speedyStreamStruct {
  const SpxPlanDev* plan = nullptr;
  int device = 0, rate = 0;
  hipStream_t hs = nullptr;
  // sliding device arrays, all starting at frame `origin`, `cap` frames long
  int64_t origin = 0, cap = 0;
  float* dFrames = nullptr;   // [cap][W]
  SpxFrameRec* dRec = nullptr;
  float* dScr = nullptr;      // [cap][4]
  float *tTension = nullptr, *tSpeed = nullptr, *tFeatures = nullptr, *tSpec = nullptr, *tNorm = nullptr;
  unsigned char* dSmall = nullptr;  // job | state | speed out
  SpxStreamDev* dJob = nullptr;
  SpxStreamState* dState = nullptr;
  float* dSpeed = nullptr;
  float* dScratchFrame = nullptr;   // speedySpectrogram: 2 frames in, spectrum out
  float* dScratchSpec = nullptr;
  int64_t T = 0;            // frames added
  int64_t time0 = -1;       // at_time of frame 0 (0 or 1), -1 = not known yet
  int64_t current_time = 0;
  int64_t tensionDone = 0;  // lowest tension frame that may still be asked for
  bool started = false;
  int64_t hookHi = 0;       // one past the highest row a hook has written (rows beyond T that must survive a reallocation)
  float preemph_prev = 0.0f;  // speedyPreemphasisFilter: the last raw sample it (or speedyAddData) has seen (speedy.c:416-425)
  float* dHook = nullptr;     // [0] scalar result of a hook kernel; [8 ..) a frame's worth of scratch floats
  std::vector<float> hSpec, hSpecAt, hNorm, hFeat, hTmp;
  bool aliased = false;       // the last speedyComputeTension answered a stale / repeated time (hAlFeat / hAlNorm hold its rows)
  std::vector<float> hAlFeat, hAlNorm;
};

// Oldest frame the device keeps: the reference remembers 21 spectra and 42 hysteresis values (speedy.c:95-97), i.e. a
// tension older than current_time - 21 cannot be computed there either.  Here the window is SPD_KEEP + Pp + F frames behind
// the NEWEST frame, whatever has or has not been asked for: a caller that only adds frames (or lags with
// speedyComputeTension) no longer keeps every row alive -- tensions older than the window are refused instead.
static int64_t oldest_kept(speedyStream s) {
  const SpxPlanDev& P = *s->plan;
  return std::max<int64_t>(0, s->T - SPD_KEEP - P.Pp - P.F - 8);
}
static bool grow(speedyStream s, int64_t need_hi) {
  // make frames [oldest_kept, need_hi) addressable; everything slides together (one frame_off in the kernels)
  const SpxPlanDev& P = *s->plan;
  const int64_t lo = std::max(oldest_kept(s), s->origin);
  if (s->dFrames && need_hi <= s->origin + s->cap && lo - s->origin <= s->cap / 2) return true;
  const int64_t ncap = std::max<int64_t>(256, 2 * (need_hi - lo));
  const int64_t keep_lo = lo, keep_hi = std::max(s->T + 1, s->hookHi);  // rows that hold data (+1: normalised row T)
  // all new arrays first; the stream's pointers, origin and cap change only when every allocation has succeeded
  struct Arr { void** p; size_t elem; int64_t stride; void* np; };
  Arr arrs[8] = {{reinterpret_cast<void**>(&s->dFrames), sizeof(float), P.W, nullptr},
                 {reinterpret_cast<void**>(&s->dRec), sizeof(SpxFrameRec), 1, nullptr},
                 {reinterpret_cast<void**>(&s->dScr), sizeof(float), 4, nullptr},
                 {reinterpret_cast<void**>(&s->tTension), sizeof(float), 1, nullptr},
                 {reinterpret_cast<void**>(&s->tSpeed), sizeof(float), 1, nullptr},
                 {reinterpret_cast<void**>(&s->tFeatures), sizeof(float), SPX_FEATURE_COUNT, nullptr},
                 {reinterpret_cast<void**>(&s->tSpec), sizeof(float), P.N, nullptr},
                 {reinterpret_cast<void**>(&s->tNorm), sizeof(float), P.W, nullptr}};
  for (Arr& a : arrs) {
    if (hipMallocAsync(&a.np, (size_t)ncap * a.stride * a.elem, s->hs) != hipSuccess) {
      (void)hipGetLastError();
      for (Arr& b : arrs) if (b.np) (void)hipFreeAsync(b.np, s->hs);
      return false;   // nothing has moved: the stream is as it was
    }
  }
  for (Arr& a : arrs) {
    unsigned char* np = static_cast<unsigned char*>(a.np);
    unsigned char* op = static_cast<unsigned char*>(*a.p);
    const size_t row = (size_t)a.stride * a.elem;
    (void)hipMemsetAsync(np, 0, (size_t)ncap * row, s->hs);
    if (op && keep_hi > keep_lo) {
      const int64_t n = std::min(keep_hi, s->origin + s->cap) - keep_lo;
      if (n > 0)
        (void)hipMemcpyAsync(np + (size_t)(keep_lo - lo) * row, op + (size_t)(keep_lo - s->origin) * row, (size_t)n * row,
                             hipMemcpyDeviceToDevice, s->hs);
    }
    if (op) (void)hipFreeAsync(op, s->hs);
    *a.p = a.np;
  }
  s->origin = lo;
  s->cap = ncap;
  return true;
}

// Launch the tension kernel (and, for a new frame, the analysis kernel before it) for this stream.
static bool launch(speedyStream s, bool new_frame, int64_t k_from, int64_t k_to) {
  const SpxPlanDev& P = *s->plan;
  if (!grow(s, s->T + 2)) { spx_internal_set_api_error("speedy: device allocation failed"); return false; }
  SpxStreamDev J;
  memset(&J, 0, sizeof(J));
  J.frame_off = -s->origin;
  J.n_frames = (int32_t)(s->T + (new_frame ? 1 : 0));
  J.frame_begin = (int32_t)s->T;
  J.channels = 1;
  J.flags = (s->started ? 0 : SPX_F_INIT) | SPX_F_TENSION_RANGE | SPX_F_NO_SPEED;
  J.speed = 2.0f; J.nonlinear = 1.0f; J.feedback = 0.0f;   // unused by the passes that run
  J.unit_time0 = (s->time0 == 0) ? 1 : 0;
  J.tension_skip = (int32_t)k_from;
  J.tension_to = (int32_t)k_to;
  if (hipMemcpyAsync(s->dJob, &J, sizeof(J), hipMemcpyHostToDevice, s->hs) != hipSuccess) return false;
  const int64_t fo = J.frame_off;
  SpxTapsDev td;
  td.tension = s->tTension - s->origin - fo; td.speed = s->tSpeed - s->origin - fo;
  td.features = s->tFeatures - (s->origin + fo) * SPX_FEATURE_COUNT;
  td.spectrogram = s->tSpec - (s->origin + fo) * P.N; td.normalized = s->tNorm - (s->origin + fo) * P.W;
  // (origin + fo == 0: the tap bases are the allocations themselves; written out so the indexing rule stays visible)
  if (new_frame)
    spx_launch_analysis_frames(P, s->dJob, 1, s->dFrames - s->origin * P.W, true, s->dRec, td, s->hs);
  spx_launch_tension(P, s->dJob, 1, s->dState, s->dRec, s->dScr, td, nullptr, nullptr, s->hs);
  if (hipGetLastError() != hipSuccess) { spx_internal_set_api_error("speedy: kernel launch failed"); return false; }
  s->started = true;
  return true;
}


// =====================================================================================================================
// The reference's test hooks between stages (speedy.h:102-133, used by speedy_test.cc:135-453).  The stages are fused
// on the device, so each hook is a small kernel of its own on the SAME device state the fused kernels use -- the energy
// and difference filter states of the stream record, the hysteresis values in the per-frame scratch array (slot of
// time t = frame t - time0), the spectra in the tap array -- with the reference's arithmetic (types and order as in
// spx_analysis.hip / spx_tension.hip, which are checked against the oracle bit for bit).  One lane does the work: these
// are per-frame scalars and a few hundred bins.
// =====================================================================================================================
struct FirstOrderFilterStruct { float* d; int device; };  // device: [0] alpha, [1] state

__global__ void hk_fof_iterate_kernel(float* f, float input, float* out) {
  if (threadIdx.x != 0) return;
  const float alpha = f[0];
  const float y = (1 - alpha) * input + alpha * f[1];   // speedy.c:74, all float
  f[1] = y;
  *out = y;
}

// speedyNormalizeByEnergy (speedy.c:628-647): energy over bins 1.., every bin scaled
__global__ void hk_normalize_kernel(const float* s, float* out, int length, float* energy_out) {
  if (threadIdx.x != 0) return;
  float e = 0.0f;
  for (int i = 1; i < length; i++) e += s[i] * s[i];
  const float eps = 2.2204e-16f;
  const float inv = (float)(1.0 / (__builtin_sqrt((double)e) + (double)eps));
  for (int i = 0; i < length; i++) out[i] = s[i] * inv;
  *energy_out = e;
}

// speedyPreemphasisFilter (speedy.c:416-425): y[i] = 1.0*x[i] - 0.97*x[i-1] in double, float store; x[-1] = prev
__global__ void hk_preemph_kernel(float* x, int n, float prev) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < n; i++) {
    const float cur = x[i];
    x[i] = (float)(1.0 * (double)cur - 0.97 * (double)prev);
    prev = cur;
  }
}

// the hysteresis value of slot `tau` (time index): frame tau - time0 of the scratch array.  The reference keeps these in a
// ring of 2*(F+P+1) entries (speedy.c:95,617): a time not written yet reads what was written one ring length (or several)
// earlier -- which matters when a hook asks for a hysteresis before the F future frames exist (speedy_test.cc:443) --
// and zero before the first write.  `hi` = the highest time written so far.
__device__ inline float hk_slot(const SpxPlanDev& P, const float* scr, int64_t origin, int64_t cap, int64_t time0, int64_t hi,
                                int64_t tau) {
  const int64_t ring = 2 * (int64_t)(P.F + P.Pp + 1);
  // the ring slot of time tau holds the LATEST time written that shares it: a time not written yet reads what was written
  // one ring length (or several) earlier, a time overwritten since reads the newer value (speedy.c:617: index = time mod ring)
  if (tau > hi) tau -= ring * ((tau - hi + ring - 1) / ring);
  else tau += ring * ((hi - tau) / ring);
  const int64_t j = tau - time0;
  return (j >= origin && j < origin + cap && j >= 0) ? scr[4 * (j - origin) + 0] : 0.0f;
}
__device__ inline float hk_hysteresis(const SpxPlanDev& P, const float* scr, int64_t origin, int64_t cap, int64_t time0, int64_t hi,
                                      int64_t k) {
  float future_max = 0.0f, past_max = 0.0f;
  for (int i = 0; i <= P.F; i++) {   // speedy.c:594-608
    float v = hk_slot(P, scr, origin, cap, time0, hi, k + i);
    v *= P.taperF[i];
    if (v > future_max) future_max = v;
  }
  for (int i = 0; i <= P.Pp; i++) {
    float v = hk_slot(P, scr, origin, cap, time0, hi, k - i);
    v *= P.taperP[i];
    if (v > past_max) past_max = v;
  }
  return (float)((double)(past_max + future_max) / 2.0);   // speedy.c:609
}
__global__ void hk_hysteresis_kernel(SpxPlanDev P, const float* scr, int64_t origin, int64_t cap, int64_t time0, int64_t hi, int64_t k,
                                     float* out) {
  if (threadIdx.x == 0) *out = hk_hysteresis(P, scr, origin, cap, time0, hi, k);
}

// speedyComputeLocalEnergy (speedy.c:510-523) for the LAST frame's spectrum (the reference reads stream->spectrogram, not
// its argument, speedy.c:515): energy low-pass iterated once more, local energy, compression, hysteresis slot of `at_time`
__global__ void hk_local_energy_kernel(SpxPlanDev P, SpxStreamState* st, float energy, float* slot, float* feat) {
  if (threadIdx.x != 0) return;
  const float lp = P.one_minus_alpha * energy + P.alpha * st->lp;   // speedy.c:74
  st->lp = lp;
  const float local = energy / lp;
  const float comp = (float)__builtin_sqrt(local > 2 ? 2.0 : (double)local);
  *slot = comp;
  if (feat) { feat[1] = lp; feat[2] = local; feat[3] = comp; }
}

// speedyComputeSpectralDifference (speedy.c:664-729) for the two spectra given: features row k, normalised spectra
__global__ void hk_spectral_difference_kernel(SpxPlanDev P, SpxStreamState* st, const float* cur, const float* last,
                                              const float* scr, int64_t origin, int64_t cap, int64_t time0, int64_t hi, int64_t k,
                                              float* feat, float* norm_row, float* norm_last, int any_order, float* tension_out) {
  if (threadIdx.x != 0) return;
  const int W = P.W;   // = fft_size / 2
  const float hyst = hk_hysteresis(P, scr, origin, cap, time0, hi, k);
  const float eps = 2.2204e-16f;
  float e = 0.0f, mx = 0.0f, e2 = 0.0f;
  for (int i = 1; i < W; i++) { e += cur[i] * cur[i]; mx = fmaxf(mx, cur[i]); }
  for (int i = 1; i < W; i++) e2 += last[i] * last[i];
  const float inv = (float)(1.0 / (__builtin_sqrt((double)e) + (double)eps));     // speedy.c:642
  const float inv2 = (float)(1.0 / (__builtin_sqrt((double)e2) + (double)eps));
  for (int i = 0; i < W; i++) { norm_row[i] = cur[i] * inv; norm_last[i] = last[i] * inv2; }
  const float lowthr = (float)(0.04 * (double)1.41421f);               // speedy.c:682
  int first_k = st->tension_first;
  const bool first_call = first_k < 0;
  if (first_call) { first_k = (int)k; st->tension_first = first_k; }  // the very first call is skipped (speedy.c:293,691)
  // any_order: a time asked again (or out of order) -- only the FIRST call ever is the skipped one, whatever its time
  const bool low = e <= lowthr || (any_order ? first_call : (int)k == first_k);
  float lsd = 0.0f, ewld = 0.0f, rel = 0.0f, sc = 0.0f;
  if (!low) {
    const float thr = (float)((double)mx / 100.0);                     // speedy.c:709
    for (int i = 1; i < W; i++) {
      double term = 0.0;
      if (cur[i] > thr && last[i] > thr) {
        const float ratio = (norm_row[i] + eps) / (norm_last[i] + eps);
        term = __builtin_fabs(spx_log((double)ratio));                 // speedy.c:715-717
      }
      lsd = (float)((double)lsd + term);                               // float +=
    }
    ewld = lsd * hyst;                                                 // speedy.c:720
  }
  const float lpf = P.one_minus_alpha * ewld + P.alpha * st->lpf;      // the difference filter runs on skipped frames too (0)
  st->lpf = lpf;
  if (!low) {
    rel = (float)((double)ewld / ((double)lpf + 0.01 * (double)123.979f));         // speedy.c:725-726
    sc = (float)fmin((double)rel, (double)(4 * 0.971975f));                        // speedy.c:727-728
  }
  feat[0] = e; feat[4] = hyst; feat[5] = low ? 1.0f : 0.0f; feat[6] = lsd; feat[7] = ewld; feat[8] = lpf; feat[9] = rel;
  feat[10] = sc; feat[13] = (float)k; feat[14] = lowthr;
  if (tension_out) {
    const float a = 0.5f, b = 0.25f, M_E_ = 0.7f, M_S = 1.0f;
    const float tension = a * (hyst - M_E_) + b * (sc - M_S);          // speedy.c:761
    feat[11] = tension;
    *tension_out = tension;
  }
}

extern "C" {

speedyStream speedyCreateStream(int sample_rate) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    spx_internal_set_api_error("speedyCreateStream: no HIP device (this library has no CPU path)");
    return nullptr;
  }
  const SpxPlanDev* plan = spx_internal_shared_plan(sample_rate, spx_internal_match_matlab());
  if (!plan) { spx_internal_set_api_error("speedyCreateStream: plan creation failed"); return nullptr; }
  if (!spx_internal_analysis_fits(*plan)) {
    spx_internal_set_api_error("speedyCreateStream: sample rate too high (the analysis tile does not fit one CU's LDS)");
    return nullptr;
  }
  speedyStream s = new speedyStreamStruct();
  s->plan = plan;
  s->rate = sample_rate;
  (void)hipGetDevice(&s->device);
  const SpxPlanDev& P = *plan;
  if (hipStreamCreateWithFlags(&s->hs, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&s->dSmall), 1024) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&s->dScratchFrame), sizeof(float) * 2 * P.W) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&s->dScratchSpec), sizeof(float) * 2 * P.N) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&s->dHook), sizeof(float) * (64 + 6 * (size_t)P.N)) != hipSuccess) {
    spx_internal_set_api_error("speedyCreateStream: device allocation failed");
    speedyDestroyStream(s);
    return nullptr;
  }
  s->dJob = reinterpret_cast<SpxStreamDev*>(s->dSmall);
  s->dState = reinterpret_cast<SpxStreamState*>(s->dSmall + 256);
  s->dSpeed = reinterpret_cast<float*>(s->dSmall + 512);
  s->hSpec.assign(P.N, 0.0f); s->hSpecAt.assign(P.N, 0.0f); s->hNorm.assign(P.N, 0.0f);
  s->hFeat.assign(SPX_FEATURE_COUNT, 0.0f); s->hTmp.assign(2 * P.N, 0.0f);
  s->hAlFeat.assign(SPX_FEATURE_COUNT, 0.0f); s->hAlNorm.assign(P.N, 0.0f);
  return s;
}

void speedyDestroyStream(speedyStream s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->hs) (void)hipStreamSynchronize(s->hs);
  for (void* p : {(void*)s->dFrames, (void*)s->dRec, (void*)s->dScr, (void*)s->tTension, (void*)s->tSpeed,
                  (void*)s->tFeatures, (void*)s->tSpec, (void*)s->tNorm, (void*)s->dSmall, (void*)s->dScratchFrame,
                  (void*)s->dScratchSpec, (void*)s->dHook})
    if (p) (void)hipFree(p);
  if (s->hs) (void)hipStreamDestroy(s->hs);
  delete s;
}

int speedyInputFrameSize(speedyStream s) { return s->plan->W; }
int speedyInputFrameStep(speedyStream s) { return s->rate / 100; }                     // speedy.c:337
int speedyFFTSize(speedyStream s) { return s->plan->N; }
float speedyBinToFreq(speedyStream s, int bin) { return bin * (s->rate / (float)s->plan->N); }   // speedy.c:347
int speedyFreqToBin(speedyStream s, float freq) { return (int)round(freq * s->plan->N / s->rate); }  // speedy.c:352
int speedyHipHysteresisFuture(speedyStream s) { return s->plan->F; }
int speedyHipHysteresisPast(speedyStream s) { return s->plan->Pp; }
int64_t speedyGetCurrentTime(speedyStream s) { return s->current_time; }

void speedyAddData(speedyStream s, const float input[], int64_t at_time) {
  (void)hipSetDevice(s->device);
  const SpxPlanDev& P = *s->plan;
  if (s->time0 < 0) {
    if (at_time != 0 && at_time != 1) { spx_internal_set_api_error("speedyAddData: the first at_time must be 0 or 1"); return; }
    s->time0 = at_time;
  }
  if (at_time != s->time0 + s->T) { spx_internal_set_api_error("speedyAddData: at_time must advance by one per call"); return; }
  if (!grow(s, s->T + 2)) { spx_internal_set_api_error("speedy: device allocation failed"); return; }
  if (hipMemcpyAsync(s->dFrames + (s->T - s->origin) * P.W, input, sizeof(float) * P.W, hipMemcpyHostToDevice, s->hs) !=
      hipSuccess)
    return;
  (void)hipStreamSynchronize(s->hs);  // `input` is the caller's again
  if (!launch(s, true, s->tensionDone, s->tensionDone)) return;  // the new frame's analysis + its energy filter
  s->preemph_prev = input[P.W - 1];  // what speedyPreemphasisFilter's state is after the reference's AddData (speedy.c:545)
  s->T++;
  s->current_time = at_time;
}

void speedyAddDataShort(speedyStream s, const int16_t input[], int64_t at_time) {
  const int W = s->plan->W;
  std::vector<float> f((size_t)W);
  for (int i = 0; i < W; i++) f[i] = (float)(input[i] / 32768.0);  // speedy.c:558
  speedyAddData(s, f.data(), at_time);
}

// A time the reference's rings no longer hold as such -- 20 or more frames behind the newest (its spectrum ring has 21
// entries, speedy.c:97,476-487) -- or a time asked again / out of order: the reference answers from whatever its rings hold
// NOW (speedy.c:198-200,484-487,594-608: every index is the time modulo the ring length), i.e. from the latest frames that
// share the slots.  Its own tests do that (speedy_test.cc:564,628: `output_time = 0` after every success).  Same here: the
// two spectra are the rows of the latest frames in slots at_time and at_time - 1 (zeros when never written, speedy.c:242-248),
// the hysteresis reads the latest value of every slot (hk_slot), and the per-call state (difference filter, "the first call
// is the skipped one") advances in call order -- one lane of hk_spectral_difference_kernel, the arithmetic of the fused path.
static int tension_any_time(speedyStream s, int64_t at_time, float* tension) {
  const SpxPlanDev& P = *s->plan;
  if (!grow(s, s->T + 2)) { spx_internal_set_api_error("speedy: device allocation failed"); return 0; }
  if (!s->started && !launch(s, false, s->tensionDone, s->tensionDone)) return 0;   // (the state record exists after the first launch)
  const int64_t ring = P.F + P.Pp + 1, ct = s->current_time, t0 = s->time0 < 0 ? 0 : s->time0;
  float* zero = s->dHook + 64 + 4 * (size_t)P.N;        // [N] zeros
  float* feat = s->dHook + 8;                            // [15]
  float* nrow = s->dHook + 64;                           // [N] normalised row of this call
  float* nlast = s->dHook + 64 + 2 * (size_t)P.N;
  (void)hipMemsetAsync(zero, 0, sizeof(float) * P.N, s->hs);
  (void)hipMemsetAsync(feat, 0, sizeof(float) * 16, s->hs);
  auto row_of = [&](int64_t tau) -> const float* {
    // latest time <= ct that shares tau's slot
    int64_t m = ((tau % ring) + ring) % ring, c = ((ct % ring) + ring) % ring;
    int64_t latest = ct - ((c - m + ring) % ring);
    const int64_t j = latest - t0;
    if (j < 0 || j >= s->T || j < s->origin) return zero;   // never written (the kept window always covers one ring length)
    return s->tSpec + (j - s->origin) * P.N;
  };
  const float* cur = row_of(at_time);
  const float* last = row_of(at_time - 1);
  hipLaunchKernelGGL(hk_spectral_difference_kernel, dim3(1), dim3(64), 0, s->hs, P, s->dState, cur, last, s->dScr, s->origin, s->cap,
                     t0, ct, at_time, feat, nrow, nlast, 1, s->dHook);
  float v = 0.0f;
  if (hipMemcpyAsync(&v, s->dHook, sizeof(float), hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipMemcpyAsync(s->hAlFeat.data(), feat, sizeof(float) * SPX_FEATURE_COUNT, hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipMemcpyAsync(s->hAlNorm.data(), nrow, sizeof(float) * P.W, hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipStreamSynchronize(s->hs) != hipSuccess)
    return 0;
  // the values speedyComputeLocalEnergy left at the last speedyAddData (features 1-3, 12: speedy.c:517-522)
  if (s->T > 0 && s->T - 1 >= s->origin) {
    const int64_t k = s->T - 1 + (s->time0 == 0 ? 0 : 1) - P.F;
    if (k >= s->origin && k >= 0) {
      float f[SPX_FEATURE_COUNT];
      (void)hipMemcpy(f, s->tFeatures + (k - s->origin) * SPX_FEATURE_COUNT, sizeof(f), hipMemcpyDeviceToHost);
      s->hAlFeat[1] = f[1]; s->hAlFeat[2] = f[2]; s->hAlFeat[3] = f[3];
    }
    s->hAlFeat[12] = (float)ct;
  }
  s->aliased = true;
  *tension = v;
  return 1;
}

int speedyComputeTension(speedyStream s, int64_t at_time, float* tension) {
  (void)hipSetDevice(s->device);
  if (s->T == 0 || !(at_time + s->plan->F <= s->current_time)) return 0;  // speedy.c:756
  // in time order, and recent enough that the reference's rings still hold this time's own frames: the fused kernels
  if (at_time < s->tensionDone || s->current_time - at_time >= s->plan->F + s->plan->Pp) return tension_any_time(s, at_time, tension);
  if (!launch(s, false, at_time, at_time + 1)) return 0;
  float v = 0.0f;
  if (hipMemcpyAsync(&v, s->tTension + (at_time - s->origin), sizeof(float), hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipStreamSynchronize(s->hs) != hipSuccess)
    return 0;
  s->tensionDone = at_time + 1;
  s->aliased = false;
  *tension = v;
  return 1;
}

float speedyComputeSpeedFromTension(float tension, float R_g, float fb, speedyStream s) {
  (void)hipSetDevice(s->device);
  if (!s->started) {  // the state record is initialised by the first launch; before it, do that here
    SpxStreamState z;
    memset(&z, 0, sizeof(z));
    z.lp = 2.14204f; z.lpf = 123.837f;  // speedy.c:263-264
    z.tension_first = -1;
    (void)hipMemcpyAsync(s->dState, &z, sizeof(z), hipMemcpyHostToDevice, s->hs);
    (void)hipStreamSynchronize(s->hs);
    s->started = true;
  }
  spx_launch_speed_from_tension(s->dState, tension, R_g, fb, s->dSpeed, s->hs);
  float v = 0.0f;
  (void)hipMemcpyAsync(&v, s->dSpeed, sizeof(float), hipMemcpyDeviceToHost, s->hs);
  (void)hipStreamSynchronize(s->hs);
  return v;
}

float* speedySpectrogram(speedyStream s, float input[]) {
  (void)hipSetDevice(s->device);
  const SpxPlanDev& P = *s->plan;
  // a one-frame scratch "stream": frame 0 = input, no pre-emphasis, spectrum tap only
  SpxStreamDev J;
  memset(&J, 0, sizeof(J));
  J.n_frames = 1; J.frame_begin = 0; J.channels = 1; J.flags = SPX_F_INIT; J.unit_time0 = 1;
  SpxStreamDev* dJ = reinterpret_cast<SpxStreamDev*>(s->dSmall + 640);
  SpxFrameRec* dR = reinterpret_cast<SpxFrameRec*>(s->dSmall + 896);
  (void)hipMemcpyAsync(dJ, &J, sizeof(J), hipMemcpyHostToDevice, s->hs);
  (void)hipMemcpyAsync(s->dScratchFrame, input, sizeof(float) * P.W, hipMemcpyHostToDevice, s->hs);
  SpxTapsDev td = {nullptr, nullptr, nullptr, s->dScratchSpec, nullptr};
  spx_launch_analysis_frames(P, dJ, 1, s->dScratchFrame, false, dR, td, s->hs);
  (void)hipMemcpyAsync(s->hTmp.data(), s->dScratchSpec, sizeof(float) * P.N, hipMemcpyDeviceToHost, s->hs);
  (void)hipStreamSynchronize(s->hs);
  return s->hTmp.data();
}

static float* fetch_row(speedyStream s, const float* base, int64_t row, int n, std::vector<float>& h, int total) {
  std::fill(h.begin(), h.end(), 0.0f);
  if (row >= s->origin && row < s->origin + s->cap) {
    (void)hipSetDevice(s->device);
    (void)hipMemcpyAsync(h.data(), base + (row - s->origin) * (int64_t)n, sizeof(float) * n, hipMemcpyDeviceToHost, s->hs);
    (void)hipStreamSynchronize(s->hs);
  }
  (void)total;
  return h.data();
}
float* speedyGetSpectrogram(speedyStream s) { return fetch_row(s, s->tSpec, s->T - 1, s->plan->N, s->hSpec, s->plan->N); }
float* speedyGetSpectrogramAtTime(speedyStream s, int64_t at_time) {
  const int64_t j = at_time - (s->time0 < 0 ? 0 : s->time0);
  if (j < 0 || j >= s->T || j < s->T - 21) { std::fill(s->hSpecAt.begin(), s->hSpecAt.end(), 0.0f); return s->hSpecAt.data(); }
  return fetch_row(s, s->tSpec, j, s->plan->N, s->hSpecAt, s->plan->N);
}
float* speedyGetNormalizedSpectrogram(speedyStream s) {
  if (s->aliased) return s->hAlNorm.data();
  return fetch_row(s, s->tNorm, s->tensionDone - 1, s->plan->W, s->hNorm, s->plan->N);
}
float* speedyGetInternalState(speedyStream s) {
  if (s->aliased) return s->hAlFeat.data();
  return fetch_row(s, s->tFeatures, s->tensionDone - 1, SPX_FEATURE_COUNT, s->hFeat, SPX_FEATURE_COUNT);
}
float speedyGetEnergyCompressed(speedyStream s) {
  if (s->T == 0) return 0.0f;
  float v = 0.0f;
  (void)hipSetDevice(s->device);
  (void)hipMemcpyAsync(&v, s->dScr + (s->T - 1 - s->origin) * 4, sizeof(float), hipMemcpyDeviceToHost, s->hs);
  (void)hipStreamSynchronize(s->hs);
  return v;
}
float speedyGetSpeechChanges(speedyStream s) { return speedyGetInternalState(s)[10]; }

}  // extern "C"

extern "C" {

static float hook_scalar(speedyStream s) {   // the result a hook kernel left in dHook[0]
  float v = 0.0f;
  (void)hipMemcpyAsync(&v, s->dHook, sizeof(float), hipMemcpyDeviceToHost, s->hs);
  (void)hipStreamSynchronize(s->hs);
  return v;
}
static int64_t hook_time0(speedyStream s) { return s->time0 < 0 ? 0 : s->time0; }
static int64_t hook_hi(speedyStream s) { return hook_time0(s) + std::max(s->T, s->hookHi) - 1; }   // highest time written so far
// make row `j` of the per-frame arrays addressable for a hook (rows beyond the frames added so far included)
static bool hook_row(speedyStream s, int64_t j) {
  if (j < 0) return false;
  if (j + 1 > s->hookHi) s->hookHi = j + 1;
  return grow(s, std::max(s->T, j) + 2) && j >= s->origin;
}
static void hook_ensure_state(speedyStream s) {
  if (s->started) return;   // the state record is initialised by the first launch; before it, do that here
  SpxStreamState z;
  memset(&z, 0, sizeof(z));
  z.lp = 2.14204f; z.lpf = 123.837f;  // speedy.c:263-264
  z.tension_first = -1;
  (void)hipMemcpyAsync(s->dState, &z, sizeof(z), hipMemcpyHostToDevice, s->hs);
  (void)hipStreamSynchronize(s->hs);
  s->started = true;
}

FirstOrderFilter CreateFirstOrderFilter(float time_constant_in_samples) {          // speedy.c:50-60
  FirstOrderFilter f = new FirstOrderFilterStruct();
  (void)hipGetDevice(&f->device);
  if (hipMalloc(reinterpret_cast<void**>(&f->d), 4 * sizeof(float)) != hipSuccess) { delete f; return nullptr; }
  DesignFirstOrderLowpassFilter(f, time_constant_in_samples);
  return f;
}
void DesignFirstOrderLowpassFilter(FirstOrderFilter f, float tc) {                 // speedy.c:62-71 (a table, like the plan's)
  (void)hipSetDevice(f->device);
  const float h[2] = {tc > 0 ? (float)exp(-1.0 / tc) : 0.0f, 0.0f};   // alpha: double exp, float store; state cleared
  (void)hipMemcpy(f->d, h, sizeof(h), hipMemcpyHostToDevice);
}
float IterateFirstOrderFilter(FirstOrderFilter f, float input) {                   // speedy.c:73-76
  (void)hipSetDevice(f->device);
  hipLaunchKernelGGL(hk_fof_iterate_kernel, dim3(1), dim3(64), 0, nullptr, f->d, input, f->d + 2);
  float v = 0.0f;
  (void)hipMemcpy(&v, f->d + 2, sizeof(float), hipMemcpyDeviceToHost);
  return v;
}
void ResetFirstOrderFilter(FirstOrderFilter f) {                                   // speedy.c:78-81
  (void)hipSetDevice(f->device);
  (void)hipMemset(f->d + 1, 0, sizeof(float));
}
void DeleteFirstOrderFilter(FirstOrderFilter f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->d) (void)hipFree(f->d);
  delete f;
}

float speedyNormalizeByEnergy(const float* spectrogram, float* normalized, int length) {   // speedy.c:628-647
  if (length <= 0) return 0.0f;
  float* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(float) * (2 * (size_t)length + 1)) != hipSuccess) return 0.0f;
  (void)hipMemcpy(d, spectrogram, sizeof(float) * length, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(hk_normalize_kernel, dim3(1), dim3(64), 0, nullptr, d, d + length, length, d + 2 * length);
  float e = 0.0f;
  (void)hipMemcpy(normalized, d + length, sizeof(float) * length, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&e, d + 2 * length, sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e;
}

void speedyPreemphasisFilter(speedyStream s, float* input, int length) {           // speedy.c:416-425, in place
  if (length <= 0) return;
  (void)hipSetDevice(s->device);
  float* d = nullptr;
  if (hipMallocAsync(reinterpret_cast<void**>(&d), sizeof(float) * (size_t)length, s->hs) != hipSuccess) return;
  (void)hipMemcpyAsync(d, input, sizeof(float) * length, hipMemcpyHostToDevice, s->hs);
  const float last_raw = input[length - 1];
  hipLaunchKernelGGL(hk_preemph_kernel, dim3(1), dim3(64), 0, s->hs, d, length, s->preemph_prev);
  (void)hipMemcpyAsync(input, d, sizeof(float) * length, hipMemcpyDeviceToHost, s->hs);
  (void)hipFreeAsync(d, s->hs);
  // ONE pre-emphasis state, as in the reference (speedy.c:416-425): speedyAddData's analysis takes its carry from the last
  // sample of the previous frame's row, so that is where the hook leaves it (the row's own spectrum is long computed)
  if (s->T > 0 && s->dFrames && s->T - 1 >= s->origin)
    (void)hipMemcpyAsync(s->dFrames + (s->T - 1 - s->origin) * s->plan->W + (s->plan->W - 1), &last_raw, sizeof(float),
                         hipMemcpyHostToDevice, s->hs);
  (void)hipStreamSynchronize(s->hs);
  s->preemph_prev = last_raw;
}

void speedyAddToHysteresisBuffer(speedyStream s, float value, int64_t at_time) {   // speedy.c:615-619
  (void)hipSetDevice(s->device);
  const int64_t j = at_time - hook_time0(s);
  if (!hook_row(s, j)) { spx_internal_set_api_error("speedyAddToHysteresisBuffer: time outside the kept range"); return; }
  (void)hipMemcpyAsync(s->dScr + (j - s->origin) * 4, &value, sizeof(float), hipMemcpyHostToDevice, s->hs);
  (void)hipStreamSynchronize(s->hs);
}
float speedyEvaluateHysteresis(speedyStream s, int64_t at_time) {                  // speedy.c:590-610
  (void)hipSetDevice(s->device);
  if (!grow(s, s->T + 2)) return 0.0f;
  hipLaunchKernelGGL(hk_hysteresis_kernel, dim3(1), dim3(64), 0, s->hs, *s->plan, s->dScr, s->origin, s->cap, hook_time0(s),
                     hook_hi(s), at_time, s->dHook);
  return hook_scalar(s);
}

void speedySaveSpectrogramData(speedyStream s, float spectrogram[], int64_t at_time) {   // speedy.c:476-483
  (void)hipSetDevice(s->device);
  const int64_t j = at_time - hook_time0(s);
  if (!hook_row(s, j)) { spx_internal_set_api_error("speedySaveSpectrogramData: time outside the kept range"); return; }
  (void)hipMemcpyAsync(s->tSpec + (j - s->origin) * s->plan->N, spectrogram, sizeof(float) * s->plan->N, hipMemcpyHostToDevice, s->hs);
  (void)hipStreamSynchronize(s->hs);
}
float* speedyGetInternalSpectrogram(speedyStream s) { return speedyGetSpectrogram(s); }                      // speedy.c:393-396
float* speedyGetInternalNormalizedSpectrogram(speedyStream s) { return speedyGetNormalizedSpectrogram(s); }  // speedy.c:398-401

void speedyComputeLocalEnergy(speedyStream s, float* spectrogram, int64_t at_time) {   // speedy.c:510-523
  (void)spectrogram;   // the reference sums stream->spectrogram, the LAST frame's spectrum, whatever is passed (speedy.c:515)
  (void)hipSetDevice(s->device);
  const int64_t j = at_time - hook_time0(s);
  if (!hook_row(s, j)) { spx_internal_set_api_error("speedyComputeLocalEnergy: time outside the kept range"); return; }
  hook_ensure_state(s);
  float energy = 0.0f;   // of the last frame added (the analysis kernel's sum, float, bin order); nothing added yet: 0
  if (s->T > 0) {
    SpxFrameRec r;
    (void)hipMemcpyAsync(&r, s->dRec + (s->T - 1 - s->origin), sizeof(r), hipMemcpyDeviceToHost, s->hs);
    (void)hipStreamSynchronize(s->hs);
    energy = r.energy;
  }
  const int64_t k = j + (s->time0 == 0 ? 0 : 1) - s->plan->F;   // the feature row that shows this frame's AddData-time values
  float* feat = (k >= s->origin && k >= 0) ? s->tFeatures + (k - s->origin) * SPX_FEATURE_COUNT : nullptr;
  hipLaunchKernelGGL(hk_local_energy_kernel, dim3(1), dim3(64), 0, s->hs, *s->plan, s->dState, energy,
                     s->dScr + (j - s->origin) * 4, feat);
  (void)hipStreamSynchronize(s->hs);
}

void speedyComputeSpectralDifference(speedyStream s, const float* spectrogram, const float* last_spectrogram, int64_t at_time) {
  (void)hipSetDevice(s->device);
  const SpxPlanDev& P = *s->plan;
  const int64_t k = at_time;   // feature / normalised-spectrum rows are indexed by tension time
  if (k < 0 || !hook_row(s, k)) { spx_internal_set_api_error("speedyComputeSpectralDifference: time outside the kept range"); return; }
  hook_ensure_state(s);
  float* dc = s->dHook + 8;
  (void)hipMemcpyAsync(dc, spectrogram, sizeof(float) * P.W, hipMemcpyHostToDevice, s->hs);
  (void)hipMemcpyAsync(dc + P.N, last_spectrogram, sizeof(float) * P.W, hipMemcpyHostToDevice, s->hs);
  hipLaunchKernelGGL(hk_spectral_difference_kernel, dim3(1), dim3(64), 0, s->hs, P, s->dState, dc, dc + P.N, s->dScr, s->origin,
                     s->cap, hook_time0(s), hook_hi(s), k, s->tFeatures + (k - s->origin) * SPX_FEATURE_COUNT,
                     s->tNorm + (k - s->origin) * P.W, dc + 2 * P.N, 0, nullptr);
  (void)hipStreamSynchronize(s->hs);
  s->aliased = false;
  if (s->tensionDone < k + 1) s->tensionDone = k + 1;   // speedyGetInternalState / speedyGetSpeechChanges now show this row
}

}  // extern "C"
