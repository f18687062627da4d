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
  std::vector<float> hSpec, hSpecAt, hNorm, hFeat, hTmp;
};

static bool grow(speedyStream s, int64_t need_hi) {
  // make frames [max(0, T - SPD_KEEP), need_hi) addressable; everything slides together (one frame_off in the kernels)
  const SpxPlanDev& P = *s->plan;
  const int64_t lo = std::max<int64_t>(0, std::min(s->T, s->tensionDone) - SPD_KEEP - P.Pp - P.F);
  if (s->dFrames && need_hi <= s->origin + s->cap && lo - s->origin <= s->cap / 2) return true;
  const int64_t ncap = std::max<int64_t>(256, 2 * (need_hi - lo));
  const int64_t keep_lo = std::max(lo, s->origin), keep_hi = s->T + 1;  // rows that hold data (+1: normalised row T)
  auto move = [&](auto*& p, int64_t stride) -> bool {
    using E = std::remove_reference_t<decltype(*p)>;
    E* np = nullptr;
    if (hipMallocAsync(reinterpret_cast<void**>(&np), (size_t)ncap * stride * sizeof(E), s->hs) != hipSuccess) return false;
    (void)hipMemsetAsync(np, 0, (size_t)ncap * stride * sizeof(E), s->hs);
    if (p && keep_hi > keep_lo) {
      const int64_t n = std::min(keep_hi, s->origin + s->cap) - keep_lo;
      if (n > 0)
        (void)hipMemcpyAsync(np + (keep_lo - lo) * stride, p + (keep_lo - s->origin) * stride, (size_t)n * stride * sizeof(E),
                             hipMemcpyDeviceToDevice, s->hs);
    }
    if (p) (void)hipFreeAsync(p, s->hs);
    p = np;
    return true;
  };
  if (!move(s->dFrames, P.W) || !move(s->dRec, 1) || !move(s->dScr, 4) || !move(s->tTension, 1) || !move(s->tSpeed, 1) ||
      !move(s->tFeatures, SPX_FEATURE_COUNT) || !move(s->tSpec, P.N) || !move(s->tNorm, P.W))
    return false;
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
      hipMalloc(reinterpret_cast<void**>(&s->dScratchSpec), sizeof(float) * 2 * P.N) != hipSuccess) {
    spx_internal_set_api_error("speedyCreateStream: device allocation failed");
    speedyDestroyStream(s);
    return nullptr;
  }
  s->dJob = reinterpret_cast<SpxStreamDev*>(s->dSmall);
  s->dState = reinterpret_cast<SpxStreamState*>(s->dSmall + 256);
  s->dSpeed = reinterpret_cast<float*>(s->dSmall + 512);
  s->hSpec.assign(P.N, 0.0f); s->hSpecAt.assign(P.N, 0.0f); s->hNorm.assign(P.N, 0.0f);
  s->hFeat.assign(SPX_FEATURE_COUNT, 0.0f); s->hTmp.assign(2 * P.N, 0.0f);
  return s;
}

void speedyDestroyStream(speedyStream s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->hs) (void)hipStreamSynchronize(s->hs);
  for (void* p : {(void*)s->dFrames, (void*)s->dRec, (void*)s->dScr, (void*)s->tTension, (void*)s->tSpeed,
                  (void*)s->tFeatures, (void*)s->tSpec, (void*)s->tNorm, (void*)s->dSmall, (void*)s->dScratchFrame,
                  (void*)s->dScratchSpec})
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
  s->T++;
  s->current_time = at_time;
}

void speedyAddDataShort(speedyStream s, const int16_t input[], int64_t at_time) {
  const int W = s->plan->W;
  std::vector<float> f((size_t)W);
  for (int i = 0; i < W; i++) f[i] = (float)(input[i] / 32768.0);  // speedy.c:558
  speedyAddData(s, f.data(), at_time);
}

int speedyComputeTension(speedyStream s, int64_t at_time, float* tension) {
  (void)hipSetDevice(s->device);
  if (s->T == 0 || !(at_time + s->plan->F <= s->current_time)) return 0;  // speedy.c:756
  if (at_time < s->tensionDone) {
    spx_internal_set_api_error("speedyComputeTension: tensions must be asked for in increasing time order, each once");
    return 0;
  }
  if (!launch(s, false, at_time, at_time + 1)) return 0;
  float v = 0.0f;
  if (hipMemcpyAsync(&v, s->tTension + (at_time - s->origin), sizeof(float), hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipStreamSynchronize(s->hs) != hipSuccess)
    return 0;
  s->tensionDone = at_time + 1;
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
  return fetch_row(s, s->tNorm, s->tensionDone - 1, s->plan->W, s->hNorm, s->plan->N);
}
float* speedyGetInternalState(speedyStream s) {
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
