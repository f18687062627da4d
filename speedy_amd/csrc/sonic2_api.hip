// The reference-compatible streaming API (include/sonic2.h) on top of the HIP kernels.
//
// Mirrors the reference shim, soniclib.c: same entry points, same units, same return conventions.  Where
// the shim keeps a ring of host buffers and calls the analysis and the TSM stage synchronously per 10 ms
// frame (soniclib.c:246-373), this implementation keeps the whole stream device-resident and, on every
// write, enqueues ONE analysis launch and ONE walk launch that cover all frames the new samples complete,
// resuming from the state record (SpxStreamState) the previous launch left in device memory.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/sonic2.h"
#include "spx_internal.h"

static thread_local std::string g_api_err;
static int g_match_matlab = 0;

template <class T>
struct DevBuf {  // growable device array, contents preserved on growth
  T* p = nullptr;
  size_t cap = 0;
  bool reserve(size_t n, size_t keep, hipStream_t st) {
    if (n <= cap) return true;
    size_t ncap = cap ? cap : 4096;
    while (ncap < n) ncap *= 2;
    T* np = nullptr;
    if (hipMalloc(&np, ncap * sizeof(T)) != hipSuccess) return false;
    if (p && keep) {
      if (hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
    }
    if (p) {
      (void)hipStreamSynchronize(st);
      (void)hipFree(p);
    }
    p = np;
    cap = ncap;
    return true;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct sonicStreamStruct {  // speedyConnectionStruct + the parts of libsonic's stream the API exposes
  const SpxPlanDev* plan = nullptr;
  int sampleRate = 0, channels = 0;
  float globalSpeed = 1.0f;         // soniclib.c:114
  bool speedupOnly = true;          // every launch so far had speed > 1 and 0 <= nonlinear factor <= 1
  float nonlinearFactor = 0.0f;     // soniclib.c:117
  float feedbackStrength = 0.1f;    // soniclib.c:122
  float rate = 1.0f;
  int bufferSize = 0;               // 0 until the first nonlinear write (soniclib.c:195, sonic_test.cc:496)
  int mode = -1;                    // -1 unknown, 0 linear, 1 nonlinear (fixed by the first write)
  bool flushed = false;
  tensionFunction cbTension = nullptr;
  speedFunction cbSpeed = nullptr;
  featuresFunction cbFeatures = nullptr;
  spectrogramFunction cbSpectrogram = nullptr, cbNormalized = nullptr;

  hipStream_t hs = nullptr;
  DevBuf<int16_t> dIn, dOut;
  DevBuf<SpxFrameRec> dRec;
  DevBuf<float> dScr;
  DevBuf<float> tTension, tSpeed, tFeatures, tSpec, tNorm;
  SpxStreamDev* dJob = nullptr;     // 1 entry
  SpxStreamState* dState = nullptr; // 1 entry
  int64_t* dNOut = nullptr;         // 1 entry

  int64_t nIn = 0;          // frames written so far
  int64_t framesDone = 0;   // analysis frames already launched
  int64_t outKnown = 0;     // frames produced, as of the last synchronisation
  int64_t outBound = 0;     // upper bound on frames produced by everything launched
  int64_t outRead = 0;      // frames already delivered to the caller
  bool dirty = false;       // launches in flight since the last synchronisation
  bool started = false;     // a job has been launched (state record valid)
  bool failed = false;
  std::vector<float> hostRow;  // callback scratch
  void* userData = nullptr;    // sonicIntSetUserData (soniclib.c:98,106)
};

static bool any_callback(sonicStream s) {
  return s->cbTension || s->cbSpeed || s->cbFeatures || s->cbSpectrogram || s->cbNormalized;
}

extern "C" {

const char* speedyHipLastError(void) { return g_api_err.c_str(); }
void speedyHipSetMatchMatlab(int on) { g_match_matlab = on ? 1 : 0; }

sonicStream sonicCreateStream(int sampleRate, int numChannels) {
  if (numChannels < 1) { g_api_err = "sonicCreateStream: numChannels < 1"; return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_api_err = "sonicCreateStream: no HIP device (this library has no CPU path)";
    return nullptr;
  }
  const SpxPlanDev* plan = spx_internal_shared_plan(sampleRate, g_match_matlab);
  if (!plan) { g_api_err = "sonicCreateStream: plan creation failed"; return nullptr; }
  sonicStream s = new sonicStreamStruct();
  s->plan = plan;
  s->sampleRate = sampleRate;
  s->channels = numChannels;
  if (hipStreamCreateWithFlags(&s->hs, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&s->dJob, sizeof(SpxStreamDev)) != hipSuccess ||
      hipMalloc(&s->dState, sizeof(SpxStreamState)) != hipSuccess ||
      hipMalloc(&s->dNOut, sizeof(int64_t)) != hipSuccess) {
    g_api_err = "sonicCreateStream: device allocation failed";
    sonicDestroyStream(s);
    return nullptr;
  }
  return s;
}

void sonicDestroyStream(sonicStream s) {
  if (!s) return;
  if (s->hs) (void)hipStreamSynchronize(s->hs);
  s->dIn.release(); s->dOut.release(); s->dRec.release(); s->dScr.release();
  s->tTension.release(); s->tSpeed.release(); s->tFeatures.release(); s->tSpec.release(); s->tNorm.release();
  if (s->dJob) (void)hipFree(s->dJob);
  if (s->dState) (void)hipFree(s->dState);
  if (s->dNOut) (void)hipFree(s->dNOut);
  if (s->hs) (void)hipStreamDestroy(s->hs);
  delete s;
}

void sonicSetRate(sonicStream s, float rate) { s->rate = rate; }
void sonicSetSpeed(sonicStream s, float speed) { s->globalSpeed = speed; }
void sonicEnableNonlinearSpeedup(sonicStream s, float f) { s->nonlinearFactor = f; }
void sonicSetDurationFeedbackStrength(sonicStream s, float f) { s->feedbackStrength = f; }
int getSonicBufferSize(sonicStream s) { return s ? s->bufferSize : 0; }
int sonicSpectrogramSize(sonicStream s) { return s ? s->plan->N : 0; }
int sonicIntGetNumChannels(sonicStream s) { return s->channels; }
int sonicIntGetSampleRate(sonicStream s) { return s->sampleRate; }
float sonicIntGetSpeed(sonicStream s) { return s->globalSpeed; }

void sonicTensionCallback(sonicStream s, tensionFunction f) { s->cbTension = f; }
tensionFunction getSonicTensionCallback(sonicStream s) { return s->cbTension; }
void sonicSpeedCallback(sonicStream s, speedFunction f) { s->cbSpeed = f; }
tensionFunction getSonicSpeedCallback(sonicStream s) { return (tensionFunction)s->cbSpeed; }
void sonicFeaturesCallback(sonicStream s, featuresFunction f) { s->cbFeatures = f; }
featuresFunction getSonicFeaturesCallback(sonicStream s) { return s->cbFeatures; }
void sonicSpectrogramCallback(sonicStream s, spectrogramFunction f) { s->cbSpectrogram = f; }
spectrogramFunction getSonicSpectrogramCallback(sonicStream s) { return s->cbSpectrogram; }
void sonicNormalizedSpectrogramCallback(sonicStream s, spectrogramFunction f) { s->cbNormalized = f; }
spectrogramFunction getSonicNormalizedSpectrogramCallback(sonicStream s) { return s->cbNormalized; }

}  // extern "C"

// Bring outKnown up to date (synchronises the stream).
static bool sync_stream(sonicStream s) {
  if (!s->dirty) return true;
  int64_t n = 0;
  if (hipMemcpyAsync(&n, s->dNOut, sizeof(n), hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipStreamSynchronize(s->hs) != hipSuccess) {
    g_api_err = std::string("stream synchronisation failed: ") + hipGetErrorString(hipGetLastError());
    s->failed = true;
    return false;
  }
  if (n == SPX_NOUT_LOST_PRODUCER) {
    g_api_err = "a producer kernel never delivered its frames (device-side poll limit reached)";
    s->failed = true;
    n = s->outKnown;
  } else if (n < 0) {
    g_api_err = "output capacity exceeded on the device";
    s->failed = true;
    n = -n;
  }
  s->outKnown = n;
  s->outBound = n;
  s->dirty = false;
  return true;
}

// Fire the monitoring callbacks for analysis calls [j0, j1) in the order of soniclib.c:297-353.
static void run_callbacks(sonicStream s, int64_t j0, int64_t j1) {
  const SpxPlanDev& P = *s->plan;
  const int N = P.N, W = P.W, F = P.F;
  s->hostRow.resize((size_t)N);
  for (int64_t j = j0; j < j1; j++) {
    const int at_time = (int)(j + 1);  // writeBufferFrameIndex at that moment
    if (s->cbSpectrogram) {
      (void)hipMemcpy(s->hostRow.data(), s->tSpec.p + (size_t)j * N, sizeof(float) * N, hipMemcpyDeviceToHost);
      s->cbSpectrogram(s, at_time, s->hostRow.data());
    }
    if (s->cbNormalized) {
      // the buffer the reference hands out here was filled by the PREVIOUS tension computation
      // (soniclib.c:303-310), i.e. tension frame j-F; before the first one it is uninitialised there, zero here
      const int64_t kprev = j - F;
      std::fill(s->hostRow.begin(), s->hostRow.end(), 0.0f);
      if (kprev >= 0)
        (void)hipMemcpy(s->hostRow.data(), s->tNorm.p + (size_t)kprev * W, sizeof(float) * W, hipMemcpyDeviceToHost);
      s->cbNormalized(s, at_time, s->hostRow.data());
    }
    const int64_t k = j - F + 1;
    if (k >= 0) {
      if (s->cbTension) {
        float v = 0;
        (void)hipMemcpy(&v, s->tTension.p + k, sizeof(float), hipMemcpyDeviceToHost);
        s->cbTension(s, (int)k, v);
      }
      if (s->cbFeatures) {
        float f[SPX_FEATURE_COUNT];
        (void)hipMemcpy(f, s->tFeatures.p + (size_t)k * SPX_FEATURE_COUNT, sizeof(f), hipMemcpyDeviceToHost);
        s->cbFeatures(s, (int)k, f);
      }
      if (s->cbSpeed) {
        float v = 0;
        (void)hipMemcpy(&v, s->tSpeed.p + k, sizeof(float), hipMemcpyDeviceToHost);
        s->cbSpeed(s, (int)k, v);
      }
    }
  }
}

// Enqueue the analysis + walk launches for everything written since the last job.
static int launch_job(sonicStream s, bool flush) {
  const SpxPlanDev& P = *s->plan;
  const bool nonlinear = s->mode == 1;
  const int64_t T = nonlinear ? spx_internal_frames_for(P, s->nIn) : 0;
  const int64_t fa = s->framesDone;
  const bool taps = nonlinear && any_callback(s);

  // output capacity: the bound of spx_internal_out_bound on the total produced since the stream start
  const int64_t bound = spx_internal_out_bound(P, s->nIn + 2 * (int64_t)P.maxRequired, s->globalSpeed, nonlinear);
  // `bound` limits the TOTAL output since the stream start, so it is the capacity to provide
  if (s->outBound < s->outKnown) s->outBound = s->outKnown;
  const int64_t need = bound > s->outBound ? bound : s->outBound;
  size_t keep = (size_t)s->outBound * s->channels;
  if (keep > s->dOut.cap) keep = s->dOut.cap;
  if (!s->dOut.reserve((size_t)need * s->channels, keep, s->hs)) return 0;
  if (nonlinear) {
    if (!s->dRec.reserve((size_t)T + 1, (size_t)fa, s->hs)) return 0;
    if (!s->dScr.reserve(4 * ((size_t)T + 1), 4 * (size_t)fa, s->hs)) return 0;
    if (taps) {
      if (!s->tTension.reserve((size_t)T + 1, (size_t)fa, s->hs) || !s->tSpeed.reserve((size_t)T + 1, (size_t)fa, s->hs) ||
          !s->tFeatures.reserve(((size_t)T + 1) * SPX_FEATURE_COUNT, (size_t)fa * SPX_FEATURE_COUNT, s->hs) ||
          !s->tSpec.reserve(((size_t)T + 1) * P.N, (size_t)fa * P.N, s->hs) ||
          !s->tNorm.reserve(((size_t)T + 1) * P.W, (size_t)fa * P.W, s->hs))
        return 0;
    }
  }
  SpxStreamDev J;
  memset(&J, 0, sizeof(J));
  J.in_off = 0; J.n_in = s->nIn; J.out_off = 0; J.out_cap = (int64_t)(s->dOut.cap / s->channels);
  J.frame_off = 0; J.n_frames = (int32_t)T; J.frame_begin = (int32_t)fa;
  J.channels = s->channels;
  J.flags = (s->started ? 0 : SPX_F_INIT) | (flush ? SPX_F_FLUSH : 0);
  J.speed = s->globalSpeed; J.nonlinear = nonlinear ? s->nonlinearFactor : 0.0f; J.feedback = s->feedbackStrength;
  // once a stream has run at a speed <= 1 its carried speed may be below 1: stay on the general kernel from then on
  if (!(J.speed > 1.0f && J.nonlinear >= 0.0f && J.nonlinear <= 1.0f)) s->speedupOnly = false;
  J.first_tile = 0;
  if (hipMemcpyAsync(s->dJob, &J, sizeof(J), hipMemcpyHostToDevice, s->hs) != hipSuccess) return 0;
  SpxTapsDev td = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (taps) {
    td.tension = s->tTension.p; td.speed = s->tSpeed.p; td.features = s->tFeatures.p;
    td.spectrogram = s->tSpec.p; td.normalized = s->tNorm.p;
  }
  if (nonlinear && T > fa) {
    const int TF = spx_analysis_tile_frames();
    const int tiles = (int)((T - fa + TF - 1) / TF);
    spx_launch_analysis(P, s->dJob, 1, tiles, s->dIn.p, s->dRec.p, td, nullptr, nullptr, s->hs);
  }
  if (nonlinear) spx_launch_tension(P, s->dJob, 1, s->dState, s->dRec.p, s->dScr.p, td, nullptr, nullptr, s->hs);
  spx_launch_walk(P, s->dJob, 1, s->channels, s->dIn.p, s->dOut.p, s->dNOut, s->dState, s->dScr.p, nullptr, s->speedupOnly,
                  s->hs);
  if (hipGetLastError() != hipSuccess) { g_api_err = "kernel launch failed"; s->failed = true; return 0; }
  s->started = true;
  s->dirty = true;
  s->outBound = need;
  s->framesDone = T;
  if (taps && T > fa) {
    (void)hipStreamSynchronize(s->hs);
    run_callbacks(s, fa, T);
  }
  return 1;
}

static int write_shorts(sonicStream s, const short* in, int sampleCount) {
  if (s->failed) return 0;
  if (s->rate != 1.0f) {
    g_api_err = "sonicSetRate != 1 is not supported (libsonic's resampler is outside the hot path)";
    return 0;
  }
  const int want = (s->nonlinearFactor != 0.0f) ? 1 : 0;  // soniclib.c:397
  if (s->mode < 0) s->mode = want;
  if (s->mode != want) {
    g_api_err = "switching between linear and nonlinear mode inside one stream is not supported";
    return 0;
  }
  if (s->flushed) {
    g_api_err = "writing after sonicFlushStream is not supported";
    return 0;
  }
  if (s->mode == 1 && s->bufferSize == 0) s->bufferSize = s->plan->B;  // sonicAllocateBuffers, soniclib.c:195
  if (!in || sampleCount <= 0) return 1;
  if (s->nIn + sampleCount >= (1ll << 30)) {
    g_api_err = "stream longer than 2^30 frames is not supported";
    return 0;
  }
  const size_t C = (size_t)s->channels;
  if (!s->dIn.reserve((size_t)(s->nIn + sampleCount) * C + 64, (size_t)s->nIn * C, s->hs)) return 0;
  if (hipMemcpyAsync(s->dIn.p + (size_t)s->nIn * C, in, sizeof(short) * (size_t)sampleCount * C,
                     hipMemcpyHostToDevice, s->hs) != hipSuccess)
    return 0;
  s->nIn += sampleCount;
  return launch_job(s, false);
}

extern "C" {

int sonicWriteShortToStream(sonicStream s, const short* in, int sampleCount) {
  return write_shorts(s, in, sampleCount);
}

int sonicWriteFloatToStream(sonicStream s, const float* in, int sampleCount) {
  if (!in || sampleCount <= 0) return write_shorts(s, nullptr, 0);
  const size_t n = (size_t)sampleCount * s->channels;
  std::vector<short> tmp(n);
  if (s->nonlinearFactor != 0.0f) {
    for (size_t i = 0; i < n; i++) tmp[i] = (short)(in[i] * 32768.0);   // soniclib.c:496
  } else {
    for (size_t i = 0; i < n; i++) tmp[i] = (short)(in[i] * 32767.0f);  // libsonic's float input scale
  }
  return write_shorts(s, tmp.data(), sampleCount);
}

int sonicSamplesAvailable(sonicStream s) {
  if (!sync_stream(s)) return 0;
  return (int)(s->outKnown - s->outRead);
}

int sonicReadShortFromStream(sonicStream s, short* out, int bufferSize) {
  if (!sync_stream(s)) return 0;
  int64_t n = s->outKnown - s->outRead;
  if (n <= 0) return 0;
  if (n > bufferSize) n = bufferSize;
  const size_t C = (size_t)s->channels;
  if (hipMemcpy(out, s->dOut.p + (size_t)s->outRead * C, sizeof(short) * (size_t)n * C, hipMemcpyDeviceToHost) !=
      hipSuccess)
    return 0;
  s->outRead += n;
  return (int)n;
}

int sonicReadFloatFromStream(sonicStream s, float* out, int bufferSize) {
  std::vector<short> tmp((size_t)(bufferSize > 0 ? bufferSize : 0) * s->channels);
  const int n = sonicReadShortFromStream(s, tmp.data(), bufferSize);
  const size_t cnt = (size_t)n * s->channels;
  for (size_t i = 0; i < cnt; i++) out[i] = tmp[i] / 32767.0f;  // libsonic's float output scale
  return n;
}

// ---- the libsonic entry points the reference's tests call directly (sonic_test.cc:370,735-750).  They address
// the TSM stage alone, which is what a stream in linear mode is here. ----
static bool linear_only(sonicStream s, const char* who) {
  if (s->mode == 1 || (s->mode < 0 && s->nonlinearFactor != 0.0f)) {
    g_api_err = std::string(who) + ": direct TSM-stage calls on a stream in nonlinear mode are not supported";
    return false;
  }
  return true;
}
sonicStream sonicIntCreateStream(int sampleRate, int numChannels) { return sonicCreateStream(sampleRate, numChannels); }
void sonicIntDestroyStream(sonicStream s) { sonicDestroyStream(s); }
void sonicIntSetSpeed(sonicStream s, float speed) { s->globalSpeed = speed; }
void sonicIntSetRate(sonicStream s, float rate) { s->rate = rate; }
int sonicIntWriteShortToStream(sonicStream s, const short* in, int n) {
  return linear_only(s, "sonicIntWriteShortToStream") ? write_shorts(s, in, n) : 0;
}
int sonicIntWriteFloatToStream(sonicStream s, const float* in, int n) {
  return linear_only(s, "sonicIntWriteFloatToStream") ? sonicWriteFloatToStream(s, in, n) : 0;
}
int sonicIntReadShortFromStream(sonicStream s, short* out, int n) { return sonicReadShortFromStream(s, out, n); }
int sonicIntReadFloatFromStream(sonicStream s, float* out, int n) { return sonicReadFloatFromStream(s, out, n); }
int sonicIntFlushStream(sonicStream s);
void sonicIntSetUserData(sonicStream s, void* p) { s->userData = p; }
void* sonicIntGetUserData(sonicStream s) { return s->userData; }

int sonicFlushStream(sonicStream s) {
  if (s->failed) return 0;
  if (s->mode < 0) s->mode = (s->nonlinearFactor != 0.0f) ? 1 : 0;
  if (s->flushed) return 1;
  const int rc = launch_job(s, true);
  s->flushed = true;
  return rc;
}
int sonicIntFlushStream(sonicStream s) { return linear_only(s, "sonicIntFlushStream") ? sonicFlushStream(s) : 0; }

}  // extern "C"
