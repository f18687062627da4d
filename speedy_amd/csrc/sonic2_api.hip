// The reference-compatible streaming API (include/sonic2.h) on top of the HIP kernels.
//
// Mirrors the reference shim, soniclib.c: same entry points, same units, same return conventions.  Where the shim
// keeps a ring of host buffers and calls the analysis and the TSM stage synchronously per 10 ms frame
// (soniclib.c:246-373), this implementation keeps a sliding part of the stream device-resident and, on every write,
// enqueues ONE analysis, ONE tension and ONE walk launch that cover all frames the new samples complete, resuming from
// the state record (SpxStreamState) the previous launch left in device memory.
//
// Memory is bounded: the device buffers slide.  Input before the oldest frame either stage can still touch (the
// analysis halo, the TSM stage's buffered input), delivered output, and frame records behind the hysteresis look-back
// are dropped -- the buffers are indexed through negative base offsets (SpxStreamDev::in_off / out_off / frame_off), so
// the kernels keep using absolute stream coordinates.  The reference holds F+2 ring buffers plus libsonic's FIFOs;
// here a stream holds O(maxRequired + chunk) frames however long it runs (tests/test_gpu_sonic2.py soak test).
//
// Life cycle as in the reference: a stream stays usable after sonicFlushStream (soniclib.c:529-552 only moves the
// shim's read index to its write index and flushes the TSM stage), the nonlinear factor is re-read on every write
// (soniclib.c:397), sonicSetSpeed takes effect at once (soniclib.c:177-183), sonicSetRate is forwarded to the TSM stage
// (soniclib.c:169-175): from the first write with a rate != 1 on, the TSM stage's output passes through the rate stage
// (spx_rate.hip) into a second sliding buffer, which is then what the stream delivers.  One deviation:
// switching between factor == 0 and factor != 0 inside one stream fails at the next write with a message
// (speedyHipLastError); it never produces wrong audio.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <cmath>
#include <vector>

#include "../../include/sonic2.h"
#include "spx_internal.h"

static thread_local std::string g_api_err;
static int g_match_matlab = 0;

// A device array holding elements [origin, origin + cap) of a conceptually unbounded sequence.  ensure(lo, hi) makes
// [lo, hi) addressable and keeps what is already valid from lo on ([lo, filled)); it slides -- a stream-ordered copy
// into a fresh allocation, the old one freed in stream order -- when hi does not fit or when more than half the
// allocation is dead prefix.  base() is the pointer that, indexed with ABSOLUTE element numbers, lands in the allocation.
// `guard` elements in front of p[0] belong to the allocation too (zeroed, never meaningful): a reader that aligns its
// first position down may touch them.
template <class T>
struct SlideBuf {
  T* p = nullptr;
  int64_t origin = 0;  // absolute index of p[0]
  int64_t cap = 0;     // elements
  int64_t filled = 0;  // absolute end of valid data (set by the owner before ensure)
  int64_t guard = 0;   // addressable elements in front of p[0]
  // does [lo, hi) fit as things are (and is the dead prefix still small)?
  bool fits(int64_t lo, int64_t hi) const {
    return p && lo >= origin && hi <= origin + cap && lo - origin <= cap / 2;
  }
  // move the window so that it starts at lo and holds at least [lo, hi), keeping [lo, filled)
  bool slide_to(int64_t lo, int64_t hi, hipStream_t st, int64_t min_cap) {
    const int64_t ncap = std::max<int64_t>(min_cap, 2 * (hi - lo));
    T* np = nullptr;
    if (hipMallocAsync(reinterpret_cast<void**>(&np), (size_t)(ncap + guard) * sizeof(T), st) != hipSuccess) return false;
    if (guard) {
      (void)hipMemsetAsync(np, 0, (size_t)guard * sizeof(T), st);
      np += guard;
    }
    if (p && filled > lo && lo >= origin) {
      if (hipMemcpyAsync(np, p + (lo - origin), (size_t)(std::min(filled, origin + cap) - lo) * sizeof(T),
                         hipMemcpyDeviceToDevice, st) != hipSuccess)
        return false;
    }
    if (p) (void)hipFreeAsync(p - guard, st);
    p = np;
    origin = lo;
    cap = ncap;
    return true;
  }
  bool ensure(int64_t lo, int64_t hi, hipStream_t st, int64_t min_cap = 4096) {
    if (p && lo < origin) lo = origin;  // what was dropped stays dropped
    if (lo < 0) lo = 0;
    if (hi < lo) hi = lo;
    if (fits(lo, hi)) return true;
    return slide_to(lo, hi, st, min_cap);
  }
  T* base() const { return p - origin; }  // only ever dereferenced at indices >= origin
  void release(hipStream_t st) {
    if (p) (void)hipFreeAsync(p - guard, st);
    p = nullptr;
    cap = 0;
  }
};

struct sonicStreamStruct {  // speedyConnectionStruct + the parts of libsonic's stream the API exposes
  const SpxPlanDev* plan = nullptr;
  int device = 0;
  int sampleRate = 0, channels = 0;
  float globalSpeed = 1.0f;         // soniclib.c:114
  bool speedupOnly = true;          // every launch so far had speed > 1 and 0 <= nonlinear factor <= 1
  float nonlinearFactor = 0.0f;     // soniclib.c:117
  float feedbackStrength = 0.1f;    // soniclib.c:122
  float rate = 1.0f;
  int bufferSize = 0;               // 0 until the first nonlinear write (soniclib.c:195, sonic_test.cc:496)
  int mode = -1;                    // -1 unknown, 0 linear, 1 nonlinear (fixed by the first write)
  tensionFunction cbTension = nullptr;
  speedFunction cbSpeed = nullptr;
  featuresFunction cbFeatures = nullptr;
  spectrogramFunction cbSpectrogram = nullptr, cbNormalized = nullptr;

  hipStream_t hs = nullptr;
  SlideBuf<int16_t> dIn, dOut;      // elements = int16 values (frames * channels); dOut = what the TSM stage produces
  SlideBuf<int16_t> dFinal;         // rate mode: what the rate stage produces = what the stream delivers
  SpxRateState* dRate = nullptr;    // device record of the rate stage (directly behind dNOut)
  bool speedSet = false;            // sonicSetSpeed since the last job (SPX_F_SPEED_SET)
  bool rateMode = false;            // a write or flush has seen rate != 1: outputs go through the rate stage from then on
  int64_t finKnown = 0;             // rate mode: final frames produced / TSM frames taken, as of the last synchronisation
  int64_t finBound = 0;
  int64_t tsmSeenKnown = 0;
  SlideBuf<SpxFrameRec> dRec;       // elements = analysis frames
  SlideBuf<float> dScr;             // 4 floats per frame
  SlideBuf<float> tTension, tSpeed, tFeatures, tSpec, tNorm;
  unsigned char* dSmall = nullptr;  // SpxStreamDev job | SpxStreamState | int64 n_out, one allocation
  SpxStreamDev* dJob = nullptr;
  SpxStreamState* dState = nullptr;
  int64_t* dNOut = nullptr;
  unsigned char* hPinned = nullptr;  // pinned staging: job table (first 256 B), then input chunk / callback rows
  size_t hPinnedBytes = 0;
  hipEvent_t evStaged = nullptr;     // the last copy out of the staging area has been consumed

  int64_t nIn = 0;          // frames written so far
  int64_t framesDone = 0;   // analysis frames already launched
  int64_t tensionDone = 0;  // tension frames already computed (or skipped for good by a flush)
  int64_t tensionSkip = 0;  // SpxStreamDev::tension_skip
  int64_t tsmShift = 0;     // SpxStreamDev::tsm_shift
  int64_t outKnown = 0;     // frames produced, as of the last synchronisation
  int64_t outBound = 0;     // upper bound on frames produced by everything launched
  int64_t outRead = 0;      // frames already delivered to the caller
  int64_t tsmBase = 0;      // TSM stage's oldest buffered frame (TSM position), as of the last synchronisation
  int writesSinceSync = 0;
  bool dirty = false;       // launches in flight since the last synchronisation
  bool started = false;     // a job has been launched (state record valid)
  bool failed = false;
  void* userData = nullptr;    // sonicIntSetUserData (soniclib.c:98,106)
};

// shared with speedy_api.hip
void spx_internal_set_api_error(const std::string& msg) { g_api_err = msg; }
int spx_internal_match_matlab() { return g_match_matlab; }

static bool any_callback(sonicStream s) {
  return s->cbTension || s->cbSpeed || s->cbFeatures || s->cbSpectrogram || s->cbNormalized;
}

#define SPX_STAGE_JOB 256  // the job table lives in the first bytes of the staging area
// Staging area in pinned host memory: waits until the previous user's copies have left it, grows on demand (contents are
// not preserved).  Returns the part behind the job-table slot.
static unsigned char* staging(sonicStream s, size_t bytes) {
  if (s->evStaged) (void)hipEventSynchronize(s->evStaged);
  bytes += SPX_STAGE_JOB;
  if (bytes > s->hPinnedBytes) {
    if (s->hPinned) (void)hipHostFree(s->hPinned);
    s->hPinned = nullptr;
    size_t n = s->hPinnedBytes ? s->hPinnedBytes : 65536;
    while (n < bytes) n *= 2;
    if (hipHostMalloc(reinterpret_cast<void**>(&s->hPinned), n, hipHostMallocDefault) != hipSuccess) {
      s->hPinnedBytes = 0;
      g_api_err = "pinned staging allocation failed";
      return nullptr;
    }
    s->hPinnedBytes = n;
  }
  return s->hPinned + SPX_STAGE_JOB;
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
  if (spx_walk_lds_bytes(*plan, numChannels, false) > 160 * 1024) {  // one CU's LDS
    g_api_err = "sonicCreateStream: too many channels for the walk kernel's LDS window";
    return nullptr;
  }
  sonicStream s = new sonicStreamStruct();
  s->plan = plan;
  (void)hipGetDevice(&s->device);
  s->sampleRate = sampleRate;
  s->channels = numChannels;
  // the walk kernels refill their window from a position aligned down by 8 frames; right behind a flush that found
  // (almost) no input that lies up to 7 frames in front of the first input frame
  s->dIn.guard = 64 * (int64_t)numChannels;
  const size_t small = 256 + sizeof(SpxStreamState) + sizeof(int64_t) + sizeof(SpxRateState) + 64;
  if (hipStreamCreateWithFlags(&s->hs, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->evStaged, hipEventDisableTiming) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&s->dSmall), small) != hipSuccess) {
    g_api_err = "sonicCreateStream: device allocation failed";
    sonicDestroyStream(s);
    return nullptr;
  }
  s->dJob = reinterpret_cast<SpxStreamDev*>(s->dSmall);
  s->dState = reinterpret_cast<SpxStreamState*>(s->dSmall + 256);
  s->dNOut = reinterpret_cast<int64_t*>(s->dSmall + 256 + sizeof(SpxStreamState));  // directly behind the state
  s->dRate = reinterpret_cast<SpxRateState*>(s->dSmall + 256 + sizeof(SpxStreamState) + sizeof(int64_t));
  return s;
}

void sonicDestroyStream(sonicStream s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->hs) (void)hipStreamSynchronize(s->hs);
  s->dIn.release(s->hs); s->dOut.release(s->hs); s->dFinal.release(s->hs); s->dRec.release(s->hs); s->dScr.release(s->hs);
  s->tTension.release(s->hs); s->tSpeed.release(s->hs); s->tFeatures.release(s->hs); s->tSpec.release(s->hs);
  s->tNorm.release(s->hs);
  if (s->hs) (void)hipStreamSynchronize(s->hs);
  if (s->dSmall) (void)hipFree(s->dSmall);
  if (s->hPinned) (void)hipHostFree(s->hPinned);
  if (s->evStaged) (void)hipEventDestroy(s->evStaged);
  if (s->hs) (void)hipStreamDestroy(s->hs);
  delete s;
}

// sonicIntSetRate also restarts the two rate positions (the dependency does; a sample waiting in its pitch buffer stays)
void sonicSetRate(sonicStream s, float rate) {
  s->rate = rate;
  if (s->rateMode) {
    (void)hipSetDevice(s->device);
    (void)hipMemsetAsync(&s->dRate->old_pos, 0, 2 * sizeof(int32_t), s->hs);
  }
}
void sonicSetSpeed(sonicStream s, float speed) { s->globalSpeed = speed; s->speedSet = true; }
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

// Bring outKnown / tsmBase up to date: ONE device-to-host copy of {state record, produced count}, one synchronisation.
static bool sync_stream(sonicStream s) {
  if (!s->dirty) return true;
  (void)hipSetDevice(s->device);
  struct { SpxStreamState st; int64_t n; SpxRateState r; } h;
  static_assert(sizeof(h) == sizeof(SpxStreamState) + sizeof(int64_t) + sizeof(SpxRateState),
                "state, count and rate record are read back in one copy");
  const size_t hbytes = s->rateMode ? sizeof(h) : sizeof(SpxStreamState) + sizeof(int64_t);
  if (hipMemcpyAsync(&h, s->dState, hbytes, hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipStreamSynchronize(s->hs) != hipSuccess) {
    g_api_err = std::string("stream synchronisation failed: ") + hipGetErrorString(hipGetLastError());
    s->failed = true;
    return false;
  }
  int64_t n = h.n;
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
  if (s->rateMode) {
    if (h.r.overflow == 2 && !s->failed) { g_api_err = "a producer kernel never delivered its frames (device-side poll limit reached)"; s->failed = true; }
    else if (h.r.overflow && !s->failed) { g_api_err = "output capacity exceeded on the device (rate stage)"; s->failed = true; }
    s->finKnown = h.r.final_n;
    s->finBound = h.r.final_n;
    s->tsmSeenKnown = h.r.tsm_seen;
  }
  s->tsmBase = h.st.w.base;
  s->dirty = false;
  s->writesSinceSync = 0;
  return true;
}

// Fire the monitoring callbacks for analysis calls [j0, j1) in the order of soniclib.c:297-353; the tension frames this
// job computed are [k_first, j1 - F + 1).  All rows the calls need are fetched with one batch of asynchronous copies
// into pinned memory and one synchronisation.
static void run_callbacks(sonicStream s, int64_t j0, int64_t j1, int64_t k_first) {
  const SpxPlanDev& P = *s->plan;
  const int N = P.N, W = P.W, F = P.F;
  const int64_t nj = j1 - j0;
  if (nj <= 0) return;
  const int64_t k_end = (j1 >= F) ? j1 - F + 1 : 0;
  const int64_t nk = std::max<int64_t>(0, k_end - k_first);
  const int64_t kp0 = std::max<int64_t>(0, j0 - F), kp1 = std::max<int64_t>(kp0, j1 - F);  // rows of tNorm the calls hand out
  const size_t o_spec = 0;
  const size_t o_norm = o_spec + (s->cbSpectrogram ? sizeof(float) * (size_t)nj * N : 0);
  const size_t o_ten = o_norm + (s->cbNormalized ? sizeof(float) * (size_t)(kp1 - kp0) * W : 0);
  const size_t o_spd = o_ten + sizeof(float) * (size_t)nk;
  const size_t o_feat = o_spd + sizeof(float) * (size_t)nk;
  const size_t total = o_feat + sizeof(float) * (size_t)nk * SPX_FEATURE_COUNT + (size_t)N * sizeof(float);
  unsigned char* h = staging(s, total);
  if (!h) return;
  float* hSpec = reinterpret_cast<float*>(h + o_spec);
  float* hNorm = reinterpret_cast<float*>(h + o_norm);
  float* hTen = reinterpret_cast<float*>(h + o_ten);
  float* hSpd = reinterpret_cast<float*>(h + o_spd);
  float* hFeat = reinterpret_cast<float*>(h + o_feat);
  float* hRow = hFeat + (size_t)nk * SPX_FEATURE_COUNT;  // scratch row handed to the normalised-spectrum callback
  if (s->cbSpectrogram)
    (void)hipMemcpyAsync(hSpec, s->tSpec.base() + (size_t)j0 * N, sizeof(float) * (size_t)nj * N, hipMemcpyDeviceToHost, s->hs);
  if (s->cbNormalized && kp1 > kp0)
    (void)hipMemcpyAsync(hNorm, s->tNorm.base() + (size_t)kp0 * W, sizeof(float) * (size_t)(kp1 - kp0) * W,
                         hipMemcpyDeviceToHost, s->hs);
  if (nk > 0) {
    (void)hipMemcpyAsync(hTen, s->tTension.base() + k_first, sizeof(float) * (size_t)nk, hipMemcpyDeviceToHost, s->hs);
    (void)hipMemcpyAsync(hSpd, s->tSpeed.base() + k_first, sizeof(float) * (size_t)nk, hipMemcpyDeviceToHost, s->hs);
    (void)hipMemcpyAsync(hFeat, s->tFeatures.base() + (size_t)k_first * SPX_FEATURE_COUNT,
                         sizeof(float) * (size_t)nk * SPX_FEATURE_COUNT, hipMemcpyDeviceToHost, s->hs);
  }
  (void)hipStreamSynchronize(s->hs);
  for (int64_t j = j0; j < j1; j++) {
    const int at_time = (int)(j + 1);  // writeBufferFrameIndex at that moment
    if (s->cbSpectrogram) s->cbSpectrogram(s, at_time, hSpec + (size_t)(j - j0) * N);
    if (s->cbNormalized) {
      // the buffer the reference hands out here was filled by the PREVIOUS tension computation
      // (soniclib.c:303-310), i.e. tension frame j-F; before the first one it is uninitialised there, zero here
      const int64_t kprev = j - F;
      std::fill(hRow, hRow + N, 0.0f);
      if (kprev >= kp0 && kprev < kp1) memcpy(hRow, hNorm + (size_t)(kprev - kp0) * W, sizeof(float) * W);
      s->cbNormalized(s, at_time, hRow);
    }
    const int64_t k = j - F + 1;
    if (k >= k_first && k < k_end) {
      if (s->cbTension) s->cbTension(s, (int)k, hTen[k - k_first]);
      if (s->cbFeatures) s->cbFeatures(s, (int)k, hFeat + (size_t)(k - k_first) * SPX_FEATURE_COUNT);
      if (s->cbSpeed) s->cbSpeed(s, (int)k, hSpd[k - k_first]);
    }
  }
}

// Enqueue the analysis + tension + walk launches for everything written since the last job.  The caller has made sure
// the staging area exists and nobody reads its job-table slot any more.
static int launch_job(sonicStream s, bool flush) {
  const SpxPlanDev& P = *s->plan;
  const bool nonlinear = s->mode == 1;
  const int64_t C = s->channels;
  const int64_t T = nonlinear ? spx_internal_frames_for(P, s->nIn) : 0;
  const int64_t fa = s->framesDone;
  const bool taps = nonlinear && any_callback(s);
  const int F = P.F, Pp = P.Pp;

  // ---- output window [outRead, need): what was produced as of the last synchronisation plus the most the TSM stage
  // can make of the input it had not consumed by then (flush padding included) ----
  const int64_t unconsumed = s->nIn + s->tsmShift + 2 * (int64_t)P.maxRequired - s->tsmBase;
  const int64_t bound = s->outKnown + spx_internal_out_bound(P, unconsumed, s->globalSpeed, nonlinear);
  if (s->outBound < s->outKnown) s->outBound = s->outKnown;
  const int64_t need = std::max(bound, s->outBound);
  s->dOut.filled = s->outBound * C;
  // rate mode: the TSM output is dead once the rate stage has taken it (one frame stays as its left neighbour's source)
  const int64_t tsmKeep = s->rateMode ? std::max<int64_t>(0, s->tsmSeenKnown - 1) : s->outRead;
  if (!s->dOut.ensure(tsmKeep * C, need * C, s->hs, 1 << 16)) return 0;
  int oldR = s->sampleRate, newR = s->sampleRate;
  if (s->rateMode) {
    newR = (int)(s->sampleRate / s->rate);               // the dependency's adjustRate: both halved down to 14 bits
    while (newR > (1 << 14) || oldR > (1 << 14)) { newR >>= 1; oldR >>= 1; }
    if (newR < 1) { g_api_err = "sonicSetRate: rate too large for this sample rate"; return 0; }
    // final frames: what is known plus the most the rate stage can make of the TSM frames it has not taken yet
    const double per = (s->rate != 1.0f) ? (double)newR / (double)oldR : 1.0;
    const int64_t fneed = std::max(s->finBound, s->finKnown + (int64_t)((double)(need - s->tsmSeenKnown + 2) * per) + 16);
    s->dFinal.filled = s->finBound * C;
    if (!s->dFinal.ensure(s->outRead * C, fneed * C, s->hs, 1 << 16)) return 0;
    s->finBound = fneed;
  }
  // ---- frame records: the tension kernel looks back Pp + 1 frames of compressed energy and one record ----
  if (nonlinear) {
    // The kernels index all per-frame arrays (records, scratch, taps) through ONE frame_off, so they slide together:
    // all of them start at frame `keep`, or none moves.
    int64_t keep = std::max<int64_t>(0, std::min(fa, s->tensionDone) - Pp - F - 4);
    if (s->dRec.p && keep < s->dRec.origin) keep = s->dRec.origin;
    const int64_t hi = T + 2;
    struct Member { SlideBuf<float>* b; int64_t stride; int64_t filled_frames; };
    Member tapm[5] = {{&s->tTension, 1, fa}, {&s->tSpeed, 1, fa}, {&s->tFeatures, SPX_FEATURE_COUNT, fa},
                      {&s->tSpec, P.N, fa}, {&s->tNorm, P.W, fa + 1}};
    bool move = !s->dRec.fits(keep, hi) || !s->dScr.fits(4 * keep, 4 * hi);
    if (taps)
      for (auto& m : tapm) move = move || !m.b->fits(keep * m.stride, hi * m.stride);
    if (move) {
      s->dRec.filled = fa; s->dScr.filled = 4 * fa;
      if (!s->dRec.slide_to(keep, hi, s->hs, 4096) || !s->dScr.slide_to(4 * keep, 4 * hi, s->hs, 4 * 4096)) return 0;
      if (taps)
        for (auto& m : tapm) {
          m.b->filled = m.filled_frames * m.stride;
          if (!m.b->slide_to(keep * m.stride, hi * m.stride, s->hs, 4096 * m.stride)) return 0;
        }
    }
  }
  if (!s->dIn.p && !s->dIn.ensure(0, 64 * C, s->hs, 1 << 16)) return 0;  // a flush before any write: the kernels still
                                                                         // get a real (empty, guarded) input array
  // ---- the job: absolute stream coordinates through (possibly negative) base offsets ----
  SpxStreamDev& J = *reinterpret_cast<SpxStreamDev*>(s->hPinned);
  static_assert(sizeof(SpxStreamDev) <= SPX_STAGE_JOB, "job table slot");
  memset(&J, 0, sizeof(J));
  J.in_off = -s->dIn.origin; J.n_in = s->nIn;
  J.out_off = -s->dOut.origin; J.out_cap = (s->dOut.origin + s->dOut.cap) / C;
  J.frame_off = -s->dRec.origin; J.n_frames = (int32_t)T; J.frame_begin = (int32_t)fa;
  J.channels = (int32_t)C;
  J.flags = (s->started ? 0 : SPX_F_INIT) | (flush ? SPX_F_FLUSH : 0) | (s->rateMode ? SPX_F_NO_TRUNC : 0) |
            (s->speedSet ? SPX_F_SPEED_SET : 0);
  s->speedSet = false;
  J.speed = s->globalSpeed; J.nonlinear = nonlinear ? s->nonlinearFactor : 0.0f; J.feedback = s->feedbackStrength;
  J.tsm_shift = s->tsmShift;
  J.tension_skip = (int32_t)s->tensionSkip;
  // once a stream has run at a speed <= 1 its carried speed may be below 1: stay on the general kernel from then on
  if (!(J.speed > 1.0f && J.nonlinear >= 0.0f && J.nonlinear <= 1.0f)) s->speedupOnly = false;
  // SPX_F_NO_TRUNC is the general walk kernel's (the mono speed-up kernel is tuned to its register budget)
  const bool speedupKernel = s->speedupOnly && !s->rateMode;
  J.first_tile = 0;
  if (hipMemcpyAsync(s->dJob, &J, sizeof(J), hipMemcpyHostToDevice, s->hs) != hipSuccess) return 0;
  (void)hipEventRecord(s->evStaged, s->hs);
  SpxTapsDev td = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const int64_t fo = J.frame_off;
  if (taps) {  // tap rows are indexed frame_off + k: give the kernels bases that make that land in the sliding buffers
    td.tension = s->tTension.base() - fo; td.speed = s->tSpeed.base() - fo;
    td.features = s->tFeatures.base() - fo * SPX_FEATURE_COUNT;
    td.spectrogram = s->tSpec.base() - fo * P.N; td.normalized = s->tNorm.base() - fo * P.W;
  }
  if (nonlinear && T > fa) {
    const int TF = P.tile_frames;
    const int tiles = (int)((T - fa + TF - 1) / TF);
    spx_launch_analysis(P, s->dJob, 1, tiles, s->dIn.p, s->dRec.p, td, nullptr, nullptr, s->hs);
  }
  if (nonlinear) spx_launch_tension(P, s->dJob, 1, s->dState, s->dRec.p, s->dScr.p, td, nullptr, nullptr, s->hs);
  spx_launch_walk(P, s->dJob, 1, (int)C, s->dIn.p, s->dOut.p, s->dNOut, s->dState, s->dScr.p, nullptr, speedupKernel,
                  s->hs);
  if (s->rateMode)
    spx_launch_rate(s->dRate, s->dState, s->dNOut, s->dOut.base(), s->dFinal.base(),
                    (s->dFinal.origin + s->dFinal.cap) / C, (int)C, oldR, newR, s->rate, s->rate == 1.0f ? 1 : 0,
                    flush ? 1 : 0, s->hs);
  if (hipGetLastError() != hipSuccess) { g_api_err = "kernel launch failed"; s->failed = true; return 0; }
  s->started = true;
  s->dirty = true;
  s->outBound = need;
  const int64_t k_first = std::max(s->tensionDone, s->tensionSkip);
  s->framesDone = T;
  if (nonlinear) s->tensionDone = std::max<int64_t>(k_first, (T >= F) ? T - F + 1 : 0);
  if (flush) {
    // soniclib.c:538-550: every complete ring buffer goes to the TSM stage at the last speed and the shim's read index
    // moves to its write index -- tension frames below it that were not computed yet never will be; sonicIntFlushStream
    // then pads 2*maxRequired zeros, which later input follows in TSM coordinates
    if (nonlinear) {
      s->tensionSkip = std::max(s->tensionSkip, s->nIn / P.B);
      s->tensionDone = std::max(s->tensionDone, s->tensionSkip);
    }
    s->tsmShift += 2 * (int64_t)P.maxRequired;
  }
  if (taps && T > fa) run_callbacks(s, fa, T, k_first);
  return 1;
}

// First use of a rate != 1: from here on the stream delivers the rate stage's output.  What the TSM stage has produced
// so far (and the caller has not read yet) becomes the head of the final buffer; the rate stage starts behind it.
static bool enter_rate_mode(sonicStream s) {
  if (s->channels > SPX_RATE_MAX_CHANNELS) {
    g_api_err = "sonicSetRate != 1 supports at most 16 channels";
    return false;
  }
  if (!sync_stream(s)) return false;
  const int64_t C = s->channels;
  const int64_t n = s->outKnown - s->outRead;
  if (!s->dFinal.ensure(s->outRead * C, (s->outKnown + 4096) * C, s->hs, 1 << 16)) return false;
  if (n > 0 && hipMemcpyAsync(s->dFinal.base() + s->outRead * C, s->dOut.base() + s->outRead * C,
                              sizeof(int16_t) * (size_t)(n * C), hipMemcpyDeviceToDevice, s->hs) != hipSuccess)
    return false;
  SpxRateState* h = reinterpret_cast<SpxRateState*>(staging(s, sizeof(SpxRateState)));
  if (!h) return false;
  memset(h, 0, sizeof(*h));
  h->tsm_seen = s->outKnown;
  h->final_n = s->outKnown;
  if (hipMemcpyAsync(s->dRate, h, sizeof(*h), hipMemcpyHostToDevice, s->hs) != hipSuccess) return false;
  (void)hipEventRecord(s->evStaged, s->hs);
  s->finKnown = s->finBound = s->tsmSeenKnown = s->outKnown;
  s->rateMode = true;
  return true;
}

// The reference stores whatever float it is given; most values outside the documented ranges have no defined behaviour
// there (a speed <= 0 makes the TSM stage's step counts negative).  Here the next write / flush refuses them.
static bool settings_ok(sonicStream s) {
  if (!(s->globalSpeed > 0.0f) || !std::isfinite(s->globalSpeed)) { g_api_err = "sonicSetSpeed: speed must be finite and > 0"; return false; }
  if (!(s->nonlinearFactor >= 0.0f && s->nonlinearFactor <= 1.0f)) {
    g_api_err = "sonicEnableNonlinearSpeedup: factor outside [0, 1] (sonic2.h:73-76; the blended speed could reach 0)";
    return false;
  }
  if (!(s->rate > 0.0f) || !std::isfinite(s->rate)) { g_api_err = "sonicSetRate: rate must be finite and > 0"; return false; }
  if (!std::isfinite(s->feedbackStrength)) { g_api_err = "sonicSetDurationFeedbackStrength: not finite"; return false; }
  return true;
}

static int write_shorts(sonicStream s, const short* in, int sampleCount) {
  if (s->failed || !settings_ok(s)) return 0;
  (void)hipSetDevice(s->device);
  if (s->rate != 1.0f && !s->rateMode && !enter_rate_mode(s)) return 0;
  const int want = (s->nonlinearFactor != 0.0f) ? 1 : 0;  // soniclib.c:397
  if (s->mode < 0) s->mode = want;
  if (s->mode != want) {
    g_api_err = "switching between linear (factor 0) and nonlinear mode inside one stream is not supported";
    return 0;
  }
  if (s->mode == 1 && s->bufferSize == 0) s->bufferSize = s->plan->B;  // sonicAllocateBuffers, soniclib.c:195
  if (!in || sampleCount <= 0) return 1;
  if (s->nIn + sampleCount + s->tsmShift >= (1ll << 30)) {
    g_api_err = "stream longer than 2^30 frames is not supported";
    return 0;
  }
  const SpxPlanDev& P = *s->plan;
  const int64_t C = s->channels;
  // Without a read the host does not know how far the TSM stage has consumed its input: look every so often
  if (++s->writesSinceSync > 64 && !sync_stream(s)) return 0;
  // oldest input frame either stage can still touch: the analysis halo (frame framesDone-1 starts at (framesDone-1)*B)
  // and the TSM stage's buffered input (the window refill aligns down by 8 frames)
  int64_t keepFrom = s->tsmBase - s->tsmShift - 16;
  if (s->mode == 1) keepFrom = std::min(keepFrom, (s->framesDone - 1) * (int64_t)P.B - 16);
  if (s->dirty || !s->started) keepFrom = std::min(keepFrom, s->dIn.origin / C);  // unknown progress: keep what is there
  if (keepFrom < 0) keepFrom = 0;
  s->dIn.filled = s->nIn * C;
  if (!s->dIn.ensure(keepFrom * C, (s->nIn + sampleCount) * C + 64, s->hs, 1 << 16)) return 0;
  const size_t bytes = sizeof(short) * (size_t)sampleCount * C;
  unsigned char* h = staging(s, bytes);
  if (!h) return 0;
  memcpy(h, in, bytes);  // the caller's buffer is free again when this call returns
  if (hipMemcpyAsync(s->dIn.base() + s->nIn * C, h, bytes, hipMemcpyHostToDevice, s->hs) != hipSuccess) return 0;
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
  return write_shorts(s, tmp.data(), sampleCount);  // copied into the pinned staging area before this returns
}

int sonicSamplesAvailable(sonicStream s) {
  if (!sync_stream(s)) return 0;
  return (int)((s->rateMode ? s->finKnown : s->outKnown) - s->outRead);
}

int sonicReadShortFromStream(sonicStream s, short* out, int bufferSize) {
  if (!sync_stream(s)) return 0;
  int64_t n = (s->rateMode ? s->finKnown : s->outKnown) - s->outRead;
  if (n <= 0) return 0;
  if (n > bufferSize) n = bufferSize;
  const size_t C = (size_t)s->channels;
  (void)hipSetDevice(s->device);
  const int16_t* src = s->rateMode ? s->dFinal.base() : s->dOut.base();
  if (hipMemcpyAsync(out, src + (size_t)s->outRead * C, sizeof(short) * (size_t)n * C, hipMemcpyDeviceToHost,
                     s->hs) != hipSuccess ||
      hipStreamSynchronize(s->hs) != hipSuccess)
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
void sonicIntSetSpeed(sonicStream s, float speed) { sonicSetSpeed(s, speed); }
void sonicIntSetRate(sonicStream s, float rate) { sonicSetRate(s, rate); }
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
  if (s->failed || !settings_ok(s)) return 0;
  (void)hipSetDevice(s->device);
  if (s->mode < 0) s->mode = (s->nonlinearFactor != 0.0f) ? 1 : 0;
  if (s->rate != 1.0f && !s->rateMode && !enter_rate_mode(s)) return 0;
  if (!staging(s, 0)) return 0;  // the job-table slot
  return launch_job(s, true);    // the stream stays usable: a later write continues behind the flush's padding
}
int sonicIntFlushStream(sonicStream s) { return linear_only(s, "sonicIntFlushStream") ? sonicFlushStream(s) : 0; }

}  // extern "C"
