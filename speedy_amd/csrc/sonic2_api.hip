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
// (spx_rate.hip) into a second sliding buffer, which is then what the stream delivers.  Switching between factor == 0 and
// factor != 0 inside one stream works as in the reference ("mixed" streams below).
//
// Two execution paths.  EAGER (this file): the handle's own launch sequence per write, on its own HIP stream -- what a
// handle with monitoring callbacks, a rate stage, sonicInt* calls or a mode switch needs.  COALESCED (sonic2_pool.hip):
// every other handle; writes and flushes are only staged, and the first call that needs results on any waiting handle
// runs ONE launch sequence for all of them (the kernels take N-stream job tables).  Same kernels, same jobs
// (spx_prepare_job / spx_finish_job serve both), same results call for call.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <cmath>
#include <vector>

#include "../../include/sonic2.h"
#include "spx_internal.h"

#include <atomic>
static thread_local std::string g_api_err;
static std::atomic<int> g_match_matlab{0};

#include "sonic2_stream.h"

// shared with speedy_api.hip
void spx_internal_set_api_error(const std::string& msg) { g_api_err = msg; }
void spx_api_error(const std::string& msg) { g_api_err = msg; }
int spx_internal_match_matlab() { return g_match_matlab.load(); }

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

sonicStream speedyHipCreateSonicStreamEx(int sampleRate, int numChannels, int matchMatlab, int coalesce);
sonicStream sonicCreateStream(int sampleRate, int numChannels) {
  return speedyHipCreateSonicStreamEx(sampleRate, numChannels, g_match_matlab.load(), -1);
}
sonicStream speedyHipCreateSonicStream(int sampleRate, int numChannels, int matchMatlab) {
  return speedyHipCreateSonicStreamEx(sampleRate, numChannels, matchMatlab, -1);
}
// coalesce: -1 = the process-wide default (speedyHipSetCoalescing / SPX_NO_POOL), 0 = this handle runs its own launch
// sequence per write, 1 = this handle is coalesced whatever the default says.  The choice is the handle's: nothing
// process-wide is written here.
sonicStream speedyHipCreateSonicStreamEx(int sampleRate, int numChannels, int matchMatlab, int coalesce) {
  if (numChannels < 1) { g_api_err = "sonicCreateStream: numChannels < 1"; return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_api_err = "sonicCreateStream: no HIP device (this library has no CPU path)";
    return nullptr;
  }
  const SpxPlanDev* plan = spx_internal_shared_plan(sampleRate, matchMatlab ? 1 : 0);
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
  s->dTsm.guard = s->dIn.guard;
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
  if (SpxPool* pool = spx_pool_for_device(s->device, coalesce)) spx_pool_adopt(pool, s);
  return s;
}

void sonicDestroyStream(sonicStream s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->pooled) spx_pool_forget(s);   // off the waiting list; its buffers were only ever used on the pool's (idle) stream
  if (s->hs) (void)hipStreamSynchronize(s->hs);
  s->dIn.release(s->hs); s->dTsm.release(s->hs); s->dOut.release(s->hs); s->dFinal.release(s->hs); s->dRec.release(s->hs); s->dScr.release(s->hs);
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
// A setting changes what the NEXT job does; work a pooled handle has only staged so far was written under the old one.
// (Round 6: through the pool's lock whatever the handle's own flags say -- with several host threads the handle's staged write may
// be part of a run ANOTHER thread's read started a moment ago: no longer "pending", not yet done; spx_pool_sync waits for that run.)
static inline void settle(sonicStream s) { if (s->pooled) (void)spx_pool_sync(s); }
void sonicSetRate(sonicStream s, float rate) {
  settle(s);
  s->rate = rate;
  if (s->rateMode) {
    (void)hipSetDevice(s->device);
    (void)hipMemsetAsync(&s->dRate->old_pos, 0, 2 * sizeof(int32_t), s->hs);
  }
}
void sonicSetSpeed(sonicStream s, float speed) { settle(s); s->globalSpeed = speed; s->tsmSpeed = speed; s->speedSet = true; }
void sonicEnableNonlinearSpeedup(sonicStream s, float f) { settle(s); s->nonlinearFactor = f; }
void sonicSetDurationFeedbackStrength(sonicStream s, float f) { settle(s); s->feedbackStrength = f; }
int getSonicBufferSize(sonicStream s) { return s ? s->bufferSize : 0; }
int sonicSpectrogramSize(sonicStream s) { return s ? s->plan->N : 0; }
int sonicIntGetNumChannels(sonicStream s) { return s->channels; }
int sonicIntGetSampleRate(sonicStream s) { return s->sampleRate; }
float sonicIntGetSpeed(sonicStream s);  // below: the TSM stage's current speed lives on the device

void sonicTensionCallback(sonicStream s, tensionFunction f) { settle(s); s->cbTension = f; }
tensionFunction getSonicTensionCallback(sonicStream s) { return s->cbTension; }
void sonicSpeedCallback(sonicStream s, speedFunction f) { settle(s); s->cbSpeed = f; }
tensionFunction getSonicSpeedCallback(sonicStream s) { return (tensionFunction)s->cbSpeed; }
void sonicFeaturesCallback(sonicStream s, featuresFunction f) { settle(s); s->cbFeatures = f; }
featuresFunction getSonicFeaturesCallback(sonicStream s) { return s->cbFeatures; }
void sonicSpectrogramCallback(sonicStream s, spectrogramFunction f) { settle(s); s->cbSpectrogram = f; }
spectrogramFunction getSonicSpectrogramCallback(sonicStream s) { return s->cbSpectrogram; }
void sonicNormalizedSpectrogramCallback(sonicStream s, spectrogramFunction f) { settle(s); s->cbNormalized = f; }
spectrogramFunction getSonicNormalizedSpectrogramCallback(sonicStream s) { return s->cbNormalized; }

}  // extern "C"

// Bring outKnown / tsmBase up to date: ONE device-to-host copy of {state record, produced count}, one synchronisation.
static bool sync_stream(sonicStream s) {
  if (s->pooled) return spx_pool_sync(s);
  if (!s->dirty) return true;
  (void)hipSetDevice(s->device);
  struct { SpxStreamState st; int64_t n; SpxRateState r; } h;
  static_assert(sizeof(h) == sizeof(SpxStreamState) + sizeof(int64_t) + sizeof(SpxRateState),
                "state, count and rate record are read back in one copy");
  const size_t hbytes = s->rateMode ? sizeof(h) : sizeof(SpxStreamState) + sizeof(int64_t);
  if (hipMemcpyAsync(&h, s->dState, hbytes, hipMemcpyDeviceToHost, s->hs) != hipSuccess ||
      hipStreamSynchronize(s->hs) != hipSuccess) {
    spx_stream_fail(s, std::string("stream synchronisation failed: ") + hipGetErrorString(hipGetLastError()));
    return false;
  }
  int64_t n = h.n;
  if (n == SPX_NOUT_LOST_PRODUCER) {
    spx_stream_fail(s, "a producer kernel never delivered its frames (device-side poll limit reached)");
    n = s->outKnown;
  } else if (n < 0) {
    spx_stream_fail(s, "output capacity exceeded on the device");
    n = -n;
  }
  s->outKnown = n;
  s->outBound = n;
  if (s->rateMode) {
    if (h.r.overflow == 2 && !s->failed) spx_stream_fail(s, "a producer kernel never delivered its frames (device-side poll limit reached)");
    else if (h.r.overflow && !s->failed) spx_stream_fail(s, "output capacity exceeded on the device (rate stage)");
    s->finKnown = h.r.final_n;
    s->finBound = h.r.final_n;
    s->tsmSeenKnown = h.r.tsm_seen;
  }
  s->tsmBase = h.st.w.base;
  s->curSpeedKnown = h.st.curSpeed;
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

// What the next job of a stream is (everything written since the last one, plus a flush): makes room in the device
// buffers (stream-ordered on hs) and fills the two job records.  `pool`: the coalesced path -- the kernels of a pooled launch
// get NULL base pointers, so every offset is the absolute element address of the handle's own allocation (minus the
// sliding origin), and the frame records live in the pool's arena.
int spx_prepare_job(sonicStream s, bool flush, bool direct, hipStream_t hs, SpxPool* pool, SpxJobPlan& J, SpxDeferred* defer) {
  const SpxPlanDev& P = *s->plan;
  const int64_t C = s->channels;
  const int F = P.F, Pp = P.Pp, B = P.B;
  // what this job is: a stream has a ring (analysis) sequence once it has been written in nonlinear mode; a job runs the
  // analysis + tension kernels and the walk kernel's per-tension events when the stream is in nonlinear mode -- except
  // the flush of a mixed stream: no new tension can exist then, its pending ring buffers (appended to the TSM input below)
  // go one by one at the last speed, which the walk kernel's ring path does given a non-zero factor
  // `direct`: a sonicInt* call -- the TSM stage alone, the shim's ring is not touched (a direct flush hands nothing over)
  const bool hasRing = s->mode == 1 || s->mixed;
  const bool nonlinear = s->mode == 1 && !(s->mixed && flush) && !direct;
  const int64_t T = hasRing ? spx_internal_frames_for(P, s->nIn) : 0;
  const int64_t fa = s->framesDone;
  const bool taps = nonlinear && any_callback(s);
  J.hasRing = hasRing; J.nonlinear = nonlinear; J.taps = taps; J.flush = flush; J.direct = direct;
  J.T = T; J.fa = fa;

  // ---- ring buffers this job hands to the TSM stage (the kernel counts the same way: spx_walk.hip events) ----
  const int64_t handedBefore = s->handedHost;
  int64_t handedAfter = handedBefore;
  if (hasRing) {
    if (nonlinear) handedAfter = std::max<int64_t>(handedBefore, (T >= F) ? T - F + 1 : 0);  // one per tension frame, soniclib.c:354-369
    if (flush && !direct) handedAfter = std::max<int64_t>(handedAfter, s->nIn / B);         // every complete buffer, :538-550
  }
  J.handedAfter = handedAfter;
  if (s->mixed && handedAfter > handedBefore) {   // they join the TSM input behind whatever the linear writes put there
    const int64_t n = (handedAfter - handedBefore) * B;
    int64_t keepT = s->tsmBase - s->tsmShift - 16;
    if (s->dirty) keepT = std::min(keepT, s->dTsm.origin / C);
    if (keepT < 0) keepT = 0;
    s->dTsm.filled = s->tPhys * C;
    if (!s->dTsm.ensure(keepT * C, (s->tPhys + n) * C + 64, hs, 1 << 16)) return 0;
    if (hipMemcpyAsync(s->dTsm.base() + s->tPhys * C, s->dIn.base() + handedBefore * B * C, sizeof(int16_t) * (size_t)(n * C),
                       hipMemcpyDeviceToDevice, hs) != hipSuccess)
      return 0;
    s->tPhys += n;
  }
  const int64_t tsmLen = s->mixed ? s->tPhys : s->nIn;   // frames of TSM input that exist (flush paddings not counted)

  // ---- output window [outRead, need): what was produced as of the last synchronisation plus the most the TSM stage
  // can make of the input it had not consumed by then (flush padding included) ----
  const int64_t unconsumed = tsmLen + s->tsmShift + 2 * (int64_t)P.maxRequired - s->tsmBase;
  const int64_t bound = s->outKnown + spx_internal_out_bound(P, unconsumed, std::min(s->globalSpeed, s->tsmSpeed), hasRing);
  if (s->outBound < s->outKnown) s->outBound = s->outKnown;
  const int64_t need = std::max(bound, s->outBound);
  J.need = need;
  s->dOut.filled = s->outBound * C;
  // rate mode: the TSM output is dead once the rate stage has taken it (one frame stays as its left neighbour's source)
  const int64_t tsmKeep = s->rateMode ? std::max<int64_t>(0, s->tsmSeenKnown - 1) : s->outRead;
  if (!s->dOut.ensure(tsmKeep * C, need * C, hs, 1 << 16, defer)) return 0;
  int oldR = s->sampleRate, newR = s->sampleRate;
  if (s->rateMode) {
    newR = (int)(s->sampleRate / s->rate);               // the dependency's adjustRate: both halved down to 14 bits
    while (newR > (1 << 14) || oldR > (1 << 14)) { newR >>= 1; oldR >>= 1; }
    if (newR < 1) { g_api_err = "sonicSetRate: rate too large for this sample rate"; return 0; }
    // final frames: what is known plus the most the rate stage can make of the TSM frames it has not taken yet
    const double per = (s->rate != 1.0f) ? (double)newR / (double)oldR : 1.0;
    const int64_t fneed = std::max(s->finBound, s->finKnown + (int64_t)((double)(need - s->tsmSeenKnown + 2) * per) + 16);
    s->dFinal.filled = s->finBound * C;
    if (!s->dFinal.ensure(s->outRead * C, fneed * C, hs, 1 << 16)) return 0;
    s->finBound = fneed;
  }
  J.oldR = oldR; J.newR = newR;
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
    if (move && pool) {
      if (!spx_pool_slide_frames(pool, s, keep, hi, fa, defer)) return 0;
    } else if (move) {
      s->dRec.filled = fa; s->dScr.filled = 4 * fa;
      if (!s->dRec.slide_to(keep, hi, hs, 4096) || !s->dScr.slide_to(4 * keep, 4 * hi, hs, 4 * 4096)) return 0;
      if (taps)
        for (auto& m : tapm) {
          m.b->filled = m.filled_frames * m.stride;
          if (!m.b->slide_to(keep * m.stride, hi * m.stride, hs, 4096 * m.stride)) return 0;
        }
    }
  }
  SlideBuf<int16_t>& tsmIn = s->mixed ? s->dTsm : s->dIn;   // what the walk kernel reads
  J.tsmIn = &tsmIn;
  if (!s->dIn.p && !s->dIn.ensure(0, 64 * C, hs, 1 << 16, defer)) return 0;  // a flush before any write: the kernels still
  if (!tsmIn.p && !tsmIn.ensure(0, 64 * C, hs, 1 << 16, defer)) return 0;    // get real (empty, guarded) input arrays
  // ---- the jobs: absolute stream coordinates through (possibly negative) base offsets.  JA = what the analysis and
  // tension kernels see (the ring sequence), JW = what the walk kernel sees (the TSM input): the same record unless the
  // stream is mixed ----
  // element offset that makes `base + off + k` land on element k of the sliding buffer: base = the allocation (eager) or
  // NULL (pooled launches: the allocation's address counted in elements; hipMalloc'ed blocks are 256-byte aligned)
  auto off16 = [&](const SlideBuf<int16_t>& b) -> int64_t {
    return (pool ? (int64_t)(reinterpret_cast<uintptr_t>(b.p) / sizeof(int16_t)) : 0) - b.origin;
  };
  SpxStreamDev& JA = J.JA;
  SpxStreamDev& JW = J.JW;
  memset(&JA, 0, sizeof(JA));
  JA.in_off = off16(s->dIn); JA.n_in = s->nIn;
  JA.out_off = off16(s->dOut); JA.out_cap = (s->dOut.origin + s->dOut.cap) / C;
  JA.frame_off = -s->dRec.origin;
  if (pool && s->dRec.p) JA.frame_off += s->dRec.p - spx_pool_arena_rec(pool);
  JA.n_frames = (int32_t)T; JA.frame_begin = (int32_t)fa;
  JA.channels = (int32_t)C;
  const int common = (flush ? SPX_F_FLUSH : 0) | (s->rateMode ? SPX_F_NO_TRUNC : 0) | (s->speedSet ? SPX_F_SPEED_SET : 0);
  s->speedSet = false;
  JA.flags = common | (s->tensionStarted ? 0 : SPX_F_INIT);
  JA.speed = s->globalSpeed; JA.nonlinear = nonlinear ? s->nonlinearFactor : 0.0f; JA.feedback = s->feedbackStrength;
  JA.tsm_shift = s->tsmShift;
  JA.tension_skip = (int32_t)s->tensionSkip;
  JA.first_tile = 0;
  JW = JA;
  JW.speed = s->tsmSpeed;
  JW.flags = common | (s->started ? 0 : SPX_F_INIT);
  if (s->mixed) {
    JW.in_off = off16(s->dTsm); JW.n_in = s->tPhys;
    JW.flags |= SPX_F_HANDED_IN | (nonlinear ? 0 : SPX_F_KEEP_SPEED);
    JW.handed_in = (int32_t)handedBefore;
    JW.ring_bufs = (int32_t)(s->nIn / B);
    if (flush && !direct) JW.nonlinear = (s->nonlinearFactor != 0.0f) ? s->nonlinearFactor : 1.0f;  // the ring path; no tension event is pending
  }
  // once a stream has run at a speed <= 1 its carried speed may be below 1: stay on the general kernel from then on
  if (!(JA.speed > 1.0f && JW.speed > 1.0f && JA.speed < SPX_FAST_MAX_SPEED && JW.speed < SPX_FAST_MAX_SPEED && JA.nonlinear >= 0.0f &&
        JA.nonlinear <= 1.0f))
    s->speedupOnly = false;
  // SPX_F_NO_TRUNC and the mixed-stream flags are the general walk kernel's (the speed-up kernels are tuned to their
  // register budget)
  J.speedupKernel = s->speedupOnly && !s->rateMode && !s->mixed;
  J.tiles = (nonlinear && T > fa) ? (int)((T - fa + P.tile_frames - 1) / P.tile_frames) : 0;
  return 1;
}

// The host's mirror of what the job just enqueued will have done to the stream.
void spx_finish_job(sonicStream s, const SpxJobPlan& J) {
  const SpxPlanDev& P = *s->plan;
  const int F = P.F;
  s->started = true;
  s->dirty = true;
  s->outBound = J.need;
  s->handedHost = J.handedAfter;
  if (J.nonlinear) s->tensionStarted = true;
  const int64_t k_first = std::max(s->tensionDone, s->tensionSkip);
  if (J.hasRing) s->framesDone = std::max(s->framesDone, J.nonlinear ? J.T : J.fa);
  if (J.nonlinear) s->tensionDone = std::max<int64_t>(k_first, (J.T >= F) ? J.T - F + 1 : 0);
  if (J.flush) {
    // soniclib.c:538-550: every complete ring buffer goes to the TSM stage at the last speed and the shim's read index
    // moves to its write index -- tension frames below it that were not computed yet never will be; sonicIntFlushStream
    // then pads 2*maxRequired zeros, which later input follows in TSM coordinates
    if (J.hasRing && !J.direct) {
      s->tensionSkip = std::max(s->tensionSkip, s->nIn / P.B);
      s->tensionDone = std::max(s->tensionDone, s->tensionSkip);
    }
    s->tsmShift += 2 * (int64_t)P.maxRequired;
  }
}

// Enqueue the analysis + tension + walk launches for everything written since the last job (the eager path: this
// handle's own launch sequence on its own HIP stream).  The caller has made sure the staging area exists and nobody
// reads its job-table slot any more.
static int launch_job(sonicStream s, bool flush, bool direct = false) {
  const SpxPlanDev& P = *s->plan;
  const int64_t C = s->channels;
  SpxJobPlan J;
  if (!spx_prepare_job(s, flush, direct, s->hs, nullptr, J)) return 0;
  static_assert(2 * 128 <= SPX_STAGE_JOB && sizeof(SpxStreamDev) <= 128, "job table slot");
  memset(s->hPinned, 0, 256);
  memcpy(s->hPinned, &J.JA, sizeof(SpxStreamDev));
  memcpy(s->hPinned + 128, &J.JW, sizeof(SpxStreamDev));
  SpxStreamDev* dJobW = reinterpret_cast<SpxStreamDev*>(reinterpret_cast<unsigned char*>(s->dJob) + 128);
  if (hipMemcpyAsync(s->dJob, s->hPinned, 256, hipMemcpyHostToDevice, s->hs) != hipSuccess) return 0;
  (void)hipEventRecord(s->evStaged, s->hs);
  SpxTapsDev td = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const int64_t fo = J.JA.frame_off;
  if (J.taps) {  // tap rows are indexed frame_off + k: give the kernels bases that make that land in the sliding buffers
    td.tension = s->tTension.base() - fo; td.speed = s->tSpeed.base() - fo;
    td.features = s->tFeatures.base() - fo * SPX_FEATURE_COUNT;
    td.spectrogram = s->tSpec.base() - fo * P.N; td.normalized = s->tNorm.base() - fo * P.W;
  }
  if (J.tiles > 0) spx_launch_analysis(P, s->dJob, 1, J.tiles, s->dIn.p, s->dRec.p, td, nullptr, nullptr, s->hs);
  if (J.nonlinear) spx_launch_tension(P, s->dJob, 1, s->dState, s->dRec.p, s->dScr.p, td, nullptr, nullptr, s->hs);
  spx_launch_walk(P, dJobW, 1, (int)C, J.tsmIn->p, s->dOut.p, s->dNOut, s->dState, s->dScr.p, nullptr, J.speedupKernel, s->hs,
                  /*short_jobs: one write, a few pitch steps -- the usual window, not the long one*/ true);
  if (s->rateMode)
    spx_launch_rate(s->dRate, s->dState, s->dNOut, s->dOut.base(), s->dFinal.base(),
                    (s->dFinal.origin + s->dFinal.cap) / C, (int)C, J.oldR, J.newR, s->rate, s->rate == 1.0f ? 1 : 0,
                    flush ? 1 : 0, s->hs);
  if (hipGetLastError() != hipSuccess) { spx_stream_fail(s, "kernel launch failed"); return 0; }
  const int64_t k_first = std::max(s->tensionDone, s->tensionSkip);
  spx_finish_job(s, J);
  if (J.taps && J.T > J.fa) run_callbacks(s, J.fa, J.T, k_first);
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

// The first write in the other mode (soniclib.c:397-399 looks at the factor on every write): from here on the TSM stage's
// input is its own sequence.  Coming from linear mode it is everything written so far and the ring sequence starts
// empty; coming from nonlinear mode it is the ring buffers handed over so far (the stage may still hold the last of
// them), while the ring sequence keeps its unhanded buffers and its partial one.
static bool enter_mixed(sonicStream s) {
  if (!sync_stream(s)) return false;
  const int64_t C = s->channels;
  if (s->mode == 0) {
    std::swap(s->dTsm, s->dIn);          // (the guards are equal)
    s->tPhys = s->nIn;
    s->nIn = 0;
  } else {
    const int64_t hi = s->handedHost * (int64_t)s->plan->B;
    int64_t lo = std::max<int64_t>(0, s->tsmBase - s->tsmShift - 16);
    lo = std::min(std::max(lo, s->dIn.origin / C), hi);
    s->dTsm.filled = 0;
    if (!s->dTsm.ensure(lo * C, hi * C + 64, s->hs, 1 << 16)) return false;
    if (hi > lo && hipMemcpyAsync(s->dTsm.base() + lo * C, s->dIn.base() + lo * C, sizeof(int16_t) * (size_t)((hi - lo) * C),
                                  hipMemcpyDeviceToDevice, s->hs) != hipSuccess)
      return false;
    s->tPhys = hi;
  }
  s->mixed = true;
  return true;
}

// The reference stores whatever float it is given; most values outside the documented ranges have no defined behaviour
// there (a speed <= 0 makes the TSM stage's step counts negative).  Here the next write / flush refuses them.
bool spx_settings_ok(sonicStream s) {
  if (!(s->globalSpeed > 0.0f) || !std::isfinite(s->globalSpeed) || !(s->tsmSpeed > 0.0f) || !std::isfinite(s->tsmSpeed)) {
    g_api_err = "sonicSetSpeed: speed must be finite and > 0";
    return false;
  }
  if (!(s->nonlinearFactor >= 0.0f && s->nonlinearFactor <= 1.0f)) {
    g_api_err = "sonicEnableNonlinearSpeedup: factor outside [0, 1] (sonic2.h:73-76; the blended speed could reach 0)";
    return false;
  }
  if (!(s->rate > 0.0f) || !std::isfinite(s->rate)) { g_api_err = "sonicSetRate: rate must be finite and > 0"; return false; }
  if (!std::isfinite(s->feedbackStrength)) { g_api_err = "sonicSetDurationFeedbackStrength: not finite"; return false; }
  return true;
}

static int write_shorts(sonicStream s, const short* in, int sampleCount, bool direct = false) {
  if (spx_stream_failed(s) || !spx_settings_ok(s)) return 0;
  const int want = (s->nonlinearFactor != 0.0f && !direct) ? 1 : 0;  // soniclib.c:397: decided anew on every write; sonicInt* bypasses
  // the coalesced path serves plain streams; anything that needs a launch sequence of its own leaves it for good
  if (s->pooled && (direct || s->rate != 1.0f || any_callback(s) || (s->mode >= 0 && s->mode != want)) && !spx_pool_leave(s))
    return 0;
  if (!s->pooled) (void)hipSetDevice(s->device);   // (a staged write makes no runtime call at all; the pool sets the device when it runs)
  if (s->rate != 1.0f && !s->rateMode && !enter_rate_mode(s)) return 0;
  if (want == 1 && !spx_internal_analysis_fits(*s->plan)) {
    g_api_err = "sample rate too high for the nonlinear path (the analysis tile does not fit one CU's LDS); linear mode only";
    return 0;
  }
  if (s->mode < 0) s->mode = want;
  if (want == 1 && s->bufferSize == 0) s->bufferSize = s->plan->B;  // sonicAllocateBuffers, soniclib.c:195
  if (!in || sampleCount <= 0) return 1;
  if (s->pooled) return spx_pool_write(s, in, sampleCount);
  if (s->mode != want) {  // the other mode from now on: the ring sequence and the TSM input part ways
    if (!s->mixed && !enter_mixed(s)) return 0;
    s->mode = want;
  }
  const SpxPlanDev& P = *s->plan;
  const int64_t C = s->channels;
  const bool toTsm = s->mixed && want == 0;   // a linear write of a mixed stream goes straight to the TSM input
  if ((toTsm ? s->tPhys : s->nIn) + sampleCount + s->tsmShift >= (1ll << 30) ||
      (s->mixed && s->tPhys + s->nIn + sampleCount + s->tsmShift >= (1ll << 30))) {
    g_api_err = "stream longer than 2^30 frames is not supported";
    return 0;
  }
  // Without a read the host does not know how far the TSM stage has consumed its input: look every so often
  if (++s->writesSinceSync > 64 && !sync_stream(s)) return 0;
  SlideBuf<int16_t>& dst = toTsm ? s->dTsm : s->dIn;
  int64_t& len = toTsm ? s->tPhys : s->nIn;
  // oldest frame still needed.  TSM input: what the stage has buffered (the window refill aligns down by 8 frames).
  // Ring sequence: the analysis halo (frame framesDone-1 starts at (framesDone-1)*B) and, on a mixed stream, the
  // buffers not handed over yet; on an unmixed stream it is the TSM input as well.
  int64_t keepFrom = s->tsmBase - s->tsmShift - 16;
  if (!toTsm) {
    if (s->mixed) keepFrom = s->handedHost * (int64_t)P.B;
    if (s->mode == 1) keepFrom = std::min(keepFrom, (s->framesDone - 1) * (int64_t)P.B - 16);
  }
  if (s->dirty || !s->started) keepFrom = std::min(keepFrom, dst.origin / C);  // unknown progress: keep what is there
  if (keepFrom < 0) keepFrom = 0;
  dst.filled = len * C;
  if (!dst.ensure(keepFrom * C, (len + sampleCount) * C + 64, s->hs, 1 << 16)) return 0;
  const size_t bytes = sizeof(short) * (size_t)sampleCount * C;
  unsigned char* h = staging(s, bytes);
  if (!h) return 0;
  memcpy(h, in, bytes);  // the caller's buffer is free again when this call returns
  if (hipMemcpyAsync(dst.base() + len * C, h, bytes, hipMemcpyHostToDevice, s->hs) != hipSuccess) return 0;
  len += sampleCount;
  return launch_job(s, false, direct);
}

extern "C" {

int sonicWriteShortToStream(sonicStream s, const short* in, int sampleCount) {
  return write_shorts(s, in, sampleCount);
}

static int write_floats(sonicStream s, const float* in, int sampleCount, bool direct) {
  if (!in || sampleCount <= 0) return write_shorts(s, nullptr, 0, direct);
  const size_t n = (size_t)sampleCount * s->channels;
  std::vector<short> tmp(n);
  if (s->nonlinearFactor != 0.0f && !direct) {
    for (size_t i = 0; i < n; i++) tmp[i] = (short)(in[i] * 32768.0);   // soniclib.c:496
  } else {
    for (size_t i = 0; i < n; i++) tmp[i] = (short)(in[i] * 32767.0f);  // libsonic's float input scale
  }
  return write_shorts(s, tmp.data(), sampleCount, direct);  // copied into the pinned staging area before this returns
}
int sonicWriteFloatToStream(sonicStream s, const float* in, int sampleCount) { return write_floats(s, in, sampleCount, false); }

// libsonic's sonicGetSpeed: what the TSM stage runs at right now -- the last setter's value, or in nonlinear mode the
// speed of the last tension frame handed over (soniclib.c:354)
float sonicIntGetSpeed(sonicStream s) {
  settle(s);
  if (s->speedSet || !s->started) return s->tsmSpeed;
  if (!sync_stream(s)) return s->tsmSpeed;
  return s->curSpeedKnown;
}

int sonicSamplesAvailable(sonicStream s) {
  if (!sync_stream(s)) return 0;
  return (int)((s->rateMode ? s->finKnown : s->outKnown) - s->outRead);
}

int sonicReadShortFromStream(sonicStream s, short* out, int bufferSize) {
  if (s->pooled) return spx_pool_read(s, out, bufferSize);
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

// ---- the libsonic entry points the reference's tests call directly (sonic_test.cc:370,735-750).  They address the TSM
// stage alone: a write bypasses the shim's ring whatever the factor, a flush leaves waiting ring buffers where they are. ----
sonicStream sonicIntCreateStream(int sampleRate, int numChannels) { return sonicCreateStream(sampleRate, numChannels); }
void sonicIntDestroyStream(sonicStream s) { sonicDestroyStream(s); }
void sonicIntSetSpeed(sonicStream s, float speed) { settle(s); s->tsmSpeed = speed; s->speedSet = true; }  // the TSM stage alone (the shim's global speed stays)
void sonicIntSetRate(sonicStream s, float rate) { sonicSetRate(s, rate); }
int sonicIntWriteShortToStream(sonicStream s, const short* in, int n) {
  return write_shorts(s, in, n, true);
}
int sonicIntWriteFloatToStream(sonicStream s, const float* in, int n) {
  return write_floats(s, in, n, true);
}
int sonicIntReadShortFromStream(sonicStream s, short* out, int n) { return sonicReadShortFromStream(s, out, n); }
int sonicIntReadFloatFromStream(sonicStream s, float* out, int n) { return sonicReadFloatFromStream(s, out, n); }
int sonicIntFlushStream(sonicStream s);
int sonicIntSamplesAvailable(sonicStream s) { return sonicSamplesAvailable(s); }
void sonicIntSetUserData(sonicStream s, void* p) { s->userData = p; }
void* sonicIntGetUserData(sonicStream s) { return s->userData; }
// libsonic's public names for the same (include/compat/sonic.h without SONIC_INTERNAL)
void sonicSetUserData(sonicStream s, void* p) { s->userData = p; }
void* sonicGetUserData(sonicStream s) { return s->userData; }
float sonicGetSpeed(sonicStream s) { return sonicIntGetSpeed(s); }
int sonicGetSampleRate(sonicStream s) { return s->sampleRate; }
int sonicGetNumChannels(sonicStream s) { return s->channels; }

int sonicFlushStream(sonicStream s) {
  if (spx_stream_failed(s) || !spx_settings_ok(s)) return 0;
  (void)hipSetDevice(s->device);
  if (s->mode < 0) s->mode = (s->nonlinearFactor != 0.0f) ? 1 : 0;
  if (s->pooled && (s->rate != 1.0f || any_callback(s) || (s->mode == 1 && s->nonlinearFactor == 0.0f)) && !spx_pool_leave(s))
    return 0;
  if (s->pooled) return spx_pool_flush(s);
  // the flush itself does not look at the factor (soniclib.c:529-552): pending ring buffers go to the TSM stage at its
  // last speed.  With the factor at 0 by now there is no nonlinear job to do that in: the stream becomes a mixed one,
  // whose flushes append the pending buffers to the TSM input
  if (s->mode == 1 && s->nonlinearFactor == 0.0f && !s->mixed && !enter_mixed(s)) return 0;
  if (s->rate != 1.0f && !s->rateMode && !enter_rate_mode(s)) return 0;
  if (!staging(s, 0)) return 0;  // the job-table slot
  return launch_job(s, true);    // the stream stays usable: a later write continues behind the flush's padding
}
int sonicIntFlushStream(sonicStream s) {
  if (spx_stream_failed(s) || !spx_settings_ok(s)) return 0;
  (void)hipSetDevice(s->device);
  if (s->pooled && !spx_pool_leave(s)) return 0;
  if (s->mode < 0) s->mode = 0;
  if (s->mode == 1 && !s->mixed && !enter_mixed(s)) return 0;  // the TSM input goes its own way from here (buffers stay in the ring)
  if (s->rate != 1.0f && !s->rateMode && !enter_rate_mode(s)) return 0;
  if (!staging(s, 0)) return 0;
  return launch_job(s, true, true);
}

}  // extern "C"
