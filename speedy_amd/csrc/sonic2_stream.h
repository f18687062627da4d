// The stream handle behind include/sonic2.h and the pieces its two execution paths share (sonic2_api.hip: one
// handle per launch sequence, "eager"; sonic2_pool.hip: many handles per launch sequence, "coalesced").
#pragma once
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/sonic2.h"
#include "spx_internal.h"

// Deferred device work of a coalesced run (sonic2_pool.hip): buffer moves are not issued as stream operations of their
// own but collected and carried out by the run's stage kernel; the old allocations are freed after that kernel.
struct SpxMove { void* dst; const void* src; uint64_t bytes; };   // dst / src 16-byte aligned, bytes even; src == nullptr: zero fill
struct SpxPool;
// device blocks of a pool's cache (sonic2_pool.hip): power-of-two sizes, never handed back to the runtime while the pool lives
void* spx_pool_block_alloc(SpxPool* pool, size_t bytes, size_t* got);
void spx_pool_block_free(SpxPool* pool, void* p, size_t bytes);
struct SpxDeferred {
  SpxPool* pool = nullptr;
  std::vector<SpxMove> moves;
  std::vector<std::pair<void*, size_t>> frees;   // blocks that return to the pool's cache after the stage kernel
};

// A device array holding elements [origin, origin + cap) of a conceptually unbounded sequence.  ensure(lo, hi) makes
// [lo, hi) addressable and keeps what is already valid from lo on ([lo, filled)); it slides -- a stream-ordered copy
// into a fresh allocation, the old one freed in stream order -- when hi does not fit or when more than half the
// allocation is dead prefix.  base() is the pointer that, indexed with ABSOLUTE element numbers, lands in the allocation.
// `guard` elements in front of p[0] belong to the allocation too (zeroed, never meaningful): a reader that aligns its
// first position down may touch them.
// With a SpxDeferred (coalesced runs) nothing is issued: the block comes from the pool's cache, the copy and the guard's
// zero fill are planned as moves of the run's stage kernel, the old block returns to the cache after it.
template <class T>
struct SlideBuf {
  T* p = nullptr;
  int64_t origin = 0;  // absolute index of p[0]
  int64_t cap = 0;     // elements
  int64_t filled = 0;  // absolute end of valid data (set by the owner before ensure)
  int64_t guard = 0;   // addressable elements in front of p[0]
  size_t block = 0;    // != 0: the allocation is a block of that size from `owner`'s cache (not the runtime's stream-ordered pool)
  SpxPool* owner = nullptr;
  // does [lo, hi) fit as things are (and is the dead prefix still small)?
  bool fits(int64_t lo, int64_t hi) const {
    return p && lo >= origin && hi <= origin + cap && lo - origin <= cap / 2;
  }
  void drop_old(hipStream_t st, SpxDeferred* defer) {
    if (!p) return;
    if (block && defer) defer->frees.push_back({p - guard, block});
    else if (block) { (void)hipStreamSynchronize(st); spx_pool_block_free(owner, p - guard, block); }   // (a stream that left its pool: rare)
    else (void)hipFreeAsync(p - guard, st);
  }
  // move the window so that it starts at lo and holds at least [lo, hi), keeping [lo, filled)
  bool slide_to(int64_t lo, int64_t hi, hipStream_t st, int64_t min_cap, SpxDeferred* defer = nullptr) {
    constexpr int64_t A = 16 / (int64_t)sizeof(T) > 0 ? 16 / (int64_t)sizeof(T) : 1;   // elements per 16 bytes
    if (defer && p && lo >= origin) lo -= (lo - origin) % A;   // deferred moves copy 16-byte units: keep source and destination aligned
    int64_t ncap = std::max<int64_t>(min_cap, 2 * (hi - lo));
    ncap = (ncap + A - 1) / A * A;
    T* np = nullptr;
    size_t nblock = 0;
    if (defer) {
      np = static_cast<T*>(spx_pool_block_alloc(defer->pool, (size_t)(ncap + guard) * sizeof(T) + 16, &nblock));
      if (!np) return false;
      ncap = (int64_t)((nblock - 16) / sizeof(T)) - guard;   // the whole block is usable
      ncap -= ncap % A;
    } else if (hipMallocAsync(reinterpret_cast<void**>(&np), (size_t)(ncap + guard) * sizeof(T) + 16, st) != hipSuccess) {
      return false;
    }
    if (guard) {
      if (defer) defer->moves.push_back({np, nullptr, (uint64_t)(((size_t)guard * sizeof(T) + 15) & ~(size_t)15)});
      else (void)hipMemsetAsync(np, 0, (size_t)guard * sizeof(T), st);
      np += guard;
    }
    if (p && filled > lo && lo >= origin) {
      const size_t bytes = (size_t)(std::min(filled, origin + cap) - lo) * sizeof(T);
      if (defer) defer->moves.push_back({np, p + (lo - origin), (uint64_t)bytes});   // exact: what lies behind is somebody else's to write
      else if (hipMemcpyAsync(np, p + (lo - origin), bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
    }
    drop_old(st, defer);
    p = np;
    origin = lo;
    cap = ncap;
    block = nblock;
    owner = defer ? defer->pool : nullptr;
    return true;
  }
  bool ensure(int64_t lo, int64_t hi, hipStream_t st, int64_t min_cap = 4096, SpxDeferred* defer = nullptr) {
    if (p && lo < origin) lo = origin;  // what was dropped stays dropped
    if (lo < 0) lo = 0;
    if (hi < lo) hi = lo;
    if (fits(lo, hi)) return true;
    return slide_to(lo, hi, st, min_cap, defer);
  }
  T* base() const { return p - origin; }  // only ever dereferenced at indices >= origin
  void release(hipStream_t st) {
    drop_old(st, nullptr);
    p = nullptr;
    cap = 0;
    block = 0;
  }
};

struct sonicStreamStruct {  // speedyConnectionStruct + the parts of libsonic's stream the API exposes
  const SpxPlanDev* plan = nullptr;
  int device = 0;
  int sampleRate = 0, channels = 0;
  float globalSpeed = 1.0f;         // soniclib.c:114
  float tsmSpeed = 1.0f;            // the speed last given to the TSM stage by a setter (sonicSetSpeed sets both, sonicIntSetSpeed this one)
  bool speedupOnly = true;          // every launch so far had speed > 1 and 0 <= nonlinear factor <= 1
  float nonlinearFactor = 0.0f;     // soniclib.c:117
  float feedbackStrength = 0.1f;    // soniclib.c:122
  float rate = 1.0f;
  int bufferSize = 0;               // 0 until the first nonlinear write (soniclib.c:195, sonic_test.cc:496)
  int mode = -1;                    // -1 unknown, 0 linear, 1 nonlinear: what the last write was (soniclib.c:397-399 decides per write)
  // A stream that has been written in both modes ("mixed"): the ring sequence (what nonlinear writes brought: the
  // analysis input, handed to the TSM stage buffer by buffer as tensions arrive) and the TSM stage's input (ring buffers
  // in hand-over order, linear writes in between, as the reference interleaves them) are two different sequences.
  // dIn / nIn stay the ring sequence; dTsm / tPhys hold the TSM input, filled by device-to-device copies of the ring
  // buffers at hand-over time and by the linear writes.
  bool mixed = false;
  int64_t tPhys = 0;                // mixed: frames of TSM input materialised so far (physical index = TSM position - tsmShift)
  int64_t handedHost = 0;           // ring buffers handed to the TSM stage so far (the host's mirror of the device count)
  bool tensionStarted = false;      // the tension kernel's filter states have been initialised
  tensionFunction cbTension = nullptr;
  speedFunction cbSpeed = nullptr;
  featuresFunction cbFeatures = nullptr;
  spectrogramFunction cbSpectrogram = nullptr, cbNormalized = nullptr;

  hipStream_t hs = nullptr;
  SlideBuf<int16_t> dIn, dOut;      // elements = int16 values (frames * channels); dOut = what the TSM stage produces
  SlideBuf<int16_t> dTsm;           // mixed streams: the TSM stage's input
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
  float curSpeedKnown = 1.0f;  // the TSM stage's speed as of the last synchronisation
  int writesSinceSync = 0;
  bool dirty = false;       // launches in flight since the last synchronisation
  bool started = false;     // a job has been launched (state record valid)
  std::atomic<bool> failed{false};
  std::string errText;         // why (written once, before `failed` is set: spx_stream_fail); surfaced by the handle's next call
  void* userData = nullptr;    // sonicIntSetUserData (soniclib.c:98,106)

  // ---- coalesced execution (sonic2_pool.hip).  A handle starts in its device's pool and stays there while nothing it
  // does needs a launch sequence of its own (callbacks, rate stage, sonicInt* calls, a change of mode inside the stream);
  // leaving is one-way.  A pooled handle has no work in flight between API calls: writes and flushes are staged on the
  // host, the pool runs them for all waiting handles in one launch sequence, synchronously.
  bool pooled = false;
  bool poolPending = false;     // on the pool's waiting list
  bool inRun = false;           // part of the pool's run in flight: every call on the handle waits for that run (pool mutex)
  bool pendingFlush = false;    // ... with a flush behind the staged writes
  int64_t devIn = 0;            // frames of input that have reached dIn (nIn counts the staged ones too)
  struct Seg { int64_t pos; size_t src_off; int64_t frames; };   // staged write: stream position, offset in the pool's pinned input area
  std::vector<Seg> segs;
  int64_t arenaStart = -1;      // dRec / dScr live in the pool's frame arena: first frame slot, slots
  int64_t arenaCap = 0;
  // the produced frames nobody has read yet, [outRead, outKnown), as the pool's gather kernel delivered them: reads of a
  // pooled handle are host copies
  std::vector<int16_t> hostOut;
  size_t hostHead = 0;          // element index of frame outRead
};

// What one job (everything new since the previous one, sonic2_api.hip) is going to do: filled by prepare_job -- which also
// makes room in the device buffers, stream-ordered on `hs` -- consumed by the launch code and by finish_job.
struct SpxJobPlan {
  SpxStreamDev JA, JW;       // the analysis / tension kernels' view and the walk kernel's (they differ on mixed streams)
  bool hasRing = false, nonlinear = false, taps = false, speedupKernel = false, flush = false, direct = false;
  int64_t T = 0, fa = 0, need = 0, handedAfter = 0;
  int oldR = 0, newR = 0;
  int tiles = 0;             // analysis tiles of this job (plan tile size)
  SlideBuf<int16_t>* tsmIn = nullptr;
};
// pool == nullptr: offsets relative to the handle's own allocations; else absolute (kernels get null base pointers) and
// frame records in the pool's arena
int spx_prepare_job(sonicStream s, bool flush, bool direct, hipStream_t hs, SpxPool* pool, SpxJobPlan& J,
                    SpxDeferred* defer = nullptr);
void spx_finish_job(sonicStream s, const SpxJobPlan& J);
void spx_api_error(const std::string& msg);
// A handle fails for good: the reason is kept IN the handle -- a coalesced run may be triggered by another thread's call, and
// speedyHipLastError() is per thread -- and repeated as the calling thread's last error.
inline void spx_stream_fail(sonicStream s, const std::string& msg) {
  if (!s->failed.load(std::memory_order_acquire)) { s->errText = msg; s->failed.store(true, std::memory_order_release); }
  spx_api_error(msg);
}
// true (and the handle's reason becomes this thread's last error) when the handle has failed
inline bool spx_stream_failed(sonicStream s) {
  if (!s->failed.load(std::memory_order_acquire)) return false;
  if (!s->errText.empty()) spx_api_error(s->errText);
  return true;
}
bool spx_settings_ok(sonicStream s);

// ---- sonic2_pool.hip ----
SpxPool* spx_pool_for_device(int device, int coalesce = -1);   // nullptr when coalescing is off (coalesce: -1 the process default, 0 / 1 this handle's own choice)
void spx_pool_adopt(SpxPool* pool, sonicStream s);     // at creation
int spx_pool_write(sonicStream s, const short* in, int sampleCount);   // stage a write
int spx_pool_flush(sonicStream s);                                     // stage a flush
bool spx_pool_sync(sonicStream s);                     // run everything that waits (if s waits); false: s has failed
bool spx_pool_leave(sonicStream s);                    // hand the stream to the eager path (runs what waits first)
void spx_pool_forget(sonicStream s);                   // at destruction: off the lists, arena slots returned
int spx_pool_read(sonicStream s, short* out, int bufferSize);
// frame-arena slide of a pooled handle's records: make [keep, hi) addressable, keeping [keep, filled)
bool spx_pool_slide_frames(SpxPool* pool, sonicStream s, int64_t keep, int64_t hi, int64_t filled, SpxDeferred* defer);
const SpxFrameRec* spx_pool_arena_rec(SpxPool* pool);

