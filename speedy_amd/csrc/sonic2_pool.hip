// Coalesced execution of the drop-in streaming API (include/sonic2.h): many sonicStream handles, ONE launch sequence.
//
// The reference serves one handle per call, synchronously on the CPU (soniclib.c:391-452 -> :246-373).  A GPU pays a
// fixed price per launch sequence (a host-to-device copy, three kernels, a synchronisation: ~150 us), so a server that
// feeds hundreds of handles in turn would spend all of it on overheads if every write ran alone.  Here a write (or a
// flush) on a plain handle -- no monitoring callbacks, no rate stage, one mode -- only copies the samples into a pinned
// host area and puts the handle on its device's waiting list.  The first call that needs a RESULT on a waiting handle
// (sonicReadShortFromStream, sonicSamplesAvailable, a setter, a second flush ...) runs everything that waits:
//
//   stage kernel   job tables + every staged write, read straight out of pinned host memory, scattered into the
//                  handles' device-resident input sequences; state records gathered into one array
//   analysis / tension / walk   the batch kernels over N-stream job tables (spx_launch_*), one sequence per
//                  (sample rate, kernel variant) group
//   gather kernel  state records back to their handles; the new output frames of every handle and its state record
//                  written straight into pinned host memory
//
// and one synchronisation.  Reads are host copies after that.  The jobs are the ones the eager path would have run
// (spx_prepare_job / spx_finish_job, sonic2_api.hip), on the same kernels; several writes of one handle that wait
// together become one job, which the kernels' event model treats like the reference treats consecutive writes.  Per-handle
// results equal the reference's call for call (tests/test_gpu_pool.py: interleaved handles against the oracle shim).
//
// Several host threads (round 6).  A stream is not thread-safe, distinct streams are independent (sonic2.h:54-84) -- so a server
// runs the reference's loop (speedy_wave.cc:199-220: write a chunk, read what is ready) on one thread per group of handles.  Until
// round 6 the pool's mutex was held across a whole run, GPU wait included: a thread could not even STAGE a write while another
// thread's run was in flight, every read ran a launch sequence for one or two handles, and T threads were no faster than one
// (14 Msamples/s whatever the number of handles).  Now (flat combining):
//   * one run at a time, but the mutex is released while the GPU works: other threads stage their writes meanwhile, into the second
//     of two pinned input areas (the run's stage kernel is still reading the first);
//   * a read that needs a result and finds a run in flight waits for it; the first thread to find the pool idle becomes the
//     combiner and runs EVERYTHING staged by any thread -- after a bounded look (a few microseconds at a time while new work keeps
//     arriving, SPX_POOL_GATHER_US in all) for the threads that were just served and are about to write again -- so T threads
//     in the write -> read order cost one launch sequence per round, not T;
//   * a call on a handle that is part of the run in flight waits for that run (its state is the run's until the results are in).
// Results per handle are what they always were: the jobs are formed per handle from that handle's own staged writes.
//
// Device memory: a handle keeps its own sliding input / output allocations (sonic2_stream.h); the kernels of a pooled
// launch get NULL base pointers and per-stream offsets that are the allocations' absolute element addresses.  The
// per-frame arrays (records, scratch) are indexed by the kernels through ONE offset per stream, so for pooled handles
// they live in a pool-wide frame arena: slot f = {rec[f], scr[4f..4f+3]}; a handle owns a power-of-two range of slots.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "sonic2_stream.h"

static std::atomic<int> g_coalesce{-1};   // -1: not decided yet (environment), 0 off, 1 on

struct PinBuf {   // growable pinned host area
  unsigned char* p = nullptr;
  size_t cap = 0;
  bool reserve(size_t need, size_t keep) {
    if (need <= cap) return true;
    size_t n = cap ? cap : (1u << 20);
    while (n < need) n *= 2;
    unsigned char* np = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&np), n, hipHostMallocDefault) != hipSuccess) return false;
    if (p && keep) memcpy(np, p, keep);
    if (p) (void)hipHostFree(p);
    p = np;
    cap = n;
    return true;
  }
};

struct PoolCopy {      // one piece of a staged write
  int16_t* dst;        // device
  uint32_t src_off;    // bytes into the pinned input area, 16-byte aligned
  uint32_t n;          // int16 values
};
struct PoolDesc {      // per waiting handle, read by the stage and gather kernels from pinned memory
  SpxStreamState* home;     // the handle's own state record (device)
  const int16_t* out_base;  // dOut.base(): indexed with absolute element numbers
  int64_t out_from;         // frames already known to the host
  int64_t out_cap;          // absolute frame capacity of the output window (SpxStreamDev::out_cap)
  int64_t res_off;          // this handle's slice of the pinned output area, in int16 values
  int64_t res_cap;          // ... and its size in frames
  int32_t channels;
  int32_t pad;
};
struct PoolResult { SpxStreamState st; int64_t n; };

struct SpxPool {
  std::mutex mu;
  std::condition_variable cv;      // a run has finished (running -> false, its handles' inRun -> false)
  bool running = false;            // a run owns hTab / hRes / dWs / the stream, and the input area it was staged in
  int waiters = 0;                 // threads inside wait_for_run (mutex)
  std::atomic<unsigned> gen{0};    // runs finished: a waiting thread polls it for a while before it blocks (wait_for_run)
  std::thread::id first_thread;    // the batching look costs a single-threaded caller nothing: only once a second thread is seen
  bool have_first = false, multi = false;
  int device = 0;
  hipStream_t hs = nullptr;
  // frame arena
  SpxFrameRec* aRec = nullptr;
  float* aScr = nullptr;
  int64_t aCap = 0, aBump = 0;
  std::vector<int64_t> freeList[48];   // by log2(slots)
  std::vector<sonicStream> members;    // handles that own arena slots
  std::vector<std::pair<int64_t, int64_t>> arenaLater;   // ranges to free once the run in preparation has been launched
  // waiting work
  std::vector<sonicStream> waiting;
  std::vector<sonicStream> run;            // the handles of the run in flight
  size_t waitingSegs = 0;
  PinBuf hInBuf[2], hTab, hRes;    // writes are staged in hInBuf[hCur]; a run reads the other one
  int hCur = 0;
  size_t hInUsed = 0;
  unsigned char* dWs = nullptr;
  size_t dWsCap = 0;
  unsigned long long runs = 0, jobs = 0;   // statistics (speedyHipPoolStats)
  SpxDeferred* defer = nullptr;            // the run in preparation
  struct Item { sonicStream s; SpxJobPlan J; std::vector<sonicStreamStruct::Seg> segs; bool flush; };
  std::vector<Item> items;                 // the run's tables (capacity kept from run to run)
  std::vector<PoolCopy> copies;
  std::vector<SpxMove> moves;
  std::vector<int64_t> res_off, res_cap;
  std::vector<void*> blocks[48];           // device block cache by log2(bytes)
  double t_prep = 0, t_tab = 0, t_launch = 0, t_wait = 0, t_post = 0;   // SPX_POOL_TIMES=1: host seconds per phase
  double t_l[5] = {0, 0, 0, 0, 0};   // ... and per launch of a one-group round: stage, analysis, tension, walk, gather
};

static SpxPool* g_pools[64];
static std::mutex g_pools_mu;

static int coalesce_default() {
  int on = g_coalesce.load();
  if (on < 0) {
    on = getenv("SPX_NO_POOL") ? 0 : 1;
    int expected = -1;
    g_coalesce.compare_exchange_strong(expected, on);   // (a speedyHipSetCoalescing call that raced us wins)
    on = g_coalesce.load();
  }
  return on;
}
SpxPool* spx_pool_for_device(int device, int coalesce) {
  const int on = coalesce < 0 ? coalesce_default() : (coalesce ? 1 : 0);
  if (!on || device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> g(g_pools_mu);
  if (!g_pools[device]) {
    SpxPool* p = new SpxPool();
    p->device = device;
    if (hipStreamCreateWithFlags(&p->hs, hipStreamNonBlocking) != hipSuccess) { delete p; return nullptr; }
    g_pools[device] = p;
  }
  return g_pools[device];
}
static SpxPool* pool_of(sonicStream s) { return g_pools[s->device]; }

void spx_pool_adopt(SpxPool* pool, sonicStream s) {
  (void)pool;
  s->pooled = true;
}
const SpxFrameRec* spx_pool_arena_rec(SpxPool* pool) { return pool->aRec; }

static int log2ceil(int64_t v) { int l = 0; while (((int64_t)1 << l) < v) l++; return l; }

// ---------------- device block cache ----------------
// Sliding buffers of pooled handles are re-allocated all the time (every few dozen writes per handle); the runtime's
// stream-ordered allocator costs 5-10 us per call, a free list costs nothing.  Pool mutex held.
void* spx_pool_block_alloc(SpxPool* P, size_t bytes, size_t* got) {
  int l = log2ceil((int64_t)bytes);
  if (l < 16) l = 16;
  *got = (size_t)1 << l;
  if (!P->blocks[l].empty()) { void* q = P->blocks[l].back(); P->blocks[l].pop_back(); return q; }
  void* q = nullptr;
  if (hipMalloc(&q, *got) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return q;
}
static void block_free_locked(SpxPool* P, void* p, size_t bytes) {
  if (p) P->blocks[log2ceil((int64_t)bytes)].push_back(p);
}
void spx_pool_block_free(SpxPool* P, void* p, size_t bytes) {   // from outside a run (a stream that has left the pool)
  std::lock_guard<std::mutex> g(P->mu);
  block_free_locked(P, p, bytes);
}

// ---------------- frame arena ----------------

static bool arena_grow(SpxPool* P, int64_t min_cap) {
  static const int64_t first = [] { const char* e = getenv("SPX_POOL_FRAMES"); return std::max<int64_t>(1024, e ? atoll(e) : (int64_t)1 << 20); }();
  int64_t ncap = P->aCap ? 2 * P->aCap : first;
  while (ncap < min_cap) ncap *= 2;
  SpxFrameRec* nr = nullptr;
  float* ns = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&nr), sizeof(SpxFrameRec) * (size_t)ncap) != hipSuccess) return false;
  if (hipMalloc(reinterpret_cast<void**>(&ns), sizeof(float) * 4 * (size_t)ncap) != hipSuccess) { (void)hipFree(nr); return false; }
  if (P->aRec) {   // everything moves; the handles' pointers follow (offsets relative to the arena base do not change)
    (void)hipMemcpyAsync(nr, P->aRec, sizeof(SpxFrameRec) * (size_t)P->aBump, hipMemcpyDeviceToDevice, P->hs);
    (void)hipMemcpyAsync(ns, P->aScr, sizeof(float) * 4 * (size_t)P->aBump, hipMemcpyDeviceToDevice, P->hs);
    (void)hipStreamSynchronize(P->hs);
    (void)hipFree(P->aRec);
    (void)hipFree(P->aScr);
  }
  if (P->defer)   // moves already planned for the run in preparation point into the old arrays
    for (SpxMove& mv : P->defer->moves) {
      auto fix = [&](const void* q) -> void* {
        const unsigned char* c = static_cast<const unsigned char*>(q);
        const unsigned char* r0 = reinterpret_cast<const unsigned char*>(P->aRec);
        const unsigned char* s0 = reinterpret_cast<const unsigned char*>(P->aScr);
        if (P->aRec && c >= r0 && c < r0 + sizeof(SpxFrameRec) * (size_t)P->aCap) return reinterpret_cast<unsigned char*>(nr) + (c - r0);
        if (P->aScr && c >= s0 && c < s0 + sizeof(float) * 4 * (size_t)P->aCap) return reinterpret_cast<unsigned char*>(ns) + (c - s0);
        return const_cast<void*>(q);
      };
      mv.dst = fix(mv.dst);
      if (mv.src) mv.src = fix(mv.src);
    }
  P->aRec = nr; P->aScr = ns; P->aCap = ncap;
  for (sonicStream m : P->members)
    if (m->arenaStart >= 0) { m->dRec.p = P->aRec + m->arenaStart; m->dScr.p = P->aScr + 4 * m->arenaStart; }
  return true;
}
static int64_t arena_alloc(SpxPool* P, int64_t slots) {   // slots: a power of two
  const int l = log2ceil(slots);
  if (!P->freeList[l].empty()) { const int64_t s = P->freeList[l].back(); P->freeList[l].pop_back(); return s; }
  if (P->aBump + slots > P->aCap && !arena_grow(P, P->aBump + slots)) return -1;
  const int64_t s = P->aBump;
  P->aBump += slots;
  return s;
}
static void arena_free(SpxPool* P, int64_t start, int64_t slots) {
  if (start >= 0 && slots > 0) P->freeList[log2ceil(slots)].push_back(start);
}

bool spx_pool_slide_frames(SpxPool* P, sonicStream s, int64_t keep, int64_t hi, int64_t filled, SpxDeferred* defer) {
  if (s->arenaStart >= 0 && keep >= s->dRec.origin) keep -= (keep - s->dRec.origin) & 1;   // moves copy 16-byte units (two records)
  int64_t ncap = 1024;
  while (ncap < 2 * (hi - keep)) ncap *= 2;
  const int64_t nstart = arena_alloc(P, ncap);   // (may move the arena: old pointers are refreshed by arena_grow)
  if (nstart < 0) { spx_api_error("frame arena allocation failed"); return false; }
  if (s->arenaStart < 0) P->members.push_back(s);
  if (s->arenaStart >= 0 && filled > keep && keep >= s->dRec.origin) {
    const int64_t n = std::min(filled, s->dRec.origin + s->dRec.cap) - keep;
    const int64_t from = s->arenaStart + (keep - s->dRec.origin);
    defer->moves.push_back({P->aRec + nstart, P->aRec + from, (uint64_t)(sizeof(SpxFrameRec) * (size_t)n)});
    defer->moves.push_back({P->aScr + 4 * nstart, P->aScr + 4 * from, (uint64_t)(sizeof(float) * 4 * (size_t)n)});
  }
  // the old range goes back to the free lists only after the run (another handle's move must not land in it before this
  // handle's move has read it: all moves of a run happen in one kernel)
  if (s->arenaStart >= 0) P->arenaLater.push_back({s->arenaStart, s->arenaCap});
  s->arenaStart = nstart; s->arenaCap = ncap;
  s->dRec.p = P->aRec + nstart; s->dRec.origin = keep; s->dRec.cap = ncap;
  s->dScr.p = P->aScr + 4 * nstart; s->dScr.origin = 4 * keep; s->dScr.cap = 4 * ncap;
  return true;
}
static void arena_release(SpxPool* P, sonicStream s) {
  if (s->arenaStart < 0) return;
  arena_free(P, s->arenaStart, s->arenaCap);
  s->arenaStart = -1; s->arenaCap = 0;
  s->dRec.p = nullptr; s->dRec.cap = 0; s->dScr.p = nullptr; s->dScr.cap = 0;
  for (size_t i = 0; i < P->members.size(); i++)
    if (P->members[i] == s) { P->members[i] = P->members.back(); P->members.pop_back(); break; }
}

// ---------------- kernels ----------------
// Job tables into the workspace, the handles' state records into the launch's state array, every staged write from the
// pinned input area (read over the host link, 16 bytes per lane) into its handle's input sequence.
__global__ void __launch_bounds__(256)
spx_pool_stage_kernel(const unsigned* __restrict__ tab_src, unsigned* __restrict__ tab_dst, unsigned n_words,
                      const PoolDesc* __restrict__ desc, SpxStreamState* __restrict__ states, unsigned n,
                      const PoolCopy* __restrict__ copies, unsigned n_copies, const unsigned char* __restrict__ hin,
                      const SpxMove* __restrict__ moves, unsigned n_moves) {
  if (blockIdx.x >= n_copies && blockIdx.x < n_copies + n_moves) {   // a buffer move (or a zero fill), 16 bytes per lane
    const SpxMove m = moves[blockIdx.x - n_copies];
    uint4* __restrict__ d = static_cast<uint4*>(m.dst);
    const uint4* __restrict__ q = static_cast<const uint4*>(m.src);
    const unsigned n16 = (unsigned)(m.bytes >> 4), tail = (unsigned)(m.bytes & 15) >> 1;   // exact: nothing behind the range is touched
    for (unsigned k = threadIdx.x; k < n16; k += 256) d[k] = q ? q[k] : make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x < tail) {
      unsigned short* dt = reinterpret_cast<unsigned short*>(d + n16);
      const unsigned short* qt = reinterpret_cast<const unsigned short*>(q + n16);
      dt[threadIdx.x] = q ? qt[threadIdx.x] : (unsigned short)0;
    }
  }
  const unsigned gt = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  for (unsigned i = gt; i < n_words; i += stride) tab_dst[i] = tab_src[i];
  constexpr unsigned SW = sizeof(SpxStreamState) / 4;
  for (unsigned i = gt; i < n * SW; i += stride) {
    const unsigned h = i / SW, w = i - h * SW;
    reinterpret_cast<unsigned*>(states + h)[w] = reinterpret_cast<const unsigned*>(desc[h].home)[w];
  }
  if (blockIdx.x < n_copies) {
    const PoolCopy c = copies[blockIdx.x];
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(hin + c.src_off);
    for (unsigned k = threadIdx.x * 8; k < c.n; k += 256 * 8) {
      const uint4 v = src[k >> 3];   // (the area is padded: a read past the write's end stays inside it)
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
      int16_t* d = c.dst + k;
      const unsigned m = c.n - k;
#pragma unroll
      for (int j = 0; j < 8; j++)
        if ((unsigned)j < m) d[j] = (int16_t)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
    }
  }
}

// State records back home and to the host; the frames each handle produced beyond `out_from` to the pinned output area.
__global__ void __launch_bounds__(256)
spx_pool_gather_kernel(const PoolDesc* __restrict__ desc, const SpxStreamState* __restrict__ states,
                       const int64_t* __restrict__ nout, PoolResult* __restrict__ res, int16_t* __restrict__ hout) {
  const unsigned i = blockIdx.x;
  const PoolDesc D = desc[i];
  const int64_t k = nout[i];
  if (blockIdx.y == 0) {
    constexpr unsigned SW = sizeof(SpxStreamState) / 4;
    if (threadIdx.x < SW) {
      const unsigned v = reinterpret_cast<const unsigned*>(states + i)[threadIdx.x];
      reinterpret_cast<unsigned*>(D.home)[threadIdx.x] = v;
      reinterpret_cast<unsigned*>(&res[i].st)[threadIdx.x] = v;
    }
    if (threadIdx.x == 0) res[i].n = k;
  }
  // a negative count flags an overflowed stream (the frames that fitted are there); INT64_MIN a lost producer
  int64_t produced = (k == INT64_MIN) ? D.out_from : (k < 0 ? -k : k);
  if (produced > D.out_cap) produced = D.out_cap;
  int64_t fresh = produced - D.out_from;
  if (fresh > D.res_cap) fresh = D.res_cap;
  const int64_t elems = fresh * D.channels;
  const int16_t* __restrict__ src = D.out_base + D.out_from * D.channels;
  int16_t* __restrict__ dst = hout + D.res_off;
  for (int64_t e = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4; e < elems; e += (int64_t)gridDim.y * 1024) {
    unsigned short v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (e + j < elems) ? (unsigned short)src[e + j] : (unsigned short)0;
    uint2 o;
    o.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
    o.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
    *reinterpret_cast<uint2*>(dst + e) = o;   // res_off is a multiple of 8 values; the slice is padded to one
  }
}

// ---------------- running what waits ----------------
static bool place_input(SpxPool* P, sonicStream s, const std::vector<sonicStreamStruct::Seg>& segs, std::vector<PoolCopy>& copies,
                        SpxDeferred* defer) {
  if (segs.empty()) return true;
  const SpxPlanDev& PL = *s->plan;
  const int64_t C = s->channels;
  SlideBuf<int16_t>& dst = s->dIn;
  // oldest frame still needed: what the TSM stage has buffered (its window refill aligns down by 8 frames) and the
  // analysis halo (frame framesDone-1 starts at (framesDone-1)*B)
  int64_t keepFrom = s->tsmBase - s->tsmShift - 16;
  if (s->mode == 1) keepFrom = std::min(keepFrom, (s->framesDone - 1) * (int64_t)PL.B - 16);
  if (!s->started) keepFrom = std::min(keepFrom, dst.origin / C);
  if (keepFrom < 0) keepFrom = 0;
  dst.filled = s->devIn * C;
  if (!dst.ensure(keepFrom * C, s->nIn * C + 64, P->hs, 1 << 17, defer)) return false;
  for (const auto& g : segs) {
    const int64_t total = g.frames * C;
    for (int64_t k = 0; k < total; k += 16384) {
      PoolCopy c;
      c.dst = dst.base() + g.pos * C + k;
      c.src_off = (uint32_t)(g.src_off + 2 * (size_t)k);
      c.n = (uint32_t)std::min<int64_t>(16384, total - k);
      copies.push_back(c);
    }
  }
  return true;
}

// A run is over (or could not be launched): its handles belong to their callers again.  Pool mutex held.
static void end_run(SpxPool* P, std::vector<sonicStream>& run) {
  for (sonicStream s : run) s->inRun = false;
  run.clear();
  P->running = false;
  P->gen.fetch_add(1, std::memory_order_release);
  P->cv.notify_all();
}

// Everything that waits, one launch sequence per (plan, kernel variant) group, one synchronisation.  Called with the pool mutex
// held through `lk`, with no run in flight; returns with it held.  The mutex is released twice on the way: for the bounded look
// for more staged work (several threads only), and while the GPU works.
static bool pool_run(SpxPool* P, std::unique_lock<std::mutex>& lk) {
  if (P->waiting.empty()) return true;
  P->running = true;   // from here on other threads stage and wait; nobody else starts a run
  if (P->multi) {
    // The combiner's look: threads whose handles the previous run served are writing again right now -- a few microseconds at a
    // time while the waiting list keeps growing, SPX_POOL_GATHER_US (default 24) in all.  One launch sequence for all of them
    // instead of one for those that happened to be staged and one for the rest.
    static const double gather_us = [] { const char* e = getenv("SPX_POOL_GATHER_US"); return e ? atof(e) : 24.0; }();
    const auto t_g = std::chrono::steady_clock::now();
    size_t seen;
    do {
      seen = P->waiting.size();
      lk.unlock();
      const auto t_s = std::chrono::steady_clock::now();
      while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_s).count() < 4.0) { }
      lk.lock();
    } while (P->waiting.size() > seen && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_g).count() < gather_us);
  }
  (void)hipSetDevice(P->device);
  typedef SpxPool::Item Item;
  // (the run's tables live in the pool and keep their capacity: no allocation on the hot path)
  std::vector<Item>& items = P->items;
  std::vector<PoolCopy>& copies = P->copies;
  items.clear();
  copies.clear();
  // the run takes the waiting list and the input area it was staged in; staging goes on in the other area
  std::vector<sonicStream>& run = P->run;
  run.swap(P->waiting);
  P->waiting.clear();
  P->waitingSegs = 0;
  const unsigned char* run_in = P->hInBuf[P->hCur].p;
  P->hCur ^= 1;
  P->hInUsed = 0;
  items.reserve(run.size());
  SpxDeferred defer;
  defer.pool = P;
  P->defer = &defer;
  const auto tp0 = std::chrono::steady_clock::now();
  for (sonicStream s : run) {
    s->poolPending = false;
    s->inRun = true;
    items.emplace_back();
    Item& it = items.back();   // prepared in place
    it.s = s;
    it.segs.clear();
    it.segs.swap(s->segs);
    it.flush = s->pendingFlush;
    s->pendingFlush = false;
    if (s->failed || !place_input(P, s, it.segs, copies, &defer) || !spx_prepare_job(s, it.flush, false, P->hs, P, it.J, &defer)) {
      if (!s->failed) spx_stream_fail(s, "preparing the staged work failed (device allocation)");
      items.pop_back();
    }
  }
  P->defer = nullptr;
  // whatever happens below, the replaced allocations are released behind everything enqueued so far and the replaced
  // arena ranges return to the free lists
  struct Releaser {
    SpxPool* P; SpxDeferred* d;
    ~Releaser() {
      for (auto& q : d->frees) block_free_locked(P, q.first, q.second);
      for (auto& r : P->arenaLater) arena_free(P, r.first, r.second);
      P->arenaLater.clear();
    }
  } releaser{P, &defer};
  const size_t n = items.size();
  if (n == 0) { end_run(P, run); return true; }
  // moves in pieces of 64 KB, one workgroup each
  std::vector<SpxMove>& moves = P->moves;
  moves.clear();
  for (const SpxMove& m : defer.moves)
    for (uint64_t k = 0; k < m.bytes; k += 65536)
      moves.push_back({static_cast<unsigned char*>(m.dst) + k, m.src ? static_cast<const unsigned char*>(m.src) + k : nullptr,
                       std::min<uint64_t>(65536, m.bytes - k)});
  const auto tp1 = std::chrono::steady_clock::now();
  // groups: same plan (sample rate, hysteresis mode), same walk kernel family
  auto before = [](const Item& a, const Item& b) {
    if (a.s->plan != b.s->plan) return a.s->plan < b.s->plan;
    return (int)a.J.speedupKernel < (int)b.J.speedupKernel;
  };
  if (!std::is_sorted(items.begin(), items.end(), before)) std::stable_sort(items.begin(), items.end(), before);   // (one kind of handle: nothing to do)
  struct Group { size_t i0, i1; int tiles; int maxC; };
  std::vector<Group> groups;
  for (size_t i = 0; i < n;) {
    size_t j = i;
    Group g = {i, i, 0, 1};
    while (j < n && items[j].s->plan == items[i].s->plan && items[j].J.speedupKernel == items[i].J.speedupKernel) {
      items[j].J.JA.first_tile = g.tiles;
      g.tiles += items[j].J.tiles;
      g.maxC = std::max(g.maxC, items[j].s->channels);
      j++;
    }
    g.i1 = j;
    groups.push_back(g);
    i = j;
  }
  // ---- pinned tables: jobsA[n] | jobsW[n] | desc[n] | copies[m]; pinned results: PoolResult[n] | output slices ----
  const size_t b_jobs = sizeof(SpxStreamDev) * n;
  const size_t o_desc = 2 * b_jobs, o_copies = o_desc + sizeof(PoolDesc) * n;
  const size_t o_moves = (o_copies + sizeof(PoolCopy) * copies.size() + 15) & ~(size_t)15;
  const size_t b_tab = o_moves + sizeof(SpxMove) * moves.size();
  auto give_up = [&](const char* why) {   // nothing was launched: the waiting handles cannot be served
    for (auto& it : items) spx_stream_fail(it.s, why);
    end_run(P, run);
    return false;
  };
  if (!P->hTab.reserve(b_tab + 64, 0)) return give_up("pinned table allocation failed");
  size_t res_elems = 0;
  int64_t max_slice = 0;
  std::vector<int64_t>& res_off = P->res_off;
  std::vector<int64_t>& res_cap = P->res_cap;
  res_off.resize(n);
  res_cap.resize(n);
  for (size_t i = 0; i < n; i++) {
    sonicStream s = items[i].s;
    res_cap[i] = std::max<int64_t>(0, items[i].J.need - s->outKnown);
    res_off[i] = (int64_t)res_elems;
    const int64_t e = res_cap[i] * s->channels;
    max_slice = std::max(max_slice, e);
    res_elems += (size_t)((e + 7) & ~(int64_t)7);
  }
  const size_t o_out = (sizeof(PoolResult) * n + 63) & ~(size_t)63;
  if (!P->hRes.reserve(o_out + res_elems * sizeof(int16_t) + 64, 0)) return give_up("pinned result allocation failed");
  // device workspace: jobsA[n] | jobsW[n] | states[n] | nout[n]
  const size_t w_states = (2 * b_jobs + 63) & ~(size_t)63, w_nout = w_states + ((sizeof(SpxStreamState) * n + 63) & ~(size_t)63);
  const size_t w_total = w_nout + sizeof(int64_t) * n;
  if (w_total > P->dWsCap) {
    if (P->dWs) (void)hipFree(P->dWs);   // (nothing of this pool is in flight between runs)
    P->dWs = nullptr;
    size_t cap = P->dWsCap ? P->dWsCap : 65536;
    while (cap < w_total) cap *= 2;
    if (hipMalloc(reinterpret_cast<void**>(&P->dWs), cap) != hipSuccess) { P->dWs = nullptr; P->dWsCap = 0; (void)hipGetLastError(); return give_up("pool workspace allocation failed"); }
    P->dWsCap = cap;
  }
  SpxStreamDev* hA = reinterpret_cast<SpxStreamDev*>(P->hTab.p);
  SpxStreamDev* hW = hA + n;
  PoolDesc* hD = reinterpret_cast<PoolDesc*>(P->hTab.p + o_desc);
  PoolCopy* hC = reinterpret_cast<PoolCopy*>(P->hTab.p + o_copies);
  for (size_t i = 0; i < n; i++) {
    sonicStream s = items[i].s;
    hA[i] = items[i].J.JA;
    hW[i] = items[i].J.JW;
    hW[i].first_tile = hA[i].first_tile;
    PoolDesc& D = hD[i];
    D.home = s->dState;
    D.out_base = s->dOut.base();
    D.out_from = s->outKnown;
    D.out_cap = items[i].J.JA.out_cap;
    D.res_off = res_off[i];
    D.res_cap = res_cap[i];
    D.channels = s->channels;
    D.pad = 0;
  }
  if (!copies.empty()) memcpy(hC, copies.data(), sizeof(PoolCopy) * copies.size());
  SpxMove* hM = reinterpret_cast<SpxMove*>(P->hTab.p + o_moves);
  if (!moves.empty()) memcpy(hM, moves.data(), sizeof(SpxMove) * moves.size());
  SpxStreamDev* dA = reinterpret_cast<SpxStreamDev*>(P->dWs);
  SpxStreamDev* dW = dA + n;
  SpxStreamState* dStates = reinterpret_cast<SpxStreamState*>(P->dWs + w_states);
  int64_t* dNout = reinterpret_cast<int64_t*>(P->dWs + w_nout);
  PoolResult* hR = reinterpret_cast<PoolResult*>(P->hRes.p);
  int16_t* hOut = reinterpret_cast<int16_t*>(P->hRes.p + o_out);

  const unsigned n_words = (unsigned)(2 * b_jobs / 4);
  const unsigned grid = std::max<unsigned>(std::max<unsigned>((unsigned)(copies.size() + moves.size()), (n_words + 255) / 256), 1u);
  const auto tp2 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(spx_pool_stage_kernel, dim3(grid), dim3(256), 0, P->hs, reinterpret_cast<const unsigned*>(hA),
                     reinterpret_cast<unsigned*>(dA), n_words, hD, dStates, (unsigned)n, hC, (unsigned)copies.size(), run_in,
                     hM, (unsigned)moves.size());
  auto tl = std::chrono::steady_clock::now();
  auto lap = [&](int k) { const auto t = std::chrono::steady_clock::now(); P->t_l[k] += std::chrono::duration<double>(t - tl).count(); tl = t; };
  lap(0);
  const SpxTapsDev no_taps = {nullptr, nullptr, nullptr, nullptr, nullptr};
  for (const Group& g : groups) {
    const SpxPlanDev& PL = *items[g.i0].s->plan;
    const int ng = (int)(g.i1 - g.i0);
    if (g.tiles > 0) spx_launch_analysis(PL, dA + g.i0, ng, g.tiles, nullptr, P->aRec, no_taps, nullptr, nullptr, P->hs);
    lap(1);
    bool any_nl = false;
    for (size_t i = g.i0; i < g.i1; i++) any_nl = any_nl || items[i].J.nonlinear;
    if (any_nl) spx_launch_tension(PL, dA + g.i0, ng, dStates + g.i0, P->aRec, P->aScr, no_taps, nullptr, nullptr, P->hs);
    lap(2);
    spx_launch_walk(PL, dW + g.i0, ng, g.maxC, nullptr, nullptr, dNout + g.i0, dStates + g.i0, P->aScr, nullptr,
                    items[g.i0].J.speedupKernel, P->hs, /*short_jobs=*/true);
    lap(3);
  }
  const unsigned gy = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, (max_slice + 8191) / 8192));
  hipLaunchKernelGGL(spx_pool_gather_kernel, dim3((unsigned)n, gy), dim3(256), 0, P->hs, hD, dStates, dNout, hR, hOut);
  lap(4);
  const hipError_t le = hipGetLastError();
  const auto tp3 = std::chrono::steady_clock::now();
  // The run is a few tens of microseconds of GPU work: waiting for it in the runtime's blocking way costs about as much
  // again in wake-up latency.  Poll the stream for a while (SPX_POOL_SPIN_US, default 2000 us), then block.
  static const long spin_us = [] { const char* e = getenv("SPX_POOL_SPIN_US"); return e ? atol(e) : 2000L; }();
  // (the mutex is free while the GPU works: other threads stage their next writes -- on handles that are not part of this run --
  // and queue up behind it; everything this run owns stays untouched until the lock is back)
  lk.unlock();
  hipError_t se = hipErrorNotReady;
  if (spin_us > 0) {
    const auto t_spin = std::chrono::steady_clock::now();
    while ((se = hipStreamQuery(P->hs)) == hipErrorNotReady) {
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_spin).count() > (double)spin_us) break;
    }
    if (se == hipErrorNotReady) (void)hipGetLastError();
  }
  if (se == hipErrorNotReady) se = hipStreamSynchronize(P->hs);
  lk.lock();
  const auto tp4 = std::chrono::steady_clock::now();
  if (le != hipSuccess || se != hipSuccess) {
    const std::string why = std::string("coalesced launch failed: ") + hipGetErrorString(le != hipSuccess ? le : se);
    for (auto& it : items) spx_stream_fail(it.s, why);
    end_run(P, run);
    return false;
  }
  for (size_t i = 0; i < n; i++) {
    sonicStream s = items[i].s;
    spx_finish_job(s, items[i].J);
    int64_t k = hR[i].n;
    if (k == SPX_NOUT_LOST_PRODUCER) {
      spx_stream_fail(s, "a producer kernel never delivered its frames (device-side poll limit reached)");
      k = s->outKnown;
    } else if (k < 0) {
      spx_stream_fail(s, "output capacity exceeded on the device");
      k = -k;
    }
    // what the gather kernel delivered: the frames beyond outKnown that fitted the output window AND the handle's slice of
    // the pinned result area (sized from the host-side bound J.need).  The host's count follows what it actually holds: a
    // device that produced more than the bound is an error of this handle, never a read past hostOut.
    const int64_t avail = std::min(k, items[i].J.JA.out_cap) - s->outKnown;
    const int64_t fresh = std::max<int64_t>(0, std::min(avail, res_cap[i]));
    if (avail > res_cap[i] && !s->failed)
      spx_stream_fail(s, "the device produced more frames than the host-side bound of the job (coalesced result slice)");
    if (fresh > 0) {
      const int16_t* src = hOut + res_off[i];
      s->hostOut.insert(s->hostOut.end(), src, src + fresh * s->channels);
    }
    s->outKnown += fresh;
    s->outBound = s->outKnown;
    s->tsmBase = hR[i].st.w.base;
    s->curSpeedKnown = hR[i].st.curSpeed;
    s->dirty = false;
    s->writesSinceSync = 0;
    s->devIn = s->nIn;
  }
  P->runs++;
  P->jobs += n;
  end_run(P, run);
  const auto tp5 = std::chrono::steady_clock::now();
  auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  P->t_prep += sec(tp0, tp1); P->t_tab += sec(tp1, tp2); P->t_launch += sec(tp2, tp3); P->t_wait += sec(tp3, tp4); P->t_post += sec(tp4, tp5);
  return true;
}

static void enlist(SpxPool* P, sonicStream s) {
  if (!s->poolPending) { s->poolPending = true; P->waiting.push_back(s); }
}
// Wait for the run in flight to end.  A run is tens of microseconds: blocking in the kernel costs a thread about as much again to
// wake up, so it polls the run counter for a while first (mutex released; SPX_POOL_SPIN_US bounds it, 200 us at most), then blocks.
static void wait_for_run(SpxPool* P, std::unique_lock<std::mutex>& lk) {
  static const double spin_us = [] { const char* e = getenv("SPX_POOL_SPIN_US"); const double v = e ? atof(e) : 2000.0; return v < 200.0 ? v : 200.0; }();
  const unsigned g0 = P->gen.load(std::memory_order_acquire);
  P->waiters++;
  // (polling is for a handful of server threads on CPUs of their own; dozens of waiters -- more threads than CPUs, most likely --
  // block at once: 256 threads on 16 CPUs served 104 Msamples/s blocking and 52 polling, profiles/r06/r6h_api_threads.txt)
  if (spin_us > 0 && P->waiters <= 32) {
    lk.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      if (P->gen.load(std::memory_order_acquire) != g0) break;
      const double waited = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (waited >= spin_us) break;
      // (more waiting threads than CPUs -- a cgroup quota does not show in the affinity mask: after a short pure spin the poll gives
      // its time slice away, so that the combiner and the threads it serves are not kept off the CPUs by their own waiters)
      if (waited > 20.0) sched_yield();
#if defined(__x86_64__)
      else __builtin_ia32_pause();
#endif
    }
    lk.lock();
  }
  if (P->gen.load(std::memory_order_acquire) == g0 && P->running) P->cv.wait(lk);
  P->waiters--;
}
// every entry point: which threads use the pool (the batching look is for several), and a handle that is part of the run in
// flight is the run's until its results are in
static void enter(SpxPool* P, sonicStream s, std::unique_lock<std::mutex>& lk) {
  if (!P->multi) {
    const std::thread::id me = std::this_thread::get_id();
    if (!P->have_first) { P->first_thread = me; P->have_first = true; }
    else if (P->first_thread != me) P->multi = true;
  }
  while (s->inRun) wait_for_run(P, lk);
}
// run what waits until `s` has nothing staged any more (another thread's run may serve it; several runs may pass first)
static bool settle_locked(SpxPool* P, sonicStream s, std::unique_lock<std::mutex>& lk) {
  bool ok = true;
  for (;;) {
    if (s->inRun || (s->poolPending && P->running)) { wait_for_run(P, lk); continue; }
    if (!s->poolPending) return ok && !s->failed.load(std::memory_order_acquire);
    ok = pool_run(P, lk) && ok;
  }
}
// ... or everything staged by anybody (bounded staging): waits for the run in flight, then runs once
static bool run_all_locked(SpxPool* P, std::unique_lock<std::mutex>& lk) {
  while (P->running) wait_for_run(P, lk);
  return pool_run(P, lk);
}

int spx_pool_write(sonicStream s, const short* in, int sampleCount) {
  SpxPool* P = pool_of(s);
  std::unique_lock<std::mutex> lk(P->mu);
  enter(P, s, lk);
  if (spx_stream_failed(s)) return 0;
  if (s->pendingFlush && (!settle_locked(P, s, lk) || spx_stream_failed(s))) return 0;   // a write behind a staged flush is the next job
  if (s->nIn + sampleCount + s->tsmShift >= (1ll << 30)) {
    spx_api_error("stream longer than 2^30 frames is not supported");
    return 0;
  }
  const size_t bytes = sizeof(short) * (size_t)sampleCount * s->channels;
  const size_t off = (P->hInUsed + 15) & ~(size_t)15;
  PinBuf& hIn = P->hInBuf[P->hCur];
  if (!hIn.reserve(off + bytes + 32, P->hInUsed)) { spx_api_error("pinned staging allocation failed"); return 0; }
  memcpy(hIn.p + off, in, bytes);   // the caller's buffer is free again when this call returns
  P->hInUsed = off + bytes;
  s->segs.push_back({s->nIn, off, (int64_t)sampleCount});
  s->nIn += sampleCount;
  P->waitingSegs++;
  enlist(P, s);
  // bounded staging: a caller that only ever writes still makes progress (and the device buffers keep sliding)
  static const size_t limit = [] { const char* e = getenv("SPX_POOL_STAGE_BYTES"); return e ? (size_t)atoll(e) : (size_t)8 << 20; }();
  if (P->hInUsed > limit || P->waitingSegs > 8192) return run_all_locked(P, lk) && !spx_stream_failed(s) ? 1 : 0;
  return 1;
}

int spx_pool_flush(sonicStream s) {
  SpxPool* P = pool_of(s);
  std::unique_lock<std::mutex> lk(P->mu);
  enter(P, s, lk);
  if (spx_stream_failed(s)) return 0;
  if (s->pendingFlush && (!settle_locked(P, s, lk) || spx_stream_failed(s))) return 0;
  s->pendingFlush = true;
  enlist(P, s);
  return 1;
}

bool spx_pool_sync(sonicStream s) {
  SpxPool* P = pool_of(s);
  std::unique_lock<std::mutex> lk(P->mu);
  enter(P, s, lk);
  if (!s->poolPending) return true;
  return settle_locked(P, s, lk);
}

int spx_pool_read(sonicStream s, short* out, int bufferSize) {
  SpxPool* P = pool_of(s);
  std::unique_lock<std::mutex> lk(P->mu);
  enter(P, s, lk);
  if (s->poolPending) (void)settle_locked(P, s, lk);
  (void)spx_stream_failed(s);   // (what the host holds is still delivered; the reason is this thread's last error)
  int64_t n = s->outKnown - s->outRead;
  if (n <= 0 || bufferSize <= 0) return 0;
  if (n > bufferSize) n = bufferSize;
  const int64_t held = (int64_t)((s->hostOut.size() - std::min(s->hostHead, s->hostOut.size())) / (size_t)s->channels);
  if (n > held) n = held;   // (a failed handle's counts may be ahead of what the host holds)
  if (n <= 0) return 0;
  const size_t cnt = (size_t)n * s->channels;
  memcpy(out, s->hostOut.data() + s->hostHead, cnt * sizeof(short));
  s->hostHead += cnt;
  s->outRead += n;
  if (s->hostHead == s->hostOut.size()) { s->hostOut.clear(); s->hostHead = 0; }
  else if (s->hostHead > (1u << 16) && s->hostHead > s->hostOut.size() / 2) {
    s->hostOut.erase(s->hostOut.begin(), s->hostOut.begin() + (ptrdiff_t)s->hostHead);
    s->hostHead = 0;
  }
  return (int)n;
}

// The stream gets a launch sequence of its own from here on (sonic2_api.hip): what waits is run, the frame records move
// from the arena into allocations of the handle, the host-side copy of the unread output is dropped (the device still
// has those frames: a pooled handle's output window starts at its read position like an eager one's).
bool spx_pool_leave(sonicStream s) {
  SpxPool* P = pool_of(s);
  std::unique_lock<std::mutex> lk(P->mu);
  enter(P, s, lk);
  (void)hipSetDevice(P->device);
  if (s->poolPending && !settle_locked(P, s, lk)) return false;
  while (P->running) wait_for_run(P, lk);   // (the copies below use the pool's stream and the arena: not beside a run)
  if (s->arenaStart >= 0) {
    SpxFrameRec* nr = nullptr;
    float* ns = nullptr;
    const int64_t cap = s->arenaCap;
    if (hipMallocAsync(reinterpret_cast<void**>(&nr), sizeof(SpxFrameRec) * (size_t)cap, P->hs) != hipSuccess ||
        hipMallocAsync(reinterpret_cast<void**>(&ns), sizeof(float) * 4 * (size_t)cap, P->hs) != hipSuccess ||
        hipMemcpyAsync(nr, s->dRec.p, sizeof(SpxFrameRec) * (size_t)cap, hipMemcpyDeviceToDevice, P->hs) != hipSuccess ||
        hipMemcpyAsync(ns, s->dScr.p, sizeof(float) * 4 * (size_t)cap, hipMemcpyDeviceToDevice, P->hs) != hipSuccess ||
        hipStreamSynchronize(P->hs) != hipSuccess) {
      spx_stream_fail(s, "leaving the coalesced path failed (device allocation)");
      return false;
    }
    const int64_t origin = s->dRec.origin;
    arena_release(P, s);
    s->dRec.p = nr; s->dRec.origin = origin; s->dRec.cap = cap;
    s->dScr.p = ns; s->dScr.origin = 4 * origin; s->dScr.cap = 4 * cap;
  }
  s->hostOut.clear(); s->hostOut.shrink_to_fit(); s->hostHead = 0;
  s->pooled = false;
  return !s->failed;
}

void spx_pool_forget(sonicStream s) {
  SpxPool* P = pool_of(s);
  std::unique_lock<std::mutex> lk(P->mu);
  enter(P, s, lk);
  while (P->running) wait_for_run(P, lk);   // (its blocks return to the cache and its arena range to the free lists: not beside a run's tables)
  if (s->poolPending) {
    for (size_t i = 0; i < P->waiting.size(); i++)
      if (P->waiting[i] == s) { P->waiting.erase(P->waiting.begin() + (ptrdiff_t)i); break; }
    P->waitingSegs -= std::min(P->waitingSegs, s->segs.size());
    s->poolPending = false;
    s->segs.clear();
    if (P->waiting.empty()) { P->hInUsed = 0; P->waitingSegs = 0; }
  }
  arena_release(P, s);
  for (SlideBuf<int16_t>* b : {&s->dIn, &s->dOut})   // (nothing of a pooled handle is in flight between runs)
    if (b->block) { block_free_locked(P, b->p - b->guard, b->block); b->p = nullptr; b->cap = 0; b->block = 0; }
  s->pooled = false;
}

extern "C" {
// Coalesced execution of plain handles (default on; SPX_NO_POOL=1 in the environment switches it off): applies to handles
// created afterwards.
void speedyHipSetCoalescing(int on) { g_coalesce.store(on ? 1 : 0); }
int speedyHipGetCoalescing(void) { return coalesce_default(); }
// Launch sequences run and jobs served by the current device's pool so far (jobs / runs = handles per launch sequence).
void speedyHipPoolStats(unsigned long long* runs, unsigned long long* jobs) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  SpxPool* P = (dev >= 0 && dev < 64) ? g_pools[dev] : nullptr;
  if (runs) *runs = P ? P->runs : 0;
  if (jobs) *jobs = P ? P->jobs : 0;
  if (P && getenv("SPX_POOL_TIMES") && P->runs)
    fprintf(stderr, "[spx pool] %llu runs, host us per run: prepare %.1f, tables %.1f, launches %.1f, wait %.1f, post %.1f\n", P->runs,
            1e6 * P->t_prep / P->runs, 1e6 * P->t_tab / P->runs, 1e6 * P->t_launch / P->runs, 1e6 * P->t_wait / P->runs,
            1e6 * P->t_post / P->runs);
  if (P && getenv("SPX_POOL_TIMES") && P->runs)
    fprintf(stderr, "[spx pool]   launches: stage %.1f, analysis %.1f, tension %.1f, walk %.1f, gather %.1f us\n", 1e6 * P->t_l[0] / P->runs,
            1e6 * P->t_l[1] / P->runs, 1e6 * P->t_l[2] / P->runs, 1e6 * P->t_l[3] / P->runs, 1e6 * P->t_l[4] / P->runs);
}
}
