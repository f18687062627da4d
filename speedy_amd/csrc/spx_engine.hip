// C-ABI of the batch engine (include/speedy_hip.h): plan tables, workspace layout, kernel launches.
#include <algorithm>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/speedy_hip.h"
#include "spx_internal.h"
#include "spx_mode.h"
#include "spx_twiddle.h"
#include "spx_twiddle_hashes.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and a queue runs its kernels
// in order.  This library alone holds up to five streams per device that must not block one another (the caller's or the
// pipeline object's run stream, the side stream, two walk streams, the pipeline's copy stream): with four queues two of them
// share one, which two depends on the order of creation, and the end-to-end rate then read 2.0, 2.6 or 4.3 ms per batch in round 4
// (profiles/r04/r05u_hw_queues.txt).  The runtime reads the variable at its first call, so the library asks for eight queues when
// it is LOADED -- unless the application has set the variable itself (never overridden), or says SPX_KEEP_HW_QUEUES=1.  A process
// that has made HIP calls before loading the library keeps what it had: INTEGRATION.md "Environment".
__attribute__((constructor(101))) static void spx_default_hw_queues() {
  if (!getenv("SPX_KEEP_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// roctx ranges around the host side of a batch call (rocprofv3 --marker-trace shows them next to the kernels).  The
// tracer library is looked up at run time: the product has no link-time dependency on it and works without it.
#include <dlfcn.h>
struct SpxRoctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  SpxRoctx() {
    // rocprofv3 follows the rocprofiler-sdk flavour of the API; the roctracer one (libroctx64) is the fallback
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_LOCAL);
    if (h) {
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
  }
};
struct SpxRange {
  static SpxRoctx& api() { static SpxRoctx a; return a; }
  explicit SpxRange(const char* name) { if (api().push) api().push(name); }
  ~SpxRange() { if (api().pop) api().pop(); }
};

#define SPX_MAX_CHUNKS 16
// The plan's ring of earlier calls (ring_wait below) and the library's walk streams per device (dev_walk_streams): the walk kernels
// of consecutive pipelined calls take turns on up to SPX_MAX_WALK_STREAMS streams, and the ring remembers twice as many calls.
#define SPX_MAX_WALK_STREAMS 4
#define SPX_RING (2 * SPX_MAX_WALK_STREAMS)
// Pinned staging slot for the small host tables of a call (job tables, tile order): the async copies read it after the
// call has returned, so it is plan-owned and reused only once its copies have retired.
struct SpxStage {
  void* p = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
};
struct spx_plan {
  SpxPlanDev dev;
  int device = 0;           // the HIP device the tables live on; calls must be made with it current
  int cu_count = 1;
  size_t lds_per_cu = 65536;
  std::mutex mu;            // one launch sequence at a time per plan: side streams, events and staging are plan-owned
  SpxStage stage[2];
  int stage_next = 0;
  // Mode trial for batch shapes where the register file admits only ONE analysis wave per SIMD beside the consumers:
  // whether the concurrent mode pays then depends on how long the walk is (16 kHz stereo: 3.4 ms concurrent, 3.8 in
  // sequence; 22.05 kHz mono: 3.0-4.6 against 2.3-3.2), so the second call of a shape runs concurrently and the third in
  // sequence, both bracketed by events on the caller's stream, and later calls take the faster.  Results do not depend on it.
  struct Trial {
    SpxModeTrial state = {-1, 0, -1};   // key, calls, choice (-1 undecided, 0 sequential, 1 concurrent): spx_mode.h
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // [mode][begin / end]
  } trial;
  std::map<long long, SpxModeResources> res_cache;   // mode_resources: per batch shape
  // spx_batch_run of more streams than CUs, split into overlapping sub-batches (run_split): the event its sub-batches' producers
  // wait for (two, taking turns), and how the call that last used a workspace was split (spx_batch_read_steps must find the states)
  hipEvent_t ev_split[2] = {nullptr, nullptr};
  unsigned split_calls = 0;
  std::map<const void*, int> split_of;
  void* tables = nullptr;  // one device allocation behind dev.tw/tw2/window/taper*
  // time-chunk pipelining of one batch call: the analysis of chunk c+1 runs on `side` while the walk of chunk c
  // runs on the caller's stream
  hipStream_t side = nullptr;
  hipStream_t side2 = nullptr;   // concurrent mode: the tension kernel's stream
  hipEvent_t ev_tension = nullptr;
  hipEvent_t ev_start = nullptr;
  hipEvent_t ev_chunk[SPX_MAX_CHUNKS] = {};
  // spx_batch_run_ahead: the walk kernels of the plan's previous FOUR calls (ring_wait / ring_record below), the started-counter
  // of the last one
  hipEvent_t ev_walk[SPX_RING] = {};   // slot = call number mod SPX_RING
  bool ev_walk_valid[SPX_RING] = {};
  void* ring_ws[SPX_RING] = {};         // the workspace and the caller's stream of the call in the slot
  hipStream_t ring_st[SPX_RING] = {};
  int ahead_calls = 0;
  hipEvent_t ev_call[2] = {nullptr, nullptr};   // the caller's stream as it stood when the last two pipelined calls were made
  const void* ahead_last_out = nullptr;
  const void* ahead_last_nout = nullptr;
  bool ev_call_valid[2] = {false, false};
  hipStream_t ev_call_st[2] = {nullptr, nullptr};   // ... and which stream each note was taken on (a detached call leaves none)
  std::vector<std::pair<const int*, int>> mixed_started;   // the same for the groups of the last mixed call (lead plan)
  const int* ahead_started = nullptr;
  int ahead_n = 0;
  // spx_batch_run_mixed: this plan's group runs on `mix`; the first plan of a call also lends the fork event and a staging slot
  hipStream_t mix = nullptr;
  hipEvent_t ev_join = nullptr, ev_fork = nullptr, ev_an = nullptr;
  std::mutex mix_mu;
  SpxStage mix_stage;
};

// Timing: one set of four HIP events per timed call, recorded on the launch stream and resolved lazily by
// spx_timing_collect (so that the timed region itself carries no host synchronisation).
// Process-wide switches are atomics; the event lists behind spx_timing_collect are guarded by g_tmu.  spx_batch_run may be
// called from several host threads (one plan per thread, or one plan shared: launches on a plan are serialised by its mutex).
static std::atomic<bool> g_timing{false};
static std::atomic<int> g_last_concurrent{0};   // spx_debug_last_call_concurrent (2 = pipelined with the previous call)
static std::atomic<int> g_concurrent{1};  // spx_set_concurrent: analysis and walk kernels on two streams, tile-flag hand-off
static std::atomic<bool> g_chunks_set{false};  // the caller chose a chunk count (spx_set_pipeline_chunks)
static std::atomic<int> g_chunks{1};  // spx_set_pipeline_chunks (measured on MI355X, 256 x 10 s: 1 -> 5.09 ms, 2 -> 5.25, 4 -> 5.54 per step)
struct EvPair { hipEvent_t a, b; int kind; };  // kind 0 = analysis launch, 1 = walk launch, 2 = tension launch
static std::mutex g_tmu;
static std::vector<EvPair> g_ev_pending;
static std::vector<hipEvent_t> g_ev_free;
static int g_calls_pending = 0;

// Concurrent mode keeps polling workgroups resident; its deadlock-freedom bound (run_impl) counts the streams of ONE
// call.  A second concurrent-mode call in flight on the same device (another plan, thread or stream) would break it, so
// per device the last concurrent call leaves an event behind, and a call that finds it unfinished on a different stream
// takes the sequential launch order instead (same results).  Calls on the same stream are ordered by the stream.
struct SpxDevGuard {
  std::mutex mu;
  hipEvent_t last = nullptr;
  hipStream_t last_stream = nullptr;
  bool valid = false;
};
static SpxDevGuard g_guard[64];

// Allocated VGPRs of a kernel (hipFuncGetAttributes, rounded up to the granule of 8), cached per function: the query is not
// free and the engine asks on every call.  Several plans may ask from several threads at once.
int spx_kernel_vgprs(const void* fn, int* scratch_bytes) {
  static std::mutex mu;
  static std::map<const void*, std::pair<int, int>> cache;
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(fn);
  if (it == cache.end()) {
    hipFuncAttributes a;
    int regs = 128, scratch = -1;
    if (hipFuncGetAttributes(&a, fn) == hipSuccess) { regs = (a.numRegs + 7) & ~7; scratch = (int)a.localSizeBytes; }
    (void)hipGetLastError();
    it = cache.emplace(fn, std::make_pair(regs, scratch)).first;
  }
  if (scratch_bytes) *scratch_bytes = it->second.second;
  return it->second.first;
}

// Cross-process half of SpxDevGuard: the concurrent mode's deadlock-freedom bound counts the polling workgroups of ONE
// call, and SpxDevGuard enforces "one such call at a time" inside a process only.  Two processes sharing a GPU (several
// ranks on one device) could both keep polling workgroups resident and close every CU to both analysis kernels.  So the
// first process that wants the mode on a device takes an exclusive advisory lock on a per-device file and keeps it for
// its lifetime; a process that finds the lock taken (or cannot create the file, or is told SPX_SHARED_GPU=1) runs its
// kernels in sequence on that device -- same results.
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
static bool device_is_ours(int dev) {
  static std::mutex mu;
  static int state[64];   // 0 unknown, 1 ours, -1 shared
  static int lock_fd[64];
  static pid_t owner = 0;
  if (dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> g(mu);
  if (owner != getpid()) {
    // first call, or a fork()ed child: the child shares the parent's open file description and with it the parent's lock --
    // both would believe the device is theirs.  The child drops the inherited descriptors (the parent keeps the lock) and
    // asks again for itself.
    for (int d = 0; d < 64; d++) { if (owner && state[d] > 0 && lock_fd[d] > 0) close(lock_fd[d]); state[d] = 0; lock_fd[d] = -1; }
    owner = getpid();
  }
  if (state[dev]) return state[dev] > 0;
  state[dev] = -1;
  if (getenv("SPX_SHARED_GPU")) return false;
  char bus[64] = "dev";
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof(bus), "ordinal%d", dev); }
  for (char* c = bus; *c; c++) if (*c == ':' || *c == '/') *c = '_';
  // The lock must be seen by every process that can use the device, so it lives in a shared directory (SPX_LOCK_DIR, else
  // /tmp).  O_NOFOLLOW: a symbolic link planted under the name is refused, never followed; the file is created 0644 and never
  // written -- another user's file is opened read-only (an advisory lock needs no write access).  Anyone who can open the
  // file can hold the lock and thereby send other processes to the sequential launch order: that costs them speed, never
  // results (SPX_DEBUG_MODE=1 prints the mode of every call).
  const char* dir = getenv("SPX_LOCK_DIR");
  if (!dir) dir = "/tmp";
  char path[256];
  snprintf(path, sizeof(path), "%s/spx_concurrent_%s.lock", dir, bus);
  int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0644);
  if (fd < 0) fd = open(path, O_RDONLY | O_CLOEXEC | O_NOFOLLOW);
  if (fd < 0) return false;
  if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); return false; }
  state[dev] = 1;   // (the descriptor stays open: the lock lives as long as the process)
  lock_fd[dev] = fd;
  return true;
}

// Upper bound on the frames a stream can produce from n_in input frames (flush padding included here).
// speed >= 1: the stage never emits more than it consumes (the nonlinear speed stays >= 1, speedy.c:772).
// speed < 1: one pitch step at speed s emits at most 2/s frames per frame consumed -- for s < 0.5 it emits
// period + n and consumes n = (int)(period*s/(1-s)) >= 1, and period/n <= 2(1-s)/s because floor(x) >= x/2 for
// x >= 1; for 0.5 <= s < 1 the ratio is (2*period + r)/(period + r) <= 2.  The nonlinear speed can sit at the
// kMinimumSpeed clamp 0.01 (speedy.c:92,776) whatever the requested speed.
int64_t spx_internal_out_bound(const SpxPlanDev& P, int64_t n_in, float speed, bool nonlinear) {
  const int64_t slack = 4 * (int64_t)P.maxRequired + 1024;
  if (speed >= 1.0f) return n_in + slack;
  double s = nonlinear ? 0.01 : (double)speed;
  if (s < 1e-4) s = 1e-4;
  return (int64_t)((double)(n_in + 2 * (int64_t)P.maxRequired) * (2.0 / s)) + slack;
}

static int dev_walk_streams(int dev, hipStream_t* w, int n);   // (defined with the launch code below)
static int walk_stream_count();
void spx_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }   // other translation units' errors reach spx_last_error
extern "C" {

const char* spx_last_error(void) { return g_err.c_str(); }
// (host-only diagnostics, no GPU needed: tests/test_oracle_twiddle.py compares the library's twiddle routine with the oracle's and
// with a 60-digit evaluation entry by entry)
void spx_debug_twiddle_entry(long k, long n, double* c, double* s) { spx_tw::sincos_2pi(k, n, c, s); }
unsigned long long spx_debug_twiddle_hash(long den, long count) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (long t = 0; t < count; t++) {
    double e[2];
    spx_tw::entry(t, den, e);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(e);
    for (int i = 0; i < 16; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
  }
  return h;
}
int spx_abi_version(void) { return 1; }

static int factor_radices(int n, int* radix) {  // DESIGN.md "DFT spec": 4s, then 2, 3s, 5s, other primes ascending
  int ns = 0;
  while (n % 4 == 0) { radix[ns++] = 4; n /= 4; }
  while (n % 2 == 0) { radix[ns++] = 2; n /= 2; }
  while (n % 3 == 0) { radix[ns++] = 3; n /= 3; }
  while (n % 5 == 0) { radix[ns++] = 5; n /= 5; }
  for (int p = 7; n > 1; p += 2)
    while (n % p == 0) { radix[ns++] = p; n /= p; }
  return ns;
}

spx_plan_t spx_plan_create(int sample_rate, int match_matlab) {
  if (sample_rate < 1000 || sample_rate > 127999) {  // the walk kernel holds <= 256 lags per search
    fail(-1, "spx_plan_create: unsupported sample rate");
    return nullptr;
  }
  spx_plan* p = new spx_plan();
  SpxPlanDev& d = p->dev;
  memset(&d, 0, sizeof(d));
  {
    hipDeviceProp_t prop;
    if (hipGetDevice(&p->device) == hipSuccess && hipGetDeviceProperties(&prop, p->device) == hipSuccess) {
      p->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
      p->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? (size_t)prop.maxSharedMemoryPerMultiProcessor : 65536;
    }
  }
  d.rate = sample_rate;
  d.B = (int)(sample_rate / 100.0);                    // speedy.c:335-338
  d.W = (int)(1.5 * sample_rate / (float)100.0);       // speedy.c:213
  d.N = 2 * d.W;
  d.F = match_matlab ? 8 : 12;                         // speedy.h:136-146
  d.Pp = match_matlab ? 12 : 8;
  d.nstages = (d.W > 1) ? factor_radices(d.W, d.radix) : 0;
  if (d.nstages > SPX_MAX_STAGES) {
    delete p;
    fail(-1, "spx_plan_create: too many DFT stages");
    return nullptr;
  }
  // (a window too large even for that leaves the plan to linear jobs -- the TSM stage alone, spx_internal_analysis_fits --
  // and a nonlinear job on it is refused; no rate below 128 kHz is)
  d.minPeriod = sample_rate / 400;
  d.maxPeriod = sample_rate / 65;
  d.maxRequired = 2 * d.maxPeriod;
  d.skip = sample_rate > 4000 ? sample_rate / 4000 : 1;
  d.alpha = (float)exp(-1.0 / (float)100.0);          // speedy.c:67 with time constant kFrameRateHz
  d.one_minus_alpha = 1 - d.alpha;                     // float, speedy.c:74

  const int W = d.W;
  // Rader applies to a prime W > 64 whose W - 1 has no prime factor above 13 (DESIGN.md "DFT spec")
  const int M = W - 1;
  bool rader = W > 64;
  for (int q = 2; rader && (long)q * q <= W; q++) if (W % q == 0) rader = false;
  if (rader) {
    int m = M;
    for (int q = 2; q <= 13; q++) while (m % q == 0) m /= q;
    rader = (m == 1);
  }
  if (rader) {
    d.rader = 1;
    d.nstagesM = factor_radices(M, d.radixM);
  }
  // The analysis tile must fit one CU's LDS: 16 frames and four transforming waves up to about 49 kHz, 8 frames up to about
  // 61 kHz; above that (round 3) fewer waves transform -- their fp64 work areas are what grows -- and the tile shrinks to 4
  // frames (its rows of log terms grow too): 2 waves up to about 100 kHz, 1 wave up to the 128 kHz the walk kernel takes.
  d.dft_waves = 4;
  d.tile_frames = spx_analysis_tile_frames();
  const int cand[5][2] = {{spx_analysis_tile_frames(), 4}, {spx_analysis_small_tile_frames(), 4}, {spx_analysis_small_tile_frames(), 2},
                          {spx_analysis_tiny_tile_frames(), 2}, {spx_analysis_tiny_tile_frames(), 1}};
  for (int c = 0; c < 5; c++) {
    d.tile_frames = cand[c][0];
    d.dft_waves = cand[c][1];
    if (spx_analysis_lds_bytes(d) <= 160 * 1024) break;
  }
  if (spx_analysis_prefers_small_tile(d)) {  // 44.1 / 48 kHz: the compiled-in kernels, two 8-frame workgroups per CU
    d.tile_frames = spx_analysis_small_tile_frames();
    d.dft_waves = 4;
  }
  const size_t n_tw = 2 * (size_t)W, n_win = (size_t)W, n_tf = d.F + 1, n_tp = d.Pp + 1;
  const size_t n_rd = rader ? 2 * (size_t)M : 0;  // doubles in each of twM and bfft
  const size_t n_ri = rader ? (size_t)M : 0;      // ints in each of perm and iperm
  const size_t n_ql = rader ? (size_t)W : 0;      // ints in qlog
  const size_t bytes = sizeof(double) * (2 * n_tw + 2 * n_rd) + sizeof(float) * (n_win + n_tf + n_tp + 8) +
                       sizeof(int) * (2 * n_ri + n_ql);
  std::vector<unsigned char> host(bytes, 0);
  double* tw = reinterpret_cast<double*>(host.data());
  double* tw2 = tw + n_tw;
  double* twM = tw2 + n_tw;
  double* bfft = twM + n_rd;
  float* win = reinterpret_cast<float*>(bfft + n_rd);
  float* tf = win + n_win;
  float* tp = tf + n_tf;
  int* perm = reinterpret_cast<int*>(tp + n_tp + 8);
  int* iperm = perm + n_ri;
  int* qlog = iperm + n_ri;
  // A twiddle factor (cos, -sin)(2 pi t / den) comes from spx_twiddle.h: IEEE double operations on the integers (t, den), no libm call --
  // the same bits on every machine (round 6; rounds 1-5 took the box's libm, round 5 "one sincos call", so GPU == oracle held on any
  // one box only).  The Hamming window's cosine likewise (speedy.c:256-258: a double expression stored as float).
  for (int t = 0; t < W; t++) {
    spx_tw::entry(t, W, &tw[2 * t]);
    spx_tw::entry(t, 2L * W, &tw2[2 * t]);
    double c = 1.0, sn = 0.0;
    if (W > 1) spx_tw::sincos_2pi(t, W - 1, &c, &sn);
    win[t] = 0.54 - 0.46 * c;  // speedy.c:256-258
  }
  if (rader) {
    for (int t = 0; t < M; t++) spx_tw::entry(t, M, &twM[2 * t]);
    int g = 2;  // smallest primitive root of W
    for (; g < W; g++) {
      long v = 1;
      int k = 0;
      do { v = (v * g) % W; k++; } while (v != 1);
      if (k == M) break;
    }
    long v = 1;
    for (int k = 0; k < M; k++) { perm[k] = (int)v; v = (v * g) % W; }
    for (int q = 0; q < M; q++) iperm[q] = perm[(M - q) % M];  // g^-q = g^(M-q)
    for (int q = 0; q < M; q++) qlog[iperm[q]] = q;
    std::vector<double> b(2 * (size_t)M);
    for (int q = 0; q < M; q++) { b[2 * q] = tw[2 * iperm[q]]; b[2 * q + 1] = tw[2 * iperm[q] + 1]; }
    spx_host_dft(M, d.radixM, d.nstagesM, twM, b.data(), bfft);
  }
  // the tables of the compiled-in window sizes are pinned (spx_twiddle_hashes.h, generated by tools/twiddle_tables.py from a 60-digit
  // evaluation): a build whose host arithmetic strays (fast-math, a contracted multiply-add) is refused here, loudly
  {
    const struct { long den, count; const double* t; } built[3] = {{W, W, tw}, {2L * W, W, tw2}, {rader ? M : 0, rader ? M : 0, twM}};
    for (const auto& b : built)
      for (const auto& pin : spx_twiddle_pins)
        if (b.count > 0 && pin.den == b.den && pin.count == b.count && spx_tw::fnv1a(b.t, 16 * (size_t)b.count) != pin.hash) {
          delete p;
          fail(-1, "spx_plan_create: a twiddle table does not hash to its pinned value (spx_twiddle_hashes.h) -- host code built with fast-math or fp contraction?");
          return nullptr;
        }
  }
  for (int i = 0; i <= d.F; i++) tf[i] = (d.F - i) / (float)d.F;    // speedy.c:597
  for (int i = 0; i <= d.Pp; i++) tp[i] = (d.Pp - i) / (float)d.Pp;  // speedy.c:604
  {
    // The library's own streams of this device -- the side stream, the two walk streams -- are created NOW, before anything the
    // process creates later (a pipeline object's run and copy streams, the caller's own): HIP maps streams onto hardware queues
    // and pipes in creation order, and with the pipeline's two streams created FIRST its resident loop read 1.15 instead of 1.03 ms
    // per batch (profiles/r05/r5c_order_probe.txt).
    hipStream_t w[SPX_MAX_WALK_STREAMS];
    (void)dev_walk_streams(p->device, w, walk_stream_count());
    (void)hipGetLastError();
  }
  if (hipMalloc(&p->tables, bytes) != hipSuccess ||
      hipMemcpy(p->tables, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
    fail(-2, "spx_plan_create: device allocation/copy failed (is a GPU visible?)");
    if (p->tables) (void)hipFree(p->tables);
    delete p;
    return nullptr;
  }
  unsigned char* base = static_cast<unsigned char*>(p->tables);
  d.tw = reinterpret_cast<const double*>(base);
  d.tw2 = d.tw + n_tw;
  d.twM = d.tw2 + n_tw;
  d.bfft = d.twM + n_rd;
  d.window = reinterpret_cast<const float*>(d.bfft + n_rd);
  d.taperF = d.window + n_win;
  d.taperP = d.taperF + n_tf;
  d.perm = reinterpret_cast<const int*>(d.taperP + n_tp + 8);
  d.iperm = d.perm + n_ri;
  d.qlog = d.iperm + n_ri;
  return p;
}

void spx_plan_destroy(spx_plan_t plan) {
  if (!plan) return;
  // (side / side2 belong to the device, not to the plan: dev_side_streams)
  if (plan->side) (void)hipStreamSynchronize(plan->side);
  if (plan->side2) (void)hipStreamSynchronize(plan->side2);
  if (plan->mix) { (void)hipStreamSynchronize(plan->mix); (void)hipStreamDestroy(plan->mix); }
  if (plan->ev_join) (void)hipEventDestroy(plan->ev_join);
  if (plan->ev_an) (void)hipEventDestroy(plan->ev_an);
  if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
  if (plan->mix_stage.done) { (void)hipEventSynchronize(plan->mix_stage.done); (void)hipEventDestroy(plan->mix_stage.done); }
  if (plan->mix_stage.p) (void)hipHostFree(plan->mix_stage.p);
  if (plan->ev_tension) (void)hipEventDestroy(plan->ev_tension);
  if (plan->ev_start) (void)hipEventDestroy(plan->ev_start);
  for (auto& e : plan->ev_chunk) if (e) (void)hipEventDestroy(e);
  for (auto& e : plan->trial.ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : plan->ev_walk) if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
  for (auto& e : plan->ev_call) if (e) (void)hipEventDestroy(e);
  for (auto& e : plan->ev_split) if (e) (void)hipEventDestroy(e);
  for (auto& g : plan->stage) {
    if (g.done) { (void)hipEventSynchronize(g.done); (void)hipEventDestroy(g.done); }
    if (g.p) (void)hipHostFree(g.p);
  }
  if (plan->tables) (void)hipFree(plan->tables);
  delete plan;
}
int spx_plan_frame_step(spx_plan_t p) { return p->dev.B; }
int spx_plan_window_size(spx_plan_t p) { return p->dev.W; }
int spx_plan_fft_size(spx_plan_t p) { return p->dev.N; }
int spx_plan_future(spx_plan_t p) { return p->dev.F; }
int spx_plan_max_required(spx_plan_t p) { return p->dev.maxRequired; }

static int64_t frames_for(const SpxPlanDev& d, int64_t n_in) {
  // frame j is sent to the analysis once sample j*B + W has been written (soniclib.c:440-444)
  if (n_in < d.W + 1) return 0;
  return (n_in - d.W - 1) / d.B + 1;
}
int64_t spx_plan_frames(spx_plan_t p, int64_t n_in) { return frames_for(p->dev, n_in); }

int64_t spx_plan_out_capacity_for(spx_plan_t p, int64_t n_in, float speed, float nonlinear) {
  return spx_internal_out_bound(p->dev, n_in, speed, nonlinear != 0.0f);
}
int64_t spx_plan_out_capacity(spx_plan_t p, int64_t n_in, float speed) {
  return spx_internal_out_bound(p->dev, n_in, speed, true);
}

}  // extern "C"
bool spx_internal_analysis_fits(const SpxPlanDev& d) { return spx_analysis_lds_bytes(d) <= 160 * 1024; }

static spx_plan* shared_plan_full(int sample_rate, int match_matlab) {
  static std::mutex mu;
  static std::map<std::pair<int, std::pair<int, int>>, spx_plan*> cache;  // (device, (rate, mode)): tables are per device
  std::lock_guard<std::mutex> g(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto key = std::make_pair(dev, std::make_pair(sample_rate, match_matlab ? 1 : 0));
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  spx_plan* p = spx_plan_create(sample_rate, match_matlab);
  if (!p) return nullptr;
  cache[key] = p;
  return p;
}
const SpxPlanDev* spx_internal_shared_plan(int sample_rate, int match_matlab) {
  spx_plan* p = shared_plan_full(sample_rate, match_matlab);
  return p ? &p->dev : nullptr;
}
int64_t spx_internal_frames_for(const SpxPlanDev& d, int64_t n_in) { return frames_for(d, n_in); }
extern "C" {

struct Layout {
  size_t off_streams, off_states, off_rec, off_scratch, off_order, off_flags, off_ready, total;
  int64_t max_tiles;
  int64_t total_frames;
};
static Layout layout_for(const SpxPlanDev& d, const spx_stream_job* jobs, int n) {
  Layout L;
  int64_t tf = 0;
  for (int i = 0; i < n; i++) tf += (jobs[i].nonlinear != 0.0f) ? frames_for(d, jobs[i].n_in) : 0;
  L.total_frames = tf;
  size_t o = 0;
  L.off_streams = o; o += ((sizeof(SpxStreamDev) * (size_t)n * SPX_MAX_CHUNKS + 255) & ~(size_t)255);
  L.off_states = o;  o += ((sizeof(SpxStreamState) * (size_t)n + 255) & ~(size_t)255);
  L.off_rec = o;     o += ((sizeof(SpxFrameRec) * (size_t)(tf + 1) + 255) & ~(size_t)255);
  L.off_scratch = o; o += ((sizeof(float) * 4 * (size_t)(tf + 1) + 255) & ~(size_t)255);
  const int TFr = std::min(spx_analysis_small_tile_frames(), d.tile_frames > 0 ? d.tile_frames : spx_analysis_small_tile_frames());  // the smallest tile a call may take: an upper bound on tiles
  L.max_tiles = tf / TFr + n + 1;  // every stream may end with a partial tile
  L.off_order = o;   o += ((sizeof(int) * (size_t)L.max_tiles + 255) & ~(size_t)255);
  L.off_flags = o;   o += ((sizeof(int) * (size_t)L.max_tiles + 255) & ~(size_t)255);
  L.off_ready = o;   o += ((sizeof(int) * ((size_t)n + 1) + 255) & ~(size_t)255);   // + the count of walk workgroups that have started
  L.total = o;
  return L;
}


// Job tables for `nch` consecutive time chunks of every stream: chunk c covers the input up to n_c frames
// (n_c = n_in for the last chunk), starts where chunk c-1 stopped (frame_begin) and carries the state record.
static int build_streams(const SpxPlanDev& d, const spx_stream_job* jobs, int n, int nch,
                         std::vector<SpxStreamDev>& v, std::vector<int>& tiles_per_chunk) {
  v.resize((size_t)n * nch);
  tiles_per_chunk.assign(nch, 0);
  int64_t fo = 0;
  const int TF = d.tile_frames;
  for (int i = 0; i < n; i++) {
    const spx_stream_job& j = jobs[i];
    if (j.channels < 1 || j.n_in < 0 || j.in_off < 0 || j.out_off < 0 || j.out_cap < 0)
      return fail(-1, "spx_batch: bad job (channels < 1 or a negative count / offset)");
    // The reference takes any float here and has no defined behaviour for most of them (a speed <= 0 makes the TSM
    // stage's step counts negative).  A job is refused unless every speed the TSM stage can be given is positive:
    if (!(j.speed > 0.0f) || !std::isfinite(j.speed)) return fail(-1, "spx_batch: speed must be finite and > 0");
    if (!(j.nonlinear >= 0.0f && j.nonlinear <= 1.0f))
      return fail(-1, "spx_batch: nonlinear factor outside [0, 1] (sonic2.h:73-76; the blended speed could reach 0)");
    if (!std::isfinite(j.feedback)) return fail(-1, "spx_batch: feedback strength is not finite");
    if (j.nonlinear != 0.0f && !spx_internal_analysis_fits(d))
      return fail(-1, "spx_batch: sample rate too high for the nonlinear path (the analysis tile does not fit one CU's LDS); linear jobs only");
    if (j.n_in >= (1ll << 30)) return fail(-1, "spx_batch: stream of 2^30 frames or more (in-kernel positions are 32-bit)");
    const bool nonlinear = j.nonlinear != 0.0f;
    const int64_t Ttot = nonlinear ? frames_for(d, j.n_in) : 0;
    if (Ttot > 0x7fffff00) return fail(-1, "spx_batch: stream too long");
    int64_t Tprev = 0;
    for (int c = 0; c < nch; c++) {
      int64_t n_c = j.n_in;
      if (c < nch - 1) n_c = (j.n_in * (c + 1) / nch) / d.B * d.B;
#ifdef SPX_TUNING
      static const int frac = [] { const char* e = getenv("SPX_CHUNK_FRAC"); return e ? atoi(e) : 0; }();   // A/B: the first of two chunks, percent
      if (frac > 0 && nch == 2 && c == 0) n_c = (j.n_in * frac / 100) / d.B * d.B;
#endif
      SpxStreamDev& s = v[(size_t)c * n + i];
      s.in_off = j.in_off; s.n_in = n_c; s.out_off = j.out_off; s.out_cap = j.out_cap;
      s.channels = j.channels; s.speed = j.speed; s.nonlinear = j.nonlinear; s.feedback = j.feedback;
      const int64_t T = nonlinear ? frames_for(d, n_c) : 0;
      s.n_frames = (int32_t)T;
      s.frame_begin = (int32_t)Tprev;
      s.flags = (c == 0 ? SPX_F_INIT : 0) | (c == nch - 1 ? SPX_F_FLUSH : 0);
      s.frame_off = fo;
      s.first_tile = tiles_per_chunk[c];
      tiles_per_chunk[c] += (int)((T - Tprev + TF - 1) / TF);
      Tprev = T;
    }
    fo += Ttot;
  }
  return 0;
}

static SpxTapsDev taps_of(const spx_taps* t) {
  SpxTapsDev d = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (t) { d.tension = t->tension; d.speed = t->speed; d.features = t->features;
           d.spectrogram = t->spectrogram; d.normalized = t->normalized; }
  return d;
}

}  // extern "C"
// Idle-start gate (run_impl): holds the analysis stream back until every workgroup of the call's walk kernel has been
// placed -- each announces itself in started[0] (= speed_ready[n_streams]) -- so the wait ends with the event it is
// for, not after a tuned delay.  Bounded (about 0.3 ms): should the two streams share a hardware queue, the walk launch
// sits behind this kernel and cannot start; the gate then gives up and the call is merely placed less well.
__global__ void spx_gate_kernel(const int* started, int n_walk, unsigned max_spins) {
  if (threadIdx.x == 0) {
    for (unsigned i = 0; i < max_spins; i++) {
      if (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_walk) break;
      __builtin_amdgcn_s_sleep(8);   // 8 * 64 clocks, about 0.25 us
    }
  }
}
// Job tables (and tile order) from the plan's pinned staging slot into the workspace, hand-off flags cleared.
__global__ void __launch_bounds__(256)
spx_stage_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst_a, unsigned n_a, unsigned* __restrict__ dst_b,
                 unsigned n_b, unsigned* __restrict__ zero_a, unsigned nz_a, unsigned* __restrict__ zero_b, unsigned nz_b,
                 const int* gate_started, int gate_n, unsigned gate_spins) {
  const unsigned stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (unsigned i = t0; i < n_a; i += stride) dst_a[i] = src[i];
  for (unsigned i = t0; i < n_b; i += stride) dst_b[i] = src[n_a + i];
  for (unsigned i = t0; i < nz_a; i += stride) zero_a[i] = 0u;
  for (unsigned i = t0; i < nz_b; i += stride) zero_b[i] = 0u;
  // ... and, for a pipelined call, the gate of spx_gate_kernel in the same launch (one kernel and one dispatch less on the
  // producers' stream, whose chain is as long as the walk streams' period since round 5): the counter is the PREVIOUS call's
  if (gate_started != nullptr && t0 == 0) {
    for (unsigned i = 0; i < gate_spins; i++) {
      if (__hip_atomic_load(gate_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gate_n) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
extern "C" {

static hipEvent_t take_event() {
  std::lock_guard<std::mutex> g(g_tmu);
  if (!g_ev_free.empty()) { hipEvent_t e = g_ev_free.back(); g_ev_free.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

}  // extern "C"
// The two side streams of the concurrent mode (analysis, tension) and of the time-chunk pipeline are per DEVICE, created once:
// HIP maps streams onto a few hardware queues in creation order, and a queue runs its kernels in order -- with side streams
// per plan, the second plan of a process got a tension stream that shared the caller stream's queue, its walk kernel waited
// behind its own tension kernel, and the call ran analysis and walk one after the other (3.5 instead of 2.4 ms per 256 x 10 s
// at 22.05 kHz, profiles/r03/r03aj_lean_22k.txt).  Only one concurrent-mode call is in flight per device anyway (SpxDevGuard);
// calls that pipeline time chunks from several host threads share the analysis stream.
static int dev_side_streams(int dev, hipStream_t* side, hipStream_t* side2) {
  static std::mutex mu;
  static hipStream_t s1[64], s2[64];
  const int d = (dev >= 0 && dev < 64) ? dev : 0;
  std::lock_guard<std::mutex> g(mu);
  if (!s1[d]) {
    if (hipStreamCreateWithFlags(&s1[d], hipStreamNonBlocking) != hipSuccess) { s1[d] = nullptr; return -1; }
    if (hipStreamCreateWithFlags(&s2[d], hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(s1[d]); s1[d] = nullptr; s2[d] = nullptr; return -1; }
  }
  *side = s1[d]; *side2 = s2[d];
  return 0;
}

// The walk kernels of consecutive pipelined calls take turns on streams of the library's (run_impl): the device's SECOND side
// stream -- idle in that order: the tension kernel runs behind the analysis on the first -- and up to three more.  Round 4 ran two
// (four streams per device in all, the caller's included: one per hardware queue of HIP's default four); since round 5 the library
// asks for eight hardware queues, and round 6 lets the mode decide how many walk launches are in flight (walk_stream_count).
static int walk_stream_count() {
  static const int n = [] {
    const char* e = spx_tuning_env("SPX_WALK_STREAMS");
    if (!e && spx_tuning_env("SPX_WALK_STREAMS3")) return 3;
    const int v = e ? atoi(e) : 2;
    return v < 1 ? 1 : (v > SPX_MAX_WALK_STREAMS ? SPX_MAX_WALK_STREAMS : v);
  }();
  return n;
}
static int dev_walk_streams(int dev, hipStream_t* w, int n) {
  static std::mutex mu;
  static hipStream_t s3[64][SPX_MAX_WALK_STREAMS];   // [.][0] unused: the first walk stream is the second side stream
  hipStream_t side = nullptr, side2 = nullptr;
  if (dev_side_streams(dev, &side, &side2)) return -1;
  const int d = (dev >= 0 && dev < 64) ? dev : 0;
  std::lock_guard<std::mutex> g(mu);
  w[0] = side2;
  for (int i = 1; i < n && i < SPX_MAX_WALK_STREAMS; i++) {
    if (!s3[d][i] && hipStreamCreateWithFlags(&s3[d][i], hipStreamNonBlocking) != hipSuccess) { s3[d][i] = nullptr; return -1; }
    w[i] = s3[d][i];
  }
  return 0;
}

// ---- the plan's ring of earlier calls (every call of a plan that walks -- plain, pipelined, mixed -- goes through it) ----
// ring_note: where the caller's stream stands when the call is MADE (an overlapped call orders its walk kernel behind the
// PREVIOUS call's note, whatever kind of call that was).
// ring_wait: which earlier calls a pipelined call's producers wait for (on `sa`).  The walk kernels of pipelined calls may run
// on up to SPX_MAX_WALK_STREAMS streams taking turns (spx_batch_run_overlapped), so "the call before" says nothing about the
// calls before that one.  The ring remembers SPX_RING = 2 x that many calls:
//   - the older half, always: SPX_MAX_WALK_STREAMS consecutive calls close every walk stream's history (everything older is
//     done), which also covers a caller that rotates more workspaces than the ring's younger half remembers;
//   - the younger half when they used THIS workspace (two workspaces taking turns: the call two back) or another
//     caller stream (then the order of the caller's stream says nothing about them).
// A caller that rotates three workspaces therefore gets its producers started while the walk kernels of BOTH previous calls are
// still running.  *waited_prev: the producers were made to wait for the call right before this one (nothing of it is in flight
// by the time they run: no gate for its walk kernel, see run_impl).
// ring_record: how a call leaves its end -- and its output buffers -- in the ring.
static int ring_note(spx_plan* plan, hipStream_t st) {
  const int cur = plan->ahead_calls & 1;
  if (!plan->ev_call[cur]) HIPCHK(hipEventCreateWithFlags(&plan->ev_call[cur], hipEventDisableTiming));
  HIPCHK(hipEventRecord(plan->ev_call[cur], st));
  plan->ev_call_valid[cur] = true;
  plan->ev_call_st[cur] = st;
  return 0;
}
static int ring_wait(spx_plan* plan, hipStream_t sa, const void* ws, hipStream_t st, bool* waited_prev) {
  const int c = plan->ahead_calls % SPX_RING;
  if (waited_prev) *waited_prev = false;
  for (int back = SPX_RING; back >= 1; back--) {
    const int j = (c + SPX_RING - back) % SPX_RING;          // slot of the call `back` calls ago
    if (!plan->ev_walk_valid[j]) continue;
    if (back > SPX_RING / 2 || plan->ring_ws[j] == ws || plan->ring_st[j] != st) {
      HIPCHK(hipStreamWaitEvent(sa, plan->ev_walk[j], 0));
      if (back == 1 && waited_prev) *waited_prev = true;
    }
  }
  return 0;
}
static bool ring_previous_in_flight(spx_plan* plan) {
  const int j = (plan->ahead_calls + SPX_RING - 1) % SPX_RING;
  const bool f = plan->ev_walk_valid[j] && hipEventQuery(plan->ev_walk[j]) == hipErrorNotReady;
  (void)hipGetLastError();
  return f;
}
static int ring_record(spx_plan* plan, hipStream_t on, void* ws, hipStream_t st, const void* out, const void* n_out, bool detached = false) {
  const int c = plan->ahead_calls % SPX_RING;
  if (!plan->ev_walk[c]) HIPCHK(hipEventCreateWithFlags(&plan->ev_walk[c], hipEventDisableTiming));
  HIPCHK(hipEventRecord(plan->ev_walk[c], on));
  if (on != st && !detached) HIPCHK(hipStreamWaitEvent(st, plan->ev_walk[c], 0));   // the caller's stream is done when the walk is
  plan->ev_walk_valid[c] = true;
  plan->ring_ws[c] = ws;
  plan->ring_st[c] = st;
  plan->ahead_last_out = out;
  plan->ahead_last_nout = n_out;
  plan->ahead_calls++;
  return 0;
}

// ---- the inputs of spx_choose_mode (spx_mode.h) ----
static SpxModeEnv mode_env() {
  // the developers' A/B switches exist in builds with -DSPX_TUNING only (spx_tuning_env); read once per process
  static const SpxModeEnv fixed = [] {
    SpxModeEnv e;
    memset(&e, 0, sizeof(e));
    e.serial = spx_tuning_env("SPX_SERIAL") != nullptr;             // kernels back to back on one stream
    e.no_lean = spx_tuning_env("SPX_NO_LEAN_WALK") != nullptr;
    e.small_tile = spx_tuning_env("SPX_TILE_SMALL") != nullptr;     // the 8-frame tile whenever concurrent
    e.ahead_any = spx_tuning_env("SPX_AHEAD_ANY") != nullptr;       // the pipelined order whatever the co-residency arithmetic says
    e.full_walk = spx_tuning_env("SPX_OVERLAP_FULL_WALK") != nullptr;   // overlapped calls keep the full walk form
    e.walk1 = spx_tuning_env("SPX_AHEAD_WALK1") != nullptr;         // pipelined calls' walk kernels on the caller's stream, one after the other
    e.no_excl = spx_tuning_env("SPX_NO_EXCLUSIVE_CU") != nullptr;
    e.trial_force = spx_tuning_env("SPX_TRIAL_FORCE") ? atoi(spx_tuning_env("SPX_TRIAL_FORCE")) : -1;
    return e;
  }();
  SpxModeEnv e = fixed;
  e.concurrent_enabled = g_concurrent.load() != 0;
  e.chunks_set = g_chunks_set.load();
  e.chunks = g_chunks.load();
  return e;
}
static bool device_ours_cb(void* ctx) { return device_is_ours(*static_cast<int*>(ctx)); }
// What kind of speeds a batch's jobs bring: speedup_only -- every job speeds up (the walk kernel specialised for speeds >= 1
// applies); any_speed -- not all do, but every speed is one the speed-up kernel's slow-down instantiations take (round 5).
struct SpxSpeedClass { int maxC; bool speedup_only, any_speed; };
static SpxSpeedClass speed_class(const spx_stream_job* jobs, int n) {
  SpxSpeedClass c = {1, true, true};
  for (int i = 0; i < n; i++) {
    if (jobs[i].channels > c.maxC) c.maxC = jobs[i].channels;
    const bool ok = jobs[i].speed > 0.0f && jobs[i].speed < SPX_FAST_MAX_SPEED && jobs[i].nonlinear >= 0.0f && jobs[i].nonlinear <= 1.0f;
    if (!(ok && jobs[i].speed > 1.0f)) c.speedup_only = false;
    if (!ok) c.any_speed = false;
  }
  if (c.speedup_only) c.any_speed = false;   // (the flag means: slow-down jobs are there)
  return c;
}
static SpxModeWalk mode_walk(const SpxPlanDev& d, int n, int maxC, bool speedup_only, bool lean, bool any_speed = false) {
  const SpxWalkConfig c = spx_walk_config(d, n, maxC, speedup_only, false, lean, any_speed);
  SpxModeWalk w;
  w.lds = c.lds; w.waves = c.waves; w.fast_kernel = c.fast_kernel; w.nwc = c.nwc;
  w.vgprs = spx_walk_vgprs(d, n, maxC, speedup_only, lean, any_speed);
  return w;
}
// (cached per plan and shape: the register queries and spx_walk_config are not free, and the engine asks on every call)
static const SpxModeResources& mode_resources(spx_plan* plan, int n, int maxC, bool speedup_only, bool any_speed = false) {
  const long long key = ((long long)n << 16) ^ ((long long)maxC << 2) ^ (any_speed ? 2 : 0) ^ (speedup_only ? 1 : 0);
  auto it = plan->res_cache.find(key);
  if (it != plan->res_cache.end()) return it->second;
  const SpxPlanDev& d = plan->dev;
  SpxModeResources R;
  memset(&R, 0, sizeof(R));
  R.cu_count = plan->cu_count;
  R.lds_per_cu = plan->lds_per_cu;
  R.walk = mode_walk(d, n, maxC, speedup_only, false, any_speed);
  R.walk_lean = R.walk;
  if (maxC == 1 && n <= plan->cu_count && R.walk.fast_kernel && R.walk.nwc > 0) {
    R.walk_lean = mode_walk(d, n, maxC, speedup_only, true, any_speed);
    R.lean_valid = true;
  }
  R.tension_lds = spx_tension_lds_bytes();
  R.tension_vgprs = spx_tension_vgprs();
  R.tile_default = d.tile_frames;
  R.tile_big = spx_analysis_tile_frames();
  R.tile_small = spx_analysis_small_tile_frames();
  SpxPlanDev d8 = d;
  d8.tile_frames = R.tile_small;
  R.an_lds_default = spx_analysis_lds_bytes(d);
  R.an_vgprs_default = spx_analysis_vgprs(d);
  R.an_lds_small = spx_analysis_lds_bytes(d8);
  R.an_vgprs_small = spx_analysis_vgprs(d8);
  if (plan->res_cache.size() > 64) plan->res_cache.clear();
  return plan->res_cache.emplace(key, R).first->second;
}
// Diagnostics: the resource numbers spx_choose_mode is fed for a batch of this shape, in the order of the fields cu_count ..
// an_vgprs_small of speedy_amd/csrc/spx_mode_table.cpp's query (22 values).  tools/kernel_resources.py writes them to
// profiles/kernel_resources.json, the CPU table test (tests/test_mode_table.py) reads them from there, and
// tests/test_gpu_parity.py checks the file against the library.
extern "C" int spx_debug_mode_resources(int sample_rate, int channels, int n_streams, int speedup_only, long long* out) {
  spx_plan* plan = shared_plan_full(sample_rate, 0);
  if (!plan || !out || n_streams < 1) return -1;
  std::lock_guard<std::mutex> g(plan->mu);
  const SpxModeResources& R = mode_resources(plan, n_streams, channels < 1 ? 1 : channels, speedup_only != 0, speedup_only == 0);
  const long long v[22] = {R.cu_count, (long long)R.lds_per_cu, (long long)R.walk.lds, R.walk.waves, R.walk.vgprs, R.walk.fast_kernel, R.walk.nwc,
                           (long long)R.walk_lean.lds, R.walk_lean.waves, R.walk_lean.vgprs, R.walk_lean.fast_kernel, R.walk_lean.nwc, R.lean_valid,
                           (long long)R.tension_lds, R.tension_vgprs, R.tile_default, R.tile_big, R.tile_small, (long long)R.an_lds_default,
                           (long long)R.an_lds_small, R.an_vgprs_default, R.an_vgprs_small};
  for (int i = 0; i < 22; i++) out[i] = v[i];
  return 0;
}

// Job tables (and, concurrent mode, the tile order) go through a plan-owned pinned slot -- the copy is asynchronous and must not
// read host memory that dies when the call returns -- and reach the workspace by ONE small kernel on `on` that also clears the
// hand-off flags: a single stream operation where two copies and two fills (each its own DMA packet with barriers around it)
// cost the concurrent mode 0.13 ms a call.  *done: the slot's event, recorded behind the kernel.
static int stage_tables(spx_plan* plan, const std::vector<SpxStreamDev>& sv, const std::vector<int>& order, SpxStreamDev* dstreams,
                        int* d_order, int* d_flags, unsigned n_flags, int* d_ready, unsigned n_ready, hipStream_t on, hipEvent_t* done,
                        const int* gate_started = nullptr, int gate_n = 0, unsigned gate_spins = 0) {
  const size_t b_sv = sizeof(SpxStreamDev) * sv.size(), b_or = sizeof(int) * order.size();
  SpxStage& G = plan->stage[plan->stage_next];
  plan->stage_next ^= 1;
  if (G.done) HIPCHK(hipEventSynchronize(G.done));  // the kernel that last read this slot (two calls ago) has retired
  else HIPCHK(hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
  if (G.cap < b_sv + b_or) {
    if (G.p) (void)hipHostFree(G.p);
    G.p = nullptr; G.cap = 0;
    const size_t cap = (b_sv + b_or) * 2 + 4096;
    HIPCHK(hipHostMalloc(&G.p, cap, hipHostMallocDefault));
    G.cap = cap;
  }
  unsigned char* hp = static_cast<unsigned char*>(G.p);
  memcpy(hp, sv.data(), b_sv);
  if (b_or) memcpy(hp + b_sv, order.data(), b_or);
  hipLaunchKernelGGL(spx_stage_kernel, dim3(64), dim3(256), 0, on, reinterpret_cast<const unsigned*>(hp),
                     reinterpret_cast<unsigned*>(dstreams), (unsigned)(b_sv / 4), reinterpret_cast<unsigned*>(d_order), (unsigned)(b_or / 4),
                     reinterpret_cast<unsigned*>(d_flags), n_flags, reinterpret_cast<unsigned*>(d_ready), n_ready, gate_started, gate_n, gate_spins);
  HIPCHK(hipEventRecord(G.done, on));
  *done = G.done;
  return 0;
}

// HIP events around a launch while spx_set_timing is on (no host synchronisation is added to the call).
struct SpxTimed {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s = nullptr;
  int kind = 0;
  SpxTimed(bool on, int kind_, hipStream_t s_) : s(s_), kind(kind_) {
    if (on) { a = take_event(); b = take_event(); (void)hipEventRecord(a, s); }
  }
  ~SpxTimed() {
    if (!a) return;
    (void)hipEventRecord(b, s);
    std::lock_guard<std::mutex> g(g_tmu);
    g_ev_pending.push_back({a, b, kind});
  }
};

// `force`: the call is one group of a mixed-rate batch (spx_batch_run_mixed): the launch mode was decided for all groups
// together, and the device guard is held by the caller.
//   ahead_sa: the group of a pipelined mixed call -- its producers go to this stream at once (spx_batch_run_mixed_ahead orders it);
//   started_out: where the group's walk workgroups count themselves in (for the next call's gate)
//   total_streams: of all groups of the mixed call; after_analysis: recorded behind the group's analysis launch (or null)
struct SpxForce { int concurrent; bool idle_start; int total_streams; hipEvent_t after_analysis; hipStream_t ahead_sa; const int** started_out; };
struct SpxCallOpts {
  const SpxForce* force = nullptr;
  bool ahead_req = false;      // spx_batch_run_ahead: pipelined with the plan's previous call where the shape allows
  bool overlap_req = false;    // spx_batch_run_overlapped: ... and its walk kernel beside the previous call's
  void* in_ready = nullptr;    // hipEvent_t: the producers wait for it (the caller's "input is there")
  bool sub = false;            // a sub-batch of a plain call the engine has split (run_split): its `out` is the whole call's, never
                               // "the previous call's output buffer handed over again"
  // The pipeline object's calls (spx_pipeline.hip): done_event (a hipEvent_t) is recorded behind everything the call enqueued --
  // and with `detached`, a call whose walk kernel goes to one of the library's walk streams does not touch hip_stream AT ALL
  // (no note, no wait for the walk kernel: the event is recorded on the walk stream).  A caller stream that carries nothing but
  // waits for walk kernels is a hardware queue whose head is a blocked barrier packet for 2 ms of every 2; depending on where
  // that queue happens to land among the process's queues, the kernels of the side stream were dispatched 50 us late after each
  // such packet (profiles/r05/r5e_trace_*.txt, r5f_queue_probe.txt: 1.14 against 1.03 ms per batch).
  void* done_event = nullptr;
  bool detached = false;
};

// One batch call: decide the launch mode (spx_choose_mode, a pure function of the inputs collected here), then execute it.
// The three launch orders give the same results:
//   in sequence  -- staging, analysis, tension, walk on the caller's stream (time chunks: the analysis of chunk c + 1 on a side
//                   stream beside the walk of chunk c);
//   concurrent   -- analysis and tension kernels on the device's two side streams, the walk kernel at once on the caller's:
//                   tiles of frames, then speeds, are handed over through flags the consumers poll;
//   ahead        -- staging, analysis and tension kernels on the side stream AT ONCE, beside the previous call's walk kernel;
//                   the walk kernel behind them on the caller's stream, or (walk2) on one of the library's two walk streams.
static int run_impl(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out,
                    int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps, void* hs, bool do_a,
                    bool do_w, const SpxCallOpts& opt = SpxCallOpts()) {
  if (!plan || !jobs || n <= 0) return fail(-1, "spx_batch: bad arguments");
  SpxRange range_(do_a && do_w ? "spx_batch_run" : (do_a ? "spx_batch_analyze" : "spx_batch_walk"));
  const SpxForce* force = opt.force;
  SpxPlanDev d = plan->dev;  // a copy: the tile size is chosen per call
  Layout L = layout_for(d, jobs, n);
  if (ws_bytes < L.total || !ws) return fail(-1, "spx_batch: workspace too small");
  const SpxSpeedClass SC = speed_class(jobs, n);
  const int maxC = SC.maxC;
  const bool speedup_only = SC.speedup_only, any_speed = SC.any_speed;
  std::lock_guard<std::mutex> plan_lock(plan->mu);
  const SpxModeResources& R = mode_resources(plan, n, maxC, speedup_only, any_speed);
  if (R.walk.lds > 160 * 1024)   // one CU's LDS; the window holds every channel of maxRequired + 64 frames at least
    return fail(-1, "spx_batch: too many channels for the walk kernel's LDS window");
  hipStream_t st = static_cast<hipStream_t>(hs);
  // ---- decide ----
  SpxModeShape S;
  S.n = n; S.max_channels = maxC; S.do_a = do_a; S.do_w = do_w; S.has_frames = L.total_frames > 0;
  S.forced = force != nullptr;
  S.force_concurrent = force && force->concurrent != 0;
  S.force_ahead = force && force->ahead_sa != nullptr;
  S.force_total_streams = force ? force->total_streams : n;
  S.ahead_req = opt.ahead_req; S.overlap_req = opt.overlap_req;
  const SpxModeEnv E = mode_env();
  SpxModeRuntime T;
  memset(&T, 0, sizeof(T));
  const int two_back = (plan->ahead_calls + SPX_RING - 2) % SPX_RING;
  T.two_workspaces = plan->ev_walk_valid[two_back] && plan->ring_ws[two_back] == ws;
  T.trial_key = ((long long)n << 40) ^ ((long long)L.total_frames << 8) ^ (maxC << 3) ^ (any_speed ? 4 : 0) ^ (speedup_only ? 2 : 0) ^ (opt.overlap_req ? 1 : 0);
  T.device_ours = device_ours_cb;
  T.device_ctx = &plan->device;
  spx_plan::Trial& TR = plan->trial;
  if (TR.state.key == T.trial_key && TR.state.choice < 0 && TR.state.calls >= 3 && TR.ev[1] && TR.ev[3] &&
      hipEventQuery(TR.ev[1]) == hipSuccess && hipEventQuery(TR.ev[3]) == hipSuccess)
    T.trial_times_ready = hipEventElapsedTime(&T.ms_seq, TR.ev[0], TR.ev[1]) == hipSuccess &&
                          hipEventElapsedTime(&T.ms_con, TR.ev[2], TR.ev[3]) == hipSuccess;
  (void)hipGetLastError();
  SpxMode M = spx_choose_mode(S, R, E, T, TR.state);
  // another concurrent-mode call still in flight on this device, on a different stream?  Then this one runs its kernels in
  // sequence (SpxDevGuard above); the guard stays locked until this call has left its own event behind.
  SpxDevGuard& guard = g_guard[(plan->device >= 0 && plan->device < 64) ? plan->device : 0];
  std::unique_lock<std::mutex> guard_lock(guard.mu, std::defer_lock);
  bool idle_start = force ? force->idle_start : false;
  if (M.want_concurrent && !force) {
    guard_lock.lock();
    const hipError_t q = guard.valid ? hipEventQuery(guard.last) : hipSuccess;
    (void)hipGetLastError();  // hipErrorNotReady is not an error of this call
    idle_start = (q == hipSuccess);   // the previous concurrent-mode call (if any) has drained
    if (guard.valid && guard.last_stream != st && q == hipErrorNotReady) {
      T.guard_busy = true;
      M = spx_choose_mode(S, R, E, T, TR.state);
      guard_lock.unlock();
    }
  }
  static const bool dbg_trial = getenv("SPX_DEBUG_TRIAL") != nullptr;
  if (dbg_trial && TR.state.choice < 0 && M.trial_next.choice >= 0 && T.trial_times_ready)
    fprintf(stderr, "[spx trial] n=%d maxC=%d: concurrent %.3f ms, in sequence %.3f ms -> %s\n", n, maxC, T.ms_con, T.ms_seq,
            M.trial_next.choice ? "concurrent" : "sequence");
  TR.state = M.trial_next;
  d.tile_frames = M.tile_frames;
  const int nch = M.nch;
  const bool concurrent = M.concurrent, ahead = M.ahead;
  // ---- execute ----
  std::vector<SpxStreamDev> sv;
  std::vector<int> tiles;
  int rc = build_streams(d, jobs, n, nch, sv, tiles);
  if (rc) return rc;
  unsigned char* w = static_cast<unsigned char*>(ws);
  SpxStreamDev* dstreams = reinterpret_cast<SpxStreamDev*>(w + L.off_streams);
  SpxStreamState* states = reinterpret_cast<SpxStreamState*>(w + L.off_states);
  SpxFrameRec* rec = reinterpret_cast<SpxFrameRec*>(w + L.off_rec);
  float* scratch = reinterpret_cast<float*>(w + L.off_scratch);
  int* d_order = reinterpret_cast<int*>(w + L.off_order);
  int* d_flags = reinterpret_cast<int*>(w + L.off_flags);
  int* d_ready = reinterpret_cast<int*>(w + L.off_ready);
  SpxTapsDev td = taps_of(taps);
  const bool timed = g_timing.load() && do_a && do_w;
  if (do_a && do_w) g_last_concurrent.store(concurrent ? 1 : (ahead ? 2 : 0), std::memory_order_relaxed);
  static const bool dbg_mode = getenv("SPX_DEBUG_MODE") != nullptr;   // one line per call: what was decided and why
  if (dbg_mode)
    fprintf(stderr, "[spx mode] rate %d n %d maxC %d: co_resident %d doubtful %d lean %d want_concurrent %d chunks %d tiles %d -> %s%s\n",
            d.rate, n, maxC, (int)M.co_resident, (int)M.doubtful, (int)M.launch_lean, (int)M.want_concurrent, nch, tiles[0],
            concurrent ? "concurrent" : (ahead ? (M.seq_ahead ? "ahead (kernels in sequence)" : "ahead") : "sequence"),
            M.walk2 ? ", walk kernels overlapping" : "");
  // the stream the walk kernel goes to: the caller's, or (walk2) one of the library's two, taking turns -- ordered behind whatever
  // the caller had queued by the PREVIOUS call (the consumer of the output this call overwrites, two buffers taking turns) and not
  // behind this call's state of the stream, which ends with the wait for the previous call's walk kernel; a caller that hands over
  // the previous call's out / n_out again gets exactly that wait
  hipStream_t stw = st;
  const bool detached = opt.detached && M.walk2 && !force;
  if (do_w && !force && !detached && ring_note(plan, st)) return -2;
  if (do_w && !force && detached) { plan->ev_call_valid[plan->ahead_calls & 1] = false; plan->ev_call_st[plan->ahead_calls & 1] = nullptr; }
  if (M.walk2) {
    hipStream_t wst[SPX_MAX_WALK_STREAMS];
    const int nws = walk_stream_count();
    if (dev_walk_streams(plan->device, wst, nws)) return fail(-1, "spx_batch: no walk streams");
    const int cur = plan->ahead_calls & 1;
    stw = wst[plan->ahead_calls % nws];
    const bool same_out = !opt.sub && (out == plan->ahead_last_out || n_out == plan->ahead_last_nout);
    // (the previous call of the plan on ANOTHER stream, or a detached one -- a pipeline object at work on the same plan: its note says
    // nothing about this caller's stream, so the walk kernel is ordered behind everything that is on it now)
    const bool foreign_prev = !plan->ev_call_valid[cur ^ 1] || plan->ev_call_st[cur ^ 1] != st;
    if (detached) { }   // (the owner of the buffers orders their consumers itself: spx_pipeline waits for done_event on the host)
    else if (same_out || (foreign_prev && plan->ahead_calls > 0)) HIPCHK(hipStreamWaitEvent(stw, plan->ev_call[cur], 0));
    else if (plan->ev_call_valid[cur ^ 1]) HIPCHK(hipStreamWaitEvent(stw, plan->ev_call[cur ^ 1], 0));
  }
  hipStream_t sa = st;  // the stream the analysis launches go to
  if (nch > 1 || concurrent || ahead) {
    if (!plan->side) {
      if (dev_side_streams(plan->device, &plan->side, &plan->side2)) return fail(-1, "spx_batch: no side streams");
      HIPCHK(hipEventCreateWithFlags(&plan->ev_start, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&plan->ev_tension, hipEventDisableTiming));
      for (auto& e : plan->ev_chunk) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    sa = M.ahead_forced ? force->ahead_sa : plan->side;
  }
  std::vector<int> order;
  if (concurrent) {
    // tile ids in launch order: tile t of every stream before tile t+1 of any
    order.reserve((size_t)tiles[0]);
    std::vector<int> cnt(n);
    int maxT = 0;
    for (int i = 0; i < n; i++) {
      cnt[i] = (i + 1 < n ? sv[i + 1].first_tile : tiles[0]) - sv[i].first_tile;
      if (cnt[i] > maxT) maxT = cnt[i];
    }
    for (int t = 0; t < maxT; t++)
      for (int i = 0; i < n; i++)
        if (t < cnt[i]) order.push_back(sv[i].first_tile + t);
  }
  bool waited_prev = false;
  if (ahead && !force) {
    // this call's producers must not touch a workspace the walk kernel of an earlier call still reads (ring_wait)
    if (ring_wait(plan, sa, ws, st, &waited_prev)) return -2;
    if (opt.in_ready) HIPCHK(hipStreamWaitEvent(sa, static_cast<hipEvent_t>(opt.in_ready), 0));
  }
  if (!ahead && opt.in_ready) HIPCHK(hipStreamWaitEvent(st, static_cast<hipEvent_t>(opt.in_ready), 0));
  hipEvent_t staged_ev = nullptr;
  // (Round 5 tried staging a DETACHED call's tables on its own, otherwise empty, run stream -- beside the previous call's producers
  // instead of in front of this call's on the producers' stream: 1.045 against 0.94 ms per step.  A stream that holds nothing but
  // waits and one small kernel is exactly the "blocked barrier packets" case of INTEGRATION.md's hardware-queue section.)
  static const bool no_gate = spx_tuning_env("SPX_NO_GATE") != nullptr;  // A/B only
  static const bool split_gate = spx_tuning_env("SPX_SPLIT_GATE") != nullptr;  // A/B only: the pipelined call's gate as a kernel of its own
  // AHEAD: the analysis must not fill the CUs before the PREVIOUS call's walk workgroups have been placed one per CU (its walk
  // kernel becomes runnable at the same moment as this analysis: when the walk before it retires) -- only while that call is
  // still in flight (then its workspace, where the counter lives, is alive by the usual contract), and not when the producers
  // wait for that very call anyway (a caller handing the same workspace over again: the counter is THIS workspace's, this
  // call's staging kernel has just cleared it, and the gate would spin its full bound for a count nobody raises -- round 4:
  // 3.6 ms per call where a plain call takes 1.6).  A longer bound than the idle-start gate's: the previous walk kernel may
  // itself be waiting for something of the caller's (an output buffer still being copied out), and a gate that gives up early
  // lets this call's analysis fill the CUs first, which costs the previous call half its speed; ~2 ms.  The gate runs at the end of
  // the staging kernel (same stream, nothing in between).
  const bool ahead_gate = ahead && !force && plan->ahead_started != nullptr && plan->ahead_n > 0 && !no_gate && !waited_prev &&
                          ring_previous_in_flight(plan);
  rc = stage_tables(plan, sv, order, dstreams, d_order, d_flags, concurrent ? (unsigned)tiles[0] : 0u, d_ready,
                    (concurrent || ahead) ? (unsigned)n + 1u : 0u, ahead ? sa : st, &staged_ev,
                    (ahead_gate && !split_gate) ? plan->ahead_started : nullptr, plan->ahead_n, 8000u);
  if (rc) return rc;
  if (M.trial_slot >= 0) {  // bracket this call on the caller's stream (spx_plan::Trial)
    hipEvent_t& e0 = TR.ev[2 * M.trial_slot];
    if (!e0) HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventRecord(e0, st));
  }
  if (sa != st && !ahead) {
    // the side stream starts after everything already queued on the caller's stream (job tables, cleared flags, and the previous
    // call's walk, which still reads the frame records this call's analysis will overwrite): the staging slot's event marks
    // exactly that point of the caller's stream
    HIPCHK(hipStreamWaitEvent(sa, staged_ev, 0));
    if (concurrent) HIPCHK(hipStreamWaitEvent(plan->side2, staged_ev, 0));
  }
  static const unsigned gate_spins = [] { const char* e = spx_tuning_env("SPX_GATE_SPINS"); return e ? (unsigned)atoi(e) : 1200u; }();
  static const bool diag_nowait = spx_tuning_env("SPX_DIAG_NOWAIT") != nullptr;  // DIAGNOSTIC ONLY: the walk reads the speeds the
  // previous identical call left in the scratch array instead of waiting for this call's (timing experiments)
  for (int c = 0; c < nch; c++) {
    SpxStreamDev* dj = dstreams + (size_t)c * n;
    // Concurrent mode on an IDLE device (the first call after a synchronisation): kernels start as their launches arrive, and the
    // analysis launch arrives a few tens of microseconds before the walk launch.  Its tiles then fill the CUs' LDS and the walk
    // workgroups land two to a CU wherever room is left -- the whole step waits for those chains (3.06 instead of 2.21 ms in 40 %
    // of such calls; with the host running ahead the walk kernel sits right behind the staging kernel in its queue and is placed
    // first by itself).  The walk kernel cannot simply be enqueued first: HIP maps streams onto a few hardware queues, and
    // producers queued behind a waiting consumer in a shared queue never start (every consumer is enqueued after its producers,
    // which is safe with any mapping).  So an idle start holds the analysis stream back with a gate kernel until the walk kernel's
    // workgroups have been placed (they count themselves in; spx_gate_kernel).
    if (concurrent && do_w && idle_start && !no_gate)
      hipLaunchKernelGGL(spx_gate_kernel, dim3(1), dim3(64), 0, sa, d_ready + n, n, gate_spins);
    if (ahead_gate && split_gate && c == 0)
      hipLaunchKernelGGL(spx_gate_kernel, dim3(1), dim3(64), 0, sa, plan->ahead_started, plan->ahead_n, 8000u);
    if (do_a && tiles[c] > 0) {
      SpxTimed tm(timed, 0, sa);
      spx_launch_analysis(d, dj, n, tiles[c], in, rec, td, concurrent ? d_order : nullptr, concurrent ? d_flags : nullptr, sa);
    }
    if (force && force->after_analysis && c == nch - 1) HIPCHK(hipEventRecord(force->after_analysis, sa));
    if (sa != st) HIPCHK(hipEventRecord(plan->ev_chunk[c], sa));
    if (sa != st && !concurrent && !ahead) HIPCHK(hipStreamWaitEvent(st, plan->ev_chunk[c], 0));
    if (do_w) {
      // frame-rate stage: after the analysis on the same stream, or -- concurrent -- beside it on its own stream, consuming tile
      // flags and publishing the count of ready speeds (AHEAD: behind the analysis on the side stream)
      hipStream_t stn = concurrent ? plan->side2 : (ahead ? sa : st);
      {
        SpxTimed tm(timed, 2, stn);
        spx_launch_tension(d, dj, n, states, rec, scratch, td, concurrent ? d_flags : nullptr, (concurrent || ahead) ? d_ready : nullptr, stn);
      }
      if (concurrent || ahead) HIPCHK(hipEventRecord(plan->ev_tension, stn));
      if (ahead) HIPCHK(hipStreamWaitEvent(stw, plan->ev_tension, 0));   // the walk kernel starts when every speed of the call is there
      {
        // Kernels in sequence, and every stream of the call (of all groups of a mixed call) can have a CU to itself: ask for more
        // than half a CU's LDS per walk workgroup, so that they DO get one each.  Walk kernels of several groups launched side by
        // side, or a walk kernel placed while another group's analysis fills the CUs, otherwise land two to a CU here and there,
        // and those chains end the call (configs[4] shard 3.30 -> 3.04 ms, profiles/r03/r03ad_config4_lds_min.txt).  (The
        // concurrent mode needs that LDS for the analysis workgroups beside the walk; its idle-start gate does this job.)
        // AHEAD: the counts are all published by the time the kernel starts -- its one poll returns at once -- and its workgroups
        // count themselves in for the next call's gate.
        SpxTimed tm(timed, 1, stw);
        spx_launch_walk(d, dj, n, maxC, in, out, n_out, states, scratch, ((concurrent && !diag_nowait) || ahead) ? d_ready : nullptr,
                        speedup_only, stw, false, M.exclusive_cu ? R.lds_per_cu / 2 + 1024 : 0, M.launch_lean, any_speed);
      }
      if (M.ahead_forced && force->started_out) *force->started_out = d_ready + n;
      if (c == nch - 1 && !force) {
        // every call of the plan that walks leaves its event in the ring (a pipelined call orders its producers behind the walk
        // kernels of the calls before it, pipelined or not)
        if (ring_record(plan, stw, ws, st, out, n_out, detached)) return -2;
        plan->ahead_started = (concurrent || ahead) ? d_ready + n : nullptr;   // (only these walk kernels count themselves in)
        plan->ahead_n = n;
        plan->mixed_started.clear();
      }
    }
    // the caller's stream is "done" only when the side launches have retired too
    if (concurrent) {
      HIPCHK(hipStreamWaitEvent(st, plan->ev_chunk[c], 0));
      HIPCHK(hipStreamWaitEvent(st, plan->ev_tension, 0));
    }
  }
  if (concurrent && !force) {
    // leave this call's completion behind for the next concurrent-mode call on the device (guard still locked)
    if (!guard.last) HIPCHK(hipEventCreateWithFlags(&guard.last, hipEventDisableTiming));
    HIPCHK(hipEventRecord(guard.last, st));
    guard.last_stream = st;
    guard.valid = true;
  }
  if (M.trial_slot >= 0) {
    hipEvent_t& e1 = TR.ev[2 * M.trial_slot + 1];
    if (!e1) HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e1, st));
  }
  if (opt.done_event && do_w) HIPCHK(hipEventRecord(static_cast<hipEvent_t>(opt.done_event), detached ? stw : st));
  if (timed) { std::lock_guard<std::mutex> g(g_tmu); g_calls_pending++; }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- a call of more streams than the device has CUs, split into sub-batches that overlap one another (round 5) ----
// A batch of 257 .. 2 x CUs streams runs its kernels in sequence as ONE call (512 streams: 2.85 ms), where two 256-stream calls in
// the overlapped order take 2.3.  So an OVERLAPPED call of that size (spx_batch_run_overlapped, the pipeline object) is cut into
// two sub-batches of equal size -- same streams, in job order; their workspaces carved from the caller's; in / out / n_out as they
// are -- issued as overlapped calls on the plan's ring: sub-batch k + 1's analysis runs beside sub-batch k's walk kernel, the
// walk kernels overlap, and so do those of consecutive calls.  Taken only where a sub-batch would run in the pipelined order with
// overlapping walk kernels (spx_choose_mode on the sub-shape says so).
// PLAIN calls are NOT split (measured, profiles/r05/r5b_scale_streams.txt): spx_batch_run promises plain stream order, so every
// sub-batch's producers must wait for the caller's stream as it stood at the call and nothing of the NEXT call can start before
// this call's last walk kernel ends -- 512 streams 3.21 ms split against 2.85 as one call, 1 024 streams 5.63 against 4.34 (the
// throughput-form walk kernel).  Above two streams per CU the throughput form wins in every order (1 024 streams: four
// overlapped 256-stream calls 4.8 ms).  (SPX_SPLIT_PLAIN / SPX_SPLIT_MAX in the tuning build: the A/B.)
#define SPX_SPLIT_LIMIT 4   // sub-batches the workspace is sized for
struct SplitPlan { int k; std::vector<int> first; std::vector<size_t> ws_off, ws_bytes; size_t total; };
static SplitPlan split_geometry(const spx_plan* plan, const spx_stream_job* jobs, int n, int max_mult) {
  SplitPlan P;
  P.k = 1; P.total = 0;
  const int cu = plan->cu_count > 0 ? plan->cu_count : 1;
  if (max_mult > SPX_SPLIT_LIMIT) max_mult = SPX_SPLIT_LIMIT;
  if (n <= cu || n > max_mult * cu) return P;
  P.k = (n + cu - 1) / cu;
  size_t o = 0;
  for (int i = 0; i < P.k; i++) {
    const int a = (int)((long long)n * i / P.k), b = (int)((long long)n * (i + 1) / P.k);
    P.first.push_back(a);
    const size_t bytes = layout_for(plan->dev, jobs + a, b - a).total;
    P.ws_off.push_back(o); P.ws_bytes.push_back(bytes);
    o += (bytes + 255) & ~(size_t)255;
  }
  P.first.push_back(n);
  P.total = o;
  return P;
}
static int run_split(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out, int64_t* n_out, void* ws,
                     size_t ws_bytes, const spx_taps* taps, void* hs, const SpxCallOpts& opt) {
  if (!plan || !jobs || n <= 0) return fail(-1, "spx_batch: bad arguments");
  static const int max_overlapped = [] { const char* e = spx_tuning_env("SPX_SPLIT_MAX"); return e ? atoi(e) : 2; }();
  static const int max_plain = [] { const char* e = spx_tuning_env("SPX_SPLIT_PLAIN"); return e ? atoi(e) : 1; }();
  const bool plain = !opt.ahead_req;
  SplitPlan P = split_geometry(plan, jobs, n, plain ? max_plain : (opt.overlap_req ? max_overlapped : 1));
  if (P.k > 1 && (g_chunks_set.load() || ws_bytes < P.total)) P.k = 1;
  if (P.k > 1) {
    // would a sub-batch take the pipelined order with overlapping walk kernels?
    const SpxSpeedClass SC = speed_class(jobs, n);
    const int maxC = SC.maxC;
    std::lock_guard<std::mutex> plan_lock(plan->mu);
    const int m = P.first[1] - P.first[0];
    const SpxModeResources& R = mode_resources(plan, m, maxC, SC.speedup_only, SC.any_speed);
    SpxModeShape S;
    memset(&S, 0, sizeof(S));
    S.n = m; S.max_channels = maxC; S.do_a = S.do_w = true; S.has_frames = true; S.force_total_streams = m;
    S.ahead_req = S.overlap_req = true;
    SpxModeRuntime T;
    memset(&T, 0, sizeof(T));
    T.trial_key = -2;
    T.device_ours = device_ours_cb;
    T.device_ctx = &plan->device;
    const SpxModeTrial none = {-1, 0, -1};
    const SpxMode M = spx_choose_mode(S, R, mode_env(), T, none);
    if (!(M.ahead && M.walk2)) P.k = 1;
  }
  { std::lock_guard<std::mutex> g(plan->mu); if (plan->split_of.size() > 64) plan->split_of.clear(); plan->split_of[ws] = P.k; }
  if (P.k <= 1) return run_impl(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, true, true, opt);
  hipStream_t st = static_cast<hipStream_t>(hs);
  void* ready = opt.in_ready;
  if (plain) {
    // plain stream order: every sub-batch's producers wait for the caller's stream as it stands now (one event per call in
    // flight would be the exact thing; two taking turns are enough: the event is waited for by the sub-batches' producer
    // streams, and those are ordered behind the previous split call's by the ring)
    std::lock_guard<std::mutex> g(plan->mu);
    hipEvent_t& e = plan->ev_split[plan->split_calls++ & 1];
    if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e, st));
    ready = e;
  }
  int64_t rows = 0;
  for (int i = 0; i < P.k; i++) {
    const int a = P.first[i], m = P.first[i + 1] - a;
    spx_taps t;
    if (taps) {
      t = *taps;
      if (t.tension) t.tension += rows;
      if (t.speed) t.speed += rows;
      if (t.features) t.features += rows * SPX_FEATURE_COUNT;
      if (t.spectrogram) t.spectrogram += rows * (int64_t)plan->dev.N;
      if (t.normalized) t.normalized += rows * (int64_t)plan->dev.W;
      for (int j = a; j < a + m; j++) rows += jobs[j].nonlinear != 0.0f ? frames_for(plan->dev, jobs[j].n_in) : 0;
    }
    SpxCallOpts o;
    o.ahead_req = o.overlap_req = true;
    o.sub = i > 0 || plain;      // (the first sub-batch of an overlapped call is ordered like the call itself: "the same out buffer again")
    o.in_ready = ready;
    if (i == P.k - 1) o.done_event = opt.done_event;   // (on hip_stream, which waits for every sub-batch's walk kernel: never detached)
    const int rc = run_impl(plan, jobs + a, m, in, out, n_out + a, static_cast<unsigned char*>(ws) + P.ws_off[i], P.ws_bytes[i],
                            taps ? &t : nullptr, hs, true, true, o);
    if (rc) return rc;   // (what the earlier sub-batches enqueued is already joined to hip_stream: ring_record)
  }
  return 0;
}

int spx_internal_run(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out, int64_t* n_out, void* ws,
                     size_t ws_bytes, const spx_taps* taps, void* hs, bool ahead, bool overlap, void* in_ready, void* done_event,
                     bool detached) {
  SpxCallOpts o;
  o.ahead_req = ahead; o.overlap_req = overlap; o.in_ready = in_ready; o.done_event = done_event; o.detached = detached;
  return run_split(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, o);
}

extern "C" {
// (the larger of the one-call layout and the sub-batch layouts: which of the two a spx_batch_run takes is decided per call)
size_t spx_batch_workspace_bytes(spx_plan_t plan, const spx_stream_job* jobs, int n_streams) {
  if (!plan || !jobs || n_streams < 1) return 0;
  return std::max(layout_for(plan->dev, jobs, n_streams).total, split_geometry(plan, jobs, n_streams, SPX_SPLIT_LIMIT).total);
}
int spx_batch_run(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out,
                  int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps, void* hs) {
  return run_split(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, SpxCallOpts());
}
int spx_batch_run_ahead(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out,
                        int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps, void* hs) {
  return spx_internal_run(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, true, false, nullptr, nullptr, false);
}
int spx_batch_run_overlapped(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out,
                             int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps, void* hs) {
  return spx_internal_run(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, true, true, nullptr, nullptr, false);
}
int spx_batch_run_ahead_when(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out,
                             int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps, void* hs, void* in_ready_event) {
  return spx_internal_run(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, true, false, in_ready_event, nullptr, false);
}
int spx_batch_analyze(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, void* ws,
                      size_t ws_bytes, const spx_taps* taps, void* hs) {
  return run_impl(plan, jobs, n, in, nullptr, nullptr, ws, ws_bytes, taps, hs, true, false);
}
int spx_batch_walk(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out,
                   int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps, void* hs) {
  return run_impl(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, false, true);
}

// ---- one call for a batch whose streams differ in sample rate (BASELINE configs[4]: 16 kHz and 22.05 kHz, mono and
// stereo, two speeds in one shard).  The reference fixes the rate per handle (soniclib.c:93, speedy.c:213-214), so any mix
// can be alive at once; here the tables and the kernel instantiations are per rate (a plan), so the batch is cut into one
// group per plan and ALL groups are launched together: every group on a HIP stream of its own (its walk kernel there, its
// analysis and tension kernels on the plan's side streams), forked from and joined to the caller's stream -- the walk
// workgroups of all groups are resident at the same time, one stream per CU as in a homogeneous batch.  The launch mode
// (concurrent: walk kernels polling for speeds beside the analysis kernels; or in sequence) is decided once for all
// groups together: the co-residency bound of run_impl counts the polling workgroups of all of them. ----
__global__ void spx_scatter_nout_kernel(const int64_t* __restrict__ src, const int* __restrict__ idx, int n, int64_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = src[i];
}
struct MixedLayout { std::vector<size_t> ws_off, ws_bytes; size_t off_nout, off_idx, total; };
static MixedLayout mixed_layout(const spx_plan_t* plans, int n_plans, const std::vector<std::vector<spx_stream_job>>& gj, int n) {
  MixedLayout M;
  size_t o = 0;
  for (int g = 0; g < n_plans; g++) {
    const size_t b = gj[g].empty() ? 0 : layout_for(plans[g]->dev, gj[g].data(), (int)gj[g].size()).total;
    M.ws_off.push_back(o); M.ws_bytes.push_back(b);
    o += (b + 255) & ~(size_t)255;
  }
  M.off_nout = o; o += ((sizeof(int64_t) * (size_t)n + 255) & ~(size_t)255);
  M.off_idx = o;  o += ((sizeof(int) * (size_t)n + 255) & ~(size_t)255);
  M.total = o;
  return M;
}
static int mixed_groups(int n_plans, const spx_stream_job* jobs, const int* plan_index, int n,
                        std::vector<std::vector<spx_stream_job>>& gj, std::vector<std::vector<int>>& gi) {
  gj.assign(n_plans, {}); gi.assign(n_plans, {});
  for (int i = 0; i < n; i++) {
    const int g = plan_index ? plan_index[i] : 0;
    if (g < 0 || g >= n_plans) return fail(-1, "spx_batch_run_mixed: plan_index out of range");
    gj[g].push_back(jobs[i]); gi[g].push_back(i);
  }
  return 0;
}
size_t spx_batch_workspace_bytes_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index,
                                       int n_streams) {
  std::vector<std::vector<spx_stream_job>> gj;
  std::vector<std::vector<int>> gi;
  if (!plans || n_plans < 1 || !jobs || n_streams < 1 || mixed_groups(n_plans, jobs, plan_index, n_streams, gj, gi)) return 0;
  return mixed_layout(plans, n_plans, gj, n_streams).total;
}
static int mixed_impl(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n,
                      const int16_t* in, int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps,
                      void* hs, bool ahead_req, void* in_ready = nullptr);
}  // extern "C"
// (spx_pipeline.hip: a pipelined mixed call whose producers wait for an "input is there" event)
int spx_internal_run_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n, const int16_t* in,
                           int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, void* hs, bool ahead, void* in_ready) {
  return mixed_impl(plans, n_plans, jobs, plan_index, n, in, out, n_out, ws, ws_bytes, nullptr, hs, ahead, in_ready);
}
extern "C" {
int spx_batch_run_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n,
                        const int16_t* in, int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, void* hs) {
  return mixed_impl(plans, n_plans, jobs, plan_index, n, in, out, n_out, ws, ws_bytes, nullptr, hs, false);
}
int spx_batch_run_mixed_taps(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n,
                             const int16_t* in, int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps,
                             void* hs) {
  return mixed_impl(plans, n_plans, jobs, plan_index, n, in, out, n_out, ws, ws_bytes, taps, hs, false);
}
// spx_batch_run_ahead for a mixed-rate batch: consecutive calls (two workspaces taking turns, one HIP stream, the same lead
// plan) pipelined -- every group's staging, analysis and tension kernels on the device's first side stream at once, beside the
// previous call's walk kernels; the groups' walk kernels on their streams as in a plain call, each behind its tension kernel.
int spx_batch_run_mixed_ahead(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n,
                              const int16_t* in, int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, void* hs) {
  return mixed_impl(plans, n_plans, jobs, plan_index, n, in, out, n_out, ws, ws_bytes, nullptr, hs, true);
}
static int mixed_impl(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n,
                      const int16_t* in, int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps,
                      void* hs, bool ahead_req, void* in_ready) {
  if (!plans || n_plans < 1 || n_plans > 8 || !jobs || n <= 0) return fail(-1, "spx_batch_run_mixed: bad arguments");
  SpxRange range_("spx_batch_run_mixed");
  std::vector<std::vector<spx_stream_job>> gj;
  std::vector<std::vector<int>> gi;
  int rc = mixed_groups(n_plans, jobs, plan_index, n, gj, gi);
  if (rc) return rc;
  const MixedLayout M = mixed_layout(plans, n_plans, gj, n);
  if (!ws || ws_bytes < M.total) return fail(-1, "spx_batch_run_mixed: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(hs);
  spx_plan* lead = plans[0];
  for (int g = 0; g < n_plans; g++) if (plans[g]->device != lead->device) return fail(-1, "spx_batch_run_mixed: plans of different devices");
  // ---- the launch mode, for all groups together (spx_choose_mixed_mode: the rules of run_impl, summed / maximised over the groups) ----
  std::vector<SpxModeGroup> G;
  for (int g = 0; g < n_plans; g++) {
    if (gj[g].empty()) continue;
    const SpxPlanDev& d = plans[g]->dev;
    const SpxSpeedClass SC = speed_class(gj[g].data(), (int)gj[g].size());
    const int maxC = SC.maxC;
    bool any_nl = false;
    for (const auto& j : gj[g]) any_nl = any_nl || j.nonlinear != 0.0f;
    SpxModeGroup mg;
    mg.n = (int)gj[g].size();
    mg.walk = mode_walk(d, mg.n, maxC, SC.speedup_only, false, SC.any_speed);
    if (mg.walk.lds > 160 * 1024) return fail(-1, "spx_batch_run_mixed: too many channels for the walk kernel's LDS window");
    mg.any_nonlinear = any_nl;
    mg.an_lds = spx_analysis_lds_bytes(d);
    mg.an_vgprs = spx_analysis_vgprs(d);
    G.push_back(mg);
  }
  const int groups = (int)G.size();
  static const int env_mixed = spx_tuning_env("SPX_MIXED_MODE") ? atoi(spx_tuning_env("SPX_MIXED_MODE")) : -1;   // tuning: 0 sequence, 1 concurrent
  static const bool no_sjf = spx_tuning_env("SPX_MIXED_NO_ORDER") != nullptr;   // A/B
  SpxModeRuntime T;
  memset(&T, 0, sizeof(T));
  T.device_ours = device_ours_cb;
  T.device_ctx = &lead->device;
  const SpxModeEnv E = mode_env();
  SpxMixedMode MM = spx_choose_mixed_mode(G.data(), groups, n, lead->cu_count, lead->lds_per_cu, spx_tension_lds_bytes(), spx_tension_vgprs(),
                                          E, env_mixed, no_sjf, ahead_req, T);
  // ---- the device guard, once for the whole call ----
  SpxDevGuard& guard = g_guard[(lead->device >= 0 && lead->device < 64) ? lead->device : 0];
  std::unique_lock<std::mutex> guard_lock(guard.mu, std::defer_lock);
  SpxForce force = {0, false, n, nullptr, nullptr, nullptr};
  if (MM.concurrent) {
    guard_lock.lock();
    const hipError_t q = guard.valid ? hipEventQuery(guard.last) : hipSuccess;
    (void)hipGetLastError();
    force.idle_start = (q == hipSuccess);
    if (guard.valid && guard.last_stream != st && q == hipErrorNotReady) {
      T.guard_busy = true;
      MM = spx_choose_mixed_mode(G.data(), groups, n, lead->cu_count, lead->lds_per_cu, spx_tension_lds_bytes(), spx_tension_vgprs(), E,
                                 env_mixed, no_sjf, ahead_req, T);
      guard_lock.unlock();
    }
  }
  const bool concurrent = MM.concurrent;
  // pipelined with the previous call (spx_batch_run_mixed_ahead): kernels in sequence, one stream per CU at most
  const bool ahead = MM.ahead;
  force.concurrent = concurrent ? 1 : 0;
  // ---- fork: every group on its plan's own stream ----
  std::lock_guard<std::mutex> lead_lock(lead->mix_mu);
  {
    // every call that goes through the lead plan's ring notes where the caller's stream stands when it is made: the next
    // spx_batch_run_overlapped on that plan orders its walk kernel behind THIS note (round 4 left it to plain calls only, and
    // an overlapped call behind a mixed one was ordered behind a stale note)
    std::lock_guard<std::mutex> ring_lock(lead->mu);
    if (ring_note(lead, st)) return -2;
  }
  if (!lead->ev_fork) HIPCHK(hipEventCreateWithFlags(&lead->ev_fork, hipEventDisableTiming));
  if (in_ready && !ahead) HIPCHK(hipStreamWaitEvent(st, static_cast<hipEvent_t>(in_ready), 0));   // (ahead: the producers' stream waits for it)
  HIPCHK(hipEventRecord(lead->ev_fork, st));
  unsigned char* w = static_cast<unsigned char*>(ws);
  int64_t* d_nout = reinterpret_cast<int64_t*>(w + M.off_nout);
  int* d_idx = reinterpret_cast<int*>(w + M.off_idx);
  // job -> group-order index table: one small pinned staging slot of the lead plan, copied by the stream
  std::vector<int> order;
  order.reserve((size_t)n);
  for (int g = 0; g < n_plans; g++) order.insert(order.end(), gi[g].begin(), gi[g].end());
  // (copied by the caller's stream -- or, pipelined, by the producers' stream below: the copy and its dispatch then are not part of
  // what the caller's stream runs between the previous call's walk kernel and this call's, 44 -> 25 us between the two)
  auto upload_idx = [&](hipStream_t on) -> int {
    SpxStage& G = lead->mix_stage;
    if (G.done) HIPCHK(hipEventSynchronize(G.done));
    else HIPCHK(hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
    if (G.cap < sizeof(int) * (size_t)n) {
      if (G.p) (void)hipHostFree(G.p);
      G.p = nullptr; G.cap = 0;
      HIPCHK(hipHostMalloc(&G.p, sizeof(int) * (size_t)n * 2 + 1024, hipHostMallocDefault));
      G.cap = sizeof(int) * (size_t)n * 2 + 1024;
    }
    memcpy(G.p, order.data(), sizeof(int) * (size_t)n);
    HIPCHK(hipMemcpyAsync(d_idx, G.p, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, on));
    HIPCHK(hipEventRecord(G.done, on));
    return 0;
  };
  if (!ahead && upload_idx(st)) return -2;
  std::vector<size_t> gpos(n_plans, 0);
  { size_t pos = 0; for (int g = 0; g < n_plans; g++) { gpos[g] = pos; pos += gj[g].size(); } }
  // taps: the rows of group g follow those of groups 0 .. g-1 (plan order, whatever order the groups are launched in); the
  // row widths of the two spectrum taps are the group's own N and W
  std::vector<spx_taps> gtaps(n_plans);
  if (taps) {
    size_t rows = 0, o_spec = 0, o_norm = 0;
    for (int g = 0; g < n_plans; g++) {
      spx_taps& t = gtaps[g];
      t.tension = taps->tension ? taps->tension + rows : nullptr;
      t.speed = taps->speed ? taps->speed + rows : nullptr;
      t.features = taps->features ? taps->features + rows * SPX_FEATURE_COUNT : nullptr;
      t.spectrogram = taps->spectrogram ? taps->spectrogram + o_spec : nullptr;
      t.normalized = taps->normalized ? taps->normalized + o_norm : nullptr;
      const size_t fr = gj[g].empty() ? 0 : (size_t)layout_for(plans[g]->dev, gj[g].data(), (int)gj[g].size()).total_frames;
      rows += fr; o_spec += fr * (size_t)plans[g]->dev.N; o_norm += fr * (size_t)plans[g]->dev.W;
    }
  }
  // Kernels in sequence: the groups' analysis kernels one after the other, the cheapest first (lowest rate: fewest frames
  // and the shortest transform), instead of all at once.  Shared, every analysis ends late and every walk kernel starts
  // late; shortest first, the first group's walk starts early and the last analysis -- alone on what the running walk
  // kernels leave -- ends no later than it did shared (configs[4] shard: 16 kHz analysis done at 0.31 instead of 0.50 ms,
  // 22.05 kHz at 0.87 instead of 0.92; the call ends with the later group's walk kernel).
  std::vector<int> ord;
  for (int g = 0; g < n_plans; g++) if (!gj[g].empty()) ord.push_back(g);
  const bool chain_analyses = MM.chain_analyses;
  if (chain_analyses) std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return plans[a]->dev.rate < plans[b]->dev.rate; });
  // Which stream a group runs on.  Kernels in sequence (every mix measured so far): the first group on the CALLER's stream,
  // the second on the device's second side stream (idle in this mode), further groups on their plans' own streams -- so the
  // usual two-rate call keeps this library at its three streams per device whatever else the process has created.  HIP maps
  // streams onto a few hardware queues, a queue runs its kernels in order, and two groups whose streams share a queue run one
  // after the other: with a stream per plan, the configs[4] shard took 4.95 instead of 3.0 ms in every process that had run
  // a concurrent-mode call before (its two side streams had taken queues; profiles/r04/r04c_c4_prefix.txt).
  hipStream_t dev_s1 = nullptr, dev_s2 = nullptr;
  if (!concurrent && (ord.size() > 1 || ahead) && dev_side_streams(lead->device, &dev_s1, &dev_s2)) return fail(-1, "spx_batch_run_mixed: no side streams");
  std::vector<const int*> started(n_plans, nullptr);
  if (ahead) {
    // the producers' stream: behind the walk kernels of the lead plan's call before the previous one (the last user of this
    // workspace when two take turns; the previous call too if it used this workspace or another stream), and -- while the
    // previous call is still in flight -- behind gate kernels that wait until its walk workgroups have been placed
    // (no gates when the producers wait for the previous call anyway -- the same workspace handed over again: its counters are
    // the ones this call's staging kernels clear, and a gate would spin its full bound for counts nobody raises)
    std::lock_guard<std::mutex> ring_lock(lead->mu);
    bool waited_prev = false;
    if (ring_wait(lead, dev_s1, ws, st, &waited_prev)) return -2;
    const bool in_flight = !waited_prev && ring_previous_in_flight(lead);
    if (in_ready) HIPCHK(hipStreamWaitEvent(dev_s1, static_cast<hipEvent_t>(in_ready), 0));   // the caller's "input is there"
    // the job -> group-order table for the scatter kernel at the call's end: behind the ring's events (the scatter kernel of the
    // call that last used this workspace is behind them), in front of the groups' producers -- every walk kernel, and with them
    // the caller's stream, is ordered behind it through the tension events
    if (upload_idx(dev_s1)) return -2;
    if (in_flight)
      for (const auto& sn : lead->mixed_started)
        if (sn.first && sn.second > 0) hipLaunchKernelGGL(spx_gate_kernel, dim3(1), dim3(64), 0, dev_s1, sn.first, sn.second, 8000u);
  }
  hipEvent_t prev_an = nullptr;
  int launch_idx = 0;
  for (int g : ord) {
    spx_plan* p = plans[g];
    hipStream_t gs = nullptr;
    if (!concurrent && launch_idx == 0) gs = st;
    else if (!concurrent && launch_idx == 1) gs = dev_s2;
    else {
      if (!p->mix) HIPCHK(hipStreamCreateWithFlags(&p->mix, hipStreamNonBlocking));
      gs = p->mix;
    }
    launch_idx++;
    if (!p->ev_join) HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    if (!p->ev_an) HIPCHK(hipEventCreateWithFlags(&p->ev_an, hipEventDisableTiming));
    if (gs != st) HIPCHK(hipStreamWaitEvent(gs, lead->ev_fork, 0));
    SpxForce f = force;
    if (ahead) {
      f.ahead_sa = dev_s1;            // (the analyses follow one another on that stream by themselves, cheapest first)
      f.started_out = &started[g];
    } else if (chain_analyses) {
      if (prev_an) HIPCHK(hipStreamWaitEvent(gs, prev_an, 0));
      f.after_analysis = p->ev_an;
      prev_an = p->ev_an;
    }
    SpxCallOpts go;
    go.force = &f;
    rc = run_impl(p, gj[g].data(), (int)gj[g].size(), in, out, d_nout + gpos[g], w + M.ws_off[g], M.ws_bytes[g], taps ? &gtaps[g] : nullptr, gs,
                  true, true, go);
    // (also when the group failed: whatever it -- and the groups before it -- enqueued on their streams still reads the
    // caller's buffers, so the caller's stream waits for it before the error is returned)
    const std::string err = rc ? g_err : std::string();
    if (gs != st && (hipEventRecord(p->ev_join, gs) != hipSuccess || hipStreamWaitEvent(st, p->ev_join, 0) != hipSuccess)) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(gs);   // no event: make sure by waiting here
      if (!rc) return fail(-2, "spx_batch_run_mixed: joining a group's stream failed");
    }
    if (rc) return fail(rc, err);
  }
  hipLaunchKernelGGL(spx_scatter_nout_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_nout, d_idx, n, n_out);
  {
    // every mixed call leaves its end in the lead plan's ring (a pipelined call orders its producers behind the two calls before it)
    std::lock_guard<std::mutex> ring_lock(lead->mu);
    if (ring_record(lead, st, ws, st, out, n_out)) return -2;
    lead->ahead_started = nullptr;
    lead->ahead_n = 0;
    lead->mixed_started.clear();
    if (ahead)
      for (int g = 0; g < n_plans; g++)
        if (started[g]) lead->mixed_started.emplace_back(started[g], (int)gj[g].size());
  }
  if (concurrent) {
    if (!guard.last) HIPCHK(hipEventCreateWithFlags(&guard.last, hipEventDisableTiming));
    HIPCHK(hipEventRecord(guard.last, st));
    guard.last_stream = st;
    guard.valid = true;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// Pitch searches per stream of the batch's last run (SpxWalkState::steps): the length of every stream's dependent chain.
// Waits for hip_stream, then copies the state records out of the workspace.
static int read_steps(const SpxPlanDev& d, const spx_stream_job* jobs, int n, const void* ws, int32_t* steps, hipStream_t st) {
  const Layout L = layout_for(d, jobs, n);
  std::vector<SpxStreamState> h((size_t)n);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpy(h.data(), static_cast<const unsigned char*>(ws) + L.off_states, sizeof(SpxStreamState) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) steps[i] = h[i].w.steps;
  return 0;
}
int spx_batch_read_steps(spx_plan_t plan, const spx_stream_job* jobs, int n, const void* ws, int32_t* steps, void* hs) {
  if (!plan || !jobs || n <= 0 || !ws || !steps) return fail(-1, "spx_batch_read_steps: bad arguments");
  int k = 1;
  { std::lock_guard<std::mutex> g(plan->mu); auto it = plan->split_of.find(ws); if (it != plan->split_of.end()) k = it->second; }
  if (k <= 1) return read_steps(plan->dev, jobs, n, ws, steps, static_cast<hipStream_t>(hs));
  // the call that last ran on this workspace was split into sub-batches (run_split): their state records sit in their slices
  const SplitPlan P = split_geometry(plan, jobs, n, SPX_SPLIT_LIMIT);
  if (P.k != k) return fail(-1, "spx_batch_read_steps: the jobs are not those of the call that last ran on this workspace");
  for (int i = 0; i < P.k; i++) {
    const int a = P.first[i], m = P.first[i + 1] - a;
    const int rc = read_steps(plan->dev, jobs + a, m, static_cast<const unsigned char*>(ws) + P.ws_off[i], steps + a, static_cast<hipStream_t>(hs));
    if (rc) return rc;
  }
  return 0;
}
int spx_batch_read_steps_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n,
                               const void* ws, int32_t* steps, void* hs) {
  if (!plans || n_plans < 1 || n_plans > 8 || !jobs || n <= 0 || !ws || !steps) return fail(-1, "spx_batch_read_steps_mixed: bad arguments");
  std::vector<std::vector<spx_stream_job>> gj;
  std::vector<std::vector<int>> gi;
  int rc = mixed_groups(n_plans, jobs, plan_index, n, gj, gi);
  if (rc) return rc;
  const MixedLayout M = mixed_layout(plans, n_plans, gj, n);
  for (int g = 0; g < n_plans; g++) {
    if (gj[g].empty()) continue;
    std::vector<int32_t> sg(gj[g].size());
    rc = read_steps(plans[g]->dev, gj[g].data(), (int)gj[g].size(), static_cast<const unsigned char*>(ws) + M.ws_off[g], sg.data(),
                    static_cast<hipStream_t>(hs));
    if (rc) return rc;
    for (size_t k = 0; k < sg.size(); k++) steps[gi[g][k]] = sg[k];
  }
  return 0;
}

// Names of the kernels a batch of this shape is served by, as a profiler prints them (without "void" and the argument
// list): "analysis;tension;walk".  bench.py keys its roofline object and profiles/pmc_traffic.json with them.
static const char* kernel_names(spx_plan_t plan, int n_streams, int max_channels, int speedup_only, bool lean);
const char* spx_batch_kernel_names(spx_plan_t plan, int n_streams, int max_channels, int speedup_only) {
  return kernel_names(plan, n_streams, max_channels, speedup_only, false);
}
// ... with the walk kernel in its lean form (no output waves): what spx_batch_run_overlapped launches when three or more
// workspaces take turns, and the concurrent mode at 22.05 kHz mono
const char* spx_batch_kernel_names_lean(spx_plan_t plan, int n_streams, int max_channels, int speedup_only) {
  return kernel_names(plan, n_streams, max_channels, speedup_only, true);
}
static const char* kernel_names(spx_plan_t plan, int n_streams, int max_channels, int speedup_only, bool lean) {
  static thread_local char buf[256];
  const SpxPlanDev& d = plan->dev;
  const SpxWalkConfig c = spx_walk_config(d, n_streams, max_channels < 1 ? 1 : max_channels, speedup_only != 0, false, lean, speedup_only == 0);
  char walk[96];
  if (c.fast_kernel && c.slow)
    snprintf(walk, sizeof(walk), "spx_walk_fast_kernel<%d, %d, 0, 0, %d>", c.nwm, c.nwc >= 4 ? 4 : 0, max_channels > 1 ? 3 : 2);
  else if (c.fast_kernel)
    {
      const bool ct_rate = d.rate == 16000 || d.rate == 22050;
      const bool lng = ct_rate && c.nwm == 4 && c.nwc >= 4 && c.wcap == 8192;   // spx_launch_walk_fast's long-window instantiations
      snprintf(walk, sizeof(walk), "spx_walk_fast_kernel<%d, %d, %d, %d, %d>", c.nwm, c.nwc >= 4 ? 4 : (c.nwc >= 2 ? 2 : (c.nwc >= 1 ? 1 : 0)),
               (ct_rate && (lng || c.wcap == ((c.nwc == 0 && c.nwm <= 2) ? 1536 : 4096))) ? d.rate : 0, lng ? 1 : 0, max_channels > 1 ? 1 : 0);
    }
  else
    snprintf(walk, sizeof(walk), "spx_walk_kernel<%d, %d>", c.nw, c.mode);
  snprintf(buf, sizeof(buf), "spx_analysis_kernel<%d, %d>;spx_tension_kernel;%s", d.tile_frames, spx_analysis_ct_window(d), walk);
  return buf;
}

int spx_debug_last_call_concurrent(void) { return g_last_concurrent.load(std::memory_order_relaxed); }
int spx_debug_kernel_vgprs(int which) {
  if (which == 0) return spx_tension_vgprs();
  const int rate = (which == 2 || which == 4) ? 22050 : 16000;
  const SpxPlanDev* P = spx_internal_shared_plan(rate, 0);
  if (!P) return -1;
  switch (which) {
    case 1: return spx_walk_vgprs(*P, 256, 1, true, false);
    case 2: return spx_walk_vgprs(*P, 256, 1, true, true);
    case 3: case 4: return spx_analysis_vgprs(*P);
    case 5: return spx_walk_vgprs(*P, 256, 2, true, false);
    default: return -1;
  }
}

// Diagnostics: what spx_launch_walk would launch for a batch of this shape, and what that kernel costs -- out[0] allocated
// VGPRs (rounded up to the granule of 8), out[1] scratch bytes per lane (spilled registers), out[2] LDS bytes per workgroup,
// out[3] the form (16 * search waves + output waves; 0 = the general kernel), out[4] waves per workgroup.
int spx_debug_walk_info(int sample_rate, int channels, int n_streams, int speedup_only, int short_jobs, int lean, int* out) {
  const SpxPlanDev* P = spx_internal_shared_plan(sample_rate, 0);
  if (!P || !out) return -1;
  const SpxWalkConfig c = spx_walk_config(*P, n_streams, channels < 1 ? 1 : channels, speedup_only != 0, short_jobs != 0, lean != 0, speedup_only == 0);
  int scratch = -1;
  out[0] = spx_walk_kernel_regs(*P, n_streams, channels, speedup_only != 0, short_jobs != 0, lean != 0, &scratch, speedup_only == 0);
  out[1] = scratch;
  out[2] = (int)c.lds;
  out[3] = c.fast_kernel ? 16 * c.nwm + c.nwc : 0;
  out[4] = c.waves;
  return 0;
}
// ... and the same for the analysis kernel of a rate (out[0] VGPRs, out[1] scratch bytes, out[2] LDS bytes) and the tension kernel
int spx_debug_analysis_info(int sample_rate, int* out) {
  const SpxPlanDev* P = spx_internal_shared_plan(sample_rate, 0);
  if (!P || !out) return -1;
  int scratch = -1;
  out[0] = spx_analysis_vgprs(*P, &scratch);
  out[1] = scratch;
  out[2] = (int)spx_analysis_lds_bytes(*P);
  return 0;
}

void spx_set_timing(int enabled) { g_timing = enabled != 0; }
void spx_set_concurrent(int on) { g_concurrent = on != 0; }
void spx_set_pipeline_chunks(int chunks) { g_chunks_set = true; g_chunks = chunks < 1 ? 1 : (chunks > SPX_MAX_CHUNKS ? SPX_MAX_CHUNKS : chunks); }
static double g_last_tension_ms = 0.0;
double spx_timing_last_tension_ms(void) { return g_last_tension_ms; }
int spx_timing_collect(double* sum_ms_analyze, double* sum_ms_walk, int* n_calls) {
  std::lock_guard<std::mutex> g(g_tmu);
  double a = 0, w = 0, t = 0;
  for (auto& ev : g_ev_pending) {
    HIPCHK(hipEventSynchronize(ev.b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (ev.kind == 0) a += ms; else if (ev.kind == 1) w += ms; else t += ms;
    g_ev_free.push_back(ev.a);
    g_ev_free.push_back(ev.b);
  }
  g_ev_pending.clear();
  g_last_tension_ms = t;
  if (sum_ms_analyze) *sum_ms_analyze = a;
  if (sum_ms_walk) *sum_ms_walk = w;
  if (n_calls) *n_calls = g_calls_pending;
  g_calls_pending = 0;
  return 0;
}

void* spx_device_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(-2, "spx_device_alloc failed"); return nullptr; }
  return p;
}
void spx_device_free(void* p) { if (p) (void)hipFree(p); }
int spx_copy_to_device(void* dst, const void* src, size_t bytes, void* hs) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(hs)));
  return 0;
}
}  // extern "C"

// ---- output packing: offsets by one workgroup (serial prefix over <= a few thousand streams), then one workgroup per
// stream copying its frames with coalesced loads/stores ----
__global__ void __launch_bounds__(256)
spx_pack_offsets_kernel(const int64_t* __restrict__ n_out, const int* __restrict__ channels,
                        const int64_t* __restrict__ caps, int n, int64_t* __restrict__ offsets) {
  // exclusive prefix sum of the streams' element counts: 256 streams per pass, Hillis-Steele scan in LDS, running carry
  __shared__ int64_t sh[256];
  __shared__ int64_t carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + t;
    int64_t v = 0;
    if (i < n) {
      const int64_t k = n_out[i];
      // a negative count flags an overflowed stream: the frames that fitted its capacity are there, the count says how
      // many there would have been; INT64_MIN a lost producer (nothing)
      int64_t f = (k == INT64_MIN ? 0 : (k > 0 ? k : -k));
      if (f > caps[i]) f = caps[i];
      v = f * channels[i];
    }
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int64_t add = (t >= d) ? sh[t - d] : 0;
      __syncthreads();
      sh[t] += add;
      __syncthreads();
    }
    const int64_t base = carry;
    if (i < n) offsets[i] = base + sh[t] - v;
    __syncthreads();
    if (t == 255) carry = base + sh[255];
    __syncthreads();
  }
  if (t == 0) offsets[n] = carry;
}
__global__ void __launch_bounds__(256)
spx_pack_copy_kernel(const int16_t* __restrict__ out, const int64_t* __restrict__ out_offs,
                     const int64_t* __restrict__ offsets, int16_t* __restrict__ packed) {
  // 4 workgroups per stream (blockIdx.y), eight 2-byte loads in flight per thread: the copy is latency-bound otherwise
  // (one load per thread at a time took 0.3 ms for the bench batch's 26.7 MB)
  const int i = blockIdx.x;
  const int16_t* src = out + out_offs[i];
  int16_t* dst = packed + offsets[i];
  const int64_t cnt = offsets[i + 1] - offsets[i];
  const int64_t stride = (int64_t)gridDim.y * 256 * 8;
  for (int64_t e0 = ((int64_t)blockIdx.y * 256 + threadIdx.x); e0 < cnt; e0 += stride) {
    int16_t v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int64_t e = e0 + (int64_t)u * gridDim.y * 256; v[u] = (e < cnt) ? src[e] : (int16_t)0; }
#pragma unroll
    for (int u = 0; u < 8; u++) { const int64_t e = e0 + (int64_t)u * gridDim.y * 256; if (e < cnt) dst[e] = v[u]; }
  }
}

extern "C" {
int spx_batch_pack_outputs(const spx_stream_job* jobs, int n, const int16_t* out, const int64_t* n_out, int16_t* packed,
                           int64_t* offsets, void* hs) {
  if (!jobs || n <= 0 || !out || !n_out || !packed || !offsets) return fail(-1, "spx_batch_pack_outputs: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(hs);
  // small per-stream tables (channels, output offsets): a stream-ordered allocation, freed in stream order after the
  // kernels that read it, so concurrent calls on other streams never share it
  const size_t need = (size_t)n * (2 * sizeof(int64_t) + sizeof(int));
  void* d_tab = nullptr;
  if (hipMallocAsync(&d_tab, need, st) != hipSuccess) return fail(-2, "spx_batch_pack_outputs: allocation failed");
  // host side of the table: a per-thread pinned slot, reused once the copy that last read it has retired
  static thread_local SpxStage G;
  if (G.done) HIPCHK(hipEventSynchronize(G.done));
  else HIPCHK(hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
  if (G.cap < need) {
    if (G.p) (void)hipHostFree(G.p);
    G.p = nullptr; G.cap = 0;
    HIPCHK(hipHostMalloc(&G.p, need * 2 + 1024, hipHostMallocDefault));
    G.cap = need * 2 + 1024;
  }
  unsigned char* h = static_cast<unsigned char*>(G.p);
  int64_t* h_off = reinterpret_cast<int64_t*>(h);
  int64_t* h_cap = h_off + n;
  int* h_ch = reinterpret_cast<int*>(h + (size_t)n * 2 * sizeof(int64_t));
  for (int i = 0; i < n; i++) { h_off[i] = jobs[i].out_off; h_cap[i] = jobs[i].out_cap; h_ch[i] = jobs[i].channels; }
  HIPCHK(hipMemcpyAsync(d_tab, h, need, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(G.done, st));
  const int64_t* d_off = reinterpret_cast<const int64_t*>(d_tab);
  const int* d_ch = reinterpret_cast<const int*>(static_cast<unsigned char*>(d_tab) + (size_t)n * 2 * sizeof(int64_t));
  hipLaunchKernelGGL(spx_pack_offsets_kernel, dim3(1), dim3(256), 0, st, n_out, d_ch, d_off + n, n, offsets);
  hipLaunchKernelGGL(spx_pack_copy_kernel, dim3(n, 4), dim3(256), 0, st, out, d_off, offsets, packed);
  (void)hipFreeAsync(d_tab, st);
  HIPCHK(hipGetLastError());
  return 0;
}

int spx_copy_to_host(void* dst, const void* src, size_t bytes, void* hs) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(hs)));
  return 0;
}
int spx_stream_synchronize(void* hs) {
  HIPCHK(hipStreamSynchronize(static_cast<hipStream_t>(hs)));
  return 0;
}

}  // extern "C"
