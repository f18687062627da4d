// C-ABI of the batch engine (include/speedy_hip.h): one batch call -- workspace layout, staging, streams, the ring of earlier calls, launches (plans: spx_plan.hip; mixed-rate calls: spx_mixed.hip).
#include "spx_engine.h"

// HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and a queue runs its kernels
// in order.  This library alone holds up to five streams per device that must not block one another (the caller's or the
// pipeline object's run stream, the side stream, two walk streams, the pipeline's copy stream): with four queues two of them
// share one, which two depends on the order of creation, and the end-to-end rate then read 2.0, 2.6 or 4.3 ms per batch in round 4
// (profiles/r04/r05u_hw_queues.txt).  The runtime reads the variable at its first call, so the library asks for eight queues when
// it is LOADED -- unless the application has set the variable itself (never overridden), or says SPX_KEEP_HW_QUEUES=1.  A process
// that has made HIP calls before loading the library keeps what it had: INTEGRATION.md "Environment".
__attribute__((constructor(101))) static void spx_default_hw_queues() {
  if (getenv("SPX_KEEP_HW_QUEUES") || getenv("GPU_MAX_HW_QUEUES")) return;
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  // (said once where somebody asked to be told what the library decides: it changed the process's environment)
  if (getenv("SPX_DEBUG_MODE")) fprintf(stderr, "[spx] GPU_MAX_HW_QUEUES was unset: set to 8 at load time (SPX_KEEP_HW_QUEUES=1 leaves the environment alone)\n");
}

thread_local std::string g_spx_err;

// Timing: one set of four HIP events per timed call, recorded on the launch stream and resolved lazily by
// spx_timing_collect (so that the timed region itself carries no host synchronisation).
// Process-wide switches are atomics; the event lists behind spx_timing_collect are guarded by g_tmu.  spx_batch_run may be
// called from several host threads (one plan per thread, or one plan shared: launches on a plan are serialised by its mutex).
std::atomic<bool> g_timing{false};
std::atomic<int> g_last_concurrent{0};   // spx_debug_last_call_concurrent (2 = pipelined with the previous call)
std::atomic<int> g_concurrent{1};  // spx_set_concurrent: analysis and walk kernels on two streams, tile-flag hand-off
std::atomic<bool> g_chunks_set{false};  // the caller chose a chunk count (spx_set_pipeline_chunks)
std::atomic<int> g_chunks{1};  // spx_set_pipeline_chunks (measured on MI355X, 256 x 10 s: 1 -> 5.09 ms, 2 -> 5.25, 4 -> 5.54 per step)
std::mutex g_tmu;
std::vector<EvPair> g_ev_pending;
std::vector<hipEvent_t> g_ev_free;
int g_calls_pending = 0;

// Concurrent mode keeps polling workgroups resident; its deadlock-freedom bound (run_impl) counts the streams of ONE
// call.  A second concurrent-mode call in flight on the same device (another plan, thread or stream) would break it, so
// per device the last concurrent call leaves an event behind, and a call that finds it unfinished on a different stream
// takes the sequential launch order instead (same results).  Calls on the same stream are ordered by the stream.
SpxDevGuard g_guard[64];

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

extern "C" {

}  // extern "C"
Layout layout_for(const SpxPlanDev& d, const spx_stream_job* jobs, int n) {
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
void spx_launch_gate(const int* started, int n_walk, unsigned max_spins, hipStream_t st) {
  hipLaunchKernelGGL(spx_gate_kernel, dim3(1), dim3(64), 0, st, started, n_walk, max_spins);
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
int dev_side_streams(int dev, hipStream_t* side, hipStream_t* side2) {
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
int walk_stream_count() {
  static const int n = [] {
    const char* e = spx_tuning_env("SPX_WALK_STREAMS");
    if (!e && spx_tuning_env("SPX_WALK_STREAMS3")) return 3;
    const int v = e ? atoi(e) : 2;
    return v < 1 ? 1 : (v > SPX_MAX_WALK_STREAMS ? SPX_MAX_WALK_STREAMS : v);
  }();
  return n;
}
int dev_walk_streams(int dev, hipStream_t* w, int n) {
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
int ring_note(spx_plan* plan, hipStream_t st) {
  const int cur = plan->ahead_calls & 1;
  if (!plan->ev_call[cur]) HIPCHK(hipEventCreateWithFlags(&plan->ev_call[cur], hipEventDisableTiming));
  HIPCHK(hipEventRecord(plan->ev_call[cur], st));
  plan->ev_call_valid[cur] = true;
  plan->ev_call_st[cur] = st;
  return 0;
}
int ring_wait(spx_plan* plan, hipStream_t sa, const void* ws, hipStream_t st, bool* waited_prev) {
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
bool ring_previous_in_flight(spx_plan* plan) {
  const int j = (plan->ahead_calls + SPX_RING - 1) % SPX_RING;
  const bool f = plan->ev_walk_valid[j] && hipEventQuery(plan->ev_walk[j]) == hipErrorNotReady;
  (void)hipGetLastError();
  return f;
}
int ring_record(spx_plan* plan, hipStream_t on, void* ws, hipStream_t st, const void* out, const void* n_out, bool detached) {
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
SpxModeEnv mode_env() {
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
bool device_ours_cb(void* ctx) { return device_is_ours(*static_cast<int*>(ctx)); }
// What kind of speeds a batch's jobs bring: speedup_only -- every job speeds up (the walk kernel specialised for speeds >= 1
// applies); any_speed -- not all do, but every speed is one the speed-up kernel's slow-down instantiations take (round 5).
SpxSpeedClass speed_class(const spx_stream_job* jobs, int n) {
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
SpxModeWalk mode_walk(const SpxPlanDev& d, int n, int maxC, bool speedup_only, bool lean, bool any_speed, bool short_window) {
  const SpxWalkConfig c = spx_walk_config(d, n, maxC, speedup_only, false, lean, any_speed, short_window);
  SpxModeWalk w;
  w.lds = c.lds; w.waves = c.waves; w.fast_kernel = c.fast_kernel; w.nwc = c.nwc;
  w.vgprs = spx_walk_vgprs(d, n, maxC, speedup_only, lean, any_speed, short_window);
  return w;
}
static bool mixed_long_window_env() {   // A/B (tuning build): overlapped mixed calls keep the 8192-frame window
  static const bool v = spx_tuning_env("SPX_MIXED_LONG_WINDOW") != nullptr;
  return v;
}
// (cached per plan and shape: the register queries and spx_walk_config are not free, and the engine asks on every call)
// (short_window: the walk kernel as an overlapped mixed call launches it -- SpxForce::no_exclusive)
static const SpxModeResources& mode_resources(spx_plan* plan, int n, int maxC, bool speedup_only, bool any_speed = false, bool short_window = false) {
  const long long key = ((long long)n << 16) ^ ((long long)maxC << 3) ^ (short_window ? 4 : 0) ^ (any_speed ? 2 : 0) ^ (speedup_only ? 1 : 0);
  auto it = plan->res_cache.find(key);
  if (it != plan->res_cache.end()) return it->second;
  const SpxPlanDev& d = plan->dev;
  SpxModeResources R;
  memset(&R, 0, sizeof(R));
  R.cu_count = plan->cu_count;
  R.lds_per_cu = plan->lds_per_cu;
  R.walk = mode_walk(d, n, maxC, speedup_only, false, any_speed, short_window);
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

// One batch call: decide the launch mode (spx_choose_mode, a pure function of the inputs collected here), then execute it.
// The three launch orders give the same results:
//   in sequence  -- staging, analysis, tension, walk on the caller's stream (time chunks: the analysis of chunk c + 1 on a side
//                   stream beside the walk of chunk c);
//   concurrent   -- analysis and tension kernels on the device's two side streams, the walk kernel at once on the caller's:
//                   tiles of frames, then speeds, are handed over through flags the consumers poll;
//   ahead        -- staging, analysis and tension kernels on the side stream AT ONCE, beside the previous call's walk kernel;
//                   the walk kernel behind them on the caller's stream, or (walk2) on one of the library's two walk streams.
int run_impl(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out,
             int64_t* n_out, void* ws, size_t ws_bytes, const spx_taps* taps, void* hs, bool do_a,
             bool do_w, const SpxCallOpts& opt) {
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
  // a call that is not a sub-batch of run_split's leaves no split record for its workspace (spx_batch_analyze / _walk and mixed
  // calls come here directly: a stale record made spx_batch_read_steps look for sub-batch slices -- ADVICE r5)
  if (!opt.split_part) plan->split_of.erase(ws);
  // The walk kernel's window: 4096 frames instead of the long one (36.9 instead of 69.7 KB of LDS per stream) for calls whose walk
  // workgroups are to share CUs with another call's AND with analysis workgroups, and that have no lean form to take:
  //   - the groups of an overlapped mixed call (SpxForce::no_exclusive): decided before the mode, the co-residency arithmetic (and
  //     with it the tile: the 22.05 kHz group fell back to the 8-frame tile with the long window's 70.5 KB) is done with that form;
  //   - multi-channel batches whose walk kernels overlap the previous call's (walk2 below; 16 kHz stereo x 256 through the pipeline
  //     object: 1.33 -> 1.21 ms per batch): at the LAUNCH only -- the arithmetic stays with the long window's numbers, which errs on
  //     the safe side (less LDS than counted) and leaves every decision where it was (done with the short window's, 22.05 kHz stereo
  //     turns "doubtful", goes to the timed trial and ends in the concurrent mode: 1.69 -> 2.9 ms through the pipeline object).
  const bool short_window_res = opt.force && opt.force->no_exclusive && !mixed_long_window_env();
  const SpxModeResources& R = mode_resources(plan, n, maxC, speedup_only, any_speed, short_window_res);
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
  const bool chunk_ahead = M.chunk_ahead && !force;   // (producers on the side stream at once; walk chunks on the caller's as always)
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
            concurrent ? "concurrent" : (ahead ? (M.seq_ahead ? "ahead (kernels in sequence)" : "ahead") : (chunk_ahead ? "sequence, producers ahead" : "sequence")),
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
  if ((ahead || chunk_ahead) && !force) {
    // this call's producers must not touch a workspace the walk kernel of an earlier call still reads (ring_wait)
    if (ring_wait(plan, sa, ws, st, &waited_prev)) return -2;
    if (opt.in_ready) HIPCHK(hipStreamWaitEvent(sa, static_cast<hipEvent_t>(opt.in_ready), 0));
  }
  if (!ahead && !chunk_ahead && opt.in_ready) HIPCHK(hipStreamWaitEvent(st, static_cast<hipEvent_t>(opt.in_ready), 0));
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
                    (concurrent || ahead) ? (unsigned)n + 1u : 0u, (ahead || chunk_ahead) ? sa : st, &staged_ev,
                    (ahead_gate && !split_gate) ? plan->ahead_started : nullptr, plan->ahead_n, 8000u);
  if (rc) return rc;
  if (M.trial_slot >= 0) {  // bracket this call on the caller's stream (spx_plan::Trial)
    hipEvent_t& e0 = TR.ev[2 * M.trial_slot];
    if (!e0) HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventRecord(e0, st));
  }
  if (sa != st && !ahead && !chunk_ahead) {
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
                        speedup_only, stw, false, (M.exclusive_cu && !(force && force->no_exclusive)) ? R.lds_per_cu / 2 + 1024 : 0, M.launch_lean, any_speed,
                        short_window_res || (!force && M.walk2 && !M.launch_lean && maxC > 1 && !mixed_long_window_env()));
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
  if (P.k <= 1) return run_impl(plan, jobs, n, in, out, n_out, ws, ws_bytes, taps, hs, true, true, opt);
  { std::lock_guard<std::mutex> g(plan->mu); if (plan->split_of.size() > 64) plan->split_of.clear(); plan->split_of[ws] = P.k; }
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
    o.split_part = true;
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

}  // extern "C"
// Pitch searches per stream of the batch's last run (SpxWalkState::steps): the length of every stream's dependent chain.
// Waits for hip_stream, then copies the state records out of the workspace.
int spx_read_steps(const SpxPlanDev& d, const spx_stream_job* jobs, int n, const void* ws, int32_t* steps, hipStream_t st) {
  const Layout L = layout_for(d, jobs, n);
  std::vector<SpxStreamState> h((size_t)n);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpy(h.data(), static_cast<const unsigned char*>(ws) + L.off_states, sizeof(SpxStreamState) * (size_t)n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) steps[i] = h[i].w.steps;
  return 0;
}
extern "C" {
int spx_batch_read_steps(spx_plan_t plan, const spx_stream_job* jobs, int n, const void* ws, int32_t* steps, void* hs) {
  if (!plan || !jobs || n <= 0 || !ws || !steps) return fail(-1, "spx_batch_read_steps: bad arguments");
  int k = 1;
  { std::lock_guard<std::mutex> g(plan->mu); auto it = plan->split_of.find(ws); if (it != plan->split_of.end()) k = it->second; }
  if (k <= 1) return spx_read_steps(plan->dev, jobs, n, ws, steps, static_cast<hipStream_t>(hs));
  // the call that last ran on this workspace was split into sub-batches (run_split): their state records sit in their slices
  const SplitPlan P = split_geometry(plan, jobs, n, SPX_SPLIT_LIMIT);
  if (P.k != k) return fail(-1, "spx_batch_read_steps: the jobs are not those of the call that last ran on this workspace");
  for (int i = 0; i < P.k; i++) {
    const int a = P.first[i], m = P.first[i + 1] - a;
    const int rc = spx_read_steps(plan->dev, jobs + a, m, static_cast<const unsigned char*>(ws) + P.ws_off[i], steps + a, static_cast<hipStream_t>(hs));
    if (rc) return rc;
  }
  return 0;
}
}  // extern "C"
