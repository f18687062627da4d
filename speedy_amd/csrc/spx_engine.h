// Host side of the batch engine, shared by its translation units (round 6: spx_engine.hip was 1 800 lines):
//   spx_plan.hip    plan tables (DFT spec twiddles, window, Rader), sizes and capacities
//   spx_engine.hip  one batch call: workspace layout, staging, the library's streams, the plan's ring of earlier calls,
//                   run_impl (decide with spx_mode.h, then launch), sub-batches
//   spx_mixed.hip   one call over several plans (BASELINE configs[4])
//   spx_diag.hip    kernel names, resource queries, timing collection, output packing, small copies
// Nothing here is part of the C-ABI (include/speedy_hip.h).
#pragma once
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

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif


// the calling thread's last error (spx_last_error)
extern thread_local std::string g_spx_err;
static inline int fail(int code, const std::string& msg) {
  g_spx_err = msg;
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
#ifdef SPX_TUNING
#define SPX_MAX_WALK_STREAMS 8   // (the developers' build: up to eight walk launches in flight, SPX_WALK_STREAMS)
#else
#define SPX_MAX_WALK_STREAMS 4
#endif
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
  SpxStage mix_stage[2];         // (two, taking turns: a pipelined mixed call's upload sits behind the batch's H2D copy, and waiting
  int mix_next = 0;              //  for the one slot's event made the host wait for that copy -- ADVICE r5)
};

// ---- process-wide state (spx_engine.hip) ----
struct EvPair { hipEvent_t a, b; int kind; };  // kind 0 = analysis launch, 1 = walk launch, 2 = tension launch
extern std::atomic<bool> g_timing;
extern std::atomic<int> g_last_concurrent;   // spx_debug_last_call_concurrent (2 = pipelined with the previous call)
extern std::atomic<int> g_concurrent;        // spx_set_concurrent
extern std::atomic<bool> g_chunks_set;       // the caller chose a chunk count (spx_set_pipeline_chunks)
extern std::atomic<int> g_chunks;
extern std::mutex g_tmu;
extern std::vector<EvPair> g_ev_pending;
extern std::vector<hipEvent_t> g_ev_free;
extern int g_calls_pending;
struct SpxDevGuard {
  std::mutex mu;
  hipEvent_t last = nullptr;
  hipStream_t last_stream = nullptr;
  bool valid = false;
};
extern SpxDevGuard g_guard[64];

// frame j is sent to the analysis once sample j*B + W has been written (soniclib.c:440-444)
static inline int64_t frames_for(const SpxPlanDev& d, int64_t n_in) {
  if (n_in < d.W + 1) return 0;
  return (n_in - d.W - 1) / d.B + 1;
}
spx_plan* shared_plan_full(int sample_rate, int match_matlab);   // spx_plan.hip: one plan per (device, rate, hysteresis mode), cached

// ---- spx_engine.hip ----
struct Layout {
  size_t off_streams, off_states, off_rec, off_scratch, off_order, off_flags, off_ready, total;
  int64_t max_tiles;
  int64_t total_frames;
};
Layout layout_for(const SpxPlanDev& d, const spx_stream_job* jobs, int n);
int dev_side_streams(int dev, hipStream_t* side, hipStream_t* side2);
int walk_stream_count();
int dev_walk_streams(int dev, hipStream_t* w, int n);
int ring_note(spx_plan* plan, hipStream_t st);
int ring_wait(spx_plan* plan, hipStream_t sa, const void* ws, hipStream_t st, bool* waited_prev);
bool ring_previous_in_flight(spx_plan* plan);
int ring_record(spx_plan* plan, hipStream_t on, void* ws, hipStream_t st, const void* out, const void* n_out, bool detached = false);
SpxModeEnv mode_env();
bool device_ours_cb(void* ctx);
struct SpxSpeedClass { int maxC; bool speedup_only, any_speed; };
SpxSpeedClass speed_class(const spx_stream_job* jobs, int n);
SpxModeWalk mode_walk(const SpxPlanDev& d, int n, int maxC, bool speedup_only, bool lean, bool any_speed = false, bool short_window = false);
void spx_launch_gate(const int* started, int n_walk, unsigned max_spins, hipStream_t st);   // the idle-start / previous-call gate kernel
// `force`: the call is one group of a mixed-rate batch (spx_batch_run_mixed): the launch mode was decided for all groups
// together, and the device guard is held by the caller.
//   ahead_sa: the group of a pipelined mixed call -- its producers go to this stream at once (spx_batch_run_mixed_ahead orders it);
//   started_out: where the group's walk workgroups count themselves in (for the next call's gate)
//   total_streams: of all groups of the mixed call; after_analysis: recorded behind the group's analysis launch (or null)
struct SpxForce { int concurrent; bool idle_start; int total_streams; hipEvent_t after_analysis; hipStream_t ahead_sa; const int** started_out;
                  bool no_exclusive;   // the walk workgroups do NOT ask for a CU of their own and take the 4096-frame window (a mixed call whose walk
                                       // kernels overlap the previous call's: two calls' walk workgroups and an analysis workgroup on a CU)
};
struct SpxCallOpts {
  const SpxForce* force = nullptr;
  bool ahead_req = false;      // spx_batch_run_ahead: pipelined with the plan's previous call where the shape allows
  bool overlap_req = false;    // spx_batch_run_overlapped: ... and its walk kernel beside the previous call's
  void* in_ready = nullptr;    // hipEvent_t: the producers wait for it (the caller's "input is there")
  bool split_part = false;     // one of the sub-batches run_split cut the call into (spx_batch_read_steps finds their state records by it)
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
int run_impl(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes,
             const spx_taps* taps, void* hs, bool do_a, bool do_w, const SpxCallOpts& opt = SpxCallOpts());
int spx_read_steps(const SpxPlanDev& d, const spx_stream_job* jobs, int n, const void* ws, int32_t* steps, hipStream_t st);
