// spx_pipeline (include/speedy_hip.h): batch after batch of one shape, host memory to host memory, with everything in between
// owned by the library -- `depth` sets of device buffers, the pinned output buffers, two HIP streams, the events.
//
// What one submit enqueues (nothing waits on the host except for the batch `depth` tickets back):
//   copy stream   wait(kernels of the batch that last used this buffer set)  ->  H2D of the input  ->  record(in)
//   library       the batch call in its overlapped order (spx_engine.hip): staging, analysis, tension kernels behind `in` on the
//                 device's side stream, the walk kernel on one of the two walk streams
//   run stream    [wait(walk kernel): the engine's]  ->  gather kernel: every stream's produced frames, densely packed at 64-byte
//                 boundaries, written STRAIGHT into the slot's pinned host buffer, with the offsets and the counts  ->  record(done)
// The reference's caller owns one buffer and one loop (speedy_wave.cc:154-242: write a chunk, read what is ready); this is that
// loop for a caller whose unit is a batch of streams.
//
// Why the gather kernel writes host memory itself: the produced size is known on the device only.  A device-to-host copy of the
// exact size needs the host to wait for the batch first (round 4's bench loop did, and the wait kept it from running ahead:
// 2.3-3.5 ms per batch with the pipelined calls); a copy of the capacity moves three times the bytes.  The kernel needs neither,
// and it is narrow (SPX_PIPE_PACK_WGS workgroups): PCIe writes are posted, a few waves keep the link busy, and the CUs stay with the
// chains of the next batches' walk kernels -- the runtime's own copy kernel is launched full-width.
// Measured against it (round 5, profiles/r05/r5s_copy_out.txt): the gathered frames to HBM first and out by hipMemcpyAsync, sized
// from the last batch that came back (the rest of a batch that outgrew the copy fetched at wait time) -- the runtime performs
// that copy with its full-width shader kernel (__amd_rocclr_copyBuffer, 0.53 - 0.56 ms per 27 MB), and the leg read 1.78 - 1.83 ms
// per batch against 1.67 - 1.70.  Not kept.
#include <string.h>

#include <string>
#include <vector>

#include "../../include/speedy_hip.h"
#include "spx_internal.h"

int spx_internal_run(spx_plan_t plan, const spx_stream_job* jobs, int n, const int16_t* in, int16_t* out, int64_t* n_out, void* ws,
                     size_t ws_bytes, const spx_taps* taps, void* hs, bool ahead, bool overlap, void* in_ready, void* done_event, bool detached);
int spx_internal_run_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n, const int16_t* in,
                           int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, void* hs, bool ahead, void* in_ready,
                           void* done_event, bool detached);
void spx_internal_set_error(const char* msg);

#ifdef SPX_TUNING
#define SPX_PIPE_MAX_DEPTH 24   // (the developers' build: deeper pipelines for the walk-stream experiments)
#else
#define SPX_PIPE_MAX_DEPTH 8
#endif
#define PIPE_ALIGN 32   // int16 values: every stream's region starts at a 64-byte boundary, in device and in host memory

// Offsets of the packed output: exclusive prefix sums of the streams' produced values, each rounded up to PIPE_ALIGN; written
// to device memory (for the copy kernel) and, with the counts, to the slot's pinned host table.  One workgroup.
__global__ void __launch_bounds__(256)
spx_pipe_offsets_kernel(const int64_t* __restrict__ n_out, const int* __restrict__ channels, const int64_t* __restrict__ caps, int n,
                        int64_t* __restrict__ d_offsets, int64_t* __restrict__ h_offsets, int64_t* __restrict__ h_counts) {
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
      h_counts[i] = k;
      // a negative count flags an overflowed stream: the frames that fitted its capacity are there; INT64_MIN a lost producer
      int64_t f = (k == INT64_MIN ? 0 : (k > 0 ? k : -k));
      if (f > caps[i]) f = caps[i];
      v = (f * channels[i] + (PIPE_ALIGN - 1)) & ~(int64_t)(PIPE_ALIGN - 1);
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
    if (i < n) { const int64_t o = base + sh[t] - v; d_offsets[i] = o; h_offsets[i] = o; }
    __syncthreads();
    if (t == 255) carry = base + sh[255];
    __syncthreads();
  }
  if (t == 0) { d_offsets[n] = carry; h_offsets[n] = carry; }
}
// The copy: workgroup b takes streams b, b + gridDim.x, ...; 16 bytes per lane and load, four loads in flight (source and
// destination regions start at 64-byte boundaries and are padded to them: the last vector of a stream may carry up to 31 values
// of the capacity region behind the produced frames -- inside the buffers on both sides).
__global__ void __launch_bounds__(256)
spx_pipe_copy_kernel(const int16_t* __restrict__ out, const int64_t* __restrict__ out_offs, const int64_t* __restrict__ d_offsets, int n,
                     int16_t* __restrict__ dst) {
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const uint4* __restrict__ s = reinterpret_cast<const uint4*>(out + out_offs[i]);
    uint4* __restrict__ d = reinterpret_cast<uint4*>(dst + d_offsets[i]);
    const int64_t nv = (d_offsets[i + 1] - d_offsets[i]) / 8;   // 16-byte vectors
    for (int64_t e0 = threadIdx.x; e0 < nv; e0 += 4 * 256) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int64_t e = e0 + u * 256; if (e < nv) v[u] = s[e]; }
#pragma unroll
      for (int u = 0; u < 4; u++) { const int64_t e = e0 + u * 256; if (e < nv) d[e] = v[u]; }
    }
  }
}

struct SpxPipeSlot {
  int16_t* d_in = nullptr;
  int16_t* d_out = nullptr;
  int64_t* d_nout = nullptr;
  int64_t* d_offsets = nullptr;   // [n + 1] packed offsets (device copy, for the gather kernel)
  void* ws = nullptr;
  int16_t* h_in = nullptr;        // pinned staging a caller may fill (spx_pipeline_host_input), allocated on first use
  int16_t* h_out = nullptr;       // pinned: the packed output
  int64_t* h_meta = nullptr;      // pinned: offsets[n + 1], counts[n]
  hipEvent_t ev_in = nullptr;     // the input has arrived in d_in
  hipEvent_t ev_done = nullptr;   // kernels and gather done: the output is in host memory, d_in / d_out may be reused
  int64_t ticket = -1;
  bool in_recorded = false;       // ev_in has been recorded at least once (h_in / d_in have a copy to wait for)
  bool in_host = false;           // this ticket's input came from host memory: consumed once ev_in has passed (device input: ev_done)
};
struct spx_pipeline {
  std::vector<spx_plan_t> plans;
  std::vector<int> plan_index;
  bool mixed = false;
  int n = 0, depth = 0;
  unsigned flags = 0;
  int device = 0;
  std::vector<spx_stream_job> jobs;   // the caller's, with the pipeline's own out_off / out_cap
  std::vector<int64_t> static_offsets;   // SPX_PIPELINE_DEVICE_OUT: the fixed layout of d_out (offsets[n] = its extent)
  size_t in_values = 0, out_values = 0, ws_bytes = 0;
  hipStream_t s_h2d = nullptr, s_run = nullptr;
  int64_t* d_tab = nullptr;           // out_off[n], cap[n] (int64), channels[n] (int)
  std::vector<SpxPipeSlot> slots;
  int64_t next_ticket = 0;
  int pack_wgs = 64;
};

static int pfail(int code, const std::string& msg) { spx_internal_set_error(msg.c_str()); return code; }
#define PCHK(expr)                                                                             \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return pfail(-2, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

static void pipeline_free(spx_pipeline* p) {
  if (!p) return;
  (void)hipStreamSynchronize(p->s_run);
  if (p->s_h2d) (void)hipStreamSynchronize(p->s_h2d);
  for (auto& S : p->slots) {
    if (S.ev_done) { if (S.ticket >= 0) (void)hipEventSynchronize(S.ev_done); (void)hipEventDestroy(S.ev_done); }
    if (S.ev_in) (void)hipEventDestroy(S.ev_in);
    if (S.d_in) (void)hipFree(S.d_in);
    if (S.d_out) (void)hipFree(S.d_out);
    if (S.d_nout) (void)hipFree(S.d_nout);
    if (S.d_offsets) (void)hipFree(S.d_offsets);
    if (S.ws) (void)hipFree(S.ws);
    if (S.h_in) (void)hipHostFree(S.h_in);
    if (S.h_out) (void)hipHostFree(S.h_out);
    if (S.h_meta) (void)hipHostFree(S.h_meta);
  }
  if (p->d_tab) (void)hipFree(p->d_tab);
  if (p->s_run) (void)hipStreamDestroy(p->s_run);
  if (p->s_h2d) (void)hipStreamDestroy(p->s_h2d);
  (void)hipGetLastError();
  delete p;
}

static int pipeline_build(spx_pipeline* p) {
  const int n = p->n;
  PCHK(hipGetDevice(&p->device));
  // the pipeline's own output layout: capacity per stream as spx_plan_out_capacity_for, regions at 64-byte boundaries
  int64_t oo = 0;
  size_t in_values = 0;
  std::vector<int64_t> tab((size_t)2 * n);
  std::vector<int> chans((size_t)n);
  p->static_offsets.resize((size_t)n + 1);
  for (int i = 0; i < n; i++) {
    spx_stream_job& j = p->jobs[i];
    if (j.channels < 1 || j.n_in < 0 || j.in_off < 0) return pfail(-1, "spx_pipeline: bad job (channels < 1 or a negative count / offset)");
    spx_plan_t pl = p->plans[p->mixed ? p->plan_index[i] : 0];
    j.out_cap = spx_plan_out_capacity_for(pl, j.n_in, j.speed, j.nonlinear);
    j.out_off = oo;
    p->static_offsets[i] = oo;
    tab[i] = oo; tab[(size_t)n + i] = j.out_cap; chans[i] = j.channels;
    oo += (j.out_cap * j.channels + (PIPE_ALIGN - 1)) & ~(int64_t)(PIPE_ALIGN - 1);
    const size_t end = (size_t)j.in_off + (size_t)j.n_in * j.channels;
    if (end > in_values) in_values = end;
  }
  p->static_offsets[n] = oo;
  p->in_values = in_values;
  p->out_values = (size_t)oo;
  p->ws_bytes = p->mixed ? spx_batch_workspace_bytes_mixed(p->plans.data(), (int)p->plans.size(), p->jobs.data(), p->plan_index.data(), n)
                         : spx_batch_workspace_bytes(p->plans[0], p->jobs.data(), n);
  if (!p->ws_bytes) return -1;
#ifdef SPX_TUNING
  // A/B: where the run stream lands among the hardware queues (dummy streams created in front of it), its priority, the null stream
  if (const char* e = getenv("SPX_PIPE_DUMMY_STREAMS")) for (int i = 0; i < atoi(e); i++) { hipStream_t d; (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking); }
  if (getenv("SPX_PIPE_NULL_STREAM")) p->s_run = nullptr;
  else if (const char* e = getenv("SPX_PIPE_PRIO")) { PCHK(hipStreamCreateWithPriority(&p->s_run, hipStreamNonBlocking, atoi(e))); }
  else
#endif
  PCHK(hipStreamCreateWithFlags(&p->s_run, hipStreamNonBlocking));
  PCHK(hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
  const size_t tab_bytes = (size_t)n * (2 * sizeof(int64_t) + sizeof(int));
  PCHK(hipMalloc(reinterpret_cast<void**>(&p->d_tab), tab_bytes));
  PCHK(hipMemcpy(p->d_tab, tab.data(), (size_t)n * 2 * sizeof(int64_t), hipMemcpyHostToDevice));
  PCHK(hipMemcpy(reinterpret_cast<unsigned char*>(p->d_tab) + (size_t)n * 2 * sizeof(int64_t), chans.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  p->slots.resize((size_t)p->depth);
  const bool host_out = !(p->flags & SPX_PIPELINE_DEVICE_OUT);
  for (auto& S : p->slots) {
    // (d_in: allocated by the first submit of host memory -- a caller whose input is device-resident never needs it)
    PCHK(hipMalloc(reinterpret_cast<void**>(&S.d_out), (p->out_values + 64) * sizeof(int16_t)));
    PCHK(hipMemset(S.d_out, 0, (p->out_values + 64) * sizeof(int16_t)));
    PCHK(hipMalloc(reinterpret_cast<void**>(&S.d_nout), (size_t)n * sizeof(int64_t)));
    PCHK(hipMemset(S.d_nout, 0, (size_t)n * sizeof(int64_t)));
    PCHK(hipMalloc(reinterpret_cast<void**>(&S.d_offsets), ((size_t)n + 1) * sizeof(int64_t)));
    PCHK(hipMalloc(&S.ws, p->ws_bytes));
    PCHK(hipMemset(S.ws, 0, p->ws_bytes));
    if (host_out) {
      PCHK(hipHostMalloc(reinterpret_cast<void**>(&S.h_out), (p->out_values + 64) * sizeof(int16_t), hipHostMallocDefault));
      PCHK(hipHostMalloc(reinterpret_cast<void**>(&S.h_meta), ((size_t)2 * n + 1) * sizeof(int64_t), hipHostMallocDefault));
    }
    PCHK(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming));
  }
  PCHK(hipDeviceSynchronize());   // the memsets above ran on the null stream; the pipeline's streams do not wait for it
#ifdef SPX_TUNING
  if (const char* e = getenv("SPX_PIPE_PACK_WGS")) p->pack_wgs = atoi(e) > 0 ? atoi(e) : p->pack_wgs;
#endif
  return 0;
}

extern "C" {

spx_pipeline_t spx_pipeline_create_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index,
                                         int n_streams, int depth, unsigned flags) {
  if (!plans || n_plans < 1 || n_plans > 8 || !jobs || n_streams < 1 || depth < 0 || depth == 1 || depth > SPX_PIPE_MAX_DEPTH) {
    pfail(-1, "spx_pipeline_create: bad arguments (depth 0 or 2 .. 8)");
    return nullptr;
  }
  spx_pipeline* p = new spx_pipeline();
  p->plans.assign(plans, plans + n_plans);
  p->mixed = plan_index != nullptr || n_plans > 1;
  if (p->mixed) {
    p->plan_index.resize((size_t)n_streams, 0);
    for (int i = 0; i < n_streams; i++) {
      p->plan_index[i] = plan_index ? plan_index[i] : 0;
      if (p->plan_index[i] < 0 || p->plan_index[i] >= n_plans) { delete p; pfail(-1, "spx_pipeline_create: plan_index out of range"); return nullptr; }
    }
  }
  p->n = n_streams;
  p->depth = depth ? depth : 4;
  p->flags = flags;
  p->jobs.assign(jobs, jobs + n_streams);
  if (pipeline_build(p)) { pipeline_free(p); return nullptr; }
  return p;
}
spx_pipeline_t spx_pipeline_create(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, int depth, unsigned flags) {
  if (!plan) { pfail(-1, "spx_pipeline_create: no plan"); return nullptr; }
  return spx_pipeline_create_mixed(&plan, 1, jobs, nullptr, n_streams, depth, flags);
}
void spx_pipeline_destroy(spx_pipeline_t p) { pipeline_free(p); }
int spx_pipeline_depth(spx_pipeline_t p) { return p ? p->depth : 0; }
size_t spx_pipeline_input_values(spx_pipeline_t p) { return p ? p->in_values : 0; }

int16_t* spx_pipeline_host_input(spx_pipeline_t p) {
  if (!p) return nullptr;
  SpxPipeSlot& S = p->slots[(size_t)(p->next_ticket % p->depth)];
  if (!S.h_in) {
    if (hipHostMalloc(reinterpret_cast<void**>(&S.h_in), (p->in_values + 64) * sizeof(int16_t), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      pfail(-2, "spx_pipeline_host_input: pinned allocation failed");
      return nullptr;
    }
  }
  if (S.in_recorded && hipEventSynchronize(S.ev_in) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // the copy that last read it
  return S.h_in;
}

int64_t spx_pipeline_submit(spx_pipeline_t p, const int16_t* in, int in_is_device) {
  if (!p || !in) return pfail(-1, "spx_pipeline_submit: bad arguments");
  int cur = 0;
  if (hipGetDevice(&cur) == hipSuccess && cur != p->device) return pfail(-1, "spx_pipeline_submit: the pipeline's device is not the current one");
  const int64_t ticket = p->next_ticket;
  SpxPipeSlot& S = p->slots[(size_t)(ticket % p->depth)];
  // at most `depth` batches in flight: the batch that last used this buffer set has finished (its output, if nobody asked for
  // it, is dropped here)
  if (S.ticket >= 0) PCHK(hipEventSynchronize(S.ev_done));
  const int16_t* dev_in = in;
  void* in_ready = nullptr;
  if (!in_is_device) {
    // (the kernels that last read d_in are behind ev_done, waited for above: the copy may start at once)
    // ONE copy stream: with two taking turns -- so that the next 82 MB copy starts while the previous one still runs and the
    // 50 - 80 us between two chained copies go -- the leg read 1.73 - 1.75 ms per batch against 1.67 (profiles/r05/r5o_copy_streams.txt).
    if (!S.d_in) {
      // + 64 values behind the input: the walk kernels' aligned window refill may read a few frames past a stream's end; zeroed
      // once, never written again
      PCHK(hipMalloc(reinterpret_cast<void**>(&S.d_in), (p->in_values + 64) * sizeof(int16_t)));
      PCHK(hipMemsetAsync(S.d_in, 0, (p->in_values + 64) * sizeof(int16_t), p->s_h2d));
    }
    PCHK(hipMemcpyAsync(S.d_in, in, p->in_values * sizeof(int16_t), hipMemcpyHostToDevice, p->s_h2d));
    PCHK(hipEventRecord(S.ev_in, p->s_h2d));
    S.in_recorded = true;
    dev_in = S.d_in;
    in_ready = S.ev_in;
  }
  int rc;
  const bool host_out = !(p->flags & SPX_PIPELINE_DEVICE_OUT);
  // With the outputs left on the device nothing of a batch has to run behind its walk kernel: the call is DETACHED from the run
  // stream (spx_engine.hip SpxCallOpts) -- the batch's event is recorded on the walk stream itself and the run stream stays empty.
  bool event_recorded = false;
  if (p->mixed) {
    // (round 6: detached like a one-plan batch -- the groups' walk kernels on the library's walk streams, two calls' worth in flight)
    rc = spx_internal_run_mixed(p->plans.data(), (int)p->plans.size(), p->jobs.data(), p->plan_index.data(), p->n, dev_in, S.d_out, S.d_nout,
                                S.ws, p->ws_bytes, p->s_run, true, in_ready, host_out ? nullptr : S.ev_done, !host_out);
    event_recorded = !host_out;
  } else {
    rc = spx_internal_run(p->plans[0], p->jobs.data(), p->n, dev_in, S.d_out, S.d_nout, S.ws, p->ws_bytes, nullptr, p->s_run, true, true, in_ready,
                          host_out ? nullptr : S.ev_done, !host_out);
    event_recorded = !host_out;
  }
  if (rc) {
    // part of the batch may have been enqueued on this buffer set: nothing of it is handed out, and nothing is left in flight
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    S.ticket = -1;
    return rc;
  }
  S.in_host = !in_is_device;
  if (host_out) {
    const int n = p->n;
    const int64_t* d_off = p->d_tab;
    const int64_t* d_cap = p->d_tab + n;
    const int* d_ch = reinterpret_cast<const int*>(p->d_tab + 2 * (size_t)n);
    hipLaunchKernelGGL(spx_pipe_offsets_kernel, dim3(1), dim3(256), 0, p->s_run, S.d_nout, d_ch, d_cap, n, S.d_offsets, S.h_meta, S.h_meta + n + 1);
    const int wgs = n < p->pack_wgs ? n : p->pack_wgs;
#ifdef SPX_TUNING
    static const bool no_gather = getenv("SPX_PIPE_NO_GATHER") != nullptr;   // DIAGNOSTIC: what the gather kernel costs the others (no output!)
    if (!no_gather)
#endif
    hipLaunchKernelGGL(spx_pipe_copy_kernel, dim3(wgs), dim3(256), 0, p->s_run, S.d_out, d_off, S.d_offsets, n, S.h_out);
  }
  if (!event_recorded) PCHK(hipEventRecord(S.ev_done, p->s_run));
  PCHK(hipGetLastError());
  S.ticket = ticket;
  p->next_ticket++;
  return ticket;
}

static SpxPipeSlot* slot_of(spx_pipeline_t p, int64_t ticket) {
  if (!p || ticket < 0 || ticket >= p->next_ticket) return nullptr;
  SpxPipeSlot& S = p->slots[(size_t)(ticket % p->depth)];
  return S.ticket == ticket ? &S : nullptr;
}
int spx_pipeline_input_consumed(spx_pipeline_t p, int64_t ticket) {
  SpxPipeSlot* S = slot_of(p, ticket);
  if (!S) return pfail(-1, "spx_pipeline_input_consumed: unknown ticket, or its buffers have been handed to a later batch");
  // host input: the copy in has read it; device input: the kernels read it until the batch is done (the walk kernel copies from it)
  if (S->in_host) PCHK(hipEventSynchronize(S->ev_in));
  else PCHK(hipEventSynchronize(S->ev_done));
  return 0;
}
int spx_pipeline_wait(spx_pipeline_t p, int64_t ticket, const int16_t** out, const int64_t** offsets, const int64_t** counts) {
  SpxPipeSlot* S = slot_of(p, ticket);
  if (!S) return pfail(-1, "spx_pipeline_wait: unknown ticket, or its buffers have been handed to a later batch");
  PCHK(hipEventSynchronize(S->ev_done));
  const bool host_out = !(p->flags & SPX_PIPELINE_DEVICE_OUT);
  if (out) *out = host_out ? S->h_out : S->d_out;
  if (offsets) *offsets = host_out ? S->h_meta : p->static_offsets.data();
  if (counts) *counts = host_out ? S->h_meta + p->n + 1 : S->d_nout;
  return 0;
}

void* spx_host_alloc(size_t bytes) {
  void* q = nullptr;
  if (hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pfail(-2, "spx_host_alloc failed"); return nullptr; }
  return q;
}
void spx_host_free(void* q) { if (q) (void)hipHostFree(q); }

}  // extern "C"
