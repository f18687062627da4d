// One call for a batch whose streams differ in sample rate (BASELINE configs[4]): a group per plan, all groups launched together
// (spx_batch_run_mixed*, spx_batch_workspace_bytes_mixed, spx_batch_read_steps_mixed of include/speedy_hip.h).
#include "spx_engine.h"

extern "C" {
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
                      void* hs, bool ahead_req, void* in_ready = nullptr, void* done_event = nullptr, bool detached = false);
}  // extern "C"
// (spx_pipeline.hip: a pipelined mixed call whose producers wait for an "input is there" event; detached: the caller owns every buffer
// of the call and orders their consumers itself behind done_event -- the call's walk kernels then run on the library's walk streams,
// beside the previous call's, and nothing of it is enqueued on hs)
int spx_internal_run_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index, int n, const int16_t* in,
                           int16_t* out, int64_t* n_out, void* ws, size_t ws_bytes, void* hs, bool ahead, void* in_ready,
                           void* done_event, bool detached) {
  return mixed_impl(plans, n_plans, jobs, plan_index, n, in, out, n_out, ws, ws_bytes, nullptr, hs, ahead, in_ready, done_event, detached);
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
                      void* hs, bool ahead_req, void* in_ready, void* done_event, bool detached_req) {
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
  SpxForce force = {0, false, n, nullptr, nullptr, nullptr, false};
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
  // ... and (round 6, a DETACHED call: the pipeline object with its outputs left on the device) the walk kernels of consecutive calls
  // overlapping as well, the way spx_batch_run_overlapped runs a one-plan batch: the first two groups' walk kernels on the library's
  // walk streams -- two pairs taking turns, so at most two calls' walk kernels are in flight --, their workgroups without the LDS
  // request that gives each a CU of its own (two calls' workgroups share the CUs), the scatter kernel and the call's events behind
  // them on the first group's walk stream, nothing on the caller's.  configs[4] shard: 1.87 - 1.97 -> see profiles/r06.
  static const bool no_mixed_walk2 = spx_tuning_env("SPX_MIXED_NO_WALK2") != nullptr;   // A/B
  const bool walk2 = spx_mixed_walk2(MM, detached_req, taps != nullptr, E) && !no_mixed_walk2;
  static_assert(SPX_MAX_WALK_STREAMS >= 4, "two pairs of walk streams taking turns");
  hipStream_t wst[SPX_MAX_WALK_STREAMS] = {nullptr};
  if (walk2 && dev_walk_streams(lead->device, wst, 4)) return fail(-1, "spx_batch_run_mixed: no walk streams");
  force.no_exclusive = walk2;
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
    SpxStage& G = lead->mix_stage[lead->mix_next];
    lead->mix_next ^= 1;
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
        if (sn.first && sn.second > 0) spx_launch_gate(sn.first, sn.second, 8000u, dev_s1);
  }
  hipEvent_t prev_an = nullptr;
  int launch_idx = 0;
  hipStream_t end_stream = nullptr;   // walk2: the first group's walk stream, where the call ends
  for (int g : ord) {
    spx_plan* p = plans[g];
    hipStream_t gs = nullptr;
    if (walk2 && launch_idx < 2) gs = wst[2 * launch_idx + (int)(lead->ahead_calls & 1)];
    else if (!concurrent && launch_idx == 0) gs = st;
    else if (!concurrent && launch_idx == 1) gs = dev_s2;
    else {
      if (!p->mix) HIPCHK(hipStreamCreateWithFlags(&p->mix, hipStreamNonBlocking));
      gs = p->mix;
    }
    launch_idx++;
    if (!p->ev_join) HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    if (!p->ev_an) HIPCHK(hipEventCreateWithFlags(&p->ev_an, hipEventDisableTiming));
    const bool on_walk_stream = walk2 && launch_idx <= 2;   // (launch_idx is already this group's + 1)
    if (gs != st && !on_walk_stream) HIPCHK(hipStreamWaitEvent(gs, lead->ev_fork, 0));
    if (!end_stream) end_stream = gs;
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
    const std::string err = rc ? g_spx_err : std::string();
    hipStream_t join_to = walk2 ? end_stream : st;
    if (gs != join_to && (hipEventRecord(p->ev_join, gs) != hipSuccess || hipStreamWaitEvent(join_to, p->ev_join, 0) != hipSuccess)) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(gs);   // no event: make sure by waiting here
      if (!rc) return fail(-2, "spx_batch_run_mixed: joining a group's stream failed");
    }
    if (rc) return fail(rc, err);
  }
  hipStream_t fin = (walk2 && end_stream) ? end_stream : st;
  hipLaunchKernelGGL(spx_scatter_nout_kernel, dim3((n + 255) / 256), dim3(256), 0, fin, d_nout, d_idx, n, n_out);
  if (done_event) HIPCHK(hipEventRecord(static_cast<hipEvent_t>(done_event), fin));
  {
    // every mixed call leaves its end in the lead plan's ring (a pipelined call orders its producers behind the two calls before it)
    std::lock_guard<std::mutex> ring_lock(lead->mu);
    if (ring_record(lead, fin, ws, st, out, n_out, walk2)) return -2;
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
    rc = spx_read_steps(plans[g]->dev, gj[g].data(), (int)gj[g].size(), static_cast<const unsigned char*>(ws) + M.ws_off[g], sg.data(),
                    static_cast<hipStream_t>(hs));
    if (rc) return rc;
    for (size_t k = 0; k < sg.size(); k++) steps[gi[g][k]] = sg[k];
  }
  return 0;
}

}  // extern "C"
