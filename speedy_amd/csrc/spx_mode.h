// The launch mode of a batch call as a PURE function (round 5): no HIP call, no global, no environment variable in here.
// run_impl / mixed_impl (spx_engine.hip) collect the inputs -- the batch's shape, the resources of the kernels that would serve
// it (LDS bytes, waves, allocated registers: hipFuncGetAttributes), the process-wide settings, the state of the plan's ring of
// earlier calls and of its mode trial -- call spx_choose_mode, and then only EXECUTE the decided mode.  The same header compiles
// into a host-only library (speedy_amd/csrc/spx_mode_table.cpp, plain g++) that tests/test_mode_table.py drives on the CPU with
// the resource numbers of profiles/kernel_resources.json, so that "one more register in the tension kernel silently switches a
// mode off" (round 3) is a failing CPU test, not a slower bench line.
//
// The rules themselves are DESIGN.md 2's: (1) deadlock freedom of the polling consumers, (2) the analysis kernel keeps two
// workgroups / two waves beside a stream's own workgroups, (3) placement of the walk workgroups one per CU.
#ifndef SPX_MODE_H_
#define SPX_MODE_H_
#include <stddef.h>

#define SPX_MODE_MAX_CHUNKS 16

struct SpxModeWalk {     // one form of the walk kernel for the batch (spx_walk_config + the kernel's allocated registers)
  size_t lds;            // bytes per workgroup (= per stream)
  int waves;             // waves per workgroup
  int vgprs;             // allocated VGPRs per lane (rounded up to the granule of 8)
  bool fast_kernel;      // spx_walk_fast_kernel
  int nwc;               // its output waves
};
struct SpxModeResources {
  int cu_count;
  size_t lds_per_cu;
  SpxModeWalk walk;        // the form spx_walk_config picks by itself
  SpxModeWalk walk_lean;   // ... with lean = true (no output waves); meaningful when lean_valid
  bool lean_valid;         // the lean form exists for this batch (a fast kernel without output waves)
  size_t tension_lds;
  int tension_vgprs;
  int tile_default, tile_big, tile_small;   // the plan's tile, the 16-frame tile, the 8-frame tile
  size_t an_lds_default, an_lds_small;      // the analysis kernel's LDS with the plan's tile / the small tile
  int an_vgprs_default, an_vgprs_small;     // ... and its allocated registers (the instantiations differ)
};
struct SpxModeShape {
  int n, max_channels;
  bool do_a, do_w;          // analysis / walk half of the call (spx_batch_run: both)
  bool has_frames;          // the batch has at least one analysis frame
  bool forced;              // one group of a mixed-rate call: the mode was decided for all groups together
  bool force_concurrent, force_ahead;
  int force_total_streams;  // streams of all groups of that call
  bool ahead_req;           // spx_batch_run_ahead / _overlapped / the pipeline object
  bool overlap_req;         // spx_batch_run_overlapped: the walk kernels of consecutive calls may overlap
};
struct SpxModeEnv {         // process-wide settings (spx_set_*) and the developers' A/B switches (all false in the shipped library)
  bool concurrent_enabled;  // spx_set_concurrent
  bool chunks_set;          // spx_set_pipeline_chunks was called
  int chunks;
  bool serial, no_lean, small_tile, ahead_any, full_walk, walk1, no_excl;
  int trial_force;          // -1: trials decide; 0 / 1: no trial
};
struct SpxModeTrial {       // plan-owned state of the timed trial of "doubtful" shapes (one analysis wave per SIMD beside the consumers)
  long long key;
  int calls;
  int choice;               // -1 undecided, 0 in sequence, 1 concurrent
};
struct SpxModeRuntime {
  bool two_workspaces;      // the plan's call two back used this call's workspace (two buffer sets taking turns)
  bool guard_busy;          // another concurrent-mode call is in flight on the device on a different stream
  bool trial_times_ready;   // both timed trial calls have finished: ms_seq / ms_con are valid
  float ms_seq, ms_con;
  long long trial_key;      // key of this call's shape
  bool (*device_ours)(void*);   // asked only when the answer matters (taking the per-device lock file is a side effect)
  void* device_ctx;
};
struct SpxMode {
  bool lean_walk;           // the arithmetic below was done with the lean walk form
  bool launch_lean;         // ... and the walk kernel is launched in it (only beside other kernels)
  int tile_frames;
  bool co_resident, doubtful;
  int trial_slot;           // 0 / 1: this call is the timed trial of the sequential / concurrent order; -1: none
  bool want_concurrent;
  bool concurrent;          // three kernels side by side, flags between them
  bool ahead, seq_ahead, ahead_forced;   // pipelined with the previous call (seq_ahead: its kernels otherwise in sequence)
  bool walk2;               // ... and its walk kernel on one of the library's two walk streams
  bool chunk_ahead;         // a call of more than two streams per CU (time chunks, throughput-form walk kernels) whose PRODUCERS go to the
                            // side stream at once: the next call's first analysis chunk beside this call's last walk chunk (round 6)
  int nch;                  // time chunks
  bool exclusive_cu;        // the walk workgroups ask for more than half a CU's LDS
  bool asked_device;        // device_ours was consulted
  SpxModeTrial trial_next;  // the trial state after this call
};

static inline bool spx_mode_fits2(const SpxModeResources& R, const SpxModeWalk& w, size_t lds_usable) {
  // a stream's walk and tension workgroups and still TWO analysis workgroups on a CU, two analysis waves on a SIMD (default tile)
  return w.lds + R.tension_lds + 2 * R.an_lds_default <= lds_usable &&
         ((w.waves + 3) / 4) * w.vgprs + R.tension_vgprs + 2 * R.an_vgprs_default <= 512;
}

// One pass of the decision with the walk form given (lean or not).
static inline SpxMode spx_mode_pass(const SpxModeShape& S, const SpxModeResources& R, const SpxModeEnv& E, const SpxModeRuntime& T,
                                    const SpxModeTrial& trial_in, bool lean_walk) {
  SpxMode M;
  M.lean_walk = lean_walk;
  M.asked_device = false;
  M.trial_next = trial_in;
  const SpxModeWalk& W = lean_walk ? R.walk_lean : R.walk;
  const bool both = S.do_a && S.do_w;
  const size_t lds_usable = R.lds_per_cu > 6144 ? R.lds_per_cu - 6144 : R.lds_per_cu;
  const size_t per_stream_lds = W.lds + R.tension_lds;
  const size_t per_stream_waves = (size_t)W.waves + 4;
  auto ours = [&]() { M.asked_device = true; return T.device_ours ? T.device_ours(T.device_ctx) : true; };
  // the tile: the small one when that is what lets two analysis workgroups sit beside a stream's workgroups
  int tile = R.tile_default;
  if (per_stream_lds + 2 * R.an_lds_default > lds_usable && both && R.tile_default == R.tile_big &&
      per_stream_lds + 2 * R.an_lds_small <= lds_usable)
    tile = R.tile_small;
  if (E.small_tile && both && tile == R.tile_big) tile = R.tile_small;
  const size_t an_lds = tile == R.tile_default ? R.an_lds_default : R.an_lds_small;
  const int an_vgprs = tile == R.tile_default ? R.an_vgprs_default : R.an_vgprs_small;
  // rule 1 (deadlock freedom) and rule 2 (worth it)
  bool co_resident = false, doubtful = false;
  if (an_lds < R.lds_per_cu) {
    const size_t lds_closing = R.lds_per_cu - an_lds + 1;
    const size_t closed = ((size_t)S.n * per_stream_lds) / lds_closing + ((size_t)S.n * per_stream_waves) / 29;
    co_resident = closed < (size_t)R.cu_count && S.n <= R.cu_count;
    if (per_stream_lds + 2 * an_lds > lds_usable) co_resident = false;
    if (co_resident) {
      const int walk_regs = ((W.waves + 3) / 4) * W.vgprs;
      if (walk_regs + R.tension_vgprs + 2 * an_vgprs > 512) {
        if (walk_regs + R.tension_vgprs + an_vgprs > 512) co_resident = false;   // not even one analysis wave
        else doubtful = true;                                                    // one: decided by trial
      }
    }
  }
  M.doubtful = doubtful;
  // the trial of doubtful shapes: first call concurrent and untimed (cold), second concurrent and timed, third in sequence and
  // timed, later calls take the faster
  int trial_slot = -1;
  if (co_resident && doubtful && both && E.concurrent_enabled && !E.serial && !S.forced) {
    SpxModeTrial& N = M.trial_next;
    if (N.key != T.trial_key) { N.key = T.trial_key; N.calls = 0; N.choice = E.trial_force; }
    if (N.choice < 0 && N.calls >= 3 && T.trial_times_ready) N.choice = T.ms_con < T.ms_seq ? 1 : 0;
    if (N.choice >= 0) co_resident = N.choice == 1;
    else if (N.calls == 0) { }
    else if (N.calls == 1) trial_slot = 1;
    else if (N.calls == 2) { trial_slot = 0; co_resident = false; }
    else co_resident = false;   // timings not in yet
    N.calls++;
  }
  M.co_resident = co_resident;
  M.trial_slot = trial_slot;
  bool want_concurrent = E.concurrent_enabled && !E.serial && co_resident && both;
  if (want_concurrent && !S.forced && !ours()) want_concurrent = false;   // another process works on this GPU
  if (S.forced) want_concurrent = S.force_concurrent && both;
  M.ahead_forced = S.forced && S.force_ahead && both;
  const bool chunks_free = !E.chunks_set || E.chunks == 1;
  // a batch whose kernels run in sequence can still be pipelined with its predecessor if ONE analysis workgroup fits beside a
  // walk workgroup that asks for a CU of its own and one analysis wave beside its waves (default tile, full walk form)
  bool seq_ahead = false;
  if (S.ahead_req && !want_concurrent && !S.forced && both && S.n <= R.cu_count && E.concurrent_enabled && !E.serial &&
      trial_slot < 0 && chunks_free && !lean_walk) {
    const size_t walk_lds = W.lds > R.lds_per_cu / 2 + 1024 ? W.lds : R.lds_per_cu / 2 + 1024;
    const int walk_regs = ((W.waves + 3) / 4) * W.vgprs;
    seq_ahead = walk_lds + R.tension_lds + R.an_lds_default <= lds_usable &&
                walk_regs + R.tension_vgprs + R.an_vgprs_default <= 512 && ours();
  }
  M.seq_ahead = seq_ahead;
  M.ahead = M.ahead_forced || seq_ahead ||
            (S.ahead_req && (want_concurrent || (E.ahead_any && both && S.n <= R.cu_count && E.concurrent_enabled)) && !doubtful &&
             !S.forced && trial_slot < 0 && chunks_free);
  // a batch without a single analysis frame (linear jobs only: the TSM stage alone) has no producers to run ahead of anything: its
  // walk kernel on the caller's stream, call after call (measured, profiles/r05/r5k_scale_configs.txt: 16 kHz linear 0.5x 2.86 ms
  // plain against 3.55 in the pipelined order)
  if (!S.has_frames && !M.ahead_forced) { M.ahead = false; seq_ahead = false; M.seq_ahead = false; }
  if (M.ahead) want_concurrent = false;
  if (want_concurrent && !S.forced && T.guard_busy) want_concurrent = false;
  M.want_concurrent = want_concurrent;
  if (!want_concurrent && (!M.ahead || seq_ahead)) tile = R.tile_default;   // no concurrency: the default tile
  M.tile_frames = tile;
  int nch = both ? E.chunks : 1;
  if (both && !E.chunks_set && !want_concurrent && S.n > 2 * R.cu_count) nch = 2;   // the throughput regime: two time chunks
  if (M.ahead) nch = 1;
  if (nch < 1) nch = 1;
  if (nch > SPX_MODE_MAX_CHUNKS) nch = SPX_MODE_MAX_CHUNKS;
  M.nch = nch;
  M.concurrent = want_concurrent && nch == 1 && S.has_frames;
  M.walk2 = S.overlap_req && !E.walk1 && M.ahead && !S.forced && !seq_ahead;
  const int total = S.forced ? S.force_total_streams : S.n;
  M.exclusive_cu = !M.concurrent && (!M.ahead || M.ahead_forced || seq_ahead) && !E.no_excl && nch == 1 && total <= R.cu_count;
  M.launch_lean = lean_walk && (M.concurrent || M.ahead);
  // Large calls asked to be pipelined (spx_batch_run_ahead / _overlapped, the pipeline object): kernels in sequence, two time chunks --
  // the analysis chunks on the side stream, the tension and walk kernels on the caller's behind them.  With the producers started at
  // once (behind the ring's events, not behind the caller's stream) the side stream runs analysis chunk after analysis chunk, call
  // after call, beside the walk chunks on the caller's stream: the period of back-to-back calls is the larger of the two kernel sums
  // instead of their overlap-less tail and head added up.
  M.chunk_ahead = S.ahead_req && !S.forced && both && !M.ahead && !want_concurrent && nch >= 2 && S.n > 2 * R.cu_count &&
                  E.concurrent_enabled && !E.serial && trial_slot < 0 && ours();
  return M;
}

// The decision.  The LEAN walk form (no output waves: one walk wave per SIMD instead of two) is taken
//   - by necessity: where the usual form leaves no room for two analysis workgroups / waves beside it and the lean one does
//     (22.05 kHz mono in the concurrent mode), and
//   - by preference: when the call's walk kernel WILL overlap the previous call's (walk2) with three or more workspaces taking
//     turns -- the walk kernels have time to spare there, what the period waits for is the analysis kernel, and beside two lean
//     walk workgroups a SIMD holds two analysis waves.  Decided AFTER `ahead` is known (round 4 decided it before, and a call
//     that then fell out of the pipelined mode ran the lean form where the full one fits).
static inline SpxMode spx_choose_mode(const SpxModeShape& S, const SpxModeResources& R, const SpxModeEnv& E, const SpxModeRuntime& T,
                                      const SpxModeTrial& trial_in) {
  const size_t lds_usable = R.lds_per_cu > 6144 ? R.lds_per_cu - 6144 : R.lds_per_cu;
  const bool lean_possible = S.do_a && S.do_w && S.max_channels == 1 && S.n <= R.cu_count && R.walk.fast_kernel && R.walk.nwc > 0 &&
                             !E.no_lean && !S.forced && R.lean_valid && R.walk_lean.fast_kernel && R.walk_lean.nwc == 0;
  const bool lean_fits = lean_possible && spx_mode_fits2(R, R.walk_lean, lds_usable);
  const bool need_lean = lean_fits && !spx_mode_fits2(R, R.walk, lds_usable);
  SpxMode M = spx_mode_pass(S, R, E, T, trial_in, need_lean);
  if (!need_lean && lean_fits && M.walk2 && S.overlap_req && !E.full_walk && !T.two_workspaces) {
    SpxMode L = spx_mode_pass(S, R, E, T, trial_in, true);
    L.asked_device = L.asked_device || M.asked_device;
    if (L.ahead && L.walk2) return L;
  }
  return M;
}

// ---- a mixed-rate call (spx_batch_run_mixed): one decision for all groups together ----
struct SpxModeGroup {      // one non-empty group (plan) of the call
  int n;
  SpxModeWalk walk;
  bool any_nonlinear;
  size_t an_lds;           // its analysis kernel (default tile)
  int an_vgprs;
};
struct SpxMixedMode {
  bool concurrent;
  bool ahead;
  bool chain_analyses;     // kernels in sequence: the groups' analysis kernels one after the other, cheapest first
  bool asked_device;
};
static inline SpxMixedMode spx_choose_mixed_mode(const SpxModeGroup* G, int groups, int n_total, int cu_count, size_t lds_per_cu,
                                                 size_t tension_lds, int tension_vgprs, const SpxModeEnv& E, int env_mixed_mode,
                                                 bool no_order, bool ahead_req, const SpxModeRuntime& T) {
  SpxMixedMode M = {false, false, false, false};
  size_t cons_lds = 0, cons_waves = 0, max_ps_lds = 0, max_an_lds = 0;
  int max_walk_regs = 0, max_an_regs = 0;
  for (int g = 0; g < groups; g++) {
    const size_t ps = G[g].walk.lds + tension_lds;
    cons_lds += (size_t)G[g].n * ps;
    cons_waves += (size_t)G[g].n * (G[g].walk.waves + 4);
    if (ps > max_ps_lds) max_ps_lds = ps;
    if (G[g].any_nonlinear) {
      if (G[g].an_lds > max_an_lds) max_an_lds = G[g].an_lds;
      if (G[g].an_vgprs > max_an_regs) max_an_regs = G[g].an_vgprs;
    }
    const int wr = ((G[g].walk.waves + 3) / 4) * G[g].walk.vgprs;
    if (wr > max_walk_regs) max_walk_regs = wr;
  }
  auto ours = [&]() { M.asked_device = true; return T.device_ours ? T.device_ours(T.device_ctx) : true; };
  bool concurrent = false;
  if (max_an_lds > 0 && max_an_lds < lds_per_cu && n_total <= cu_count) {
    const size_t closed = cons_lds / (lds_per_cu - max_an_lds + 1) + cons_waves / 29;
    concurrent = closed < (size_t)cu_count;
    const size_t lds_usable = lds_per_cu > 6144 ? lds_per_cu - 6144 : lds_per_cu;
    if (max_ps_lds + 2 * max_an_lds > lds_usable) concurrent = false;
    if (max_walk_regs + tension_vgprs + 2 * max_an_regs > 512) concurrent = false;
  }
  if (!E.concurrent_enabled || E.serial || groups == 0) concurrent = false;
  if (env_mixed_mode >= 0 && max_an_lds > 0) concurrent = env_mixed_mode == 1;
  if (concurrent && !ours()) concurrent = false;
  if (concurrent && T.guard_busy) concurrent = false;   // (before `ahead`: a call that hits the busy guard is still pipelined -- ADVICE r5)
  M.ahead = ahead_req && !concurrent && E.concurrent_enabled && !E.serial && groups > 0 && n_total <= cu_count && ours();
  M.concurrent = concurrent;
  M.chain_analyses = !concurrent && !no_order && groups > 1;
  return M;
}

// ... and whether the walk kernels of consecutive mixed calls overlap (round 6): a pipelined call (`ahead` above) that is DETACHED -- its
// caller owns every buffer and orders their consumers behind the call's event itself: the pipeline object with its outputs left on the
// device -- and hands over no taps.  Its first two groups' walk kernels then go to the library's walk streams, two pairs taking turns,
// in the 4 + 4 form with the 4096-frame window and without the LDS request for a CU of their own: two calls' walk workgroups and an
// analysis workgroup share a CU (configs[4] shard through the pipeline object: 1.87 - 1.97 -> 1.56 ms per step).
static inline bool spx_mixed_walk2(const SpxMixedMode& M, bool detached, bool taps, const SpxModeEnv& E) {
  return M.ahead && !M.concurrent && detached && !taps && !E.walk1;
}

#endif  // SPX_MODE_H_
