// Host-only build of the launch-mode decision (spx_mode.h) for tests/test_mode_table.py: plain g++, no HIP.  The query and the
// answer are flat arrays of 64-bit integers so that the Python side mirrors them by NAME (the two name lists below are exported).
#include <string.h>

#include "spx_mode.h"

#define SPX_Q_FIELDS(X)                                                                                                         \
  X(n) X(max_channels) X(do_a) X(do_w) X(has_frames) X(forced) X(force_concurrent) X(force_ahead) X(force_total_streams)        \
  X(ahead_req) X(overlap_req) X(cu_count) X(lds_per_cu) X(walk_lds) X(walk_waves) X(walk_vgprs) X(walk_fast) X(walk_nwc)        \
  X(lean_lds) X(lean_waves) X(lean_vgprs) X(lean_fast) X(lean_nwc) X(lean_valid) X(tension_lds) X(tension_vgprs)                \
  X(tile_default) X(tile_big) X(tile_small) X(an_lds_default) X(an_lds_small) X(an_vgprs_default) X(an_vgprs_small)             \
  X(concurrent_enabled) X(chunks_set) X(chunks) X(serial) X(no_lean) X(small_tile) X(ahead_any) X(full_walk) X(walk1)           \
  X(no_excl) X(trial_force) X(trial_state_key) X(trial_calls) X(trial_choice) X(two_workspaces) X(guard_busy)                   \
  X(trial_times_ready) X(us_seq) X(us_con) X(trial_key) X(device_ours)
#define SPX_A_FIELDS(X)                                                                                                         \
  X(lean_walk) X(launch_lean) X(tile_frames) X(co_resident) X(doubtful) X(trial_slot) X(want_concurrent) X(concurrent)          \
  X(ahead) X(seq_ahead) X(ahead_forced) X(walk2) X(nch) X(exclusive_cu) X(asked_device) X(next_key) X(next_calls) X(next_choice)

enum {
#define X(f) Q_##f,
  SPX_Q_FIELDS(X)
#undef X
  Q_COUNT
};
enum {
#define X(f) A_##f,
  SPX_A_FIELDS(X)
#undef X
  A_COUNT
};

static bool ours_cb(void* ctx) { return *static_cast<long long*>(ctx) != 0; }

extern "C" {
const char* spx_mode_table_query_fields(void) {
  return
#define X(f) #f " "
      SPX_Q_FIELDS(X)
#undef X
      ;
}
const char* spx_mode_table_answer_fields(void) {
  return
#define X(f) #f " "
      SPX_A_FIELDS(X)
#undef X
      ;
}
int spx_mode_table_eval(const long long* q, int nq, long long* a, int na) {
  if (nq != Q_COUNT || na != A_COUNT) return -1;
  SpxModeShape S;
  SpxModeResources R;
  SpxModeEnv E;
  SpxModeRuntime T;
  memset(&S, 0, sizeof(S)); memset(&R, 0, sizeof(R)); memset(&E, 0, sizeof(E)); memset(&T, 0, sizeof(T));
  S.n = (int)q[Q_n]; S.max_channels = (int)q[Q_max_channels]; S.do_a = q[Q_do_a]; S.do_w = q[Q_do_w]; S.has_frames = q[Q_has_frames];
  S.forced = q[Q_forced]; S.force_concurrent = q[Q_force_concurrent]; S.force_ahead = q[Q_force_ahead];
  S.force_total_streams = (int)q[Q_force_total_streams]; S.ahead_req = q[Q_ahead_req]; S.overlap_req = q[Q_overlap_req];
  R.cu_count = (int)q[Q_cu_count]; R.lds_per_cu = (size_t)q[Q_lds_per_cu];
  R.walk = {(size_t)q[Q_walk_lds], (int)q[Q_walk_waves], (int)q[Q_walk_vgprs], q[Q_walk_fast] != 0, (int)q[Q_walk_nwc]};
  R.walk_lean = {(size_t)q[Q_lean_lds], (int)q[Q_lean_waves], (int)q[Q_lean_vgprs], q[Q_lean_fast] != 0, (int)q[Q_lean_nwc]};
  R.lean_valid = q[Q_lean_valid];
  R.tension_lds = (size_t)q[Q_tension_lds]; R.tension_vgprs = (int)q[Q_tension_vgprs];
  R.tile_default = (int)q[Q_tile_default]; R.tile_big = (int)q[Q_tile_big]; R.tile_small = (int)q[Q_tile_small];
  R.an_lds_default = (size_t)q[Q_an_lds_default]; R.an_lds_small = (size_t)q[Q_an_lds_small];
  R.an_vgprs_default = (int)q[Q_an_vgprs_default]; R.an_vgprs_small = (int)q[Q_an_vgprs_small];
  E.concurrent_enabled = q[Q_concurrent_enabled]; E.chunks_set = q[Q_chunks_set]; E.chunks = (int)q[Q_chunks]; E.serial = q[Q_serial];
  E.no_lean = q[Q_no_lean]; E.small_tile = q[Q_small_tile]; E.ahead_any = q[Q_ahead_any]; E.full_walk = q[Q_full_walk];
  E.walk1 = q[Q_walk1]; E.no_excl = q[Q_no_excl]; E.trial_force = (int)q[Q_trial_force];
  const SpxModeTrial trial = {q[Q_trial_state_key], (int)q[Q_trial_calls], (int)q[Q_trial_choice]};
  T.two_workspaces = q[Q_two_workspaces]; T.guard_busy = q[Q_guard_busy]; T.trial_times_ready = q[Q_trial_times_ready];
  T.ms_seq = (float)q[Q_us_seq] / 1000.0f; T.ms_con = (float)q[Q_us_con] / 1000.0f; T.trial_key = q[Q_trial_key];
  long long ours = q[Q_device_ours];
  T.device_ours = ours_cb; T.device_ctx = &ours;
  const SpxMode M = spx_choose_mode(S, R, E, T, trial);
  a[A_lean_walk] = M.lean_walk; a[A_launch_lean] = M.launch_lean; a[A_tile_frames] = M.tile_frames; a[A_co_resident] = M.co_resident;
  a[A_doubtful] = M.doubtful; a[A_trial_slot] = M.trial_slot; a[A_want_concurrent] = M.want_concurrent; a[A_concurrent] = M.concurrent;
  a[A_ahead] = M.ahead; a[A_seq_ahead] = M.seq_ahead; a[A_ahead_forced] = M.ahead_forced; a[A_walk2] = M.walk2; a[A_nch] = M.nch;
  a[A_exclusive_cu] = M.exclusive_cu; a[A_asked_device] = M.asked_device; a[A_next_key] = M.trial_next.key;
  a[A_next_calls] = M.trial_next.calls; a[A_next_choice] = M.trial_next.choice;
  return 0;
}
// a mixed-rate call: groups[g] = {n, walk_lds, walk_waves, walk_vgprs, any_nonlinear, an_lds, an_vgprs}; out = {concurrent, ahead, chain_analyses, asked_device}
int spx_mode_table_eval_mixed(const long long* groups, int n_groups, int n_total, int cu_count, long long lds_per_cu, long long tension_lds,
                              int tension_vgprs, int concurrent_enabled, int serial, int env_mixed_mode, int no_order, int ahead_req,
                              int guard_busy, int device_ours, long long* out4) {
  SpxModeGroup G[8];
  if (n_groups < 0 || n_groups > 8) return -1;
  for (int g = 0; g < n_groups; g++) {
    const long long* p = groups + 7 * g;
    G[g].n = (int)p[0];
    G[g].walk = {(size_t)p[1], (int)p[2], (int)p[3], true, 4};
    G[g].any_nonlinear = p[4] != 0;
    G[g].an_lds = (size_t)p[5];
    G[g].an_vgprs = (int)p[6];
  }
  SpxModeEnv E;
  memset(&E, 0, sizeof(E));
  E.concurrent_enabled = concurrent_enabled; E.serial = serial; E.chunks = 1; E.trial_force = -1;
  SpxModeRuntime T;
  memset(&T, 0, sizeof(T));
  long long ours = device_ours;
  T.guard_busy = guard_busy; T.device_ours = ours_cb; T.device_ctx = &ours;
  const SpxMixedMode M = spx_choose_mixed_mode(G, n_groups, n_total, cu_count, (size_t)lds_per_cu, (size_t)tension_lds, tension_vgprs, E,
                                               env_mixed_mode, no_order != 0, ahead_req != 0, T);
  out4[0] = M.concurrent; out4[1] = M.ahead; out4[2] = M.chain_analyses; out4[3] = M.asked_device;
  return 0;
}
// spx_mixed_walk2 for a decided mixed mode {concurrent, ahead}
int spx_mode_table_mixed_walk2(int concurrent, int ahead, int detached, int taps, int walk1) {
  SpxMixedMode M = {concurrent != 0, ahead != 0, false, false};
  SpxModeEnv E;
  memset(&E, 0, sizeof(E));
  E.walk1 = walk1 != 0;
  return spx_mixed_walk2(M, detached != 0, taps != 0, E) ? 1 : 0;
}
}
