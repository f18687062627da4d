// Plans (include/speedy_hip.h spx_plan_*): the constant tables of one (sample rate, hysteresis mode) -- the DFT spec's twiddles
// (spx_twiddle.h: machine-independent), the Hamming window, the hysteresis tapers, Rader's tables for a prime window -- and the
// sizes and capacities derived from the rate.  Reference: speedy.c:206-299 (speedyCreateStream), speedy.h:136-146.
#include "spx_engine.h"
#include "spx_twiddle.h"
#include "spx_twiddle_hashes.h"

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

void spx_internal_set_error(const char* msg) { g_spx_err = msg ? msg : ""; }   // other translation units' errors reach spx_last_error
extern "C" {

const char* spx_last_error(void) { return g_spx_err.c_str(); }
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
  for (auto& g : plan->mix_stage) {
    if (g.done) { (void)hipEventSynchronize(g.done); (void)hipEventDestroy(g.done); }
    if (g.p) (void)hipHostFree(g.p);
  }
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

int64_t spx_plan_frames(spx_plan_t p, int64_t n_in) { return frames_for(p->dev, n_in); }

int64_t spx_plan_out_capacity_for(spx_plan_t p, int64_t n_in, float speed, float nonlinear) {
  return spx_internal_out_bound(p->dev, n_in, speed, nonlinear != 0.0f);
}
int64_t spx_plan_out_capacity(spx_plan_t p, int64_t n_in, float speed) {
  return spx_internal_out_bound(p->dev, n_in, speed, true);
}

}  // extern "C"
bool spx_internal_analysis_fits(const SpxPlanDev& d) { return spx_analysis_lds_bytes(d) <= 160 * 1024; }

spx_plan* shared_plan_full(int sample_rate, int match_matlab) {
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
