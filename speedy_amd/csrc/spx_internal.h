// Internal definitions shared by the HIP kernels and the C-ABI host code.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

// A/B switches of the developers' tools (tools/ab_lib.sh, tools/scale_sweep.sh ...) exist only in builds with -DSPX_TUNING
// (`make -C speedy_amd/csrc tuning` -> speedy_amd/lib/ab/libspeedy_hip_tuning.so).  The shipped library reads the documented
// environment variables only (INTEGRATION.md "Environment"): SPX_NO_POOL, SPX_POOL_*, SPX_SHARED_GPU, SPX_LOCK_DIR,
// SPX_DEBUG_MODE, SPX_DEBUG_TRIAL.
#ifdef SPX_TUNING
static inline const char* spx_tuning_env(const char* name) { return getenv(name); }
#else
static inline const char* spx_tuning_env(const char*) { return nullptr; }
#endif

// Speeds the speed-up walk kernel (spx_walk_fast.hip) serves: its division for the step lengths is the IEEE sequence without
// the scaling half, exact for speed - 1 in [2^-60, 2^80) (fast_div).  Anything beyond (a speed of 1e18 turns every step into a
// failed one anyway) runs on the general kernel.
#define SPX_FAST_MAX_SPEED 1.0e18f

#define SPX_MAX_STAGES 16
#define SPX_WAVE 64
#define SPX_BLOCK 256
#define SPX_FEATURE_COUNT 15  // speedy.h:115

// Constant tables and sizes of one plan; passed BY VALUE to every kernel.
struct SpxPlanDev {
  int rate;
  int B;  // frame step   = rate/100              (speedy.c:335-338)
  int W;  // window size  = (int)(1.5*rate/100)   (speedy.c:213)
  int N;  // fft size     = 2*W                   (speedy.c:214)
  int F;  // kTemporalHysteresisFuture            (speedy.h:136-146)
  int Pp; // kTemporalHysteresisPast
  int nstages;
  int radix[SPX_MAX_STAGES];
  int minPeriod, maxPeriod, maxRequired, skip;  // libsonic limits
  int tile_frames;  // frames per analysis tile (spx_analysis_tile_frames)
  int dft_waves;    // waves of an analysis workgroup that transform frames (4; fewer for the windows of the highest sample rates)
  float alpha;            // (float)exp(-1.0/100)             (speedy.c:67,287)
  float one_minus_alpha;  // (1 - alpha) evaluated in float   (speedy.c:74)
  const double* tw;    // [W]  (cos, -sin)(2 pi t / W)
  const double* tw2;   // [W]  (cos, -sin)(2 pi k / 2W)
  const float* window; // [W]  Hamming                         (speedy.c:256-258)
  const float* taperF; // [F+1]  (F-i)/(float)F               (speedy.c:597)
  const float* taperP; // [P+1]  (P-i)/(float)P               (speedy.c:604)
  // Rader's algorithm for a large prime W whose W-1 is smooth (44.1 kHz: W = 661, 660 = 4*3*5*11; DESIGN.md "DFT
  // spec"): the W-point transform as a cyclic convolution of length M = W-1 done with an M-point plan.
  int rader;                     // 0: plain mixed-radix stages
  int nstagesM;
  int radixM[SPX_MAX_STAGES];
  const double* twM;   // [M]  (cos, -sin)(2 pi t / M)
  const double* bfft;  // [M]  M-point transform of b[q] = tw[iperm[q]]
  const int* perm;     // [M]  perm[p]  = g^p  mod W, g the smallest primitive root
  const int* iperm;    // [M]  iperm[q] = g^-q mod W
  const int* qlog;     // [W]  qlog[k] = the q with g^-q = k (k = 1 .. M; entry 0 unused)
};

// Per-stream job in device memory.  A job covers "everything new since the last call": batch jobs start
// from scratch (SPX_F_INIT) and end the stream (SPX_F_FLUSH) in one go; the streaming API issues a job
// per sonicWrite*/sonicFlush call with the state record carried in device memory between calls.
#define SPX_F_INIT 1   // ignore the stored state, start a fresh stream
#define SPX_F_FLUSH 2  // after the new input: sonicFlushStream (soniclib.c:529-552)
#define SPX_F_TENSION_RANGE 4  // unit-level API: compute exactly the tension frames [tension_skip, tension_to)
#define SPX_F_NO_SPEED 8       // unit-level API: no speed / duration pass (speedyComputeSpeedFromTension is its own call)
#define SPX_F_NO_TRUNC 16      // a rate stage follows (sonicSetRate != 1): the flush does not truncate the TSM output -- the
                               // dependency truncates the FINAL output there -- and leaves `flush_remaining` in the state record
#define SPX_F_SPEED_SET 32     // sonicSetSpeed was called since the last job: it reaches the TSM stage at once, in nonlinear
                               // mode too (soniclib.c:177-182) -- a flush right behind it hands out the ring buffers at that speed
#define SPX_F_KEEP_SPEED 64    // a linear job on a stream that has been nonlinear (soniclib.c:397-399 short-circuits per write): the
                               // TSM stage keeps the speed it was last given instead of taking the global one
#define SPX_F_HANDED_IN 128    // ... and on such a stream the ring-buffer counts come with the job (handed_in, ring_bufs): its TSM input is a
                               // sequence of its own, n_in describes that one
struct SpxStreamDev {
  int64_t in_off, n_in, out_off, out_cap;  // n_in = input frames present so far (from the stream start)
  int64_t frame_off;    // index of this stream's analysis frame 0 in the per-frame arrays
  int32_t n_frames;     // analysis frames available with n_in samples
  int32_t frame_begin;  // analysis frames already done by earlier jobs
  int32_t channels;
  int32_t flags;
  float speed, nonlinear, feedback;
  int32_t first_tile;   // index of this job's first analysis tile
  // Streams that are written again after sonicFlushStream (soniclib.c:529-552 leaves the stream usable):
  int64_t tsm_shift;     // frames of flush padding the TSM stage has seen so far: TSM position = input frame + tsm_shift
  int32_t tension_skip;  // tension frames below this index that were not computed before the last flush never are:
                         // the shim's read index jumps to its write index there (soniclib.c:538-550)
  // The unit-level API of include/speedy.h (reference speedy.h:61-133) adds frames and asks for tensions one call at
  // a time, with its own time base:
  int32_t unit_time0;    // 0: frame j is added at time j + 1 (the shim, soniclib.c:288-296); 1: at time j (speedyAddData(.., j))
  int32_t tension_to;    // SPX_F_TENSION_RANGE: compute tension frames [tension_skip, tension_to)
  int32_t handed_in;     // SPX_F_HANDED_IN: ring buffers handed to the TSM stage before this job ...
  int32_t ring_bufs;     // ... and complete ring buffers that exist (n_in is the TSM input's length there, not the ring's)
  int32_t pad4;
};

// TSM-stage state (libsonic's stream struct) in absolute stream coordinates.
struct SpxWalkState {
  int64_t base;    // absolute index of the first sample still buffered in the TSM stage
  int64_t out_n;   // frames produced since the stream start
  int64_t avail;   // frames handed to the TSM stage so far
  int remaining;   // remainingInputToCopy
  int prevPeriod, prevMinDiff;
  int overflow;
  int prevPeriod_toggle;  // which of the two LDS lag-sum buffers the next pitch step uses (walk kernel internal)
  int steps;       // pitch searches (findPitchPeriod calls) since the stream started: the length of its chain of dependent
                   // steps (diagnostic -- bench.py's latency roofline divides the walk kernel's time by it)
};
// Everything a stream carries from one job to the next.
struct SpxStreamState {
  SpxWalkState w;
  float lp;        // energy low-pass state          (speedy.c:166,288)
  float lpf;       // difference low-pass state      (speedy.c:167,291)
  float cur_dur;   // speedy.c:170
  float des_dur;   // speedy.c:171
  float curSpeed;  // speed currently set in the TSM stage
  int handed;      // ring buffers handed to the TSM stage so far (readBufferFrameIndex, soniclib.c:73)
  int flush_remaining;  // frames the TSM stage held when its last flush began (the rate stage's expected length needs it)
  int flush_out_mark;   // ... and the frames it had produced by then (events of the same job before the flush included)
  int tension_first;    // time index of the first tension frame ever computed (-1: none yet): that call alone is "skipped"
                        // (skip_frame_count starts at 1, speedy.c:293,691) -- frame 0 unless a flush came before it
  int pad2_;
};

// Rate stage (sonicSetRate != 1: the dependency's adjustRate, oracle/orc_sonic.c adjust_rate): the TSM stage's output
// is the input sequence, one sample always stays behind as the next call's left neighbour.
#define SPX_RATE_MAX_CHANNELS 16
struct SpxRateState {
  int64_t tsm_seen;   // TSM output frames already taken
  int64_t final_n;    // frames of final output produced so far
  int32_t old_pos, new_pos;  // oldRatePosition / newRatePosition
  int32_t has_left;   // a sample is waiting in the pitch buffer
  int32_t overflow;
  int16_t left[SPX_RATE_MAX_CHANNELS];
};
// One workgroup: takes TSM output frames [tsm_seen, |*tsm_n|) from `tsm` and appends to `fin` -- resampled at new_rate /
// old_rate (already reduced to 14 bits), or copied when `bypass`.  `flush`: sonicIntFlushStream's truncation to the
// expected length and emptying of the pitch buffer.
void spx_launch_rate(SpxRateState* rs, const SpxStreamState* st, const int64_t* tsm_n, const int16_t* tsm, int16_t* fin,
                     int64_t fin_cap, int channels, int old_rate, int new_rate, float rate, int bypass, int flush,
                     hipStream_t hs);

// Per-analysis-frame record written by the analysis kernel and consumed by the walk kernel.
struct SpxFrameRec {
  float energy;  // sum_{i=1}^{N/2-1} s[i]^2, float, index order  (speedy.c:513-516 == :633-640)
  float lsd;     // local spectral difference with cur = this frame, last = previous (speedy.c:711-719)
};

// Scratch per analysis frame used by the walk kernel's frame-rate passes.
struct SpxTapsDev {
  float* tension;
  float* speed;
  float* features;
  float* spectrogram;
  float* normalized;
};

void spx_launch_analysis(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, int n_tiles,
                         const int16_t* in, SpxFrameRec* rec, SpxTapsDev taps, const int* tile_order, int* tile_flags,
                         hipStream_t st);
// The same kernel fed with explicit float frames (unit-level API): frame j = frames[j*W .. j*W + W), pre-emphasis state
// carried from the previous frame's last sample (speedy.c:416-425); preemph = false: the frame is windowed as it is
// (speedySpectrogram, speedy.c:438-473).
void spx_launch_analysis_frames(const SpxPlanDev& P, const SpxStreamDev* streams, int n_tiles, const float* frames,
                                bool preemph, SpxFrameRec* rec, SpxTapsDev taps, hipStream_t st);
// Frame-rate stage: energy / hysteresis / difference filters -> tension -> speed per tension frame (scratch[4k+3]) and
// the tension, speed and feature taps.  tile_flags / speed_ready: the concurrent-mode hand-off (nullptr = sequential).
void spx_launch_tension(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, SpxStreamState* states,
                        const SpxFrameRec* rec, float* scratch, SpxTapsDev taps, const int* tile_flags, int* speed_ready,
                        hipStream_t st);
// speedup_only: every job has speed > 1 and 0 <= nonlinear <= 1 (and a streamed job has never had another setting),
// so the time-scale stage only ever sees speeds >= 1: selects the walk kernel specialised for that.
// any_speed (round 5; the batch engine): the jobs do NOT all speed up, but every speed the time-scale stage can be given is a valid
// one below SPX_FAST_MAX_SPEED -- the speed-up kernel's instantiations that also run libsonic's insertPitchPeriod serve the batch
// (spx_walk_fast.hip, MC + 2) instead of the general kernel.
void spx_launch_walk(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, int max_channels,
                     const int16_t* in, int16_t* out, int64_t* n_out, SpxStreamState* states, const float* scratch,
                     const int* speed_ready, bool speedup_only, hipStream_t st, bool short_jobs = false, size_t lds_min = 0,
                     bool lean = false, bool any_speed = false, bool short_window = false);
size_t spx_analysis_lds_bytes(const SpxPlanDev& P);
int spx_analysis_ct_window(const SpxPlanDev& P);
// The DFT of the spec run on the host (same operation order as the kernel): used to build the Rader tables.
void spx_host_dft(int n, const int* radix, int nstages, const double* tw, const double* in, double* out);
size_t spx_walk_lds_bytes(const SpxPlanDev& P, int max_channels, bool speedup_only);
// What spx_launch_walk will do for a batch (kernel variant, waves and LDS per stream).
struct SpxWalkConfig {
  int mode;          // 0 general, 1 mono speed-up, 2 multi-channel speed-up
  bool fast_kernel;  // mode 1 on spx_walk_fast_kernel
  int nw;            // waves per stream of spx_walk_kernel
  int nwm, nwc, wcap;  // spx_walk_fast_kernel: search waves, output waves, window frames
  int waves;         // waves per stream of the kernel that will run
  size_t lds;        // its LDS bytes per stream
  bool slow;         // the fast kernel's instantiation that also serves speeds below 1
};
// short_jobs: the streams bring a few pitch steps each (coalesced sonic2.h writes): latency form whatever their number
// lean: no output waves (and the usual window) although the streams have a CU each -- the search waves do the output work:
// one walk wave per SIMD instead of two, which is what lets two analysis waves of 168 registers sit beside it (22.05 kHz)
// short_window: the 4 + 4 form with the usual 4096-frame window where it would take the long one (36.9 instead of 69.7 KB of LDS per
// stream: two calls' walk workgroups AND an analysis workgroup on a CU -- a mixed call whose walk kernels overlap the previous call's)
SpxWalkConfig spx_walk_config(const SpxPlanDev& P, int n_streams, int max_channels, bool speedup_only, bool short_jobs = false,
                              bool lean = false, bool any_speed = false, bool short_window = false);
// spx_walk_fast.hip
size_t spx_walk_fast_lds_bytes(const SpxPlanDev& P, int wcap);
bool spx_walk_fast_supports(const SpxPlanDev& P, int nwm);
void spx_launch_walk_fast(const SpxPlanDev& P, const SpxStreamDev* streams, int n_streams, const int16_t* in,
                          int16_t* out, int64_t* n_out, SpxStreamState* states, const float* scratch,
                          const int* speed_ready, int nwm, int nwc, int wcap, int max_channels, hipStream_t st, size_t lds_min = 0,
                          bool slow = false);
// n_out value of a stream whose producer kernel never delivered (concurrent mode poll limit): not an overflow
#define SPX_NOUT_LOST_PRODUCER INT64_MIN
size_t spx_tension_lds_bytes();
// VGPRs per lane the hardware allocates to a wave of the kernel that would be launched (hipFuncGetAttributes, rounded up
// to the allocation granule of 8): the engine's co-residency rule needs them
// (helper: allocated VGPRs of a kernel, cached per function -- the query is not free and the engine asks on every call)
int spx_kernel_vgprs(const void* fn, int* scratch_bytes = nullptr);   // spx_engine.hip (one cache, behind a mutex)
int spx_tension_vgprs();
int spx_analysis_vgprs(const SpxPlanDev& P, int* scratch_bytes = nullptr);
int spx_walk_vgprs(const SpxPlanDev& P, int n_streams, int max_channels, bool speedup_only, bool lean = false, bool any_speed = false,
                   bool short_window = false);
int spx_walk_kernel_regs(const SpxPlanDev& P, int n_streams, int max_channels, bool speedup_only, bool short_jobs, bool lean,
                         int* scratch_bytes, bool any_speed = false, bool short_window = false);
// speedyComputeSpeedFromTension (speedy.c:768-788) on the stream's state record: *speed_out = requested speed, the
// duration sums of the record advance.
void spx_launch_speed_from_tension(SpxStreamState* state, float tension, float Rg, float feedback, float* speed_out,
                                   hipStream_t st);
int spx_analysis_tile_frames();
int spx_analysis_small_tile_frames();
int spx_analysis_tiny_tile_frames();
int spx_analysis_prefers_small_tile(const SpxPlanDev& P);

// Shared, cached plan per (sample rate, hysteresis mode); owned by the library for the process lifetime.
struct spx_plan;
int64_t spx_internal_out_bound(const SpxPlanDev& P, int64_t n_in, float speed, bool nonlinear);
const SpxPlanDev* spx_internal_shared_plan(int sample_rate, int match_matlab);
// false above about 61 kHz: the analysis tile does not fit one CU's LDS, the plan serves linear (TSM-only) work
bool spx_internal_analysis_fits(const SpxPlanDev& d);
int64_t spx_internal_frames_for(const SpxPlanDev& d, int64_t n_in);
