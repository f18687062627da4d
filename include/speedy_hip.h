/* speedy_hip.h — the thin C-ABI between host code and the hand-written HIP (gfx950) kernels of the
 * Speedy hot path:   analysis -> tension -> speed -> pitch-synchronous overlap-add.
 *
 * Plain C: pointers and sizes only, no HIP/torch types.  `hip_stream` arguments are a hipStream_t cast
 * to void* (NULL = the default stream).  Every pointer marked DEVICE must be HBM-resident on the
 * current device; HOST pointers are ordinary host memory.
 *
 * What each entry point replaces in the reference (google/speedy):
 *   spx_plan_create        speedyCreateStream's tables (speedy.c:206-299: window :256-258, FFT plan
 *                          :269-277, filter alphas :287-292) + libsonic's period limits
 *   spx_batch_analyze      speedyAddDataShort ... speedyComputeLocalEnergy for every frame of every
 *                          stream (speedy.c:553-565,416-425,438-473,510-523), the mono mix of
 *                          sonicSendDataToSpeedy (soniclib.c:262-287) and the per-frame part of
 *                          speedyComputeSpectralDifference (speedy.c:628-647,705-719)
 *   spx_batch_walk         the frame-sequential rest: filters, hysteresis, tension, speed
 *                          (speedy.c:73-76,590-610,682-728,752-788; soniclib.c:339-345) and the TSM stage the
 *                          reference reaches through sonicIntSetSpeed / sonicIntWriteShortToStream /
 *                          sonicIntFlushStream (soniclib.c:354,369,538-551)
 *   spx_batch_run          = analyze + walk: one whole sonicWriteShortToStream ... sonicFlushStream
 *                          life cycle (speedy_wave.cc:154-242) for N independent streams at once
 * The reference-compatible streaming API on top of this is include/sonic2.h.
 */
#ifndef SPEEDY_HIP_H_
#define SPEEDY_HIP_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef SPX_FEATURE_COUNT
#define SPX_FEATURE_COUNT 15 /* speedy.h:115 */
#endif

typedef struct spx_plan* spx_plan_t;

/* Per-stream job description (HOST array).  Offsets are in int16 elements from the batch base pointers. */
typedef struct {
  int64_t in_off;   /* first input sample of this stream, in shorts, inside `in`                      */
  int64_t n_in;     /* input length in multi-channel sample frames (sonic2.h:56-59 units)             */
  int64_t out_off;  /* first output sample, in shorts, inside `out`                                   */
  int64_t out_cap;  /* output capacity in multi-channel sample frames                                 */
  int32_t channels; /* interleaved channel count                                                      */
  float speed;      /* sonicSetSpeed                (soniclib.c:177)                                  */
  float nonlinear;  /* sonicEnableNonlinearSpeedup  (soniclib.c:555); 0 = linear: TSM stage only      */
  float feedback;   /* sonicSetDurationFeedbackStrength (soniclib.c:565)                              */
} spx_stream_job;

/* Optional debug taps (DEVICE, any may be NULL): the reference's five callbacks (sonic2.h:104-125).
 * Row r of stream s lives at index tap_off[s] + r, with tap_off[s] = sum over earlier streams of their
 * analysis-frame counts (spx_plan_frames).  Taps are excluded from the roofline byte count. */
typedef struct {
  float* tension;     /* [frames]        tension of read-buffer k  (soniclib.c:320-324)                */
  float* speed;       /* [frames]        speed handed to the TSM stage (soniclib.c:349-354)            */
  float* features;    /* [frames][15]    speedyGetInternalState at tension time (soniclib.c:325-329)   */
  float* spectrogram; /* [frames][N]     |DFT| of analysis frame j (soniclib.c:297-302)                */
  float* normalized;  /* [frames][N/2]   normalised spectrum used for tension k (speedy.c:673-675)     */
} spx_taps;

/* ---- errors: 0 = ok; otherwise a negative code, text via spx_last_error() (thread-local) ---- */
const char* spx_last_error(void);
int spx_abi_version(void);

/* ---- plan: tables for one (sample rate, hysteresis mode) pair ---- */
spx_plan_t spx_plan_create(int sample_rate, int match_matlab);
void spx_plan_destroy(spx_plan_t plan);
int spx_plan_frame_step(spx_plan_t plan);   /* speedyInputFrameStep, speedy.c:335-338 */
int spx_plan_window_size(spx_plan_t plan);  /* speedyInputFrameSize, speedy.c:330-333 */
int spx_plan_fft_size(spx_plan_t plan);     /* speedyFFTSize,        speedy.c:340-343 */
int spx_plan_future(spx_plan_t plan);       /* kTemporalHysteresisFuture, speedy.h:136-146 */
int spx_plan_max_required(spx_plan_t plan); /* libsonic maxRequired = 2*(rate/65) */
/* Number of analysis frames the shim schedules for n_in input frames (soniclib.c:440-444). */
int64_t spx_plan_frames(spx_plan_t plan, int64_t n_in);
/* Safe output capacity (frames) for n_in input frames at `speed`: n_in + slack for speed >= 1; for a slow-down
 * (n_in + flush padding) * 2/s + slack with s = speed for a linear job and s = 0.01 (the kMinimumSpeed clamp,
 * speedy.c:92,776) when the nonlinear path may drive the speed.  The three-argument form assumes nonlinear. */
int64_t spx_plan_out_capacity(spx_plan_t plan, int64_t n_in, float speed);
int64_t spx_plan_out_capacity_for(spx_plan_t plan, int64_t n_in, float speed, float nonlinear);

/* ---- batch execution ---- */
/* Bytes of DEVICE scratch needed for a batch (depends only on the jobs' lengths). */
size_t spx_batch_workspace_bytes(spx_plan_t plan, const spx_stream_job* jobs, int n_streams);

/* Whole life cycle for n_streams independent streams, asynchronous on hip_stream:
 *   in   DEVICE  int16 input samples (read once)
 *   out  DEVICE  int16 output samples (written once)
 *   n_out DEVICE int64[n_streams]: produced frames per stream (negative = out_cap overflow)
 * Returns 0 when the launches were enqueued. */
int spx_batch_run(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, const int16_t* in,
                  int16_t* out, int64_t* n_out, void* workspace, size_t workspace_bytes,
                  const spx_taps* taps, void* hip_stream);

/* spx_batch_run for a caller that issues batch after batch (round 4): consecutive calls are software-pipelined.  This call's
 * analysis and tension kernels are enqueued on a stream of the library's and start AT ONCE -- beside the walk kernel of the
 * previous call, which is still running on hip_stream -- and its own walk kernel follows on hip_stream with every speed ready.
 * Same results as spx_batch_run.  The caller's side of the contract:
 *   - `in` is complete in device memory when the call is made (the analysis does not wait for work queued on hip_stream), and
 *     nothing still pending on hip_stream touches workspace, taps, out or n_out: the call's staging, analysis and tension kernels
 *     write the workspace and the taps on the library's stream at once;
 *   - consecutive calls alternate (at least) two workspaces / out / n_out buffers; a call that hands over the previous call's
 *     workspace again waits for that call to finish instead (correct, no overlap);
 *   - hip_stream is the same stream call after call; when it has drained, every kernel of every call issued on it has.
 * Batches that do not fit the shape (kernels that do not fit side by side ...) run exactly as spx_batch_run would -- with one
 * exception since round 6: a call of more than two streams per CU (two time ranges, kernels in sequence) still starts its PRODUCERS
 * at once on the library's stream, so that call k + 1's first analysis range runs beside call k's last walk range (2 048 x 10 s per
 * call, call after call: 6.21 -> 5.85 ms); the contract above is what makes that legal. */
int spx_batch_run_ahead(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, const int16_t* in,
                        int16_t* out, int64_t* n_out, void* workspace, size_t workspace_bytes,
                        const spx_taps* taps, void* hip_stream);

/* spx_batch_run_ahead with the WALK kernels of consecutive calls overlapping as well: a call's walk kernel runs on a stream of
 * the library's (two, taking turns) and hip_stream only waits for it, so the walk of call k + 1 starts as soon as its own
 * speeds are there, beside the walk of call k -- two walk workgroups per CU run at nearly full speed each, and the longest
 * chains of one batch no longer hold the next batch back (BASELINE configs[3]: 1.60 -> 1.39 ms per batch with two buffer sets
 * taking turns, 1.16 with three: a call's producers then start while the walk kernels of both previous calls are running, and
 * the walk kernel is launched in its lean form, which leaves the analysis kernel two waves per SIMD beside two walk workgroups).
 * The price is a RELAXED stream order, on top of spx_batch_run_ahead's contract:
 *   - work the caller enqueues on hip_stream AFTER call k is ordered behind call k's kernels as always (it sees call k's output),
 *     but call k + 1's walk kernel is ordered only behind what was on hip_stream when call k was MADE: whatever the caller
 *     enqueues between call k and call k + 1 must not touch call k + 1's buffers (two buffer sets taking turns, each consumed
 *     right behind its own call, satisfy this by construction; a call that hands over the previous call's out / n_out or
 *     workspace again is ordered behind everything, as spx_batch_run_ahead would order it).
 * Everything else as spx_batch_run_ahead; shapes outside the mode run as spx_batch_run would. */
int spx_batch_run_overlapped(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, const int16_t* in,
                             int16_t* out, int64_t* n_out, void* workspace, size_t workspace_bytes,
                             const spx_taps* taps, void* hip_stream);

/* The same for input that is still on its way when the call is made (a host-to-device copy on another stream):
 * in_ready_event is a hipEvent_t the caller has recorded behind whatever completes `in`; the call's producers wait for it
 * (NULL: as spx_batch_run_ahead). */
int spx_batch_run_ahead_when(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, const int16_t* in,
                             int16_t* out, int64_t* n_out, void* workspace, size_t workspace_bytes,
                             const spx_taps* taps, void* hip_stream, void* in_ready_event);

/* ---- the owning pipeline (round 5): batch after batch of one shape, host memory to host memory ----
 * What the reference's caller loop does per stream -- write a chunk, read what is ready, again (speedy_wave.cc:154-242) -- for a
 * caller that feeds BATCHES: spx_pipeline_submit hands over one batch of input (host or device memory), spx_pipeline_wait returns
 * that batch's output.  The object OWNS everything in between: `depth` sets of device buffers (input, output, workspace), pinned
 * host buffers for the output, its HIP streams and events.  Inside, the library issues the host-to-device copy, the three kernels
 * in the overlapped order of spx_batch_run_overlapped (batch k + 1's analysis beside batch k's walk kernel, the walk kernels of
 * consecutive batches overlapping) and a gather kernel that writes the produced frames of all streams, densely packed, straight
 * into pinned host memory -- no device-to-host copy is enqueued and no host thread waits inside submit, so copies in, kernels and
 * copies out of up to `depth` batches are in flight at once.  The relaxed stream order that spx_batch_run_overlapped asks its
 * caller to respect is an implementation detail here: nobody else can touch the buffers.
 *   jobs      the shape of every batch: n_in, channels, speed, nonlinear, feedback and in_off (where stream i starts in a batch's
 *             input, in int16 values); out_off / out_cap are ignored (the pipeline lays the outputs out itself, capacity =
 *             spx_plan_out_capacity_for)
 *   plans / plan_index   as in spx_batch_run_mixed, for batches that mix sample rates (spx_pipeline_create: one plan)
 *   depth     buffer sets, 2 .. 8 (0 = the default, 4); 3 or more let the walk kernels of consecutive batches overlap fully
 *   flags     SPX_PIPELINE_DEVICE_OUT: the outputs stay in device memory (no gather, no copy out).  Such a pipeline's calls are
 *             detached from its run stream, and since round 6 that holds for mixed-rate batches too: the groups' walk kernels of
 *             consecutive batches overlap on the library's walk streams (BASELINE configs[4] shard: 1.87 - 1.97 -> 1.56 ms per batch)
 * Not thread-safe: one host thread (or external locking) per pipeline; several pipelines may be alive at once. */
typedef struct spx_pipeline* spx_pipeline_t;
#define SPX_PIPELINE_DEVICE_OUT 1u
spx_pipeline_t spx_pipeline_create(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, int depth, unsigned flags);
spx_pipeline_t spx_pipeline_create_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index,
                                         int n_streams, int depth, unsigned flags);
void spx_pipeline_destroy(spx_pipeline_t p);
int spx_pipeline_depth(spx_pipeline_t p);
/* int16 values of one batch's input: max over streams of in_off + n_in * channels. */
size_t spx_pipeline_input_values(spx_pipeline_t p);
/* Pinned host staging (spx_pipeline_input_values() int16 values) for the batch the NEXT submit hands over: a caller that produces
 * its input there saves a host copy and gets the full link rate.  Blocks until the copy that last read this buffer (`depth`
 * submits ago) has finished.  Any other host pointer (pinned or pageable) may be given to submit as well. */
int16_t* spx_pipeline_host_input(spx_pipeline_t p);
/* Hand over one batch.  in: HOST memory (in_is_device = 0) -- must stay unchanged until the batch's input has been copied
 * (spx_pipeline_input_consumed, or the batch's spx_pipeline_wait) -- or DEVICE memory (in_is_device = 1), complete when the call is
 * made and unchanged until the batch's spx_pipeline_wait returns; a device buffer is passed to the kernels as it is, and their
 * aligned window loads may touch up to 64 int16 values behind the last stream's end: it must be ALLOCATED for
 * spx_pipeline_input_values() + 64 values (the content of the padding does not matter; the same holds for the `in` of the
 * spx_batch_* calls).  Returns the batch's ticket (0, 1, 2 ...) or a negative error.
 * At most `depth` batches are in flight: the call waits for the batch `depth` tickets back first. */
int64_t spx_pipeline_submit(spx_pipeline_t p, const int16_t* in, int in_is_device);
/* Blocks until the input handed over with `ticket` may be overwritten: the copy in has finished (host input), the batch's kernels
 * have finished (device input: they read it to the end).  0, or a negative error (unknown ticket). */
int spx_pipeline_input_consumed(spx_pipeline_t p, int64_t ticket);
/* Wait for a batch.  On return
 *   *out      the output samples: pinned HOST memory owned by the pipeline (DEVICE memory with SPX_PIPELINE_DEVICE_OUT)
 *   *offsets  HOST int64[n_streams + 1]: stream i's samples are out[offsets[i] .. offsets[i] + counts[i] * channels[i]); every
 *             offset is a multiple of 32 values (64 bytes); offsets[n_streams] = the extent of the packed output
 *   *counts   int64[n_streams]: produced frames per stream as spx_batch_run's n_out (negative = capacity overflow); HOST memory,
 *             DEVICE memory with SPX_PIPELINE_DEVICE_OUT
 * valid until `depth` more batches have been submitted.  Tickets may be waited for in any order, each at most once per buffer
 * life; a ticket whose buffers have been handed to a later batch returns an error. */
int spx_pipeline_wait(spx_pipeline_t p, int64_t ticket, const int16_t** out, const int64_t** offsets, const int64_t** counts);
/* Pinned host memory for callers without HIP headers (the input of spx_pipeline_submit at the full link rate). */
void* spx_host_alloc(size_t bytes);
void spx_host_free(void* p);

/* The two stages separately (same arguments); spx_batch_run = analyze then walk. */
int spx_batch_analyze(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, const int16_t* in,
                      void* workspace, size_t workspace_bytes, const spx_taps* taps, void* hip_stream);
int spx_batch_walk(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, const int16_t* in,
                   int16_t* out, int64_t* n_out, void* workspace, size_t workspace_bytes,
                   const spx_taps* taps, void* hip_stream);

/* One call for a batch whose streams differ in SAMPLE RATE (the reference fixes the rate per handle, soniclib.c:93 /
 * speedy.c:213-214, so any mix can be alive at once; BASELINE configs[4] mixes 16 kHz and 22.05 kHz).  plans[k] serves the
 * jobs with plan_index[i] == k (all plans on the current device, at most 8); in / out / n_out as in spx_batch_run, n_out
 * indexed like jobs.  All groups are launched together -- the walk workgroups of every group are resident at the same
 * time -- forked from and joined to hip_stream.  Results per stream are those of spx_batch_run on its own plan. */
size_t spx_batch_workspace_bytes_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs,
                                       const int* plan_index, int n_streams);
int spx_batch_run_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index,
                        int n_streams, const int16_t* in, int16_t* out, int64_t* n_out, void* workspace,
                        size_t workspace_bytes, void* hip_stream);

/* The same with the debug taps.  Row order: streams grouped by plan index ascending, job order inside a group; row r of a
 * stream lives at index tap_off[s] + r with tap_off[s] = the analysis-frame counts (spx_plan_frames on the stream's own plan,
 * 0 for a linear job) of every stream in front of it in THAT order.  spectrogram / normalized rows are as wide as the
 * stream's own plan makes them (N, W): their element offsets are the sums of frames x width of the streams in front. */
int spx_batch_run_mixed_taps(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index,
                             int n_streams, const int16_t* in, int16_t* out, int64_t* n_out, void* workspace,
                             size_t workspace_bytes, const spx_taps* taps, void* hip_stream);
/* spx_batch_run_mixed for a caller that issues mixed batch after mixed batch (see spx_batch_run_ahead: same contract -- `in`
 * complete when the call is made, two workspaces / out / n_out taking turns, one hip_stream, and the same plans[0] call after
 * call): every group's analysis and tension kernels start at once on a stream of the library's, beside the previous call's walk
 * kernels.  Calls of more streams than the device has CUs run exactly as spx_batch_run_mixed would. */
int spx_batch_run_mixed_ahead(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index,
                              int n_streams, const int16_t* in, int16_t* out, int64_t* n_out, void* workspace,
                              size_t workspace_bytes, void* hip_stream);

/* Timing hooks for bench.py: while enabled, every spx_batch_run records HIP events on hip_stream around
 * each of its kernels (no host synchronisation is added to the call).  spx_timing_collect waits for the
 * recorded events, returns the summed kernel milliseconds (over all launches of each kernel) and the number of
 * spx_batch_run calls since the last collect. */
void spx_set_timing(int enabled);
/* spx_batch_run splits every stream into `chunks` consecutive time ranges and overlaps the analysis of range c+1 (on an internal
 * HIP stream) with the walk of range c (on hip_stream); results are identical for any value, the state record is carried exactly
 * as in the streaming API.  NEVER CALLED: the library chooses -- one range up to two streams per CU (on MI355X the extra launch
 * tails cost more than the overlap gains at 256 x 10 s), two above that.  A count set here is binding for every later call of the
 * process, 1 included (a call of 2 048 streams then runs as one range: 6.31 instead of 6.21 ms). */
void spx_set_pipeline_chunks(int chunks);
/* Concurrent mode of spx_batch_run (default on): the analysis kernel and the frame-rate (tension) kernel run on two
 * internal HIP streams, the walk kernel at the same time on hip_stream; tiles of frames and then per-frame speeds are
 * handed over through device-scope flags as they become ready.  Results are identical with it on or off. */
void spx_set_concurrent(int on);
int spx_timing_collect(double* sum_ms_analyze, double* sum_ms_walk, int* n_calls);
/* "analysis;tension;walk": the kernels (template arguments included, as a profiler prints them) that serve a batch of
 * n_streams streams with at most max_channels channels; speedup_only = every job has speed > 1. */
const char* spx_batch_kernel_names(spx_plan_t plan, int n_streams, int max_channels, int speedup_only);
/* The same with the walk kernel in its lean form (what spx_batch_run_overlapped launches when three or more workspaces take turns). */
const char* spx_batch_kernel_names_lean(spx_plan_t plan, int n_streams, int max_channels, int speedup_only);
/* Diagnostics: the form of the walk kernel launched last in this process, 16 * search waves + output waves (68 = the usual
 * 4 + 4, 64 = the lean form of the concurrent mode at 22.05 kHz mono, 32 = the throughput form), 0 = the general kernel. */
int spx_debug_last_walk_form(void);
/* Diagnostics: allocated VGPRs (hipFuncGetAttributes, rounded up to the granule of 8) of the kernels whose register budgets
 * decide the concurrent mode.  which: 0 tension, 1 walk 16 kHz mono (4 + 4 waves, long window), 2 walk 22.05 kHz mono lean
 * (4 + 0), 3 analysis 16 kHz, 4 analysis 22.05 kHz, 5 walk 16 kHz multi-channel (4 + 4).  -1 for an unknown index. */
int spx_debug_kernel_vgprs(int which);
/* Diagnostics: the walk kernel a batch of this shape would be served by -- out[0] allocated VGPRs, out[1] scratch bytes per lane
 * (spilled registers), out[2] LDS bytes per workgroup, out[3] its form (as spx_debug_last_walk_form), out[4] waves per workgroup --
 * and the analysis kernel of a sample rate (out[0 .. 2] alike).  tests/test_gpu_parity.py pins these per (rate, channels, batch
 * size): the engine's choice of launch mode is arithmetic over them. */
int spx_debug_walk_info(int sample_rate, int channels, int n_streams, int speedup_only, int short_jobs, int lean, int* out5);
int spx_debug_analysis_info(int sample_rate, int* out3);
/* Diagnostics: the 22 resource numbers the engine's launch-mode decision (speedy_amd/csrc/spx_mode.h, a pure function) is fed for a
 * batch of this shape: CUs, LDS per CU, the walk kernel's form as picked and in its lean form (LDS, waves, VGPRs, fast kernel?, output
 * waves; lean form exists?), the tension kernel's LDS and VGPRs, the analysis tiles (the plan's, 16, 8 frames) and the analysis
 * kernel's LDS and VGPRs with the plan's tile and with the small one.  profiles/kernel_resources.json holds them for the shapes
 * tests/test_mode_table.py decides on the CPU. */
int spx_debug_mode_resources(int sample_rate, int channels, int n_streams, int speedup_only, long long* out22);
/* Diagnostics: the walk kernel's five-instruction division for the candidate step lengths against the IEEE quotient, for
 * `denominators` random speeds (speed - 1 log-uniform in [2^exp_lo, 2^exp_hi)) x every count 1 .. 4096 x both numerator forms;
 * returns the number of mismatches (0 is the only acceptable answer), -1 on a runtime error. */
long long spx_debug_fdiv_check(unsigned seed, unsigned denominators, int exp_lo, int exp_hi);
/* The cross-fade's quotient (spx_walk_fast.hip xfade_quot: reciprocal by Newton steps, fused multiply-add, truncating conversion)
 * against the integer division, for every n in [n_lo, n_hi] (1 .. 4096) and every numerator k n + {-1, 0, 1}, |k| <= 32768.
 * Returns the number of mismatches, -1 on a bad range or a runtime error. */
long long spx_debug_xfade_check(int n_lo, int n_hi);
/* The same for the analysis kernel's scale-free fp32 / fp64 divisions and its fp64 square root (spx_log.h): `threads` x
 * `per_thread` pseudo-random operands of the ranges the kernel feeds them; the number of results that differ from the IEEE
 * sequences' (0 is the only acceptable answer), -1 on a runtime error. */
long long spx_debug_arith_check(unsigned seed, unsigned threads, unsigned per_thread);
/* The analysis kernel's natural log ("log spec v2", DESIGN.md 4a: its argument is always a float quotient, speedy.c:716-717) over
 * EVERY positive normal float: block b = the 2^20 float patterns b << 20 ..; sums[b] (HOST uint64[2040], blocks 8 .. 2039 filled) =
 * the sum of the results' bit patterns modulo 2^64.  The oracle computes the same sums (oracle/orc_logcheck.c): equal sums, block
 * for block, are bit-equality on the whole domain.  Returns 0, negative on error. */
int spx_debug_log_check(unsigned first_block, unsigned end_block, unsigned long long* sums);
/* Diagnostics: 1 if the last spx_batch_run / analyze+walk call of this process took the concurrent three-kernel mode, 2 if it was
 * pipelined with the previous call (spx_batch_run_ahead), 0 if it launched its kernels in sequence (another process holds the
 * device's concurrent-mode lock, a tuning variable, the batch shape). */
int spx_debug_last_call_concurrent(void);
/* Sum over the same calls of the frame-rate (tension) kernel's time, as of the last spx_timing_collect. */
double spx_timing_last_tension_ms(void);

/* Diagnostics: the number of pitch searches (libsonic's findPitchPeriod calls) each stream's walk has run since the stream started (a batch job starts it) -- the length of the stream's chain of dependent steps, which is what bounds a call with one stream per CU.
 *   workspace  DEVICE  the workspace that call ran with;  steps  HOST  int32[n_streams].  Waits for hip_stream first. */
int spx_batch_read_steps(spx_plan_t plan, const spx_stream_job* jobs, int n_streams, const void* workspace,
                         int32_t* steps, void* hip_stream);
int spx_batch_read_steps_mixed(const spx_plan_t* plans, int n_plans, const spx_stream_job* jobs, const int* plan_index,
                               int n_streams, const void* workspace, int32_t* steps, void* hip_stream);

/* ---- plain device-memory helpers (so that C/C++ hosts need no HIP headers) ---- */
void* spx_device_alloc(size_t bytes);
void spx_device_free(void* p);
/* Gather the produced frames of a finished batch into one contiguous buffer so that a single device-to-host copy
 * moves them: packed[offsets[i] .. offsets[i+1]) = the n_out[i]*channels[i] int16 values of stream i.
 *   jobs      HOST  the job table the batch ran with (out_off, channels)
 *   out,n_out DEVICE as written by spx_batch_run
 *   packed    DEVICE int16, at least sum(max(n_out,0)*channels) values (the size of `out` always suffices)
 *   offsets   DEVICE int64[n_streams + 1], written here: exclusive prefix sums in int16 values
 * Runs on hip_stream after the batch; returns 0 or a negative error. */
int spx_batch_pack_outputs(const spx_stream_job* jobs, int n_streams, const int16_t* out, const int64_t* n_out,
                           int16_t* packed, int64_t* offsets, void* hip_stream);
int spx_copy_to_device(void* dst, const void* src, size_t bytes, void* hip_stream);
int spx_copy_to_host(void* dst, const void* src, size_t bytes, void* hip_stream);
int spx_stream_synchronize(void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEEDY_HIP_H_ */
