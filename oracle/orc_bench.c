/* CPU baseline runner (TEST INFRASTRUCTURE, like everything under oracle/): the oracle's restatement of the
 * reference path -- the caller loop of speedy_wave.cc:154-242 (write 1000 frames, read, ... flush, drain) -- over
 * many independent streams on a pool of POSIX threads, one stream per task (streams share nothing:
 * soniclib.c:61-82, speedy.c:130-176).  bench.py compiles this file together with the oracle sources with
 * -O3 -march=native -ffp-contract=off on the machine it runs on and times it as `cpu_baseline` (kind "port").
 * Never linked into, or called from, the product. */
#include <malloc.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "orc_sonic2.h"

typedef struct {
  const short* in;      /* n_streams inputs of n_in frames each, back to back */
  long n_in;
  int n_streams, rate, channels, match_matlab, chunk;
  float speed, nonlinear, feedback;
  volatile int next;    /* task counter */
  long* out_frames;     /* per stream: frames produced */
  uint32_t* out_crc;    /* per stream: CRC-32 of the produced int16 bytes (for a cross-check against the GPU) */
} bench_job;

static uint32_t crc_table[256];
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    crc_table[i] = c;
  }
}
static uint32_t crc_update(uint32_t crc, const void* data, size_t n) {
  const unsigned char* p = (const unsigned char*)data;
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}

static void run_one(const bench_job* J, int i, short* scratch) {
  const int C = J->channels, chunk = J->chunk;
  const short* x = J->in + (size_t)i * J->n_in * C;
  orc_sonicStream s = orc_sonicCreateStream(J->rate, C, J->match_matlab);
  orc_sonicSetSpeed(s, J->speed);
  orc_sonicEnableNonlinearSpeedup(s, J->nonlinear);
  orc_sonicSetDurationFeedbackStrength(s, J->feedback);
  long produced = 0;
  uint32_t crc = 0;
  for (long pos = 0; pos < J->n_in; pos += chunk) {
    const int n = (int)((J->n_in - pos < chunk) ? J->n_in - pos : chunk);
    orc_sonicWriteShortToStream(s, x + pos * C, n);
    const int got = orc_sonicReadShortFromStream(s, scratch, chunk);
    crc = crc_update(crc, scratch, sizeof(short) * (size_t)got * C);
    produced += got;
  }
  orc_sonicFlushStream(s);
  int got;
  do {
    got = orc_sonicReadShortFromStream(s, scratch, chunk);
    crc = crc_update(crc, scratch, sizeof(short) * (size_t)got * C);
    produced += got;
  } while (got > 0);
  orc_sonicDestroyStream(s);
  J->out_frames[i] = produced;
  J->out_crc[i] = crc;
}

static void* worker(void* arg) {
  bench_job* J = (bench_job*)arg;
  short* scratch = (short*)malloc(sizeof(short) * (size_t)J->chunk * J->channels);
  for (;;) {
    const int i = __sync_fetch_and_add(&J->next, 1);
    if (i >= J->n_streams) break;
    run_one(J, i, scratch);
  }
  free(scratch);
  return NULL;
}

/* Runs n_streams streams of n_in frames each on `threads` threads; returns the wall seconds of the parallel
 * region (thread creation included, stream setup included: that is the reference's per-stream cost too). */
double orc_bench_run(const short* in, long n_in, int n_streams, int rate, int channels, float speed, float nonlinear,
                     float feedback, int match_matlab, int chunk, int threads, long* out_frames, uint32_t* out_crc) {
  crc_init();
  /* The oracle grows its FIFOs with realloc; by default glibc serves such blocks with mmap/munmap and trims the heap
   * on free, and with a few hundred threads those system calls (and the page faults of the fresh pages) serialise on
   * the process's address-space lock: 256 threads then run 9x one thread.  Keep the blocks in the per-thread arenas. */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 64 << 20);
  bench_job J;
  memset(&J, 0, sizeof(J));
  J.in = in; J.n_in = n_in; J.n_streams = n_streams; J.rate = rate; J.channels = channels;
  J.match_matlab = match_matlab; J.chunk = chunk; J.speed = speed; J.nonlinear = nonlinear; J.feedback = feedback;
  J.next = 0; J.out_frames = out_frames; J.out_crc = out_crc;
  if (threads < 1) threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int started = 0;
  for (int t = 0; t < threads; t++)
    if (pthread_create(&th[started], NULL, worker, &J) == 0) started++;
  if (started == 0) worker(&J);
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(th);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- the same runner with per-frame HASHES of the taps (tools/r11_probe.py `oracle` mode): what the GPU's spectrogram / features /
 * tension / speed taps must equal, frame for frame, on as many streams as the host has time for.  Test infrastructure.
 *   spec_hash[i * max_frames + j]  = sum over the 2W floats of analysis frame j's spectrogram row of their bit patterns (int32 -> int64)
 *   tap_hash[i * max_frames + k]   = sum_{c=0..14} (c + 1) bits(features[c]) + 31 bits(tension) + 37 bits(speed) of tension frame k
 *   n_spec[i], n_tap[i]            = rows seen                                                                                    */
static __thread int64_t* tl_spec;
static __thread int64_t* tl_tap;
static __thread int tl_max, tl_nspec, tl_ntap, tl_nbins;
static int64_t fbits(float f) { int32_t u; memcpy(&u, &f, 4); return (int64_t)u; }
static void cb_spec(orc_sonicStream s, int time, float* row) {   /* (rows in the order they are handed out: row j = analysis frame j) */
  (void)s; (void)time;
  if (tl_nspec >= tl_max) return;
  int64_t h = 0;
  for (int b = 0; b < tl_nbins; b++) h += fbits(row[b]);
  tl_spec[tl_nspec++] = h;
}
static void cb_feat(orc_sonicStream s, int time, float* f) {
  (void)s;
  if (time < 0 || time >= tl_max) return;
  int64_t h = 0;
  for (int c = 0; c < 15; c++) h += (int64_t)(c + 1) * fbits(f[c]);
  tl_tap[time] += h;
  if (time + 1 > tl_ntap) tl_ntap = time + 1;
}
static void cb_tension(orc_sonicStream s, int time, float v) { (void)s; if (time >= 0 && time < tl_max) tl_tap[time] += 31 * fbits(v); }
static void cb_speed(orc_sonicStream s, int time, float v) { (void)s; if (time >= 0 && time < tl_max) tl_tap[time] += 37 * fbits(v); }

typedef struct {
  bench_job J;
  int max_frames;
  int64_t* spec_hash;
  int64_t* tap_hash;
  int* n_spec;
  int* n_tap;
} hash_job;

static void* hash_worker(void* arg) {
  hash_job* H = (hash_job*)arg;
  bench_job* J = &H->J;
  short* scratch = (short*)malloc(sizeof(short) * (size_t)J->chunk * J->channels);
  for (;;) {
    const int i = __sync_fetch_and_add(&J->next, 1);
    if (i >= J->n_streams) break;
    const int C = J->channels, chunk = J->chunk;
    const short* x = J->in + (size_t)i * J->n_in * C;
    orc_sonicStream s = orc_sonicCreateStream(J->rate, C, J->match_matlab);
    tl_spec = H->spec_hash + (size_t)i * H->max_frames;
    tl_tap = H->tap_hash + (size_t)i * H->max_frames;
    tl_max = H->max_frames; tl_nspec = 0; tl_ntap = 0;
    tl_nbins = orc_sonicSpectrogramSize(s);
    orc_sonicSpectrogramCallback(s, cb_spec);
    orc_sonicFeaturesCallback(s, cb_feat);
    orc_sonicTensionCallback(s, cb_tension);
    orc_sonicSpeedCallback(s, cb_speed);
    orc_sonicSetSpeed(s, J->speed);
    orc_sonicEnableNonlinearSpeedup(s, J->nonlinear);
    orc_sonicSetDurationFeedbackStrength(s, J->feedback);
    for (long pos = 0; pos < J->n_in; pos += chunk) {
      const int n = (int)((J->n_in - pos < chunk) ? J->n_in - pos : chunk);
      orc_sonicWriteShortToStream(s, x + pos * C, n);
      (void)orc_sonicReadShortFromStream(s, scratch, chunk);
    }
    orc_sonicFlushStream(s);
    while (orc_sonicReadShortFromStream(s, scratch, chunk) > 0) { }
    orc_sonicDestroyStream(s);
    H->n_spec[i] = tl_nspec;
    H->n_tap[i] = tl_ntap;
  }
  free(scratch);
  return NULL;
}

double orc_bench_run_hashed(const short* in, long n_in, int n_streams, int rate, int channels, float speed, float nonlinear,
                            float feedback, int match_matlab, int chunk, int threads, int max_frames, int64_t* spec_hash,
                            int64_t* tap_hash, int* n_spec, int* n_tap) {
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 64 << 20);
  hash_job H;
  memset(&H, 0, sizeof(H));
  H.J.in = in; H.J.n_in = n_in; H.J.n_streams = n_streams; H.J.rate = rate; H.J.channels = channels;
  H.J.match_matlab = match_matlab; H.J.chunk = chunk; H.J.speed = speed; H.J.nonlinear = nonlinear; H.J.feedback = feedback;
  H.max_frames = max_frames; H.spec_hash = spec_hash; H.tap_hash = tap_hash; H.n_spec = n_spec; H.n_tap = n_tap;
  memset(spec_hash, 0, sizeof(int64_t) * (size_t)n_streams * max_frames);
  memset(tap_hash, 0, sizeof(int64_t) * (size_t)n_streams * max_frames);
  if (threads < 1) threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int started = 0;
  for (int t = 0; t < threads; t++)
    if (pthread_create(&th[started], NULL, hash_worker, &H) == 0) started++;
  if (started == 0) hash_worker(&H);
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(th);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
