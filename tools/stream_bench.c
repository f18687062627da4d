/* Many sonicStream handles on one host thread through include/sonic2.h (plain C99, no HIP headers): the many-handle
 * throughput of the drop-in API.  Every round writes `chunk` frames to each handle and then reads from each -- the
 * reference CLI's loop (speedy_wave.cc:199-231) turned sideways, as a server with N live streams runs it.
 *
 *   stream_bench STREAMS [SECONDS=10] [CHUNK=1000] [SPEED=3.5] [NONLINEAR=1] [ORDER=rounds|percall|threads:T] [RATE=16000]
 *
 * ORDER=percall reads right after each write, handle by handle -- the reference's own call order (speedy_wave.cc:199-220,
 * sonic_test.cc:384-392) on one thread: one launch sequence per handle per write.
 * ORDER=threads:T (round 6): T host threads, each with STREAMS / T handles of its own, each running that same write -> read loop
 * over its handles -- a server's shape for an API whose streams are independent but not thread-safe (sonic2.h:54-84).  The
 * library combines what the threads stage into common launch sequences (sonic2_pool.hip, flat combining).
 * Prints one line of JSON. */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sonic2.h"

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* speech-like test signal: a pitch-modulated harmonic source under a syllable envelope with pauses, plus a little noise */
static void synth(short* x, int n, int rate, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  double ph = 0.0, f0 = 110.0 + (double)(seed % 90);
  int i, h;
  for (i = 0; i < n; i++) {
    double t = (double)i / rate, env, v = 0.0;
    s = s * 1664525u + 1013904223u;
    f0 += ((double)((s >> 16) & 0xff) - 127.5) * 0.004;
    if (f0 < 90.0) f0 = 90.0;
    if (f0 > 250.0) f0 = 250.0;
    ph += 2.0 * 3.14159265358979 * f0 / rate;
    env = 0.5 * (1.0 + sin(2.0 * 3.14159265358979 * (3.0 + (seed % 3)) * t));
    if (env < 0.2) env = 0.0;
    for (h = 1; h <= 10; h++) v += sin(h * ph) / h;
    v = 0.22 * env * v + 0.004 * ((double)((s >> 8) & 0xffff) / 32768.0 - 1.0);
    x[i] = (short)(v * 32767.0 * 0.4);
  }
}

typedef struct {
  sonicStream* h;       /* this thread's handles */
  int n_h, first;       /* their number, index of the first (source selection) */
  short** src;
  int n_src, n, chunk;
  long long total_out, rounds;
  unsigned long long crc;
  int failed;
  pthread_barrier_t* start;
} worker_t;

/* one thread of ORDER=threads:T: write a chunk to a handle, read what is ready, next handle (speedy_wave.cc:199-220 per handle) */
static void* thread_main(void* arg) {
  worker_t* w = (worker_t*)arg;
  short* out = (short*)malloc(sizeof(short) * (size_t)(w->chunk > 8192 ? w->chunk : 8192));
  int pos, i, got;
  pthread_barrier_wait(w->start);
  for (pos = w->chunk; pos < w->n; pos += w->chunk) {
    const int k = pos + w->chunk <= w->n ? w->chunk : w->n - pos;
    for (i = 0; i < w->n_h; i++) {
      if (sonicWriteShortToStream(w->h[i], w->src[(w->first + i) % w->n_src] + pos, k) != 1) { w->failed = 1; free(out); return NULL; }
      got = sonicReadShortFromStream(w->h[i], out, w->chunk);
      w->total_out += got;
      if (got > 0) w->crc = w->crc * 1315423911ull + (unsigned short)out[got - 1];
    }
    w->rounds++;
  }
  for (i = 0; i < w->n_h; i++) sonicFlushStream(w->h[i]);
  for (i = 0; i < w->n_h; i++)
    while ((got = sonicReadShortFromStream(w->h[i], out, 8192)) > 0) w->total_out += got;
  free(out);
  return NULL;
}

int main(int argc, char** argv) {
  const int streams = argc > 1 ? atoi(argv[1]) : 16;
  const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
  const int chunk = argc > 3 ? atoi(argv[3]) : 1000;
  const float speed = argc > 4 ? (float)atof(argv[4]) : 3.5f;
  const float nonlinear = argc > 5 ? (float)atof(argv[5]) : 1.0f;
  const int percall = argc > 6 && strcmp(argv[6], "percall") == 0;
  const int threads = (argc > 6 && strncmp(argv[6], "threads:", 8) == 0) ? atoi(argv[6] + 8) : 0;
  const int rate = argc > 7 ? atoi(argv[7]) : 16000;
  const int n = (int)(seconds * rate);
  const int n_src = 8;
  short* src[8];
  sonicStream* h = (sonicStream*)calloc((size_t)streams, sizeof(sonicStream));
  short* out = (short*)malloc(sizeof(short) * (size_t)(chunk > 8192 ? chunk : 8192));
  long long total_out = 0, rounds = 0;
  unsigned long long crc = 0;
  double t0, t1, worst = 0.0;
  int i, pos;
  if (streams < 1 || n < 1 || chunk < 1 || !h || !out) return 2;
  for (i = 0; i < n_src; i++) {
    src[i] = (short*)malloc(sizeof(short) * (size_t)n);
    synth(src[i], n, rate, 17u + (unsigned)i);
  }
  for (i = 0; i < streams; i++) {
    h[i] = sonicCreateStream(rate, 1);
    if (!h[i]) { fprintf(stderr, "sonicCreateStream: %s\n", speedyHipLastError()); return 1; }
    sonicSetSpeed(h[i], speed);
    sonicEnableNonlinearSpeedup(h[i], nonlinear);
    sonicSetDurationFeedbackStrength(h[i], 0.0f);
  }
  /* one warm-up round on every handle (plans, pinned areas, code objects), not timed */
  for (i = 0; i < streams; i++) sonicWriteShortToStream(h[i], src[i % n_src], chunk < n ? chunk : n);
  for (i = 0; i < streams; i++) total_out += sonicReadShortFromStream(h[i], out, 8192);
  total_out = 0;
  if (threads > 0) {
    const int T = threads < streams ? threads : streams;
    pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    worker_t* w = (worker_t*)calloc((size_t)T, sizeof(worker_t));
    pthread_barrier_t start;
    int t, bad = 0;
    pthread_barrier_init(&start, NULL, (unsigned)T + 1u);
    for (t = 0; t < T; t++) {
      const int a = (int)((long long)streams * t / T), b = (int)((long long)streams * (t + 1) / T);
      w[t].h = h + a; w[t].n_h = b - a; w[t].first = a; w[t].src = src; w[t].n_src = n_src; w[t].n = n; w[t].chunk = chunk;
      w[t].start = &start;
      if (pthread_create(&th[t], NULL, thread_main, &w[t]) != 0) { fprintf(stderr, "pthread_create failed\n"); return 1; }
    }
    pthread_barrier_wait(&start);
    t0 = now_s();
    for (t = 0; t < T; t++) pthread_join(th[t], NULL);
    t1 = now_s();
    for (t = 0; t < T; t++) { total_out += w[t].total_out; crc ^= w[t].crc * (unsigned long long)(t + 1); bad |= w[t].failed; if (w[t].rounds > rounds) rounds = w[t].rounds; }
    if (bad) { fprintf(stderr, "a thread's write failed: %s\n", speedyHipLastError()); return 1; }
    {
      const double dt = t1 - t0, in_frames = (double)streams * (double)(n - chunk);
      unsigned long long runs = 0, jobs = 0;
      speedyHipPoolStats(&runs, &jobs);
      printf("{\"streams\": %d, \"threads\": %d, \"rate\": %d, \"seconds_each\": %.1f, \"chunk\": %d, \"speed\": %.2f, \"nonlinear\": %.2f, "
             "\"order\": \"threads\", \"wall_s\": %.4f, \"msamples_per_s\": %.2f, \"x_realtime_per_stream\": %.1f, "
             "\"us_per_round\": %.1f, \"frames_out\": %lld, \"launch_sequences\": %llu, \"handles_per_sequence\": %.1f, \"crc\": %llu}\n",
             streams, T, rate, seconds, chunk, speed, nonlinear, dt, in_frames / dt * 1e-6, ((double)(n - chunk) / rate) / dt,
             dt / (double)(rounds ? rounds : 1) * 1e6, total_out, runs, runs ? (double)jobs / (double)runs : 0.0, crc);
    }
    for (i = 0; i < streams; i++) sonicDestroyStream(h[i]);
    return 0;
  }
  t0 = now_s();
  for (pos = chunk; pos < n; pos += chunk) {
    const int w = pos + chunk <= n ? chunk : n - pos;
    const double r0 = now_s();
    double r1;
    if (percall) {
      for (i = 0; i < streams; i++) {
        int got;
        if (sonicWriteShortToStream(h[i], src[i % n_src] + pos, w) != 1) { fprintf(stderr, "write: %s\n", speedyHipLastError()); return 1; }
        got = sonicReadShortFromStream(h[i], out, chunk);
        total_out += got;
        if (got > 0) crc = crc * 1315423911ull + (unsigned short)out[got - 1];
      }
    } else {
      for (i = 0; i < streams; i++)
        if (sonicWriteShortToStream(h[i], src[i % n_src] + pos, w) != 1) { fprintf(stderr, "write: %s\n", speedyHipLastError()); return 1; }
      for (i = 0; i < streams; i++) {
        const int got = sonicReadShortFromStream(h[i], out, chunk);
        total_out += got;
        if (got > 0) crc = crc * 1315423911ull + (unsigned short)out[got - 1];
      }
    }
    r1 = now_s() - r0;
    if (r1 > worst) worst = r1;
    rounds++;
  }
  for (i = 0; i < streams; i++) sonicFlushStream(h[i]);
  for (i = 0; i < streams; i++) {
    int got;
    while ((got = sonicReadShortFromStream(h[i], out, 8192)) > 0) total_out += got;
  }
  t1 = now_s();
  {
    const double dt = t1 - t0, in_frames = (double)streams * (double)(n - chunk);
    unsigned long long runs = 0, jobs = 0;
    speedyHipPoolStats(&runs, &jobs);
    printf("{\"streams\": %d, \"rate\": %d, \"seconds_each\": %.1f, \"chunk\": %d, \"speed\": %.2f, \"nonlinear\": %.2f, "
           "\"order\": \"%s\", \"wall_s\": %.4f, \"msamples_per_s\": %.2f, \"x_realtime_per_stream\": %.1f, "
           "\"us_per_round\": %.1f, \"worst_round_us\": %.1f, \"frames_out\": %lld, \"launch_sequences\": %llu, "
           "\"handles_per_sequence\": %.1f, \"crc\": %llu}\n",
           streams, rate, seconds, chunk, speed, nonlinear, percall ? "percall" : "rounds", dt, in_frames / dt * 1e-6,
           ((double)(n - chunk) / rate) / dt, dt / (double)(rounds ? rounds : 1) * 1e6, worst * 1e6, total_out, runs,
           runs ? (double)jobs / (double)runs : 0.0, crc);
  }
  for (i = 0; i < streams; i++) sonicDestroyStream(h[i]);
  return 0;
}
