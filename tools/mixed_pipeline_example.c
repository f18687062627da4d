/* INTEGRATION.md section 2, "batches that mix sample rates", as a program: plain C99 over include/speedy_hip.h, no HIP headers.
 *
 *   mixed_pipeline_example IN.raw STREAMS BATCHES DEPTH
 *
 * One GPU's kind of BASELINE configs[4] shard in small: IN.raw (int16 PCM) is handed over as STREAMS independent streams --
 * stream i at 16 kHz if i is even, else 22.05 kHz; mono if (i / 2) is even, else stereo (the samples read as interleaved pairs);
 * speed 1.5 if (i / 4) is even, else 3.5; nonlinear 1 -- first through ONE spx_batch_run_mixed call (the reference result), then
 * BATCHES times through an spx_pipeline created with spx_pipeline_create_mixed and SPX_PIPELINE_DEVICE_OUT: input resident in
 * device memory, outputs left there, the library's walk kernels of consecutive batches overlapping (round 6).  Every stream of every
 * batch must equal the plain call's.  Exit code 0 = ok.  Used by tests/test_gpu_cli.py and tools/asan_host.sh. */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "speedy_hip.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, spx_last_error()); return 2; } } while (0)

int main(int argc, char** argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: %s IN.raw STREAMS BATCHES DEPTH\n", argv[0]);
    return 1;
  }
  const int n = atoi(argv[2]), batches = atoi(argv[3]), depth = atoi(argv[4]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const size_t values = (size_t)bytes / 4 * 2;   /* an even number of int16 values: mono frames, or half as many stereo frames */
  int16_t* pcm = (int16_t*)malloc(values * sizeof(int16_t) + 2);
  if (n < 1 || batches < 1 || values < 2 || fread(pcm, sizeof(int16_t), values, f) != values) { fprintf(stderr, "bad arguments / short read\n"); return 1; }
  fclose(f);

  spx_plan_t plans[2] = {spx_plan_create(16000, 0), spx_plan_create(22050, 0)};
  if (!plans[0] || !plans[1]) { fprintf(stderr, "spx_plan_create: %s\n", spx_last_error()); return 2; }
  spx_stream_job* jobs = (spx_stream_job*)calloc((size_t)n, sizeof(spx_stream_job));
  int* plan_index = (int*)calloc((size_t)n, sizeof(int));
  int64_t out_values = 0;
  for (int i = 0; i < n; i++) {
    const int ch = ((i / 2) % 2 == 0) ? 1 : 2;
    plan_index[i] = i % 2;
    jobs[i].in_off = (int64_t)i * (int64_t)values;
    jobs[i].n_in = (int64_t)values / ch;
    jobs[i].channels = ch;
    jobs[i].speed = ((i / 4) % 2 == 0) ? 1.5f : 3.5f;
    jobs[i].nonlinear = 1.0f;
    jobs[i].feedback = 0.0f;
    jobs[i].out_off = out_values;   /* (the plain call's layout; the pipeline lays its outputs out itself) */
    jobs[i].out_cap = spx_plan_out_capacity_for(plans[plan_index[i]], jobs[i].n_in, jobs[i].speed, jobs[i].nonlinear);
    out_values += jobs[i].out_cap * ch;
  }
  const size_t in_values = (size_t)n * values;

  /* the input, resident in device memory: 64 values allocated behind the last stream's end (include/speedy_hip.h) */
  int16_t* host_in = (int16_t*)calloc(in_values + 64, sizeof(int16_t));
  for (int i = 0; i < n; i++) memcpy(host_in + (size_t)i * values, pcm, values * sizeof(int16_t));
  int16_t* d_in = (int16_t*)spx_device_alloc((in_values + 64) * sizeof(int16_t));
  int16_t* d_out = (int16_t*)spx_device_alloc((size_t)out_values * sizeof(int16_t) + 64);
  int64_t* d_nout = (int64_t*)spx_device_alloc((size_t)n * sizeof(int64_t));
  const size_t wsb = spx_batch_workspace_bytes_mixed(plans, 2, jobs, plan_index, n);
  void* d_ws = spx_device_alloc(wsb);
  if (!d_in || !d_out || !d_nout || !d_ws || !wsb) { fprintf(stderr, "device allocation: %s\n", spx_last_error()); return 2; }
  CHECK(spx_copy_to_device(d_in, host_in, (in_values + 64) * sizeof(int16_t), NULL));
  CHECK(spx_stream_synchronize(NULL));

  /* the reference result: one plain mixed call */
  CHECK(spx_batch_run_mixed(plans, 2, jobs, plan_index, n, d_in, d_out, d_nout, d_ws, wsb, NULL));
  int16_t* want = (int16_t*)malloc((size_t)out_values * sizeof(int16_t) + 2);
  int64_t* want_n = (int64_t*)malloc((size_t)n * sizeof(int64_t));
  CHECK(spx_copy_to_host(want, d_out, (size_t)out_values * sizeof(int16_t), NULL));
  CHECK(spx_copy_to_host(want_n, d_nout, (size_t)n * sizeof(int64_t), NULL));
  CHECK(spx_stream_synchronize(NULL));
  for (int i = 0; i < n; i++)
    if (want_n[i] < 0) { fprintf(stderr, "plain call, stream %d: output capacity exceeded\n", i); return 3; }

  spx_pipeline_t pipe = spx_pipeline_create_mixed(plans, 2, jobs, plan_index, n, depth, SPX_PIPELINE_DEVICE_OUT);
  if (!pipe) { fprintf(stderr, "spx_pipeline_create_mixed: %s\n", spx_last_error()); return 2; }
  const int d = spx_pipeline_depth(pipe);
  if (spx_pipeline_input_values(pipe) != in_values) { fprintf(stderr, "unexpected input size\n"); return 3; }
  int64_t* tickets = (int64_t*)malloc((size_t)batches * sizeof(int64_t));
  int64_t* got_n = (int64_t*)malloc((size_t)n * sizeof(int64_t));
  int16_t* got = NULL;
  size_t got_cap = 0;
  const double t0 = now_ms();
  double waited_ms = 0.0;
  for (int k = 0; k < batches + d - 1; k++) {
    if (k < batches) {
      tickets[k] = spx_pipeline_submit(pipe, d_in, /*in_is_device=*/1);
      if (tickets[k] < 0) { fprintf(stderr, "spx_pipeline_submit: %s\n", spx_last_error()); return 2; }
    }
    const int w = k - (d - 1);
    if (w < 0) continue;
    const int16_t* out;            /* device memory */
    const int64_t *offsets, *counts;   /* offsets: host memory; counts: device memory */
    CHECK(spx_pipeline_wait(pipe, tickets[w], &out, &offsets, &counts));
    const double t1 = now_ms();
    const size_t extent = (size_t)offsets[n];
    if (extent > got_cap) { free(got); got = (int16_t*)malloc(extent * sizeof(int16_t) + 2); got_cap = extent; }
    CHECK(spx_copy_to_host(got_n, counts, (size_t)n * sizeof(int64_t), NULL));
    CHECK(spx_copy_to_host(got, out, extent * sizeof(int16_t), NULL));
    CHECK(spx_stream_synchronize(NULL));
    for (int i = 0; i < n; i++) {
      const size_t len = (size_t)want_n[i] * (size_t)jobs[i].channels;
      if (offsets[i] % 32 != 0 || got_n[i] != want_n[i] || memcmp(got + offsets[i], want + jobs[i].out_off, len * sizeof(int16_t)) != 0) {
        fprintf(stderr, "batch %d stream %d differs from the plain mixed call (%lld frames against %lld)\n", w, i, (long long)got_n[i], (long long)want_n[i]);
        return 3;
      }
    }
    waited_ms += now_ms() - t1;   /* (the checking copies are not the pipeline's time) */
  }
  const double ms = (now_ms() - t0 - waited_ms) / batches;
  int64_t total = 0;
  for (int i = 0; i < n; i++) total += want_n[i];
  printf("%d batches of %d mixed streams (16 / 22.05 kHz, mono / stereo, 1.5x / 3.5x) x %zu values -> %lld frames out per batch; "
         "%.3f ms per batch, input and outputs resident on the device (%d buffer sets); every stream equals spx_batch_run_mixed's\n",
         batches, n, values, (long long)total, ms, d);
  fflush(stdout);
  spx_pipeline_destroy(pipe);
  spx_device_free(d_in); spx_device_free(d_out); spx_device_free(d_nout); spx_device_free(d_ws);
  spx_plan_destroy(plans[0]); spx_plan_destroy(plans[1]);
  free(jobs); free(plan_index); free(pcm); free(host_in); free(want); free(want_n); free(tickets); free(got_n); free(got);
  return 0;
}
