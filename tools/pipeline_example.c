/* INTEGRATION.md section 2, "batch after batch", as a program: plain C99 over include/speedy_hip.h, no HIP headers, no C++.
 *
 *   pipeline_example IN.raw RATE CHANNELS SPEED NONLINEAR COPIES BATCHES DEPTH OUT.raw
 *
 * The caller loop of the reference (speedy_wave.cc:154-242: write a chunk, read what is ready, again) with a BATCH of streams
 * as its unit: IN.raw (interleaved int16 PCM) is handed over COPIES times per batch as independent streams, BATCHES batches go
 * through one spx_pipeline of DEPTH buffer sets -- spx_pipeline_submit, and DEPTH - 1 submits later spx_pipeline_wait -- and
 * every stream of every batch must come back identical; the first one is written to OUT.raw.  Even batches are produced in the
 * pipeline's own pinned staging buffer (spx_pipeline_host_input), odd ones come from a buffer of the caller's (spx_host_alloc).
 * Exit code 0 = ok.  Used by tests/test_gpu_cli.py::test_c_pipeline_example. */
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

int main(int argc, char** argv) {
  if (argc != 10) {
    fprintf(stderr, "usage: %s IN.raw RATE CHANNELS SPEED NONLINEAR COPIES BATCHES DEPTH OUT.raw\n", argv[0]);
    return 1;
  }
  const int rate = atoi(argv[2]), channels = atoi(argv[3]), copies = atoi(argv[6]), batches = atoi(argv[7]), depth = atoi(argv[8]);
  const float speed = (float)atof(argv[4]), nonlinear = (float)atof(argv[5]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  int16_t* pcm = (int16_t*)malloc((size_t)bytes + 2);
  if (fread(pcm, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);
  const int64_t n_in = bytes / 2 / channels;
  const size_t per_stream = (size_t)n_in * channels;

  spx_plan_t plan = spx_plan_create(rate, /*match_matlab=*/0);
  if (!plan) { fprintf(stderr, "spx_plan_create: %s\n", spx_last_error()); return 2; }
  spx_stream_job* jobs = (spx_stream_job*)calloc((size_t)copies, sizeof(spx_stream_job));
  for (int i = 0; i < copies; i++) {
    jobs[i].in_off = (int64_t)i * (int64_t)per_stream;   /* where stream i starts in a batch's input; the outputs are the pipeline's business */
    jobs[i].n_in = n_in;
    jobs[i].channels = channels;
    jobs[i].speed = speed;
    jobs[i].nonlinear = nonlinear;
    jobs[i].feedback = 0.0f;
  }
  spx_pipeline_t pipe = spx_pipeline_create(plan, jobs, copies, depth, /*flags=*/0);
  if (!pipe) { fprintf(stderr, "spx_pipeline_create: %s\n", spx_last_error()); return 2; }
  const int d = spx_pipeline_depth(pipe);
  const size_t in_values = spx_pipeline_input_values(pipe);
  if (in_values != per_stream * (size_t)copies) { fprintf(stderr, "unexpected input size\n"); return 3; }
  int16_t* mine = (int16_t*)spx_host_alloc(in_values * sizeof(int16_t));   /* pinned: the copy runs at the link's rate */
  if (!mine) { fprintf(stderr, "spx_host_alloc: %s\n", spx_last_error()); return 2; }
  for (int i = 0; i < copies; i++) memcpy(mine + (size_t)i * per_stream, pcm, per_stream * sizeof(int16_t));

  int16_t* first = NULL;
  int64_t first_n = -1;
  int64_t* tickets = (int64_t*)malloc((size_t)batches * sizeof(int64_t));
  const double t0 = now_ms();
  for (int k = 0; k < batches + d - 1; k++) {
    if (k < batches) {
      const int16_t* in = mine;
      if (k % 2 == 0) {   /* produce this batch in the pipeline's staging buffer */
        int16_t* stage = spx_pipeline_host_input(pipe);
        if (!stage) { fprintf(stderr, "spx_pipeline_host_input: %s\n", spx_last_error()); return 2; }
        for (int i = 0; i < copies; i++) memcpy(stage + (size_t)i * per_stream, pcm, per_stream * sizeof(int16_t));
        in = stage;
      }
      tickets[k] = spx_pipeline_submit(pipe, in, /*in_is_device=*/0);
      if (tickets[k] < 0) { fprintf(stderr, "spx_pipeline_submit: %s\n", spx_last_error()); return 2; }
    }
    const int w = k - (d - 1);   /* the batch submitted d - 1 submits ago: up to d batches are in flight */
    if (w >= 0) {
      const int16_t* out;
      const int64_t *offsets, *counts;
      if (spx_pipeline_wait(pipe, tickets[w], &out, &offsets, &counts) != 0) { fprintf(stderr, "spx_pipeline_wait: %s\n", spx_last_error()); return 2; }
      for (int i = 0; i < copies; i++) {
        if (counts[i] < 0) { fprintf(stderr, "batch %d stream %d: output capacity exceeded\n", w, i); return 3; }
        if (first_n < 0) {
          first_n = counts[i];
          first = (int16_t*)malloc((size_t)first_n * channels * sizeof(int16_t) + 2);
          memcpy(first, out + offsets[i], (size_t)first_n * channels * sizeof(int16_t));
        }
        if (offsets[i] % 32 != 0 || counts[i] != first_n || memcmp(out + offsets[i], first, (size_t)first_n * channels * sizeof(int16_t)) != 0) {
          fprintf(stderr, "batch %d stream %d differs from the first stream of the first batch\n", w, i);
          return 3;
        }
      }
    }
  }
  const double ms = (now_ms() - t0) / batches;
  f = fopen(argv[9], "wb");
  if (!f) { perror(argv[9]); return 1; }
  fwrite(first, sizeof(int16_t), (size_t)first_n * channels, f);
  fclose(f);
  printf("%d batches of %d streams x %lld frames -> %lld frames out each; %.3f ms per batch, host memory to host memory (%d buffer sets)\n",
         batches, copies, (long long)n_in, (long long)first_n, ms, d);
  spx_pipeline_destroy(pipe);
  spx_host_free(mine);
  spx_plan_destroy(plan);
  free(jobs); free(pcm); free(first); free(tickets);
  return 0;
}
