/* INTEGRATION.md section 2 as a program: plain C99 over include/speedy_hip.h, no HIP headers, no C++.
 *
 *   batch_example IN.raw RATE CHANNELS SPEED NONLINEAR COPIES SPLIT OUT.raw
 *
 * IN.raw = interleaved int16 PCM.  The same utterance is submitted COPIES times as independent streams of one
 * spx_batch_run call (SPLIT = 0) or of spx_batch_analyze followed by spx_batch_walk (SPLIT = 1); every stream's output
 * must be identical, the first one is written to OUT.raw, gathered on the device with spx_batch_pack_outputs.
 * Exit code 0 = ok.  Used by tests/test_gpu_cli.py::test_c_batch_example. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "speedy_hip.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    if ((call) != 0) {                                                               \
      fprintf(stderr, "%s failed: %s\n", #call, spx_last_error());                   \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 9) {
    fprintf(stderr, "usage: %s IN.raw RATE CHANNELS SPEED NONLINEAR COPIES SPLIT OUT.raw\n", argv[0]);
    return 1;
  }
  const int rate = atoi(argv[2]), channels = atoi(argv[3]), copies = atoi(argv[6]), split = atoi(argv[7]);
  const float speed = (float)atof(argv[4]), nonlinear = (float)atof(argv[5]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  int16_t* host_in = (int16_t*)malloc((size_t)bytes + 2);
  if (fread(host_in, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);
  const int64_t n_in = bytes / 2 / channels;

  if (spx_abi_version() != 1) { fprintf(stderr, "unexpected ABI version %d\n", spx_abi_version()); return 1; }
  spx_plan_t plan = spx_plan_create(rate, /*match_matlab=*/0);
  if (!plan) { fprintf(stderr, "spx_plan_create: %s\n", spx_last_error()); return 2; }
  const int64_t cap = spx_plan_out_capacity_for(plan, n_in, speed, nonlinear);
  spx_stream_job* jobs = (spx_stream_job*)calloc((size_t)copies, sizeof(spx_stream_job));
  for (int i = 0; i < copies; i++) {
    jobs[i].in_off = (int64_t)i * n_in * channels;
    jobs[i].n_in = n_in;
    jobs[i].out_off = (int64_t)i * cap * channels;
    jobs[i].out_cap = cap;
    jobs[i].channels = channels;
    jobs[i].speed = speed;
    jobs[i].nonlinear = nonlinear;
    jobs[i].feedback = 0.0f;
  }
  const size_t in_bytes = (size_t)copies * (size_t)n_in * channels * sizeof(int16_t);
  const size_t out_bytes = (size_t)copies * (size_t)cap * channels * sizeof(int16_t);
  const size_t wsb = spx_batch_workspace_bytes(plan, jobs, copies);
  void* ws = spx_device_alloc(wsb);
  int16_t* d_in = (int16_t*)spx_device_alloc(in_bytes + 128);
  int16_t* d_out = (int16_t*)spx_device_alloc(out_bytes);
  int16_t* d_packed = (int16_t*)spx_device_alloc(out_bytes);
  int64_t* d_nout = (int64_t*)spx_device_alloc((size_t)copies * sizeof(int64_t));
  int64_t* d_offsets = (int64_t*)spx_device_alloc((size_t)(copies + 1) * sizeof(int64_t));
  if (!ws || !d_in || !d_out || !d_packed || !d_nout || !d_offsets) { fprintf(stderr, "device allocation failed\n"); return 2; }
  for (int i = 0; i < copies; i++)
    CHECK(spx_copy_to_device(d_in + (size_t)i * n_in * channels, host_in, (size_t)n_in * channels * sizeof(int16_t), NULL));
  if (split) {
    CHECK(spx_batch_analyze(plan, jobs, copies, d_in, ws, wsb, NULL, NULL));
    CHECK(spx_batch_walk(plan, jobs, copies, d_in, d_out, d_nout, ws, wsb, NULL, NULL));
  } else {
    CHECK(spx_batch_run(plan, jobs, copies, d_in, d_out, d_nout, ws, wsb, NULL, NULL));
  }
  CHECK(spx_batch_pack_outputs(jobs, copies, d_out, d_nout, d_packed, d_offsets, NULL));
  int64_t* host_nout = (int64_t*)malloc((size_t)copies * sizeof(int64_t));
  int64_t* host_off = (int64_t*)malloc((size_t)(copies + 1) * sizeof(int64_t));
  CHECK(spx_copy_to_host(host_nout, d_nout, (size_t)copies * sizeof(int64_t), NULL));
  CHECK(spx_copy_to_host(host_off, d_offsets, (size_t)(copies + 1) * sizeof(int64_t), NULL));
  CHECK(spx_stream_synchronize(NULL));
  int16_t* host_packed = (int16_t*)malloc((size_t)host_off[copies] * sizeof(int16_t) + 2);
  CHECK(spx_copy_to_host(host_packed, d_packed, (size_t)host_off[copies] * sizeof(int16_t), NULL));
  CHECK(spx_stream_synchronize(NULL));
  for (int i = 0; i < copies; i++) {
    if (host_nout[i] < 0) { fprintf(stderr, "stream %d: output capacity exceeded\n", i); return 3; }
    if (host_off[i + 1] - host_off[i] != host_nout[i] * channels) { fprintf(stderr, "stream %d: offsets disagree with n_out\n", i); return 3; }
    if (host_nout[i] != host_nout[0] ||
        memcmp(host_packed + host_off[i], host_packed, (size_t)host_nout[0] * channels * sizeof(int16_t)) != 0) {
      fprintf(stderr, "stream %d differs from stream 0\n", i);
      return 3;
    }
  }
  f = fopen(argv[8], "wb");
  if (!f) { perror(argv[8]); return 1; }
  fwrite(host_packed, sizeof(int16_t), (size_t)host_nout[0] * channels, f);
  fclose(f);
  printf("%d streams x %lld frames in -> %lld frames out each\n", copies, (long long)n_in, (long long)host_nout[0]);
  spx_device_free(ws); spx_device_free(d_in); spx_device_free(d_out); spx_device_free(d_packed);
  spx_device_free(d_nout); spx_device_free(d_offsets);
  spx_plan_destroy(plan);
  free(jobs); free(host_in); free(host_nout); free(host_off); free(host_packed);
  return 0;
}
