/* Random call sequences over many sonicStream handles through include/sonic2.h (plain C99): memory-safety runs of the
 * library's host code (make -C speedy_amd/csrc asan-host builds it with ASan + UBSan; tools/asan_host.sh runs this on
 * the GPU box) and a self-check that needs no oracle -- the SAME schedule is run twice, once with coalesced execution
 * (sonic2_pool.hip) and once with every handle on its own launch sequences, and every handle must deliver the same
 * bytes at every read.
 *
 *   api_fuzz [SEED=1] [HANDLES=24] [CALLS=4000]
 * Calls: writes of 1 .. 3000 frames (short and float), reads of 1 .. 4096, flushes, sonicSetSpeed / Rate /
 * EnableNonlinearSpeedup (also 0 <-> non-zero) / SetDurationFeedbackStrength, monitoring callbacks switched on mid-stream,
 * sonicInt* writes and flushes, sonicSamplesAvailable, destroy + re-create. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sonic2.h"

static uint64_t rng_state;
static unsigned rnd(void) {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (unsigned)(rng_state >> 33);
}
static unsigned rndn(unsigned n) { return rnd() % n; }

#define MAXH 64
static uint32_t crc_of[MAXH];
static long long frames_of[MAXH];
static int cb_count;
static void on_tension(sonicStream s, int t, float v) { (void)s; (void)t; (void)v; cb_count++; }

static uint32_t crc_update(uint32_t c, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  size_t i;
  int k;
  c = ~c;
  for (i = 0; i < n; i++) {
    c ^= b[i];
    for (k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
  }
  return ~c;
}

static int run(unsigned seed, int nh, int calls, int coalesce) {
  static short in[3000 * 3], out[4096 * 3];
  static float fin[3000 * 3], fout[4096 * 3];
  sonicStream h[MAXH];
  int ch[MAXH], rate[MAXH];
  int i, c;
  const int rates[4] = {16000, 16000, 22050, 8000};
  speedyHipSetCoalescing(coalesce);
  rng_state = 0x9E3779B97F4A7C15ull ^ seed;
  memset(crc_of, 0, sizeof(crc_of));
  memset(frames_of, 0, sizeof(frames_of));
  for (i = 0; i < nh; i++) {
    ch[i] = 1 + (int)(rndn(5) == 0) + (int)(rndn(11) == 0);
    rate[i] = rates[rndn(4)];
    h[i] = sonicCreateStream(rate[i], ch[i]);
    if (!h[i]) { fprintf(stderr, "create: %s\n", speedyHipLastError()); return 1; }
    sonicSetSpeed(h[i], 1.2f + 0.1f * (float)rndn(30));
    sonicEnableNonlinearSpeedup(h[i], rndn(4) ? 1.0f : 0.0f);
    sonicSetDurationFeedbackStrength(h[i], rndn(3) ? 0.0f : 0.1f);
  }
  for (c = 0; c < calls; c++) {
    const int k = (int)rndn((unsigned)nh);
    const unsigned op = rndn(100);
    int j, n, got;
    if (op < 45) {          /* write */
      double ph = 0.013 * (double)(k + 1);
      n = 1 + (int)(rndn(4) ? rndn(1200) : rndn(3000));
      for (j = 0; j < n * ch[k]; j++) {   /* a cheap voiced-looking signal: two sines of a per-handle pitch, slowly gated */
        const int t = (int)(frames_of[k] & 0x7fffffff) + j / ch[k];
        const int v = (int)(9000.0 * ((double)((t * (k + 3)) % 160) / 80.0 - 1.0)) + (int)(3000.0 * ((double)((t * 7) % 50) / 25.0 - 1.0));
        in[j] = (short)(((t / 4000) % 5 == 4) ? v / 64 : v);
        (void)ph;
      }
      frames_of[k] += n;
      if (rndn(8) == 0) {
        for (j = 0; j < n * ch[k]; j++) fin[j] = (float)in[j] / 32768.0f;
        if (sonicWriteFloatToStream(h[k], fin, n) != 1) { fprintf(stderr, "write float: %s\n", speedyHipLastError()); return 1; }
      } else if (sonicWriteShortToStream(h[k], in, n) != 1) { fprintf(stderr, "write: %s\n", speedyHipLastError()); return 1; }
    } else if (op < 75) {   /* read */
      n = 1 + (int)rndn(4096);
      if (rndn(10) == 0) {
        got = sonicReadFloatFromStream(h[k], fout, n);
        for (j = 0; j < got * ch[k]; j++) out[j] = (short)(fout[j] * 32767.0f);
      } else got = sonicReadShortFromStream(h[k], out, n);
      if (got < 0 || got > n) { fprintf(stderr, "read returned %d of %d\n", got, n); return 1; }
      crc_of[k] = crc_update(crc_of[k], out, sizeof(short) * (size_t)got * (size_t)ch[k]);
      crc_of[k] = crc_update(crc_of[k], &got, sizeof(got));
    } else if (op < 80) {
      if (sonicFlushStream(h[k]) != 1) { fprintf(stderr, "flush: %s\n", speedyHipLastError()); return 1; }
    } else if (op < 84) {
      sonicSetSpeed(h[k], rndn(5) ? 1.1f + 0.1f * (float)rndn(35) : 0.5f + 0.05f * (float)rndn(9));
    } else if (op < 86) {
      sonicEnableNonlinearSpeedup(h[k], rndn(3) ? 1.0f : (rndn(2) ? 0.0f : 0.5f));
    } else if (op < 88) {
      sonicSetDurationFeedbackStrength(h[k], rndn(2) ? 0.0f : 0.2f);
    } else if (op < 90) {
      got = sonicSamplesAvailable(h[k]);
      crc_of[k] = crc_update(crc_of[k], &got, sizeof(got));
    } else if (op < 92) {
      sonicSetRate(h[k], rndn(2) ? 1.0f : (rndn(2) ? 1.25f : 0.8f));
    } else if (op < 94) {
      sonicTensionCallback(h[k], rndn(3) ? on_tension : (tensionFunction)0);
    } else if (op < 96) {
      n = 1 + (int)rndn(900);
      for (j = 0; j < n * ch[k]; j++) in[j] = (short)((j * 37 + k * 101) % 7001 - 3500);
      if (sonicIntWriteShortToStream(h[k], in, n) != 1) { fprintf(stderr, "int write: %s\n", speedyHipLastError()); return 1; }
    } else if (op < 97) {
      sonicIntFlushStream(h[k]);
    } else if (op < 99) {   /* destroy with whatever is staged, start over */
      sonicDestroyStream(h[k]);
      h[k] = sonicCreateStream(rate[k], ch[k]);
      if (!h[k]) { fprintf(stderr, "re-create: %s\n", speedyHipLastError()); return 1; }
      sonicSetSpeed(h[k], 2.0f + 0.25f * (float)rndn(8));
      sonicEnableNonlinearSpeedup(h[k], 1.0f);
      crc_of[k] = crc_update(crc_of[k], "new", 3);
    } else {
      const float sp = sonicIntGetSpeed(h[k]);
      crc_of[k] = crc_update(crc_of[k], &sp, sizeof(sp));
    }
  }
  for (i = 0; i < nh; i++) {   /* drain everything */
    int got;
    sonicFlushStream(h[i]);
    while ((got = sonicReadShortFromStream(h[i], out, 4096)) > 0)
      crc_of[i] = crc_update(crc_of[i], out, sizeof(short) * (size_t)got * (size_t)ch[i]);
    sonicDestroyStream(h[i]);
  }
  return 0;
}

int main(int argc, char** argv) {
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
  int nh = argc > 2 ? atoi(argv[2]) : 24;
  const int calls = argc > 3 ? atoi(argv[3]) : 4000;
  uint32_t a[MAXH];
  int i, bad = 0;
  if (nh < 1) nh = 1;
  if (nh > MAXH) nh = MAXH;
  if (run(seed, nh, calls, 1)) return 1;
  memcpy(a, crc_of, sizeof(a));
  if (run(seed, nh, calls, 0)) return 1;
  for (i = 0; i < nh; i++)
    if (a[i] != crc_of[i]) { fprintf(stderr, "handle %d: coalesced %08x, own launch sequences %08x\n", i, a[i], crc_of[i]); bad++; }
  printf("api_fuzz seed %u: %d handles, %d calls, %d callbacks fired, %d handles differ between the two execution paths\n", seed, nh, calls,
         cb_count, bad);
  fflush(stdout);
  /* no exit handlers: under the sanitizer build the HSA runtime's own teardown trips a check inside the sanitizer's device
   * allocator now and then (after every stream is destroyed; nothing of this library is on that stack) */
  _Exit(bad ? 1 : 0);
}
