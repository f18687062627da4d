/* Test infrastructure: log spec v2 (orc_log_v2_f32, orc_speedy.c) over EVERY positive normal float -- 2 130 706 432 arguments --
 *   (1) against glibc's log (< 1 ulp from the true value itself): the histogram of |v2 - libm| in ulps; the bar is <= 1 ulp;
 *   (2) a checksum per block of 2^20 float patterns (sum of the results' bit patterns mod 2^64), which the GPU's
 *       spx_debug_log_check reproduces block for block (tests/test_gpu_parity.py): GPU == oracle on the whole domain.
 * Threads: POSIX, one block at a time per thread.
 *   orc_logcheck [threads]            -> one JSON line: counts per ulp distance, worst argument
 *   (library use: orc_logcheck_run)                                                                                         */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

double orc_log_v2_f32(float x);

#define BLOCK_BITS 20
#define FIRST_BLOCK (0x00800000u >> BLOCK_BITS) /* 2^-126: the first normal float */
#define END_BLOCK (0x7f800000u >> BLOCK_BITS)   /* +inf */

typedef struct {
  uint32_t next, end;
  pthread_mutex_t mu;
  uint64_t* sums;      /* [END_BLOCK], or NULL */
  int against_libm;
  uint64_t hist[4];    /* |difference| = 0, 1, 2, >= 3 ulps */
  uint32_t worst_bits;
  uint64_t worst;
} job_t;

static uint64_t dbits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static void* worker(void* arg) {
  job_t* J = (job_t*)arg;
  uint64_t hist[4] = {0, 0, 0, 0}, worst = 0;
  uint32_t worst_bits = 0;
  for (;;) {
    pthread_mutex_lock(&J->mu);
    const uint32_t b = J->next < J->end ? J->next++ : UINT32_MAX;
    pthread_mutex_unlock(&J->mu);
    if (b == UINT32_MAX) break;
    uint64_t sum = 0;
    for (uint32_t i = 0; i < (1u << BLOCK_BITS); i++) {
      const uint32_t bits = (b << BLOCK_BITS) | i;
      float x;
      memcpy(&x, &bits, 4);
      const double v = orc_log_v2_f32(x);
      const uint64_t vb = dbits(v);
      sum += vb;
      if (J->against_libm) {
        const uint64_t lb = dbits(log((double)x));
        /* same sign unless one of them is a zero (x = 1: both +0): the patterns of same-signed doubles are ordered */
        uint64_t d = vb > lb ? vb - lb : lb - vb;
        if ((vb ^ lb) >> 63) d = (vb & 0x7fffffffffffffffull) + (lb & 0x7fffffffffffffffull);
        hist[d > 3 ? 3 : d]++;
        if (d > worst) { worst = d; worst_bits = bits; }
      }
    }
    if (J->sums) J->sums[b] = sum;
  }
  pthread_mutex_lock(&J->mu);
  for (int k = 0; k < 4; k++) J->hist[k] += hist[k];
  if (worst > J->worst) { J->worst = worst; J->worst_bits = worst_bits; }
  pthread_mutex_unlock(&J->mu);
  return NULL;
}

/* blocks [first, end) of 2^20 patterns each (FIRST_BLOCK = 8 .. END_BLOCK = 2040 covers every positive normal float);
 * sums: uint64[2040] indexed by block or NULL; hist4 / worst2 (worst distance, its argument's pattern): filled when against_libm */
int orc_logcheck_run(unsigned first, unsigned end, int threads, int against_libm, uint64_t* sums, uint64_t* hist4, uint64_t* worst2) {
  if (first < FIRST_BLOCK) first = FIRST_BLOCK;
  if (end > END_BLOCK) end = END_BLOCK;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  job_t J;
  memset(&J, 0, sizeof(J));
  J.next = first; J.end = end; J.sums = sums; J.against_libm = against_libm;
  pthread_mutex_init(&J.mu, NULL);
  pthread_t th[256];
  for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &J);
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&J.mu);
  if (hist4) memcpy(hist4, J.hist, sizeof(J.hist));
  if (worst2) { worst2[0] = J.worst; worst2[1] = J.worst_bits; }
  return 0;
}

#ifdef ORC_LOGCHECK_MAIN
int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 8;
  uint64_t hist[4], worst[2];
  orc_logcheck_run(FIRST_BLOCK, END_BLOCK, threads, 1, NULL, hist, worst);
  printf("{\"arguments\": %llu, \"ulps_from_glibc_log\": {\"0\": %llu, \"1\": %llu, \"2\": %llu, \"3+\": %llu}, \"worst_ulps\": %llu, "
         "\"worst_argument_bits\": \"0x%08llx\"}\n",
         (unsigned long long)(hist[0] + hist[1] + hist[2] + hist[3]), (unsigned long long)hist[0], (unsigned long long)hist[1],
         (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)worst[0], (unsigned long long)worst[1]);
  return hist[2] + hist[3] ? 1 : 0;
}
#endif
