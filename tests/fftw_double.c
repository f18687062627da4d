/* TEST DOUBLE, not FFTW: three functions with libfftw3's names over a naive O(N^2) double-precision DFT, built by
 * tests/test_oracle_fftw_backend.py into tests/_fftw_double/libfftw3.so.3 so that the oracle's run-time FFTW probe
 * (oracle/orc_speedy.c orc_fftw_available / orc_set_fft_backend, bench.py cpu_baseline_fftw) can be exercised on an image that
 * has no libfftw3.  It pins nothing about the reference and is never used as a baseline: bench.py on a box without the real
 * library reports {"fftw": "absent"}. */
#include <math.h>
#include <stdlib.h>
typedef struct { int n, sign; double* in; double* out; } plan_t;
void* fftw_plan_dft_1d(int n, void* in, void* out, int sign, unsigned flags) {
  (void)flags;
  plan_t* p = (plan_t*)malloc(sizeof(plan_t));
  p->n = n; p->sign = sign; p->in = (double*)in; p->out = (double*)out;
  return p;
}
void fftw_execute(void* pl) {
  plan_t* p = (plan_t*)pl;
  const int n = p->n;
  for (int k = 0; k < n; k++) {
    long double re = 0, im = 0;
    for (int t = 0; t < n; t++) {
      const long double a = (long double)p->sign * 2.0L * 3.14159265358979323846264338327950288L * (long double)(((long)k * t) % n) / n;
      const long double c = cosl(a), s = sinl(a);
      re += p->in[2 * t] * c - p->in[2 * t + 1] * s;
      im += p->in[2 * t] * s + p->in[2 * t + 1] * c;
    }
    p->out[2 * k] = (double)re; p->out[2 * k + 1] = (double)im;
  }
}
void fftw_destroy_plan(void* pl) { free(pl); }
