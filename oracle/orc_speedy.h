/* ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * CPU restatement of the reference's Speedy analysis (reference speedy.c / speedy.h).
 * The function set mirrors speedy.h:61-133 one-to-one (prefix orc_), with the compile-time
 * MATCH_MATLAB switch (speedy.h:136-146) turned into a stream-creation argument.
 *
 * Parity status: PINNED for the analysis path by the reference's own known-answer tests
 * (speedy_test.cc:135-530, restated in tests/test_oracle_kat.py) and by its Matlab fixtures
 * (speedy_test.cc:859-1057, restated in tests/test_oracle_matlab_fixture.py).
 * The FFT library the reference links (FFTW3 / kissfft, speedy.c:39-43) is NOT in the image, so the
 * DFT is this repo's own double-precision mixed-radix Stockham transform (DESIGN.md "DFT spec").
 */
#ifndef ORC_SPEEDY_H_
#define ORC_SPEEDY_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FEATURE_COUNT 15 /* speedy.h:115 */

struct orc_speedy; /* speedy.c:130-176 */
typedef struct orc_speedy* orc_speedyStream;

/* speedy.c:50-88 */
typedef struct {
  float state;
  float alpha;
} orc_fof;
void orc_fof_design(orc_fof* f, float time_constant_in_samples);
float orc_fof_iterate(orc_fof* f, float input);
void orc_fof_reset(orc_fof* f);
void orc_fof_set_state(orc_fof* f, float s);

/* speedy.c:206-299; match_matlab selects (future,past) = (8,12) instead of (12,8). */
orc_speedyStream orc_speedyCreateStream(int sample_rate, int match_matlab);
void orc_speedyDestroyStream(orc_speedyStream s);
int orc_speedyInputFrameSize(orc_speedyStream s);
int orc_speedyInputFrameStep(orc_speedyStream s);
int orc_speedyFFTSize(orc_speedyStream s);
int orc_speedyHysteresisFuture(orc_speedyStream s);
int orc_speedyHysteresisPast(orc_speedyStream s);
float orc_speedyBinToFreq(orc_speedyStream s, int bin);
int orc_speedyFreqToBin(orc_speedyStream s, float freq);

void orc_speedyAddData(orc_speedyStream s, const float* input, int64_t at_time);
void orc_speedyAddDataShort(orc_speedyStream s, const int16_t* input, int64_t at_time);
int orc_speedyComputeTension(orc_speedyStream s, int64_t at_time, float* tension);
float orc_speedyComputeSpeedFromTension(float tension, float R_g, float duration_feedback_strength,
                                        orc_speedyStream s);
int64_t orc_speedyGetCurrentTime(orc_speedyStream s);

float* orc_speedySpectrogram(orc_speedyStream s, float* input);
float orc_speedyEvaluateHysteresis(orc_speedyStream s, int64_t at_time);
void orc_speedyAddToHysteresisBuffer(orc_speedyStream s, float value, int64_t at_time);
void orc_speedyComputeSpectralDifference(orc_speedyStream s, const float* spectrogram,
                                         const float* last_spectrogram, int64_t at_time);
void orc_speedyComputeLocalEnergy(orc_speedyStream s, float* spectrogram, int64_t at_time);
void orc_speedySaveSpectrogramData(orc_speedyStream s, float* spectrogram, int64_t at_time);
float* orc_speedyGetSpectrogramAtTime(orc_speedyStream s, int64_t at_time);
void orc_speedyPreemphasisFilter(orc_speedyStream s, float* input, int length);
float* orc_speedyGetNormalizedSpectrogram(orc_speedyStream s);
float* orc_speedyGetSpectrogram(orc_speedyStream s);
float* orc_speedyGetInternalState(orc_speedyStream s);
float orc_speedyGetEnergyCompressed(orc_speedyStream s);
float orc_speedyGetSpeechChanges(orc_speedyStream s);
float orc_speedyNormalizeByEnergy(const float* spectrogram, float* normalized, int length);

/* ---- DFT building blocks (exposed so tests can check them against a naive DFT) ---- */
/* Natural log with a fixed, libm-independent operation sequence (DESIGN.md "log spec"). */
double orc_log(double x);          /* log spec v1: the fdlibm sequence, any double */
double orc_log_v2_f32(float x);    /* log spec v2: a positive normal float */
double orc_log_spec(double x);     /* what the analysis calls: v2 for positive normal floats, v1 otherwise */
void orc_set_log_spec(int v);      /* 1: v1 for every argument (A/B against round 1-4's spec); 2: the default */
int orc_get_log_spec(void);
void orc_set_dft_spec(int v);      /* 1: the unfused transform of rounds 1-4; 2 (default): multiply-add pairs fused */
int orc_get_dft_spec(void);
/* the twiddle tables and the Hamming window's cosine: 3 (default, round 6): orc_twiddle.h -- IEEE double operations only, the same bits
 * on every machine; 2: the box's libm (one sincos call per entry), as rounds 1-5.  Read when a stream / plan is created. */
/* the reference's shipped FFT (FFTW, double): used for a second CPU baseline when the box has libfftw3 (dlopen at run time) */
int orc_fftw_available(void);
int orc_set_fft_backend(int v);   /* 0: the port's transform (default); 1: FFTW if available.  Returns the backend in force */
void orc_set_twiddle_spec(int v);
int orc_get_twiddle_spec(void);
void orc_twiddle_entry(long k, long n, double* c, double* s);   /* cos, sin (2 pi k / n) */
unsigned long long orc_twiddle_hash(long den, long count);      /* FNV-1a of the table (cos, -sin)(2 pi t / den), t < count */
/* Forward complex DFT of length n (any n >= 1): in/out are interleaved re,im doubles. */
void orc_dft_forward(int n, const double* in, double* out);
/* O(n^2) definition, long-double accumulation; the checker for orc_dft_forward. */
void orc_dft_naive(int n, const double* in, double* out);
/* |DFT_{2W}(x zero-padded)| for real float x[W], all 2W bins, via the packed W-point transform. */
void orc_spectrum_magnitudes(int W, const float* x, float* mags);

#ifdef __cplusplus
}
#endif
#endif
