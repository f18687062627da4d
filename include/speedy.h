/* speedy.h — unit-level API of the MI355X-native Speedy analysis, the secondary boundary of SURVEY.md section 8b.
 *
 * Same names, argument meaning and return conventions as the reference's speedy.h:61-100 (implemented there by
 * speedy.c), for the functions a caller drives the analysis with one frame at a time -- what the reference's
 * speedy_test.cc does (:197-254, :331-373, :457-530, :534-757, :859-1057) and what its shim does underneath
 * (soniclib.c:288-345).  Every frame goes through the same HIP kernels as the batch path (spx_analysis_kernel fed with
 * explicit float frames, spx_tension_kernel with an explicit tension range); nothing is computed on the host.
 *
 * The hysteresis shape (-DMATCH_MATLAB in the reference, speedy.h:136-146) is the process-wide default of
 * speedyHipSetMatchMatlab (include/sonic2.h) read at speedyCreateStream.
 *
 * Call pattern (as in the reference's tests and shim): speedyAddData with at_time = t0, t0+1, t0+2, ... (t0 = 0 or 1),
 * speedyComputeTension with increasing at_time, each at most once; times may be left out (the shim does after a flush,
 * soniclib.c:538-550) and are then never computed -- as in the reference, the first call that succeeds is the one treated
 * as a low-energy frame (speedy.c:293,691).  A time asked again, out of order, or 20 or more frames behind the newest
 * frame is answered as the reference answers it: from what its 21-entry spectrum ring and 42-entry hysteresis ring hold at
 * that moment (speedy.c:198-200,484-487,594-608 -- its own tests do this, speedy_test.cc:564,628).  Anything else returns 0 /
 * is ignored with a message in speedyHipLastError().
 *
 * The reference's test hooks between stages (speedy.h:102-133) are provided too, see below: with the stages fused on the
 * device there is no host-visible hand-off to hook, so each is a small kernel of its own on the shared device state. */
#ifndef SPEEDY_HIP_SPEEDY_H_
#define SPEEDY_HIP_SPEEDY_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

struct speedyStreamStruct;
typedef struct speedyStreamStruct* speedyStream;

speedyStream speedyCreateStream(int sample_rate);                            /* speedy.c:206-299 */
void speedyDestroyStream(speedyStream stream);                               /* speedy.c:302-328 */
int speedyInputFrameSize(speedyStream stream);                               /* speedy.c:330-333, samples */
int speedyInputFrameStep(speedyStream stream);                               /* speedy.c:335-338, samples */
int speedyFFTSize(speedyStream stream);                                      /* speedy.c:340-343 */
float speedyBinToFreq(speedyStream stream, int bin_number);                  /* speedy.c:345-348 */
int speedyFreqToBin(speedyStream stream, float freq);                        /* speedy.c:350-353 */

/* One analysis frame of speedyInputFrameSize() samples (speedy.c:553-565 / its float twin): pre-emphasis with the state
 * carried from the previous call, Hamming window, |DFT|, frame energy, energy low-pass, compression. */
void speedyAddData(speedyStream stream, const float input[], int64_t at_time);
void speedyAddDataShort(speedyStream stream, const int16_t input[], int64_t at_time);
/* 1 and *tension when at_time + kTemporalHysteresisFuture <= current time, else 0 (speedy.c:752-766). */
int speedyComputeTension(speedyStream stream, int64_t at_time, float* tension);
float speedyComputeSpeedFromTension(float tension, float R_g, float duration_feedback_strength,
                                    speedyStream stream);                    /* speedy.c:768-788 */
int64_t speedyGetCurrentTime(speedyStream stream);                           /* speedy.c:736-739 */

/* |DFT_N| of one already pre-emphasised frame, no state touched (speedy.c:438-473); library-owned array of N floats. */
float* speedySpectrogram(speedyStream stream, float input[]);
/* Library-owned arrays, valid until the next call on the stream: the spectrum of the last frame added (N floats), of
 * the frame added at `at_time` (the last 21 are kept, speedy.c:97,476-487), the normalised spectrum the last tension
 * used (N/2 computed bins, the rest zero), and the 15 internal-state values of speedy.c:106-124 as of the last tension
 * (their AddData-time entries -- energy low-pass, local, compressed, time -- are those of the frame added at
 * at_time + kTemporalHysteresisFuture, i.e. the latest frame when tensions are asked for as soon as they are ready). */
float* speedyGetSpectrogram(speedyStream stream);
float* speedyGetSpectrogramAtTime(speedyStream stream, int64_t at_time);
float* speedyGetNormalizedSpectrogram(speedyStream stream);
#define kFeatureValueCount 15
float* speedyGetInternalState(speedyStream stream);
float speedyGetEnergyCompressed(speedyStream stream);                        /* of the last frame added */
float speedyGetSpeechChanges(speedyStream stream);                           /* of the last tension */
/* ---- the reference's hooks between stages (speedy.h:102-133), which its unit tests drive directly (speedy_test.cc:135-453).
 * The stages are fused on the device; each hook is a small kernel of its own on the same device state the fused kernels
 * use (filter states, hysteresis values, spectra), with the reference's arithmetic.  Times follow speedyAddData's. ---- */
void speedyComputeSpectralDifference(speedyStream stream, const float* spectrogram, const float* last_spectrogram,
                                     int64_t at_time);                       /* speedy.c:664-729; fft_size/2 bins of each are read */
void speedyComputeLocalEnergy(speedyStream stream, float* spectrogram, int64_t at_time);  /* speedy.c:510-523: the LAST frame's
                                                                                 spectrum, whatever is passed (speedy.c:515) */
void speedySaveSpectrogramData(speedyStream stream, float spectrogram[], int64_t at_time);   /* speedy.c:476-483 */
void speedyPreemphasisFilter(speedyStream stream, float* input, int length);  /* speedy.c:416-425, in place, state carried */
float speedyEvaluateHysteresis(speedyStream stream, int64_t at_time);         /* speedy.c:590-610 */
void speedyAddToHysteresisBuffer(speedyStream stream, float value, int64_t at_time);         /* speedy.c:615-619 */
float* speedyGetInternalSpectrogram(speedyStream stream);                     /* = speedyGetSpectrogram (speedy.c:393-396) */
float* speedyGetInternalNormalizedSpectrogram(speedyStream stream);           /* = speedyGetNormalizedSpectrogram (:398-401) */
float speedyNormalizeByEnergy(const float* spectrogram, float* normalized, int length);      /* speedy.c:628-647 */

/* A first-order low-pass filter (speedy.c:50-88); its state lives on the device, each call is one tiny launch. */
struct FirstOrderFilterStruct;
typedef struct FirstOrderFilterStruct* FirstOrderFilter;
FirstOrderFilter CreateFirstOrderFilter(float time_constant_in_samples);
void DesignFirstOrderLowpassFilter(FirstOrderFilter fof, float time_constant_in_samples);
float IterateFirstOrderFilter(FirstOrderFilter fof, float input);
void ResetFirstOrderFilter(FirstOrderFilter fof);
void DeleteFirstOrderFilter(FirstOrderFilter fof);

/* kTemporalHysteresisFuture / Past of this stream (compile-time constants in the reference, speedy.h:136-146). */
int speedyHipHysteresisFuture(speedyStream stream);
int speedyHipHysteresisPast(speedyStream stream);

/* The reference's compile-time constants, for callers that use them as such (speedy_test.cc:756 compares the measured
 * latency with kTemporalHysteresisFuture).  Here the shape is chosen at RUN time -- speedyHipSetMatchMatlab /
 * speedyHipCreateSonicStream (include/sonic2.h) -- so a caller built with -DMATCH_MATLAB must also select it once at
 * start-up; the macros only restate which pair that build means (speedy.h:136-146). */
#ifndef kTemporalHysteresisFuture
#ifdef MATCH_MATLAB
#define kTemporalHysteresisFuture 8  /* frames */
#define kTemporalHysteresisPast 12   /* frames */
#else
#define kTemporalHysteresisFuture 12 /* frames */
#define kTemporalHysteresisPast 8    /* frames */
#endif
#endif

#ifdef __cplusplus
}
#endif
#endif /* SPEEDY_HIP_SPEEDY_H_ */
