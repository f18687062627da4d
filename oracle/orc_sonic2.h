/* ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.
 *
 * CPU restatement of the reference's public shim API (reference sonic2.h:54-125, soniclib.c), i.e. the
 * drop-in boundary itself, on top of orc_speedy (analysis, pinned) and orc_sonic (TSM, parity unpinned).
 * Same functions, same argument meaning, prefix orc_; MATCH_MATLAB is a creation-time argument.
 * The library-side printf of soniclib.c:201,219 is not reproduced.
 */
#ifndef ORC_SONIC2_H_
#define ORC_SONIC2_H_
#include "orc_sonic.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef void (*orc_tensionFunction)(orc_sonicStream, int time, float tension);
typedef void (*orc_speedFunction)(orc_sonicStream, int time, float speed);
typedef void (*orc_featuresFunction)(orc_sonicStream, int time, float* features);
typedef void (*orc_spectrogramFunction)(orc_sonicStream, int time, float* spectrogram);
/* Oracle-only tap: every buffer handed to the TSM stage, with the speed in force (soniclib.c:354,369). */
typedef void (*orc_handoffFunction)(orc_sonicStream, int time, float speed, const short* buffer, int frames);

orc_sonicStream orc_sonicCreateStream(int sampleRate, int numChannels, int match_matlab); /* soniclib.c:93 */
void orc_sonicDestroyStream(orc_sonicStream s);                                             /* :141 */
int orc_sonicWriteShortToStream(orc_sonicStream s, const short* in, int sampleCount);       /* :391 */
int orc_sonicReadShortFromStream(orc_sonicStream s, short* out, int bufferSize);            /* :519 */
int orc_sonicWriteFloatToStream(orc_sonicStream s, const float* in, int sampleCount);       /* :457 */
int orc_sonicReadFloatFromStream(orc_sonicStream s, float* out, int bufferSize);            /* :524 */
void orc_sonicSetRate(orc_sonicStream s, float rate);                                       /* :169 */
void orc_sonicSetSpeed(orc_sonicStream s, float speed);                                     /* :177 */
int orc_sonicFlushStream(orc_sonicStream s);                                                /* :529 */
void orc_sonicEnableNonlinearSpeedup(orc_sonicStream s, float factor);                      /* :555 */
void orc_sonicSetDurationFeedbackStrength(orc_sonicStream s, float factor);                 /* :565 */
int orc_getSonicBufferSize(orc_sonicStream s);                                              /* :672 */
int orc_sonicSpectrogramSize(orc_sonicStream s);                                            /* :661 */
void orc_sonicTensionCallback(orc_sonicStream s, orc_tensionFunction f);                    /* :573 */
void orc_sonicSpeedCallback(orc_sonicStream s, orc_speedFunction f);                        /* :588 */
void orc_sonicFeaturesCallback(orc_sonicStream s, orc_featuresFunction f);                  /* :603 */
void orc_sonicSpectrogramCallback(orc_sonicStream s, orc_spectrogramFunction f);            /* :625 */
void orc_sonicNormalizedSpectrogramCallback(orc_sonicStream s, orc_spectrogramFunction f);  /* :645 */
void orc_sonicHandoffCallback(orc_sonicStream s, orc_handoffFunction f);

/* The caller loop of the reference's CLI (speedy_wave.cc:154-242, compress_sound) as one call:
 * create -> setSpeed -> enable(nonlinear) -> setFeedback -> {write <=chunk; read <=chunk} -> flush -> drain.
 * out must hold out_capacity frames; taps may be NULL.  tension/speed taps hold one float per frame
 * (capacity tap_capacity), features 15 floats per frame.  Returns frames produced, or -1 on overflow.
 * *n_taps receives the number of tension frames. */
long orc_compress_sound(const short* in, long n_in, int sampleRate, int numChannels, float speed,
                        float nonlinear, float feedback, int match_matlab, int chunk, short* out,
                        long out_capacity, float* tension_tap, float* speed_tap, float* features_tap,
                        long tap_capacity, long* n_taps);

#ifdef __cplusplus
}
#endif
#endif
