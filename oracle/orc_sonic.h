/* ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.
 *
 * CPU restatement of the time-scale-modification library the reference calls through
 * `sonicInt*` (call sites: reference soniclib.c:94,98,106,144-145,174,182,354,369,398,464,521,526,
 * 547,551; sonic_test.cc:370,735-750).
 *
 * PARITY UNPINNED at bit level: the arithmetic lives in the third-party dependency
 * github.com/waywardgeek/sonic (sonic.c / sonic.h), which the reference Makefile clones un-pinned
 * (`SONIC_DIR=../sonic`, Makefile:7,74: `git clone --recursive https://github.com/waywardgeek/sonic.git`,
 * no commit, no tag) and which is NOT present under /root/reference.  This file restates that library's
 * published PICOLA/AMDF algorithm (the long-standing "classic" revision: remainingInputToCopy
 * bookkeeping, speed applied per processing pass).  It is anchored on what the
 * reference itself requires of the dependency: the API shape at the call sites above and the property
 * tests of sonic_classic_test.cc / sonic_test.cc (lengths, Teager purity, mono==stereo bit identity),
 * restated in tests/test_oracle_sonic_properties.py.  No golden int16 vectors exist in the reference.
 *
 * Scope: speed != 1 (skip / insert pitch periods) and rate != 1 (the classic revision's linear-interpolation
 * resampler behind sonicIntSetRate; newer revisions use a windowed sinc -- unpinned like everything here);
 * volume = pitch = 1.
 */
#ifndef ORC_SONIC_H_
#define ORC_SONIC_H_
#ifdef __cplusplus
extern "C" {
#endif

struct orc_sonic;
typedef struct orc_sonic* orc_sonicStream;

orc_sonicStream orc_sonicIntCreateStream(int sampleRate, int numChannels);
void orc_sonicIntDestroyStream(orc_sonicStream s);
void orc_sonicIntSetUserData(orc_sonicStream s, void* p);
void* orc_sonicIntGetUserData(orc_sonicStream s);
int orc_sonicIntGetNumChannels(orc_sonicStream s);
int orc_sonicIntGetSampleRate(orc_sonicStream s);
void orc_sonicIntSetSpeed(orc_sonicStream s, float speed);
float orc_sonicIntGetSpeed(orc_sonicStream s);
void orc_sonicIntSetRate(orc_sonicStream s, float rate);
int orc_sonicIntWriteShortToStream(orc_sonicStream s, const short* samples, int numSamples);
int orc_sonicIntWriteFloatToStream(orc_sonicStream s, const float* samples, int numSamples);
int orc_sonicIntReadShortFromStream(orc_sonicStream s, short* samples, int maxSamples);
int orc_sonicIntReadFloatFromStream(orc_sonicStream s, float* samples, int maxSamples);
int orc_sonicIntFlushStream(orc_sonicStream s);
int orc_sonicIntSamplesAvailable(orc_sonicStream s);

/* Test taps: number of pitch steps taken so far and the period chosen at each (ring of last n). */
long orc_sonicIntStepCount(orc_sonicStream s);
/* Record every chosen period into a caller buffer (NULL disables). */
void orc_sonicIntSetPeriodLog(orc_sonicStream s, int* log, long capacity);

#ifdef __cplusplus
}
#endif
#endif
