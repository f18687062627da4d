/* sonic2.h — drop-in public API of the MI355X-native Speedy hot path.
 *
 * Same names, argument meaning, units and return conventions as the reference's public header
 * (google/speedy sonic2.h:54-125, implemented there by soniclib.c); a program written against the
 * reference links against libspeedy_hip.so unchanged.  Written fresh for this repo: the reference
 * header pulls in the third-party libsonic's sonic.h (sonic2.h:34-35), which is not needed here because
 * the TSM stage lives in the HIP kernels.
 *
 * Units: sampleCount / bufferSize / returned counts are MULTI-CHANNEL sample frames; buffers hold
 * count*numChannels interleaved values (sonic2.h:56-59).  Float samples are in (-1,1) (sonic2.h:64).
 * Errors: NULL from create on failure, write returns 1 on success / 0 on failure, read returns the number
 * of frames delivered (0 = nothing ready), no exceptions cross this boundary.
 * A stream is not thread-safe; distinct streams are independent.
 *
 * Execution model: a write stages its samples; the kernels for the newly completed 10 ms frames of ALL
 * streams with staged work run as one launch sequence when a result is first asked for (see
 * speedyHipSetCoalescing below; streams with callbacks, a rate stage or mode switches run their own
 * sequence per write).  The sequence of output samples, and the count available after each write,
 * equal the reference's.
 */
#ifndef SPEEDY_HIP_SONIC2_H_
#define SPEEDY_HIP_SONIC2_H_

#ifdef __cplusplus
extern "C" {
#endif

#ifndef SPEEDY_HIP_SONICSTREAM_DECLARED   /* (include/compat/sonic.h declares it too) */
#define SPEEDY_HIP_SONICSTREAM_DECLARED
struct sonicStreamStruct;
typedef struct sonicStreamStruct* sonicStream;
#endif

sonicStream sonicCreateStream(int sampleRate, int numChannels);            /* soniclib.c:93-134  */
void sonicDestroyStream(sonicStream stream);                               /* soniclib.c:141-167 */
int sonicWriteShortToStream(sonicStream stream, const short* inBuffer, int sampleCount); /* :391-452 */
int sonicReadShortFromStream(sonicStream stream, short* outBuffer, int bufferSize);      /* :519-522 */
int sonicWriteFloatToStream(sonicStream stream, const float* inBuffer, int sampleCount); /* :457-517 */
int sonicReadFloatFromStream(sonicStream stream, float* outBuffer, int bufferSize);      /* :524-527 */
void sonicSetRate(sonicStream stream, float rate);                         /* soniclib.c:169-175 */
void sonicSetSpeed(sonicStream stream, float speed);                       /* soniclib.c:177-183 */
int sonicFlushStream(sonicStream stream);                                  /* soniclib.c:529-552 */

/* 0 = purely linear speed-up (default), 1 = the standard Speedy nonlinear speed-up (soniclib.c:555-562).
 * Two-argument form, as implemented by the reference. */
void sonicEnableNonlinearSpeedup(sonicStream stream, float nonlinearFactor);
void sonicSetDurationFeedbackStrength(sonicStream stream, float factor);   /* soniclib.c:565-571 */
int getSonicBufferSize(sonicStream stream);                                /* soniclib.c:672-680 */
int sonicSpectrogramSize(sonicStream stream);                              /* soniclib.c:661-669 */

/* Monitoring callbacks (sonic2.h:104-125); `time` is in internal buffer counts. */
typedef void (*tensionFunction)(sonicStream myStream, int time, float tension);
void sonicTensionCallback(sonicStream stream, tensionFunction f);
tensionFunction getSonicTensionCallback(sonicStream stream);
typedef void (*speedFunction)(sonicStream myStream, int time, float speed);
void sonicSpeedCallback(sonicStream stream, speedFunction f);
tensionFunction getSonicSpeedCallback(sonicStream stream);
typedef void (*featuresFunction)(sonicStream myStream, int time, float* features);
void sonicFeaturesCallback(sonicStream stream, featuresFunction f);
featuresFunction getSonicFeaturesCallback(sonicStream stream);
typedef void (*spectrogramFunction)(sonicStream myStream, int time, float* spectrogram);
void sonicSpectrogramCallback(sonicStream stream, spectrogramFunction f);
spectrogramFunction getSonicSpectrogramCallback(sonicStream stream);
void sonicNormalizedSpectrogramCallback(sonicStream stream, spectrogramFunction f);
spectrogramFunction getSonicNormalizedSpectrogramCallback(sonicStream stream);

/* The libsonic ("sonicInt*") entry points the reference reaches through sonic.h and its tests call directly
 * (sonic_test.cc:370,735-750; soniclib.c:94-182).  They address the TSM stage alone = a stream in linear mode. */
int sonicIntGetNumChannels(sonicStream stream);
int sonicIntGetSampleRate(sonicStream stream);
float sonicIntGetSpeed(sonicStream stream);
sonicStream sonicIntCreateStream(int sampleRate, int numChannels);
void sonicIntDestroyStream(sonicStream stream);
void sonicIntSetSpeed(sonicStream stream, float speed);
void sonicIntSetRate(sonicStream stream, float rate);
int sonicIntWriteShortToStream(sonicStream stream, const short* samples, int numSamples);
int sonicIntWriteFloatToStream(sonicStream stream, const float* samples, int numSamples);
int sonicIntReadShortFromStream(sonicStream stream, short* samples, int maxSamples);
int sonicIntReadFloatFromStream(sonicStream stream, float* samples, int maxSamples);
int sonicIntFlushStream(sonicStream stream);
int sonicIntSamplesAvailable(sonicStream stream);
void sonicIntSetUserData(sonicStream stream, void* userData);
void* sonicIntGetUserData(sonicStream stream);

/* ---- extensions (not in the reference) ---- */
/* The reference selects the temporal-hysteresis shape at COMPILE time (-DMATCH_MATLAB, speedy.h:136-146);
 * here it is a process-wide default read at sonicCreateStream.  0 (default) = the shipped library's
 * (future,past) = (12,8); 1 = the test builds' (8,12). */
void speedyHipSetMatchMatlab(int on);
/* sonicCreateStream with the hysteresis shape given explicitly (no process-wide state involved). */
sonicStream speedyHipCreateSonicStream(int sampleRate, int numChannels, int matchMatlab);
/* Coalesced execution (default on; SPX_NO_POOL=1 in the environment = off), for streams created afterwards.  A plain
 * stream -- no monitoring callbacks, rate 1, one mode, no sonicInt* calls -- only STAGES its writes and flushes; the
 * first call that needs a result on such a stream (a read, sonicSamplesAvailable, a setter ...) runs everything that is
 * staged on ANY stream of the device in one launch sequence.  Per-stream results are those of the reference call for
 * call; a server that writes to all its streams and then reads from all of them pays one launch sequence per round
 * instead of one per stream.  Streams of one device may be used from different threads (the pool is locked). */
void speedyHipSetCoalescing(int on);
int speedyHipGetCoalescing(void);
/* sonicCreateStream with the hysteresis shape AND the execution path chosen for this handle alone: coalesce = -1 the
 * process-wide default above, 0 = the handle runs its own launch sequence per write, 1 = coalesced.  Nothing process-wide
 * is read or written for an explicit 0 / 1, so handles of both kinds can be created from several threads at once. */
sonicStream speedyHipCreateSonicStreamEx(int sampleRate, int numChannels, int matchMatlab, int coalesce);
/* Launch sequences run / stream jobs served by the current device's pool so far. */
void speedyHipPoolStats(unsigned long long* runs, unsigned long long* jobs);
/* Frames currently readable without blocking on new input. */
int sonicSamplesAvailable(sonicStream stream);
/* Text of the last failure on this thread. */
const char* speedyHipLastError(void);

#ifdef __cplusplus
}
#endif
#endif /* SPEEDY_HIP_SONIC2_H_ */
