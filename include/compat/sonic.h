/* sonic.h (compat) — the part of libsonic's public header that the reference's callers see.
 *
 * The reference's sonic2.h:34-35 does `#define SONIC_INTERNAL 1; #include "sonic.h"`: it expects the third-party
 * waywardgeek/sonic header to declare `sonicStream` and, under SONIC_INTERNAL, to rename every libsonic function to
 * sonicInt* so that the shim can re-declare the public names (sonic2.h:22-50).  The reference's CLI and tests include
 * it directly as well (speedy_wave.cc:24, sonic_test.cc:37).  libsonic is not part of the reference tree (Makefile:7,74);
 * in this repo the time-scale-modification stage lives in the HIP kernels and libspeedy_hip.so exports the sonicInt*
 * entry points the shim and the tests use (include/sonic2.h "sonicInt*"), so this header is all a reference caller needs
 * from libsonic:  g++ -Iinclude/compat -Iinclude  its_own_source.cc  -lspeedy_hip.
 *
 * Written fresh from libsonic's published API (function names and argument meaning only); functions libspeedy_hip.so
 * does not provide (pitch, volume, chord pitch, quality, unsigned-char I/O, sonicChange*Speed) are not declared, so a
 * caller that needs them fails at compile time, not at run time.
 */
#ifndef SPEEDY_HIP_COMPAT_SONIC_H_
#define SPEEDY_HIP_COMPAT_SONIC_H_

#ifdef __cplusplus
extern "C" {
#endif

#ifdef SONIC_INTERNAL
/* the library's own names are the sonicInt* ones; the shim header re-declares the public names (sonic2.h:37-50) */
#define sonicCreateStream sonicIntCreateStream
#define sonicDestroyStream sonicIntDestroyStream
#define sonicSetUserData sonicIntSetUserData
#define sonicGetUserData sonicIntGetUserData
#define sonicWriteFloatToStream sonicIntWriteFloatToStream
#define sonicWriteShortToStream sonicIntWriteShortToStream
#define sonicReadFloatFromStream sonicIntReadFloatFromStream
#define sonicReadShortFromStream sonicIntReadShortFromStream
#define sonicFlushStream sonicIntFlushStream
#define sonicSamplesAvailable sonicIntSamplesAvailable
#define sonicGetSpeed sonicIntGetSpeed
#define sonicSetSpeed sonicIntSetSpeed
#define sonicSetRate sonicIntSetRate
#define sonicGetSampleRate sonicIntGetSampleRate
#define sonicGetNumChannels sonicIntGetNumChannels
#endif

#ifndef SPEEDY_HIP_SONICSTREAM_DECLARED
#define SPEEDY_HIP_SONICSTREAM_DECLARED
struct sonicStreamStruct;
typedef struct sonicStreamStruct* sonicStream;
#endif

/* Sample counts are multi-channel frames; float samples lie in (-1, 1). */
sonicStream sonicCreateStream(int sampleRate, int numChannels);
void sonicDestroyStream(sonicStream stream);
void sonicSetUserData(sonicStream stream, void* userData);
void* sonicGetUserData(sonicStream stream);
int sonicWriteFloatToStream(sonicStream stream, const float* samples, int numSamples);
int sonicWriteShortToStream(sonicStream stream, const short* samples, int numSamples);
int sonicReadFloatFromStream(sonicStream stream, float* samples, int maxSamples);
int sonicReadShortFromStream(sonicStream stream, short* samples, int maxSamples);
int sonicFlushStream(sonicStream stream);
int sonicSamplesAvailable(sonicStream stream);
float sonicGetSpeed(sonicStream stream);
void sonicSetSpeed(sonicStream stream, float speed);
void sonicSetRate(sonicStream stream, float rate);
int sonicGetSampleRate(sonicStream stream);
int sonicGetNumChannels(sonicStream stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEEDY_HIP_COMPAT_SONIC_H_ */
