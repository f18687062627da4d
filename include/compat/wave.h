/* wave.h (compat) — the WAV file helpers the reference's CLI and tests take from libsonic (speedy_wave.cc:27,
 * 162-233; sonic_test.cc:37,266-285; speedy_test.cc:181-188).  libsonic's wave.c is not part of the reference tree;
 * libspeedy_hip.so exports these five functions (speedy_amd/csrc/wave_compat.cpp: RIFF / PCM 16-bit only, host code).
 * Sample counts are multi-channel frames; buffers hold count * numChannels interleaved shorts. */
#ifndef SPEEDY_HIP_COMPAT_WAVE_H_
#define SPEEDY_HIP_COMPAT_WAVE_H_

#ifdef __cplusplus
extern "C" {
#endif

struct waveFileStruct;
typedef struct waveFileStruct* waveFile;

/* NULL when the file cannot be opened or is not PCM16 RIFF/WAVE. */
waveFile openInputWaveFile(const char* fileName, int* sampleRate, int* numChannels);
waveFile openOutputWaveFile(const char* fileName, int sampleRate, int numChannels);
/* Writes the sizes into the header of an output file.  1 on success, 0 on failure. */
int closeWaveFile(waveFile file);
/* Frames read (0 at the end of the data chunk). */
int readFromWaveFile(waveFile file, short* buffer, int maxSamples);
/* 1 on success, 0 on failure. */
int writeToWaveFile(waveFile file, short* buffer, int numSamples);

#ifdef __cplusplus
}
#endif
#endif /* SPEEDY_HIP_COMPAT_WAVE_H_ */
