// include/compat/wave.h: the five WAV helpers the reference's callers expect from libsonic (speedy_wave.cc:162-233).
// Host code only; RIFF / PCM 16-bit, any chunk order, little-endian hosts.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/compat/wave.h"

struct waveFileStruct {
  FILE* f = nullptr;
  bool writing = false;
  int channels = 1;
  uint32_t data_left = 0;      // reading: bytes of the data chunk not delivered yet
  uint32_t data_written = 0;   // writing: bytes of samples so far
};

static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

extern "C" {

waveFile openInputWaveFile(const char* fileName, int* sampleRate, int* numChannels) {
  FILE* f = fopen(fileName, "rb");
  if (!f) return nullptr;
  unsigned char head[12];
  if (!rd(f, head, 12) || memcmp(head, "RIFF", 4) || memcmp(head + 8, "WAVE", 4)) { fclose(f); return nullptr; }
  bool have_fmt = false;
  uint16_t ch = 0;
  uint32_t sr = 0;
  for (;;) {
    unsigned char ck[8];
    uint32_t size;
    if (!rd(f, ck, 8)) { fclose(f); return nullptr; }
    memcpy(&size, ck + 4, 4);
    if (!memcmp(ck, "fmt ", 4) && size >= 16) {
      unsigned char b[16];
      uint16_t fmt, bits;
      if (!rd(f, b, 16)) { fclose(f); return nullptr; }
      memcpy(&fmt, b, 2); memcpy(&ch, b + 2, 2); memcpy(&sr, b + 4, 4); memcpy(&bits, b + 14, 2);
      if (fmt != 1 || bits != 16 || ch < 1) { fclose(f); return nullptr; }
      have_fmt = true;
      if (fseek(f, (long)(size - 16 + (size & 1)), SEEK_CUR) != 0) { fclose(f); return nullptr; }
    } else if (!memcmp(ck, "data", 4)) {
      if (!have_fmt) { fclose(f); return nullptr; }
      waveFile w = new waveFileStruct();
      w->f = f;
      w->channels = ch;
      w->data_left = size;
      if (sampleRate) *sampleRate = (int)sr;
      if (numChannels) *numChannels = ch;
      return w;
    } else if (fseek(f, (long)(size + (size & 1)), SEEK_CUR) != 0) {
      fclose(f);
      return nullptr;
    }
  }
}

waveFile openOutputWaveFile(const char* fileName, int sampleRate, int numChannels) {
  if (numChannels < 1 || sampleRate < 1) return nullptr;
  FILE* f = fopen(fileName, "wb");
  if (!f) return nullptr;
  const uint32_t zero = 0, fmtsz = 16, sr = (uint32_t)sampleRate, br = (uint32_t)sampleRate * (uint32_t)numChannels * 2u;
  const uint16_t fmt = 1, ch = (uint16_t)numChannels, align = (uint16_t)(numChannels * 2), bits = 16;
  bool ok = fwrite("RIFF", 1, 4, f) == 4 && fwrite(&zero, 4, 1, f) == 1 && fwrite("WAVEfmt ", 1, 8, f) == 8 &&
            fwrite(&fmtsz, 4, 1, f) == 1 && fwrite(&fmt, 2, 1, f) == 1 && fwrite(&ch, 2, 1, f) == 1 &&
            fwrite(&sr, 4, 1, f) == 1 && fwrite(&br, 4, 1, f) == 1 && fwrite(&align, 2, 1, f) == 1 &&
            fwrite(&bits, 2, 1, f) == 1 && fwrite("data", 1, 4, f) == 4 && fwrite(&zero, 4, 1, f) == 1;
  if (!ok) { fclose(f); return nullptr; }
  waveFile w = new waveFileStruct();
  w->f = f;
  w->writing = true;
  w->channels = numChannels;
  return w;
}

int closeWaveFile(waveFile w) {
  if (!w) return 0;
  bool ok = true;
  if (w->writing) {   // the two sizes the header left open: RIFF chunk (offset 4) and data chunk (offset 40)
    const uint32_t riff = 36 + w->data_written, bytes = w->data_written;
    ok = fseek(w->f, 4, SEEK_SET) == 0 && fwrite(&riff, 4, 1, w->f) == 1 && fseek(w->f, 40, SEEK_SET) == 0 &&
         fwrite(&bytes, 4, 1, w->f) == 1;
  }
  ok = (fclose(w->f) == 0) && ok;
  delete w;
  return ok ? 1 : 0;
}

int readFromWaveFile(waveFile w, short* buffer, int maxSamples) {
  if (!w || w->writing || maxSamples <= 0) return 0;
  const uint32_t frame = (uint32_t)w->channels * 2u;
  uint32_t want = (uint32_t)maxSamples * frame;
  if (want > w->data_left) want = w->data_left - w->data_left % frame;
  const size_t got = fread(buffer, 1, want, w->f);
  const size_t frames = got / frame;
  w->data_left -= (uint32_t)(frames * frame);
  if (got < want) w->data_left = 0;   // truncated file
  return (int)frames;
}

int writeToWaveFile(waveFile w, short* buffer, int numSamples) {
  if (!w || !w->writing) return 0;
  if (numSamples <= 0) return 1;
  const size_t n = (size_t)numSamples * (size_t)w->channels;
  if (fwrite(buffer, 2, n, w->f) != n) return 0;
  w->data_written += (uint32_t)(n * 2);
  return 1;
}

}  // extern "C"
