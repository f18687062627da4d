// speedy_wave_hip — command-line caller of the MI355X library through the reference's own C API.
//
// Plain C++ (g++, no HIP headers): it includes only include/sonic2.h and links libspeedy_hip.so, i.e. it is
// the host program a user of the reference would already have.  The processing loop is the caller loop of
// the reference's CLI (speedy_wave.cc:154-242, compress_sound): create, set speed / nonlinear / feedback,
// optional monitoring callbacks, {read <=1000 frames; write; read <=1000; write out}, flush, drain.
// WAV I/O is a minimal RIFF/PCM16 reader/writer (the reference uses libsonic's wave.c, which is not in its tree).
//
//   speedy_wave_hip --input in.wav --output out.wav [--speed 3.5] [--nonlinear 1|0] [--match_matlab]
//                   [--match_nonlinear | --length seconds] [--duration_feedback_strength 0.0]
//                   [--tension_file t] [--speed_file s] [--features_file f] [--spectrogram_file g]
//                   [--normalized_spectrogram_file n]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "sonic2.h"

static FILE* g_tension_fp = nullptr;
static FILE* g_speed_fp = nullptr;
static void tension_saver(sonicStream, int, float v) { if (g_tension_fp) fprintf(g_tension_fp, "%g\n", v); }
static void speed_saver(sonicStream, int, float v) { if (g_speed_fp) fprintf(g_speed_fp, "%g\n", v); }

static bool read_wav(const std::string& path, std::vector<int16_t>* data, int* rate, int* channels) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<unsigned char> b;
  unsigned char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + n);
  fclose(f);
  if (b.size() < 12 || memcmp(&b[0], "RIFF", 4) || memcmp(&b[8], "WAVE", 4)) return false;
  size_t pos = 12;
  bool have_fmt = false;
  while (pos + 8 <= b.size()) {
    uint32_t size;
    memcpy(&size, &b[pos + 4], 4);
    const unsigned char* body = &b[pos + 8];
    if (!memcmp(&b[pos], "fmt ", 4) && size >= 16) {
      uint16_t fmt, ch, bits;
      uint32_t sr;
      memcpy(&fmt, body, 2); memcpy(&ch, body + 2, 2); memcpy(&sr, body + 4, 4); memcpy(&bits, body + 14, 2);
      if (fmt != 1 || bits != 16) return false;
      *rate = (int)sr; *channels = ch; have_fmt = true;
    } else if (!memcmp(&b[pos], "data", 4)) {
      size_t avail = b.size() - (pos + 8);
      if (size > avail) size = (uint32_t)avail;
      data->resize(size / 2);
      memcpy(data->data(), body, (size / 2) * 2);
      return have_fmt;
    }
    pos += 8 + size + (size & 1);
  }
  return false;
}

static bool write_wav(const std::string& path, const std::vector<int16_t>& data, int rate, int channels) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  uint32_t bytes = (uint32_t)(data.size() * 2), riff = 36 + bytes, fmtsz = 16, sr = rate, br = rate * channels * 2;
  uint16_t fmt = 1, ch = (uint16_t)channels, align = (uint16_t)(channels * 2), bits = 16;
  fwrite("RIFF", 1, 4, f); fwrite(&riff, 4, 1, f); fwrite("WAVEfmt ", 1, 8, f); fwrite(&fmtsz, 4, 1, f);
  fwrite(&fmt, 2, 1, f); fwrite(&ch, 2, 1, f); fwrite(&sr, 4, 1, f); fwrite(&br, 4, 1, f);
  fwrite(&align, 2, 1, f); fwrite(&bits, 2, 1, f); fwrite("data", 1, 4, f); fwrite(&bytes, 4, 1, f);
  fwrite(data.data(), 2, data.size(), f);
  fclose(f);
  return true;
}

struct Taps {
  std::string tension, speed, features, spectrogram, normalized;
};
static FILE* g_features_fp = nullptr;
static FILE* g_spectrogram_fp = nullptr;
static FILE* g_normalized_fp = nullptr;
static int g_spectrum_bins = 0;
static void row_saver(FILE* fp, const float* v, int n) {
  if (!fp) return;
  for (int i = 0; i < n; i++) fprintf(fp, i + 1 < n ? "%g " : "%g\n", v[i]);
}
static void features_saver(sonicStream, int, float* v) { row_saver(g_features_fp, v, 15); /* speedy.h:115 */ }
static void spectrogram_saver(sonicStream, int, float* v) { row_saver(g_spectrogram_fp, v, g_spectrum_bins); }
static void normalized_saver(sonicStream, int, float* v) { row_saver(g_normalized_fp, v, g_spectrum_bins); }
static FILE* open_tap(const std::string& path) { return path.empty() ? nullptr : fopen(path.c_str(), "w"); }

// One pass of the caller loop of speedy_wave.cc:154-242.  Returns the achieved speed-up
// (frames read / frames produced); writes a WAV only when out_path is non-empty, and attaches the
// monitoring taps only on such a final nonlinear pass (speedy_wave.cc:179-187).
static double compress_sound(const std::vector<int16_t>& in, int rate, int channels, double speed, double nonlinear,
                             double feedback, const std::string& out_path, const Taps& taps) {
  sonicStream s = sonicCreateStream(rate, channels);
  if (!s) { fprintf(stderr, "sonicCreateStream failed: %s\n", speedyHipLastError()); exit(1); }
  sonicSetSpeed(s, (float)speed);
  sonicEnableNonlinearSpeedup(s, nonlinear > 0.0);
  sonicSetDurationFeedbackStrength(s, (float)feedback);
  if (nonlinear > 0.0 && !out_path.empty()) {
    g_spectrum_bins = sonicSpectrogramSize(s);
    if ((g_tension_fp = open_tap(taps.tension))) sonicTensionCallback(s, tension_saver);
    if ((g_speed_fp = open_tap(taps.speed))) sonicSpeedCallback(s, speed_saver);
    if ((g_features_fp = open_tap(taps.features))) sonicFeaturesCallback(s, features_saver);
    if ((g_spectrogram_fp = open_tap(taps.spectrogram))) sonicSpectrogramCallback(s, spectrogram_saver);
    if ((g_normalized_fp = open_tap(taps.normalized))) sonicNormalizedSpectrogramCallback(s, normalized_saver);
  }
  const int maxSamples = 1000;
  std::vector<int16_t> outbuf((size_t)maxSamples * channels), out;
  const long total = (long)(in.size() / channels);
  long produced = 0;
  for (long pos = 0; pos < total; pos += maxSamples) {
    const int n = (int)((total - pos < maxSamples) ? total - pos : maxSamples);
    if (sonicWriteShortToStream(s, &in[(size_t)pos * channels], n) <= 0) {
      fprintf(stderr, "Tried writing %d samples to sonicWrite and failed: %s\n", n, speedyHipLastError());
      exit(1);
    }
    const int got = sonicReadShortFromStream(s, outbuf.data(), maxSamples);
    out.insert(out.end(), outbuf.begin(), outbuf.begin() + (size_t)got * channels);
    produced += got;
  }
  sonicFlushStream(s);
  int got;
  do {
    got = sonicReadShortFromStream(s, outbuf.data(), maxSamples);
    out.insert(out.end(), outbuf.begin(), outbuf.begin() + (size_t)got * channels);
    produced += got;
  } while (got > 0);
  sonicDestroyStream(s);
  FILE** fps[] = {&g_tension_fp, &g_speed_fp, &g_features_fp, &g_spectrogram_fp, &g_normalized_fp};
  for (FILE** fp : fps) if (*fp) { fclose(*fp); *fp = nullptr; }
  if (!out_path.empty() && !write_wav(out_path, out, rate, channels)) {
    fprintf(stderr, "Can't open %s for speedy output.\n", out_path.c_str());
    exit(1);
  }
  printf("Compress_sound read %ld frames, and output %ld frames with nonlinear=%g.\n", total, produced, nonlinear);
  return produced ? (double)total / (double)produced : 0.0;
}

int main(int argc, char** argv) {
  std::string in_path, out_path;
  Taps taps;
  double speed = 3.5, nonlinear = 1.0, feedback = 0.0, desired_length = 0.0;  // speedy_wave.cc:32-37 defaults
  int match_matlab = 0, match_nonlinear = 0;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); }
      return argv[++i];
    };
    if (a == "--input") in_path = next("--input");
    else if (a == "--output") out_path = next("--output");
    else if (a == "--speed") speed = atof(next("--speed"));
    else if (a == "--nonlinear") nonlinear = atof(next("--nonlinear"));
    else if (a == "--linear") nonlinear = 0.0;
    else if (a == "--match_nonlinear") match_nonlinear = 1;
    else if (a == "--length") desired_length = atof(next("--length"));
    else if (a == "--duration_feedback_strength") feedback = atof(next("--duration_feedback_strength"));
    else if (a == "--tension_file") taps.tension = next("--tension_file");
    else if (a == "--speed_file") taps.speed = next("--speed_file");
    else if (a == "--features_file") taps.features = next("--features_file");
    else if (a == "--spectrogram_file") taps.spectrogram = next("--spectrogram_file");
    else if (a == "--normalized_spectrogram_file") taps.normalized = next("--normalized_spectrogram_file");
    else if (a == "--match_matlab") match_matlab = 1;
    else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
  }
  if (in_path.empty() || (out_path.empty() && (match_nonlinear || desired_length > 0))) {
    fprintf(stderr, "usage: speedy_wave_hip --input in.wav [--output out.wav] [--speed S] [--nonlinear 1|0] "
                    "[--match_nonlinear] [--length seconds] ...\n");
    return 2;
  }
  std::vector<int16_t> in;
  int rate = 0, channels = 0;
  if (!read_wav(in_path, &in, &rate, &channels)) { fprintf(stderr, "Can't open %s for speedy input.\n", in_path.c_str()); return 1; }
  printf("Read %d channel data at a sample rate of %d.\n", channels, rate);
  speedyHipSetMatchMatlab(match_matlab);

  // Two-pass drivers of speedy_wave.cc:424-461: a probing nonlinear pass without output first.
  if (match_nonlinear) {
    // Find the overall rate the nonlinear algorithm achieves for this file; the final pass then runs at
    // that rate with the user's own --nonlinear (normally 0.0, speedy_wave.cc:62).
    speed = compress_sound(in, rate, channels, speed, 1.0, feedback, "", taps);
  } else if (desired_length > 0) {
    const long total = (long)(in.size() / channels);
    const double want = ((float)total / (float)rate) / desired_length;
    printf("Read %ld frames, and trying to speed up with a factor of %g.\n", total, want);
    const double got = compress_sound(in, rate, channels, want, 1.0, feedback, "", taps);
    speed = want * (want / got);
    printf("First scaling by %g gave a speed of %g.\n", want, got);
  }
  const double actual = compress_sound(in, rate, channels, speed, nonlinear, feedback, out_path, taps);
  printf("Actual speedup: %g\n", actual);
  return 0;
}
