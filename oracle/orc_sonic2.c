/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_sonic2.h).
 *
 * Restatement of the reference shim, soniclib.c: ring of `bufferCount` buffers of frameStep
 * multi-channel samples, analysis frames of 1.5 x frameStep scheduled as soon as their last sample
 * arrives, audio delayed by kTemporalHysteresisFuture frames, one (setSpeed, write) pair per buffer
 * into the TSM stage.  Line citations are into reference soniclib.c.
 */
#include "orc_sonic2.h"

#include <assert.h>
#include <stdlib.h>
#include <string.h>

#include "orc_speedy.h"

typedef struct { /* soniclib.c:61-82 */
  orc_speedyStream speedy;
  float globalSpeed, nonlinearFactor, feedbackStrength, sampleRate;
  int channelCount, bufferCount, bufferSize;
  short** bufferList;
  short* speedyInputBuffer;
  int readIdx, speedyIdx, writeIdx, writeLoc;
  orc_tensionFunction returnTension;
  orc_speedFunction returnSpeed;
  orc_featuresFunction returnFeatures;
  orc_spectrogramFunction returnSpectrogram, returnNormalizedSpectrogram;
  orc_handoffFunction returnHandoff;
  float lastSpeedSet;
} orc_conn;

static orc_conn* conn_of(orc_sonicStream s) { return (orc_conn*)orc_sonicIntGetUserData(s); }

orc_sonicStream orc_sonicCreateStream(int sampleRate, int numChannels, int match_matlab) { /* :93-134 */
  orc_sonicStream s = orc_sonicIntCreateStream(sampleRate, numChannels);
  if (!s) return NULL;
  orc_conn* c = (orc_conn*)calloc(1, sizeof(orc_conn));
  if (!c) { orc_sonicIntDestroyStream(s); return NULL; }
  orc_sonicIntSetUserData(s, c);
  c->speedy = orc_speedyCreateStream(sampleRate, match_matlab);
  if (!c->speedy) { orc_sonicDestroyStream(s); return NULL; }
  c->globalSpeed = 1.0;
  c->sampleRate = sampleRate;
  c->channelCount = numChannels;
  c->nonlinearFactor = 0.0;
  c->feedbackStrength = 0.1; /* :122 */
  c->lastSpeedSet = 1.0f;
  return s;
}

void orc_sonicDestroyStream(orc_sonicStream s) { /* :141-167 */
  assert(s);
  orc_conn* c = conn_of(s);
  orc_sonicIntDestroyStream(s);
  if (!c) return;
  if (c->speedy) orc_speedyDestroyStream(c->speedy);
  if (c->bufferList) {
    for (int i = 0; i < c->bufferCount; i++) free(c->bufferList[i]);
    free(c->bufferList);
  }
  free(c->speedyInputBuffer);
  free(c);
}

void orc_sonicSetRate(orc_sonicStream s, float rate) { /* :169-175 */
  conn_of(s)->sampleRate = rate;
  orc_sonicIntSetRate(s, rate);
}
void orc_sonicSetSpeed(orc_sonicStream s, float speed) { /* :177-183 */
  conn_of(s)->globalSpeed = speed;
  conn_of(s)->lastSpeedSet = speed;
  orc_sonicIntSetSpeed(s, speed);
}

static int allocate_buffers(orc_sonicStream s, int sampleCount) { /* :186-233 */
  orc_conn* c = conn_of(s);
  c->bufferSize = orc_speedyInputFrameStep(c->speedy);
  int count = sampleCount / c->bufferSize + 1;
  int minCount = 2 + orc_speedyHysteresisFuture(c->speedy); /* kMinBufferSize, :91 */
  if (count < minCount) count = minCount;
  c->bufferCount = count;
  c->bufferList = (short**)calloc(count, sizeof(short*));
  if (!c->bufferList) return 0;
  for (int i = 0; i < count; i++) {
    c->bufferList[i] = (short*)calloc(c->bufferSize, sizeof(short) * c->channelCount);
    if (!c->bufferList[i]) return 0;
  }
  c->speedyInputBuffer = (short*)calloc(orc_speedyInputFrameSize(c->speedy), sizeof(short));
  return c->speedyInputBuffer != NULL;
}

static void send_data_to_speedy(orc_sonicStream s) { /* :246-373 */
  orc_conn* c = conn_of(s);
  int W = orc_speedyInputFrameSize(c->speedy);
  int B = c->bufferSize, C = c->channelCount;
  int full = W / B, partial = W - B * full;
  short* bp = c->speedyInputBuffer;
  for (int i = 0; i < full; i++) { /* :265-276, mono mix = integer mean, truncating */
    const short* wp = c->bufferList[(c->speedyIdx + i) % c->bufferCount];
    for (int j = 0; j < B; j++) {
      int sum = 0;
      for (int k = 0; k < C; k++) sum += wp[j * C + k];
      *bp++ = (short)(sum / C);
    }
  }
  const short* wp = c->bufferList[(c->speedyIdx + full) % c->bufferCount]; /* :278-287 */
  for (int i = 0; i < partial; i++) {
    int sum = 0;
    for (int k = 0; k < C; k++) sum += wp[i * C + k];
    *bp++ = (short)(sum / C);
  }
  c->speedyIdx++;
  orc_speedyAddDataShort(c->speedy, c->speedyInputBuffer, c->writeIdx); /* :295 */
  if (c->returnSpectrogram) c->returnSpectrogram(s, c->writeIdx, orc_speedyGetSpectrogram(c->speedy));
  if (c->returnNormalizedSpectrogram)
    c->returnNormalizedSpectrogram(s, c->writeIdx, orc_speedyGetNormalizedSpectrogram(c->speedy));
  float tension = 0.0;
  if (orc_speedyComputeTension(c->speedy, c->readIdx, &tension)) { /* :317 */
    if (c->returnTension) c->returnTension(s, c->readIdx, tension);
    if (c->returnFeatures) c->returnFeatures(s, c->readIdx, orc_speedyGetInternalState(c->speedy));
    float newRate = orc_speedyComputeSpeedFromTension(tension, c->globalSpeed, c->feedbackStrength, c->speedy);
    float globalSpeed = c->globalSpeed;
    newRate = newRate * c->nonlinearFactor + globalSpeed * (1 - c->nonlinearFactor); /* :344-345 */
    if (c->returnSpeed) c->returnSpeed(s, c->readIdx, newRate);
    orc_sonicIntSetSpeed(s, newRate); /* :354 */
    c->lastSpeedSet = newRate;
    short* readBuffer = c->bufferList[c->readIdx % c->bufferCount];
    if (c->returnHandoff) c->returnHandoff(s, c->readIdx, newRate, readBuffer, B);
    orc_sonicIntWriteShortToStream(s, readBuffer, B); /* :369 */
    c->readIdx++;
  }
}

static int write_common(orc_sonicStream s, const short* inS, const float* inF, int sampleCount) { /* :391-517 */
  orc_conn* c = conn_of(s);
  if (!c->nonlinearFactor) { /* :397-399, :463-465 */
    return inS ? orc_sonicIntWriteShortToStream(s, inS, sampleCount)
               : orc_sonicIntWriteFloatToStream(s, inF, sampleCount);
  }
  if (!c->bufferList) allocate_buffers(s, sampleCount);
  int W = orc_speedyInputFrameSize(c->speedy);
  int B = c->bufferSize, C = c->channelCount;
  int full = W / B, partialNeeded = W - full * B;
  while ((inS || inF) && sampleCount > 0) {
    short* wb = c->bufferList[c->writeIdx % c->bufferCount];
    for (int j = 0; j < C; j++) {
      if (inS) wb[c->writeLoc * C + j] = *inS++;
      else wb[c->writeLoc * C + j] = (short)(*inF++ * 32768.0); /* :496 */
    }
    c->writeLoc++;
    sampleCount--;
    if (c->writeIdx >= c->speedyIdx + full && c->writeLoc == partialNeeded + 1) send_data_to_speedy(s);
    if (c->writeLoc >= B) { c->writeLoc = 0; c->writeIdx++; }
  }
  return 1;
}
int orc_sonicWriteShortToStream(orc_sonicStream s, const short* in, int n) { return write_common(s, in, NULL, n); }
int orc_sonicWriteFloatToStream(orc_sonicStream s, const float* in, int n) { return write_common(s, NULL, in, n); }
int orc_sonicReadShortFromStream(orc_sonicStream s, short* out, int n) { return orc_sonicIntReadShortFromStream(s, out, n); }
int orc_sonicReadFloatFromStream(orc_sonicStream s, float* out, int n) { return orc_sonicIntReadFloatFromStream(s, out, n); }

int orc_sonicFlushStream(orc_sonicStream s) { /* :529-552 */
  orc_conn* c = conn_of(s);
  while (c->readIdx < c->writeIdx) {
    short* cur = c->bufferList[c->readIdx % c->bufferCount];
    if (c->returnHandoff) c->returnHandoff(s, c->readIdx, c->lastSpeedSet, cur, c->bufferSize);
    orc_sonicIntWriteShortToStream(s, cur, c->bufferSize);
    c->readIdx++;
  }
  return orc_sonicIntFlushStream(s);
}

void orc_sonicEnableNonlinearSpeedup(orc_sonicStream s, float f) { conn_of(s)->nonlinearFactor = f; }
void orc_sonicSetDurationFeedbackStrength(orc_sonicStream s, float f) { conn_of(s)->feedbackStrength = f; }
void orc_sonicTensionCallback(orc_sonicStream s, orc_tensionFunction f) { conn_of(s)->returnTension = f; }
void orc_sonicSpeedCallback(orc_sonicStream s, orc_speedFunction f) { conn_of(s)->returnSpeed = f; }
void orc_sonicFeaturesCallback(orc_sonicStream s, orc_featuresFunction f) { conn_of(s)->returnFeatures = f; }
void orc_sonicSpectrogramCallback(orc_sonicStream s, orc_spectrogramFunction f) { conn_of(s)->returnSpectrogram = f; }
void orc_sonicNormalizedSpectrogramCallback(orc_sonicStream s, orc_spectrogramFunction f) {
  conn_of(s)->returnNormalizedSpectrogram = f;
}
void orc_sonicHandoffCallback(orc_sonicStream s, orc_handoffFunction f) { conn_of(s)->returnHandoff = f; }
int orc_sonicSpectrogramSize(orc_sonicStream s) { return s ? orc_speedyFFTSize(conn_of(s)->speedy) : 0; }
int orc_getSonicBufferSize(orc_sonicStream s) { return s ? conn_of(s)->bufferSize : 0; }

/* ---- compress_sound (speedy_wave.cc:154-242) as one call, with taps collected through callbacks ---- */
typedef struct {
  float *tension, *speed, *features;
  long cap, n_tension, n_speed, n_features;
} orc_taps;
static __thread orc_taps* g_taps;
static void tap_tension(orc_sonicStream s, int t, float v) {
  (void)s; (void)t;
  if (g_taps->tension && g_taps->n_tension < g_taps->cap) g_taps->tension[g_taps->n_tension] = v;
  g_taps->n_tension++;
}
static void tap_speed(orc_sonicStream s, int t, float v) {
  (void)s; (void)t;
  if (g_taps->speed && g_taps->n_speed < g_taps->cap) g_taps->speed[g_taps->n_speed] = v;
  g_taps->n_speed++;
}
static void tap_features(orc_sonicStream s, int t, float* f) {
  (void)s; (void)t;
  if (g_taps->features && g_taps->n_features < g_taps->cap)
    memcpy(g_taps->features + g_taps->n_features * ORC_FEATURE_COUNT, f, sizeof(float) * ORC_FEATURE_COUNT);
  g_taps->n_features++;
}

long orc_compress_sound(const short* in, long n_in, int sampleRate, int numChannels, float speed,
                        float nonlinear, float feedback, int match_matlab, int chunk, short* out,
                        long out_capacity, float* tension_tap, float* speed_tap, float* features_tap,
                        long tap_capacity, long* n_taps) {
  orc_taps taps = {tension_tap, speed_tap, features_tap, tap_capacity, 0, 0, 0};
  g_taps = &taps;
  orc_sonicStream s = orc_sonicCreateStream(sampleRate, numChannels, match_matlab);
  if (!s) return -1;
  orc_sonicSetSpeed(s, speed);
  orc_sonicEnableNonlinearSpeedup(s, nonlinear);
  orc_sonicSetDurationFeedbackStrength(s, feedback);
  if (nonlinear != 0 && (tension_tap || speed_tap || features_tap || n_taps)) {
    orc_sonicTensionCallback(s, tap_tension);
    orc_sonicSpeedCallback(s, tap_speed);
    orc_sonicFeaturesCallback(s, tap_features);
  }
  long produced = 0;
  short* scratch = (short*)malloc(sizeof(short) * (size_t)chunk * numChannels);
  int overflow = 0;
  for (long pos = 0; pos < n_in; pos += chunk) {
    int n = (int)((n_in - pos < chunk) ? n_in - pos : chunk);
    orc_sonicWriteShortToStream(s, in + pos * numChannels, n);
    int got = orc_sonicReadShortFromStream(s, scratch, chunk);
    if (produced + got > out_capacity) { overflow = 1; break; }
    memcpy(out + produced * numChannels, scratch, sizeof(short) * (size_t)got * numChannels);
    produced += got;
  }
  if (!overflow) {
    orc_sonicFlushStream(s);
    int got;
    do {
      got = orc_sonicReadShortFromStream(s, scratch, chunk);
      if (produced + got > out_capacity) { overflow = 1; break; }
      memcpy(out + produced * numChannels, scratch, sizeof(short) * (size_t)got * numChannels);
      produced += got;
    } while (got > 0);
  }
  free(scratch);
  orc_sonicDestroyStream(s);
  if (n_taps) *n_taps = taps.n_tension;
  g_taps = NULL;
  return overflow ? -1 : produced;
}
