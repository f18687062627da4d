/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see orc_sonic.h for why and for the anchors).
 *
 * Restatement of the published PICOLA/AMDF time-scale modification of github.com/waywardgeek/sonic
 * (un-pinned, un-vendored dependency of the reference: Makefile:7,74), as the reference drives it
 * through sonicInt* (soniclib.c:354,369,521,547,551).  Written from the algorithm description in
 * SURVEY.md Appendix A; structure (absolute stream positions, growable vectors) is this repo's own.
 *
 * All sample arithmetic is integer; the only floating point is the float speed in
 *   n = (long)(period / (speed - 1.0f))                       speed >= 2
 *   remaining = (int)(period * (2.0f - speed) / (speed - 1.0f))   1 < speed < 2
 * and the mirrored slow-down forms, evaluated in float exactly as written.
 */
#include "orc_sonic.h"

#include <stdlib.h>
#include <string.h>

#define ORC_MIN_PITCH 65
#define ORC_MAX_PITCH 400
#define ORC_AMDF_FREQ 4000

typedef struct {
  short* d;
  long n;   /* frames (multi-channel samples) held */
  long cap; /* frames allocated */
} orc_vec;

struct orc_sonic {
  orc_vec in, out;
  orc_vec pitch;            /* rate != 1: speed-changed samples waiting to be resampled (the dependency's pitchBuffer) */
  int oldRatePosition, newRatePosition;
  short* down; /* mono / decimated scratch, maxRequired entries */
  void* userData;
  float speed, rate;
  int numChannels, sampleRate;
  int minPeriod, maxPeriod, maxRequired;
  int remainingInputToCopy;
  int prevPeriod, prevMinDiff;
  long steps;
  int* periodLog;
  long periodLogCap;
};

static int vec_reserve(orc_vec* v, long extra, int ch) {
  if (v->n + extra > v->cap) {
    long ncap = v->cap + (v->cap >> 1) + extra;
    short* nd = (short*)realloc(v->d, sizeof(short) * ncap * ch);
    if (!nd) return 0;
    v->d = nd;
    v->cap = ncap;
  }
  return 1;
}

orc_sonicStream orc_sonicIntCreateStream(int sampleRate, int numChannels) {
  orc_sonicStream s = (orc_sonicStream)calloc(1, sizeof(struct orc_sonic));
  if (!s) return NULL;
  s->speed = 1.0f;
  s->rate = 1.0f;
  s->numChannels = numChannels;
  s->sampleRate = sampleRate;
  s->minPeriod = sampleRate / ORC_MAX_PITCH;
  s->maxPeriod = sampleRate / ORC_MIN_PITCH;
  s->maxRequired = 2 * s->maxPeriod;
  s->down = (short*)calloc(s->maxRequired, sizeof(short));
  if (!s->down) { free(s); return NULL; }
  return s;
}
void orc_sonicIntDestroyStream(orc_sonicStream s) {
  if (!s) return;
  free(s->in.d); free(s->out.d); free(s->pitch.d); free(s->down); free(s);
}
void orc_sonicIntSetUserData(orc_sonicStream s, void* p) { s->userData = p; }
void* orc_sonicIntGetUserData(orc_sonicStream s) { return s->userData; }
int orc_sonicIntGetNumChannels(orc_sonicStream s) { return s->numChannels; }
int orc_sonicIntGetSampleRate(orc_sonicStream s) { return s->sampleRate; }
void orc_sonicIntSetSpeed(orc_sonicStream s, float speed) { s->speed = speed; }
float orc_sonicIntGetSpeed(orc_sonicStream s) { return s->speed; }
void orc_sonicIntSetRate(orc_sonicStream s, float rate) {
  s->rate = rate;
  s->oldRatePosition = 0;
  s->newRatePosition = 0;
}
int orc_sonicIntSamplesAvailable(orc_sonicStream s) { return (int)s->out.n; }
long orc_sonicIntStepCount(orc_sonicStream s) { return s->steps; }
void orc_sonicIntSetPeriodLog(orc_sonicStream s, int* log, long capacity) {
  s->periodLog = log;
  s->periodLogCap = capacity;
}

static int copy_to_output(orc_sonicStream s, const short* src, int n) {
  if (!vec_reserve(&s->out, n, s->numChannels)) return 0;
  memcpy(s->out.d + s->out.n * s->numChannels, src, sizeof(short) * (size_t)n * s->numChannels);
  s->out.n += n;
  return 1;
}

/* Mean of `skip` consecutive frames over all channels (truncating integer division). */
static void down_sample(orc_sonicStream s, const short* samples, int skip) {
  int count = s->maxRequired / skip;
  int per = s->numChannels * skip;
  for (int i = 0; i < count; i++) {
    int value = 0;
    for (int j = 0; j < per; j++) value += *samples++;
    value /= per;
    s->down[i] = (short)value;
  }
}

/* AMDF over lags [minP, maxP]: best = first lag minimising diff/lag, worst = first maximising. */
static int amdf_search(const short* x, int minP, int maxP, int* retMin, int* retMax) {
  int best = 0, worst = 255;
  unsigned long minDiff = 1, maxDiff = 0;
  for (int p = minP; p <= maxP; p++) {
    unsigned long diff = 0;
    for (int i = 0; i < p; i++) {
      int a = x[i], b = x[i + p];
      diff += (unsigned long)(a >= b ? a - b : b - a);
    }
    if (best == 0 || diff * (unsigned long)best < minDiff * (unsigned long)p) { minDiff = diff; best = p; }
    if (diff * (unsigned long)worst > maxDiff * (unsigned long)p) { maxDiff = diff; worst = p; }
  }
  *retMin = (int)(minDiff / (unsigned long)best);
  *retMax = (int)(maxDiff / (unsigned long)worst);
  return best;
}

static int prev_period_better(orc_sonicStream s, int minDiff, int maxDiff) {
  if (minDiff == 0 || s->prevPeriod == 0) return 0;
  if (maxDiff > minDiff * 3) return 0;              /* a clear match this time */
  if (minDiff * 2 <= s->prevMinDiff * 3) return 0;  /* not much worse than last time */
  return 1;
}

static int find_pitch_period(orc_sonicStream s, const short* samples) {
  int minDiff, maxDiff, period, ret;
  int skip = (s->sampleRate > ORC_AMDF_FREQ) ? s->sampleRate / ORC_AMDF_FREQ : 1;
  if (s->numChannels == 1 && skip == 1) {
    period = amdf_search(samples, s->minPeriod, s->maxPeriod, &minDiff, &maxDiff);
  } else {
    down_sample(s, samples, skip);
    period = amdf_search(s->down, s->minPeriod / skip, s->maxPeriod / skip, &minDiff, &maxDiff);
    if (skip != 1) {
      period *= skip;
      int lo = period - (skip << 2), hi = period + (skip << 2);
      if (lo < s->minPeriod) lo = s->minPeriod;
      if (hi > s->maxPeriod) hi = s->maxPeriod;
      if (s->numChannels == 1) {
        period = amdf_search(samples, lo, hi, &minDiff, &maxDiff);
      } else {
        down_sample(s, samples, 1);
        period = amdf_search(s->down, lo, hi, &minDiff, &maxDiff);
      }
    }
  }
  ret = prev_period_better(s, minDiff, maxDiff) ? s->prevPeriod : period;
  s->prevMinDiff = minDiff;
  s->prevPeriod = period;
  return ret;
}

/* out[t] = (down[t]*(n-t) + up[t]*t) / n per channel, int arithmetic, truncating division. */
static void overlap_add(int n, int ch, short* out, const short* rampDown, const short* rampUp) {
  for (int c = 0; c < ch; c++) {
    for (int t = 0; t < n; t++) {
      int d = rampDown[t * ch + c], u = rampUp[t * ch + c];
      out[t * ch + c] = (short)((d * (n - t) + u * t) / n);
    }
  }
}

static int skip_pitch_period(orc_sonicStream s, const short* samples, float speed, int period) {
  long n;
  int ch = s->numChannels;
  if (speed >= 2.0f) {
    n = (long)(period / (speed - 1.0f));
  } else {
    n = period;
    s->remainingInputToCopy = (int)(period * (2.0f - speed) / (speed - 1.0f));
  }
  if (!vec_reserve(&s->out, n, ch)) return 0;
  overlap_add((int)n, ch, s->out.d + s->out.n * ch, samples, samples + (long)period * ch);
  s->out.n += n;
  return (int)n;
}

static int insert_pitch_period(orc_sonicStream s, const short* samples, float speed, int period) {
  long n;
  int ch = s->numChannels;
  if (speed < 0.5f) {
    n = (long)(period * speed / (1.0f - speed));
  } else {
    n = period;
    s->remainingInputToCopy = (int)(period * (2.0f * speed - 1.0f) / (1.0f - speed));
  }
  if (!vec_reserve(&s->out, period + n, ch)) return 0;
  short* out = s->out.d + s->out.n * ch;
  memcpy(out, samples, sizeof(short) * (size_t)period * ch);
  overlap_add((int)n, ch, out + (long)period * ch, samples + (long)period * ch, samples);
  s->out.n += period + n;
  return (int)n;
}

static int change_speed(orc_sonicStream s, float speed) {
  long numSamples = s->in.n;
  long position = 0;
  int maxRequired = s->maxRequired;
  if (s->in.n < maxRequired) return 1;
  do {
    int newSamples;
    if (s->remainingInputToCopy > 0) {
      newSamples = s->remainingInputToCopy;
      if (newSamples > maxRequired) newSamples = maxRequired;
      if (!copy_to_output(s, s->in.d + position * s->numChannels, newSamples)) return 0;
      s->remainingInputToCopy -= newSamples;
      position += newSamples;
    } else {
      const short* samples = s->in.d + position * s->numChannels;
      int period = find_pitch_period(s, samples);
      if (s->periodLog && s->steps < s->periodLogCap) s->periodLog[s->steps] = period;
      s->steps++;
      if (speed > 1.0) {
        newSamples = skip_pitch_period(s, samples, speed, period);
        position += period + newSamples;
      } else {
        newSamples = insert_pitch_period(s, samples, speed, period);
        position += newSamples;
      }
    }
    if (newSamples == 0) return 0;
  } while (position + maxRequired <= numSamples);
  long remaining = s->in.n - position;
  if (remaining > 0)
    memmove(s->in.d, s->in.d + position * s->numChannels, sizeof(short) * (size_t)remaining * s->numChannels);
  s->in.n = remaining;
  return 1;
}

/* Rate stage (playback-rate change by resampling), the classic revision of the dependency: everything the speed stage
 * has just appended to the output is moved to a pitch buffer and re-emitted by LINEAR interpolation at
 * newSampleRate/oldSampleRate, both halved until they fit 14 bits; the two positions run modulo the (reduced) rates;
 * one input sample always stays behind (the interpolation needs its right neighbour).  Integer arithmetic, truncating
 * division.  PARITY UNPINNED like the rest of this file (newer revisions of the dependency interpolate with a windowed
 * sinc instead); the reference only forwards sonicSetRate (soniclib.c:169-175) and tests nothing about it. */
static short interpolate(orc_sonicStream s, const short* in, int oldSampleRate, int newSampleRate) {
  short left = *in, right = in[s->numChannels];
  int position = s->newRatePosition * oldSampleRate;
  int leftPosition = s->oldRatePosition * newSampleRate;
  int rightPosition = (s->oldRatePosition + 1) * newSampleRate;
  int ratio = rightPosition - position;
  int width = rightPosition - leftPosition;
  return (short)((ratio * left + (width - ratio) * right) / width);
}

static int adjust_rate(orc_sonicStream s, float rate, long originalNumOutput) {
  int newSampleRate = (int)(s->sampleRate / rate);
  int oldSampleRate = s->sampleRate;
  int ch = s->numChannels;
  long position;
  while (newSampleRate > (1 << 14) || oldSampleRate > (1 << 14)) { newSampleRate >>= 1; oldSampleRate >>= 1; }
  if (s->out.n == originalNumOutput) return 1;
  /* move the new samples to the pitch buffer */
  long moved = s->out.n - originalNumOutput;
  if (!vec_reserve(&s->pitch, moved, ch)) return 0;
  memcpy(s->pitch.d + s->pitch.n * ch, s->out.d + originalNumOutput * ch, sizeof(short) * (size_t)moved * ch);
  s->pitch.n += moved;
  s->out.n = originalNumOutput;
  for (position = 0; position < s->pitch.n - 1; position++) {
    while ((s->oldRatePosition + 1) * newSampleRate > s->newRatePosition * oldSampleRate) {
      if (!vec_reserve(&s->out, 1, ch)) return 0;
      short* out = s->out.d + s->out.n * ch;
      const short* in = s->pitch.d + position * ch;
      for (int i = 0; i < ch; i++) out[i] = interpolate(s, in + i, oldSampleRate, newSampleRate);
      s->newRatePosition++;
      s->out.n++;
    }
    s->oldRatePosition++;
    if (s->oldRatePosition == oldSampleRate) {
      s->oldRatePosition = 0;
      s->newRatePosition = 0;  /* == newSampleRate here, by construction */
    }
  }
  /* remove the consumed pitch samples */
  if (position > 0) {
    long rem = s->pitch.n - position;
    if (rem > 0) memmove(s->pitch.d, s->pitch.d + position * ch, sizeof(short) * (size_t)rem * ch);
    s->pitch.n = rem;
  }
  return 1;
}

static int process_input(orc_sonicStream s) {
  long originalNumOutput = s->out.n;
  float speed = s->speed;
  float rate = s->rate;
  if (speed > 1.00001 || speed < 0.99999) {
    change_speed(s, speed);
  } else {
    if (!copy_to_output(s, s->in.d, (int)s->in.n)) return 0;
    s->in.n = 0;
  }
  if (rate != 1.0f) return adjust_rate(s, rate, originalNumOutput);
  return 1;
}

int orc_sonicIntWriteShortToStream(orc_sonicStream s, const short* samples, int numSamples) {
  if (numSamples > 0 && samples) {
    if (!vec_reserve(&s->in, numSamples, s->numChannels)) return 0;
    memcpy(s->in.d + s->in.n * s->numChannels, samples, sizeof(short) * (size_t)numSamples * s->numChannels);
    s->in.n += numSamples;
  }
  return process_input(s);
}

int orc_sonicIntWriteFloatToStream(orc_sonicStream s, const float* samples, int numSamples) {
  if (numSamples > 0 && samples) {
    if (!vec_reserve(&s->in, numSamples, s->numChannels)) return 0;
    short* dst = s->in.d + s->in.n * s->numChannels;
    long count = (long)numSamples * s->numChannels;
    for (long i = 0; i < count; i++) dst[i] = (short)(samples[i] * 32767.0f);
    s->in.n += numSamples;
  }
  return process_input(s);
}

int orc_sonicIntReadShortFromStream(orc_sonicStream s, short* samples, int maxSamples) {
  long n = s->out.n, rem = 0;
  if (n == 0) return 0;
  if (n > maxSamples) { rem = n - maxSamples; n = maxSamples; }
  memcpy(samples, s->out.d, sizeof(short) * (size_t)n * s->numChannels);
  if (rem > 0) memmove(s->out.d, s->out.d + n * s->numChannels, sizeof(short) * (size_t)rem * s->numChannels);
  s->out.n = rem;
  return (int)n;
}

int orc_sonicIntReadFloatFromStream(orc_sonicStream s, float* samples, int maxSamples) {
  long n = s->out.n, rem = 0;
  if (n == 0) return 0;
  if (n > maxSamples) { rem = n - maxSamples; n = maxSamples; }
  long count = n * s->numChannels;
  for (long i = 0; i < count; i++) samples[i] = s->out.d[i] / 32767.0f;
  if (rem > 0) memmove(s->out.d, s->out.d + n * s->numChannels, sizeof(short) * (size_t)rem * s->numChannels);
  s->out.n = rem;
  return (int)n;
}

int orc_sonicIntFlushStream(orc_sonicStream s) {
  int maxRequired = s->maxRequired;
  long remaining = s->in.n;
  float speed = s->speed;
  float rate = s->rate;
  long expected = s->out.n + (int)((remaining / speed + s->pitch.n) / rate + 0.5f);
  if (!vec_reserve(&s->in, 2 * maxRequired, s->numChannels)) return 0;
  memset(s->in.d + remaining * s->numChannels, 0, sizeof(short) * 2 * (size_t)maxRequired * s->numChannels);
  s->in.n += 2 * maxRequired;
  if (!orc_sonicIntWriteShortToStream(s, NULL, 0)) return 0;
  if (s->out.n > expected) s->out.n = expected;
  s->in.n = 0;
  s->remainingInputToCopy = 0;
  s->pitch.n = 0;
  return 1;
}
