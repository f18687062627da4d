// Rate stage behind sonicSetRate != 1 (sonic2.h:70; the reference forwards it to the TSM dependency, soniclib.c:169-175):
// the dependency's adjustRate -- linear interpolation of the speed stage's output at newSampleRate / oldSampleRate --
// as a data-parallel kernel.  The sequential original (oracle/orc_sonic.c adjust_rate) walks the input with two
// positions; in closed form, with t the input position and k the output count since the rate was set, output k is
// emitted at t = floor(k*old/new) as (ratio*in[t] + (new - ratio)*in[t+1]) / new, ratio = (t+1)*new - k*old, and an
// input sample is consumed only once its right neighbour exists -- so one sample stays behind between calls.
// PARITY UNPINNED like the whole TSM stage (DESIGN.md "Oracle"): bit-exact against the oracle's restatement.
#include "spx_internal.h"

__global__ void __launch_bounds__(256)
spx_rate_kernel(SpxRateState* rs, const SpxStreamState* st, const int64_t* tsm_n_p, const int16_t* __restrict__ tsm,
                int16_t* __restrict__ fin, int64_t fin_cap, int C, int oldR, int newR, float rate, int bypass, int flush) {
  __shared__ SpxRateState S;
  __shared__ int s_over;
  const int tid = threadIdx.x;
  if (tid == 0) { S = *rs; s_over = 0; }
  __syncthreads();
  int64_t tn = *tsm_n_p;
  int over = S.overflow;
  if (tn == SPX_NOUT_LOST_PRODUCER) { tn = S.tsm_seen; over = 2; }
  else if (tn < 0) { tn = -tn; over = over ? over : 1; }
  const int64_t seen = S.tsm_seen;
  const int64_t newCount = tn > seen ? tn - seen : 0;
  const int64_t finalBefore = S.final_n;
  const int hasLeft = S.has_left;
  int64_t nOut = 0;
  int64_t o0 = S.old_pos, k0 = S.new_pos, kEnd = k0;
  const int64_t M = hasLeft + newCount;
  if (bypass) {
    nOut = newCount;
    for (int64_t e = tid; e < newCount * C; e += 256) {
      const int64_t f = finalBefore + e / C;
      if (f < fin_cap) fin[finalBefore * C + e] = tsm[seen * C + e];
      else s_over = 1;
    }
  } else {
    if (M >= 2) {
      kEnd = ((o0 + M - 1) * (int64_t)newR + oldR - 1) / oldR;  // outputs k with floor(k*old/new) <= o0 + M - 2
      nOut = kEnd - k0;
    }
    for (int64_t j = tid; j < nOut; j += 256) {
      const int64_t k = k0 + j;
      const int64_t t = (k * oldR) / newR;
      const int64_t p = t - o0;                          // position in [left?] ++ tsm[seen ..)
      const int ratio = (int)((t + 1) * newR - k * oldR);
      const int64_t f = finalBefore + j;
      if (f >= fin_cap) { s_over = 1; continue; }
      for (int c = 0; c < C; c++) {
        const int l = (p == 0 && hasLeft) ? (int)S.left[c] : (int)tsm[(seen + p - hasLeft) * C + c];
        const int r = (int)tsm[(seen + p + 1 - hasLeft) * C + c];
        fin[f * C + c] = (int16_t)((ratio * l + (newR - ratio) * r) / newR);
      }
    }
  }
  __syncthreads();  // every thread has read S.left / S.* before thread 0 rewrites the record
  if (tid == 0) {
    SpxRateState R = S;
    R.tsm_seen = tn;
    R.final_n = finalBefore + nOut;
    if (!bypass && M >= 1) {
      const int64_t t_end = o0 + M - 1;
      const int64_t wraps = t_end / oldR;
      R.old_pos = (int32_t)(t_end - wraps * oldR);
      R.new_pos = (int32_t)(kEnd - wraps * (int64_t)newR);
      if (newCount > 0)
        for (int c = 0; c < C; c++) R.left[c] = tsm[(tn - 1) * C + c];
      R.has_left = 1;
    }
    if (flush) {
      // sonicIntFlushStream: expected = numOutputSamples + (int)((remaining/speed + numPitchSamples)/rate + 0.5f), computed
      // before the padded input is processed; the output is cut back to it and the pitch buffer emptied
      // "before" = after everything this job's ordinary events produced (the shim hands its remaining ring buffers to
      // the TSM stage before it flushes it, soniclib.c:538-551): TSM frames up to flush_out_mark
      const float speed = st->curSpeed;
      int64_t mark = st->flush_out_mark;
      if (mark < seen) mark = seen;
      if (mark > tn) mark = tn;
      const int64_t MA = hasLeft + (mark - seen);
      int64_t outA, leftA;
      if (bypass) { outA = mark - seen; leftA = hasLeft; }
      else { outA = (MA >= 2) ? ((o0 + MA - 1) * (int64_t)newR + oldR - 1) / oldR - k0 : 0; leftA = MA >= 1 ? 1 : 0; }
      const int64_t expected = finalBefore + outA + (int)(((float)st->flush_remaining / speed + (float)leftA) / rate + 0.5f);
      if (R.final_n > expected) R.final_n = expected;
      R.has_left = 0;
    }
    R.overflow = (over || s_over) ? (over == 2 ? 2 : 1) : 0;
    *rs = R;
  }
}

void spx_launch_rate(SpxRateState* rs, const SpxStreamState* st, const int64_t* tsm_n, const int16_t* tsm, int16_t* fin,
                     int64_t fin_cap, int channels, int old_rate, int new_rate, float rate, int bypass, int flush,
                     hipStream_t hs) {
  hipLaunchKernelGGL(spx_rate_kernel, dim3(1), dim3(256), 0, hs, rs, st, tsm_n, tsm, fin, fin_cap, channels, old_rate,
                     new_rate, rate, bypass, flush);
}
