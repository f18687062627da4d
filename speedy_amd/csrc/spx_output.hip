// Deferred output (round 6; spx_internal.h "Deferred output"): every sample the walk forms without output waves used to produce on
// their search waves -- a11: libsonic's overlapAdd cross-fades and its copies; call sites soniclib.c:369,551 --
// is produced HERE, from the 16-byte records those kernels now write: a wave per record, a lane per sample, no dependence between
// any two samples.  On the chain a step's cross-fade cost its reciprocal, two LDS reads, the quotient and a store in every search
// wave (a ninth of the lean form's 1.65 ms per 256 x 10 s); here the same arithmetic (xfade_rcp / xfade_num / xfade_quot of
// spx_walk_common.h: bit-identical by construction) runs at the chip's width, reading the two ramps from the input in HBM (L2:
// the walk kernel read those lines moments ago).  Algorithmic bytes: the output written once (2 C bytes per produced frame) and
// up to two input reads per cross-faded frame; a 256 x 10 s batch: 27 MB written, <= 54 MB read, 5 MB of records.
#include "spx_walk_common.h"

// A wave takes 64 consecutive records -- one 16-byte load per lane -- and expands them four at a time: the fields of a record come
// out of the lanes' registers (v_readlane), the input loads of four records are in flight together, and nothing a wave does waits
// for a load of the record before.  (First version: a record per wave and turn, its fields loaded, then its samples, then stored --
// three dependent round trips per record, 81 records per wave: 87 us per 256 x 10 s batch, on the walk stream.)  A step's
// cross-fade is n = period / (speed - 1) frames: at most 64 from speed 4.8 up whatever the period, 30 - 100 at 3.5x -- one pass of
// the wave for most records, a loop for the rest.
template <bool MCH>
__device__ __forceinline__ void expand_record(const int16_t* __restrict__ in, int16_t* __restrict__ out, int C, int lane, pos_t out_cap,
                                              pos_t limit, int o, int src, int n, int period) {
  const pos_t room = out_cap - o;
  const int nv = room > n ? n : (room < 0 ? 0 : (int)room);
  if (period != 0) {
    // cross-fade (libsonic overlapAdd): out[t] = (down[t] (n - t) + up[t] t) / n, integer, truncating toward zero
    if (n <= 0) return;
    const double inv = xfade_rcp(n);
    if (!MCH || C == 1) {
      const int16_t* __restrict__ rd = in + src;
      const int16_t* __restrict__ ru = rd + period;
      const int realD = (int)(limit - src), realU = realD - period;
      for (int t = lane; t < nv; t += 64) {
        const int d = (t < realD) ? (int)rd[t] : 0, u = (t < realU) ? (int)ru[t] : 0;
        out[(size_t)o + t] = (int16_t)xfade_quot(d * (n - t) + u * t, inv);
      }
    } else {
      const int16_t* __restrict__ rd = in + (size_t)src * C;
      const int16_t* __restrict__ ru = in + (size_t)(src + period) * C;
      int16_t* __restrict__ dst = out + (size_t)o * C;
      const int total = nv * C;
      const unsigned invC = (0x10000u + (unsigned)C - 1u) / (unsigned)C;  // e / C for e < 8192, C <= 8
      const int realD = (int)((limit - src) * C), realU = (int)((limit - src - period) * C);
      for (int e = lane; e < total; e += 64) {
        const int t = (C == 2) ? (e >> 1) : (int)(((unsigned)e * invC) >> 16);
        const int d = (e < realD) ? (int)rd[e] : 0, u = (e < realU) ? (int)ru[e] : 0;
        dst[e] = (int16_t)xfade_quot(d * (n - t) + u * t, inv);
      }
    }
  } else {
    // copy: n frames from TSM position src
    int16_t* __restrict__ dst = out + (size_t)o * C;
    const int16_t* __restrict__ sp = in + (size_t)src * C;
    const pos_t real = (limit - src) * C;
    const int total = nv * C;
    for (int e = lane; e < total; e += 64) dst[e] = (e < real) ? sp[e] : (int16_t)0;
  }
}

template <bool MCH>
__global__ void __launch_bounds__(256)
spx_output_kernel(const SpxStreamDev* __restrict__ streams, const int16_t* __restrict__ in_base, int16_t* __restrict__ out_base,
                  const SpxOutRec* __restrict__ recs, const int* __restrict__ counts, int parts) {
  const int sidx = (int)blockIdx.x / parts, part = (int)blockIdx.x - sidx * parts;
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)threadIdx.x >> 6);
  const SpxStreamDev& S = streams[sidx];
  const int C = MCH ? S.channels : 1;
  const int16_t* __restrict__ in = in_base + S.in_off - S.tsm_shift * C;   // indexed by (TSM position) * C + channel
  int16_t* __restrict__ out = out_base + S.out_off;
  const int nrec = uni(counts[2 * sidx]);
  const pos_t flushLim = uni(counts[2 * sidx + 1]);
  const pos_t writtenLim = (pos_t)(S.n_in + S.tsm_shift);
  const pos_t out_cap = (pos_t)(S.out_cap > 0x7fffffff ? 0x7fffffff : S.out_cap);
  const int4* __restrict__ R = reinterpret_cast<const int4*>(recs + S.rec_off);
  for (int r0 = 64 * (part * 4 + wave); r0 < nrec; r0 += 256 * parts) {
    const int have = nrec - r0 < 64 ? nrec - r0 : 64;
    const int4 mine = lane < have ? R[r0 + lane] : make_int4(0, 0, 0, 0);
    for (int k = 0; k < have; k += 4) {
      int o[4], src[4], n[4], period[4];
      pos_t lim[4];
      bool simple = true;   // four mono cross-fades of at most 64 frames inside the output: the common case, loads first
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int kk = k + u < have ? k + u : have - 1;
        o[u] = __builtin_amdgcn_readlane(mine.x, kk);
        src[u] = __builtin_amdgcn_readlane(mine.y, kk);
        const int nf = __builtin_amdgcn_readlane(mine.z, kk);
        period[u] = __builtin_amdgcn_readlane(mine.w, kk);
        n[u] = k + u < have ? (nf & (SPX_REC_FLUSH_LIMIT - 1)) : 0;   // (past the end: an empty record)
        lim[u] = (nf & SPX_REC_FLUSH_LIMIT) ? flushLim : writtenLim;   // input from here on reads as the flush's zero padding
        simple = simple && (n[u] == 0 || (period[u] != 0 && n[u] <= 64 && o[u] + n[u] <= out_cap));
      }
      if (simple && (!MCH || C == 1)) {
        int d[4], uu[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int realD = (int)(lim[u] - src[u]), realU = realD - period[u];
          const int16_t* __restrict__ rd = in + src[u];
          d[u] = (lane < n[u] && lane < realD) ? (int)rd[lane] : 0;
          uu[u] = (lane < n[u] && lane < realU) ? (int)rd[period[u] + lane] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (n[u] > 0) {   // uniform
            const double inv = xfade_rcp(n[u]);
            if (lane < n[u]) out[(size_t)o[u] + lane] = (int16_t)xfade_quot(d[u] * (n[u] - lane) + uu[u] * lane, inv);
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (k + u < have) expand_record<MCH>(in, out, C, lane, out_cap, lim[u], o[u], src[u], n[u], period[u]);
      }
    }
  }
}

void spx_launch_outputs(const SpxStreamDev* streams, int n_streams, int max_channels, const int16_t* in, int16_t* out,
                        const SpxOutRec* recs, const int* counts, hipStream_t st) {
  if (n_streams <= 0) return;
  // a 10 s stream writes about 1 300 records = 21 waves' worth; enough workgroups that the chip is full whatever the batch
  const int parts = n_streams <= 1024 ? 4 : (n_streams <= 4096 ? 2 : 1);
  if (max_channels > 1)
    hipLaunchKernelGGL(spx_output_kernel<true>, dim3(n_streams * parts), dim3(256), 0, st, streams, in, out, recs, counts, parts);
  else
    hipLaunchKernelGGL(spx_output_kernel<false>, dim3(n_streams * parts), dim3(256), 0, st, streams, in, out, recs, counts, parts);
}
