// Micro-benchmark (diagnostic, not part of the product): cost of an "all lags at full rate" AMDF phase on one CU.
// One workgroup of 512 threads per CU; LDS holds 4 copies of a 4096-sample u16 window, copy k shifted by k samples, so that
// any run of 4 samples is one aligned ds_read_b64.  Task of a lane: (lag p, pairs [j0, j0+n)), n <= PPL, dealt so that all
// lags 40..246 are covered (16 kHz).  Per "step": every lane sums |s[o+i] - s[o+i+p]| over its range with v_sad_u16,
// ds_add_u32 into the lag's sum, workgroup barrier.  Prints cycles per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define WCAP 4096
#define CSTR (WCAP * 2 + 64)   // bytes per copy: == 64 mod 256 -> consecutive lags land on distinct bank pairs
struct Task { int p, j0, n; };
__global__ void __launch_bounds__(512) k(const Task* tasks, int steps, int ppl, unsigned long long* cyc, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned* sums = reinterpret_cast<unsigned*>(lds);            // 512 entries
  unsigned char* win = lds + 2048;
  const int tid = threadIdx.x;
  for (int i = tid; i < WCAP; i += 512)
    for (int c = 0; c < 4; c++) if (i >= c) *reinterpret_cast<unsigned short*>(win + c * CSTR + 2 * (i - c)) = (unsigned short)((i * 2654435761u) >> 17);
  sums[tid] = 0;
  const Task T = tasks[tid];
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned acc_all = 0;
  int o = 7;
  for (int s = 0; s < steps; s++) {
    const int ea = o + 2 * T.j0, eb = ea + T.p;
    const uint2* ap = reinterpret_cast<const uint2*>(win + (ea & 3) * CSTR + 2 * (ea & ~3));
    const uint2* bp = reinterpret_cast<const uint2*>(win + (eb & 3) * CSTR + 2 * (eb & ~3));
    unsigned d = 0;
    for (int j = 0; j < ppl; j += 16) {   // flights of 8 b64 reads per operand (16 pairs)
      uint2 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { a[u] = ap[j / 2 + u]; b[u] = bp[j / 2 + u]; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const bool on0 = j + 2 * u < T.n, on1 = j + 2 * u + 1 < T.n;
        d = __builtin_amdgcn_sad_u16(a[u].x, on0 ? b[u].x : a[u].x, d);
        d = __builtin_amdgcn_sad_u16(a[u].y, on1 ? b[u].y : a[u].y, d);
      }
    }
    atomicAdd(&sums[T.p], d);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    acc_all += sums[(tid * 7) & 255];
    o = 7 + ((o * 13 + (acc_all & 3)) & 1023);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + tid] = acc_all;
}
int main(int argc, char** argv) {
  const int minP = 40, maxP = 246;
  int ppl = argc > 1 ? atoi(argv[1]) : 38;
  std::vector<Task> t;
  // chunk-major order: chunk c of every lag that has it, lags ascending
  for (int c = 0;; c++) {
    bool any = false;
    for (int p = minP; p <= maxP; p++) {
      const int np = (p + 1) / 2;
      if (c * ppl < np) { any = true; t.push_back({p, c * ppl, np - c * ppl < ppl ? np - c * ppl : ppl}); }
    }
    if (!any) break;
  }
  printf("ppl %d tasks %zu\n", ppl, t.size());
  if (t.size() > 512) { printf("too many tasks\n"); return 1; }
  while (t.size() < 512) t.push_back({minP, 0, 0});
  Task* dt; unsigned long long* dc; unsigned* ds;
  hipMalloc(&dt, sizeof(Task) * 512); hipMalloc(&dc, 8 * 256); hipMalloc(&ds, 4 * 512 * 256);
  hipMemcpy(dt, t.data(), sizeof(Task) * 512, hipMemcpyHostToDevice);
  const int steps = 2000;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 2048 + 4 * CSTR + 1024, 0, dt, steps, ppl, dc, ds);
    hipDeviceSynchronize();
  }
  unsigned long long h[256]; hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; i++) s += h[i];
  printf("cycles per step (two barriers included): %.0f\n", s / 256 / steps);
  return 0;
}
