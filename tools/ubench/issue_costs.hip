// Micro-benchmark (developer tool, not product): what single instructions cost in the walk kernel's regime on gfx950 --
// one workgroup per CU, NW waves of which only the first NS execute the measured code while the rest wait at a barrier.
// Prints shader cycles per instruction for dependent / independent SALU and VALU chains, v_readlane -> SALU, LDS read
// latency, ds_add + wait, s_barrier, taken branches, exec save/restore, the DPP min reduction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_costs.hip -o /tmp/issue_costs && /tmp/issue_costs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

template <int TEST>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int ns, int* sink, int iters) {
  extern __shared__ unsigned lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < 4096; t += blockDim.x) lds[t] = t * 7 + 1;
  __syncthreads();
  unsigned long long acc = 0;
  int s = iters, v = lane;
  unsigned addr = (unsigned)(lane * 4);
  if (wave < ns) {
    for (int it = 0; it < iters; it++) {
      const unsigned long long t0 = now();
      if (TEST == 0) {  // dependent SALU chain
        asm volatile(REP64("s_add_i32 %0, %0, 1\n\t") : "+s"(s));
      } else if (TEST == 1) {  // independent SALU
        int a = s, b = s, c = s, d = s;
        asm volatile(REP8(REP8("s_add_i32 %0, %0, 1\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %2, %2, 1\n\ts_add_i32 %3, %3, 1\n\t")) : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
        s = a + b + c + d;
      } else if (TEST == 2) {  // dependent VALU chain
        asm volatile(REP64("v_add_u32 %0, %0, 1\n\t") : "+v"(v));
      } else if (TEST == 3) {  // independent VALU
        int a = v, b = v, c = v, d = v;
        asm volatile(REP8(REP8("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1\n\t")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        v = a + b + c + d;
      } else if (TEST == 4) {  // v_readlane -> SALU -> VALU round trip
        asm volatile(REP64("v_readlane_b32 %1, %0, 3\n\ts_add_i32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(v), "+s"(s));
      } else if (TEST == 5) {  // dependent LDS read chain (latency)
        unsigned a = addr;
        asm volatile(REP64("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xffc, %0\n\t") : "+v"(a));
        v += a;
      } else if (TEST == 6) {  // 8 independent LDS reads in flight then wait
        unsigned a0, a1, a2, a3, a4, a5, a6, a7;
        asm volatile(REP8("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %8 offset0:2 offset1:3\n\tds_read2_b32 %2, %8 offset0:4 offset1:5\n\tds_read2_b32 %3, %8 offset0:6 offset1:7\n\t"
                          "ds_read2_b32 %4, %8 offset0:8 offset1:9\n\tds_read2_b32 %5, %8 offset0:10 offset1:11\n\tds_read2_b32 %6, %8 offset0:12 offset1:13\n\tds_read2_b32 %7, %8 offset0:14 offset1:15\n\ts_waitcnt lgkmcnt(0)\n\t")
                     : "=&v"(*(unsigned long long*)&a0), "=&v"(*(unsigned long long*)&a1), "=&v"(*(unsigned long long*)&a2), "=&v"(*(unsigned long long*)&a3),
                       "=&v"(*(unsigned long long*)&a4), "=&v"(*(unsigned long long*)&a5), "=&v"(*(unsigned long long*)&a6), "=&v"(*(unsigned long long*)&a7)
                     : "v"(addr));
        v += a0 + a7;
      } else if (TEST == 15) {  // 8 independent 8-byte LDS reads in flight (the same bytes as test 6), then wait
        unsigned long long a0, a1, a2, a3, a4, a5, a6, a7;
        asm volatile(REP8("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                          "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\ts_waitcnt lgkmcnt(0)\n\t")
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7)
                     : "v"(addr * 2));
        v += (int)(a0 + a7);
      } else if (TEST == 7) {  // ds_add + wait
        asm volatile(REP64("ds_add_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : : "v"(addr), "v"(v) : "memory");
      } else if (TEST == 9) {  // taken branches
        asm volatile(REP64("s_branch 1f\n\ts_nop 0\n\t1:\n\t") : : : "memory");
      } else if (TEST == 10) {  // exec save/restore around one VALU
        asm volatile(REP64("s_and_saveexec_b64 s[40:41], vcc\n\tv_add_u32 %0, %0, 1\n\ts_or_b64 exec, exec, s[40:41]\n\t") : "+v"(v) : : "s40", "s41");
      } else if (TEST == 11) {  // DPP min reduction + readlane
        unsigned x = (unsigned)v;
        asm volatile(REP8("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                          "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                          "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                          "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                          "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                          "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                          "s_nop 1\n\tv_readlane_b32 %1, %0, 63\n\ts_nop 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(x), "+s"(s));
        v = (int)x;
      } else if (TEST == 12) {  // dependent fp64 fma chain
        double d = (double)v;
        asm volatile(REP64("v_fma_f64 %0, %0, %0, %0\n\t") : "+v"(d));
        v += (int)d;
      } else if (TEST == 13) {  // v_sad_u16 dependent chain
        asm volatile(REP64("v_sad_u16 %0, %1, %1, %0\n\t") : "+v"(v) : "v"(addr));
      } else if (TEST == 14) {  // SALU alternating with VALU, independent
        asm volatile(REP64("s_add_i32 %1, %1, 1\n\tv_add_u32 %0, %0, 1\n\t") : "+v"(v), "+s"(s));
      }
      acc += now() - t0;
    }
  }
  if (TEST == 8) {  // s_barrier with every wave of the workgroup taking part
    for (int it = 0; it < iters; it++) {
      const unsigned long long t0 = now();
      asm volatile(REP64("s_barrier\n\t") ::: "memory");
      acc += now() - t0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
  if (s == 12345 && v == 777) sink[0] = s + v;
}

#define RUN(T, name, per)                                                                                          \
  do {                                                                                                             \
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(64 * nw), 16384, 0, d_out, ns, d_sink, iters);                        \
    hipDeviceSynchronize();                                                                                        \
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);                                                         \
    printf("nw=%2d ns=%2d  %-44s %7.1f cycles each\n", nw, ns, name, (double)h[0] / iters / (per));                \
  } while (0)

int main(int argc, char** argv) {
  unsigned long long* d_out; int* d_sink; unsigned long long h[256];
  hipMalloc(&d_out, sizeof(h)); hipMalloc(&d_sink, 64);
  const int iters = 50;
  const int cfg[][2] = {{8, 4}, {8, 8}, {4, 4}, {1, 1}, {16, 16}};
  for (auto& c : cfg) {
    const int nw = c[0], ns = c[1];
    RUN(0, "dependent s_add_i32", 64.0);
    RUN(1, "independent s_add_i32 (4 chains)", 256.0);
    RUN(2, "dependent v_add_u32", 64.0);
    RUN(3, "independent v_add_u32 (4 chains)", 256.0);
    RUN(14, "s_add + v_add pair (independent)", 64.0);
    RUN(4, "v_readlane -> s_add -> v_add (dependent triple)", 64.0);
    RUN(5, "ds_read_b32 -> wait -> v_and (dependent)", 64.0);
    RUN(6, "8x ds_read2_b32 in flight + wait (per group)", 8.0);
    RUN(15, "8x ds_read_b64 in flight + wait (per group)", 8.0);
    RUN(7, "ds_add_u32 + wait", 64.0);
    RUN(8, "s_barrier (all waves)", 64.0);
    RUN(9, "taken s_branch + skipped s_nop", 64.0);
    RUN(10, "saveexec + v_add + restore", 64.0);
    RUN(11, "DPP min (6 stages) + readlane + v_add", 8.0);
    RUN(12, "dependent v_fma_f64", 64.0);
    RUN(13, "dependent v_sad_u16", 64.0);
  }
  return 0;
}
