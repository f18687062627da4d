// Diagnostics and small services of the C-ABI (include/speedy_hip.h): the names and resources of the kernels a batch is served by,
// the process-wide switches, timing collection, output packing, copies.
#include "spx_engine.h"

extern "C" {
// Names of the kernels a batch of this shape is served by, as a profiler prints them (without "void" and the argument
// list): "analysis;tension;walk".  bench.py keys its roofline object and profiles/pmc_traffic.json with them.
static const char* kernel_names(spx_plan_t plan, int n_streams, int max_channels, int speedup_only, bool lean);
const char* spx_batch_kernel_names(spx_plan_t plan, int n_streams, int max_channels, int speedup_only) {
  return kernel_names(plan, n_streams, max_channels, speedup_only, false);
}
// ... with the walk kernel in its lean form (no output waves): what spx_batch_run_overlapped launches when three or more
// workspaces take turns, and the concurrent mode at 22.05 kHz mono
const char* spx_batch_kernel_names_lean(spx_plan_t plan, int n_streams, int max_channels, int speedup_only) {
  return kernel_names(plan, n_streams, max_channels, speedup_only, true);
}
static const char* kernel_names(spx_plan_t plan, int n_streams, int max_channels, int speedup_only, bool lean) {
  static thread_local char buf[256];
  const SpxPlanDev& d = plan->dev;
  const SpxWalkConfig c = spx_walk_config(d, n_streams, max_channels < 1 ? 1 : max_channels, speedup_only != 0, false, lean, speedup_only == 0);
  char walk[96];
  if (c.fast_kernel && c.slow)
    snprintf(walk, sizeof(walk), "spx_walk_fast_kernel<%d, %d, 0, 0, %d>", c.nwm, c.nwc >= 4 ? 4 : 0, max_channels > 1 ? 3 : 2);
  else if (c.fast_kernel)
    {
      const bool ct_rate = d.rate == 16000 || d.rate == 22050;
      const bool lng = ct_rate && c.nwm == 4 && c.nwc >= 4 && c.wcap == 8192;   // spx_launch_walk_fast's long-window instantiations
      snprintf(walk, sizeof(walk), "spx_walk_fast_kernel<%d, %d, %d, %d, %d>", c.nwm, c.nwc >= 4 ? 4 : (c.nwc >= 2 ? 2 : (c.nwc >= 1 ? 1 : 0)),
               (ct_rate && (lng || c.wcap == ((c.nwc == 0 && c.nwm <= 2) ? 1536 : 4096))) ? d.rate : 0, lng ? 1 : 0, max_channels > 1 ? 1 : 0);
    }
  else
    snprintf(walk, sizeof(walk), "spx_walk_kernel<%d, %d>", c.nw, c.mode);
  snprintf(buf, sizeof(buf), "spx_analysis_kernel<%d, %d>;spx_tension_kernel;%s", d.tile_frames, spx_analysis_ct_window(d), walk);
  return buf;
}

int spx_debug_last_call_concurrent(void) { return g_last_concurrent.load(std::memory_order_relaxed); }
int spx_debug_kernel_vgprs(int which) {
  if (which == 0) return spx_tension_vgprs();
  const int rate = (which == 2 || which == 4) ? 22050 : 16000;
  const SpxPlanDev* P = spx_internal_shared_plan(rate, 0);
  if (!P) return -1;
  switch (which) {
    case 1: return spx_walk_vgprs(*P, 256, 1, true, false);
    case 2: return spx_walk_vgprs(*P, 256, 1, true, true);
    case 3: case 4: return spx_analysis_vgprs(*P);
    case 5: return spx_walk_vgprs(*P, 256, 2, true, false);
    default: return -1;
  }
}

// Diagnostics: what spx_launch_walk would launch for a batch of this shape, and what that kernel costs -- out[0] allocated
// VGPRs (rounded up to the granule of 8), out[1] scratch bytes per lane (spilled registers), out[2] LDS bytes per workgroup,
// out[3] the form (16 * search waves + output waves; 0 = the general kernel), out[4] waves per workgroup.
int spx_debug_walk_info(int sample_rate, int channels, int n_streams, int speedup_only, int short_jobs, int lean, int* out) {
  const SpxPlanDev* P = spx_internal_shared_plan(sample_rate, 0);
  if (!P || !out) return -1;
  const SpxWalkConfig c = spx_walk_config(*P, n_streams, channels < 1 ? 1 : channels, speedup_only != 0, short_jobs != 0, lean != 0, speedup_only == 0);
  int scratch = -1;
  out[0] = spx_walk_kernel_regs(*P, n_streams, channels, speedup_only != 0, short_jobs != 0, lean != 0, &scratch, speedup_only == 0);
  out[1] = scratch;
  out[2] = (int)c.lds;
  out[3] = c.fast_kernel ? 16 * c.nwm + c.nwc : 0;
  out[4] = c.waves;
  return 0;
}
// ... and the same for the analysis kernel of a rate (out[0] VGPRs, out[1] scratch bytes, out[2] LDS bytes) and the tension kernel
int spx_debug_analysis_info(int sample_rate, int* out) {
  const SpxPlanDev* P = spx_internal_shared_plan(sample_rate, 0);
  if (!P || !out) return -1;
  int scratch = -1;
  out[0] = spx_analysis_vgprs(*P, &scratch);
  out[1] = scratch;
  out[2] = (int)spx_analysis_lds_bytes(*P);
  return 0;
}

void spx_set_timing(int enabled) { g_timing = enabled != 0; }
void spx_set_concurrent(int on) { g_concurrent = on != 0; }
void spx_set_pipeline_chunks(int chunks) { g_chunks_set = true; g_chunks = chunks < 1 ? 1 : (chunks > SPX_MAX_CHUNKS ? SPX_MAX_CHUNKS : chunks); }
static double g_last_tension_ms = 0.0;
double spx_timing_last_tension_ms(void) { return g_last_tension_ms; }
int spx_timing_collect(double* sum_ms_analyze, double* sum_ms_walk, int* n_calls) {
  std::lock_guard<std::mutex> g(g_tmu);
  double a = 0, w = 0, t = 0;
  for (auto& ev : g_ev_pending) {
    HIPCHK(hipEventSynchronize(ev.b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (ev.kind == 0) a += ms; else if (ev.kind == 1) w += ms; else t += ms;
    g_ev_free.push_back(ev.a);
    g_ev_free.push_back(ev.b);
  }
  g_ev_pending.clear();
  g_last_tension_ms = t;
  if (sum_ms_analyze) *sum_ms_analyze = a;
  if (sum_ms_walk) *sum_ms_walk = w;
  if (n_calls) *n_calls = g_calls_pending;
  g_calls_pending = 0;
  return 0;
}

void* spx_device_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(-2, "spx_device_alloc failed"); return nullptr; }
  return p;
}
void spx_device_free(void* p) { if (p) (void)hipFree(p); }
int spx_copy_to_device(void* dst, const void* src, size_t bytes, void* hs) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(hs)));
  return 0;
}
}  // extern "C"

// ---- output packing: offsets by one workgroup (serial prefix over <= a few thousand streams), then one workgroup per
// stream copying its frames with coalesced loads/stores ----
__global__ void __launch_bounds__(256)
spx_pack_offsets_kernel(const int64_t* __restrict__ n_out, const int* __restrict__ channels,
                        const int64_t* __restrict__ caps, int n, int64_t* __restrict__ offsets) {
  // exclusive prefix sum of the streams' element counts: 256 streams per pass, Hillis-Steele scan in LDS, running carry
  __shared__ int64_t sh[256];
  __shared__ int64_t carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + t;
    int64_t v = 0;
    if (i < n) {
      const int64_t k = n_out[i];
      // a negative count flags an overflowed stream: the frames that fitted its capacity are there, the count says how
      // many there would have been; INT64_MIN a lost producer (nothing)
      int64_t f = (k == INT64_MIN ? 0 : (k > 0 ? k : -k));
      if (f > caps[i]) f = caps[i];
      v = f * channels[i];
    }
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int64_t add = (t >= d) ? sh[t - d] : 0;
      __syncthreads();
      sh[t] += add;
      __syncthreads();
    }
    const int64_t base = carry;
    if (i < n) offsets[i] = base + sh[t] - v;
    __syncthreads();
    if (t == 255) carry = base + sh[255];
    __syncthreads();
  }
  if (t == 0) offsets[n] = carry;
}
__global__ void __launch_bounds__(256)
spx_pack_copy_kernel(const int16_t* __restrict__ out, const int64_t* __restrict__ out_offs,
                     const int64_t* __restrict__ offsets, int16_t* __restrict__ packed) {
  // 4 workgroups per stream (blockIdx.y), eight 2-byte loads in flight per thread: the copy is latency-bound otherwise
  // (one load per thread at a time took 0.3 ms for the bench batch's 26.7 MB)
  const int i = blockIdx.x;
  const int16_t* src = out + out_offs[i];
  int16_t* dst = packed + offsets[i];
  const int64_t cnt = offsets[i + 1] - offsets[i];
  const int64_t stride = (int64_t)gridDim.y * 256 * 8;
  for (int64_t e0 = ((int64_t)blockIdx.y * 256 + threadIdx.x); e0 < cnt; e0 += stride) {
    int16_t v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int64_t e = e0 + (int64_t)u * gridDim.y * 256; v[u] = (e < cnt) ? src[e] : (int16_t)0; }
#pragma unroll
    for (int u = 0; u < 8; u++) { const int64_t e = e0 + (int64_t)u * gridDim.y * 256; if (e < cnt) dst[e] = v[u]; }
  }
}

extern "C" {
int spx_batch_pack_outputs(const spx_stream_job* jobs, int n, const int16_t* out, const int64_t* n_out, int16_t* packed,
                           int64_t* offsets, void* hs) {
  if (!jobs || n <= 0 || !out || !n_out || !packed || !offsets) return fail(-1, "spx_batch_pack_outputs: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(hs);
  // small per-stream tables (channels, output offsets): a stream-ordered allocation, freed in stream order after the
  // kernels that read it, so concurrent calls on other streams never share it
  const size_t need = (size_t)n * (2 * sizeof(int64_t) + sizeof(int));
  void* d_tab = nullptr;
  if (hipMallocAsync(&d_tab, need, st) != hipSuccess) return fail(-2, "spx_batch_pack_outputs: allocation failed");
  // host side of the table: a per-thread pinned slot, reused once the copy that last read it has retired
  static thread_local SpxStage G;
  if (G.done) HIPCHK(hipEventSynchronize(G.done));
  else HIPCHK(hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
  if (G.cap < need) {
    if (G.p) (void)hipHostFree(G.p);
    G.p = nullptr; G.cap = 0;
    HIPCHK(hipHostMalloc(&G.p, need * 2 + 1024, hipHostMallocDefault));
    G.cap = need * 2 + 1024;
  }
  unsigned char* h = static_cast<unsigned char*>(G.p);
  int64_t* h_off = reinterpret_cast<int64_t*>(h);
  int64_t* h_cap = h_off + n;
  int* h_ch = reinterpret_cast<int*>(h + (size_t)n * 2 * sizeof(int64_t));
  for (int i = 0; i < n; i++) { h_off[i] = jobs[i].out_off; h_cap[i] = jobs[i].out_cap; h_ch[i] = jobs[i].channels; }
  HIPCHK(hipMemcpyAsync(d_tab, h, need, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(G.done, st));
  const int64_t* d_off = reinterpret_cast<const int64_t*>(d_tab);
  const int* d_ch = reinterpret_cast<const int*>(static_cast<unsigned char*>(d_tab) + (size_t)n * 2 * sizeof(int64_t));
  hipLaunchKernelGGL(spx_pack_offsets_kernel, dim3(1), dim3(256), 0, st, n_out, d_ch, d_off + n, n, offsets);
  hipLaunchKernelGGL(spx_pack_copy_kernel, dim3(n, 4), dim3(256), 0, st, out, d_off, offsets, packed);
  (void)hipFreeAsync(d_tab, st);
  HIPCHK(hipGetLastError());
  return 0;
}

int spx_copy_to_host(void* dst, const void* src, size_t bytes, void* hs) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(hs)));
  return 0;
}
int spx_stream_synchronize(void* hs) {
  HIPCHK(hipStreamSynchronize(static_cast<hipStream_t>(hs)));
  return 0;
}

}  // extern "C"
