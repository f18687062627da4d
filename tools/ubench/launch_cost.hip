// Host cost of a kernel launch on this platform: empty kernel, small and plan-sized (by-value struct) arguments, one
// stream; and the same five-kernel chain submitted as a hipGraph.  hipcc --offload-arch=gfx950 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[320]; };
__global__ void k0() {}
__global__ void k1(int* p, int n) { if (n < 0) *p = n; }
__global__ void k2(Big b, int* p, int n) { if (n < 0) *p = b.b[0]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; (void)hipMalloc(&d, 4);
  Big b = {};
  for (int rep = 0; rep < 2; rep++) {
    const int N = 2000;
    double t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k0, dim3(256), dim3(256), 0, s);
    double t1 = now(); (void)hipStreamSynchronize(s);
    double t2 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k1, dim3(256), dim3(256), 0, s, d, i);
    double t3 = now(); (void)hipStreamSynchronize(s);
    double t4 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k2, dim3(256), dim3(256), 36000, s, b, d, i);
    double t5 = now(); (void)hipStreamSynchronize(s);
    // launch 5 then sync, repeatedly (the pool's pattern)
    double t6 = now();
    for (int i = 0; i < 400; i++) { for (int j = 0; j < 5; j++) hipLaunchKernelGGL(k2, dim3(256), dim3(256), 36000, s, b, d, i); (void)hipStreamSynchronize(s); }
    double t7 = now();
    printf("per launch (queued back to back): no args %.2f us, 2 args %.2f us, 320-byte struct + LDS %.2f us; 5 launches + sync: %.1f us per round\n",
           (t1 - t0) / N, (t3 - t2) / N, (t5 - t4) / N, (t7 - t6) / 400);
  }
  // the same 5-kernel chain as a graph
  hipGraph_t g; (void)hipGraphCreate(&g, 0);
  hipGraphNode_t prev = nullptr;
  int n = 1;
  void* args[3] = {&b, &d, &n};
  for (int j = 0; j < 5; j++) {
    hipKernelNodeParams kp = {};
    kp.func = reinterpret_cast<void*>(k2); kp.gridDim = dim3(256); kp.blockDim = dim3(256); kp.sharedMemBytes = 36000; kp.kernelParams = args;
    hipGraphNode_t node;
    if (hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp) != hipSuccess) { printf("add node failed\n"); return 1; }
    prev = node;
  }
  hipGraphExec_t ge;
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (int i = 0; i < 400; i++) { (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s); }
    double t1 = now();
    printf("graph of the 5 kernels: launch + sync %.1f us per round\n", (t1 - t0) / 400);
  }
  return 0;
}
