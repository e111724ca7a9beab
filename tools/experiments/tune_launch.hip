// Tuning harness (not product): cost of one dependent kernel boundary on this GPU, eager stream vs captured hipGraph, for an empty
// kernel, a one-workgroup kernel that touches memory, and a 1184-workgroup kernel.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_launch.hip -o /tmp/tune_launch && /tmp/tune_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ __launch_bounds__(256) void k_chain(const float* in, float* out, int n) {      // dependent: out = f(in), one workgroup
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += in[i];
  out[threadIdx.x] = s;
}

template <typename F>
static void measure(const char* name, int n, F launch) {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) launch(s, i);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  hipEventRecord(e0, s);
  for (int i = 0; i < n; ++i) launch(s, i);
  hipEventRecord(e1, s);
  auto t1 = std::chrono::steady_clock::now();
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
  // graph
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(s, i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float gms; hipEventElapsedTime(&gms, e0, e1);
  printf("%-28s eager %6.2f us/kernel (host enqueue %5.2f us)   graph %6.2f us/kernel\n", name, ms * 1e3 / n, host_us, gms * 1e3 / (3 * n));
  fflush(stdout);
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
}

int main() {
  float *a, *b; hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMemset(a, 0, 1 << 20); hipMemset(b, 0, 1 << 20);
  const int n = 1000;
  measure("empty <<<1,64>>>", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
  measure("empty <<<1184,256>>>", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1184), dim3(256), 0, s); });
  measure("touch <<<1,64>>>", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, a); });
  measure("touch <<<1184,256>>>", n, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_touch, dim3(1184), dim3(256), 0, s, a); });
  measure("chain 3584 floats <<<1,256>>>", n, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, 3584); });
  if (hipGetLastError() != hipSuccess) printf("HIP ERROR\n");
  return 0;
}
