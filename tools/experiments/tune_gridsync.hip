// Micro-benchmark (not product): cost of a grid-wide barrier inside one cooperative kernel on MI355X, against the cost of a kernel
// boundary (back-to-back dependent launches of an empty kernel).   hipcc --offload-arch=gfx950 -O3 tools/tune_gridsync.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
namespace cg = cooperative_groups;

__global__ void k_sync(int n, float* out) {
  cg::grid_group g = cg::this_grid();
  float v = threadIdx.x;
  for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.f; g.sync(); }
  if (v == -1.f) out[0] = v;
}

// hand-rolled barrier: one atomic per workgroup on a monotonically increasing counter, spin on an agent-scope load
__global__ void k_atomic(int n, unsigned* ctr, float* out) {
  float v = threadIdx.x;
  const unsigned nb = gridDim.x;
  for (int i = 0; i < n; ++i) {
    v = v * 1.0001f + 1.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(i + 1) * nb;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (v == -1.f) out[0] = v;
}

__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }

int main() {
  float* out; hipMalloc(&out, 64);
  unsigned* ctr; hipMalloc(&ctr, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int blocks : {64, 256, 512}) {
    for (int threads : {64, 256}) {
      int n = 200;
      void* args[] = {&n, &out};
      hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(threads), args, 0, 0);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipError_t e = hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(threads), args, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      printf("cg grid.sync   blocks %3d x %3d threads: %.2f us per sync (%s)\n", blocks, threads, ms * 1e3 / n, hipGetErrorString(e));
      hipMemset(ctr, 0, 4);
      hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(threads), 0, 0, n, ctr, out);
      hipDeviceSynchronize();
      hipMemset(ctr, 0, 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(threads), 0, 0, n, ctr, out);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      printf("atomic barrier blocks %3d x %3d threads: %.2f us per barrier\n", blocks, threads, ms * 1e3 / n);
    }
  }
  for (int blocks : {64, 256, 1024}) {
    int n = 500;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel, %4d blocks: %.2f us per back-to-back launch\n", blocks, ms * 1e3 / n);
  }
  return 0;
}
