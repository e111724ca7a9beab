// Prototype (not product): b = 1 GEMV as whole-row streaming + v_dot2 (bf16) with x held in registers.
// Each wave owns R rows at a time; lanes read consecutive 16-B chunks of a row (1 KiB per wave instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[i]), __builtin_bit_cast(bf16x2, x[i]), acc, false);
  return acc;
}

// K = NCH * 512 exactly (3584 = 7 * 512).  grid-stride over row groups.
template <int R, int NCH, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void rowdot_kernel(const unsigned short* W, const unsigned short* X, float* Y, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = NCH * 512;
  u32x4 xr[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) xr[c] = *(const u32x4*)(X + c * 512 + lane * 8);
  const int ngroups = N / R;
  for (int g = blockIdx.x * WAVES + wave; g < ngroups; g += gridDim.x * WAVES) {
    const unsigned short* base = W + (size_t)g * R * K + lane * 8;
    u32x4 w[R][NCH];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const u32x4* p = (const u32x4*)(base + (size_t)r * K + c * 512);
        w[r][c] = NT ? __builtin_nontemporal_load(p) : *p;
      }
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[r] = dot8(w[r][c], xr[c], acc[r]);
    }
    // butterfly: R values over 64 lanes
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) Y[g * R + r] = acc[r];
    }
  }
}

int main() {
  const int N = 37888, K = 3584, iters = 40;
  const double bytes = (double)N * K * 2;
  std::vector<void*> W(6);
  for (auto& w : W) { hipMalloc(&w, (size_t)bytes); hipMemset(w, 0x3c, (size_t)bytes); }
  void* X; hipMalloc(&X, K * 2); hipMemset(X, 0x3c, K * 2);
  float* Y; hipMalloc(&Y, N * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch(W[i % W.size()]);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch(W[i % W.size()]);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.1f us  %5.2f TB/s\n", name, ms * 1e3 / iters, bytes / (ms * 1e3 / iters) / 1e6);
  };
#define L(KERN, GRID, BLK) [&](void* w) { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(BLK), 0, 0, (const unsigned short*)w, (const unsigned short*)X, Y, N); }
  time("rowdot R=2 waves=4 grid=2048 nt", L((rowdot_kernel<2, 7, 4, true>), 2048, 256));
  time("rowdot R=2 waves=4 grid=2048 plain", L((rowdot_kernel<2, 7, 4, false>), 2048, 256));
  time("rowdot R=4 waves=4 grid=2048 nt", L((rowdot_kernel<4, 7, 4, true>), 2048, 256));
  time("rowdot R=4 waves=4 grid=1024 nt", L((rowdot_kernel<4, 7, 4, true>), 1024, 256));
  time("rowdot R=4 waves=8 grid=1024 nt", L((rowdot_kernel<4, 7, 8, true>), 1024, 512));
  time("rowdot R=4 waves=4 grid=4736 nt", L((rowdot_kernel<4, 7, 4, true>), 4736, 256));
  time("rowdot R=8 waves=4 grid=1184 nt", L((rowdot_kernel<8, 7, 4, true>), 1184, 256));
  time("rowdot R=8 waves=4 grid=1024 nt", L((rowdot_kernel<8, 7, 4, true>), 1024, 256));
  time("rowdot R=4 waves=4 grid=2048 plain", L((rowdot_kernel<4, 7, 4, false>), 2048, 256));
  time("rowdot R=1 waves=8 grid=2048 nt", L((rowdot_kernel<1, 7, 8, true>), 2048, 512));
  return 0;
}
