// Micro-benchmark (not product): what one all-to-all hand-off of an activation row costs INSIDE a persistent launch on MI355X, in the form the
// decode engine uses (csrc/engine.hip): every workgroup (one per CU, 8 waves) publishes its share of the row as 8-byte {tag, value} granules
// (sc1 stores, no flag), one wave (or all eight) of every workgroup sweeps the whole row with sc1 loads until every tag equals the epoch,
// stages the values in LDS and the workgroup meets at a barrier.  Rounds are chained (round r + 1 publishes only after round r's gather), so
// time / rounds = one edge: last publish -> every CU ready.  Optionally waves 0..6 stream weights (non-temporal 16-B loads) in every round, to
// price the edge beside the CU's own weight stream.
//   hipcc --offload-arch=gfx950 -O3 tools/tune_handoff.hip -o tools/bin/tune_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void st_granule(u64* p, unsigned epoch, unsigned v) {
  __hip_atomic_store(p, ((u64)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_granule(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// N granules in all (N % gridDim.x == 0, N % 64 == 0); sweep_waves = 1 (wave 7 sweeps everything) or 8 (every wave sweeps N / 8);
// bg16 = 16-byte loads per lane per round of the streaming waves (0..6): bg16 KiB per wave per round
template <int BG>
__global__ __launch_bounds__(512) void handoff_kernel(u64* buf, int N, int rounds, int sweep_waves, const u32x4* stream, size_t stream_per_wg,
                                                       unsigned* err, float* sink) {
  constexpr int bg16 = BG, MAXK = BG > 0 ? BG : 1;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  unsigned* xs = (unsigned*)lds_raw;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per = N / gridDim.x;
  float acc = 0.f;
  const u32x4* sp = stream + (size_t)blockIdx.x * stream_per_wg + (size_t)wave * (stream_per_wg / 8) + lane;
  size_t soff = 0;
  const size_t swrap = stream_per_wg / 8 - 64 * (size_t)(bg16 > 0 ? bg16 : 1);
  for (int r = 0; r < rounds; ++r) {
    const unsigned epoch = (unsigned)r + 1u;
    u64* b = buf + (size_t)(r & 1) * N;
    // background stream: issue first, consume after the barrier (loads in flight across the hand-off, like a prefetching engine)
    u32x4 w[MAXK];
    if (wave < 7 && bg16 > 0) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < bg16) w[k] = __builtin_nontemporal_load(sp + soff + (size_t)k * 64);
      soff += (size_t)64 * bg16;
      if (soff >= swrap) soff = 0;
    }
    if (wave == 0 && lane < per) st_granule(b + (size_t)blockIdx.x * per + lane, epoch, (unsigned)(blockIdx.x * per + lane) ^ epoch);
    const bool sweeper = sweep_waves == 8 || wave == 7;
    if (sweeper) {
      const int n_mine = sweep_waves == 8 ? N / 8 : N;
      const int base = sweep_waves == 8 ? wave * n_mine : 0;
      unsigned spins = 0;
      for (int g0 = 0; g0 < n_mine; g0 += 64 * 16) {          // passes of 16 loads per lane (8 KiB of granules)
        for (;;) {
          bool ok = true;
          unsigned v[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int i = g0 + k * 64 + lane;
            const u64 x = i < n_mine ? ld_granule(b + base + i) : ((u64)epoch << 32);
            v[k] = (unsigned)x; ok &= (unsigned)(x >> 32) == epoch;
          }
          if (__all(ok)) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int i = g0 + k * 64 + lane; if (i < n_mine) xs[base + i] = v[k]; }
            break;
          }
          if (++spins > (1u << 22)) { if (lane == 0) atomicOr(err, 1u); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
    // consume: one LDS word + the streamed weights
    acc += (float)(xs[(lane * 7 + r) % N] & 0xff);
    if (wave < 7 && bg16 > 0) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < bg16) acc += (float)(w[k].x & 1u) + (float)(w[k].w & 1u);
    }
    __syncthreads();
  }
  if (acc == -1.f) sink[0] = acc;
}

// baseline: the same streaming with NO hand-off (what the CU's stream costs alone)
template <int BG>
__global__ __launch_bounds__(512) void stream_kernel(int rounds, const u32x4* stream, size_t stream_per_wg, float* sink) {
  constexpr int bg16 = BG, MAXK = BG > 0 ? BG : 1;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float acc = 0.f;
  const u32x4* sp = stream + (size_t)blockIdx.x * stream_per_wg + (size_t)wave * (stream_per_wg / 8) + lane;
  size_t soff = 0;
  const size_t swrap = stream_per_wg / 8 - 64 * (size_t)(bg16 > 0 ? bg16 : 1);
  for (int r = 0; r < rounds; ++r) {
    u32x4 w[MAXK];
    if (wave < 7) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < bg16) w[k] = __builtin_nontemporal_load(sp + soff + (size_t)k * 64);
      soff += (size_t)64 * bg16;
      if (soff >= swrap) soff = 0;
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < bg16) acc += (float)(w[k].x & 1u) + (float)(w[k].w & 1u);
    }
  }
  if (acc == -1.f) sink[0] = acc;
}

__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }

int main() {
  int n_cu = 0; CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  printf("CUs %d\n", n_cu);
  const int grid = n_cu;
  const int NMAX = 9472 * 2;
  u64* buf; CHECK(hipMalloc(&buf, (size_t)2 * NMAX * 8));
  unsigned* err; CHECK(hipMalloc(&err, 64)); CHECK(hipMemset(err, 0, 64));
  float* sink; CHECK(hipMalloc(&sink, 64));
  const size_t stream_per_wg = (size_t)8 * 1024 * 1024 / 16;      // 8 MiB per workgroup = 2 GiB in all
  u32x4* stream; CHECK(hipMalloc(&stream, stream_per_wg * 16 * grid));
  CHECK(hipMemset(stream, 1, stream_per_wg * 16 * grid));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t lds = 96 * 1024;
  typedef void (*kh_t)(u64*, int, int, int, const u32x4*, size_t, unsigned*, float*);
  typedef void (*ks_t)(int, const u32x4*, size_t, float*);
  auto pick = [&](int bg) -> kh_t { return bg == 0 ? handoff_kernel<0> : bg == 2 ? handoff_kernel<2> : bg == 8 ? handoff_kernel<8> : handoff_kernel<16>; };
  auto pick_s = [&](int bg) -> ks_t { return bg == 2 ? stream_kernel<2> : bg == 8 ? stream_kernel<8> : stream_kernel<16>; };
  for (int bg : {0, 2, 8, 16}) CHECK(hipFuncSetAttribute((const void*)pick(bg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto run = [&](int N, int rounds, int sw, int bg16) -> float {
    kh_t kh = pick(bg16);
    CHECK(hipMemset(buf, 0, (size_t)2 * NMAX * 8));
    hipLaunchKernelGGL(kh, dim3(grid), dim3(512), lds, 0, buf, N, 8, sw, stream, stream_per_wg, err, sink);      // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(buf, 0, (size_t)2 * NMAX * 8));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kh, dim3(grid), dim3(512), lds, 0, buf, N, rounds, sw, stream, stream_per_wg, err, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f;
  };
  // granule counts: 512 = a 64-CU group's q/k/v share as pairs x 256 CUs scaled; 1792 = a 3584-wide bf16 row as pairs; 9472 = the 18944-wide
  // activation row as pairs.  N must divide by the grid (256) and by 64 x 8.
  for (int N : {512, 1024, 2048, 9216}) {
    for (int sw : {1, 8}) {
      for (int bg : {0, 2, 8, 16}) {
        const float t1 = run(N, 50, sw, bg), t2 = run(N, 250, sw, bg);
        const float per = (t2 - t1) / 200.f;
        // the stream alone, same rounds
        float base = 0.f;
        if (bg) {
          ks_t ks = pick_s(bg);
          hipLaunchKernelGGL(ks, dim3(grid), dim3(512), 0, 0, 8, stream, stream_per_wg, sink);
          CHECK(hipDeviceSynchronize());
          float a, b;
          CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(ks, dim3(grid), dim3(512), 0, 0, 50, stream, stream_per_wg, sink); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&a, e0, e1));
          CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(ks, dim3(grid), dim3(512), 0, 0, 250, stream, stream_per_wg, sink); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&b, e0, e1));
          base = (b - a) * 1e3f / 200.f;
        }
        printf("N %5d granules (%5.1f KB)  sweep waves %d  stream %3d KiB/wave/round (%6.1f MB/round chip): %6.2f us per round   [stream alone %6.2f us; launch+50 rounds %7.1f us]\n",
               N, N * 8 / 1024.f, sw, bg, bg * 1024.f * 7 * grid / 1e6f, per, base, t1);
      }
    }
  }
  unsigned h_err = 0; CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
  printf("timeouts: %u\n", h_err);
  // a kernel boundary for comparison
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel boundary: %.2f us\n", ms * 1e3 / 500);
  return 0;
}
