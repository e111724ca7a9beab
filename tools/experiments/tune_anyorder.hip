// Tuning harness (not product): what does hipExtAnyOrderLaunch change on this GPU?
//  1. ORDER: kernel A spins (few waves, or several resident rounds of waves); kernel B records the clock at which its first wave starts.  In-order B starts
//     ~2.5 us after A's last wave ends.  Measured with the flag on B: ~0.3 us after -- B still does not overtake A.
//  2. VISIBILITY: A writes a buffer (each workgroup its slice), B reads the slice written by the workgroup 1 / 3 / 5 ids further (another XCD: workgroup ids
//     go round the 8 XCDs) and counts stale words, over many rounds with a new value each; plain stores + plain loads, and write-through stores (sc0 sc1,
//     drained before the wave ends) + cache-bypassing loads (sc0 sc1).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_anyorder.hip -o tools/bin/tune_anyorder && tools/bin/tune_anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void k_spin(unsigned long long* t_end, long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) atomicMax(t_end, wall_clock64());
}
__global__ void k_mark(unsigned long long* t_start) {
  if (threadIdx.x == 0) atomicMin(t_start, wall_clock64());
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int AUX_SYS = 17;        // sc0 | sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr int SLICE = 256 * 4;      // words per workgroup: 16 bytes per thread

template <bool WT>
__global__ __launch_bounds__(256) void k_write(unsigned* buf, unsigned value, int nwg) {
  const u32x4_t v = {value, value ^ 0x5a5a5a5au, value + threadIdx.x, value + blockIdx.x};
  if constexpr (WT) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_of(buf, (size_t)nwg * SLICE * 4), (blockIdx.x * SLICE + threadIdx.x * 4) * 4, 0, AUX_SYS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    *reinterpret_cast<u32x4_t*>(buf + blockIdx.x * SLICE + threadIdx.x * 4) = v;
  }
}
template <bool BYPASS>
__global__ __launch_bounds__(256) void k_read(const unsigned* buf, unsigned value, int nwg, int shift, unsigned* stale) {
  const int src = (blockIdx.x + shift) % nwg;
  u32x4_t v;
  if constexpr (BYPASS) v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(buf, (size_t)nwg * SLICE * 4), (src * SLICE + threadIdx.x * 4) * 4, 0, AUX_SYS);
  else v = *reinterpret_cast<const u32x4_t*>(buf + src * SLICE + threadIdx.x * 4);
  const bool ok = v[0] == value && v[1] == (value ^ 0x5a5a5a5au) && v[2] == value + threadIdx.x && v[3] == value + (unsigned)src;
  if (!ok) atomicAdd(stale, 1u);
}

__global__ void k_pub(unsigned* flag, unsigned epoch) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_poll(const unsigned* flag, unsigned epoch) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    for (int i = 0; i < 2000000; ++i) {
      const unsigned f = lane < 28 ? __hip_atomic_load(flag + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
      if (__all((int)(f - epoch) >= 0)) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
__global__ void k_body(unsigned* sink) {
  if (threadIdx.x == 0 && blockIdx.x == 0xffffff) sink[0] = 1u;
}

int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long* d; hipMalloc(&d, 16);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);     // kHz
  // ---- 1. order
  for (int shape = 0; shape < 2; ++shape) {
    const int wgs = shape == 0 ? 256 : 20000, threads = shape == 0 ? 64 : 256;
    const double spin_us = shape == 0 ? 200.0 : 20.0;
    const long long ticks = (long long)(spin_us * rate / 1000.0);
    for (int mode = 0; mode < 2; ++mode) {
      double lo = 1e30, hi = -1e30;
      for (int rep = 0; rep < 10; ++rep) {
        unsigned long long init[2] = {0ull, ~0ull};
        hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(threads), 0, s, d, ticks);
        if (mode == 0) hipLaunchKernelGGL(k_mark, dim3(256), dim3(64), 0, s, d + 1);
        else hipExtLaunchKernelGGL(k_mark, dim3(256), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d + 1);
        hipStreamSynchronize(s);
        unsigned long long out[2]; hipMemcpy(out, d, 16, hipMemcpyDeviceToHost);
        const double gap_us = ((double)out[1] - (double)out[0]) * 1000.0 / rate;
        lo = gap_us < lo ? gap_us : lo; hi = gap_us > hi ? gap_us : hi;
      }
      printf("order, A = %d x %d threads spinning %.0f us each, B %s: B's first wave starts %.2f .. %.2f us after A's last wave ends (10 runs; negative = overtook)\n",
             wgs, threads, spin_us, mode ? "any-order" : "in-order ", lo, hi);
    }
  }
  // ---- 2. visibility
  const int nwg = 4096;
  unsigned *buf, *stale; hipMalloc(&buf, (size_t)nwg * SLICE * 4); hipMalloc(&stale, 4);
  hipMemset(buf, 0, (size_t)nwg * SLICE * 4);
  for (int variant = 0; variant < 4; ++variant) {
    const bool wt = variant & 1, any = variant & 2;
    hipMemset(stale, 0, 4);
    const int rounds = 300;
    for (int r = 0; r < rounds; ++r) {
      const unsigned value = 0x1000u * (variant + 1) + r * 7u + 1u;
      const int shift = 1 + 2 * (r % 3);
      if (wt) hipLaunchKernelGGL(k_write<true>, dim3(nwg), dim3(256), 0, s, buf, value, nwg);
      else hipLaunchKernelGGL(k_write<false>, dim3(nwg), dim3(256), 0, s, buf, value, nwg);
      if (any) {
        if (wt) hipExtLaunchKernelGGL(k_read<true>, dim3(nwg), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const unsigned*)buf, value, nwg, shift, stale);
        else hipExtLaunchKernelGGL(k_read<false>, dim3(nwg), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const unsigned*)buf, value, nwg, shift, stale);
      } else {
        if (wt) hipLaunchKernelGGL(k_read<true>, dim3(nwg), dim3(256), 0, s, (const unsigned*)buf, value, nwg, shift, stale);
        else hipLaunchKernelGGL(k_read<false>, dim3(nwg), dim3(256), 0, s, (const unsigned*)buf, value, nwg, shift, stale);
      }
    }
    hipStreamSynchronize(s);
    unsigned n; hipMemcpy(&n, stale, 4, hipMemcpyDeviceToHost);
    printf("visibility, %s stores / %s loads, reader launched %s: %u stale 16-byte reads of %d (%d rounds)\n", wt ? "write-through (sc0 sc1, drained)" : "plain",
           wt ? "bypassing (sc0 sc1)" : "plain", any ? "any-order" : "in-order ", n, rounds * nwg * 256, rounds);
  }
  // ---- 3. chain cost: [A (28 workgroups, publishes a flag) ; B (512 x 448 threads, polls the flag when launched any-order) ; C (4736 x 256 threads)] x 200,
  //         all in-order against B any-order: what the packets around an out-of-order one cost
  {
    unsigned* flag; hipMalloc(&flag, 256); hipMemset(flag, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned epoch = 0;
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, s);
        for (int i = 0; i < 200; ++i) {
          ++epoch;
          hipLaunchKernelGGL(k_pub, dim3(28), dim3(128), 0, s, flag, epoch);
          if (mode == 0) hipLaunchKernelGGL(k_poll, dim3(512), dim3(448), 0, s, (const unsigned*)flag, epoch);
          else hipExtLaunchKernelGGL(k_poll, dim3(512), dim3(448), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const unsigned*)flag, epoch);
          if (mode == 2) hipExtLaunchKernelGGL(k_body, dim3(4736), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag + 32);
          else hipLaunchKernelGGL(k_body, dim3(4736), dim3(256), 0, s, flag + 32);
        }
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("chain A ; B ; C x 200: %s: %.2f us per chain\n", mode == 0 ? "all in-order              " : mode == 1 ? "B any-order (polls A's flag)" :
               "B and C any-order (C unordered: price only)", ms * 1000.0 / 200);
      }
    }
  }
  printf("hipGetLastError: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
