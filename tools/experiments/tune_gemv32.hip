// Tuning harness (not product): skinny GEMM Y[b,N] = X[b,K] W[N,K]^T for b <= 32 on the decode shapes of Qwen2-7B, to decide the
// operand layouts of the decode GEMV.  Variants: x row-major (fragment-shaped 16 rows x 64 B loads) vs x PACKED in MFMA fragment
// order ([chunk][half][nb][lane][8]: every wave load is 1 KiB contiguous), W row-major vs W PACKED ([tile16][chunk][half][lane][8]),
// NTILE row tiles per workgroup, K chunks in flight per wave.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_gemv32.hip -o /tmp/tune_gemv32 && /tmp/tune_gemv32
#include "../../omchat_amd/csrc/common.h"
#include <cstdio>
#include <vector>
void omchat_set_error(const std::string& s) { fprintf(stderr, "ERR %s\n", s.c_str()); }

template <typename T, int NTILE, int NB, int WAVES, int UNROLL, bool XPACK, bool WPACK, bool NTL, bool XFAKE = false>
__global__ __launch_bounds__(WAVES * 64) void sk_kernel(const T* __restrict__ W, const T* __restrict__ X, float* __restrict__ Y, int N, int K, int b,
                                                        int ksplit) {
  typedef typename V8<T>::type frag_t;
  __shared__ float red[WAVES][NTILE * NB][256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int tile0 = blockIdx.x * NTILE;
  const int nchunk_all = K / 64;
  const int c_lo = (int)(((long)nchunk_all * blockIdx.y) / ksplit), c_hi = (int)(((long)nchunk_all * (blockIdx.y + 1)) / ksplit);
  f32x4 acc[NTILE][NB];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const T* wbase[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    if (WPACK) wbase[t] = W + (size_t)(tile0 + t) * nchunk_all * 1024 + lane * 8;
    else { int r = (tile0 + t) * 16 + fr; r = r < N ? r : N - 1; wbase[t] = W + (size_t)r * K + fg * 8; }
  }
  const T* xbase[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (XPACK) xbase[nb] = X + nb * 512 + lane * 8;
    else xbase[nb] = X + (size_t)(nb * 16 + fr < b ? nb * 16 + fr : 0) * K + fg * 8;
  }
  for (int c0 = c_lo + wave * UNROLL; c0 < c_hi; c0 += WAVES * UNROLL) {
    frag_t wf[UNROLL][NTILE][2], xf[UNROLL][NB][2];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int c = c0 + u < c_hi ? c0 + u : c_hi - 1;          // tail: recompute the last chunk (timing harness only)
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const T* p = WPACK ? wbase[t] + (size_t)c * 1024 + h * 512 : wbase[t] + c * 64 + h * 32;
          if (NTL) wf[u][t][h] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(p));
          else wf[u][t][h] = *reinterpret_cast<const frag_t*>(p);
        }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const T* p = XPACK ? xbase[nb] + (size_t)((XFAKE ? (c & 1) : c) * 2 + h) * NB * 512 : xbase[nb] + c * 64 + h * 32;      // XFAKE: x stays in L1 (bound without x traffic; results wrong)
          xf[u][nb][h] = *reinterpret_cast<const frag_t*>(p);
        }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          acc[t][nb] = mfma16(wf[u][t][0], xf[u][nb][0], acc[t][nb]);
          acc[t][nb] = mfma16(wf[u][t][1], xf[u][nb][1], acc[t][nb]);
        }
  }
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t * NB + nb][(fg * 4 + r) * 16 + fr] = acc[t][nb][r];
  __syncthreads();
  // all waves share the reduction: element e of tile slot ts
  for (int i = threadIdx.x; i < NTILE * NB * 256; i += WAVES * 64) {
    const int ts = i >> 8, e = i & 255;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += red[w][ts][e];
    const int t = ts / NB, nb = ts % NB, n = (tile0 + t) * 16 + (e >> 4), bi = nb * 16 + (e & 15);
    if (n < N && bi < b) Y[((size_t)blockIdx.y * b + bi) * N + n] = s;
  }
}

struct Shape { const char* name; int N, K, ks; };
// x-stationary persistent variant (K = 64 * WAVES * NCH): every wave keeps the x fragments of ITS chunks in registers for the whole launch,
// workgroups walk 16-row weight tiles (stride gridDim.x), the waves' partial tiles meet in LDS (parity double buffer, one barrier per
// tile), the next tile's weight fragments are loaded before the reduction of the current one.  No x re-reads at all.
template <typename T, int NB, int WAVES, int NCH>
__global__ __launch_bounds__(WAVES * 64) void xs_kernel(const T* __restrict__ W, const T* __restrict__ X, float* __restrict__ Y, int N, int K, int b) {
  typedef typename V8<T>::type frag_t;
  __shared__ float red[2][WAVES][NB][256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunk_all = K / 64, n_tiles = N / 16;
  frag_t xf[NCH][NB][2];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int h = 0; h < 2; ++h) xf[i][nb][h] = *reinterpret_cast<const frag_t*>(X + ((size_t)((wave + WAVES * i) * 2 + h) * NB + nb) * 512 + lane * 8);
  auto load_w = [&](frag_t (&wf)[NCH][2], int tile) {
    const T* base = W + (size_t)tile * nchunk_all * 1024 + lane * 8;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) wf[i][h] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(base + (size_t)(wave + WAVES * i) * 1024 + h * 512));
  };
  auto finish = [&](frag_t (&wf)[NCH][2], int tile, int par) {
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[nb] = mfma16(wf[i][0], xf[i][nb][0], acc[nb]);
        acc[nb] = mfma16(wf[i][1], xf[i][nb][1], acc[nb]);
      }
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[par][wave][nb][(fg * 4 + r) * 16 + fr] = acc[nb][r];
    __syncthreads();
    for (int i = threadIdx.x; i < NB * 256; i += WAVES * 64) {
      const int nb = i >> 8, e = i & 255;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) sum += red[par][w][nb][e];
      const int n = tile * 16 + (e >> 4), bi = nb * 16 + (e & 15);
      if (bi < b) Y[(size_t)bi * N + n] = sum;
    }
  };
  frag_t wa[NCH][2], wb[NCH][2];
  int tile = blockIdx.x;
  if (tile < n_tiles) load_w(wa, tile);
  for (; tile < n_tiles; tile += 2 * gridDim.x) {
    const int t2 = tile + gridDim.x, t3 = tile + 2 * gridDim.x;
    if (t2 < n_tiles) load_w(wb, t2);
    finish(wa, tile, 0);
    if (t2 < n_tiles) {
      if (t3 < n_tiles) load_w(wa, t3);
      finish(wb, t2, 1);
    }
  }
}

template <int NB, int WAVES, int NCH>
float run_xs(const Shape& sh, int b, const std::vector<void*>& W, void* X, void* Y, int iters, int grid) {
  auto k = xs_kernel<bf16, NB, WAVES, NCH>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 0, 0, (const bf16*)W[i % W.size()], (const bf16*)X, (float*)Y, sh.N, sh.K, b);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 0, 0, (const bf16*)W[i % W.size()], (const bf16*)X, (float*)Y, sh.N, sh.K, b);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("LAUNCH ERROR\n");
  return ms * 1e3f / iters;
}



template <int NTILE, int NB, int WAVES, int UNROLL, bool XPACK, bool WPACK, bool NTL, bool XFAKE = false>
float run(const Shape& sh, int b, const std::vector<void*>& W, void* X, void* Y, int iters) {
  dim3 grid((sh.N / 16 + NTILE - 1) / NTILE, sh.ks);
  auto k = sk_kernel<bf16, NTILE, NB, WAVES, UNROLL, XPACK, WPACK, NTL, XFAKE>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(WAVES * 64), 0, 0, (const bf16*)W[i % W.size()], (const bf16*)X, (float*)Y, sh.N, sh.K, b, sh.ks);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(WAVES * 64), 0, 0, (const bf16*)W[i % W.size()], (const bf16*)X, (float*)Y, sh.N, sh.K, b, sh.ks);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("LAUNCH ERROR\n");
  return ms * 1e3f / iters;
}

#define RUN(NT_, NB_, WV, UN, XP, WP, NTL_) do { float us = run<NT_, NB_, WV, UN, XP, WP, NTL_>(sh, b, W, X, Y, iters); \
  printf("  %-8s b=%-2d NTILE=%d NB=%d waves=%-2d unroll=%d xpack=%d wpack=%d nt=%d ks=%d : %7.1f us  %5.2f TB/s\n", sh.name, b, NT_, NB_, WV, UN, (int)XP, (int)WP, (int)NTL_, \
         sh.ks, us, bytes / us / 1e6); fflush(stdout); } while (0)

int main() {
  const int iters = 30;
  Shape shapes[] = {{"gateup", 37888, 3584, 1}, {"down", 3584, 18944, 8}, {"qkv", 4608, 3584, 1}, {"o", 3584, 3584, 3}, {"lm_head", 152064, 3584, 1}};
  for (auto& sh : shapes) {
    const double bytes = (double)sh.N * sh.K * 2;
    int nbuf = (int)(1.2e9 / bytes) + 2; nbuf = nbuf > 12 ? 12 : nbuf;
    std::vector<void*> W(nbuf);
    for (auto& w : W) { hipMalloc(&w, (size_t)bytes); hipMemset(w, 0x3c, (size_t)bytes); }
    void *X, *Y; hipMalloc(&X, (size_t)sh.K * 2 * 32); hipMemset(X, 0x3c, (size_t)sh.K * 2 * 32); hipMalloc(&Y, (size_t)sh.N * 4 * 32 * 8);
    printf("%s N=%d K=%d (%.1f MB, %zu buffers)\n", sh.name, sh.N, sh.K, bytes / 1e6, W.size());
    {
      const int b = 32;
      RUN(2, 2, 8, 4, false, false, false);      // what the product kernel does today
      RUN(2, 2, 8, 4, true, false, false);
      RUN(4, 2, 8, 2, true, false, false);
      RUN(2, 2, 8, 4, true, true, false);
      RUN(2, 2, 8, 4, true, true, true);
      RUN(4, 2, 8, 2, true, true, true);
      RUN(4, 2, 4, 2, true, true, true);
      RUN(4, 2, 4, 4, true, true, true);
      RUN(2, 2, 4, 4, true, true, true);
      RUN(2, 2, 4, 8, true, true, true);
      RUN(1, 2, 4, 8, true, true, true);
      RUN(1, 2, 8, 8, true, true, true);
      RUN(4, 2, 8, 4, true, true, true);
      RUN(8, 2, 8, 2, true, true, true);
      RUN(8, 2, 8, 1, true, true, true);
      RUN(8, 2, 4, 2, true, true, true);
      if (sh.K == 3584) {
        for (int grid : {256, 512, 296, 1184})
          { float us = run_xs<2, 8, 7>(sh, b, W, X, Y, iters, grid); printf("  %-8s b=32 x-stationary waves=8 nch=7 grid=%-4d: %7.1f us %5.2f TB/s\n", sh.name, grid, us, bytes / us / 1e6); }
      }
      { float us = run<2, 2, 8, 4, true, true, true, true>(sh, b, W, X, Y, iters); printf("  %-8s b=32 NTILE=2 waves=8 unroll=4 XFAKE (x from L1): %7.1f us %5.2f TB/s\n", sh.name, us, bytes / us / 1e6); }
      { float us = run<4, 2, 4, 2, true, true, true, true>(sh, b, W, X, Y, iters); printf("  %-8s b=32 NTILE=4 waves=4 unroll=2 XFAKE (x from L1): %7.1f us %5.2f TB/s\n", sh.name, us, bytes / us / 1e6); }
    }
    {
      const int b = 1;
      RUN(2, 1, 8, 4, false, false, false);
      RUN(2, 1, 8, 4, true, true, true);
      RUN(4, 1, 8, 4, true, true, true);
      RUN(4, 1, 4, 4, true, true, true);
      RUN(2, 1, 4, 8, true, true, true);
      RUN(1, 1, 4, 8, true, true, true);
      RUN(1, 1, 4, 16, true, true, true);
      RUN(2, 1, 8, 8, true, true, true);
    }
    for (auto& w : W) hipFree(w);
    hipFree(X); hipFree(Y);
  }
  return 0;
}
