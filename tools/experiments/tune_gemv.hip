// Tuning harness (not product): sweeps weight-streaming GEMV variants on the decode shapes of Qwen2-7B.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_gemv.hip -o /tmp/tune_gemv && /tmp/tune_gemv
#include "../../omchat_amd/csrc/gemv.hip"
#include <cstdio>
#include <vector>
void omchat_set_error(const std::string& s) { fprintf(stderr, "ERR %s\n", s.c_str()); }

__global__ void stream_read_kernel(const u32x4* p, size_t n16, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

struct Shape { const char* name; int N, K, epi, ntile, ks; };

template <int NTILE, int EPI, int WAVES, int UNROLL, bool NTL>
float run(const Shape& sh, const std::vector<void*>& W, void* X, void* Y, int iters) {
  GemvP p{X, nullptr, Y, nullptr, nullptr, sh.K, sh.K, EPI == EPI_SWIGLU ? sh.N / 2 : sh.N, 0, 1, sh.N, sh.K, 0, sh.ks};
  dim3 grid((sh.N + NTILE * 16 - 1) / (NTILE * 16), sh.ks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { p.W = W[i % W.size()]; hipLaunchKernelGGL((gemv_kernel<bf16, NTILE, EPI, WAVES, UNROLL, NTL>), grid, dim3(WAVES * 64), 0, 0, p); }
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) { p.W = W[i % W.size()]; hipLaunchKernelGGL((gemv_kernel<bf16, NTILE, EPI, WAVES, UNROLL, NTL>), grid, dim3(WAVES * 64), 0, 0, p); }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

#define RUN(NT_, EPI_, WV, UN, NTL_) do { float us = run<NT_, EPI_, WV, UN, NTL_>(sh, W, X, Y, iters); \
  printf("  %-10s NTILE=%d waves=%d unroll=%d nt=%d ks=%d : %7.1f us  %6.2f TB/s\n", sh.name, NT_, WV, UN, (int)NTL_, sh.ks, us, bytes / us / 1e6); } while (0)

int main() {
  const int iters = 40;
  // ceiling: plain streaming read of 2 GiB
  {
    size_t n = (size_t)2 << 30; void* buf; hipMalloc(&buf, n); hipMemset(buf, 1, n); unsigned* o; hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int g : {1024, 2048, 4096, 8192}) {
      hipLaunchKernelGGL(stream_read_kernel, dim3(g), dim3(256), 0, 0, (const u32x4*)buf, n / 16, o);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(stream_read_kernel, dim3(g), dim3(256), 0, 0, (const u32x4*)buf, n / 16, o);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("stream read 2 GiB grid=%d: %.2f TB/s\n", g, n * 5.0 / ms / 1e9);
    }
    hipFree(buf);
  }
  Shape shapes[] = {{"gateup", 37888, 3584, EPI_SWIGLU, 2, 1}, {"down", 3584, 18944, EPI_PARTIAL, 1, 6}, {"down", 3584, 18944, EPI_PARTIAL, 1, 4},
                    {"down", 3584, 18944, EPI_PARTIAL, 1, 8}, {"o", 3584, 3584, EPI_PARTIAL, 1, 3}, {"o", 3584, 3584, EPI_PARTIAL, 1, 2},
                    {"qkv", 4608, 3584, EPI_NONE, 1, 1}, {"qkv", 4608, 3584, EPI_PARTIAL, 1, 2}, {"lm_head", 152064, 3584, EPI_NONE, 2, 1}};
  for (auto& sh : shapes) {
    const double bytes = (double)sh.N * sh.K * 2;
    const int nbuf = (int)(1.2e9 / bytes) + 2;
    std::vector<void*> W(nbuf > 12 ? 12 : nbuf);
    for (auto& w : W) { hipMalloc(&w, (size_t)bytes); hipMemset(w, 0x3c, (size_t)bytes); }
    void *X, *Y; hipMalloc(&X, (size_t)sh.K * 2 * 16); hipMemset(X, 0x3c, (size_t)sh.K * 2 * 16); hipMalloc(&Y, (size_t)sh.N * 4 * 16 * 8);
    printf("%s N=%d K=%d (%.1f MB, %zu buffers)\n", sh.name, sh.N, sh.K, bytes / 1e6, W.size());
    if (sh.epi == EPI_SWIGLU) {
      RUN(2, EPI_SWIGLU, 8, 4, true); RUN(2, EPI_SWIGLU, 8, 4, false); RUN(2, EPI_SWIGLU, 8, 2, true); RUN(2, EPI_SWIGLU, 8, 8, true);
      RUN(2, EPI_SWIGLU, 4, 4, true); RUN(2, EPI_SWIGLU, 4, 8, true); RUN(2, EPI_SWIGLU, 16, 2, true); RUN(2, EPI_SWIGLU, 16, 4, true);
    } else if (sh.epi == EPI_PARTIAL) {
      RUN(1, EPI_PARTIAL, 8, 4, true); RUN(1, EPI_PARTIAL, 8, 4, false); RUN(1, EPI_PARTIAL, 8, 8, true); RUN(1, EPI_PARTIAL, 4, 4, true);
      RUN(1, EPI_PARTIAL, 4, 8, true); RUN(1, EPI_PARTIAL, 16, 4, true); RUN(2, EPI_PARTIAL, 8, 4, true);
    } else if (sh.ntile == 2) {
      RUN(2, EPI_NONE, 8, 4, true); RUN(2, EPI_NONE, 8, 8, true); RUN(2, EPI_NONE, 4, 4, true); RUN(2, EPI_NONE, 16, 4, true);
    } else {
      RUN(1, EPI_NONE, 8, 4, true); RUN(1, EPI_NONE, 8, 8, true); RUN(1, EPI_NONE, 4, 4, true); RUN(1, EPI_NONE, 16, 2, true);
    }
    for (auto& w : W) hipFree(w);
    hipFree(X); hipFree(Y);
  }
  return 0;
}
