// Tuning harness (not product): whole-row streaming GEMV variants on the decode shapes.
#include "../../omchat_amd/csrc/gemv.hip"
#include <cstdio>
#include <vector>
void omchat_set_error(const std::string& s) { fprintf(stderr, "ERR %s\n", s.c_str()); }
struct Shape { const char* name; int N, K, epi; };
template <int EPI, int RR, int WAVES>
float run(const Shape& sh, int ks, int gridcap, const std::vector<void*>& W, void* X, void* Y, int iters) {
  GemvP p{X, nullptr, Y, nullptr, nullptr, sh.K, sh.K, EPI == EPI_SWIGLU ? sh.N / 2 : sh.N, 0, 1, sh.N, sh.K, 0, ks};
  const int n_out = EPI == EPI_SWIGLU ? sh.N / 2 : sh.N;
  int grid = (((n_out + RR - 1) / RR) + WAVES - 1) / WAVES;
  grid = grid > gridcap ? gridcap : grid;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { p.W = W[i % W.size()]; hipLaunchKernelGGL((gemv_rows_kernel<bf16, EPI, RR, WAVES>), dim3(grid, ks), dim3(WAVES * 64), 0, 0, p); }
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) { p.W = W[i % W.size()]; hipLaunchKernelGGL((gemv_rows_kernel<bf16, EPI, RR, WAVES>), dim3(grid, ks), dim3(WAVES * 64), 0, 0, p); }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}
#define RUN(EPI_, RR_, WV, KS, CAP) do { float us = run<EPI_, RR_, WV>(sh, KS, CAP, W, X, Y, iters); \
  printf("  %-8s R=%d waves=%d ks=%d cap=%d : %7.1f us  %5.2f TB/s\n", sh.name, RR_, WV, KS, CAP, us, bytes / us / 1e6); } while (0)
int main() {
  const int iters = 40;
  Shape shapes[] = {{"gateup", 37888, 3584, EPI_SWIGLU}, {"down", 3584, 18944, EPI_PARTIAL}, {"o", 3584, 3584, EPI_PARTIAL}, {"qkv", 4608, 3584, EPI_NONE},
                    {"lm_head", 152064, 3584, EPI_NONE}};
  for (auto& sh : shapes) {
    const double bytes = (double)sh.N * sh.K * 2;
    int nbuf = (int)(1.2e9 / bytes) + 2; nbuf = nbuf > 12 ? 12 : nbuf;
    std::vector<void*> W(nbuf);
    for (auto& w : W) { hipMalloc(&w, (size_t)bytes); hipMemset(w, 0x3c, (size_t)bytes); }
    void *X, *Y; hipMalloc(&X, (size_t)sh.K * 2); hipMemset(X, 0x3c, (size_t)sh.K * 2); hipMalloc(&Y, (size_t)sh.N * 4 * 8);
    printf("%s N=%d K=%d (%.1f MB)\n", sh.name, sh.N, sh.K, bytes / 1e6);
    if (sh.epi == EPI_SWIGLU) {
      RUN(EPI_SWIGLU, 4, 4, 1, 2048); RUN(EPI_SWIGLU, 2, 4, 1, 2048); RUN(EPI_SWIGLU, 4, 8, 1, 1024); RUN(EPI_SWIGLU, 2, 8, 1, 2048); RUN(EPI_SWIGLU, 4, 4, 1, 1024);
      RUN(EPI_SWIGLU, 1, 8, 1, 2048); RUN(EPI_SWIGLU, 4, 2, 1, 4096);
    } else if (sh.epi == EPI_PARTIAL && sh.K > 8192) {
      RUN(EPI_PARTIAL, 4, 4, 5, 2048); RUN(EPI_PARTIAL, 2, 4, 5, 2048); RUN(EPI_PARTIAL, 4, 4, 8, 2048); RUN(EPI_PARTIAL, 2, 4, 8, 2048); RUN(EPI_PARTIAL, 4, 8, 5, 2048);
      RUN(EPI_PARTIAL, 8, 4, 5, 2048); RUN(EPI_PARTIAL, 8, 4, 8, 2048); RUN(EPI_PARTIAL, 1, 8, 5, 2048);
    } else if (sh.epi == EPI_PARTIAL) {
      RUN(EPI_PARTIAL, 4, 4, 3, 2048); RUN(EPI_PARTIAL, 2, 4, 3, 2048); RUN(EPI_PARTIAL, 2, 4, 2, 2048); RUN(EPI_PARTIAL, 2, 4, 1, 2048); RUN(EPI_PARTIAL, 1, 4, 1, 2048);
      RUN(EPI_PARTIAL, 1, 8, 2, 2048); RUN(EPI_PARTIAL, 4, 4, 1, 2048); RUN(EPI_PARTIAL, 2, 8, 1, 2048);
    } else {
      RUN(EPI_NONE, 4, 4, 1, 2048); RUN(EPI_NONE, 2, 4, 1, 2048); RUN(EPI_NONE, 1, 4, 1, 2048); RUN(EPI_NONE, 2, 8, 1, 2048); RUN(EPI_NONE, 1, 8, 1, 2048); RUN(EPI_NONE, 8, 4, 1, 2048);
    }
    for (auto& w : W) hipFree(w);
    hipFree(X); hipFree(Y);
  }
  return 0;
}
