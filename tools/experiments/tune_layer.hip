// Diagnostic harness (not product): the one-launch decoder layer of csrc/decode_layer.hip alone at the configs[1] decode shape (Qwen2-7B
// widths, ~3.6 k keys): back-to-back time per launch over 28 different layers' weights (cold, as in the model) and in-kernel phase stamps
// (s_memrealtime, 100 MHz) of wave 7 and wave 0 of every workgroup.  Compiled WITH the stamps (OMCHAT_FUSED_STAMPS); the library has none.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOMCHAT_FUSED_STAMPS tools/tune_layer.hip -o tools/bin/tune_layer
#include "../../omchat_amd/csrc/experiments/decode_layer.hip"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

void omchat_set_error(const std::string& s) { fprintf(stderr, "error: %s\n", s.c_str()); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(bf16* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16)(((float)(h & 0xffff) / 32768.f - 1.f) * scale);
  }
}

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 3648;
  const int NL = 28, QH = 28, KVH = 4, H = 3584, qd = 3584, kvd = 512, It = 18944, CAP = 4096, qkvd = qd + 2 * kvd;
  const int G = device_cus();
  struct Lw { bf16 *ln1, *ln2, *wqkv, *bqkv, *wo, *wgu, *wd, *kc, *vc; };
  std::vector<Lw> lw(NL);
  auto mk = [&](bf16** p, size_t n, unsigned seed, float sc) { CK(hipMalloc(p, n * 2)); hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, *p, n, seed, sc); };
  for (int i = 0; i < NL; ++i) {
    mk(&lw[i].ln1, H, 11 + i, 1.f); mk(&lw[i].ln2, H, 12 + i, 1.f); mk(&lw[i].wqkv, (size_t)qkvd * H, 13 + i, 0.02f); mk(&lw[i].bqkv, qkvd, 14 + i, 0.02f);
    mk(&lw[i].wo, (size_t)H * qd, 15 + i, 0.02f); mk(&lw[i].wgu, (size_t)2 * It * H, 16 + i, 0.02f); mk(&lw[i].wd, (size_t)H * It, 17 + i, 0.02f);
    mk(&lw[i].kc, (size_t)KVH * CAP * 128, 18 + i, 1.f); mk(&lw[i].vc, (size_t)KVH * CAP * 128, 19 + i, 1.f);
  }
  bf16* x; mk(&x, H, 5, 1.0f);
  float* rope; CK(hipMalloc(&rope, (size_t)CAP * 128 * 4));
  std::vector<float> tab((size_t)CAP * 128);
  for (int i = 0; i < 64; ++i) for (int pos = 0; pos < CAP; ++pos) { const float ang = pos * powf(1e6f, -(2.f * i) / 128.f); tab[((size_t)pos * 64 + i) * 2] = cosf(ang); tab[((size_t)pos * 64 + i) * 2 + 1] = sinf(ang); }
  CK(hipMemcpy(rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  const size_t wsb = decode_layer_ws_bytes(QH, H, qd, kvd, It);
  void* ws; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  unsigned* err; CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
  u64* dbg; CK(hipMalloc(&dbg, (size_t)G * 32 * 8)); CK(hipMemset(dbg, 0, (size_t)G * 32 * 8));
  unsigned epoch = 0;
  auto launch = [&](int i, void* d) {
    DecodeLayerArgs a{lw[i].ln1, lw[i].ln2, lw[i].wqkv, lw[i].bqkv, lw[i].wo, lw[i].wgu, lw[i].wd, lw[i].kc, lw[i].vc, (int64_t)CAP * 128, x,
                      H, qd, kvd, It, QH, KVH, L, rope, CAP, 1e-6f, 0.08838834764831845f, ws, ++epoch, err, 2000};
    a.dbg = d;
    if (launch_decode_layer(OMCHAT_BF16, a, 0)) exit(1);
  };
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < NL; ++i) launch(i, nullptr);
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int t = 0; t < 4; ++t) for (int i = 0; i < NL; ++i) launch(i, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("L = %d: one-launch layer, 28 layers' weights in turn: %.2f us per layer launch  (%.3f ms per 28 layers)\n", L, ms * 1e3 / (4 * NL), ms / 4);
  }
  const char* names[32] = {"w7 K/V requested", "w7 q/k/v gathered", "w7 tile done", "w7 partials stored", "w7 merge sweep done", "w7 merged published",
                           "w7 attn row swept", "w7 x2 row swept", "w0 qkv rows published", "w0 o_proj done, x2 published", "w0 gate|up done", "w7 act row swept",
                           "w0 down rows stored"};
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < 5; ++i) launch(i, nullptr);
    launch(5, dbg);
    CK(hipDeviceSynchronize());
    std::vector<u64> h((size_t)G * 32);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    u64 t0 = ~0ull;
    for (int c = 0; c < G; ++c) for (int sl = 0; sl < 13; ++sl) if (h[(size_t)c * 32 + sl]) t0 = std::min(t0, h[(size_t)c * 32 + sl]);
    printf("stamped launch %d (us after the earliest stamp; min / median / max over workgroups):\n", rep);
    const int order[13] = {8, 0, 1, 2, 3, 4, 5, 6, 9, 7, 10, 11, 12};
    for (int oi = 0; oi < 13; ++oi) {
      const int sl = order[oi];
      std::vector<double> v;
      for (int c = 0; c < G; ++c) if (h[(size_t)c * 32 + sl]) v.push_back((double)(h[(size_t)c * 32 + sl] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("  %-30s n=%3zu  min %6.2f  median %6.2f  max %6.2f\n", names[sl], v.size(), v.front(), v[v.size() / 2], v.back());
    }
    // who is slow?  the gate|up phase (stamp 10 - stamp 9) by c % 8 (workgroups that share an XCD under round-robin dispatch) and by c / 32
    {
      double by8[8] = {0}, by32[8] = {0}; int n8[8] = {0}, n32[8] = {0};
      for (int c = 0; c < G; ++c) {
        if (!h[(size_t)c * 32 + 10] || !h[(size_t)c * 32 + 9]) continue;
        const double d = (double)(h[(size_t)c * 32 + 10] - h[(size_t)c * 32 + 9]) * 0.01;
        by8[c % 8] += d; ++n8[c % 8]; by32[(c / 32) % 8] += d; ++n32[(c / 32) % 8];
      }
      printf("  gate|up phase by c %% 8:");  for (int i = 0; i < 8; ++i) printf(" %5.1f", by8[i] / (n8[i] ? n8[i] : 1));
      printf("\n  gate|up phase by c / 32:"); for (int i = 0; i < 8; ++i) printf(" %5.1f", by32[i] / (n32[i] ? n32[i] : 1));
      printf("\n  slowest 12 workgroups:");
      std::vector<std::pair<double, int>> v;
      for (int c = 0; c < G; ++c) if (h[(size_t)c * 32 + 10]) v.push_back({(double)(h[(size_t)c * 32 + 10] - t0) * 0.01, c});
      std::sort(v.begin(), v.end());
      for (size_t i = v.size() >= 12 ? v.size() - 12 : 0; i < v.size(); ++i) printf(" %d(%.1f)", v[i].second, v[i].first);
      printf("\n  fastest 12:");
      for (size_t i = 0; i < 12 && i < v.size(); ++i) printf(" %d(%.1f)", v[i].second, v[i].first);
      printf("\n");
    }
    CK(hipMemset(dbg, 0, (size_t)G * 32 * 8));
  }
  unsigned he = 0; CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
  printf("timeout bits: %u\n", he);
  return 0;
}
