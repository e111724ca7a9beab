// Diagnostic harness (not product): the fused attention + merge + o_proj launch of csrc/fused_decode.hip alone, at the configs[1] decode shape
// (28 / 4 heads, 3584 wide, ~3.6 k keys), timed back to back and with in-kernel phase stamps (s_memrealtime, 100 MHz) of wave 7 and wave 0 of
// every workgroup.  Compiled WITH the stamps (OMCHAT_FUSED_STAMPS); the library build has none.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOMCHAT_FUSED_STAMPS tools/tune_fused.hip -o tools/bin/tune_fused
#include "../../omchat_amd/csrc/experiments/fused_decode.hip"
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
  const int QH = 28, KVH = 4, H = 3584, qd = 3584, CAP = 4096;
  const int G = device_cus();
  bf16 *kc, *vc, *qkv, *wo, *x;
  float* rope;
  CK(hipMalloc(&kc, (size_t)KVH * CAP * 128 * 2)); CK(hipMalloc(&vc, (size_t)KVH * CAP * 128 * 2));
  CK(hipMalloc(&qkv, (size_t)(QH + 2 * KVH) * 128 * 2)); CK(hipMalloc(&wo, (size_t)H * qd * 2)); CK(hipMalloc(&x, (size_t)H * 2));
  CK(hipMalloc(&rope, (size_t)CAP * 128 * 4));
  hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, kc, (size_t)KVH * CAP * 128, 1u, 1.0f);
  hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, vc, (size_t)KVH * CAP * 128, 2u, 1.0f);
  hipLaunchKernelGGL(fill_bf16, dim3(64), dim3(256), 0, 0, qkv, (size_t)(QH + 2 * KVH) * 128, 3u, 1.0f);
  hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, wo, (size_t)H * qd, 4u, 0.02f);
  hipLaunchKernelGGL(fill_bf16, dim3(64), dim3(256), 0, 0, x, (size_t)H, 5u, 1.0f);
  std::vector<float> tab((size_t)CAP * 128);
  for (int i = 0; i < 64; ++i) for (int pos = 0; pos < CAP; ++pos) { const float ang = pos * powf(1e6f, -(2.f * i) / 128.f); tab[((size_t)pos * 64 + i) * 2] = cosf(ang); tab[((size_t)pos * 64 + i) * 2 + 1] = sinf(ang); }
  CK(hipMemcpy(rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  void* ws; const size_t wsb = fused_decode_ws_bytes(QH); CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  unsigned* err; CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
  u64* dbg; CK(hipMalloc(&dbg, (size_t)G * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)G * 16 * 8));
  float* attn_ws; CK(hipMalloc(&attn_ws, attn_decode_ws_bytes(1, QH, CAP)));

  AttnDecodeArgs a{};
  a.Q = qkv; a.q_sb = (QH + 2 * KVH) * 128; a.q_sh = 128;
  a.K = kc; a.k_sb = (int64_t)KVH * CAP * 128; a.k_sh = (int64_t)CAP * 128; a.k_sr = 128;
  a.V = vc; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
  a.O = nullptr; a.o_sb = qd; a.o_sh = 128;
  a.batch = 1; a.q_heads = QH; a.kv_heads = KVH; a.L = L; a.kv_len = nullptr; a.scale = 0.08838834764831845f;
  a.ws = attn_ws; a.ws_bytes = attn_decode_ws_bytes(1, QH, CAP);
  a.rope = rope; a.rope_max = CAP; a.pos = nullptr;
  a.k_new = qkv + QH * 128; a.v_new = qkv + (QH + KVH) * 128; a.new_sb = (QH + 2 * KVH) * 128;
  unsigned epoch = 0;
  auto launch = [&](void* d) {
    FusedDecodeArgs fa{wo, qd, x, H, qd, ws, ++epoch, err, 2000};
    fa.dbg = d;
    if (launch_attn_oproj_fused(OMCHAT_BF16, a, fa, 0)) exit(1);
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch(nullptr);
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) launch(nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("L = %d: fused launch, back to back: %.2f us per launch\n", L, ms * 1e3 / 200);
  }
  // a big streaming kernel in between (cold caches / realistic neighbour), then one stamped launch
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, wo, (size_t)H * qd, 4u, 0.02f);
    launch(dbg);
    CK(hipDeviceSynchronize());
    std::vector<u64> h((size_t)G * 16);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    u64 t0 = ~0ull;
    for (int c = 0; c < G; ++c) { if (h[(size_t)c * 16 + 0]) t0 = std::min(t0, h[(size_t)c * 16 + 0]); if (h[(size_t)c * 16 + 8]) t0 = std::min(t0, h[(size_t)c * 16 + 8]); }
    const char* names[16] = {"w7 entry", "w7 tile done", "w7 partials stored", "w7 merge sweep done", "w7 merged published", "w7 row sweep done", "w7 final barrier", "",
                             "w0 entry", "w0 W loads issued", "w0 barrier (b) passed", "w0 row sweep done", "w0 rows stored", "", "", ""};
    printf("stamped launch %d (us after the first workgroup's entry; median / max over workgroups):\n", rep);
    for (int sl = 0; sl < 13; ++sl) {
      if (!names[sl][0]) continue;
      std::vector<double> v;
      for (int c = 0; c < G; ++c) if (h[(size_t)c * 16 + sl]) v.push_back((double)(h[(size_t)c * 16 + sl] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("  %-24s n=%3zu  min %6.2f  median %6.2f  max %6.2f\n", names[sl], v.size(), v.front(), v[v.size() / 2], v.back());
    }
    CK(hipMemset(dbg, 0, (size_t)G * 16 * 8));
  }
  unsigned he = 0; CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
  printf("timeout bits: %u\n", he);
  return 0;
}
