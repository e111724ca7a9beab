// Debug harness (not product): the one-launch decoder layer against the separate launches, piece by piece: q/k/v values, split-KV partials
// (O, m, l), merged attention row, x + attn.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg_layer_parts.hip omchat_amd/csrc/{attention,gemv}.hip
#include "../../omchat_amd/csrc/experiments/decode_layer.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
  const int QH = argc > 1 ? atoi(argv[1]) : 4, KVH = argc > 2 ? atoi(argv[2]) : 2, H = argc > 3 ? atoi(argv[3]) : 256, It = argc > 4 ? atoi(argv[4]) : 512;
  const int L = argc > 5 ? atoi(argv[5]) : 22, CAP = 4096;
  const int qd = QH * 128, kvd = KVH * 128, qkvd = qd + 2 * kvd;
  bf16 *ln1, *ln2, *wqkv, *bqkv, *wo, *wgu, *wd, *kc, *vc, *kc2, *vc2, *x, *x2, *qkv, *ao;
  auto mk = [&](bf16** p, size_t n, unsigned seed, float sc) { CK(hipMalloc(p, n * 2)); hipLaunchKernelGGL(fill_bf16, dim3(256), dim3(256), 0, 0, *p, n, seed, sc); };
  mk(&ln1, H, 11, 1.f); mk(&ln2, H, 12, 1.f); mk(&wqkv, (size_t)qkvd * H, 13, 0.05f); mk(&bqkv, qkvd, 14, 0.05f); mk(&wo, (size_t)H * qd, 15, 0.05f);
  mk(&wgu, (size_t)2 * It * H, 16, 0.05f); mk(&wd, (size_t)H * It, 17, 0.05f); mk(&kc, (size_t)KVH * CAP * 128, 18, 1.f); mk(&vc, (size_t)KVH * CAP * 128, 19, 1.f);
  mk(&x, H, 5, 1.0f); CK(hipMalloc(&x2, H * 2)); CK(hipMalloc(&qkv, qkvd * 2)); CK(hipMalloc(&ao, qd * 2));
  CK(hipMalloc(&kc2, (size_t)KVH * CAP * 128 * 2)); CK(hipMalloc(&vc2, (size_t)KVH * CAP * 128 * 2));
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(kc2, kc, (size_t)KVH * CAP * 128 * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(vc2, vc, (size_t)KVH * CAP * 128 * 2, hipMemcpyDeviceToDevice));
  CK(hipMemcpy(x2, x, H * 2, hipMemcpyDeviceToDevice));
  float* rope; CK(hipMalloc(&rope, (size_t)CAP * 128 * 4));
  std::vector<float> tab((size_t)CAP * 128);
  for (int i = 0; i < 64; ++i) for (int pos = 0; pos < CAP; ++pos) { const float ang = pos * powf(1e6f, -(2.f * i) / 128.f); tab[((size_t)pos * 64 + i) * 2] = cosf(ang); tab[((size_t)pos * 64 + i) * 2 + 1] = sinf(ang); }
  CK(hipMemcpy(rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  // ---- reference: qkv GEMV with the norm, split-KV attention + merge
  GemvArgs g{x, H, wqkv, H, qkv, qkvd, 1, qkvd, H, bqkv, nullptr, 0, EPI_NONE, 0};
  g.norm_w = ln1; g.norm_eps = 1e-6f;
  if (launch_gemv(OMCHAT_BF16, g, 0)) return 1;
  float* aws; const size_t awsb = attn_decode_ws_bytes(1, QH, CAP); CK(hipMalloc(&aws, awsb)); CK(hipMemset(aws, 0, awsb));
  AttnDecodeArgs a{};
  a.Q = qkv; a.q_sb = qkvd; a.q_sh = 128;
  a.K = kc; a.k_sb = (int64_t)KVH * CAP * 128; a.k_sh = (int64_t)CAP * 128; a.k_sr = 128;
  a.V = vc; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
  a.O = ao; a.o_sb = qd; a.o_sh = 128;
  a.batch = 1; a.q_heads = QH; a.kv_heads = KVH; a.L = L; a.kv_len = nullptr; a.scale = 0.08838834764831845f;
  a.ws = aws; a.ws_bytes = awsb; a.rope = rope; a.rope_max = CAP;
  a.k_new = qkv + qd; a.v_new = qkv + qd + kvd; a.new_sb = qkvd;
  if (launch_attn_decode(OMCHAT_BF16, a, 0)) return 1;
  CK(hipDeviceSynchronize());
  // ---- the one-launch layer on copies
  const size_t wsb = decode_layer_ws_bytes(QH, H, qd, kvd, It);
  void* ws; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  unsigned* err; CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
  DecodeLayerArgs d{ln1, ln2, wqkv, bqkv, wo, wgu, wd, kc2, vc2, (int64_t)CAP * 128, x2, H, qd, kvd, It, QH, KVH, L, rope, CAP, 1e-6f, 0.08838834764831845f, ws, 7u, err, 2000};
  if (!decode_layer_ok(d)) { printf("geometry not supported\n"); return 1; }
  if (launch_decode_layer(OMCHAT_BF16, d, 0)) return 1;
  CK(hipDeviceSynchronize());
  unsigned he = 0; CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost)); printf("timeout bits %u\n", he);
  std::vector<u64> hw(wsb / 8); CK(hipMemcpy(hw.data(), ws, wsb, hipMemcpyDeviceToHost));
  std::vector<unsigned short> hq(qkvd); CK(hipMemcpy(hq.data(), qkv, qkvd * 2, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < qkvd; ++i) if ((unsigned short)hw[i] != hq[i]) { if (bad < 8) printf("qkv[%d]: layer %04x ref %04x\n", i, (unsigned)(unsigned short)hw[i], hq[i]); ++bad; }
  printf("q/k/v values: %d of %d differ\n", bad, qkvd);
  // partials
  const int ns = (L + 63) / 64;
  std::vector<float> hp(awsb / 4); CK(hipMemcpy(hp.data(), aws, awsb, hipMemcpyDeviceToHost));
  const u64* pg = hw.data() + qkvd;
  bad = 0;
  for (int h = 0; h < QH; ++h) for (int s = 0; s < ns; ++s) for (int dd = 0; dd < 130; ++dd) {
    const unsigned mine = (unsigned)pg[((size_t)h * 64 + s) * 132 + dd];
    unsigned ref; memcpy(&ref, &hp[((size_t)h * ns + s) * 132 + dd], 4);
    if (mine != ref) { if (bad < 12) { float fm, fr; memcpy(&fm, &mine, 4); memcpy(&fr, &ref, 4); printf("partial head %d split %d col %d: layer %g ref %g\n", h, s, dd, fm, fr); } ++bad; }
  }
  printf("partials: %d of %d differ\n", bad, QH * ns * 130);
  // cache rows appended
  std::vector<unsigned short> k1((size_t)KVH * CAP * 128), k2(k1.size());
  CK(hipMemcpy(k1.data(), kc, k1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(k2.data(), kc2, k2.size() * 2, hipMemcpyDeviceToHost));
  bad = 0; for (size_t i = 0; i < k1.size(); ++i) if (k1[i] != k2[i]) { if (bad < 4) printf("K cache elem %zu (head %zu row %zu col %zu): layer %04x ref %04x\n", i, i / (CAP * 128), (i / 128) % CAP, i % 128, k2[i], k1[i]); ++bad; }
  printf("K cache: %d differ\n", bad);
  {
    // the inputs of the first differing elements of head 0's appended row: raw k (both halves), cos / sin as the table holds them
    const int pp = L - 1; int shown = 0;
    for (int col = 0; col < 128 && shown < 6; ++col) {
      const size_t i = (size_t)pp * 128 + col;
      if (k1[i] == k2[i]) continue;
      const int j = col & 63;
      printf("col %d: k[col] %04x  k[partner %d] %04x  cos %.9g sin %.9g   layer %04x ref %04x\n", col, hq[qd + col], col ^ 64, hq[qd + (col ^ 64)],
             tab[((size_t)pp * 64 + j) * 2], tab[((size_t)pp * 64 + j) * 2 + 1], k2[i], k1[i]);
      ++shown;
    }
  }
  CK(hipMemcpy(k1.data(), vc, k1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(k2.data(), vc2, k2.size() * 2, hipMemcpyDeviceToHost));
  bad = 0; for (size_t i = 0; i < k1.size(); ++i) if (k1[i] != k2[i]) ++bad;
  printf("V cache: %d differ\n", bad);
  // merged row
  std::vector<unsigned short> hao(qd); CK(hipMemcpy(hao.data(), ao, qd * 2, hipMemcpyDeviceToHost));
  const u64* ag = pg + (size_t)QH * 64 * 132;
  bad = 0; for (int i = 0; i < qd; ++i) { const unsigned short m = (unsigned short)(ag[i >> 1] >> (16 * (i & 1))); if (m != hao[i]) ++bad; }
  printf("merged attention row: %d of %d differ\n", bad, qd);
  return 0;
}
