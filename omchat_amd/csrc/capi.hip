// Op-level C ABI entry points (unit parity tests / micro benches) and the FlashAttention-shaped seam.
#include "kernels.h"
#include <stdlib.h>
#include "../../include/omchat_hip.h"
#include <math.h>
#include <vector>

#define S(x) ((hipStream_t)(x))
#if OMCHAT_EXPERIMENTS
int g_launch_any_order = 0;      // tuning key 41: timing probe, see kernels.h
#endif

extern "C" int omchat_op_gemm(int dtype, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                              const void* bias, const void* ls, const void* resid, int ldr, int epi, int force_tile, void* stream) {
  GemmArgs g{A, lda, W, ldw, C, ldc, M, N, K, bias, ls, resid, ldr, epi, force_tile, nullptr, 0, -1};
  return launch_gemm(dtype, g, S(stream));
}

// round 6: the GEMM with the folded-norm row scale and / or the statistics epilogues (epi 7 = LS_RESID + stats, 8 = NONE + stats);
// *nslots (host int, may be null) receives the number of statistics slots the launch wrote (one per wave tile of the tile kernel that ran)
extern "C" int omchat_op_gemm_fused(int dtype, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                    const void* bias, const void* ls, const void* resid, int ldr, int epi, int force_tile, const float* row_scale,
                                    const float* rs_stats, int rs_ld, int rs_nslots, int rs_dim, float rs_eps,
                                    float* stats, int stats_ld, int* nslots, void* stream) {
  GemmArgs g{A, lda, W, ldw, C, ldc, M, N, K, bias, ls, resid, ldr, epi, force_tile, nullptr, 0, -1};
  g.row_scale = row_scale; g.stats = stats; g.stats_ld = stats_ld; g.stats_nslots = nslots;
  g.rs_stats = rs_stats; g.rs_ld = rs_ld; g.rs_nslots = rs_nslots; g.rs_dim = rs_dim; g.rs_eps = rs_eps;
  return launch_gemm(dtype, g, S(stream));
}
extern "C" int omchat_op_stats_finish(const float* stats, int ld, int slot0, int nslots, int ngroups, int rows, int dim, float eps, float* out, void* stream) {
  return launch_stats_finish(stats, ld, slot0, nslots, ngroups, rows, dim, eps, out, S(stream));
}
extern "C" int omchat_op_row_sumsq(int dtype, const void* x, int ldx, int rows, int H, float* stats, void* stream) {
  return launch_row_sumsq(dtype, x, ldx, rows, H, stats, S(stream));
}
extern "C" int omchat_op_fold_cols(int dtype, const void* W, const void* n, void* out, int rows, int cols, void* stream) {
  return launch_fold_cols(dtype, W, n, out, rows, cols, S(stream));
}
// the K half of the joint-head q / k norm straight from the qkv GEMM's statistics slots (q slots [0, nslots), k slots [nslots, 2 nslots)); leaves the q sums
extern "C" int omchat_op_vit_knorm_slots(int dtype, void* k, int ld, const void* wk, int rows, int C, int C_total, float eps, const float* stats, int stats_ld,
                                         int nslots, float* sumsq_q, void* stream) {
  return launch_vit_knorm_slots(dtype, k, ld, wk, rows, C, C_total, eps, stats, stats_ld, nslots, sumsq_q, S(stream));
}
// omchat_mha_fwd on a packed qkv [B, S, 3, H, 128] whose Q is still RAW: the Q half of InternAttention's joint-head norm is applied on load from
// sumsq [B * S][stride] (element 0 of a row: the sum of squares of its q channels; K must already be normalised: omchat_op_vit_knorm_slots)
extern "C" int omchat_op_mha_qnorm(int dtype, const void* qkv, int B, int Sq, int H, const float* sumsq, int stride, int dim, const void* wq,
                                   float eps, float q_scale, void* out, void* stream) {
  OM_CHECK(qkv && out && sumsq && wq, "null argument");
  AttnArgs a{};
  const int64_t row = (int64_t)3 * H * 128;
  a.Q = qkv; a.q_sb = Sq * row; a.q_sh = 128; a.q_sr = row;
  a.K = (const char*)qkv + (size_t)H * 128 * 2; a.k_sb = a.q_sb; a.k_sh = 128; a.k_sr = row;
  a.V = (const char*)qkv + (size_t)2 * H * 128 * 2; a.v_sb = a.q_sb; a.v_sh = 128; a.v_sr = row;
  a.O = out; a.o_sb = (int64_t)Sq * H * 128; a.o_sh = 128; a.o_sr = (int64_t)H * 128;
  a.batch = B; a.q_heads = H; a.kv_heads = H; a.Sq = Sq; a.Skv = Sq; a.kv_len = nullptr; a.causal = 0; a.q_pos0 = 0;
  a.scale = 1.0f;
  a.qn_sumsq = sumsq; a.qn_stride = stride; a.qn_dim = dim; a.qn_w = wq; a.qn_eps = eps; a.qn_scale = q_scale;
  return launch_attn_prefill(dtype, a, S(stream));
}

// The tuning keys are PROCESS-GLOBAL switches for tests and measurements (A/B of kernel forms, launch shapes): they mutate state shared by
// every context of the process, so production callers must not touch them -- the call is refused unless the process opted in with
// OMCHAT_ALLOW_TUNING=1 in its environment (tests/conftest.py, bench.py --tuning and tools/gpu_job.sh set it).  Nothing in omchat_amd/ sets a key.
extern "C" int omchat_op_set_tuning(int key, int value) {
  const char* allow = getenv("OMCHAT_ALLOW_TUNING");
  if (!allow || allow[0] != '1') {
    omchat_set_error("omchat_op_set_tuning: process-global test / measurement hook; set OMCHAT_ALLOW_TUNING=1 to use it (see include/omchat_hip.h)");
    return 1;
  }
  if (key == 0) { gemm_set_skew(value); return 0; }
  if (key == 1) { gemv_set_force_mfma(value); return 0; }
  if (key == 5) { gemm_set_autotune(value); return 0; }
  if (key == 4) { model_set_ar_min_rows(value); return 0; }
  if (key == 6) { model_set_pack_replica(value); return 0; }
  if (key == 8) { attn_set_v2(value); return 0; }
  if (key == 9) { model_set_fuse_peer_norm(value); return 0; }
  if (key == 10) { attn_set_tpw(value); return 0; }
  if (key == 11) { gemv_set_no_xs(value); return 0; }
  if (key == 12) { attn_set_klds(value); return 0; }
  if (key == 25) { attn_set_dma(value); return 0; }
  if (key == 26) { attn_set_dma_slots(value); return 0; }
  if (key == 27) { attn_set_dma_rot(value); return 0; }
  if (key == 28) { gemv_set_skew(value); return 0; }
  if (key == 29) { model_set_tp_f32(value); return 0; }
  if (key == 30) { attn_set_hsplit(value); return 0; }
  if (key == 33) { attn_set_mha_xcd(value); return 0; }
  if (key == 16) { gemv_set_norm_loop(value); return 0; }
  if (key == 17) { gemv_set_rows_balance(value); return 0; }
  if (key == 19) { attn_set_merge_mid_min(value); return 0; }
  if (key == 21) { attn_set_merge_dg(value); return 0; }
  if (key == 13) { gemm_set_persist(value); return 0; }
  if (key == 14) { model_set_norm_in_gemv(value); return 0; }
  if (key == 22) { model_set_fuse_attn_oproj(value); return 0; }
  if (key == 23) { model_set_decode_layer(value); return 0; }
  if (key == 24) { gemv_set_dyn(value); return 0; }
  if (key == 34) { gemv_set_shard_shapes(value); return 0; }
  if (key == 35) { model_set_shard_as_tp1(value); return 0; }
  if (key == 36) { attn_set_kg(value); return 0; }
  if (key == 37) { gemm_set_wide_store(value); return 0; }
  if (key == 43) { gemm_set_skip_dead(value); return 0; }
  if (key == 44) { model_set_vit_fused(value); return 0; }
  if (key == 45) { model_set_tp_sp(value); return 0; }
  if (key == 46) { attn_set_peel(value); return 0; }
  if (key == 47) { attn_set_kv8_tpw(value); return 0; }
  if (key == 48) { attn_set_kv8_fuse(value); return 0; }
  if (key == 38) { gemv_set_gu_rr(value); return 0; }
  if (key == 39) { gemv_set_longk_direct(value); return 0; }
  if (key == 40) { norm_set_wave(value); return 0; }
#if OMCHAT_EXPERIMENTS
  if (key == 41) { g_launch_any_order = value; return 0; }
  if (key == 42) { model_set_ao_oproj(value); return 0; }
#endif
  omchat_set_error("omchat_op_set_tuning: unknown key");
  return 1;
}

extern "C" int omchat_allreduce_noop(void*, void*, size_t, int, void*) { return 0; }

extern "C" int omchat_gemm_tune_load(const char* path) { OM_CHECK(path, "null path"); return gemm_tune_load(path); }
extern "C" int omchat_gemm_tune_dump(const char* path) { OM_CHECK(path, "null path"); return gemm_tune_dump(path); }
extern "C" long omchat_gemm_tune_runs(void) { return gemm_tune_runs(); }

extern "C" size_t omchat_op_gemm_sk_ws(void) { return gemm_sk_ws_bytes(); }

extern "C" int omchat_op_gemm_sk(int dtype, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                 const void* bias, const void* ls, const void* resid, int ldr, int epi, int force_tile, void* ws,
                                 size_t ws_bytes, int stream_k, void* stream) {
  GemmArgs g{A, lda, W, ldw, C, ldc, M, N, K, bias, ls, resid, ldr, epi, force_tile, ws, ws_bytes, stream_k};
  return launch_gemm(dtype, g, S(stream));
}

extern "C" int omchat_op_gemv(int dtype, const void* X, int ldx, const void* W, int ldw, void* Y, int ldy, int b, int N, int K,
                              const void* bias, const void* resid, int ldr, int epi, int out_f32, void* stream) {
  GemvArgs g{X, ldx, W, ldw, Y, ldy, b, N, K, bias, resid, ldr, epi, out_f32};
  return launch_gemv(dtype, g, S(stream));
}

// batch-1 GEMV with the preceding RMSNorm in registers: y = epi(W RMSNorm(x; norm_w, eps)), x the RAW row [K] (K <= 4096)
extern "C" int omchat_op_gemv_norm(int dtype, const void* X, const void* W, int ldw, void* Y, int N, int K, const void* norm_w, float eps,
                                   const void* bias, int epi, int out_f32, void* stream) {
  OM_CHECK(X && W && Y && norm_w, "null argument");
  GemvArgs g{X, K, W, ldw, Y, N, 1, N, K, bias, nullptr, 0, epi, out_f32};
  g.norm_w = norm_w; g.norm_eps = eps;
  return launch_gemv(dtype, g, S(stream));
}

// packed-operand batched GEMV (gemv.hip: gemv_pk_kernel): X row-major [b, K] and W row-major [N, K] are packed into scratch here
// (launch_pack_x / launch_pack_w), then Y = epi(X W^T); y_packed != 0 returns the SwiGLU output in the packed x layout
extern "C" int omchat_op_gemv_packed(int dtype, const void* X, int ldx, const void* W, int ldw, void* Y, int ldy, int b, int N, int K,
                                     const void* bias, int epi, int out_f32, int ksplit, int w_packed, int y_packed, void* stream) {
  OM_CHECK(b >= 1 && b <= 32 && K % 64 == 0, "1 <= b <= 32, K % 64 == 0");
  const int NB = b > 16 ? 2 : 1;
  void *xp = nullptr, *wp = nullptr;
  OM_HIP(hipMalloc(&xp, (size_t)NB * 16 * K * 2));
  int rc = launch_pack_x(dtype, X, ldx, b, K, xp, S(stream));
  if (rc == 0 && w_packed) {
    if (hipMalloc(&wp, (size_t)N * K * 2) != hipSuccess) { hipFree(xp); omchat_set_error("hipMalloc failed"); return 2; }
    rc = launch_pack_w(dtype, W, ldw, N, K, wp, S(stream));
  }
  if (rc == 0) {
    GemvArgs g{xp, K, w_packed ? wp : W, ldw, Y, ldy, b, N, K, bias, nullptr, 0, epi, out_f32, ksplit, 0, nullptr, 1, w_packed, y_packed};
    rc = launch_gemv(dtype, g, S(stream));
  }
  hipStreamSynchronize(S(stream));
  hipFree(xp); if (wp) hipFree(wp);
  return rc;
}
extern "C" int omchat_op_pack_x(int dtype, const void* X, int ldx, int b, int K, void* out, void* stream) { return launch_pack_x(dtype, X, ldx, b, K, out, S(stream)); }

// fp8 x fp8 GEMM (BASELINE configs[4]): A8 [M, K], W8 [N, K] e4m3 bytes with per-row fp32 scales -> C (dtype) = epi(sa[m] sw[n] A8 W8^T)
extern "C" int omchat_op_gemm_fp8(int dtype, const void* A8, const float* a_scale, const void* W8, const float* w_scale, void* C, int ldc, int M,
                                  int N, int K, const void* bias, const void* resid, int ldr, int epi, void* stream) {
  GemmArgs g{A8, K, W8, K, C, ldc, M, N, K, bias, nullptr, resid, ldr, epi, 0, nullptr, 0, -1, 1, a_scale, w_scale};
  return launch_gemm(dtype, g, S(stream));
}
// per-row e4m3 quantisation of activations [rows, H]: norm_w != NULL applies the RMSNorm first (launch_rmsnorm_q8)
extern "C" int omchat_op_quant_rows_fp8(int dtype, const void* x, const void* norm_w, float eps, void* y8, float* scale, int rows, int H, void* stream) {
  if (norm_w) return launch_rmsnorm_q8(dtype, x, H, norm_w, y8, H, scale, rows, H, eps, S(stream));
  return launch_quant_rows_q8(dtype, x, H, y8, H, scale, rows, H, S(stream));
}

// decode: x = T(x + T(sum_s part[s])) in place, xn = RMSNorm(x) * w; part fp32 [ks][rows][H] (the split-K slices of the skinny GEMM)
extern "C" int omchat_op_resid_rmsnorm(int dtype, void* x, int ldx, const float* part, int ks, const void* w, void* xn, int ldn, int rows, int H,
                                       float eps, int pack_nb, void* stream) {
  OM_CHECK(x && part, "null argument");
  return launch_resid_rmsnorm(dtype, x, ldx, part, ks, w, xn, ldn, rows, H, eps, S(stream), pack_nb);
}
extern "C" int omchat_op_rmsnorm(int dtype, const void* x, const void* w, void* y, int rows, int H, float eps, void* stream) {
  return launch_rmsnorm(dtype, x, H, w, y, H, rows, H, eps, S(stream));
}

extern "C" int omchat_op_vit_qknorm(int dtype, void* qkv, int ld, const void* wq, const void* wk, int rows, int C, int C_total, float eps,
                                    float q_scale, void* stream) {
  return launch_vit_qknorm(dtype, qkv, ld, wq, wk, rows, C, C_total, eps, q_scale, nullptr, S(stream));
}

extern "C" int omchat_op_attn_prefill(int dtype, const void* q, const void* k, const void* v, void* out, int b, int Sq, int Skv, int Hq,
                                      int Hkv, const int32_t* kv_len, int causal, int q_pos0, float scale, void* stream) {
  AttnArgs a{};
  a.Q = q; a.q_sb = (int64_t)Sq * Hq * 128; a.q_sh = 128; a.q_sr = (int64_t)Hq * 128;
  a.K = k; a.k_sb = (int64_t)Hkv * Skv * 128; a.k_sh = (int64_t)Skv * 128; a.k_sr = 128;
  a.V = v; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
  a.O = out; a.o_sb = a.q_sb; a.o_sh = 128; a.o_sr = a.q_sr;
  a.batch = b; a.q_heads = Hq; a.kv_heads = Hkv; a.Sq = Sq; a.Skv = Skv; a.kv_len = kv_len; a.causal = causal; a.q_pos0 = q_pos0; a.scale = scale;
  return launch_attn_prefill(dtype, a, S(stream));
}

extern "C" size_t omchat_op_attn_decode_ws(int b, int Hq, int L) { return attn_decode_ws_bytes(b, Hq, L); }

extern "C" int omchat_op_attn_decode(int dtype, const void* q, const void* k, const void* v, void* out, int b, int Hq, int Hkv, int cap, int L,
                                     const int32_t* kv_len, float scale, void* ws, size_t ws_bytes, void* stream) {
  AttnDecodeArgs a{};
  a.Q = q; a.q_sb = (int64_t)Hq * 128; a.q_sh = 128;
  a.K = k; a.k_sb = (int64_t)Hkv * cap * 128; a.k_sh = (int64_t)cap * 128; a.k_sr = 128;
  a.V = v; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
  a.O = out; a.o_sb = a.q_sb; a.o_sh = 128;
  a.batch = b; a.q_heads = Hq; a.kv_heads = Hkv; a.L = L; a.kv_len = kv_len; a.scale = scale; a.ws = (float*)ws; a.ws_bytes = ws_bytes;
  return launch_attn_decode(dtype, a, S(stream));
}

extern "C" int omchat_op_attn_decode_kv8(int dtype, const void* q, const void* k8, const void* v8, const float* ks, const float* vs, void* out, int b,
                                         int Hq, int Hkv, int cap, int L, const int32_t* kv_len, float scale, void* ws, size_t ws_bytes, void* stream) {
  if (!k8 || !v8 || !ks || !vs) { omchat_set_error("omchat_op_attn_decode_kv8: null cache or scale pointer"); return 1; }
  AttnDecodeArgs a{};
  a.Q = q; a.q_sb = (int64_t)Hq * 128; a.q_sh = 128;
  a.K = k8; a.k_sb = (int64_t)Hkv * cap * 128; a.k_sh = (int64_t)cap * 128; a.k_sr = 128;      // strides in cache elements = bytes
  a.V = v8; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
  a.k_scale = ks; a.v_scale = vs; a.scale_sb = (int64_t)Hkv * cap; a.scale_sh = cap;
  a.O = out; a.o_sb = a.q_sb; a.o_sh = 128;
  a.batch = b; a.q_heads = Hq; a.kv_heads = Hkv; a.L = L; a.kv_len = kv_len; a.scale = scale; a.ws = (float*)ws; a.ws_bytes = ws_bytes;
  return launch_attn_decode(dtype, a, S(stream));
}

static float* rope_table_device(int max_pos, float theta) {
  std::vector<float> tab((size_t)max_pos * 128);
  for (int i = 0; i < 64; ++i) {
    const float inv = (float)(1.0 / pow((double)theta, (double)((float)(2 * i) / 128.0f)));
    for (int p = 0; p < max_pos; ++p) {
      const float ang = (float)p * inv;
      tab[((size_t)p * 64 + i) * 2] = (float)cos((double)ang);
      tab[((size_t)p * 64 + i) * 2 + 1] = (float)sin((double)ang);
    }
  }
  float* d = nullptr;
  if (hipMalloc(&d, tab.size() * 4) != hipSuccess) return nullptr;
  if (hipMemcpy(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
  return d;
}

// One decode step's attention over the e4m3 cache INCLUDING the new token's RoPE + append, as the decoder layer issues it (model.hip): qkv
// [b][(Hq + 2 Hkv) * 128] raw projections of the new token; sequence i holds kv_len[i] keys counting the new one (NULL: L for every sequence), the
// new row goes to position kv_len[i] - 1 of k8 / v8 / ks / vs and of the 16-bit caches k16 / v16.  Long launches take the walking form, which does
// the rotation, the append and the quantisation itself (tuning keys 47 / 48); the others run omchat_op_rope_kv_q8's launch in front (q rotated in
// place in qkv then).  pos = kv_len - 1 as a device array (needed by that launch when kv_len != NULL).
extern "C" int omchat_op_attn_decode_kv8_append(int dtype, void* qkv, float theta, void* k8, void* v8, float* ks, float* vs, void* k16, void* v16,
                                                void* out, int b, int Hq, int Hkv, int cap, int L, const int32_t* kv_len, const int32_t* pos, float scale,
                                                void* ws, size_t ws_bytes, void* stream) {
  OM_CHECK(qkv && k8 && v8 && ks && vs && k16 && v16 && out, "null argument");
  OM_CHECK(L >= 1 && L <= cap, "L exceeds the cache capacity");
  OM_CHECK((kv_len == nullptr) == (pos == nullptr), "kv_len and pos come together");
  float* tab = rope_table_device(L, theta);
  OM_CHECK(tab, "rope table allocation failed");
  const int qkvd = (Hq + 2 * Hkv) * 128;
  AttnDecodeArgs a{};
  a.Q = qkv; a.q_sb = qkvd; a.q_sh = 128;
  a.K = k8; a.k_sb = (int64_t)Hkv * cap * 128; a.k_sh = (int64_t)cap * 128; a.k_sr = 128;
  a.V = v8; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
  a.k_scale = ks; a.v_scale = vs; a.scale_sb = (int64_t)Hkv * cap; a.scale_sh = cap;
  a.O = out; a.o_sb = (int64_t)Hq * 128; a.o_sh = 128;
  a.batch = b; a.q_heads = Hq; a.kv_heads = Hkv; a.L = L; a.kv_len = kv_len; a.scale = scale; a.ws = (float*)ws; a.ws_bytes = ws_bytes;
  int rc = 0;
  if (attn_decode_kv8_fuses_rope(b, Hkv, L, false)) {
    a.rope = tab; a.rope_max = L;
    a.k_new = (const char*)qkv + (size_t)Hq * 128 * 2; a.v_new = (const char*)qkv + (size_t)(Hq + Hkv) * 128 * 2; a.new_sb = qkvd;
    a.k16_w = k16; a.v16_w = v16;
  } else {
    RopeArgs r{qkv, qkvd, b, 1, Hq, Hkv, pos, L - 1, tab, L, k16, v16, (int64_t)Hkv * cap * 128, (int64_t)cap * 128};
    r.k8 = k8; r.v8 = v8; r.ks = ks; r.vs = vs; r.s_sb = (int64_t)Hkv * cap; r.s_sh = cap;
    rc = launch_rope_kv(dtype, r, S(stream));
  }
  if (!rc) rc = launch_attn_decode(dtype, a, S(stream));
  hipStreamSynchronize(S(stream));
  hipFree(tab);
  return rc;
}

static int op_rope_kv_impl(int dtype, void* qkv, int b, int Sq, int Hq, int Hkv, int pos0, float theta, void* kcache, void* vcache, int cap,
                           void* k8, void* v8, float* ks, float* vs, void* stream);
extern "C" int omchat_op_rope_kv(int dtype, void* qkv, int b, int Sq, int Hq, int Hkv, int pos0, float theta, void* kcache, void* vcache, int cap,
                                 void* stream) {
  return op_rope_kv_impl(dtype, qkv, b, Sq, Hq, Hkv, pos0, theta, kcache, vcache, cap, nullptr, nullptr, nullptr, nullptr, stream);
}
// the same with the appended rows also quantised to the e4m3 cache (k8 / v8 [b, Hkv, cap, 128] bytes, ks / vs [b, Hkv, cap] fp32 scales):
// what a decode step with the fp8 KV cache launches (round 3; before: RoPE + append, then a quantiser launch over the new rows)
extern "C" int omchat_op_rope_kv_q8(int dtype, void* qkv, int b, int Sq, int Hq, int Hkv, int pos0, float theta, void* kcache, void* vcache, int cap,
                                    void* k8, void* v8, float* ks, float* vs, void* stream) {
  OM_CHECK(k8 && v8 && ks && vs, "null fp8 cache argument");
  return op_rope_kv_impl(dtype, qkv, b, Sq, Hq, Hkv, pos0, theta, kcache, vcache, cap, k8, v8, ks, vs, stream);
}
static int op_rope_kv_impl(int dtype, void* qkv, int b, int Sq, int Hq, int Hkv, int pos0, float theta, void* kcache, void* vcache, int cap,
                           void* k8, void* v8, float* ks, float* vs, void* stream) {
  OM_CHECK(pos0 + Sq <= cap, "positions exceed cache capacity");
  const int max_pos = pos0 + Sq;
  float* d = rope_table_device(max_pos, theta);
  OM_CHECK(d, "rope table allocation failed");
  RopeArgs r{qkv, (Hq + 2 * Hkv) * 128, b * Sq, Sq, Hq, Hkv, nullptr, pos0, d, max_pos, kcache, vcache, (int64_t)Hkv * cap * 128, (int64_t)cap * 128};
  r.k8 = k8; r.v8 = v8; r.ks = ks; r.vs = vs; r.s_sb = (int64_t)Hkv * cap; r.s_sh = cap;
  int rc = launch_rope_kv(dtype, r, S(stream));
  hipStreamSynchronize(S(stream));
  hipFree(d);
  return rc;
}

extern "C" int omchat_op_argmax(const float* logits, int b, int V, int32_t* out, void* stream) {
  void* scratch = nullptr;
  OM_HIP(hipMalloc(&scratch, argmax_scratch_bytes(b)));
  int rc = launch_argmax(logits, V, b, V, out, scratch, S(stream));
  hipStreamSynchronize(S(stream));
  hipFree(scratch);
  return rc;
}

extern "C" int omchat_op_fill_uniform(int dtype, void* dst, int64_t n, uint64_t key, float scale, float offset, void* stream) {
  return launch_fill_uniform(dtype, dst, n, key, scale, offset, S(stream));
}

// FlashAttention.forward(qkv[B,S,3,H,D]) -> out[B,S,H,D]  (intern_vit_6b/flash_attention.py:30-75); D = 128 only.
extern "C" int omchat_mha_fwd(const void* qkv, int B, int Sq, int H, float softmax_scale, int causal, void* out, int dtype, void* stream) {
  OM_CHECK(qkv && out, "null argument");
  AttnArgs a{};
  const int64_t row = (int64_t)3 * H * 128;
  a.Q = qkv; a.q_sb = Sq * row; a.q_sh = 128; a.q_sr = row;
  a.K = (const char*)qkv + (size_t)H * 128 * 2; a.k_sb = a.q_sb; a.k_sh = 128; a.k_sr = row;
  a.V = (const char*)qkv + (size_t)2 * H * 128 * 2; a.v_sb = a.q_sb; a.v_sh = 128; a.v_sr = row;
  a.O = out; a.o_sb = (int64_t)Sq * H * 128; a.o_sh = 128; a.o_sr = (int64_t)H * 128;
  a.batch = B; a.q_heads = H; a.kv_heads = H; a.Sq = Sq; a.Skv = Sq; a.kv_len = nullptr; a.causal = causal; a.q_pos0 = 0;
  a.scale = softmax_scale > 0.f ? softmax_scale : 0.08838834764831845f;
  return launch_attn_prefill(dtype, a, S(stream));
}

// packed qkv [B, S, 3, H, D] (D = 128 or 64) with optional per-sequence valid lengths (keys >= seqlens[b] masked; the caller zeroes
// the rows of padded queries, as flash-attn's pad_input does)
extern "C" int omchat_mha_fwd_varlen(const void* qkv, int B, int Sq, int H, int D, const int32_t* seqlens, float softmax_scale, int causal,
                                     void* out, int dtype, void* stream) {
  OM_CHECK(qkv && out, "null argument");
  OM_CHECK(D == 128 || D == 64, "head_dim must be 128 or 64");
  AttnArgs a{};
  const int64_t row = (int64_t)3 * H * D;
  a.Q = qkv; a.q_sb = Sq * row; a.q_sh = D; a.q_sr = row;
  a.K = (const char*)qkv + (size_t)H * D * 2; a.k_sb = a.q_sb; a.k_sh = D; a.k_sr = row;
  a.V = (const char*)qkv + (size_t)2 * H * D * 2; a.v_sb = a.q_sb; a.v_sh = D; a.v_sr = row;
  a.O = out; a.o_sb = (int64_t)Sq * H * D; a.o_sh = D; a.o_sr = (int64_t)H * D;
  a.batch = B; a.q_heads = H; a.kv_heads = H; a.Sq = Sq; a.Skv = Sq; a.kv_len = seqlens; a.causal = causal; a.q_pos0 = 0;
  a.scale = softmax_scale > 0.f ? softmax_scale : (D == 128 ? 0.08838834764831845f : 0.125f);
  a.head_dim = D;
  return launch_attn_prefill(dtype, a, S(stream));
}

extern "C" int omchat_op_quant_fp8(int dtype, const void* W, int N, int K, void* W8, float* scale, void* stream) {
  return launch_quant_fp8_rows(dtype, W, K, N, K, W8, K, scale, S(stream));
}

extern "C" int omchat_op_gemv_fp8(int dtype, const void* X, const void* W8, const float* scale, void* Y, int N, int K, const void* bias,
                                  const void* resid, int epi, int out_f32, int ksplit, void* stream) {
  OM_CHECK(scale, "scale missing");
  GemvArgs g{X, K, W8, K, Y, N, 1, N, K, bias, resid, N, epi, out_f32, ksplit, 0, scale};
  return launch_gemv(dtype, g, S(stream));
}

extern "C" int omchat_op_attn_prefill_d(int dtype, const void* q, const void* k, const void* v, void* out, int b, int Sq, int Skv, int Hq,
                                        int Hkv, int D, const int32_t* kv_len, int causal, int q_pos0, float scale, void* stream) {
  AttnArgs a{};
  a.Q = q; a.q_sb = (int64_t)Sq * Hq * D; a.q_sh = D; a.q_sr = (int64_t)Hq * D;
  a.K = k; a.k_sb = (int64_t)Hkv * Skv * D; a.k_sh = (int64_t)Skv * D; a.k_sr = D;
  a.V = v; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = D;
  a.O = out; a.o_sb = a.q_sb; a.o_sh = D; a.o_sr = a.q_sr;
  a.batch = b; a.q_heads = Hq; a.kv_heads = Hkv; a.Sq = Sq; a.Skv = Skv; a.kv_len = kv_len; a.causal = causal; a.q_pos0 = q_pos0; a.scale = scale;
  a.head_dim = D;
  return launch_attn_prefill(dtype, a, S(stream));
}

extern "C" int omchat_op_layernorm(int dtype, const void* x, const void* w, const void* b, void* y, int rows, int H, float eps, void* stream) {
  return launch_layernorm(dtype, x, H, w, b, y, H, rows, H, eps, S(stream));
}
