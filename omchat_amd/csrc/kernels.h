// Internal launcher interface (C++) shared by the op-level C ABI (capi.cpp) and the model loops (model.cpp).
// Every launcher enqueues on `stream` and returns 0 / sets omchat_last_error().  dtype: OMCHAT_F16 | OMCHAT_BF16.
#pragma once
// Round-4 one-launch forms of the batch-1 decode layer (decode_layer.hip, fused_decode.hip) and the work-stealing gate|up GEMV: measured
// SLOWER than the launches they replace and kept for the record only.  They are compiled in with
// -DOMCHAT_EXPERIMENTS=1 (`python -m omchat_amd.build --twin ab_lib/experiments -DOMCHAT_EXPERIMENTS=1`); the product library has stubs
// that refuse, tuning keys 22 / 23 / 24 then do nothing, and omchat_has_experiments() reports which build is loaded.
#ifndef OMCHAT_EXPERIMENTS
#define OMCHAT_EXPERIMENTS 0
#endif
#include "common.h"
#if OMCHAT_EXPERIMENTS
// TIMING PROBE ONLY (tuning key 41, experiments build): every kernel launch of the library goes out with hipExtAnyOrderLaunch.  On this GPU the flag does not
// let a kernel overtake its predecessor (tools/experiments/tune_anyorder.hip: it still starts after the predecessor's last wave) but the boundary shrinks from 2.5 to
// 0.3 us -- the release / acquire cache maintenance between the two is what goes.  Without it a consumer on another XCD may read stale lines, so results
// under this key are NOT valid; it prices what boundaries without cache maintenance would return.
#include <hip/hip_ext.h>
extern int g_launch_any_order;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                                      \
  do {                                                                                                                                         \
    if (g_launch_any_order)                                                                                                                    \
      hipExtLaunchKernelGGL((kernelName), dim3(numBlocks), dim3(numThreads), (memPerBlock), (streamId), nullptr, nullptr, hipExtAnyOrderLaunch, \
                            __VA_ARGS__);                                                                                                      \
    else                                                                                                                                       \
      hipLaunchKernelGGLInternal((kernelName), numBlocks, numThreads, memPerBlock, streamId, __VA_ARGS__);                                     \
  } while (0)
#endif

// ------------------------------------------------------------------------------------------------ GEMM
// C[M,N] = epilogue(A[M,K] @ W[N,K]^T), fp32 accumulate on MFMA.  K % 64 == 0, lda/ldw % 8 == 0.
enum {
  EPI_NONE = 0,      // C = T(acc + bias?)
  EPI_GELU = 1,      // C = T(gelu_erf(T(acc + bias?)))                       (InternMLP fc1, projector linear_1)
  EPI_LS_RESID = 2,  // C = T(resid + T(T(acc + bias?) * ls))                 (ViT: x + branch * ls)
  EPI_RESID = 3,     // C = T(resid + T(acc + bias?))                         (decoder o_proj / down_proj)
  EPI_SWIGLU = 4,    // W rows interleaved [16 gate | 16 up]: C[M,N/2] = T(T(silu(T(g))) * T(u))
  EPI_PARTIAL = 5,   // skinny GEMM only: raw fp32 K-slice sums Y[ksplit][b][ldy] (split-K across workgroups)
  EPI_F32OUT = 6,    // MFMA GEMM only: C is float [M, ldc]: the raw fp32 accumulators (tensor parallelism: row-parallel partial sums that are
                     // all-reduced in fp32 and finished by launch_tp_finish; ldc % 4 == 0, C 16-byte aligned)
  // round 6 (the ViT's norms folded into its GEMMs): the same epilogues that ALSO leave, per row and per wave-column block ("slot"), the sum of
  // squares of the 16-bit values they store -- GemmArgs.stats, SLOT-MAJOR fp32 [slots][stats_ld >= M], slot = first column of the wave tile / its width.  The
  // consumer (the next GEMM's row scale, the ViT q / k norm) sums a row's slots in slot order: a fixed order, so the statistics are deterministic
  EPI_LS_RESID_STATS = 7,
  EPI_NONE_STATS = 8,
};
struct GemmArgs {
  const void* A; int lda;
  const void* W; int ldw;
  void* C; int ldc;
  int M, N, K;
  const void* bias;   // [N] or null
  const void* ls;     // [N] layer-scale (EPI_LS_RESID)
  const void* resid;  // [M, ldr] (EPI_LS_RESID / EPI_RESID); may alias C
  int ldr;
  int epi;
  int force_tile;     // 0 auto, 1 = 128x128, 2 = 256x256 (staggered 4-phase), 3 = 256x192, 4 = 224x256, 5 = 192x256, 6 = 256x256 (one barrier)
  void* sk_ws; size_t sk_ws_bytes;   // stream-K workspace (gemm_sk_ws_bytes()), null = data-parallel only
  int stream_k;       // 0 auto (when a workspace is given), -1 never, 1 required
  // fp8 x fp8 MFMA (BASELINE configs[4], "fp8 MFMA weights"): A [M, K] and W [N, K] are OCP e4m3 bytes (lda / ldw in bytes) with one fp32
  // scale per row each; C = epi(a_scale[m] * w_scale[n] * sum_k A8 W8) in the 16-bit type.  K % 128 == 0.  256x256 tile, 128-deep K
  // steps: the same LDS bytes per step as the 16-bit kernel feed twice the MFMAs.
  int f8;
  const float* a_scale; const float* w_scale;
  // round 6: C = epi(row_scale[m] * (A W^T) + bias ...): the RMSNorm in front of a ViT projection as a per-row factor on the fp32 accumulators
  // (x * rsqrt(mean x^2 + eps) commutes with the product; the norm WEIGHT is folded into W's columns at load time: model.hip fold_norm_weights)
  const float* row_scale = nullptr;
  // ... or finished inside the launch from the producing GEMM's statistics slots: row_scale[m] = rsqrt(sum_{s < rs_nslots} rs_stats[s * rs_ld + m] / rs_dim + rs_eps)
  // (once per tile, into LDS, after the K loop: no launch between the producer and this GEMM)
  const float* rs_stats = nullptr; int rs_ld = 0, rs_nslots = 0, rs_dim = 0; float rs_eps = 0.f;
  // EPI_*_STATS: per-row partial sums of squares of the stored outputs (slot-major); *stats_nslots receives the number of slots the launch(es) wrote
  float* stats = nullptr; int stats_ld = 0; int* stats_nslots = nullptr;
};
int launch_gemm(int dtype, const GemmArgs& a, hipStream_t stream);
// finishes the per-slot partials of an EPI_*_STATS GEMM (or of launch_row_sumsq); statistics are SLOT-MAJOR, stats[slot * ld + m], ld >= rows:
// for group g < ngroups, t_g = sum_{s < nslots} stats[(slot0 + g * nslots + s) * ld + m] in slot order;  dim > 0: out[m * ngroups + g] = rsqrt(t_g / dim + eps) (a GEMM row scale),
// dim == 0: out[m * ngroups + g] = t_g (the [rows, 2] sums launch_vit_qknorm and the attention's q norm take)
int launch_stats_finish(const float* stats, int ld, int slot0, int nslots, int ngroups, int rows, int dim, float eps, float* out, hipStream_t s);
// stats[m] = sum_c x[m][c]^2 (slot 0 of slot-major statistics): the statistics of a residual stream that no GEMM epilogue produced (the ViT's embeddings)
int launch_row_sumsq(int dtype, const void* x, int ldx, int rows, int H, float* stats, hipStream_t s);
size_t gemm_sk_ws_bytes();
void gemm_set_skew(int v);
void gemm_set_persist(int v);
void gemm_set_wide_store(int v);      // tuning key 37
void gemm_set_skip_dead(int v);       // tuning key 43

// ------------------------------------------------------------------------------------------------ skinny GEMM (decode)
// Y[b,N] = epilogue(X[b,K] @ W[N,K]^T) for b <= 16: weight-streaming, HBM-bound.  K % 64 == 0.
struct GemvArgs {
  const void* X; int ldx;
  const void* W; int ldw;
  void* Y; int ldy;         // T, or float when out_f32
  int b, N, K;
  const void* bias;
  const void* resid; int ldr;
  int epi;                  // EPI_NONE | EPI_RESID | EPI_SWIGLU | EPI_PARTIAL
  int out_f32;
  int ksplit;               // K slices across workgroups (EPI_PARTIAL), <= 1 = none
  int force_mfma;           // 1 = always the MFMA form (tests / A-B); default: b == 1 uses the whole-row streaming form
  const float* w_scale;     // non-null: W is OCP e4m3 bytes [N][ldw] with one fp32 scale per row (weight-only fp8, b == 1 only)
  // batched decode (b <= 32) with operands in MFMA fragment order (common.h: packed_x_index / packed_w_index): every wave load is 1 KiB
  // contiguous instead of 16 rows x 64 B (measured, tools/experiments/tune_gemv32.hip: gate|up at b = 32 84 -> 53 us, down 50 -> 28 us)
  int x_packed;             // X is packed with NB = b > 16 ? 2 : 1 (ldx ignored)
  int w_packed;             // W is the packed replica (ldw ignored); needs x_packed
  int y_packed;             // EPI_SWIGLU only: write Y in the packed x layout of the consumer (same NB; ldy ignored)
  // b == 1, ksplit <= 1, whole-row form: X is the RAW hidden row and y = epi(W RMSNorm(X; norm_w, norm_eps)) -- the norm runs in registers
  const void* norm_w; float norm_eps;
  // optional: 65 x 64 unsigned (65 lines of 256 bytes) of ZEROED device memory owned by the caller and used by no other launch at the same time (the kernel leaves them
  // zero): the loop form of a norm GEMV then takes its outputs from atomic work counters (gemv_rows_norm_dyn_kernel) instead of equal shares
  void* dyn_ctr = nullptr;
  // batched x-stationary form, EPI_RESID (round 5): the result rows are ALSO written in the packed x layout here (same NB), un-normalised --
  // the consuming gate|up GEMV normalises them in registers (norm_w with x_packed)
  void* y_pack = nullptr;
  // experiments build, measurement only: 8 x 64-bit clock stamps of this launch's layer (see model.hip dbg_stamps)
  unsigned long long* dbg = nullptr;
};
// row-major [rows <= 32][K] -> packed x (tests, tools); row-major W [N][ldw] -> packed replica (N % 16 == 0)
int launch_pack_x(int dtype, const void* X, int ldx, int b, int K, void* out, hipStream_t s);
int launch_pack_w(int dtype, const void* W, int ldw, int N, int K, void* out, hipStream_t s);
int launch_gemv(int dtype, const GemvArgs& a, hipStream_t stream);
// experiments build (tuning key 42): the batch-1 o_proj GEMV launched out of order behind the split-KV merge, waiting on its per-head completion flags
int launch_gemv_wait(int dtype, const GemvArgs& a, const unsigned* flags, unsigned epoch, int nflags, unsigned* err, int mode, hipStream_t s);
// per-row (output channel) symmetric quantisation of W [N][ldw] (T) to OCP e4m3: scale[n] = absmax_n / 448 (1 if the row is 0),
// W8[n][k] = e4m3_rne(W[n][k] / scale[n])
int launch_quant_fp8_rows(int dtype, const void* W, int ldw, int N, int K, void* W8, int ld8, float* scale, hipStream_t stream);
void gemv_set_force_mfma(int v);
int gemv_get_force_mfma();
void gemm_set_autotune(int v);
int gemm_tune_load(const char* path);
int gemm_tune_dump(const char* path);
long gemm_tune_runs();
void model_set_ar_min_rows(int v);
void model_set_tp_f32(int v);
void model_set_pack_replica(int v);
void model_set_norm_in_gemv(int v);
void model_set_vit_fused(int v);        // tuning key 44
void model_set_tp_sp(int v);            // tuning key 45
// out[r][c] = T(W[r][c] * n[c]): a norm weight folded into the columns of the linear map that follows it (model.hip ensure_vit_folded)
int launch_fold_cols(int dtype, const void* W, const void* n, void* out, int rows, int cols, hipStream_t s);

// ------------------------------------------------------------------------------------------------ norms
// y = T(w * T(x * rsqrt(mean(x^2) + eps)))  (InternRMSNorm / Qwen2RMSNorm), rows of width H (H % 8 == 0, H <= 16384)
int launch_layernorm(int dtype, const void* x, int ldx, const void* w, const void* b, void* y, int ldy, int rows, int H, float eps, hipStream_t stream);
// pack_nb != 0: y is written in the packed x layout (common.h) with NB = pack_nb instead of row-major (rows <= 16 * pack_nb)
int launch_rmsnorm(int dtype, const void* x, int ldx, const void* w, void* y, int ldy, int rows, int H, float eps, hipStream_t s, int pack_nb = 0);
// RMSNorm whose output goes to an fp8 x fp8 GEMM: y8[row] = e4m3(T(w * T(x * rsqrt(..))) / s_row), s_row = absmax / 448 (per token)
int launch_rmsnorm_q8(int dtype, const void* x, int ldx, const void* w, void* y8, int ldy, float* scale, int rows, int H, float eps, hipStream_t s);
// per-row e4m3 quantisation of an activation matrix [rows, H] (16-bit) -> bytes + one fp32 scale per row
int launch_quant_rows_q8(int dtype, const void* x, int ldx, void* y8, int ldy, float* scale, int rows, int H, hipStream_t s);
// decode: x = T(x + T(sum_s part[s])) in place, then xn = rmsnorm(x) * w (w == null: skip the norm).  part fp32 [ks][rows][H]
int launch_resid_rmsnorm(int dtype, void* x, int ldx, const float* part, int ks, const void* w, void* xn, int ldn, int rows, int H, float eps,
                         hipStream_t s, int pack_nb = 0);
// sequence-parallel tensor parallelism (round 6): x = T(x + y) in place on `rows` rows (y: the 16-bit sum of a row-parallel projection's partials),
// then xn = RMSNorm(x) * w (b == null), LayerNorm(x) * w + b, or nothing (w == null)
int launch_resid16_norm(int dtype, void* x, int ldx, const void* y, int ldy, const void* w, const void* b, void* xn, int ldn, int rows, int H, float eps,
                        hipStream_t s);
// ViT joint-head q/k RMSNorm in place on the fused qkv buffer [rows, 3C] (q = cols [0,C), k = [C,2C)); q is also
// multiplied by q_scale with the reference's rounding (modeling_intern_vit.py:143-148).
// sumsq_in: optional [rows,2] fp32 externally reduced sum of squares (tensor parallel); C_total = divisor.
int launch_vit_qknorm(int dtype, void* qkv, int ld, const void* wq, const void* wk, int rows, int C, int C_total,
                      float eps, float q_scale, const float* sumsq_in, hipStream_t s, int only_k = 0);      // only_k: the K half alone (round 6: Q is normed where the attention loads it)
int launch_vit_qk_sumsq(int dtype, const void* qkv, int ld, int rows, int C, float* sumsq_out, hipStream_t s);
// round 6: the K half of that norm straight from the qkv GEMM's statistics slots (slot-major stats [2 * nslots][stats_ld >= rows]: q slots [0, nslots),
// k slots [nslots, 2 nslots), nslots <= 64): k (the K columns of the fused buffer, row stride ld) = T(w_k * T(k * rsqrt(sum of the k slots / C_total + eps))) in place, and
// sumsq_q[row] = sum of the q slots for the attention kernel's q norm on load (AttnArgs.qn_sumsq)
int launch_vit_knorm_slots(int dtype, void* k, int ld, const void* wk, int rows, int C, int C_total, float eps, const float* stats, int stats_ld, int nslots,
                           float* sumsq_q, hipStream_t s);
// ------------------------------------------------------------------------------------------------ attention
struct AttnArgs {
  const void* Q; int64_t q_sb, q_sh, q_sr;     // strides in elements: batch, head, row(token)
  const void* K; int64_t k_sb, k_sh, k_sr;
  const void* V; int64_t v_sb, v_sh, v_sr;
  void* O; int64_t o_sb, o_sh, o_sr;
  int batch, q_heads, kv_heads;
  int Sq, Skv;              // common lengths; per-batch overrides below
  const int* kv_len;        // optional device [batch] (valid keys per sequence), null -> Skv
  const int* kv_start;      // optional device [batch]: keys < kv_start[b] are masked (left-padded batches), null -> 0
  int causal;               // key j visible to query i iff kv_start <= j < kv_len and (!causal or j <= i + q_pos0)
  int q_pos0;               // absolute position of query row 0 (0 for a fresh prefill)
  float scale;              // applied to scores in fp32
  int head_dim;             // 128 (0 = 128) or 64 (InternViT-300M)
  // round 6, MHA only (q_heads == kv_heads, head_dim 128): the Q half of the ViT's joint-head q / k norm applied on load -- see attn_common.h AttnP
  const float* qn_sumsq = nullptr; int qn_stride = 0, qn_dim = 0; const void* qn_w = nullptr; float qn_eps = 0.f, qn_scale = 1.f;
};
int launch_attn_prefill(int dtype, const AttnArgs& a, hipStream_t s);
// query rows i < kv_start[b] of a left-padded batch (no visible key): O = sum_j T(1 / Skv) * V[j] over ALL Skv keys, the reference's eager
// attention on a fully masked row (modeling_qwen2.py:150-172 with every score at finfo.min); needs a.kv_start
int launch_attn_uniform_rows(int dtype, const AttnArgs& a, hipStream_t s);
void attn_set_v2(int v);
void model_set_fuse_peer_norm(int v);
void attn_set_tpw(int v);
void attn_set_klds(int v);
void attn_set_dma(int v);
void attn_set_dma_slots(int v);
void attn_set_dma_rot(int v);
void attn_set_hsplit(int v);
void attn_set_mha_xcd(int v);
void attn_set_kg(int v);      // tuning key 36
void attn_set_peel(int v);    // tuning key 46
void attn_set_kv8_tpw(int v); // tuning key 47
void attn_set_kv8_fuse(int v); // tuning key 48
bool attn_decode_kv8_fuses_rope(int batch, int kv_heads, int L, bool masked);
void attn_set_merge_mid_min(int v);
void attn_set_merge_dg(int v);
void gemv_set_norm_loop(int v);
void gemv_set_dyn(int v);
void gemv_set_skew(int v);
void gemv_set_rows_balance(int v);
void gemv_set_no_xs(int v);
void gemv_set_shard_shapes(int v);      // tuning key 34
void gemv_set_gu_rr(int v);             // tuning key 38
void gemv_set_longk_direct(int v);      // tuning key 39
void norm_set_wave(int v);              // tuning key 40
void model_set_ao_oproj(int v);         // tuning key 42
int gemv_get_shard_shapes();

// decode: one query token per sequence, q heads grouped per kv head; split-KV partials + merge.
struct AttnDecodeArgs {
  const void* Q; int64_t q_sb, q_sh;            // [b, q_heads, 128]
  const void* K; int64_t k_sb, k_sh, k_sr;      // cache [b, kv_heads, L, 128]
  const void* V; int64_t v_sb, v_sh, v_sr;
  void* O; int64_t o_sb, o_sh;                  // [b, q_heads, 128]
  int batch, q_heads, kv_heads;
  int L;                                        // kv length used when kv_len == null
  const int* kv_len;                            // optional device [batch]; null = every sequence holds exactly L keys (one dependent load less)
  float scale;
  float* ws; size_t ws_bytes;                   // workspace for partials
  // optional fused RoPE + KV append of the token being decoded (replaces rope_kv for S = 1): q is rotated in registers,
  // the split that owns position pos[b] rotates k, appends k/v to the cache and uses them from LDS
  const float* rope; int rope_max;              // cos/sin table [max_pos][64][2] or null (= q, K, V already rotated/appended)
  const int* pos;                               // device [batch]; kv_len[b] must be pos[b] + 1
  const void* k_new; const void* v_new; int64_t new_sb;   // raw k / v of the new token: [b][kv_heads*128] views, batch stride
  int o_pack_nb;                                // != 0: O is written as packed x ([b][q_heads*128] rows, common.h) for the o_proj GEMV
  // fp8 KV cache (BASELINE configs[4]): K / V point at OCP e4m3 bytes in the SAME [b, kv_heads, L, 128] layout (strides in elements),
  // with one fp32 scale per (sequence, kv head, key): k_scale / v_scale [b][kv_heads][scale_cap]; rope must be null (the new token is
  // rotated, appended and quantised before the call)
  const float* k_scale; const float* v_scale; int64_t scale_sb, scale_sh;
  // ... unless attn_decode_kv8_fuses_rope(batch, kv_heads, L, masked) (round 6): then `rope`, `k_new`, `v_new` are given as for the 16-bit cache
  // and the launch rotates q / k itself, appends the new rows to the e4m3 cache + scales (written through K / V / k_scale / v_scale) and to the
  // 16-bit cache k16_w / v16_w (same [b, kv_heads, L, 128] strides) -- rope_kv_kernel's bytes, no launch in front
  void* k16_w = nullptr; void* v16_w = nullptr;
  // padded batch as the reference decodes it (omchat_decode_step_masked): key j of sequence b is visible iff key_mask[b * mask_sb + j] != 0
  // (device bytes, rows zero-padded to mask_sb % 64 == 0); `pos` then holds the RoPE positions (not kv_len - 1); kv_len must be null
  const unsigned char* key_mask; int64_t mask_sb;
  // experiments build (tuning key 42): the merge launch publishes done_flags[head] = done_epoch after an agent-scope release (batch 1, <= 64 splits only)
  unsigned* done_flags = nullptr; unsigned done_epoch = 0; int done_mode = 0;
  unsigned long long* done_dbg = nullptr;      // experiments build, measurement only: clock stamp of the merge's end
};
// quantise rows [pos0, pos1) of every (sequence < b, kv head) of a 16-bit cache [b_cap, kv_heads, cap, 128] into the fp8 cache + scales
// (pos1 = null-terminated per sequence: rows >= len[b] are skipped when len != null)
int launch_kv_quant(int dtype, const void* kc, const void* vc, void* k8, void* v8, float* ks, float* vs, int b, int kv_heads, int64_t c_sb, int64_t c_sh,
                    int64_t s_sb, int64_t s_sh, const int* pos_lo, int pos0, const int* len, int max_rows, hipStream_t s);
size_t attn_decode_ws_bytes(int batch, int q_heads, int max_len);
int launch_attn_decode(int dtype, const AttnDecodeArgs& a, hipStream_t s);

// batch-1 decode on one GPU (fused_decode.hip, round 4): launch_attn_decode + the o_proj GEMV with EPI_RESID as ONE launch, same bits.
// x [H] is the residual stream (read, x + o_proj(attn) written in place), Wo [H][qd] row-major; ws = fused_decode_ws_bytes(q_heads) bytes
// of zero-initialised device memory owned by the caller and used by no other launch at the same time; epoch: a value that no earlier launch
// on the same ws has used (monotonic counter, never 0); err: device word that collects time-out bits (0 = every hand-off completed).
struct FusedDecodeArgs {
  const void* Wo; int ldw;
  void* x; int H, qd;
  void* ws; unsigned epoch;
  unsigned* err; int timeout_ms;
  void* dbg = nullptr;      // diagnostic build of tools/experiments/tune_fused.hip only: phase stamps [CUs][16]; the library never sets it
};
size_t fused_decode_ws_bytes(int q_heads);
bool attn_oproj_fused_ok(const AttnDecodeArgs& a, int H, int qd);
int launch_attn_oproj_fused(int dtype, const AttnDecodeArgs& a, const FusedDecodeArgs& f, hipStream_t s);
void model_set_fuse_attn_oproj(int v);
void model_set_decode_layer(int v);
void model_set_shard_as_tp1(int v);      // tuning key 35

// batch-1 decode on one GPU (decode_layer.hip, round 4): ONE decoder layer -- the six launches qkv GEMV (+ input RMSNorm), split-KV
// attention (+ RoPE, cache append), merge, o_proj (+ residual), gate|up GEMV (+ post-attention RMSNorm, SwiGLU), down_proj (+ residual) -- as
// ONE launch with in-launch hand-offs; the same bits.  Weights row-major as the context holds them (wgu: gate / up interleaved in 16-row
// blocks); x [H] is the residual stream, read and overwritten with the layer's output; kc / vc this layer's cache [kv_heads][cap][128] with
// head stride k_sh elements; kv_len = keys after this step (the new token at kv_len - 1).  ws = decode_layer_ws_bytes(...) bytes of
// zero-initialised device memory used by no other launch at the same time; epoch: a value no earlier launch on the same ws has used (never 0).
struct DecodeLayerArgs {
  const void *ln1, *ln2, *wqkv, *bqkv, *wo, *wgu, *wd;
  void *kc, *vc; int64_t k_sh;
  void* x;
  int H, qd, kvd, It, q_heads, kv_heads, kv_len;
  const float* rope; int rope_max;
  float eps, scale;
  void* ws; unsigned epoch;
  unsigned* err; int timeout_ms;
  void* dbg = nullptr;      // diagnostic build only
};
size_t decode_layer_ws_bytes(int q_heads, int H, int qd, int kvd, int It);
bool decode_layer_ok(const DecodeLayerArgs& a);
int launch_decode_layer(int dtype, const DecodeLayerArgs& a, hipStream_t s);

// ------------------------------------------------------------------------------------------------ RoPE + KV append
// qkv [rows, (nq + 2 nkv) * 128] post-bias; rotate q in place, write rotated k and raw v into the caches at
// position pos[row] of sequence seq[row] (rows are b-major: row = b_idx * S + s).
struct RopeArgs {
  void* qkv; int ld;
  int rows, S;                 // rows = b * S
  int nq, nkv;
  const int* pos;              // device [rows] absolute positions, or null -> pos0 + (row % S)
  int pos0;
  const float* cos_sin;        // table [max_pos][64][2] fp32 (cos, sin)
  int max_pos;
  void* kcache; void* vcache;  // [b, nkv, cap, 128]
  int64_t c_sb, c_sh;          // cache strides (elements); row stride 128
  // optional (fp8 KV cache, decode): the appended k / v rows are ALSO quantised here (what launch_kv_quant would do to them in a launch
  // of its own): e4m3 bytes into k8 / v8 (same [b, nkv, cap, 128] layout, strides in elements = bytes), one fp32 scale per row into
  // ks / vs [b][nkv][s_sh]
  void* k8 = nullptr; void* v8 = nullptr; float* ks = nullptr; float* vs = nullptr; int64_t s_sb = 0, s_sh = 0;
  // cache slot of row r when it differs from the RoPE position (masked decode of a padded batch: every row appends at the COMMON slot and
  // rotates to its own position, omchat_arch.py:61-70): slot0 + (r % S); -1 = the position itself
  int slot0 = -1;
};
int launch_rope_kv(int dtype, const RopeArgs& a, hipStream_t s);

// ------------------------------------------------------------------------------------------------ data movement
// im2col for Conv2d(3->C, k=p, s=p): pixels [B,3,HW,HW] -> cols [B*g*g, Kpad] (zero padded K)
int launch_im2col(int dtype, const void* pixels, void* cols, int B, int HW, int patch, int Kpad, hipStream_t s);
// x[b, 0] = cls + pos[0];  x[b, 1+p] = pe[b*np + p] + pos[1+p]
int launch_vit_assemble(int dtype, const void* pe, const void* cls, const void* pos, void* x, int B, int np, int C, hipStream_t s);
// out[r] = table[idx[r]] (idx >= 0) | feats[-1 - idx[r]] (idx <= -1, > PAD) | 0 (idx == INT_MIN)
// out[M, N] = epi(sum[M, N] (fp32), bias, ls, resid) with the rounding points of the GEMM epilogues (EPI_NONE / EPI_RESID / EPI_LS_RESID)
int launch_tp_finish(int dtype, const float* sum, const void* bias, const void* ls, const void* resid, void* out, int M, int N, int epi, hipStream_t s);
int launch_gather_rows(int dtype, const int* idx, const void* table, const void* feats, void* out, int rows, int H, hipStream_t s);
// strided row copy (drop CLS etc.): dst[r] = src[map(r)]
int launch_copy_rows(int dtype, const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int rows, int H,
                     int group, int skip, hipStream_t s);   // src row = (r / group) * (group + skip) + skip + r % group
// argmax over fp32 logits [b, V] -> int32 [b] (first index wins ties)
size_t argmax_scratch_bytes(int b);
// adv_pos / adv_len (optional, device [b]): incremented by one in the second stage -- the decode step's position bookkeeping without a launch of its own
int launch_argmax(const float* logits, int ld, int b, int V, int* out, void* scratch, hipStream_t s, int* adv_pos = nullptr, int* adv_len = nullptr);   // scratch: argmax_scratch_bytes(b)
// deterministic synthetic fill (bit-identical to omchat_amd/synth.py::uniform)
int launch_fill_uniform(int dtype, void* dst, int64_t n, uint64_t key, float scale, float offset, hipStream_t s);
int launch_cast_f32(int dtype, const void* src, float* dst, int64_t n, hipStream_t s);
