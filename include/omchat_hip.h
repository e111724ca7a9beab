/* omchat_hip.h -- C ABI of libomchat_hip.so: the MI355X (gfx950) implementation of the OmChat inference hot path
 * (InternViT vision tower -> mlp2x_gelu projector -> image-token splice -> Qwen2 prefill -> greedy decode).
 *
 * The reference (om-ai-lab/OmChat) has no FFI of its own: its seams are Python class boundaries.  Each entry point
 * below names the reference interface it stands behind (file:line relative to the reference repo; Qwen2 lines refer
 * to transformers/models/qwen2/modeling_qwen2.py which the reference inherits at
 * omchat/model/language_model/omchat_qwen2.py:7,22,29).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions: plain pointers + sizes, no framework types.  Device pointers unless stated.  All 16-bit tensors use
 * the context's compute dtype (OMCHAT_F16 / OMCHAT_BF16), row-major.  `stream` is a hipStream_t (NULL = default
 * stream); every call only enqueues work on it.  Return 0 on success; otherwise omchat_last_error() (thread-local)
 * explains, and the Python mirror re-raises ValueError / RuntimeError like the reference does.
 */
#ifndef OMCHAT_HIP_H
#define OMCHAT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMCHAT_F16 0
#define OMCHAT_BF16 1
#define OMCHAT_F32 2            /* accepted as a SOURCE dtype by omchat_load_tensor and as an output dtype of omchat_preproc_anyres */
#define OMCHAT_PAD_ROW INT32_MIN /* splice index: zero row */

typedef struct omchat_ctx omchat_ctx;

/* Per-rank (already tensor-parallel-local) model geometry.  Field names follow InternVisionConfig
 * (multimodal_encoder/intern_vit_6b/configuration_intern_vit.py:63-83) and Qwen2Config. */
typedef struct {
  /* vision tower */
  int v_hidden;        /* hidden_size (3200) */
  int v_heads;         /* LOCAL attention heads (25 at TP=1; zero-padded shards otherwise), head_dim is 128 */
  int v_qk_channels;   /* divisor of the joint q/k RMSNorm = full hidden_size (3200) (modeling_intern_vit.py:143-146) */
  int v_mlp;           /* LOCAL intermediate_size (12800 / tp) */
  int v_layers;        /* num_hidden_layers (45) */
  int v_patch;         /* 14 */
  int v_image;         /* 448 */
  float v_eps;         /* layer_norm_eps 1e-6 */
  /* decoder */
  int t_hidden;        /* 3584 */
  int t_layers;        /* 28 */
  int t_heads;         /* LOCAL query heads */
  int t_kv_heads;      /* LOCAL kv heads */
  int t_mlp;           /* LOCAL intermediate_size */
  int t_vocab;         /* LOCAL lm_head rows */
  int t_vocab_total;   /* embed_tokens rows (replicated) */
  float t_eps;         /* rms_norm_eps */
  float rope_theta;    /* 1e6 */
  /* capacities */
  int max_seq;         /* KV positions per sequence */
  int max_batch;       /* sequences */
  int max_tiles;       /* ViT tiles per launch batch */
  int max_prefill_rows;/* b * S rows of one prefill call */
  int dtype;           /* OMCHAT_F16 | OMCHAT_BF16 */
  /* tower variant (0 = the InternViT-6B defaults, so older callers that zero the tail are unchanged):
   * InternViT-300M (intern_vit_300m/configuration_intern_vit.py:60-80) = head_dim 64, LayerNorm, no q/k norm */
  int v_head_dim;      /* 0 or 128 | 64 */
  int v_norm_type;     /* 0 = InternRMSNorm, 1 = nn.LayerNorm (weight + bias) */
  int v_no_qk_norm;    /* 0 = joint-head q/k RMSNorm (qk_normalization=True), 1 = none */
} omchat_config;

const char* omchat_last_error(void);
const char* omchat_version(void);

/* ---- loader: stands behind load_pretrained_model (omchat/model/builder.py:22-35) -------------------------------- */
/* rccl_comm: ncclComm_t of the tensor-parallel group or NULL (tp_size == 1, or a peer group is attached with omchat_ctx_set_peer). */
int omchat_ctx_create(const omchat_config* cfg, int tp_rank, int tp_size, void* rccl_comm, omchat_ctx** out);
void omchat_ctx_destroy(omchat_ctx* ctx);
/* name: omchat-native checkpoint key, e.g. "model.layers.3.mlp.gate_proj.weight";
 * data: host OR device pointer to the (rank-local) tensor, contiguous; src_dtype: ctx dtype or OMCHAT_F32. */
int omchat_load_tensor(omchat_ctx* ctx, const char* name, const void* data, const int64_t* shape, int ndim, int src_dtype);
/* fills every weight with omchat_amd/synth.py's deterministic generator (same values as the host generator). */
int omchat_fill_synthetic(omchat_ctx* ctx, uint64_t seed);
/* number of tensors still missing (0 = ready); names of missing tensors are put in omchat_last_error(). */
int omchat_weights_missing(omchat_ctx* ctx);
size_t omchat_device_bytes(omchat_ctx* ctx);

/* ---- tower: InternVITVisionTower.forward + feature_select (internVIT_encoder.py:35-56) -------------------------- */
/* pixels [n_tiles,3,v_image,v_image]; out [n_tiles, ntok, v_hidden], ntok = patches (+1 when keep_cls).
 * select_layer indexes hidden_states like the reference (0 = embeddings, -1 = last layer). */
int omchat_vit_forward(omchat_ctx* ctx, const void* pixels, int n_tiles, int select_layer, int keep_cls, void* out, void* stream);
/* ---- projector: build_vision_projector('mlp2x_gelu').forward (multimodal_projector/builder.py:54-61) ------------ */
int omchat_projector_forward(omchat_ctx* ctx, const void* in, int rows, void* out, void* stream);
/* ---- encode_images (omchat_arch.py:50-53): tower (select_layer, patch features) then projector ------------------ */
int omchat_encode_images(omchat_ctx* ctx, const void* pixels, int n_tiles, int select_layer, void* out, void* stream);

/* ---- splice: prepare_inputs_labels_for_multimodal (omchat_arch.py:55-209) --------------------------------------- */
/* Host-side plan (integers only).  ids int64 [b,T] with -200 sentinels; mask uint8 [b,T] or NULL; n_tok rows per
 * tile; padding_side 0 = right, 1 = left; max_length <= 0 = none.  Writes src_index int32 [b * S_out] (>= 0: token
 * id, <= -1: feature row -1-k, OMCHAT_PAD_ROW: zero row), lengths int32 [b]; returns S_out via *S_out.
 * Call with src_index == NULL to size the output.  n_tiles_avail is checked like the reference's running index.
 * vocab > 0: an unmasked id outside [0, vocab) other than the -200 sentinel returns 4 (the reference raises IndexError from
 * embed_tokens, omchat_arch.py:139); vocab <= 0 skips the check. */
int omchat_splice_plan(const int64_t* ids, const uint8_t* mask, int b, int T, int n_tok, int n_tiles_avail,
                       int padding_side, int max_length, int32_t* src_index, int32_t* lengths, int* S_out, int vocab);
/* Device gather: embeds[r] = embed_tokens[idx] | feats[row] | 0.  src_index device int32 [rows]. */
int omchat_splice_gather(omchat_ctx* ctx, const int32_t* src_index, const void* feats, void* embeds, int rows, void* stream);

/* ---- decoder: OmChatQwen2ForCausalLM.forward (omchat_qwen2.py:45-89) -> Qwen2ForCausalLM.forward --------------- */
/* Prefill step 0: embeds [b, S, t_hidden] (right-padded rows), lengths host int32 [b].  Resets the KV cache of
 * sequences 0..b-1.  logits_last: device fp32 [b, t_vocab] for the last valid position of each sequence (or NULL).
 * hidden_out: optional [b, S, t_hidden] post-final-norm hidden states (test hook).  */
int omchat_prefill(omchat_ctx* ctx, const void* embeds, int b, int S, const int32_t* lengths, float* logits_last,
                   void* hidden_out, void* stream);
/* Prefill of a LEFT-padded batch (config.tokenizer_padding_side == "left", omchat_arch.py:176-184): row i holds its lengths[i] tokens at
 * [S - lengths[i], S).  As in the reference, position_ids are dropped (:206-207) so RoPE runs on arange(S) for every row, the padded
 * keys are masked, and logits_last is the position S - 1 of every row (what generate reads).  A padded query row sees no key; its
 * attention output is the uniform average of the sequence's V rows, as the reference's eager CPU attention computes it (every score at
 * finfo.min).  Decode steps after it go through omchat_decode_step_masked (omchat_decode_step would place the rows per sequence and is
 * refused). */
int omchat_prefill_left(omchat_ctx* ctx, const void* embeds, int b, int S, const int32_t* lengths, float* logits_last,
                        void* hidden_out, void* stream);
/* Decode step >= 1 (omchat_qwen2.py:92-111, omchat_arch.py:61-70): one token per sequence, appended at kv_len.
 * tokens device int32 [b]; logits device fp32 [b, t_vocab] or NULL; next_tokens device int32 [b] (greedy argmax,
 * first index wins) or NULL. */
int omchat_decode_step(omchat_ctx* ctx, const int32_t* tokens, int b, float* logits, int32_t* next_tokens, void* stream);
/* Decode step of a PADDED batch exactly as the reference computes it (omchat_arch.py:61-70 as HF generate drives it, `images` passed on
 * every step): the new token of EVERY row is appended at the common cache length (S of the padded prefill + the masked steps so far), is
 * rotated to positions[i] (host int32 [b]; the reference's sum(attention_mask) - 1) and attends the cache slots j <= slots with
 * key_mask[i * mask_ld + j] != 0 (host bytes [b][mask_ld], mask_ld >= slots + 1: the token-level mask padded with ones -- it hides real
 * prompt slots and exposes padded ones once images expanded the rows differently).  Valid after omchat_prefill (right padding) and
 * omchat_prefill_left, for all b rows of that prefill, on one GPU (16-bit or e4m3 KV cache); omchat_decode_step and this entry cannot be
 * mixed after one prefill (they place the cache rows differently).  Synchronises the stream. */
int omchat_decode_step_masked(omchat_ctx* ctx, const int32_t* tokens, int b, const int32_t* positions, const uint8_t* key_mask, int mask_ld,
                              float* logits, int32_t* next_tokens, void* stream);
/* The same step WITHOUT per-step host data, for the loop HF generate drives (single_inference.py:53-62): there the decode branch pads the
 * token-level mask with ones up to the cache length and every generated token appends another one (omchat_arch.py:63-69), so over the cache
 * slots the key mask of every step is [the prompt's token-level mask | ones] and position_ids = sum(mask) - 1 grows by one per step.
 * omchat_masked_decode_begin (once after the prefill; synchronises): key_mask host bytes [b][mask_ld], the first mask_cols columns are taken,
 * every slot behind them counts as visible; positions host int32 [b] = the FIRST step's position_ids.  omchat_decode_step_masked_next (every
 * step; no host buffer, no synchronisation): appends at the common slot, rotates to the device-resident positions and advances them.
 * omchat_kv_rewind takes the positions back with the slots.  A call of omchat_decode_step_masked in between needs a new begin. */
int omchat_masked_decode_begin(omchat_ctx* ctx, int b, const int32_t* positions, const uint8_t* key_mask, int mask_ld, int mask_cols, void* stream);
int omchat_decode_step_masked_next(omchat_ctx* ctx, const int32_t* tokens, int b, float* logits, int32_t* next_tokens, void* stream);
/* Experimental one-launch forms of the batch-1 decode layer (NOT in the product build since round 5: compile with -DOMCHAT_EXPERIMENTS=1;
 * measured slower than the six launches): with tuning key 23 a decoder layer is ONE launch with in-launch hand-offs (csrc/decode_layer.hip), with key 22 attention +
 * merge + o_proj are one launch (csrc/fused_decode.hip); same bits as the separate launches either way.  launches: how many such launches
 * this context has issued; timeout_bits: sticky bits of hand-offs that gave up after their wall-clock budget (0 = none; otherwise the
 * affected steps' results are wrong).  Synchronises. */
int omchat_fused_status(omchat_ctx* ctx, long* launches, unsigned* timeout_bits);
/* 1 when the library was built with -DOMCHAT_EXPERIMENTS=1 (those one-launch forms and the work-stealing gate|up GEMV compiled in), 0 for
 * the product build, in which tuning keys 22 / 23 / 24 select nothing. */
int omchat_has_experiments(void);
/* lm_head on arbitrary hidden rows (Qwen2ForCausalLM.forward :462-465): hidden [n, t_hidden] -> fp32 [n, t_vocab] */
int omchat_lm_head(omchat_ctx* ctx, const void* hidden, int n, float* logits, void* stream);
/* greedy pick (HF generate with do_sample=False: argmax of the last position, first index wins): logits fp32
 * [b, t_vocab] (rank-local slice under tensor parallelism; the (max, index) pairs are exchanged) -> int32 [b] */
int omchat_greedy(omchat_ctx* ctx, const float* logits, int b, int32_t* next_tokens, void* stream);
int omchat_kv_lengths(omchat_ctx* ctx, int32_t* out, int b);      /* host copy of the current KV lengths */
/* take back the last n decode steps of sequences 0..b-1 (generate() enqueues step k + 1 before it has read token k on the host, as the
 * reference's HF loop cannot; when token k ends the generation -- EOS, a stopping criterion -- that step is forgotten).  Synchronises. */
int omchat_kv_rewind(omchat_ctx* ctx, int b, int n, void* stream);

/* ---- decode step as a hipGraph ------------------------------------------------------------------------------------ */
/* With on != 0, omchat_decode_step on a TP = 1 context (b <= 32) replays one captured graph per step instead of issuing its
 * ~230 kernel launches (same kernels, same results: tests compare bit for bit).  Captured on a context-owned stream that is
 * ordered after / before the caller's `stream` with events; re-captured when a sequence outgrows the captured split-KV grid
 * (every 1024 tokens) or the batch size / fp8 mode changes.  While profiling is enabled every 8th step runs eagerly so the
 * HIP-event brackets still sample the timed region.
 * Measured on ROCm 7.2 / MI355X (bench.py --graph): replay is SLOWER than the eager stream (3.21 vs 2.96 ms per token at
 * TP = 1: graph nodes are dispatched with a barrier packet each), so it is opt-in -- useful when the host, not the GPU,
 * bounds the step (small per-rank kernels under tensor parallelism, a busy Python thread). */
int omchat_enable_decode_graph(omchat_ctx* ctx, int on);
int omchat_decode_graph_stats(omchat_ctx* ctx, long* steps, long* replays, long* captures);

/* ---- weight-only fp8 for decode (BASELINE configs[4]) -------------------------------------------- */
/* Builds (once) an OCP e4m3 replica of the decoder weights that a decode step streams (fused qkv, o, gate|up, down,
 * lm_head): per output row scale = absmax / 448, W8 = e4m3_rne(W / scale).  With on != 0, batch-1 decode steps on a
 * context (any tensor-parallel degree: each rank quantises its own shard, one scale per local output row) read these bytes (half the
 * HBM traffic of the 16-bit weights) and apply the scale after the fp32
 * row reduction; prefill and b > 1 steps keep the 16-bit weights.  Not part of the reference (it has no quantised
 * path): parity is against the oracle run on the de-quantised weights. */
int omchat_enable_fp8_decode(omchat_ctx* ctx, int on);

/* ---- fp8 KV cache and fp8 x fp8 prefill GEMMs (BASELINE configs[4]: long video context, "fp8 MFMA weights") ------ */
/* omchat_enable_fp8_kv: after the next prefill the decode steps read keys and values as OCP e4m3 bytes (57 344 -> 28 672 bytes per
 * cached token, + 2 x 4 kv heads x 28 layers fp32 scales) with one scale per (layer, sequence, kv head, position) = absmax / 448; the
 * token being decoded is rotated, appended and quantised before its attention.  The 16-bit cache stays (prefill attention reads it).
 * omchat_enable_fp8_prefill: the qkv and gate|up GEMMs of the prefill run as fp8 x fp8 MFMA: activations quantised per token by the
 * RMSNorm kernel, weights from the e4m3 replica (per output row); o_proj / down_proj keep 16-bit operands.
 * Neither is part of the reference: parity is stated against the oracle on de-quantised operands (tests/test_gpu_fp8.py). */
int omchat_enable_fp8_kv(omchat_ctx* ctx, int on);
int omchat_enable_fp8_prefill(omchat_ctx* ctx, int on);

/* ---- measurement: HIP-event timing of the dominant kernel classes, recorded on the launch stream ------------------ */
#define OMCHAT_PROF_DECODE_GATEUP 0   /* decode gate|up weight-streaming GEMV (+SwiGLU), one launch per layer per token */
#define OMCHAT_PROF_PREFILL_GATEUP 1  /* decoder prefill gate|up MFMA GEMM (+SwiGLU), one launch per layer */
#define OMCHAT_PROF_VIT_FC1 2         /* ViT fc1 MFMA GEMM (+bias+GELU), one launch per layer */
#define OMCHAT_PROF_CATS 3
int omchat_prof_enable(omchat_ctx* ctx, int on);
/* synchronises the device, folds the recorded event pairs into (total_ms, launches) for `cat`; reset != 0 clears. */
int omchat_prof_read(omchat_ctx* ctx, int cat, double* total_ms, long* launches, int reset);

/* ---- the one native op seam of the reference: FlashAttention.forward (intern_vit_6b/flash_attention.py:30-75) --- */
/* qkv packed [B, S, 3, H, 128] -> out [B, S, H, 128]; softmax_scale <= 0 means 1/sqrt(128). */
int omchat_mha_fwd(const void* qkv, int B, int S, int H, float softmax_scale, int causal, void* out, int dtype, void* stream);
/* same seam with head dim D = 128 | 64 and the key_padding_mask branch (:56-67) for right-padded batches: seqlens int32 [B] on the
 * device (NULL = all S); keys >= seqlens[b] are masked, rows of padded queries are left to the caller to zero (pad_input) */
int omchat_mha_fwd_varlen(const void* qkv, int B, int S, int H, int D, const int32_t* seqlens, float softmax_scale, int causal,
                          void* out, int dtype, void* stream);

/* ---- op-level entry points (unit parity tests, benches) --------------------------------------------------------- */
/* C[M, N] = epi(A[M, K] W[N, K]^T) on MFMA, fp32 accumulate.  Contract (checked, an error otherwise): K % 64 == 0; lda, ldw % 8 == 0 and
 * A, W 16-byte aligned; N % 4 == 0 (the epilogue owns four consecutive columns per lane); C and resid 8-byte aligned with ldc, ldr % 4 == 0
 * (8-byte epilogue accesses); bias and ls 8-byte aligned; EPI_SWIGLU: N % 32 == 0 and C is [M, N / 2]. */
int omchat_op_gemm(int dtype, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                   const void* bias, const void* ls, const void* resid, int ldr, int epi, int force_tile, void* stream);
/* Round 6: the vision tower's norms folded into its GEMMs (csrc/model.hip vit_layer_fused; InternVisionEncoderLayer.forward,
 * modeling_intern_vit.py:210-222 with InternRMSNorm :39-44 and the joint-head q / k norm :143-148).  Op-level entries of its pieces:
 * omchat_op_gemm_fused = omchat_op_gemm with (a) a per-row factor on the fp32 accumulators, C = epi(s[m] * (A W^T) ...) -- the RMSNorm in front
 * of a projection -- given either finished (row_scale [M] fp32) or as the producer's statistics slots (rs_stats, s[m] = rsqrt(sum of slots
 * [0, rs_nslots) / rs_dim + rs_eps), finished per tile inside the launch); both null = no factor; (b) epi 7 (= 2, layer-scale + residual) /
 * 8 (= 0) that also leave stats fp32: per row and per wave-column block ("slot") the sum of squares of the 16-bit values stored.  ALL statistics
 * buffers are SLOT-MAJOR: element (slot s, row m) at [s * ld + m], ld >= M (the 16 rows of an MFMA fragment are one 64-byte run for the producer,
 * and a consumer's wave reads 64 consecutive rows of a slot); *nslots = the number of slots written (one per wave tile of the tile kernel(s) that
 * ran: 64 or 112 columns each; epi 8 runs on 64-column wave tiles only, so the thirds of a fused q | k | v output own whole slots).  omchat_op_stats_finish: for group g < ngroups, t_g = sum of slots
 * [slot0 + g * nslots, slot0 + (g + 1) * nslots) of a row; dim > 0: out[m * ngroups + g] = rsqrt(t_g / dim + eps), dim == 0: out = t_g (a
 * stand-alone finishing launch: tests and callers outside the fused layer).  omchat_op_row_sumsq: stats[m] = sum_c x[m][c]^2 (slot 0).
 * omchat_op_fold_cols: out[r][c] = T(W[r][c] * n[c]) (a norm weight folded into the columns of the linear map behind it).
 * omchat_op_vit_knorm_slots: the K half of the joint-head norm from the qkv GEMM's slots (q slots [0, nslots), k slots [nslots, 2 nslots),
 * nslots <= 64): k = T(w_k * T(k * rsqrt(sum k slots / C_total + eps))) in place on rows of stride ld, sumsq_q[m] = sum of the q slots.
 * omchat_op_mha_qnorm: omchat_mha_fwd on packed qkv [B, S, 3, H, 128] whose Q is RAW -- q = T(T(w_q * T(q * rstd_q)) * q_scale) is applied
 * where the kernel loads Q, rstd_q from sumsq[(b * S + s) * stride]. */
int omchat_op_gemm_fused(int dtype, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                         const void* bias, const void* ls, const void* resid, int ldr, int epi, int force_tile, const float* row_scale,
                         const float* rs_stats, int rs_ld, int rs_nslots, int rs_dim, float rs_eps,
                         float* stats, int stats_ld, int* nslots, void* stream);
int omchat_op_stats_finish(const float* stats, int ld, int slot0, int nslots, int ngroups, int rows, int dim, float eps, float* out, void* stream);
int omchat_op_row_sumsq(int dtype, const void* x, int ldx, int rows, int H, float* stats, void* stream);
int omchat_op_fold_cols(int dtype, const void* W, const void* n, void* out, int rows, int cols, void* stream);
int omchat_op_vit_knorm_slots(int dtype, void* k, int ld, const void* wk, int rows, int C, int C_total, float eps, const float* stats, int stats_ld,
                              int nslots, float* sumsq_q, void* stream);
int omchat_op_mha_qnorm(int dtype, const void* qkv, int B, int Sq, int H, const float* sumsq, int stride, int dim, const void* wq,
                        float eps, float q_scale, void* out, void* stream);
/* same with the stream-K tail enabled: ws from omchat_op_gemm_sk_ws() bytes of device memory; stream_k 0 = auto, 1 = required */
/* experiment knobs -- PROCESS-GLOBAL test / measurement hooks shared by every context of the process: refused (error return) unless the
 * process environment has OMCHAT_ALLOW_TUNING=1; nothing in the product sets a key.  (key 0: start skew of the 256x256 GEMM workgroups, in units of ~1024 cycles; key 1: 1 = skinny GEMM
 * always takes the MFMA form, 0 = batch 1 takes the whole-row streaming form; key 4: row count from which a tensor-parallel
 * row-parallel projection is pipelined against its all-reduce in 2 chunks (3x: 4 chunks), default 1024; key 5: 0 = GEMM tile
 * shapes from the cost model instead of the first-use measurement; key 6: 0 = batched decode steps (2 <= b <= 32) read the
 * row-major weights instead of building the packed replica (+ one more copy of the decoder weights); key 8: 0 = first-generation
 * 16x16x32 prefill attention kernel; key 9: 0 = tensor-parallel decode over the peer transport keeps the all-reduce of the split-K
 * slices and the residual + RMSNorm as two launches instead of omchat_peer_resid_rmsnorm; key 10: key tiles (of 64) per wave of the
 * split-KV decode attention, 0 = chosen from the grid size (1 for single sequences, up to 4 for large batches); key 11: 1 = the batched
 * decode GEMV never takes its x-stationary persistent form; key 12: 1 = the batched decode attention loads its K tiles as whole rows
 * through LDS instead of fragment-shaped straight to registers (same bits; measured neutral); key 13: 1 = multi-round 256x256 GEMMs
 * take the persistent one-workgroup-per-CU form that overlaps the next tile's prologue with the epilogue (same bits; measured neutral);
 * key 14: batch-1 decode, bit 0 = the post-attention RMSNorm runs inside the gate|up GEMV, bit 1 = the input / final RMSNorm inside the
 * qkv / lm_head GEMV with down_proj un-split (default 3; 0 = both residual + RMSNorm launches stay); bit 2 (experiments build only; off by
 * default in BOTH builds, so the twin's batched decode is the product's) = batched steps (2 <= b <= 32) take the seven-launch layer -- o_proj
 * un-split with the residual in its epilogue, the post-attention norm in the x-stationary gate|up GEMV -- measured slower: 4.57-4.87 vs 4.28 ms;
 * key 16: which of those norm-in-GEMV launches take the loop form (one resident round of workgroups, three register buffers per wave):
 * bit 0 = gate|up, bit 1 = qkv, bit 2 = e4m3 gate|up, bit 3 = lm_head (default 0 since round 5 -- key 38; every form gives the same bits);
 * key 17: 1 (default) = batch-1 o_proj / qkv launches whose rows deal evenly to two workgroups per CU use N / (2 CUs) waves per workgroup;
 * key 19: split-KV merges with more partials per head than this take the 512-thread form (default 64: at 57 partials the 128-thread
 * one-batch form is faster, 4.9 vs 5.9 us);
 * key 21: column groups per head in the split-KV merge: 1 (default) = four workgroups per head beyond 256 partials (33 k keys: 21 -> ~7 us
 * per launch), 2 = also two workgroups per head for 65..256 partials (neutral), 0 = one workgroup per head;
 * key 22: 1 = a batch-1 decode step on one GPU (16-bit weights and cache, <= 4096 keys) runs attention + merge + o_proj as one launch
 * (fused_decode.hip), 0 (default) = as three launches (same bits);
 * key 23: 1 = such a step runs every decoder layer as ONE launch (decode_layer.hip: qkv, attention, merge, o_proj, gate|up, down
 * with in-launch hand-offs; measured slower: 98 vs 91.5 us per layer), 0 (default) = six launches per layer (same bits);
 * key 24: 1 = the batch-1 gate|up GEMV takes its shares from per-XCD work queues (gemv_rows_norm_dyn_kernel; eager steps only; same
 * bits; measured 80-331 us against 45: the returning atomic drains the wave's load ring), 0 (default) = equal static shares;
 * key 25: 1 (default) = decode attention launches with enough keys (>= 6 tiles of 32 keys per CU: b >= 4 at 3.7 k keys, b = 1 from ~12 k)
 * take the LDS-DMA ring form (attention.hip: attn_decode_dma_kernel; 16-bit cache), 0 = the register multi-tile / one-tile kernels;
 * key 26: low byte = resident waves per CU that form's split count is sized for (0 = by launch size: 2 or 4), next byte = ring stages 2..4;
 * key 27 / 28: experiments (rotated tile order of that form; per-(blockIdx % 8) share deltas of the batch-1 gate|up GEMV, eight nibbles
 * d + 8: measured 1 % at best, profiles/r04_n);
 * key 30: GQA prefill attention, heaviest causal query-block ranks issued as two workgroups of half the kv group's heads each: -1 (default) =
 * a quarter of the ranks while the launch is at most one workgroup per CU (S <= 2048 for one Qwen2-7B sequence: 61.2 -> 53.7 us), 0 = off,
 * n > 0 = n ranks whatever the size (same bits either way);
 * key 33: 1 (default) = MHA prefill attention launches (the ViT) use a one-dimensional grid that keeps the query blocks of a (head, sequence) pair on
 * one XCD (K / V fetched once: 355 -> 82 MB per launch at 3 tiles), 0 = (query block, head, sequence) grid (same bits);
 * key 29: tensor parallelism, 1 = the row-parallel projections of the ViT / the prefill are all-reduced as fp32 partial sums and the
 * epilogue (bias, layer scale, residual) is applied once to the sum, so TP = N differs from TP = 1 in fp32 summation order only (twice
 * the bytes on the links); 0 (default) = every rank applies the epilogue to its own partial and 16-bit results are summed.  TP = 8 f16
 * logit error against TP = 1: 4.97e-3 with, 5.32e-3 without (profiles/r04_w, r04_x));
 * key 34: launch shapes for the SHARD widths of a tensor-parallel rank's decode GEMVs (default 15): bit 0 = the x-stationary form, one tile per
 * workgroup, for the short qkv shard of a batched step; bit 1 = split-K slices of >= 16 chunks and one chunk per wave per step for short-K
 * o_proj / down_proj shards; bit 2 = one (gate, up) pair per wave for the short batch-1 gate|up shard; bit 3 = the x-stationary form, one
 * (gate, up) tile per workgroup, for a batched gate|up shard of at most one tile per CU (0 = the round-4 shapes; same sums up to fp32 order
 * where the slice count changes);
 * key 35 (experiments build): a rank context whose exchanges are no-ops takes the one-GPU launch structures on its shard widths (measurement);
 * key 36 (experiments build): 2 = the MHA prefill attention splits the keys between two wave groups of an eight-wave workgroup (measured slower);
 * key 43: 1 (default) = the MFMA GEMMs issue no operand reads and no MFMAs for 64-row blocks / waves that lie wholly beyond M (the 13th row tile of the
 * 3-tile ViT holds 3 valid rows of 256), 0 = full issue (same bits; in the model -0.7 % of the ViT, inside the noise: profiles/r06_a_gemm_dead_row_skip_ab.txt);
 * key 44: 1 (default) = the RMSNorm vision tower at TP = 1 runs the fused layer of round 6 (norm1 / norm2 as a row scale of the qkv / fc1 GEMM with the
 * norm weight folded into the GEMM weight, statistics from the producing GEMM's epilogue finished per tile inside the consuming GEMM, q norm on load in the
 * attention kernel, K norm from the qkv epilogue's statistics: six launches), 0 = the eight launches of round 5;
 * key 46: 1 (default) = MHA prefill attention whose key count is a whole number of 64-key tiles + 1 (the ViT: 1025) folds the odd key into the
 * online softmax's initial state (m = q . k_last, l = 1, O = v_last) instead of giving it a tile step of its own (-5.9 % of the launch), 0 = 17 tile steps;
 * key 47: 64-key tiles per wave of the decode attention over the e4m3 KV cache: 0 (default) = by the launch's size (one wave per tile below two
 * one-tile waves per CU; from there a wave walks 3 tiles with the next one prefetched into registers, and four such waves share a workgroup and
 * fold their states in LDS when that leaves <= 64 partials per head for the one-round-trip merge and still fills half the CUs: 33 k keys 21.1 ->
 * 14.9 us per attention + merge), 1 = always one wave per tile, 2 / 3 / 4 = tiles per wave forced, 11 = 3 tiles x 4 waves forced;
 * key 48: 1 (default) = that walking form also rotates q / k and appends + quantises the new token's k / v rows (no RoPE + append launch in front of
 * the attention of a long-context e4m3 decode step; the same bytes), 0 = separate launch;
 * key 45: 1 (default) = sequence-parallel norms under tensor parallelism (omchat_ctx_sp_stats), 0 = one all-reduce per sub-block and replicated norms;
 * key 37: 1 (default) = the GEMM epilogues store 16 bytes per lane (two column blocks exchanged between lane pairs), 0 = 8 bytes (same bits);
 * key 38: (gate, up) pairs per wave of the batch-1 gate|up GEMV's non-loop norm form: 1 (default: thousands of small workgroups that the dispatcher
 * re-balances over the XCDs; decode 2.650 -> 2.597 ms per token against the loop form), 2, 3 (same bits); 16 x n sets the e4m3 replica's form
 * (default 1: 1.700 -> 1.658 ms per token against 4);
 * key 39 (experiments build): n = 1 / 2 / 4: the batch-1 long-K GEMV (down_proj) loads x from L2 in every wave instead of staging it in LDS, as
 * workgroups of n waves (same bits; measured slower: 2.70 against 2.605 ms per token);
 * key 40 (experiments build): 1 = RMSNorm and ViT q / k norm launches of >= 256 rows (H <= 4096) take one wave per row -- all loads up front, no LDS,
 * no barrier; same bits, measured no faster -- 0 (default) = one workgroup per row;
 * key 41 (experiments build, TIMING PROBE: results are INVALID under it): 1 = every launch goes out with hipExtAnyOrderLaunch (consumers may overtake
 * producers): prices the launch boundaries of a step;
 * key 28 (the static-skew form < 256 exists in the experiments build only: the product build IGNORES values below 256): < 256: workgroups moved from every odd XCD's share of the batch-1 gate|up launch to every even XCD's (contiguous pair ranges per XCD; same bits;
 * -0.6 % of the step at 24); >= 256: eight nibbles of per-XCD share deltas of the loop form (round 4);
 * key 42 (experiments build, prototype): bit 0 = the batch-1 o_proj GEMV is launched out of order behind the split-KV merge and waits on its per-head completion
 * flags (gemv_rows_wait_kernel; bit-identical results); bit 1 = cache-bypassing loads of the row instead of an agent-scope acquire, bit 2 = write-through store
 * in the merge instead of a release, bit 3 = slower poll, bit 4 = collect in-kernel clock stamps of the six launches of a layer (printed by omchat_fused_status); measured not faster (7: 2.618 against 2.608 ms per token) */
/* batch-1 skinny GEMM with the RMSNorm that precedes it computed in the registers of every wave: y = epi(W RMSNorm(x; norm_w, eps)),
 * x the raw hidden row [K], K <= 4096, epi NONE / SWIGLU (transformers modeling_qwen2.py:247-252 + :46-48; the decode step uses it for
 * post_attention_layernorm + gate|up) */
int omchat_op_gemv_norm(int dtype, const void* X, const void* W, int ldw, void* Y, int N, int K, const void* norm_w, float eps,
                        const void* bias, int epi, int out_f32, void* stream);
int omchat_op_set_tuning(int key, int value);
/* GEMM tile choices are measured on first use of a (dtype, epilogue, ceil(M/256), N, K) class; load / dump persist them as text
 * (returns the number of entries, -1 when the file cannot be opened); omchat_gemm_tune_runs = measurements done by this process */
int omchat_gemm_tune_load(const char* path);
int omchat_gemm_tune_dump(const char* path);
long omchat_gemm_tune_runs(void);
size_t omchat_op_gemm_sk_ws(void);
int omchat_op_gemm_sk(int dtype, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                      const void* bias, const void* ls, const void* resid, int ldr, int epi, int force_tile, void* ws,
                      size_t ws_bytes, int stream_k, void* stream);
int omchat_op_gemv(int dtype, const void* X, int ldx, const void* W, int ldw, void* Y, int ldy, int b, int N, int K,
                   const void* bias, const void* resid, int ldr, int epi, int out_f32, void* stream);
/* batched decode GEMV with packed operands (MFMA fragment order, csrc/common.h): packs X [b <= 32, K] (and W [N, K] when
 * w_packed) into scratch, then Y = epi(X W^T); epi EPI_NONE (T or fp32 out) | EPI_SWIGLU (y_packed: output in the packed x layout
 * of the consumer) | EPI_PARTIAL (fp32 [ksplit][b][ldy]).  omchat_op_pack_x: row-major -> packed x with NB = b > 16 ? 2 : 1. */
int omchat_op_gemv_packed(int dtype, const void* X, int ldx, const void* W, int ldw, void* Y, int ldy, int b, int N, int K,
                          const void* bias, int epi, int out_f32, int ksplit, int w_packed, int y_packed, void* stream);
int omchat_op_pack_x(int dtype, const void* X, int ldx, int b, int K, void* out, void* stream);
/* fp8 x fp8 MFMA GEMM (BASELINE configs[4] "fp8 MFMA weights"): A8 [M, K] and W8 [N, K] are OCP e4m3 bytes with one fp32 scale per row;
 * C (dtype) = epi(a_scale[m] * w_scale[n] * A8 W8^T), epilogues as omchat_op_gemm (ls unused).  K % 128 == 0.
 * omchat_op_quant_rows_fp8: per-row (per token) quantisation of activations, optionally behind an RMSNorm (norm_w != NULL). */
int omchat_op_gemm_fp8(int dtype, const void* A8, const float* a_scale, const void* W8, const float* w_scale, void* C, int ldc, int M, int N,
                       int K, const void* bias, const void* resid, int ldr, int epi, void* stream);
int omchat_op_quant_rows_fp8(int dtype, const void* x, const void* norm_w, float eps, void* y8, float* scale, int rows, int H, void* stream);
/* fp8 pieces: W [N][K] (dtype) -> W8 [N][K] e4m3 bytes + scale [N]; y[N] = (W8 . x) * scale (+ epilogue), batch 1;
 * epi EPI_PARTIAL writes fp32 slices [ksplit][N] */
int omchat_op_quant_fp8(int dtype, const void* W, int N, int K, void* W8, float* scale, void* stream);
int omchat_op_gemv_fp8(int dtype, const void* X, const void* W8, const float* scale, void* Y, int N, int K, const void* bias,
                       const void* resid, int epi, int out_f32, int ksplit, void* stream);
int omchat_op_rmsnorm(int dtype, const void* x, const void* w, void* y, int rows, int H, float eps, void* stream);
/* the decode step's fused residual add + RMSNorm (modeling_qwen2.py:283-296 + 247-252): x[rows, H] = T(x + T(sum_s part[s])) in place,
 * part = fp32 split-K slices [ks][rows][H] of the projection, then xn = T(w * T(x * rsqrt(mean(x^2) + eps))) (w == NULL: skip);
 * pack_nb != 0 writes xn in the packed x layout of the batched GEMV */
int omchat_op_resid_rmsnorm(int dtype, void* x, int ldx, const float* part, int ks, const void* w, void* xn, int ldn, int rows, int H,
                            float eps, int pack_nb, void* stream);
int omchat_op_vit_qknorm(int dtype, void* qkv, int ld, const void* wq, const void* wk, int rows, int C, int C_total,
                         float eps, float q_scale, void* stream);
/* q [b,Sq,Hq,128], k/v [b,Hkv,Skv,128] (cache layout), out [b,Sq,Hq,128]; kv_len device int32 [b] or NULL */
int omchat_op_attn_prefill(int dtype, const void* q, const void* k, const void* v, void* out, int b, int Sq, int Skv,
                           int Hq, int Hkv, const int32_t* kv_len, int causal, int q_pos0, float scale, void* stream);
/* same with head dim D = 128 or 64 (InternViT-300M attention, intern_vit_300m/modeling_intern_vit.py:138-155) */
int omchat_op_attn_prefill_d(int dtype, const void* q, const void* k, const void* v, void* out, int b, int Sq, int Skv,
                             int Hq, int Hkv, int D, const int32_t* kv_len, int causal, int q_pos0, float scale, void* stream);
/* nn.LayerNorm over the last dim (weight + bias), fp32 statistics, one rounding */
int omchat_op_layernorm(int dtype, const void* x, const void* w, const void* b, void* y, int rows, int H, float eps, void* stream);
/* q [b,Hq,128], k/v [b,Hkv,cap,128]; kv_len device int32 [b] or NULL (-> L); ws from omchat_op_attn_decode_ws */
size_t omchat_op_attn_decode_ws(int b, int Hq, int L);
int omchat_op_attn_decode(int dtype, const void* q, const void* k, const void* v, void* out, int b, int Hq, int Hkv,
                          int cap, int L, const int32_t* kv_len, float scale, void* ws, size_t ws_bytes, void* stream);
/* the same over an e4m3 KV cache (omchat_enable_fp8_kv; BASELINE configs[4]): k8 / v8 [b,Hkv,cap,128] OCP e4m3 bytes, ks / vs [b,Hkv,cap] fp32, one
 * scale per cached row (as omchat_op_rope_kv_q8 writes them): score = (q . k8) * ks in fp32, the value scale folded into the probabilities */
int omchat_op_attn_decode_kv8(int dtype, const void* q, const void* k8, const void* v8, const float* ks, const float* vs, void* out, int b,
                              int Hq, int Hkv, int cap, int L, const int32_t* kv_len, float scale, void* ws, size_t ws_bytes, void* stream);
/* one decode step's attention over the e4m3 cache INCLUDING the new token's RoPE + append + quantisation, as the decoder layer issues it: qkv
 * [b][(Hq + 2 Hkv) * 128] raw projections; sequence i holds kv_len[i] keys counting the new one (NULL: L), whose rows go to position kv_len[i] - 1
 * of k8 / v8 / ks / vs and of the 16-bit caches k16 / v16 [b,Hkv,cap,128]; pos = kv_len - 1 (device, NULL with kv_len).  Launches long enough for
 * the walking form (tuning key 47) do all of it in the attention launch (key 48; omchat_op_rope_kv_q8's bytes), the others run that launch in front */
int omchat_op_attn_decode_kv8_append(int dtype, void* qkv, float theta, void* k8, void* v8, float* ks, float* vs, void* k16, void* v16, void* out,
                                     int b, int Hq, int Hkv, int cap, int L, const int32_t* kv_len, const int32_t* pos, float scale, void* ws,
                                     size_t ws_bytes, void* stream);
/* qkv [rows, (Hq+2Hkv)*128]; rows = b*S; positions pos0 + s; caches [b,Hkv,cap,128]; theta: rope base */
int omchat_op_rope_kv(int dtype, void* qkv, int b, int S, int Hq, int Hkv, int pos0, float theta, void* kcache, void* vcache,
                      int cap, void* stream);
/* the same with the appended rows ALSO quantised into an e4m3 KV cache (k8 / v8 [b, Hkv, cap, 128] bytes, ks / vs [b, Hkv, cap] fp32: per row
 * s = absmax / 448 (1 for a zero row), bytes = e4m3_rne(x / s) of the 16-bit values stored in kcache / vcache) -- the decode step of the fp8
 * KV cache mode (BASELINE configs[4]) */
int omchat_op_rope_kv_q8(int dtype, void* qkv, int b, int S, int Hq, int Hkv, int pos0, float theta, void* kcache, void* vcache, int cap,
                         void* k8, void* v8, float* ks, float* vs, void* stream);
int omchat_op_argmax(const float* logits, int b, int V, int32_t* out, void* stream);
int omchat_op_fill_uniform(int dtype, void* dst, int64_t n, uint64_t key, float scale, float offset, void* stream);

/* ---- image front-end: process_anyres_image (omchat/mm_utils.py:119-158) + CLIPImageProcessor ---- */
/* (internVIT_encoder.py:25-29) on the device, bit-identical to the reference's PIL/numpy result.
 * plan (host only): select_best_resolution (mm_utils.py:12-39) over `pinpoints` [n_pin][2] = (w, h); n_tiles = 1 thumbnail
 * + the tiles of the best resolution.  anyres: `rgb` uint8 [H][W][3] (host, or device when rgb_on_device) ->
 * pixels_out device [n_tiles][3][tile][tile] in `dtype` (OMCHAT_F16 / BF16 / F32): tile 0 = Image.resize((tile, tile)),
 * then the row-major tiles of resize_and_pad_image (mm_utils.py:42-74); every value = (u/255 - mean)/std with
 * transformers' rounding order.  Blocks until the staging copies are done (the kernels stay ordered on `stream`). */
int omchat_preproc_plan(int W, int H, const int* pinpoints, int n_pin, int tile, int* best_w, int* best_h, int* n_tiles);
int omchat_preproc_anyres(int dtype, const void* rgb, int rgb_on_device, int W, int H, int best_w, int best_h, int tile,
                          const float mean[3], const float std_[3], void* pixels_out, void* stream);
/* dynamic_preprocess + process_dynamic_image (mm_utils.py:276-323, the OmChat-2.1 / InternViT-300M tiling): plain resize to
 * (grid_w*tile, grid_h*tile) -- aspect not preserved --, row-major tiles, thumbnail FIRST when `thumbnail` (the reference
 * adds it when the grid has more than one block); pixels_out [thumbnail + grid_w*grid_h][3][tile][tile].  The grid choice
 * (find_closest_aspect_ratio, :326-339) is host integer/float logic in omchat_amd/mm_utils.py. */
int omchat_preproc_dynamic(int dtype, const void* rgb, int rgb_on_device, int W, int H, int grid_w, int grid_h, int tile, int thumbnail,
                           const float mean[3], const float std_[3], void* pixels_out, void* stream);
/* host-only pieces exposed for the CPU parity tests: Pillow's fixed-point resampling taps (Resample.c precompute_coeffs +
 * normalize_coeffs_8bpc, BICUBIC): bounds [out][2] = (first source index, taps), kk [out][*ksize]; and the 3 x 256
 * rescale+normalize table */
int omchat_resample_coeffs(int in_size, int out_size, int* ksize, int* bounds, int* kk, int kk_cap);
int omchat_normalize_lut(const float mean[3], const float std_[3], float lut[768]);

/* ---- tensor-parallel bootstrap (RCCL over xGMI; one process per GPU) -------------------------------------------- */
int omchat_comm_unique_id(char id[128]);
int omchat_comm_init(const char id[128], int rank, int size, void** comm_out);
/* the collective of the tensor-parallel data path, callable on its own: in-place sum over the ranks of `comm` */
int omchat_comm_allreduce(void* comm, void* buf, size_t count, int dtype, void* stream);
void omchat_comm_destroy(void* comm);

int omchat_comm_count(void* comm, int* nranks);                  /* ncclCommCount: ranks of the communicator */

/* ---- peer all-reduce over IPC-mapped buffers (xGMI P2P), csrc/comm.hip ------------------------------------------ */
/* The decode-sized collective of SURVEY.md 8e: one-shot (every rank reads every rank's slot and sums in rank order: all ranks
 * get identical bits) up to `oneshot_max` bytes, two-shot (reduce-scatter + all-gather by peer reads) above it.  One process
 * per GPU: omchat_peer_create allocates the rank's shared buffer (capacity `cap_bytes` per message piece; longer messages are
 * cut into pieces) and returns its 64-byte hipIpcMemHandle_t; the caller ships the handles of all ranks (any bootstrap channel;
 * tp.py uses torch.distributed) and passes the `size` x 64-byte table to omchat_peer_connect.  omchat_peer_connect_local is the
 * same-process variant (rank contexts in threads: pointers instead of IPC handles). */
typedef struct omchat_peer omchat_peer;
int omchat_peer_create(int rank, int size, size_t cap_bytes, omchat_peer** out, char handle_out[64]);
int omchat_peer_connect(omchat_peer* p, const char* all_handles);
void* omchat_peer_base(omchat_peer* p);
int omchat_peer_connect_local(omchat_peer* p, void* const* bases);
/* fast != 0: drop the one-lane system-scope release / acquire fences around the flag barrier (payload stays sc0 sc1);
 * oneshot_max_bytes != 0: one-shot / two-shot switch point (default 256 KiB); max_blocks != 0: workgroups per call (default 64,
 * <= 128; every workgroup of a call must be resident on every rank at once, so rank contexts that SHARE a GPU keep it small) */
int omchat_peer_set_mode(omchat_peer* p, int fast, size_t oneshot_max_bytes, int max_blocks);
size_t omchat_peer_capacity(omchat_peer* p);
/* in-place sum over the ranks, `count` elements of OMCHAT_F16 / BF16 / F32; buf 16-byte aligned, byte count % 16 == 0 */
int omchat_peer_allreduce(omchat_peer* p, void* buf, size_t count, int dtype, void* stream);
/* the two halves of that all-reduce as collectives of their own (round 6, sequence-parallel norms): buf = [size][blk_count] elements (dtype as
 * above; block bytes a multiple of 16).  reduce_scatter: block `rank` of buf becomes the sum over the ranks of their block `rank` (rank order,
 * fp32, one rounding: the bits of omchat_peer_allreduce), the other blocks are left unchanged.  all_gather: block `rank` of every rank is copied into
 * block `rank` of every other rank's buf.  Same stream / ordering rules as omchat_peer_allreduce. */
int omchat_peer_reduce_scatter(omchat_peer* p, void* buf, size_t blk_count, int dtype, void* stream);
int omchat_peer_all_gather(omchat_peer* p, void* buf, size_t blk_count, int dtype, void* stream);
/* Tensor-parallel decode, the all-reduce fused with its consumer: part = this rank's fp32 split-K slices [ks][rows][H] of a row-parallel
 * projection (ks <= 8, rows <= 128, ks * rows * H * 4 <= capacity).  On return (stream order) x[rows, H] = T(x + T(sum over ranks and
 * slices)) on every rank and, when w != NULL, xn = T(w * T(x * rsqrt(mean(x^2) + eps))) (pack_nb != 0: in the packed x layout of the
 * batched GEMV).  Same bits as omchat_peer_allreduce(part) followed by the local residual + RMSNorm launch, one launch instead of two.
 * (reference: the residual adds and RMSNorms of transformers modeling_qwen2.py:283-296, 247-252) */
int omchat_peer_resid_rmsnorm(omchat_peer* p, int dtype, void* x, int ldx, const float* part, int ks, const void* w, void* xn, int ldn,
                              int rows, int H, float eps, int pack_nb, void* stream);
/* synchronises the device; *err_out = 1 when a barrier spin timed out (a peer died or never arrived) since the last call */
int omchat_peer_error(omchat_peer* p, int* err_out);
void omchat_peer_destroy(omchat_peer* p);
/* use `peer` for the tensor-parallel sums of `ctx`: messages <= max_bytes (0 = keep 256 KiB), every size when all_sizes != 0
 * or when the context has no RCCL communicator */
int omchat_ctx_set_peer(omchat_ctx* ctx, omchat_peer* peer, size_t max_bytes, int all_sizes);
int omchat_ctx_comm_stats(omchat_ctx* ctx, long* peer_calls, long* rccl_calls);
/* Sequence-parallel norms under tensor parallelism (round 6; default on, tuning key 45 = 0 restores the all-reduce form): a row-parallel
 * projection of the vision tower / the prefill ends in a reduce-scatter over row blocks (rank r receives the summed rows it owns: block r of
 * ceil(rows / tp_size) rows per chunk), the owner adds the residual and normalises ITS rows, an all-gather hands every rank the normalised
 * activation; the residual stream stays row-sharded between sub-blocks and is gathered once at the end.  Same link bytes as the all-reduce, the
 * replicated RMSNorm / LayerNorm work divided by tp_size.  RCCL: ncclReduceScatter / ncclAllGather in place; hook / peer transports emulate both
 * with all-reduces.  The counters say which form ran. */
int omchat_ctx_sp_stats(omchat_ctx* ctx, long* reduce_scatters, long* all_gathers);
/* in-place sum of a caller buffer over the context's tensor-parallel group (same transports as the model's own all-reduces); used for
 * the data-parallel vision tower: every rank encodes its share of the tiles into a zero-filled feature buffer and the sum gathers them */
int omchat_ctx_allreduce(omchat_ctx* ctx, void* buf, size_t count, int dtype, void* stream);

/* Test seam: replace the RCCL all-reduce of a tensor-parallel context by a caller-supplied function (sum over ranks,
 * in place, `count` elements of dtype OMCHAT_F16/BF16/F32, ordered on `stream`).  Lets the whole TP dataflow be
 * verified with several rank contexts on ONE GPU; production contexts never set it. */
typedef int (*omchat_allreduce_fn)(void* user, void* buf, size_t count, int dtype, void* stream);
int omchat_set_allreduce_hook(omchat_ctx* ctx, omchat_allreduce_fn fn, void* user);
/* Measurement seam (bench.py --shard-of N): an omchat_allreduce_fn that does nothing.  A context created with tp_size = N and this
 * hook runs ONE rank's share of the tensor-parallel work (its GEMM / GEMV / attention shard shapes and launch sequence) on a single
 * GPU with the exchanges removed: kernel time per rank without an N-GPU node.  The results are NOT the model's outputs. */
int omchat_allreduce_noop(void* user, void* buf, size_t count, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
