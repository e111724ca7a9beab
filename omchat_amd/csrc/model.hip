// Model-level loops of the OmChat hot path on one GPU (one tensor-parallel rank): context, weight routing, workspaces,
// KV cache, and the kernel sequences for the vision tower, projector, prefill and decode.  Host C++; every FLOP runs
// in the HIP kernels of this directory.
#include "kernels.h"
#include "../../include/omchat_hip.h"
#include <rccl/rccl.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

static thread_local std::string g_err;
void omchat_set_error(const std::string& s) { g_err = s; }
extern "C" const char* omchat_last_error(void) { return g_err.c_str(); }
extern "C" const char* omchat_version(void) { return "omchat_hip 0.1 (gfx950)"; }

namespace {

__global__ void cast_from_f32_f16(const float* s, f16* d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (f16)s[i];
}
__global__ void cast_from_f32_bf16(const float* s, bf16* d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (bf16)s[i];
}
__global__ void add_positions_kernel(int* pos, int b, int d) {
  const int i = threadIdx.x;
  if (i < b) pos[i] += d;
}
__global__ void advance_lens_kernel(int* pos, int* len, int b) {
  const int i = threadIdx.x;
  if (i < b) { pos[i] += 1; len[i] += 1; }
}
__global__ void last_row_index_kernel(const int* len, int S, int b, int* idx) {
  const int i = threadIdx.x;
  if (i < b) idx[i] = i * S + len[i] - 1;
}

// vocab-parallel greedy: every rank contributes (max logit, global index) per sequence; summed into a zeroed table
__global__ void tp_argmax_scatter_kernel(const float* logits, int ld, const int* local_idx, int b, int rank, int v_local, float* table) {
  const int i = threadIdx.x;
  if (i < b) {
    table[((size_t)rank * b + i) * 2] = logits[(size_t)i * ld + local_idx[i]];
    table[((size_t)rank * b + i) * 2 + 1] = (float)(rank * v_local + local_idx[i]);
  }
}
__global__ void tp_argmax_pick_kernel(const float* table, int b, int size, int* out) {
  const int i = threadIdx.x;
  if (i < b) {
    float best = table[(size_t)i * 2]; int bi = (int)table[(size_t)i * 2 + 1];
    for (int r = 1; r < size; ++r) {            // ranks hold ascending index ranges: strict > keeps the first index on ties
      const float v = table[((size_t)r * b + i) * 2];
      if (v > best) { best = v; bi = (int)table[((size_t)r * b + i) * 2 + 1]; }
    }
    out[i] = bi;
  }
}

constexpr int DEC_KS_MAX = 8;

enum RouteKind { R_PLAIN = 0, R_GATE = 1, R_UP = 2, R_PATCH = 3 };
struct Route {
  void* dst = nullptr;
  int64_t rows = 0, cols = 0;     // logical source shape (2-D view)
  int64_t dst_ld = 0;             // destination row stride (elements)
  int kind = R_PLAIN;
  bool loaded = false;
  float synth_std = 0.02f, synth_off = 0.f;
};

}  // namespace

#if !OMCHAT_EXPERIMENTS
// product build: the round-4 one-launch experiments are not compiled in (kernels.h); their entry points refuse
size_t fused_decode_ws_bytes(int) { return 0; }
bool attn_oproj_fused_ok(const AttnDecodeArgs&, int, int) { return false; }
int launch_attn_oproj_fused(int, const AttnDecodeArgs&, const FusedDecodeArgs&, hipStream_t) { omchat_set_error("fused decode launch: build with -DOMCHAT_EXPERIMENTS=1"); return 1; }
size_t decode_layer_ws_bytes(int, int, int, int, int) { return 0; }
bool decode_layer_ok(const DecodeLayerArgs&) { return false; }
int launch_decode_layer(int, const DecodeLayerArgs&, hipStream_t) { omchat_set_error("one-launch decode layer: build with -DOMCHAT_EXPERIMENTS=1"); return 1; }
#endif
extern "C" int omchat_has_experiments(void) { return OMCHAT_EXPERIMENTS; }
int g_shard_as_tp1 = 0;          // omchat_op_set_tuning key 35 (experiments build): see decode_body
void model_set_shard_as_tp1(int v) { g_shard_as_tp1 = v; }
int g_decode_layer = 0;          // omchat_op_set_tuning key 23: 1 = batch-1 decode on one GPU runs each decoder layer as ONE launch (decode_layer.hip); 0 = six launches (same bits)
void model_set_decode_layer(int v) { g_decode_layer = v; }
int g_fuse_attn_oproj = 0;      // omchat_op_set_tuning key 22: 1 = batch-1 decode on one GPU runs split-KV attention + merge + o_proj as ONE launch (fused_decode.hip); 0 = three launches (A/B, same bits)
void model_set_fuse_attn_oproj(int v) { g_fuse_attn_oproj = v; }
int g_ao_oproj = 0;            // omchat_op_set_tuning key 42 (experiments build, prototype): batch-1 o_proj launched out of order behind the split-KV merge (gemv_rows_wait_kernel)
void model_set_ao_oproj(int v) { g_ao_oproj = v; }
int g_fuse_peer_norm = 1;      // omchat_op_set_tuning key 9: 0 = tensor-parallel decode keeps the all-reduce and the residual + RMSNorm as two launches (A/B)
void model_set_fuse_peer_norm(int v) { g_fuse_peer_norm = v; }

struct omchat_ctx {
  omchat_config c;
  int dt = OMCHAT_BF16;
  int tp_rank = 0, tp_size = 1;
  ncclComm_t comm = nullptr;
  std::vector<void*> allocs;
  size_t bytes = 0;
  std::unordered_map<std::string, Route> routes;

  // derived geometry
  int v_np = 0, v_ntok = 0, v_Cq = 0, v_kpad = 0, v_hd = 128;
  int t_qdim = 0, t_kvdim = 0, t_qkvdim = 0;

  // weights (device, compute dtype)
  struct VitLayer { void *ls1, *ls2, *n1, *n2, *n1b, *n2b, *wqkv, *qn, *kn, *wproj, *bproj, *w1, *b1, *w2, *b2; };
  struct DecLayer { void *ln1, *ln2, *wqkv, *bqkv, *wo, *wgu, *wd; };
  // weight-only fp8 replica of the decode-streamed decoder weights (omchat_enable_fp8_decode): OCP e4m3 bytes + one fp32
  // scale per output row; batch-1 decode steps stream these instead of the 16-bit weights, prefill keeps the 16-bit ones
  struct DecLayer8 { void *wqkv, *wo, *wgu, *wd; float *sqkv, *so, *sgu, *sd; };
  std::vector<DecLayer8> dl8;
  // packed replica of the same weights for BATCHED decode steps (2 <= b <= 32): MFMA fragment order, every wave load 1 KiB contiguous
  // (gemv.hip: gemv_pk_kernel; +14 GB at OmChat-13B, built on the first batched step; omchat_op_set_tuning key 6 = 0 disables it)
  struct DecLayerP { void *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wd = nullptr; };
  std::vector<DecLayerP> dlp;
  void* t_lmP = nullptr;
  bool pk_ready = false, pk_unavailable = false;
  void* t_lm8 = nullptr; float* t_lm8_s = nullptr;
  bool fp8_decode = false, fp8_stale = false;
  // BASELINE configs[4]: fp8 KV cache for decode (e4m3 bytes in the layout of the 16-bit cache + one fp32 scale per (layer, sequence,
  // kv head, position)) and fp8 x fp8 MFMA prefill GEMMs (qkv and gate|up: the activations come quantised per token from the RMSNorm)
  void *k8cache = nullptr, *v8cache = nullptr;
  float *ks8 = nullptr, *vs8 = nullptr;
  bool fp8_kv = false, kv8_valid = false, fp8_prefill = false;
  void* tw_q8 = nullptr; float* tw_q8s = nullptr;
  int64_t scale_layer_stride() const { return (int64_t)c.max_batch * c.t_kv_heads * c.max_seq; }
  // decode step as a hipGraph (omchat_enable_decode_graph): ~230 launches per token replayed as one graph launch.  Captured on a
  // context-owned stream (the caller's may be the legacy null stream, which cannot capture) with context-owned token / logits
  // buffers so that every kernel argument is replay-invariant; the split-KV attention grid is captured for `cap_len` keys
  // (empty splits exit at once) and the graph is re-captured when a sequence outgrows it.
  struct DecodeGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int cap_len = 0; };
  std::unordered_map<int, DecodeGraph> graphs;      // key = b * 2 + fp8
  bool graph_on = false;
  hipStream_t graph_stream = nullptr;
  hipEvent_t graph_ev_in = nullptr, graph_ev_out = nullptr;
  int32_t *d_tok_in = nullptr, *d_tok_out = nullptr;
  long graph_steps = 0, graph_replays = 0, graph_captures = 0;
  void *v_cls = nullptr, *v_pos = nullptr, *v_wpatch = nullptr, *v_bpatch = nullptr;
  std::vector<VitLayer> vl;
  void *p_w0 = nullptr, *p_b0 = nullptr, *p_w2 = nullptr, *p_b2 = nullptr;
  void *t_embed = nullptr, *t_norm = nullptr, *t_lm = nullptr;
  std::vector<DecLayer> dl;
  float* rope = nullptr;          // [max_seq][64][2]

  // workspaces
  void *vw_cols = nullptr, *vw_pe = nullptr, *vw_x = nullptr, *vw_x2 = nullptr, *vw_xn = nullptr, *vw_qkv = nullptr, *vw_ao = nullptr,
       *vw_h = nullptr, *vw_feat = nullptr, *vw_proj = nullptr;
  float* vw_sumsq = nullptr;
  // round 6, fused ViT layer (vit_run): statistics slots left by the GEMM epilogues, the row scale finished from them, and the copies of
  // the qkv / fc1 weights with norm1 / norm2's weight folded into their columns (W'[n][k] = T(W[n][k] * w_norm[k])); rebuilt after a reload
  float *vw_stats_x = nullptr, *vw_stats_qk = nullptr, *vw_rstd = nullptr;
  int vw_stats_x_ld = 0, vw_stats_qk_ld = 0, vw_stats_x_ld_used = 1;      // (_used: slots of x's statistics written by the last producer)
  struct VitFold { void *wqkv = nullptr, *w1 = nullptr; };
  std::vector<VitFold> vfold;
  bool vfold_stale = true;
  void *tw_x = nullptr, *tw_x2 = nullptr, *tw_xn = nullptr, *tw_qkv = nullptr, *tw_ao = nullptr, *tw_act = nullptr, *tw_last = nullptr;
  float* tw_logits = nullptr;
  void* sk_ws = nullptr; size_t sk_ws_bytes = 0;      // stream-K slabs + flags of the MFMA GEMM
  float* tp_table = nullptr;
  float* tp_f32_ws = nullptr; size_t tp_f32_bytes = 0;      // fp32 partial sums of a row-parallel projection (tuning key 29), grown on demand
  void* arg_scratch = nullptr;
  float* tw_part = nullptr;       // split-K fp32 slices of the decode o_proj / down_proj [KS_MAX][max_batch][H]
  float* tw_attn_ws = nullptr;
  size_t tw_attn_ws_bytes = 0;
  // fused attention + merge + o_proj launch of the batch-1 decode step (fused_decode.hip): granule buffers, sticky time-out word, and the
  // launch counter that tags the granules of one launch
  void* fd_ws = nullptr;
  unsigned* fd_err = nullptr;
  unsigned fd_epoch = 0;
  long n_fused_launches = 0;
  // one-launch decoder layer (decode_layer.hip): granule buffers; shares the error word and the launch counter above
  void* dl_ws = nullptr;
  long n_layer_launches = 0;
  unsigned long long* dbg_stamps = nullptr;      // experiments build: [layers][8] clock stamps of the last decode step (tuning key 42 bit 4)
  unsigned* dyn_ctr = nullptr;      // [layers][65 * 64] work counters of the dynamic gate|up GEMV (gemv_rows_norm_dyn_kernel), zero between launches
  int *d_pos = nullptr, *d_len = nullptr, *d_idx = nullptr, *d_start = nullptr;
  bool left_padded = false;
  // decode of a padded batch as the reference computes it (omchat_decode_step_masked): every row's cache holds pre_S + masked_steps slots;
  // dec_mode: 0 = no decode step since the prefill, 1 = omchat_decode_step (per-sequence lengths), 2 = omchat_decode_step_masked
  int pre_S = 0, pre_b = 0, masked_steps = 0, dec_mode = 0;
  unsigned char* d_mask = nullptr; int64_t mask_sb = 0;
  bool mask_on_device = false;      // d_mask holds [prompt mask | ones] for the whole cache (omchat_masked_decode_begin)
  void *kcache = nullptr, *vcache = nullptr;   // [layers][max_batch][kv_heads][max_seq][128]
  std::vector<int> h_len;
  // optional per-kernel-class HIP-event timing (bench.py roofline): category -> event pairs recorded on the launch stream
  struct Prof { std::vector<hipEvent_t> ev; size_t used = 0; double ms = 0; long count = 0; };
  bool prof_on = false;
  Prof prof[OMCHAT_PROF_CATS];
  void prof_mark(int cat, hipStream_t s) {
    if (!prof_on) return;
    Prof& p = prof[cat];
    if (p.used == p.ev.size()) { hipEvent_t e; hipEventCreate(&e); p.ev.push_back(e); }
    hipEventRecord(p.ev[p.used++], s);
  }
  void* stage_f32 = nullptr; size_t stage_f32_bytes = 0;
  void* stage_t = nullptr; size_t stage_t_bytes = 0;

  int alloc(void** p, size_t n) {
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(p, n);
    if (e != hipSuccess) { omchat_set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e)); return 2; }
    allocs.push_back(*p);
    bytes += n;
    return 0;
  }
  size_t esz() const { return 2; }
  int64_t cache_layer_stride() const { return (int64_t)c.max_batch * c.t_kv_heads * c.max_seq * 128; }
  int64_t cache_sb() const { return (int64_t)c.t_kv_heads * c.max_seq * 128; }
  int64_t cache_sh() const { return (int64_t)c.max_seq * 128; }

  omchat_allreduce_fn hook = nullptr;
  void* hook_user = nullptr;
  // tensor-parallel prefill / ViT: the all-reduce of a row-parallel projection runs on its own stream, one row chunk behind the GEMM
  static constexpr int AR_CHUNKS = 4;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_chunk[AR_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_comm_done = nullptr;
  // sequence-parallel form: the all-gather of row chunk i is done (communication stream); the NEXT column-parallel GEMM is issued per row chunk
  // behind these events, so the exchange of chunk i + 1 also runs under the consumer's chunk i (gemm_sp / gemm_after_sp)
  hipEvent_t ev_ag[AR_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};
  int sp_pend_nch = 0, sp_pend_rows_per = 0;
  // transports of the tensor-parallel sum, in order of precedence: the test hook; the peer (IPC / xGMI) all-reduce of comm.hip for
  // messages up to peer_max bytes (one-shot: decode-sized) or for every size when there is no RCCL communicator / peer_all is set;
  // RCCL otherwise.  Every buffer passed here is context-owned with >= 16 bytes of slack, so the peer path may round the count up
  // to whole 16-byte pieces (the extra elements are summed and never read).
  omchat_peer* peer = nullptr;
  size_t peer_max = 256 * 1024;
  bool peer_all = false;
  long n_ar_peer = 0, n_ar_rccl = 0;
  int allreduce_any(void* buf, size_t count, int dtype, hipStream_t s) {
    if (tp_size == 1) return 0;
    if (hook) return hook(hook_user, buf, count, dtype, s);
    const size_t esz = dtype == OMCHAT_F32 ? 4 : 2;
    if (peer && (count * esz <= peer_max || !comm || peer_all)) {
      const size_t per16 = 16 / esz;
      ++n_ar_peer;
      return omchat_peer_allreduce(peer, buf, (count + per16 - 1) / per16 * per16, dtype, s);
    }
    if (!comm) { omchat_set_error("tensor-parallel context without a transport: pass an RCCL communicator or call omchat_ctx_set_peer"); return 1; }
    ++n_ar_rccl;
    const ncclDataType_t t = dtype == OMCHAT_F32 ? ncclFloat32 : (dtype == OMCHAT_F16 ? ncclFloat16 : ncclBfloat16);
    ncclResult_t r = ncclAllReduce(buf, buf, count, t, ncclSum, comm, s);
    if (r != ncclSuccess) { omchat_set_error(std::string("ncclAllReduce: ") + ncclGetErrorString(r)); return 3; }
    return 0;
  }
  // decode: sum of the split-K slices over the ranks + residual + RMSNorm.  With the peer transport that is ONE launch
  // (omchat_peer_resid_rmsnorm, same bits); with the hook or RCCL: all-reduce of the slices, then the local kernel.
  long n_fused_norm = 0;
  int reduce_resid_rmsnorm(void* x, int ldx, float* part, int ks, const void* w, void* xn, int ldn, int rows, int H, float eps, int pack_nb,
                           hipStream_t s) {
    const size_t bytes = (size_t)ks * rows * H * 4;
    if (tp_size > 1 && !hook && peer && g_fuse_peer_norm && rows <= 128 && bytes <= omchat_peer_capacity(peer) &&
        (bytes <= peer_max || !comm || peer_all)) {
      ++n_fused_norm;
      return omchat_peer_resid_rmsnorm(peer, dt, x, ldx, part, ks, w, xn, ldn, rows, H, eps, pack_nb, s);
    }
    if (tp_size > 1) { const int rc = allreduce_any(part, (size_t)ks * rows * H, OMCHAT_F32, s); if (rc) return rc; }
    return launch_resid_rmsnorm(dt, x, ldx, part, ks, w, xn, ldn, rows, H, eps, s, pack_nb);
  }
  // Sequence-parallel form (round 6): buf = [tp_size][blk_rows][N] 16-bit rows.  reduce_scatter_rows: afterwards block tp_rank holds the sum over
  // the ranks (the other blocks are undefined); all_gather_rows: every rank contributes block tp_rank, afterwards all blocks are whole everywhere.
  // RCCL: ncclReduceScatter / ncclAllGather in place -- together the bytes of ONE all-reduce; peer transport: the two halves of its two-shot
  // all-reduce as kernels of their own (comm.hip omchat_peer_reduce_scatter / omchat_peer_all_gather).  The test hook has all-reduce only:
  // reduce-scatter = all-reduce (every block comes back summed), all-gather = zero the foreign blocks, then all-reduce (x + 0 is exact).
  long n_rs = 0, n_ag = 0;
  bool sp_native(size_t bytes) const { return !hook && comm && !(peer && (bytes <= peer_max || peer_all)); }
  bool sp_peer(size_t bytes) const { return !hook && peer && (bytes <= peer_max || !comm || peer_all); }
  int reduce_scatter_rows(void* buf, int blk_rows, int N, hipStream_t s) {
    if (tp_size == 1) return 0;
    ++n_rs;
    const size_t blk = (size_t)blk_rows * N;
    if (sp_peer(blk * tp_size * 2)) { ++n_ar_peer; return omchat_peer_reduce_scatter(peer, buf, blk, dt, s); }
    if (!sp_native(blk * tp_size * 2)) return allreduce_any(buf, blk * tp_size, dt, s);
    ++n_ar_rccl;
    ncclResult_t r = ncclReduceScatter(buf, (char*)buf + (size_t)tp_rank * blk * 2, blk, dt == OMCHAT_F16 ? ncclFloat16 : ncclBfloat16, ncclSum, comm, s);
    if (r != ncclSuccess) { omchat_set_error(std::string("ncclReduceScatter: ") + ncclGetErrorString(r)); return 3; }
    return 0;
  }
  int all_gather_rows(void* buf, int blk_rows, int N, hipStream_t s) {
    if (tp_size == 1) return 0;
    ++n_ag;
    const size_t blk = (size_t)blk_rows * N;
    if (sp_peer(blk * tp_size * 2)) { ++n_ar_peer; return omchat_peer_all_gather(peer, buf, blk, dt, s); }
    if (!sp_native(blk * tp_size * 2)) {
      if (hook == omchat_allreduce_noop) return 0;      // bench.py --shard-of: one rank's compute with the exchanges removed
      if (tp_rank > 0 && hipMemsetAsync(buf, 0, (size_t)tp_rank * blk * 2, s) != hipSuccess) { omchat_set_error("all_gather_rows: memset"); return 2; }
      if (tp_rank + 1 < tp_size && hipMemsetAsync((char*)buf + (size_t)(tp_rank + 1) * blk * 2, 0, (size_t)(tp_size - 1 - tp_rank) * blk * 2, s) != hipSuccess) {
        omchat_set_error("all_gather_rows: memset"); return 2;
      }
      return allreduce_any(buf, blk * tp_size, dt, s);
    }
    ++n_ar_rccl;
    ncclResult_t r = ncclAllGather((char*)buf + (size_t)tp_rank * blk * 2, buf, blk, dt == OMCHAT_F16 ? ncclFloat16 : ncclBfloat16, comm, s);
    if (r != ncclSuccess) { omchat_set_error(std::string("ncclAllGather: ") + ncclGetErrorString(r)); return 3; }
    return 0;
  }
  int allreduce(void* buf, size_t count, hipStream_t s) { return allreduce_any(buf, count, dt, s); }
  int allreduce_f32(float* buf, size_t count, hipStream_t s) { return allreduce_any(buf, count, OMCHAT_F32, s); }
};

namespace {

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

int add_route(omchat_ctx* ctx, const std::string& name, void** slot, int64_t rows, int64_t cols, float std_, float off,
              int kind = R_PLAIN, void* dst_override = nullptr, int64_t dst_ld = -1) {
  Route r;
  r.rows = rows; r.cols = cols; r.kind = kind; r.synth_std = std_; r.synth_off = off;
  if (dst_override) {
    r.dst = dst_override;
  } else {
    TRY(ctx->alloc(&r.dst, (size_t)rows * cols * 2));
    if (slot) *slot = r.dst;
  }
  r.dst_ld = dst_ld < 0 ? cols : dst_ld;
  ctx->routes[name] = r;
  return 0;
}

int build(omchat_ctx* ctx) {
  const omchat_config& c = ctx->c;
  const int C = c.v_hidden, I = c.v_mlp;
  ctx->v_np = (c.v_image / c.v_patch) * (c.v_image / c.v_patch);
  ctx->v_ntok = ctx->v_np + 1;
  ctx->v_hd = c.v_head_dim ? c.v_head_dim : 128;
  ctx->v_Cq = c.v_heads * ctx->v_hd;
  const int K = 3 * c.v_patch * c.v_patch;
  ctx->v_kpad = cdiv(K, 64) * 64;
  const std::string TW = "model.vision_tower.vision_tower.";

  // ---- vision tower weights
  if (c.v_layers > 0) {
    TRY(add_route(ctx, TW + "embeddings.class_embedding", &ctx->v_cls, 1, C, 0.02f, 0.f));
    TRY(add_route(ctx, TW + "embeddings.position_embedding", &ctx->v_pos, ctx->v_ntok, C, 0.02f, 0.f));
    TRY(ctx->alloc(&ctx->v_wpatch, (size_t)C * ctx->v_kpad * 2));
    OM_HIP(hipMemset(ctx->v_wpatch, 0, (size_t)C * ctx->v_kpad * 2));
    TRY(add_route(ctx, TW + "embeddings.patch_embedding.weight", nullptr, C, K, 0.02f, 0.f, R_PATCH, ctx->v_wpatch, ctx->v_kpad));
    TRY(add_route(ctx, TW + "embeddings.patch_embedding.bias", &ctx->v_bpatch, 1, C, 0.02f, 0.f));
    ctx->vl.resize(c.v_layers);
    for (int j = 0; j < c.v_layers; ++j) {
      const std::string P = TW + "encoder.layers." + std::to_string(j) + ".";
      auto& L = ctx->vl[j];
      TRY(add_route(ctx, P + "ls1", &L.ls1, 1, C, 0.02f, 0.1f));
      TRY(add_route(ctx, P + "ls2", &L.ls2, 1, C, 0.02f, 0.1f));
      TRY(add_route(ctx, P + "norm1.weight", &L.n1, 1, C, 0.05f, 1.f));
      TRY(add_route(ctx, P + "norm2.weight", &L.n2, 1, C, 0.05f, 1.f));
      L.n1b = L.n2b = L.qn = L.kn = nullptr;
      if (c.v_norm_type == 1) {
        TRY(add_route(ctx, P + "norm1.bias", &L.n1b, 1, C, 0.02f, 0.f));
        TRY(add_route(ctx, P + "norm2.bias", &L.n2b, 1, C, 0.02f, 0.f));
      }
      TRY(add_route(ctx, P + "attn.qkv.weight", &L.wqkv, 3 * ctx->v_Cq, C, 0.02f, 0.f));
      if (!c.v_no_qk_norm) {
        TRY(add_route(ctx, P + "attn.q_norm.weight", &L.qn, 1, ctx->v_Cq, 0.05f, 1.f));
        TRY(add_route(ctx, P + "attn.k_norm.weight", &L.kn, 1, ctx->v_Cq, 0.05f, 1.f));
      }
      TRY(add_route(ctx, P + "attn.proj.weight", &L.wproj, C, ctx->v_Cq, 0.02f, 0.f));
      TRY(add_route(ctx, P + "attn.proj.bias", &L.bproj, 1, C, 0.02f, 0.f));
      TRY(add_route(ctx, P + "mlp.fc1.weight", &L.w1, I, C, 0.02f, 0.f));
      TRY(add_route(ctx, P + "mlp.fc1.bias", &L.b1, 1, I, 0.02f, 0.f));
      TRY(add_route(ctx, P + "mlp.fc2.weight", &L.w2, C, I, 0.02f, 0.f));
      TRY(add_route(ctx, P + "mlp.fc2.bias", &L.b2, 1, C, 0.02f, 0.f));
    }
    const int H = c.t_hidden;
    TRY(add_route(ctx, "model.mm_projector.0.weight", &ctx->p_w0, H, C, 0.02f, 0.f));
    TRY(add_route(ctx, "model.mm_projector.0.bias", &ctx->p_b0, 1, H, 0.02f, 0.f));
    TRY(add_route(ctx, "model.mm_projector.2.weight", &ctx->p_w2, H, H, 0.02f, 0.f));
    TRY(add_route(ctx, "model.mm_projector.2.bias", &ctx->p_b2, 1, H, 0.02f, 0.f));
  }

  // ---- decoder weights (q/k/v fused; gate/up interleaved in 16-row blocks for the SwiGLU epilogue)
  const int H = c.t_hidden, It = c.t_mlp;
  ctx->t_qdim = c.t_heads * 128; ctx->t_kvdim = c.t_kv_heads * 128; ctx->t_qkvdim = ctx->t_qdim + 2 * ctx->t_kvdim;
  if (c.t_layers > 0) {
    OM_CHECK(It % 16 == 0, "t_mlp must be a multiple of 16");
    TRY(add_route(ctx, "model.embed_tokens.weight", &ctx->t_embed, c.t_vocab_total, H, 0.02f, 0.f));
    TRY(add_route(ctx, "model.norm.weight", &ctx->t_norm, 1, H, 0.05f, 1.f));
    TRY(add_route(ctx, "lm_head.weight", &ctx->t_lm, c.t_vocab, H, 0.02f, 0.f));
    ctx->dl.resize(c.t_layers);
    for (int i = 0; i < c.t_layers; ++i) {
      const std::string P = "model.layers." + std::to_string(i) + ".";
      auto& L = ctx->dl[i];
      TRY(ctx->alloc(&L.wqkv, (size_t)ctx->t_qkvdim * H * 2));
      TRY(ctx->alloc(&L.bqkv, (size_t)ctx->t_qkvdim * 2));
      TRY(ctx->alloc(&L.wgu, (size_t)2 * It * H * 2));
      char* wq = (char*)L.wqkv; char* bq = (char*)L.bqkv;
      TRY(add_route(ctx, P + "self_attn.q_proj.weight", nullptr, ctx->t_qdim, H, 0.02f, 0.f, R_PLAIN, wq));
      TRY(add_route(ctx, P + "self_attn.k_proj.weight", nullptr, ctx->t_kvdim, H, 0.02f, 0.f, R_PLAIN, wq + (size_t)ctx->t_qdim * H * 2));
      TRY(add_route(ctx, P + "self_attn.v_proj.weight", nullptr, ctx->t_kvdim, H, 0.02f, 0.f, R_PLAIN, wq + (size_t)(ctx->t_qdim + ctx->t_kvdim) * H * 2));
      TRY(add_route(ctx, P + "self_attn.q_proj.bias", nullptr, 1, ctx->t_qdim, 0.02f, 0.f, R_PLAIN, bq));
      TRY(add_route(ctx, P + "self_attn.k_proj.bias", nullptr, 1, ctx->t_kvdim, 0.02f, 0.f, R_PLAIN, bq + (size_t)ctx->t_qdim * 2));
      TRY(add_route(ctx, P + "self_attn.v_proj.bias", nullptr, 1, ctx->t_kvdim, 0.02f, 0.f, R_PLAIN, bq + (size_t)(ctx->t_qdim + ctx->t_kvdim) * 2));
      TRY(add_route(ctx, P + "self_attn.o_proj.weight", &L.wo, H, ctx->t_qdim, 0.02f, 0.f));
      TRY(add_route(ctx, P + "mlp.gate_proj.weight", nullptr, It, H, 0.02f, 0.f, R_GATE, L.wgu));
      TRY(add_route(ctx, P + "mlp.up_proj.weight", nullptr, It, H, 0.02f, 0.f, R_UP, L.wgu));
      TRY(add_route(ctx, P + "mlp.down_proj.weight", &L.wd, H, It, 0.02f, 0.f));
      TRY(add_route(ctx, P + "input_layernorm.weight", &L.ln1, 1, H, 0.05f, 1.f));
      TRY(add_route(ctx, P + "post_attention_layernorm.weight", &L.ln2, 1, H, 0.05f, 1.f));
    }
    // RoPE table: Qwen2RotaryEmbedding (modeling_qwen2.py:64-102): inv_freq and angles in fp32, cos/sin fp32
    std::vector<float> tab((size_t)c.max_seq * 128);
    for (int i = 0; i < 64; ++i) {
      const float inv = (float)(1.0 / pow((double)c.rope_theta, (double)((float)(2 * i) / 128.0f)));
      for (int pos = 0; pos < c.max_seq; ++pos) {
        const float ang = (float)pos * inv;
        tab[((size_t)pos * 64 + i) * 2] = (float)cos((double)ang);
        tab[((size_t)pos * 64 + i) * 2 + 1] = (float)sin((double)ang);
      }
    }
    TRY(ctx->alloc((void**)&ctx->rope, tab.size() * 4));
    OM_HIP(hipMemcpy(ctx->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  }

  // ---- workspaces
  if (c.v_layers > 0) {
    const size_t Bm = (size_t)c.max_tiles, M = Bm * ctx->v_ntok;
    TRY(ctx->alloc(&ctx->vw_cols, Bm * ctx->v_np * ctx->v_kpad * 2));
    TRY(ctx->alloc(&ctx->vw_pe, Bm * ctx->v_np * C * 2));
    const size_t SL = 64;      // row slack: the last row block of a sequence-parallel reduce-scatter reaches up to tp_size - 1 rows beyond M
    TRY(ctx->alloc(&ctx->vw_x, (M + SL) * C * 2));
    TRY(ctx->alloc(&ctx->vw_x2, (M + SL) * C * 2));
    TRY(ctx->alloc(&ctx->vw_xn, (M + SL) * C * 2));
    TRY(ctx->alloc(&ctx->vw_qkv, M * 3 * ctx->v_Cq * 2));
    TRY(ctx->alloc(&ctx->vw_ao, M * ctx->v_Cq * 2));
    TRY(ctx->alloc(&ctx->vw_h, M * I * 2));
    TRY(ctx->alloc(&ctx->vw_feat, M * C * 2));
    TRY(ctx->alloc(&ctx->vw_proj, M * c.t_hidden * 2));
    TRY(ctx->alloc((void**)&ctx->vw_sumsq, M * 2 * 4 + 64));
    // statistics slots (slot-major [slot][M]): one per wave-column block of the producing GEMM, at least 48 columns wide (gemm.hip launch_epi_stats)
    ctx->vw_stats_x_ld = ctx->vw_stats_qk_ld = (int)((M + 63) / 64 * 64);
    TRY(ctx->alloc((void**)&ctx->vw_stats_x, (size_t)(cdiv(C, 48) + 2) * ctx->vw_stats_x_ld * 4));            // one slot per wave tile (>= 48 columns)
    TRY(ctx->alloc((void**)&ctx->vw_stats_qk, (size_t)(cdiv(3 * ctx->v_Cq, 64) + 1) * ctx->vw_stats_qk_ld * 4));   // qkv: one per 64
    TRY(ctx->alloc((void**)&ctx->vw_rstd, M * 4 + 64));
  }
  if (c.t_layers > 0) {
    const size_t R = std::max<size_t>(32, (size_t)(c.max_prefill_rows > c.max_batch ? c.max_prefill_rows : c.max_batch));      // >= 32: packed decode operands
    TRY(ctx->alloc(&ctx->tw_x, (R + 64) * H * 2));       // (+ 64 rows: slack of the sequence-parallel row blocks, see vw_x)
    TRY(ctx->alloc(&ctx->tw_x2, (R + 64) * H * 2));
    TRY(ctx->alloc(&ctx->tw_xn, (R + 64) * H * 2));
    TRY(ctx->alloc(&ctx->tw_qkv, R * ctx->t_qkvdim * 2));
    TRY(ctx->alloc(&ctx->tw_ao, R * ctx->t_qdim * 2));
    TRY(ctx->alloc(&ctx->tw_act, R * It * 2));
    TRY(ctx->alloc(&ctx->tw_last, (size_t)c.max_batch * H * 2));
    TRY(ctx->alloc((void**)&ctx->tw_logits, (size_t)c.max_batch * c.t_vocab * 4));
    ctx->tw_attn_ws_bytes = attn_decode_ws_bytes(c.max_batch, c.t_heads, c.max_seq);
    TRY(ctx->alloc((void**)&ctx->tw_attn_ws, ctx->tw_attn_ws_bytes));
    TRY(ctx->alloc((void**)&ctx->tw_part, (size_t)DEC_KS_MAX * c.max_batch * H * 4));
    if (OMCHAT_EXPERIMENTS) {
      const size_t fb = fused_decode_ws_bytes(c.t_heads);
      TRY(ctx->alloc(&ctx->fd_ws, fb));
      TRY(ctx->alloc((void**)&ctx->fd_err, 64));
      OM_HIP(hipMemset(ctx->fd_ws, 0, fb));
      OM_HIP(hipMemset(ctx->fd_err, 0, 64));
      const size_t lb = decode_layer_ws_bytes(c.t_heads, H, ctx->t_qdim, ctx->t_kvdim, It);
      TRY(ctx->alloc(&ctx->dl_ws, lb));
      OM_HIP(hipMemset(ctx->dl_ws, 0, lb));
      TRY(ctx->alloc((void**)&ctx->dyn_ctr, (size_t)c.t_layers * 65 * 64 * 4));
      OM_HIP(hipMemset(ctx->dyn_ctr, 0, (size_t)c.t_layers * 65 * 64 * 4));
      TRY(ctx->alloc((void**)&ctx->dbg_stamps, (size_t)c.t_layers * 256));
      OM_HIP(hipMemset(ctx->dbg_stamps, 0, (size_t)c.t_layers * 256));
    }
    TRY(ctx->alloc(&ctx->arg_scratch, argmax_scratch_bytes(c.max_batch)));
    TRY(ctx->alloc((void**)&ctx->tp_table, (size_t)ctx->tp_size * c.max_batch * 2 * 4));
    TRY(ctx->alloc((void**)&ctx->d_pos, (size_t)c.max_batch * 4));
    TRY(ctx->alloc((void**)&ctx->d_len, (size_t)c.max_batch * 4));
    TRY(ctx->alloc((void**)&ctx->d_idx, (size_t)c.max_batch * 4));
    TRY(ctx->alloc((void**)&ctx->d_start, (size_t)c.max_batch * 4));
    const size_t cache = (size_t)c.t_layers * ctx->cache_layer_stride() * 2;
    TRY(ctx->alloc(&ctx->kcache, cache));
    TRY(ctx->alloc(&ctx->vcache, cache));
    OM_HIP(hipMemset(ctx->kcache, 0, cache));
    OM_HIP(hipMemset(ctx->vcache, 0, cache));
    ctx->h_len.assign(c.max_batch, 0);
  }
  return 0;
}

int ensure_stage(omchat_ctx* ctx, size_t f32_bytes, size_t t_bytes) {
  if (f32_bytes > ctx->stage_f32_bytes) {
    if (ctx->stage_f32) hipFree(ctx->stage_f32);
    OM_HIP(hipMalloc(&ctx->stage_f32, f32_bytes));
    ctx->stage_f32_bytes = f32_bytes;
  }
  if (t_bytes > ctx->stage_t_bytes) {
    if (ctx->stage_t) hipFree(ctx->stage_t);
    OM_HIP(hipMalloc(&ctx->stage_t, t_bytes));
    ctx->stage_t_bytes = t_bytes;
  }
  return 0;
}

// copy a contiguous [rows, cols] T tensor (host or device) into its routed destination
int place(omchat_ctx* ctx, Route& r, const void* src_t) {
  const size_t rowb = (size_t)r.cols * 2;
  if (r.kind == R_PLAIN || r.kind == R_PATCH) {
    OM_HIP(hipMemcpy2D(r.dst, (size_t)r.dst_ld * 2, src_t, rowb, rowb, (size_t)r.rows, hipMemcpyDefault));
  } else {
    // gate block j (16 rows) -> fused rows [32j, 32j+16); up block j -> [32j+16, 32j+32)
    char* d = (char*)r.dst + (r.kind == R_UP ? 16 * rowb : 0);
    OM_HIP(hipMemcpy2D(d, 32 * rowb, src_t, 16 * rowb, 16 * rowb, (size_t)r.rows / 16, hipMemcpyDefault));
  }
  r.loaded = true;
  ctx->vfold_stale = true;      // (any tensor: the folded qkv / fc1 copies of the vision tower are rebuilt before the next tower pass)
  return 0;
}

}  // namespace

extern "C" int omchat_ctx_create(const omchat_config* cfg, int tp_rank, int tp_size, void* rccl_comm, omchat_ctx** out) {
  OM_CHECK(cfg && out, "null argument");
  OM_CHECK(cfg->dtype == OMCHAT_F16 || cfg->dtype == OMCHAT_BF16, "dtype must be OMCHAT_F16 or OMCHAT_BF16");
  OM_CHECK(tp_size >= 1 && tp_rank >= 0 && tp_rank < tp_size, "bad tensor-parallel rank/size");
  OM_CHECK(cfg->v_head_dim == 0 || cfg->v_head_dim == 128 || cfg->v_head_dim == 64, "v_head_dim must be 128 or 64");
  OM_CHECK(cfg->v_norm_type == 0 || cfg->v_norm_type == 1, "v_norm_type must be 0 (RMSNorm) or 1 (LayerNorm)");
  OM_CHECK(cfg->v_layers == 0 || (cfg->v_hidden % 64 == 0 && cfg->v_mlp % 64 == 0 && cfg->v_image % cfg->v_patch == 0),
           "vision dims: hidden/mlp % 64, image % patch");
  OM_CHECK(cfg->t_layers == 0 || (cfg->t_hidden % 64 == 0 && cfg->t_mlp % 64 == 0 && cfg->t_heads % cfg->t_kv_heads == 0),
           "text dims: hidden/mlp % 64, heads % kv_heads");
  OM_CHECK(cfg->t_layers == 0 || cfg->t_heads / cfg->t_kv_heads <= 16, "GQA group must be <= 16");
  int dev_count = 0;
  if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) {
    omchat_set_error("omchat_ctx_create: no HIP device (the HIP path has no CPU fallback)");
    return 2;
  }
  omchat_ctx* ctx = new omchat_ctx();
  ctx->c = *cfg; ctx->dt = cfg->dtype; ctx->tp_rank = tp_rank; ctx->tp_size = tp_size; ctx->comm = (ncclComm_t)rccl_comm;
  int rc = build(ctx);
  if (rc == 0 && tp_size > 1) {      // communication stream + events of the pipelined all-reduce (gemm_allreduce)
    auto mk = [&]() -> int {
      OM_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
      for (auto& e : ctx->ev_chunk) OM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto& e : ctx->ev_ag) OM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      OM_HIP(hipEventCreateWithFlags(&ctx->ev_comm_done, hipEventDisableTiming));
      return 0;
    };
    rc = mk();
  }
  if (rc) { omchat_ctx_destroy(ctx); return rc; }
  *out = ctx;
  return 0;
}

extern "C" void omchat_ctx_destroy(omchat_ctx* ctx) {
  if (!ctx) return;
  for (void* p : ctx->allocs) hipFree(p);
  for (auto& pr : ctx->prof) for (hipEvent_t e : pr.ev) hipEventDestroy(e);
  for (auto& kv : ctx->graphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
  }
  if (ctx->graph_ev_in) (void)hipEventDestroy(ctx->graph_ev_in);
  if (ctx->graph_ev_out) (void)hipEventDestroy(ctx->graph_ev_out);
  if (ctx->graph_stream) (void)hipStreamDestroy(ctx->graph_stream);
  for (hipEvent_t e : ctx->ev_chunk) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_ag) if (e) (void)hipEventDestroy(e);
  if (ctx->ev_comm_done) (void)hipEventDestroy(ctx->ev_comm_done);
  if (ctx->tp_f32_ws) (void)hipFree(ctx->tp_f32_ws);
  if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
  if (ctx->stage_f32) hipFree(ctx->stage_f32);
  if (ctx->stage_t) hipFree(ctx->stage_t);
  delete ctx;
}

extern "C" size_t omchat_device_bytes(omchat_ctx* ctx) { return ctx ? ctx->bytes : 0; }

extern "C" int omchat_load_tensor(omchat_ctx* ctx, const char* name, const void* data, const int64_t* shape, int ndim, int src_dtype) {
  OM_CHECK(ctx && name && data && shape, "null argument");
  auto it = ctx->routes.find(name);
  OM_CHECK(it != ctx->routes.end(), std::string("unknown tensor name: ") + name);
  Route& r = it->second;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= shape[i];
  OM_CHECK(n == r.rows * r.cols, std::string("shape mismatch for ") + name + ": expected " + std::to_string(r.rows * r.cols) +
                                     " elements, got " + std::to_string(n));
  {   // the leading dims must multiply to the routed row count (catches a transposed [cols, rows] tensor of the right size)
    bool split_ok = r.rows == 1 || ndim == 1;
    int64_t lead = 1;
    for (int i = 0; i < ndim && !split_ok; ++i) { lead *= shape[i]; split_ok = lead == r.rows; }
    OM_CHECK(split_ok, std::string("shape mismatch for ") + name + ": expected [" + std::to_string(r.rows) + ", " + std::to_string(r.cols) + "]");
  }
  if (!ctx->dl8.empty()) ctx->fp8_stale = true;      // the e4m3 replica no longer matches the 16-bit weights
  ctx->pk_ready = false;                              // nor does the packed replica
  if (src_dtype == ctx->dt) return place(ctx, r, data);
  OM_CHECK(src_dtype == OMCHAT_F32, "source dtype must be the context dtype or OMCHAT_F32");
  TRY(ensure_stage(ctx, (size_t)n * 4, (size_t)n * 2));
  OM_HIP(hipMemcpy(ctx->stage_f32, data, (size_t)n * 4, hipMemcpyDefault));
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (ctx->dt == OMCHAT_F16) hipLaunchKernelGGL(cast_from_f32_f16, dim3(grid), dim3(256), 0, 0, (const float*)ctx->stage_f32, (f16*)ctx->stage_t, n);
  else hipLaunchKernelGGL(cast_from_f32_bf16, dim3(grid), dim3(256), 0, 0, (const float*)ctx->stage_f32, (bf16*)ctx->stage_t, n);
  OM_LAUNCH_CHECK();
  OM_HIP(hipDeviceSynchronize());
  return place(ctx, r, ctx->stage_t);
}

static uint64_t fnv1a64(const std::string& s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (unsigned char ch : s) { h ^= ch; h *= 0x100000001B3ull; }
  return h;
}

extern "C" int omchat_fill_synthetic(omchat_ctx* ctx, uint64_t seed) {
  OM_CHECK(ctx, "null ctx");
  // tp_size > 1: every rank fills its LOCAL shapes (right geometry for benches; not a sharding of the TP=1 values)
  size_t maxn = 0;
  for (auto& kv : ctx->routes) maxn = std::max(maxn, (size_t)(kv.second.rows * kv.second.cols));
  TRY(ensure_stage(ctx, 16, maxn * 2));
  for (auto& kv : ctx->routes) {
    Route& r = kv.second;
    const int64_t n = r.rows * r.cols;
    const float scale = (float)((double)r.synth_std * sqrt(3.0));
    TRY(launch_fill_uniform(ctx->dt, ctx->stage_t, n, fnv1a64(kv.first) ^ seed, scale, r.synth_off, 0));
    OM_HIP(hipDeviceSynchronize());
    TRY(place(ctx, r, ctx->stage_t));
  }
  return 0;
}

extern "C" int omchat_weights_missing(omchat_ctx* ctx) {
  int miss = 0;
  std::string names;
  for (auto& kv : ctx->routes)
    if (!kv.second.loaded) { if (miss < 8) names += kv.first + " "; ++miss; }
  if (miss) omchat_set_error("missing tensors: " + names + (miss > 8 ? "..." : ""));
  return miss;
}

// ---------------------------------------------------------------------------------------------------------
// vision tower
// ---------------------------------------------------------------------------------------------------------
static int gemm(omchat_ctx* ctx, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const void* bias,
                const void* ls, const void* resid, int ldr, int epi, hipStream_t s) {
  GemmArgs g{A, lda, W, ldw, C, ldc, M, N, K, bias, ls, resid, ldr, epi, 0, nullptr, 0, -1};
  return launch_gemm(ctx->dt, g, s);
}

int g_norm_in_gemv = 3;       // omchat_op_set_tuning key 14: bit 0 = the post-attention RMSNorm runs inside the gate|up GEMV, bit 1 = the input / final norm
                              // inside qkv / lm_head with down_proj un-split (0 = batch-1 decode keeps both residual + RMSNorm launches: A/B);
                              // bit 2 (experiments build only, measured slower: 4.57-4.87 vs 4.28 ms per step) = the seven-launch batched layer
void model_set_norm_in_gemv(int v) { g_norm_in_gemv = v; }
int g_pack_replica = 1;       // omchat_op_set_tuning key 6: 0 = batched decode reads the row-major weights (packed x only)
void model_set_pack_replica(int v) { g_pack_replica = v; }
int g_ar_min_rows = 1024;      // rows from which a projection is pipelined in 2 (x3: 4) chunks; tests lower it (omchat_op_set_tuning key 4)
void model_set_ar_min_rows(int v) { g_ar_min_rows = v > 8 ? v : 8; }

// Row-parallel projection (proj / fc2 / o_proj / down_proj under tensor parallelism): Y = epi(A W^T) holds this rank's partial
// sums and is all-reduced in place.  The rows are cut into up to AR_CHUNKS chunks of whole 256-row tiles; the all-reduce of
// chunk i is enqueued on the communication stream behind an event and runs under the GEMM of chunk i + 1 (RCCL's ring kernels
// take a few CUs, the GEMM the rest).  The launch stream resumes after the last chunk's all-reduce.
int g_tp_f32 = 0;      // omchat_op_set_tuning key 29: 1 = row-parallel projections of the prefill / the ViT are all-reduced as fp32 partial sums and the
                       // epilogue (bias, layer scale, residual) is applied once to the sum (launch_tp_finish); 0 (default) = every rank applies the
                       // epilogue to its own partial and the 16-bit results are summed (half the bytes on the links)
void model_set_tp_f32(int v) { g_tp_f32 = v; }

// bias / resid: the true operands on EVERY rank (the 16-bit path hands them to rank 0 only)
static int gemm_allreduce(omchat_ctx* ctx, const void* A, int lda, const void* W, int ldw, void* Y, int N, int M, int K, const void* bias_all,
                          const void* ls, const void* resid_all, int epi, hipStream_t s) {
  const bool lead = ctx->tp_rank == 0;
  const void* bias = lead ? bias_all : nullptr;
  const void* resid = lead ? resid_all : nullptr;
  int nch = M >= 3 * g_ar_min_rows ? 4 : (M >= g_ar_min_rows ? 2 : 1);
  if (!ctx->comm_stream) nch = 1;
  // A chunk must still fill the GPU: cutting the 3-tile ViT (M = 3075: 169 tiles of 256^2) or the single-sequence prefill (S = 3584: 196
  // tiles) into 1024-row chunks ran every projection as 3-4 launches on 52-78 of the 256 CUs (bench.py --shard-of 8, profiles/r03_a:
  // 3 x 34 us for a projection that takes ~40 us in one launch).  The all-reduce of such a message is then exposed, but it is shorter
  // than the lost GEMM time.  (Tests lower g_ar_min_rows below one tile to force chunking on tiny shapes: the rule is skipped there.)
  if (g_ar_min_rows >= 256) {
    const int n_cu = device_cus();
    const long tiles = (long)cdiv(M, 256) * cdiv(N, 256);
    while (nch > 1 && tiles / nch < n_cu) nch >>= 1;
  }
  const int align = g_ar_min_rows >= 256 ? 256 : 8;
  const int rows_per = nch == 1 ? M : cdiv(cdiv(M, nch), align) * align;
  // fp32 partial sums (tuning key 29): GEMM -> raw accumulators, fp32 all-reduce, one epilogue on the sum
  const bool f32 = g_tp_f32 && N % 4 == 0 && (epi == EPI_NONE || epi == EPI_RESID || epi == EPI_LS_RESID);
  if (f32) {
    const size_t need = (size_t)M * N * 4;
    if (ctx->tp_f32_bytes < need) {
      OM_HIP(hipStreamSynchronize(s));
      if (ctx->comm_stream) OM_HIP(hipStreamSynchronize(ctx->comm_stream));
      if (ctx->tp_f32_ws) (void)hipFree(ctx->tp_f32_ws);
      ctx->tp_f32_ws = nullptr; ctx->tp_f32_bytes = 0;
      OM_HIP(hipMalloc((void**)&ctx->tp_f32_ws, need));
      ctx->tp_f32_bytes = need;
    }
  }
  if (nch == 1) {
    if (f32) {
      TRY(gemm(ctx, A, lda, W, ldw, ctx->tp_f32_ws, N, M, N, K, nullptr, nullptr, nullptr, 0, EPI_F32OUT, s));
      TRY(ctx->allreduce_f32(ctx->tp_f32_ws, (size_t)M * N, s));
      return launch_tp_finish(ctx->dt, ctx->tp_f32_ws, bias_all, ls, resid_all, Y, M, N, epi, s);
    }
    TRY(gemm(ctx, A, lda, W, ldw, Y, N, M, N, K, bias, ls, resid, N, epi, s));
    return ctx->allreduce(Y, (size_t)M * N, s);
  }
  int i = 0;
  for (int r0 = 0; r0 < M; r0 += rows_per, ++i) {
    const int rows = std::min(rows_per, M - r0);
    char* y = (char*)Y + (size_t)r0 * N * 2;
    if (f32) {
      float* part = ctx->tp_f32_ws + (size_t)r0 * N;
      TRY(gemm(ctx, (const char*)A + (size_t)r0 * lda * 2, lda, W, ldw, part, N, rows, N, K, nullptr, nullptr, nullptr, 0, EPI_F32OUT, s));
      OM_HIP(hipEventRecord(ctx->ev_chunk[i], s));
      OM_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_chunk[i], 0));
      TRY(ctx->allreduce_f32(part, (size_t)rows * N, ctx->comm_stream));
      TRY(launch_tp_finish(ctx->dt, part, bias_all, ls, resid_all ? (const char*)resid_all + (size_t)r0 * N * 2 : nullptr, y, rows, N, epi, ctx->comm_stream));
      continue;
    }
    TRY(gemm(ctx, (const char*)A + (size_t)r0 * lda * 2, lda, W, ldw, y, N, rows, N, K, bias, ls,
             resid ? (const char*)resid + (size_t)r0 * N * 2 : nullptr, N, epi, s));
    OM_HIP(hipEventRecord(ctx->ev_chunk[i], s));
    OM_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_chunk[i], 0));
    TRY(ctx->allreduce(y, (size_t)rows * N, ctx->comm_stream));
  }
  OM_HIP(hipEventRecord(ctx->ev_comm_done, ctx->comm_stream));
  OM_HIP(hipStreamWaitEvent(s, ctx->ev_comm_done, 0));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Sequence-parallel norms (round 6; tuning key 45, default on under tensor parallelism).  The all-reduce form above leaves the WHOLE residual
// stream on every rank, and every rank then normalises all M rows: RMSNorm (2 x 88 us per ViT layer at 24 tiles) does not shrink with the group
// size -- 16 % of a TP = 8 ViT layer (profiles/r03_a).  Here a row-parallel projection ends in a reduce-scatter over ROW BLOCKS: rank r receives the
// summed rows it owns, adds the residual and normalises THOSE rows (M / tp_size of them, one launch), and an all-gather hands every rank the
// normalised activation the next column-parallel GEMM needs.  Same bytes on the links as the all-reduce (reduce-scatter + all-gather ARE its two
// halves), the replicated norm work divided by the group size.  The residual stream itself stays row-sharded between sub-blocks (rank r's copy is
// current on its own rows only) and is gathered once at the end of the tower / the prefill.  Megatron-LM's sequence parallelism restricted to the
// norms: the GEMMs and the attention keep their head / column shards.  No reference counterpart (builder.py:22-25 is device_map="auto").
//   rows of a chunk [r0, r0 + rows): block = cdiv(rows, tp_size) rows per rank; rank r owns [r0 + r * block, r0 + min((r + 1) * block, rows))
//   partials: every rank T(T(acc + (rank 0 ? bias : 0)) * ls) (ViT) or T(acc) (decoder); the owner adds the residual AFTER the sum -- one rounding
//   point moved against the all-reduce form (rank 0 added it before), inside the evaluation-order noise of 16-bit TP
// ---------------------------------------------------------------------------------------------------------
int g_tp_sp = 1;
void model_set_tp_sp(int v) { g_tp_sp = v; }

struct SpPlan { int nch, rows_per; };
static SpPlan sp_plan(omchat_ctx* ctx, int M, int N) {      // the chunk rule of gemm_allreduce (a chunk must still fill the GPU)
  int nch = M >= 3 * g_ar_min_rows ? 4 : (M >= g_ar_min_rows ? 2 : 1);
  if (!ctx->comm_stream) nch = 1;
  if (g_ar_min_rows >= 256) {
    const int n_cu = device_cus();
    const long tiles = (long)cdiv(M, 256) * cdiv(N, 256);
    while (nch > 1 && tiles / nch < n_cu) nch >>= 1;
  }
  const int unit = 8 * ctx->tp_size;                                    // whole row blocks inside every chunk but the last
  const int align = g_ar_min_rows >= 256 ? cdiv(256, unit) * unit : unit;
  const int rows_per = nch == 1 ? M : cdiv(cdiv(M, nch), align) * align;
  return {nch, rows_per};
}

// x (the residual stream, current on this rank's rows) <- x + sum over ranks of epi'(A W^T);  xn <- norm(x) on every row of every rank (nw == null: none)
static int gemm_sp(omchat_ctx* ctx, const void* A, int lda, const void* W, int ldw, void* P, void* x, int N, int M, int K, const void* bias_all, const void* ls,
                   int epi, const void* nw, const void* nb, void* xn, float eps, hipStream_t s) {
  const bool lead = ctx->tp_rank == 0;
  const SpPlan pl = sp_plan(ctx, M, N);
  const bool pipe = pl.nch > 1;
  hipStream_t cs = pipe ? ctx->comm_stream : s;
  int i = 0;
  for (int r0 = 0; r0 < M; r0 += pl.rows_per, ++i) {
    const int rows = std::min(pl.rows_per, M - r0);
    const int blk = cdiv(rows, ctx->tp_size);
    char* p = (char*)P + (size_t)r0 * N * 2;
    // the partial: bias on rank 0 only, layer scale on every rank (it distributes over the sum), NO residual
    TRY(gemm(ctx, (const char*)A + (size_t)r0 * lda * 2, lda, W, ldw, p, N, rows, N, K, lead ? bias_all : nullptr, ls, nullptr, 0,
             epi == EPI_LS_RESID ? EPI_LS_RESID : EPI_NONE, s));
    if (pipe) {
      OM_HIP(hipEventRecord(ctx->ev_chunk[i], s));
      OM_HIP(hipStreamWaitEvent(cs, ctx->ev_chunk[i], 0));
    }
    TRY(ctx->reduce_scatter_rows(p, blk, N, cs));
    const int o0 = ctx->tp_rank * blk, on = std::max(0, std::min(blk, rows - o0));      // my rows of this chunk
    char* xo = (char*)x + (size_t)(r0 + o0) * N * 2;
    char* no = nw ? (char*)xn + (size_t)(r0 + o0) * N * 2 : nullptr;
    TRY(launch_resid16_norm(ctx->dt, xo, N, p + (size_t)o0 * N * 2, N, nw, nb, no, N, on, N, eps, cs));
    if (nw) TRY(ctx->all_gather_rows((char*)xn + (size_t)r0 * N * 2, blk, N, cs));
    if (pipe && nw) OM_HIP(hipEventRecord(ctx->ev_ag[i], cs));
  }
  if (pipe && nw) {
    // the launch stream does NOT wait here: the consumer of xn (the next column-parallel GEMM: gemm_after_sp) goes out per row chunk behind
    // ev_ag[i], so chunk i + 1's reduce-scatter / norm / all-gather also runs under the consumer's GEMM of chunk i
    ctx->sp_pend_nch = pl.nch; ctx->sp_pend_rows_per = pl.rows_per;
  } else if (pipe) {
    OM_HIP(hipEventRecord(ctx->ev_comm_done, cs));
    OM_HIP(hipStreamWaitEvent(s, ctx->ev_comm_done, 0));
  }
  return 0;
}

// every exchange of a pending sequence-parallel sub-block is behind the launch stream (no consumer took it chunk by chunk)
static int sp_drain(omchat_ctx* ctx, hipStream_t s) {
  for (int i = 0; i < ctx->sp_pend_nch; ++i) OM_HIP(hipStreamWaitEvent(s, ctx->ev_ag[i], 0));
  ctx->sp_pend_nch = 0;
  return 0;
}

// C = epi(A W^T + bias) where A [M, lda] is the normalised activation a sequence-parallel sub-block is still gathering: one GEMM per row chunk of
// its plan, each behind that chunk's all-gather (a chunk holds >= 256 tiles of the producer, so the consumer's chunk fills the GPU too)
static int gemm_after_sp(omchat_ctx* ctx, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const void* bias, int epi,
                         hipStream_t s) {
  // a consumer chunk must hold two whole rounds of tiles: below that the partial last round of every chunk costs more than the exchange it hides
  // (bench.py --shard-of 4, configs[2]: the ViT's fc1 as four chunks of 312 tiles 380 ms per rank against 372 with one launch)
  const long chunk_tiles = (long)cdiv(ctx->sp_pend_rows_per, 256) * cdiv(N, 256);
  if (ctx->sp_pend_nch <= 1 || (g_ar_min_rows >= 256 && chunk_tiles < 2L * device_cus())) {      // (tests lower g_ar_min_rows: the rule is skipped there)
    TRY(sp_drain(ctx, s));
    return gemm(ctx, A, lda, W, ldw, C, ldc, M, N, K, bias, nullptr, nullptr, 0, epi, s);
  }
  const int nch = ctx->sp_pend_nch, rows_per = ctx->sp_pend_rows_per;
  ctx->sp_pend_nch = 0;
  int i = 0;
  for (int r0 = 0; r0 < M; r0 += rows_per, ++i) {
    const int rows = std::min(rows_per, M - r0);
    OM_HIP(hipStreamWaitEvent(s, ctx->ev_ag[i < nch ? i : nch - 1], 0));
    TRY(gemm(ctx, (const char*)A + (size_t)r0 * lda * 2, lda, W, ldw, (char*)C + (size_t)r0 * ldc * 2, ldc, rows, N, K, bias, nullptr, nullptr, 0, epi, s));
  }
  return 0;
}

// the row-sharded residual stream made whole on every rank (end of the tower / of the prefill): one all-gather per chunk of the same plan
static int sp_gather_x(omchat_ctx* ctx, void* x, int N, int M, hipStream_t s) {
  const SpPlan pl = sp_plan(ctx, M, N);
  for (int r0 = 0; r0 < M; r0 += pl.rows_per) {
    const int rows = std::min(pl.rows_per, M - r0);
    TRY(ctx->all_gather_rows((char*)x + (size_t)r0 * N * 2, cdiv(rows, ctx->tp_size), N, s));
  }
  return 0;
}

int g_vit_fused = 1;      // omchat_op_set_tuning key 44: 1 = the ViT layer with its norms folded into the GEMMs (round 6), 0 = the round-5 launches
void model_set_vit_fused(int v) { g_vit_fused = v; }

// The fused ViT layer needs norm1 / norm2's WEIGHT inside the qkv / fc1 weights: y = w * (x * rstd) feeds a linear map, so
// (w * x * rstd) W^T = rstd * (x (W diag(w))^T): W' = T(W * w) column-wise, once per load; rstd becomes the GEMM's row scale.
static int ensure_vit_folded(omchat_ctx* ctx, hipStream_t s) {
  if (!ctx->vfold_stale && !ctx->vfold.empty()) return 0;
  const omchat_config& c = ctx->c;
  const int C = c.v_hidden, I = c.v_mlp, Cq = ctx->v_Cq;
  if (ctx->vfold.empty()) {
    ctx->vfold.resize(c.v_layers);
    for (auto& f : ctx->vfold) {
      TRY(ctx->alloc(&f.wqkv, (size_t)3 * Cq * C * 2));
      TRY(ctx->alloc(&f.w1, (size_t)I * C * 2));
    }
  }
  for (int j = 0; j < c.v_layers; ++j) {
    TRY(launch_fold_cols(ctx->dt, ctx->vl[j].wqkv, ctx->vl[j].n1, ctx->vfold[j].wqkv, 3 * Cq, C, s));
    TRY(launch_fold_cols(ctx->dt, ctx->vl[j].w1, ctx->vl[j].n2, ctx->vfold[j].w1, I, C, s));
  }
  ctx->vfold_stale = false;
  return 0;
}

static int gemm_x(omchat_ctx* ctx, GemmArgs g, hipStream_t s) { return launch_gemm(ctx->dt, g, s); }

// One InternVisionEncoderLayer (modeling_intern_vit.py:210-222) as SIX launches (round 6; TP = 1, RMSNorm tower):
//   qkv GEMM (A = raw x, W' = Wqkv diag(norm1.w), row scale rstd1 finished per tile from x's slots, epilogue leaves the q / k sum-of-squares slots)
//   K norm (from its slots; leaves the q sums)   attention (Q normed + scaled where it is loaded)   proj GEMM (+b, *ls1, +x; leaves x's slots)
//   fc1 GEMM (A = raw x, W' = W1 diag(norm2.w), row scale rstd2 from x's slots, +b, GELU)   fc2 GEMM (+b, *ls2, +x; leaves x's slots)
// against the round-5 layer's rmsnorm, qkv, q/k norm, attention, proj, rmsnorm, fc1, fc2: two 39 MB norm passes and half of the 79 MB
// q / k pass are gone, and nothing was added between the GEMMs (a first form with three statistics-finishing launches of 4.8 us each gave
// the whole gain back: profiles/r06_c).  Numerics: x * rstd is no longer rounded to 16 bits before the product (one rounding FEWER than the reference,
// N2) and the norm weight is rounded into W' instead of into the activation; the q / k norm keeps the reference's rounding points, its
// statistics are summed per 64-column block first (fp32 order).  Guarded by the reference goldens and the full-depth oracle fixture.
static int vit_layer_fused(omchat_ctx* ctx, int j, void* x, int B, bool last, hipStream_t s) {
  const omchat_config& c = ctx->c;
  const int C = c.v_hidden, I = c.v_mlp, Cq = ctx->v_Cq, ntok = ctx->v_ntok, M = B * ntok, hd = ctx->v_hd;
  auto& L = ctx->vl[j];
  auto& F = ctx->vfold[j];
  const float qscale = 0.08838834764831845f;      // head_dim ** -0.5 (modeling_intern_vit.py:118), head dim 128
  int ns_x = 0;
  if (j == 0) {      // the embeddings came from vit_assemble: one slot
    TRY(launch_row_sumsq(ctx->dt, x, C, M, C, ctx->vw_stats_x, s));
    ctx->vw_stats_x_ld_used = 1;
  }
  {
    GemmArgs g{x, C, F.wqkv, C, ctx->vw_qkv, 3 * Cq, M, 3 * Cq, C, nullptr, nullptr, nullptr, 0, EPI_NONE_STATS, 2, nullptr, 0, -1};
    // norm1 as the row scale, finished per tile from the slots x's producer left
    g.rs_stats = ctx->vw_stats_x; g.rs_ld = ctx->vw_stats_x_ld; g.rs_nslots = ctx->vw_stats_x_ld_used; g.rs_dim = C; g.rs_eps = c.v_eps;
    g.stats = ctx->vw_stats_qk; g.stats_ld = ctx->vw_stats_qk_ld;
    TRY(gemm_x(ctx, g, s));      // (tile 2: 64-column slots, so q = slots [0, Cq / 64), k = [Cq / 64, 2 Cq / 64))
  }
  // K is normalised in place from its slots; the q slots' sum goes to vw_sumsq [M] for the attention kernel's q norm on load
  const int qs = Cq / 64;
  TRY(launch_vit_knorm_slots(ctx->dt, (char*)ctx->vw_qkv + (size_t)Cq * 2, 3 * Cq, L.kn, M, Cq, c.v_qk_channels, c.v_eps, ctx->vw_stats_qk, ctx->vw_stats_qk_ld, qs,
                             ctx->vw_sumsq, s));
  AttnArgs a{};
  a.scale = 1.0f;
  a.Q = ctx->vw_qkv; a.q_sb = (int64_t)ntok * 3 * Cq; a.q_sh = hd; a.q_sr = 3 * Cq;
  a.K = (const char*)ctx->vw_qkv + (size_t)Cq * 2; a.k_sb = a.q_sb; a.k_sh = hd; a.k_sr = 3 * Cq;
  a.V = (const char*)ctx->vw_qkv + (size_t)2 * Cq * 2; a.v_sb = a.q_sb; a.v_sh = hd; a.v_sr = 3 * Cq;
  a.O = ctx->vw_ao; a.o_sb = (int64_t)ntok * Cq; a.o_sh = hd; a.o_sr = Cq;
  a.batch = B; a.q_heads = c.v_heads; a.kv_heads = c.v_heads; a.Sq = ntok; a.Skv = ntok; a.kv_len = nullptr; a.causal = 0; a.q_pos0 = 0;
  a.head_dim = hd;
  a.qn_sumsq = ctx->vw_sumsq; a.qn_stride = 1; a.qn_dim = c.v_qk_channels; a.qn_w = L.qn; a.qn_eps = c.v_eps; a.qn_scale = qscale;
  TRY(launch_attn_prefill(ctx->dt, a, s));
  {
    GemmArgs g{ctx->vw_ao, Cq, L.wproj, Cq, x, C, M, C, Cq, L.bproj, L.ls1, x, C, EPI_LS_RESID_STATS, 0, nullptr, 0, -1};
    g.stats = ctx->vw_stats_x; g.stats_ld = ctx->vw_stats_x_ld; g.stats_nslots = &ns_x;
    TRY(gemm_x(ctx, g, s));
  }
  ctx->prof_mark(OMCHAT_PROF_VIT_FC1, s);
  {
    GemmArgs g{x, C, F.w1, C, ctx->vw_h, I, M, I, C, L.b1, nullptr, nullptr, 0, EPI_GELU, 0, nullptr, 0, -1};
    g.rs_stats = ctx->vw_stats_x; g.rs_ld = ctx->vw_stats_x_ld; g.rs_nslots = ns_x; g.rs_dim = C; g.rs_eps = c.v_eps;      // norm2
    TRY(gemm_x(ctx, g, s));
  }
  ctx->prof_mark(OMCHAT_PROF_VIT_FC1, s);
  {
    GemmArgs g{ctx->vw_h, I, L.w2, I, x, C, M, C, I, L.b2, L.ls2, x, C, last ? EPI_LS_RESID : EPI_LS_RESID_STATS, 0, nullptr, 0, -1};
    if (!last) { g.stats = ctx->vw_stats_x; g.stats_ld = ctx->vw_stats_x_ld; g.stats_nslots = &ns_x; }
    TRY(gemm_x(ctx, g, s));
    if (!last) ctx->vw_stats_x_ld_used = ns_x;
  }
  return 0;
}

static int vit_run(omchat_ctx* ctx, const void* pixels, int B, int n_layers, hipStream_t s, void** x_out) {
  const omchat_config& c = ctx->c;
  const int C = c.v_hidden, I = c.v_mlp, Cq = ctx->v_Cq, np = ctx->v_np, ntok = ctx->v_ntok, M = B * ntok;
  const bool lead = ctx->tp_rank == 0;
  const bool fused = g_vit_fused && ctx->tp_size == 1 && c.v_norm_type == 0 && !c.v_no_qk_norm && ctx->v_hd == 128 && Cq % 64 == 0 && Cq / 64 <= 256;
  if (fused) TRY(ensure_vit_folded(ctx, s));
  // InternVisionEmbeddings.forward (modeling_intern_vit.py:90-102)
  TRY(launch_im2col(ctx->dt, pixels, ctx->vw_cols, B, c.v_image, c.v_patch, ctx->v_kpad, s));
  TRY(gemm(ctx, ctx->vw_cols, ctx->v_kpad, ctx->v_wpatch, ctx->v_kpad, ctx->vw_pe, C, B * np, C, ctx->v_kpad, ctx->v_bpatch, nullptr, nullptr, 0, EPI_NONE, s));
  TRY(launch_vit_assemble(ctx->dt, ctx->vw_pe, ctx->v_cls, ctx->v_pos, ctx->vw_x, B, np, C, s));
  void* x = ctx->vw_x;
  void* y = ctx->vw_x2;
  const bool sp = g_tp_sp && ctx->tp_size > 1 && !g_tp_f32;      // sequence-parallel norms: gemm_sp
  for (int j = 0; j < n_layers; ++j) {
    if (fused) { TRY(vit_layer_fused(ctx, j, x, B, j + 1 == n_layers, s)); continue; }
    auto& L = ctx->vl[j];
    // InternVisionEncoderLayer.forward (modeling_intern_vit.py:210-222)
    const int hd = ctx->v_hd;
    const float qscale = hd == 128 ? 0.08838834764831845f : 0.125f;      // head_dim ** -0.5 (modeling_intern_vit.py:118)
    auto norm = [&](const void* w, const void* b) -> int {      // InternRMSNorm, or nn.LayerNorm for the 300M tower (NORM2FN)
      if (c.v_norm_type == 1) return launch_layernorm(ctx->dt, x, C, w, b, ctx->vw_xn, C, M, C, c.v_eps, s);
      return launch_rmsnorm(ctx->dt, x, C, w, ctx->vw_xn, C, M, C, c.v_eps, s);
    };
    if (!sp || j == 0) TRY(norm(L.n1, L.n1b));      // (sequence-parallel: the previous layer's fc2 left norm1(x) in vw_xn)
    if (sp) TRY(gemm_after_sp(ctx, ctx->vw_xn, C, L.wqkv, C, ctx->vw_qkv, 3 * Cq, M, 3 * Cq, C, nullptr, EPI_NONE, s));
    else TRY(gemm(ctx, ctx->vw_xn, C, L.wqkv, C, ctx->vw_qkv, 3 * Cq, M, 3 * Cq, C, nullptr, nullptr, nullptr, 0, EPI_NONE, s));
    AttnArgs a{};
    if (!c.v_no_qk_norm) {
      const float* sumsq = nullptr;
      if (ctx->tp_size > 1) {     // joint-head norm: sum of squares over ALL ranks' heads
        TRY(launch_vit_qk_sumsq(ctx->dt, ctx->vw_qkv, 3 * Cq, M, Cq, ctx->vw_sumsq, s));
        TRY(ctx->allreduce_f32(ctx->vw_sumsq, (size_t)M * 2, s));
        sumsq = ctx->vw_sumsq;
      }
      TRY(launch_vit_qknorm(ctx->dt, ctx->vw_qkv, 3 * Cq, L.qn, L.kn, M, Cq, c.v_qk_channels, c.v_eps, qscale, sumsq, s));
      a.scale = 1.0f;     // q was pre-scaled by the q/k norm kernel (reference order, modeling_intern_vit.py:148)
    } else {
      a.scale = qscale;     // q * scale is exact for head_dim 64 (2^-3), so scaling the fp32 scores is the same value
    }
    a.Q = ctx->vw_qkv; a.q_sb = (int64_t)ntok * 3 * Cq; a.q_sh = hd; a.q_sr = 3 * Cq;
    a.K = (const char*)ctx->vw_qkv + (size_t)Cq * 2; a.k_sb = a.q_sb; a.k_sh = hd; a.k_sr = 3 * Cq;
    a.V = (const char*)ctx->vw_qkv + (size_t)2 * Cq * 2; a.v_sb = a.q_sb; a.v_sh = hd; a.v_sr = 3 * Cq;
    a.O = ctx->vw_ao; a.o_sb = (int64_t)ntok * Cq; a.o_sh = hd; a.o_sr = Cq;
    a.batch = B; a.q_heads = c.v_heads; a.kv_heads = c.v_heads; a.Sq = ntok; a.Skv = ntok; a.kv_len = nullptr; a.causal = 0; a.q_pos0 = 0;
    a.head_dim = hd;
    TRY(launch_attn_prefill(ctx->dt, a, s));
    if (ctx->tp_size == 1) {
      TRY(gemm(ctx, ctx->vw_ao, Cq, L.wproj, Cq, x, C, M, C, Cq, L.bproj, L.ls1, x, C, EPI_LS_RESID, s));
    } else if (sp) {
      TRY(gemm_sp(ctx, ctx->vw_ao, Cq, L.wproj, Cq, y, x, C, M, Cq, L.bproj, L.ls1, EPI_LS_RESID, L.n2, L.n2b, ctx->vw_xn, c.v_eps, s));
    } else {
      TRY(gemm_allreduce(ctx, ctx->vw_ao, Cq, L.wproj, Cq, y, C, M, Cq, L.bproj, L.ls1, x, EPI_LS_RESID, s));
      std::swap(x, y);
    }
    if (!sp) TRY(norm(L.n2, L.n2b));
    ctx->prof_mark(OMCHAT_PROF_VIT_FC1, s);
    if (sp) TRY(gemm_after_sp(ctx, ctx->vw_xn, C, L.w1, C, ctx->vw_h, I, M, I, C, L.b1, EPI_GELU, s));
    else TRY(gemm(ctx, ctx->vw_xn, C, L.w1, C, ctx->vw_h, I, M, I, C, L.b1, nullptr, nullptr, 0, EPI_GELU, s));
    ctx->prof_mark(OMCHAT_PROF_VIT_FC1, s);
    if (ctx->tp_size == 1) {
      TRY(gemm(ctx, ctx->vw_h, I, L.w2, I, x, C, M, C, I, L.b2, L.ls2, x, C, EPI_LS_RESID, s));
    } else if (sp) {
      const bool more = j + 1 < n_layers;      // the next layer's norm1 on the owned rows; after the last layer the stream is gathered instead
      TRY(gemm_sp(ctx, ctx->vw_h, I, L.w2, I, y, x, C, M, I, L.b2, L.ls2, EPI_LS_RESID, more ? ctx->vl[j + 1].n1 : nullptr, more ? ctx->vl[j + 1].n1b : nullptr,
                  ctx->vw_xn, c.v_eps, s));
      if (!more) TRY(sp_gather_x(ctx, x, C, M, s));
    } else {
      TRY(gemm_allreduce(ctx, ctx->vw_h, I, L.w2, I, y, C, M, I, L.b2, L.ls2, x, EPI_LS_RESID, s));
      std::swap(x, y);
    }
  }
  TRY(sp_drain(ctx, s));
  *x_out = x;
  return 0;
}

static int resolve_layer(omchat_ctx* ctx, int select_layer, int* idx) {
  const int L = ctx->c.v_layers;
  const int i = select_layer >= 0 ? select_layer : L + 1 + select_layer;
  OM_CHECK(i >= 0 && i <= L, "select_layer out of range");
  *idx = i;
  return 0;
}

extern "C" int omchat_vit_forward(omchat_ctx* ctx, const void* pixels, int n_tiles, int select_layer, int keep_cls, void* out, void* stream) {
  OM_CHECK(ctx && pixels && out, "null argument");
  OM_CHECK(ctx->c.v_layers > 0, "context has no vision tower");
  OM_CHECK(n_tiles >= 0, "negative tile count");
  OM_CHECK(omchat_weights_missing(ctx) == 0, std::string(omchat_last_error()));
  int idx; TRY(resolve_layer(ctx, select_layer, &idx));
  hipStream_t s = (hipStream_t)stream;
  const omchat_config& c = ctx->c;
  const int C = c.v_hidden, np = ctx->v_np, ntok = ctx->v_ntok;
  const size_t px_tile = (size_t)3 * c.v_image * c.v_image * 2;
  const int out_tok = keep_cls ? ntok : np;
  for (int t0 = 0; t0 < n_tiles; t0 += c.max_tiles) {
    const int B = std::min(c.max_tiles, n_tiles - t0);
    void* x;
    TRY(vit_run(ctx, (const char*)pixels + t0 * px_tile, B, idx, s, &x));
    // feature_select (internVIT_encoder.py:35-43): 'patch' drops CLS, 'cls_patch' keeps it
    char* o = (char*)out + (size_t)t0 * out_tok * C * 2;
    if (keep_cls) TRY(launch_copy_rows(ctx->dt, x, C, o, C, B * ntok, C, ntok, 0, s));
    else TRY(launch_copy_rows(ctx->dt, x, C, o, C, B * np, C, np, 1, s));
  }
  return 0;
}

extern "C" int omchat_projector_forward(omchat_ctx* ctx, const void* in, int rows, void* out, void* stream) {
  OM_CHECK(ctx && in && out, "null argument");
  OM_CHECK(ctx->c.v_layers > 0, "context has no projector");
  hipStream_t s = (hipStream_t)stream;
  const int C = ctx->c.v_hidden, H = ctx->c.t_hidden;
  const int cap = ctx->c.max_tiles * ctx->v_ntok;
  for (int r0 = 0; r0 < rows; r0 += cap) {
    const int R = std::min(cap, rows - r0);
    const char* a = (const char*)in + (size_t)r0 * C * 2;
    char* o = (char*)out + (size_t)r0 * H * 2;
    // Linear + GELU + Linear (multimodal_projector/builder.py:57-61)
    TRY(gemm(ctx, a, C, ctx->p_w0, C, ctx->vw_proj, H, R, H, C, ctx->p_b0, nullptr, nullptr, 0, EPI_GELU, s));
    TRY(gemm(ctx, ctx->vw_proj, H, ctx->p_w2, H, o, H, R, H, H, ctx->p_b2, nullptr, nullptr, 0, EPI_NONE, s));
  }
  return 0;
}

extern "C" int omchat_encode_images(omchat_ctx* ctx, const void* pixels, int n_tiles, int select_layer, void* out, void* stream) {
  OM_CHECK(ctx && pixels && out, "null argument");
  OM_CHECK(ctx->c.v_layers > 0, "context has no vision tower");
  OM_CHECK(omchat_weights_missing(ctx) == 0, std::string(omchat_last_error()));
  int idx; TRY(resolve_layer(ctx, select_layer, &idx));
  hipStream_t s = (hipStream_t)stream;
  const omchat_config& c = ctx->c;
  const int C = c.v_hidden, H = c.t_hidden, np = ctx->v_np;
  const size_t px_tile = (size_t)3 * c.v_image * c.v_image * 2;
  for (int t0 = 0; t0 < n_tiles; t0 += c.max_tiles) {
    const int B = std::min(c.max_tiles, n_tiles - t0);
    void* x;
    TRY(vit_run(ctx, (const char*)pixels + t0 * px_tile, B, idx, s, &x));
    TRY(launch_copy_rows(ctx->dt, x, C, ctx->vw_feat, C, B * np, C, np, 1, s));
    TRY(omchat_projector_forward(ctx, ctx->vw_feat, B * np, (char*)out + (size_t)t0 * np * H * 2, s));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// splice
// ---------------------------------------------------------------------------------------------------------
extern "C" int omchat_splice_plan(const int64_t* ids, const uint8_t* mask, int b, int T, int n_tok, int n_tiles_avail, int padding_side,
                                  int max_length, int32_t* src_index, int32_t* lengths, int* S_out, int vocab) {
  OM_CHECK(ids && S_out && b > 0 && T > 0 && n_tok >= 0, "bad argument");
  // embed_tokens raises IndexError for an id outside [0, vocab) (omchat_arch.py:139); the device gather would read out of bounds
  if (vocab > 0)
    for (size_t i = 0; i < (size_t)b * T; ++i) {
      if (mask && !mask[i]) continue;
      if (ids[i] != -200 && (ids[i] < 0 || ids[i] >= vocab)) {
        omchat_set_error("omchat_splice_plan: token id " + std::to_string(ids[i]) + " out of range [0, " + std::to_string(vocab) + ") (index out of range in embed_tokens)");
        return 4;
      }
    }
  // pass 1: lengths (omchat_arch.py:115-164); a row without sentinels still consumes one tile slot (:122-129)
  std::vector<int> len(b);
  int cur = 0, S = 0, total_img = 0;
  for (int i = 0; i < b; ++i) {
    int n = 0, n_img = 0;
    for (int t = 0; t < T; ++t) {
      if (mask && !mask[(size_t)i * T + t]) continue;
      if (ids[(size_t)i * T + t] == -200) { n += n_tok; ++n_img; } else ++n;
    }
    cur += n_img == 0 ? 1 : n_img;
    total_img += n_img;
    if (max_length > 0 && n > max_length) n = max_length;
    len[i] = n;
    S = std::max(S, n);
  }
  // text-only call (no tiles, no sentinels) = plain embed_tokens lookup (the reference short-circuits before the splice, :59-70)
  OM_CHECK(cur <= n_tiles_avail || n_tok == 0 || (n_tiles_avail == 0 && total_img == 0), "more <image> sentinels than image tiles (reference: IndexError on image_features)");
  *S_out = S;
  if (lengths) for (int i = 0; i < b; ++i) lengths[i] = len[i];
  if (!src_index) return 0;
  cur = 0;
  for (int i = 0; i < b; ++i) {
    int32_t* row = src_index + (size_t)i * S;
    const int off = padding_side == 1 ? S - len[i] : 0;       // left padding (:176-184) vs right (:185-193)
    for (int s = 0; s < S; ++s) row[s] = INT32_MIN;
    int w = 0, n_img = 0;
    for (int t = 0; t < T; ++t) {
      if (mask && !mask[(size_t)i * T + t]) continue;
      const int64_t id = ids[(size_t)i * T + t];
      if (id == -200) {
        for (int k = 0; k < n_tok; ++k, ++w)
          if (w < len[i]) row[off + w] = -1 - (cur * n_tok + k);
        ++cur; ++n_img;
      } else {
        if (w < len[i]) row[off + w] = (int32_t)id;
        ++w;
      }
    }
    if (n_img == 0) ++cur;
  }
  return 0;
}

extern "C" int omchat_splice_gather(omchat_ctx* ctx, const int32_t* src_index, const void* feats, void* embeds, int rows, void* stream) {
  OM_CHECK(ctx && src_index && embeds, "null argument");
  OM_CHECK(ctx->t_embed, "context has no decoder");
  return launch_gather_rows(ctx->dt, src_index, ctx->t_embed, feats, embeds, rows, ctx->c.t_hidden, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------------------------
static int lm_head_rows(omchat_ctx* ctx, const void* hidden, int n, float* logits, hipStream_t s, bool fp8 = false, bool packed = false) {
  const int H = ctx->c.t_hidden, V = ctx->c.t_vocab;
  for (int r0 = 0; r0 < n; r0 += 32) {
    const int R = std::min(32, n - r0);
    GemvArgs g{(const char*)hidden + (size_t)r0 * H * 2, H, ctx->t_lm, H, logits + (size_t)r0 * V, V, R, V, H, nullptr, nullptr, 0, EPI_NONE, 1};
    if (fp8 && n == 1) { g.W = ctx->t_lm8; g.w_scale = ctx->t_lm8_s; }
    if (packed) {          // hidden is in the packed x layout (n <= 32)
      g.x_packed = 1;
      if (ctx->pk_ready) { g.W = ctx->t_lmP; g.w_packed = 1; }
    }
    TRY(launch_gemv(ctx->dt, g, s));
  }
  return 0;
}

static int ensure_fp8_weights(omchat_ctx* ctx) {
  const omchat_config& c = ctx->c;
  if (ctx->dl8.empty() || ctx->fp8_stale) {
    OM_CHECK(omchat_weights_missing(ctx) == 0, "load the weights before quantising them");
    const int H = c.t_hidden, It = c.t_mlp, qkvd = ctx->t_qkvdim, qd = ctx->t_qdim;
    ctx->dl8.resize(c.t_layers);
    auto quant = [&](const void* W, int N, int K, void** w8, float** sc) -> int {
      if (!*w8) {      // first build allocates; a rebuild after omchat_load_tensor re-quantises in place
        TRY(ctx->alloc(w8, (size_t)N * K));
        TRY(ctx->alloc((void**)sc, (size_t)N * 4));
      }
      return launch_quant_fp8_rows(ctx->dt, W, K, N, K, *w8, K, *sc, nullptr);
    };
    for (int i = 0; i < c.t_layers; ++i) {
      auto& L = ctx->dl[i]; auto& Q = ctx->dl8[i];
      TRY(quant(L.wqkv, qkvd, H, &Q.wqkv, &Q.sqkv));
      TRY(quant(L.wo, H, qd, &Q.wo, &Q.so));
      TRY(quant(L.wgu, 2 * It, H, &Q.wgu, &Q.sgu));
      TRY(quant(L.wd, H, It, &Q.wd, &Q.sd));
    }
    TRY(quant(ctx->t_lm, c.t_vocab, H, &ctx->t_lm8, (float**)&ctx->t_lm8_s));
    OM_HIP(hipDeviceSynchronize());
    ctx->fp8_stale = false;
  }
  return 0;
}

extern "C" int omchat_enable_fp8_decode(omchat_ctx* ctx, int on) {
  OM_CHECK(ctx, "null context");
  OM_CHECK(ctx->c.t_layers > 0, "context has no decoder");
  if (!on) { ctx->fp8_decode = false; return 0; }
  TRY(ensure_fp8_weights(ctx));
  ctx->fp8_decode = true;
  return 0;
}

extern "C" int omchat_enable_fp8_kv(omchat_ctx* ctx, int on) {
  OM_CHECK(ctx, "null context");
  const omchat_config& c = ctx->c;
  OM_CHECK(c.t_layers > 0, "context has no decoder");
  ctx->kv8_valid = false;            // takes effect with the next prefill
  if (!on) { ctx->fp8_kv = false; return 0; }
  if (!ctx->k8cache) {
    const size_t bytes = (size_t)c.t_layers * ctx->cache_layer_stride();
    const size_t sc = (size_t)c.t_layers * ctx->scale_layer_stride() * 4;
    TRY(ctx->alloc(&ctx->k8cache, bytes)); TRY(ctx->alloc(&ctx->v8cache, bytes));
    TRY(ctx->alloc((void**)&ctx->ks8, sc)); TRY(ctx->alloc((void**)&ctx->vs8, sc));
  }
  ctx->fp8_kv = true;
  return 0;
}

extern "C" int omchat_enable_fp8_prefill(omchat_ctx* ctx, int on) {
  OM_CHECK(ctx, "null context");
  const omchat_config& c = ctx->c;
  OM_CHECK(c.t_layers > 0, "context has no decoder");
  if (!on) { ctx->fp8_prefill = false; return 0; }
  OM_CHECK(c.t_hidden % 128 == 0, "fp8 x fp8 prefill GEMMs need hidden_size % 128 == 0");
  TRY(ensure_fp8_weights(ctx));
  if (!ctx->tw_q8) {
    const size_t R = std::max<size_t>(32, (size_t)(c.max_prefill_rows > c.max_batch ? c.max_prefill_rows : c.max_batch));
    TRY(ctx->alloc(&ctx->tw_q8, R * c.t_hidden));
    TRY(ctx->alloc((void**)&ctx->tw_q8s, R * 4));
  }
  ctx->fp8_prefill = true;
  return 0;
}

// greedy argmax over (rank-local) logits; under tensor parallelism the (max, index) pairs are exchanged
static int greedy_pick(omchat_ctx* ctx, const float* lg, int b, int32_t* next_tokens, hipStream_t s, bool advance = false) {
  const omchat_config& c = ctx->c;
  TRY(launch_argmax(lg, c.t_vocab, b, c.t_vocab, next_tokens, ctx->arg_scratch, s, advance ? ctx->d_pos : nullptr, advance ? ctx->d_len : nullptr));
  if (ctx->tp_size > 1) {
    const size_t n = (size_t)ctx->tp_size * b * 2;
    OM_HIP(hipMemsetAsync(ctx->tp_table, 0, n * 4, s));
    hipLaunchKernelGGL(tp_argmax_scatter_kernel, dim3(1), dim3(64 > b ? 64 : b), 0, s, lg, c.t_vocab, next_tokens, b, ctx->tp_rank, c.t_vocab, ctx->tp_table);
    TRY(ctx->allreduce_f32(ctx->tp_table, n, s));
    hipLaunchKernelGGL(tp_argmax_pick_kernel, dim3(1), dim3(64 > b ? 64 : b), 0, s, ctx->tp_table, b, ctx->tp_size, next_tokens);
  }
  return 0;
}

extern "C" int omchat_greedy(omchat_ctx* ctx, const float* logits, int b, int32_t* next_tokens, void* stream) {
  OM_CHECK(ctx && logits && next_tokens && b >= 1 && b <= ctx->c.max_batch, "bad argument");
  return greedy_pick(ctx, logits, b, next_tokens, (hipStream_t)stream);
}

extern "C" int omchat_lm_head(omchat_ctx* ctx, const void* hidden, int n, float* logits, void* stream) {
  OM_CHECK(ctx && hidden && logits, "null argument");
  return lm_head_rows(ctx, hidden, n, logits, (hipStream_t)stream);
}

static int prefill_impl(omchat_ctx* ctx, const void* embeds, int b, int S, const int32_t* lengths, float* logits_last, void* hidden_out,
                        void* stream, bool left) {
  OM_CHECK(ctx && embeds && lengths, "null argument");
  const omchat_config& c = ctx->c;
  OM_CHECK(c.t_layers > 0, "context has no decoder");
  OM_CHECK(b >= 1 && b <= c.max_batch, "batch exceeds max_batch");
  OM_CHECK(S >= 1 && S <= c.max_seq, "sequence exceeds max_seq");
  OM_CHECK((int64_t)b * S <= c.max_prefill_rows, "b * S exceeds max_prefill_rows");
  OM_CHECK(omchat_weights_missing(ctx) == 0, std::string(omchat_last_error()));
  for (int i = 0; i < b; ++i) OM_CHECK(lengths[i] >= 1 && lengths[i] <= S, "lengths must be in [1, S]");
  hipStream_t s = (hipStream_t)stream;
  if (ctx->fp8_prefill && ctx->fp8_stale) TRY(ensure_fp8_weights(ctx));      // weights were reloaded: re-quantise in place
  const int H = c.t_hidden, It = c.t_mlp, rows = b * S, qkvd = ctx->t_qkvdim, qd = ctx->t_qdim;
  const bool lead = ctx->tp_rank == 0;

  std::vector<int> pos(b), len1(b), klen(b), kstart(b);
  // left-padded batch (omchat_arch.py:176-184): row i holds its n_i tokens at [S - n_i, S); the reference drops position_ids
  // (:206-207), so RoPE runs on arange(S) for every row, the padded keys are masked, and every row's last token sits at S - 1
  for (int i = 0; i < b; ++i) {
    ctx->h_len[i] = left ? 0 : lengths[i];            // 0 = no decode after a left-padded prefill (see omchat_decode_step)
    pos[i] = lengths[i]; len1[i] = lengths[i] + 1;
    klen[i] = left ? S : lengths[i]; kstart[i] = left ? S - lengths[i] : 0;
  }
  ctx->left_padded = left;
  ctx->pre_S = S; ctx->pre_b = b; ctx->masked_steps = 0; ctx->dec_mode = 0; ctx->mask_on_device = false;
  // d_len holds the valid key range end during prefill; switched to (len + 1, pos = len) for the decode steps at the end
  OM_HIP(hipMemcpyAsync(ctx->d_len, klen.data(), (size_t)b * 4, hipMemcpyHostToDevice, s));
  if (left) OM_HIP(hipMemcpyAsync(ctx->d_start, kstart.data(), (size_t)b * 4, hipMemcpyHostToDevice, s));

  void* x = ctx->tw_x;
  void* y = ctx->tw_x2;
  OM_HIP(hipMemcpyAsync(x, embeds, (size_t)rows * H * 2, hipMemcpyDeviceToDevice, s));
  // sequence-parallel norms (gemm_sp): not with the fp8 prefill (its norm writes e4m3 + scales) nor with fp32 partial sums (tuning key 29)
  const bool sp = g_tp_sp && ctx->tp_size > 1 && !g_tp_f32 && !(ctx->fp8_prefill && !ctx->dl8.empty() && !ctx->fp8_stale);
  for (int i = 0; i < c.t_layers; ++i) {
    auto& L = ctx->dl[i];
    char* kc = (char*)ctx->kcache + (size_t)i * ctx->cache_layer_stride() * 2;
    char* vc = (char*)ctx->vcache + (size_t)i * ctx->cache_layer_stride() * 2;
    // Qwen2DecoderLayer.forward (modeling_qwen2.py:269-298)
    const bool f8p = ctx->fp8_prefill && !ctx->dl8.empty() && !ctx->fp8_stale;
    // fp8 x fp8 MFMA for the two column-parallel GEMMs (BASELINE configs[4]): the RMSNorm writes e4m3 + one scale per token, the
    // weights are the e4m3 replica with one scale per output row; o_proj / down_proj keep the 16-bit operands
    auto gemm_f8 = [&](const void* W8, const float* sw, void* Cb, int ldc, int N, const void* bias, int epi) -> int {
      GemmArgs g{ctx->tw_q8, H, W8, H, Cb, ldc, rows, N, H, bias, nullptr, nullptr, 0, epi, 0, nullptr, 0, -1, 1, ctx->tw_q8s, sw};
      return launch_gemm(ctx->dt, g, s);
    };
    if (f8p) {
      TRY(launch_rmsnorm_q8(ctx->dt, x, H, L.ln1, ctx->tw_q8, H, ctx->tw_q8s, rows, H, c.t_eps, s));
      TRY(gemm_f8(ctx->dl8[i].wqkv, ctx->dl8[i].sqkv, ctx->tw_qkv, qkvd, qkvd, L.bqkv, EPI_NONE));
    } else {
      if (!sp || i == 0) TRY(launch_rmsnorm(ctx->dt, x, H, L.ln1, ctx->tw_xn, H, rows, H, c.t_eps, s));      // (sequence-parallel: down_proj of layer i - 1 left it)
      if (sp) TRY(gemm_after_sp(ctx, ctx->tw_xn, H, L.wqkv, H, ctx->tw_qkv, qkvd, rows, qkvd, H, L.bqkv, EPI_NONE, s));
      else TRY(gemm(ctx, ctx->tw_xn, H, L.wqkv, H, ctx->tw_qkv, qkvd, rows, qkvd, H, L.bqkv, nullptr, nullptr, 0, EPI_NONE, s));
    }
    RopeArgs r{ctx->tw_qkv, qkvd, rows, S, c.t_heads, c.t_kv_heads, nullptr, 0, ctx->rope, c.max_seq, kc, vc, ctx->cache_sb(), ctx->cache_sh()};
    TRY(launch_rope_kv(ctx->dt, r, s));
    AttnArgs a{};
    a.Q = ctx->tw_qkv; a.q_sb = (int64_t)S * qkvd; a.q_sh = 128; a.q_sr = qkvd;
    a.K = kc; a.k_sb = ctx->cache_sb(); a.k_sh = ctx->cache_sh(); a.k_sr = 128;
    a.V = vc; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
    a.O = ctx->tw_ao; a.o_sb = (int64_t)S * qd; a.o_sh = 128; a.o_sr = qd;
    a.batch = b; a.q_heads = c.t_heads; a.kv_heads = c.t_kv_heads; a.Sq = S; a.Skv = S; a.kv_len = ctx->d_len; a.causal = 1; a.q_pos0 = 0;
    a.kv_start = left ? ctx->d_start : nullptr;
    a.scale = 0.08838834764831845f;
    TRY(launch_attn_prefill(ctx->dt, a, s));
    // left-padded batch: a padded query row sees no key at all; the reference's eager attention (the CPU path) then attends EVERY key of
    // the sequence with weight 1 / S (all scores are finfo.min -> softmax uniform, future keys included), and these rows' K / V in the
    // next layers are what a masked decode step exposes (omchat_arch.py:61-70).  The flash kernel leaves exactly 0 there; fill them.
    if (left && ctx->tp_size == 1) TRY(launch_attn_uniform_rows(ctx->dt, a, s));
    if (ctx->tp_size == 1) {
      TRY(gemm(ctx, ctx->tw_ao, qd, L.wo, qd, x, H, rows, H, qd, nullptr, nullptr, x, H, EPI_RESID, s));
    } else if (sp) {
      TRY(gemm_sp(ctx, ctx->tw_ao, qd, L.wo, qd, y, x, H, rows, qd, nullptr, nullptr, EPI_RESID, L.ln2, nullptr, ctx->tw_xn, c.t_eps, s));
    } else {
      TRY(gemm_allreduce(ctx, ctx->tw_ao, qd, L.wo, qd, y, H, rows, qd, nullptr, nullptr, x, EPI_RESID, s));
      std::swap(x, y);
    }
    if (f8p) TRY(launch_rmsnorm_q8(ctx->dt, x, H, L.ln2, ctx->tw_q8, H, ctx->tw_q8s, rows, H, c.t_eps, s));
    else if (!sp) TRY(launch_rmsnorm(ctx->dt, x, H, L.ln2, ctx->tw_xn, H, rows, H, c.t_eps, s));
    ctx->prof_mark(OMCHAT_PROF_PREFILL_GATEUP, s);
    if (f8p) TRY(gemm_f8(ctx->dl8[i].wgu, ctx->dl8[i].sgu, ctx->tw_act, It, 2 * It, nullptr, EPI_SWIGLU));
    else if (sp) TRY(gemm_after_sp(ctx, ctx->tw_xn, H, L.wgu, H, ctx->tw_act, It, rows, 2 * It, H, nullptr, EPI_SWIGLU, s));
    else TRY(gemm(ctx, ctx->tw_xn, H, L.wgu, H, ctx->tw_act, It, rows, 2 * It, H, nullptr, nullptr, nullptr, 0, EPI_SWIGLU, s));
    ctx->prof_mark(OMCHAT_PROF_PREFILL_GATEUP, s);
    if (ctx->tp_size == 1) {
      TRY(gemm(ctx, ctx->tw_act, It, L.wd, It, x, H, rows, H, It, nullptr, nullptr, x, H, EPI_RESID, s));
    } else if (sp) {
      const bool more = i + 1 < c.t_layers;
      TRY(gemm_sp(ctx, ctx->tw_act, It, L.wd, It, y, x, H, rows, It, nullptr, nullptr, EPI_RESID, more ? ctx->dl[i + 1].ln1 : nullptr, nullptr, ctx->tw_xn, c.t_eps, s));
      if (!more) TRY(sp_gather_x(ctx, x, H, rows, s));
    } else {
      TRY(gemm_allreduce(ctx, ctx->tw_act, It, L.wd, It, y, H, rows, It, nullptr, nullptr, x, EPI_RESID, s));
      std::swap(x, y);
    }
  }
  TRY(sp_drain(ctx, s));
  if (hidden_out) TRY(launch_rmsnorm(ctx->dt, x, H, ctx->t_norm, hidden_out, H, rows, H, c.t_eps, s));
  if (logits_last) {
    // only the last valid position feeds generation (Qwen2ForCausalLM.forward :462-465 projects all; same values)
    hipLaunchKernelGGL(last_row_index_kernel, dim3(1), dim3(64 > b ? 64 : b), 0, s, ctx->d_len, S, b, ctx->d_idx);
    TRY(launch_gather_rows(ctx->dt, ctx->d_idx, x, nullptr, ctx->tw_last, b, H, s));
    TRY(launch_rmsnorm(ctx->dt, ctx->tw_last, H, ctx->t_norm, ctx->tw_last, H, b, H, c.t_eps, s));
    TRY(lm_head_rows(ctx, ctx->tw_last, b, logits_last, s));
  }
  ctx->kv8_valid = false;
  if (ctx->fp8_kv) {      // fp8 KV cache for the decode steps: quantise what this prefill wrote -- ALL S slots of every row: the masked decode
                          // of a padded batch exposes padded slots too (omchat_arch.py:61-70), and the per-sequence step overwrites them as it appends
    for (int i = 0; i < c.t_layers; ++i) {
      const size_t off = (size_t)i * ctx->cache_layer_stride(), so = (size_t)i * ctx->scale_layer_stride();
      TRY(launch_kv_quant(ctx->dt, (char*)ctx->kcache + off * 2, (char*)ctx->vcache + off * 2, (char*)ctx->k8cache + off, (char*)ctx->v8cache + off,
                          ctx->ks8 + so, ctx->vs8 + so, b, c.t_kv_heads, ctx->cache_sb(), ctx->cache_sh(), (int64_t)c.t_kv_heads * c.max_seq, c.max_seq,
                          nullptr, 0, nullptr, S, s));
    }
    ctx->kv8_valid = true;
  }
  OM_HIP(hipMemcpyAsync(ctx->d_pos, pos.data(), (size_t)b * 4, hipMemcpyHostToDevice, s));
  OM_HIP(hipMemcpyAsync(ctx->d_len, len1.data(), (size_t)b * 4, hipMemcpyHostToDevice, s));
  OM_HIP(hipStreamSynchronize(s));     // pos/len1 are stack vectors
  return 0;
}

extern "C" int omchat_prefill(omchat_ctx* ctx, const void* embeds, int b, int S, const int32_t* lengths, float* logits_last, void* hidden_out,
                              void* stream) {
  return prefill_impl(ctx, embeds, b, S, lengths, logits_last, hidden_out, stream, false);
}

extern "C" int omchat_prefill_left(omchat_ctx* ctx, const void* embeds, int b, int S, const int32_t* lengths, float* logits_last,
                                   void* hidden_out, void* stream) {
  return prefill_impl(ctx, embeds, b, S, lengths, logits_last, hidden_out, stream, true);
}

// (re)build the packed weight replica of the decode-streamed decoder weights; synchronous, never inside a graph capture.
// The replica is optional (+14 GB at OmChat-13B): when it does not fit, the context remembers that (pk_unavailable) and batched decode
// keeps streaming the row-major weights (w_packed = 0), which every kernel still supports.
static int ensure_packed(omchat_ctx* ctx) {
  if (!g_pack_replica || ctx->pk_ready || ctx->pk_unavailable) return 0;
  const omchat_config& c = ctx->c;
  const int H = c.t_hidden, It = c.t_mlp, qkvd = ctx->t_qkvdim, qd = ctx->t_qdim;
  if (qkvd % 16 || H % 16 || c.t_vocab % 16) return 0;          // odd geometry: stay on the row-major weights
  if (ctx->dlp.empty()) {
    // allocate into locals and commit only when every buffer exists: a half-built table must never reach launch_pack_w
    std::vector<omchat_ctx::DecLayerP> fresh(c.t_layers);
    std::vector<void*> got;
    void* lmP = nullptr;
    auto grab = [&](void** p, size_t n) -> bool {
      if (hipMalloc(p, n) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
      got.push_back(*p);
      return true;
    };
    bool ok = true;
    for (auto& P : fresh) {
      ok = ok && grab(&P.wqkv, (size_t)qkvd * H * 2) && grab(&P.wo, (size_t)H * qd * 2) && grab(&P.wgu, (size_t)2 * It * H * 2) &&
           grab(&P.wd, (size_t)H * It * 2);
      if (!ok) break;
    }
    ok = ok && grab(&lmP, (size_t)c.t_vocab * H * 2);
    if (!ok) {
      for (void* p : got) (void)hipFree(p);
      ctx->pk_unavailable = true;
      return 0;
    }
    for (void* p : got) ctx->allocs.push_back(p);
    ctx->bytes += (size_t)c.t_layers * ((size_t)qkvd * H + (size_t)H * qd + (size_t)2 * It * H + (size_t)H * It) * 2 + (size_t)c.t_vocab * H * 2;
    ctx->dlp.swap(fresh);
    ctx->t_lmP = lmP;
  }
  for (int i = 0; i < c.t_layers; ++i) {
    auto& L = ctx->dl[i]; auto& P = ctx->dlp[i];
    TRY(launch_pack_w(ctx->dt, L.wqkv, H, qkvd, H, P.wqkv, nullptr));
    TRY(launch_pack_w(ctx->dt, L.wo, qd, H, qd, P.wo, nullptr));
    TRY(launch_pack_w(ctx->dt, L.wgu, H, 2 * It, H, P.wgu, nullptr));
    TRY(launch_pack_w(ctx->dt, L.wd, It, H, It, P.wd, nullptr));
  }
  TRY(launch_pack_w(ctx->dt, ctx->t_lm, H, c.t_vocab, H, ctx->t_lmP, nullptr));
  OM_HIP(hipDeviceSynchronize());
  ctx->pk_ready = true;
  return 0;
}

// One decode step on stream s.  Lmax = upper bound of the key count (sizes the split-KV grid; the kernels read the true
// lengths from d_len).  Every argument is a context pointer or a step-invariant scalar when called for graph capture.
// exact_len: every sequence holds exactly Lmax keys after this step (eager launches only): the attention launches then take the length
// as a kernel argument instead of loading d_len first -- one dependent memory round trip less in two latency-bound launches per layer.
static int decode_body(omchat_ctx* ctx, const int32_t* tokens, int b, int Lmax, float* logits, int32_t* next_tokens, hipStream_t s, bool allow_prof,
                       bool exact_len = false, bool masked = false) {
  const omchat_config& c = ctx->c;
  const int H = c.t_hidden, It = c.t_mlp, qkvd = ctx->t_qkvdim, qd = ctx->t_qdim;
  const bool lead = ctx->tp_rank == 0;
  void* x = ctx->tw_x;
  void* y = ctx->tw_x2;
  TRY(launch_gather_rows(ctx->dt, tokens, ctx->t_embed, nullptr, x, b, H, s));      // embed_tokens
  // weight-only fp8 replica (omchat_enable_fp8_decode): batch-1 steps stream e4m3 bytes + per-row scales
  const bool f8 = ctx->fp8_decode && b == 1;      // under tensor parallelism every rank streams the e4m3 replica of its own shard (round 3)
  // batched steps (2 <= b <= 32): the activations that feed a GEMV (tw_xn, tw_ao, tw_act) are produced in the packed x layout
  // (common.h) by their producers, and the weights come from the packed replica when it exists (ensure_packed)
  const int pk = (b > 1 && b <= 32 && H % 64 == 0 && qd % 64 == 0 && It % 64 == 0) ? (b > 16 ? 2 : 1) : 0;
  const bool wpk = pk && ctx->pk_ready;
  static const omchat_ctx::DecLayerP noneP{};
  auto gemv_args = [&](const void* X, int ldx, const void* W, int K, void* Y, int ldy, int R, int N, const void* bias, const void* resid, int epi,
                       int ks, const void* W8, const float* sc, const void* WP, bool ypk) {
    GemvArgs g{X, ldx, W, K, Y, ldy, R, N, K, bias, resid, H, epi, 0, ks};
    if (f8 && W8) { g.W = W8; g.w_scale = sc; }
    if (pk) { g.x_packed = 1; g.y_packed = ypk && epi == EPI_SWIGLU; if (wpk && WP) { g.W = WP; g.w_packed = 1; } }
    return g;
  };
  auto gemv = [&](const void* X, int ldx, const void* W, int K, void* Y, int ldy, int N, const void* bias, const void* resid, int epi,
                  const void* W8 = nullptr, const float* sc = nullptr, const void* WP = nullptr, bool ypk = false) -> int {
    for (int r0 = 0; r0 < b; r0 += 32) {
      const int R = std::min(32, b - r0);
      const GemvArgs g = gemv_args((const char*)X + (size_t)r0 * ldx * 2, ldx, W, K, (char*)Y + (size_t)r0 * ldy * 2, ldy, R, N, bias,
                                   resid ? (const char*)resid + (size_t)r0 * H * 2 : nullptr, epi, 0, W8, sc, WP, ypk);
      TRY(launch_gemv(ctx->dt, g, s));
    }
    return 0;
  };
  // split-K over workgroups: fp32 slices [ks][b][H], summed by the fused residual + RMSNorm kernel
  auto gemv_partial = [&](const void* X, int ldx, const void* W, int K, int ks, const void* W8 = nullptr, const float* sc = nullptr,
                          const void* WP = nullptr) -> int {
    OM_CHECK(b <= 32, "split-K decode path handles b <= 32");
    const GemvArgs g = gemv_args(X, ldx, W, K, ctx->tw_part, H, b, H, nullptr, nullptr, EPI_PARTIAL, ks, W8, sc, WP, false);
    TRY(launch_gemv(ctx->dt, g, s));
    // tensor parallelism: the slices hold this rank's partial sums; reduce_resid_rmsnorm sums them over the ranks in fp32 (<= 3 x 3584
    // floats at batch 1, latency-bound like any small message) together with the residual + RMSNorm, identically on every rank
    return 0;
  };
  const bool fused = b <= 32;
  // K slices: batch 1 (whole-row streaming form) wants <= 8 chunks of 512 per slice and >= ~2500 waves in the grid;
  // the MFMA form (b > 1) wants ~2-3 workgroups per CU
  // (tools/experiments/tune_rows.hip: down_proj 18944 -> 8 slices 23.3 us vs 5 slices 27.6 us; o_proj is latency-bound, 1-3 slices alike)
  auto ks_rows = [&](int K) { const int nch = cdiv(K, 512); return nch >= 16 ? std::min(DEC_KS_MAX, nch) : std::max(1, std::min(3, nch)); };
  // batched o_proj: K = 3584 cuts into two exact slices for the x-stationary form (40 + 16 chunks, gemv_xs_split_kernel) when the packed
  // replica is in use; otherwise ~2 workgroups per CU for the MFMA form
  // shard widths of a tensor-parallel rank (round 5): a K of <= 64 chunks is cut into slices of >= 16 chunks only (o_proj K = 512: one
  // slice, down_proj K = 2368: two) -- eight slices of 4-5 chunks left seven of a workgroup's eight waves without a chunk, and every slice
  // is 4 b H bytes that the residual + RMSNorm launch (under TP: the peer exchange) has to read
  auto ks_short = [&](int K) { return std::max(1, std::min(DEC_KS_MAX, (K / 64) / 16)); };
  const bool shard_ks = (gemv_get_shard_shapes() & 2) != 0;
  const int ks_o = b == 1 ? ks_rows(qd) : (wpk && qd == 3584 ? 2 : (shard_ks && qd / 64 <= 64) ? ks_short(qd) :
                                           std::max(1, std::min(DEC_KS_MAX, std::min(qd / 64, cdiv(512, cdiv(H, 16))))));
  const int ks_d = b == 1 ? ks_rows(It) : (shard_ks && It / 64 <= 64) ? ks_short(It) :
                                          std::max(1, std::min(DEC_KS_MAX, std::min(It / 64, 3 * cdiv(512, cdiv(H, 16)))));
  // batch 1, one GPU (round 3): the two residual + RMSNorm launches of a layer disappear.  o_proj and down_proj run without split-K and
  // write x + attn / x + mlp themselves (EPI_RESID, in place; down_proj's K = 18944 through gemv_rows_longk_kernel), and each RMSNorm runs
  // inside the projection that consumes it (gemv_rows_norm_kernel: qkv, gate|up, lm_head): six dependent launches per layer instead of
  // eight.  n2 = the post-attention norm (tuning key 14 bit 0), n1 = the input norm of the next layer / the final norm (bit 1).
  // (only while the whole-row GEMV form is in use: with tuning key 1 -- force the MFMA form -- the norm has no registers to live in and
  // the step keeps its residual + RMSNorm launches)
  // measurement only (experiments build, bench.py --shard-of N --tuning 35=1): a rank context whose exchanges are no-ops takes the ONE-GPU launch
  // structures on its shard widths -- what removing launches could buy a tensor-parallel rank if its two exchanges per layer were free
  const bool tp1_like = ctx->tp_size == 1 || (OMCHAT_EXPERIMENTS && g_shard_as_tp1 && ctx->hook == omchat_allreduce_noop);
  const bool n2 = (g_norm_in_gemv & 1) && fused && b == 1 && tp1_like && qd <= 4096 && H <= 4096 && !gemv_get_force_mfma();
  const bool n1 = n2 && (g_norm_in_gemv & 2) && It <= 32768 && It % 8 == 0;
  // batched steps on one GPU (round 5, key 14 bit 2; -DOMCHAT_EXPERIMENTS=1 builds only: measured slower, gemv.hip): o_proj un-split in the x-stationary form writes x + attn itself (row-major in place AND the
  // packed raw copy), and the post-attention RMSNorm runs in the registers of the gate|up GEMV: seven launches per layer instead of eight
  const bool nb2 = OMCHAT_EXPERIMENTS && (g_norm_in_gemv & 4) && fused && wpk && ctx->tp_size == 1 && !f8 && (qd >> 6) == 56 && (H >> 6) == 56 && qd % 64 == 0 &&
                   H / 16 <= device_cus() && (2 * It) / 32 >= 4 * device_cus();
  if (fused && !n1) TRY(launch_rmsnorm(ctx->dt, x, H, ctx->dl[0].ln1, ctx->tw_xn, H, b, H, c.t_eps, s, pk));
#if OMCHAT_EXPERIMENTS
  if ((g_ao_oproj & 16) && ctx->dbg_stamps) OM_HIP(hipMemsetAsync(ctx->dbg_stamps, 0, (size_t)c.t_layers * 256, s));      // measurement: the stamps of THIS step only
#endif
  for (int i = 0; i < c.t_layers; ++i) {
    auto& L = ctx->dl[i];
    static const omchat_ctx::DecLayer8 none8{};
    const omchat_ctx::DecLayer8& Q = f8 ? ctx->dl8[i] : none8;
    const omchat_ctx::DecLayerP& P = wpk ? ctx->dlp[i] : noneP;
    char* kc = (char*)ctx->kcache + (size_t)i * ctx->cache_layer_stride() * 2;
    char* vc = (char*)ctx->vcache + (size_t)i * ctx->cache_layer_stride() * 2;
    if (!fused) TRY(launch_rmsnorm(ctx->dt, x, H, L.ln1, ctx->tw_xn, H, b, H, c.t_eps, s));
    // batch 1, one GPU (round 4): the whole layer as ONE launch with in-launch hand-offs (decode_layer.hip; the same bits as the six
    // launches below).  Eager steps only: the launch is tagged with a per-launch counter, which a captured graph would freeze.
    if (g_decode_layer && n1 && exact_len && !masked && !f8 && !(ctx->fp8_kv && ctx->kv8_valid) && ctx->dl_ws) {      // (the layer kernel has no key mask)
      DecodeLayerArgs d{L.ln1, L.ln2, L.wqkv, L.bqkv, L.wo, L.wgu, L.wd, kc, vc, ctx->cache_sh(), x, H, qd, ctx->t_kvdim, It, c.t_heads, c.t_kv_heads, Lmax,
                        ctx->rope, c.max_seq, c.t_eps, 0.08838834764831845f, ctx->dl_ws, ++ctx->fd_epoch, ctx->fd_err, 2000};
      if (decode_layer_ok(d)) {
        const bool mark_l = allow_prof && i == c.t_layers / 2;      // HIP-event bracket on one layer per token (bench.py roofline)
        if (mark_l) ctx->prof_mark(OMCHAT_PROF_DECODE_GATEUP, s);
        TRY(launch_decode_layer(ctx->dt, d, s));
        if (mark_l) ctx->prof_mark(OMCHAT_PROF_DECODE_GATEUP, s);
        ++ctx->n_layer_launches;
        continue;
      }
      --ctx->fd_epoch;
    }
    if (n1) {
      GemvArgs g = gemv_args(x, H, L.wqkv, H, ctx->tw_qkv, qkvd, 1, qkvd, L.bqkv, nullptr, EPI_NONE, 0, Q.wqkv, Q.sqkv, nullptr, false);
      g.norm_w = L.ln1; g.norm_eps = c.t_eps;
      if (OMCHAT_EXPERIMENTS && (g_ao_oproj & 16) && ctx->dbg_stamps) g.dbg = ctx->dbg_stamps + (size_t)i * 32;
      TRY(launch_gemv(ctx->dt, g, s));
    } else {
      TRY(gemv(ctx->tw_xn, H, L.wqkv, H, ctx->tw_qkv, qkvd, qkvd, L.bqkv, nullptr, EPI_NONE, Q.wqkv, Q.sqkv, P.wqkv));
    }
    // RoPE + KV append are fused into the attention kernel (q rotated in registers, the split that owns the new
    // position rotates k and appends k / v)
    AttnDecodeArgs a{};
    a.Q = ctx->tw_qkv; a.q_sb = qkvd; a.q_sh = 128;
    a.K = kc; a.k_sb = ctx->cache_sb(); a.k_sh = ctx->cache_sh(); a.k_sr = 128;
    a.V = vc; a.v_sb = a.k_sb; a.v_sh = a.k_sh; a.v_sr = 128;
    a.O = ctx->tw_ao; a.o_sb = qd; a.o_sh = 128;
    a.batch = b; a.q_heads = c.t_heads; a.kv_heads = c.t_kv_heads; a.L = Lmax; a.kv_len = ctx->d_len; a.scale = 0.08838834764831845f;
    a.ws = ctx->tw_attn_ws; a.ws_bytes = ctx->tw_attn_ws_bytes;
    a.rope = ctx->rope; a.rope_max = c.max_seq; a.pos = ctx->d_pos;
    a.k_new = (const char*)ctx->tw_qkv + (size_t)qd * 2; a.v_new = (const char*)ctx->tw_qkv + (size_t)(qd + ctx->t_kvdim) * 2; a.new_sb = qkvd;
    if (exact_len) a.kv_len = nullptr;
    if (masked) { a.key_mask = ctx->d_mask; a.mask_sb = ctx->mask_sb; }      // d_pos holds the given RoPE positions, every row has Lmax keys
    if (ctx->fp8_kv && ctx->kv8_valid) {
      // fp8 KV cache: rotate q / k and append to the 16-bit cache with the prefill's kernel, quantise the new row, then attend over e4m3
      // keys and values (two small launches more per layer than the fused 16-bit path; this mode is for long contexts)
      // (round 3: the new row is quantised inside the RoPE + append launch -- one launch less per layer than rope_kv + kv_quant)
      const size_t off = (size_t)i * ctx->cache_layer_stride(), so = (size_t)i * ctx->scale_layer_stride();
      RopeArgs r{ctx->tw_qkv, qkvd, b, 1, c.t_heads, c.t_kv_heads, ctx->d_pos, 0, ctx->rope, c.max_seq, kc, vc, ctx->cache_sb(), ctx->cache_sh()};
      r.k8 = (char*)ctx->k8cache + off; r.v8 = (char*)ctx->v8cache + off; r.ks = ctx->ks8 + so; r.vs = ctx->vs8 + so;
      r.s_sb = (int64_t)c.t_kv_heads * c.max_seq; r.s_sh = c.max_seq;
      if (masked) r.slot0 = Lmax - 1;                              // padded batch: common cache slot, per-row RoPE positions from d_pos
      else if (exact_len) { r.pos = nullptr; r.pos0 = Lmax - 1; }      // the position by value: one dependent load less in front of the table read
      // round 6: from ~9 k keys on the attention launch walks its tiles and rotates / appends / quantises the new rows itself (the same bytes)
      if (attn_decode_kv8_fuses_rope(b, c.t_kv_heads, Lmax, masked)) { a.k16_w = kc; a.v16_w = vc; a.pos = nullptr; }
      else {
        TRY(launch_rope_kv(ctx->dt, r, s));
        a.rope = nullptr; a.pos = nullptr; a.k_new = nullptr; a.v_new = nullptr;
      }
      a.K = (char*)ctx->k8cache + off; a.V = (char*)ctx->v8cache + off;
      a.k_scale = ctx->ks8 + so; a.v_scale = ctx->vs8 + so; a.scale_sb = (int64_t)c.t_kv_heads * c.max_seq; a.scale_sh = c.max_seq;
    }
    a.o_pack_nb = fused ? pk : 0;
    // batch 1, one GPU (round 4): attention + merge + o_proj (+ residual) as ONE launch with in-launch hand-offs (fused_decode.hip; the same
    // bits as the three launches).  Eager steps only: the launch is tagged with a per-launch counter, which a captured graph would freeze.
    bool ao_oproj = false;
    const bool fuse_ao = g_fuse_attn_oproj && n2 && exact_len && !f8 && a.rope && ctx->fd_ws && attn_oproj_fused_ok(a, H, qd);
    if (fuse_ao) {
      FusedDecodeArgs fa{L.wo, qd, x, H, qd, ctx->fd_ws, ++ctx->fd_epoch, ctx->fd_err, 2000};
      TRY(launch_attn_oproj_fused(ctx->dt, a, fa, s));
      ++ctx->n_fused_launches;
    } else {
#if OMCHAT_EXPERIMENTS
      // tuning key 42 (prototype): o_proj out of order behind the merge, which publishes per-head completion flags (fd_ws doubles as the flag words:
      // key 22 and key 42 are not meant to be on together)
      ao_oproj = (g_ao_oproj & 1) && n2 && !f8 && !masked && ctx->fd_ws && Lmax <= 4096 && H == 3584 && qd == 3584 && c.t_heads <= 64 && !ctx->graph_on;
      if (ao_oproj) { a.done_flags = (unsigned*)ctx->fd_ws; a.done_epoch = ++ctx->fd_epoch; a.done_mode = g_ao_oproj; }
      if ((g_ao_oproj & 16) && n2 && ctx->dbg_stamps) a.done_dbg = ctx->dbg_stamps + (size_t)i * 32;
#endif
      TRY(launch_attn_decode(ctx->dt, a, s));
    }
    // batch 1, one GPU (round 3): o_proj without split-K writes x + attn itself (EPI_RESID, in place) and the post-attention RMSNorm runs
    // in the registers of the gate|up GEMV's waves (gemv.hip: norm_w): seven dependent launches per layer instead of eight
    if (fuse_ao) {
      // x + attn is already in place
    } else if (ao_oproj) {
      GemvArgs g = gemv_args(ctx->tw_ao, qd, L.wo, qd, x, H, 1, H, nullptr, x, EPI_RESID, 0, nullptr, nullptr, nullptr, false);
      if ((g_ao_oproj & 16) && ctx->dbg_stamps) g.dbg = ctx->dbg_stamps + (size_t)i * 32;
      TRY(launch_gemv_wait(ctx->dt, g, (const unsigned*)ctx->fd_ws, ctx->fd_epoch, c.t_heads, ctx->fd_err, g_ao_oproj, s));
    } else if (n2 && OMCHAT_EXPERIMENTS && (g_ao_oproj & 16) && ctx->dbg_stamps && !f8) {
      GemvArgs g = gemv_args(ctx->tw_ao, qd, L.wo, qd, x, H, 1, H, nullptr, x, EPI_RESID, 0, nullptr, nullptr, nullptr, false);
      g.dbg = ctx->dbg_stamps + (size_t)i * 32;
      TRY(launch_gemv(ctx->dt, g, s));
    } else if (n2) {
      TRY(gemv(ctx->tw_ao, qd, L.wo, qd, x, H, H, nullptr, x, EPI_RESID, Q.wo, Q.so));
    } else if (nb2) {
      GemvArgs g = gemv_args(ctx->tw_ao, qd, L.wo, qd, x, H, b, H, nullptr, x, EPI_RESID, 0, nullptr, nullptr, P.wo, false);
      g.y_pack = ctx->tw_xn;
      TRY(launch_gemv(ctx->dt, g, s));
    } else if (fused) {
      TRY(gemv_partial(ctx->tw_ao, qd, L.wo, qd, ks_o, Q.wo, Q.so, P.wo));
      TRY(ctx->reduce_resid_rmsnorm(x, H, ctx->tw_part, ks_o, L.ln2, ctx->tw_xn, H, b, H, c.t_eps, pk, s));
    } else if (ctx->tp_size == 1) {
      TRY(gemv(ctx->tw_ao, qd, L.wo, qd, x, H, H, nullptr, x, EPI_RESID));
    } else {
      TRY(gemv(ctx->tw_ao, qd, L.wo, qd, y, H, H, nullptr, lead ? x : nullptr, lead ? EPI_RESID : EPI_NONE));
      TRY(ctx->allreduce(y, (size_t)b * H, s));
      std::swap(x, y);
    }
    if (!fused) TRY(launch_rmsnorm(ctx->dt, x, H, L.ln2, ctx->tw_xn, H, b, H, c.t_eps, s));
    // HIP-event bracket on ONE layer per token only: each event record costs ~1-2 us of launch-stream time, and 56 of them
    // per token would themselves slow the measured decode by a few per cent
    const bool mark = allow_prof && i == c.t_layers / 2;
    if (mark) ctx->prof_mark(OMCHAT_PROF_DECODE_GATEUP, s);
    if (n2) {
      GemvArgs g = gemv_args(x, H, L.wgu, H, ctx->tw_act, It, 1, 2 * It, nullptr, nullptr, EPI_SWIGLU, 0, Q.wgu, Q.sgu, nullptr, false);
      g.norm_w = L.ln2; g.norm_eps = c.t_eps;
      if (ctx->dyn_ctr && !ctx->graph_on) g.dyn_ctr = ctx->dyn_ctr + (size_t)i * 65 * 64;
      if (OMCHAT_EXPERIMENTS && (g_ao_oproj & 16) && ctx->dbg_stamps) g.dbg = ctx->dbg_stamps + (size_t)i * 32;
      TRY(launch_gemv(ctx->dt, g, s));
    } else if (nb2) {
      GemvArgs g = gemv_args(ctx->tw_xn, H, L.wgu, H, ctx->tw_act, It, b, 2 * It, nullptr, nullptr, EPI_SWIGLU, 0, nullptr, nullptr, P.wgu, true);
      g.norm_w = L.ln2; g.norm_eps = c.t_eps;
      TRY(launch_gemv(ctx->dt, g, s));
    } else {
      TRY(gemv(ctx->tw_xn, H, L.wgu, H, ctx->tw_act, It, 2 * It, nullptr, nullptr, EPI_SWIGLU, Q.wgu, Q.sgu, P.wgu, true));
    }
    if (mark) ctx->prof_mark(OMCHAT_PROF_DECODE_GATEUP, s);
    if (n1 && OMCHAT_EXPERIMENTS && (g_ao_oproj & 16) && ctx->dbg_stamps && !f8) {
      GemvArgs g = gemv_args(ctx->tw_act, It, L.wd, It, x, H, 1, H, nullptr, x, EPI_RESID, 0, nullptr, nullptr, nullptr, false);
      g.dbg = ctx->dbg_stamps + (size_t)i * 32;
      TRY(launch_gemv(ctx->dt, g, s));
    } else if (n1) {
      TRY(gemv(ctx->tw_act, It, L.wd, It, x, H, H, nullptr, x, EPI_RESID, Q.wd, Q.sd));
    } else if (fused) {
      TRY(gemv_partial(ctx->tw_act, It, L.wd, It, ks_d, Q.wd, Q.sd, P.wd));
      const void* nw = i + 1 < c.t_layers ? ctx->dl[i + 1].ln1 : ctx->t_norm;      // next layer's input norm, or the final norm
      TRY(ctx->reduce_resid_rmsnorm(x, H, ctx->tw_part, ks_d, nw, ctx->tw_xn, H, b, H, c.t_eps, pk, s));
    } else if (ctx->tp_size == 1) {
      TRY(gemv(ctx->tw_act, It, L.wd, It, x, H, H, nullptr, x, EPI_RESID));
    } else {
      TRY(gemv(ctx->tw_act, It, L.wd, It, y, H, H, nullptr, lead ? x : nullptr, lead ? EPI_RESID : EPI_NONE));
      TRY(ctx->allreduce(y, (size_t)b * H, s));
      std::swap(x, y);
    }
  }
  if (!fused)   TRY(launch_rmsnorm(ctx->dt, x, H, ctx->t_norm, ctx->tw_xn, H, b, H, c.t_eps, s));
  float* lg = logits ? logits : ctx->tw_logits;
  if (n1) {
    GemvArgs g{x, H, ctx->t_lm, H, lg, c.t_vocab, 1, c.t_vocab, H, nullptr, nullptr, 0, EPI_NONE, 1};
    if (f8) { g.W = ctx->t_lm8; g.w_scale = ctx->t_lm8_s; }
    g.norm_w = ctx->t_norm; g.norm_eps = c.t_eps;
    TRY(launch_gemv(ctx->dt, g, s));
  } else {
    TRY(lm_head_rows(ctx, ctx->tw_xn, b, lg, s, f8, fused && pk));
  }
  // the position bookkeeping (pos += 1, len += 1) rides in the argmax's second stage when the step picks a token (one launch less per token)
  if (next_tokens) TRY(greedy_pick(ctx, lg, b, next_tokens, s, true));
  else hipLaunchKernelGGL(advance_lens_kernel, dim3(1), dim3(64 > b ? 64 : b), 0, s, ctx->d_pos, ctx->d_len, b);
  OM_LAUNCH_CHECK();
  return 0;
}

static void destroy_graph(omchat_ctx::DecodeGraph& g) {
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g = omchat_ctx::DecodeGraph{};
}

extern "C" int omchat_enable_decode_graph(omchat_ctx* ctx, int on) {
  OM_CHECK(ctx, "null ctx");
  OM_CHECK(ctx->c.t_layers > 0, "context has no decoder");
  if (on && !ctx->graph_stream) {
    OM_HIP(hipStreamCreateWithFlags(&ctx->graph_stream, hipStreamNonBlocking));
    OM_HIP(hipEventCreateWithFlags(&ctx->graph_ev_in, hipEventDisableTiming));
    OM_HIP(hipEventCreateWithFlags(&ctx->graph_ev_out, hipEventDisableTiming));
    TRY(ctx->alloc((void**)&ctx->d_tok_in, (size_t)ctx->c.max_batch * 4));
    TRY(ctx->alloc((void**)&ctx->d_tok_out, (size_t)ctx->c.max_batch * 4));
  }
  ctx->graph_on = on != 0;
  return 0;
}

extern "C" int omchat_decode_graph_stats(omchat_ctx* ctx, long* steps, long* replays, long* captures) {
  OM_CHECK(ctx, "null ctx");
  if (steps) *steps = ctx->graph_steps;
  if (replays) *replays = ctx->graph_replays;
  if (captures) *captures = ctx->graph_captures;
  return 0;
}

extern "C" int omchat_decode_step(omchat_ctx* ctx, const int32_t* tokens, int b, float* logits, int32_t* next_tokens, void* stream) {
  OM_CHECK(ctx && tokens, "null argument");
  const omchat_config& c = ctx->c;
  OM_CHECK(c.t_layers > 0, "context has no decoder");
  OM_CHECK(b >= 1 && b <= c.max_batch, "batch exceeds max_batch");
  int Lmax = 0;
  OM_CHECK(!ctx->left_padded, "per-sequence decode after a left-padded prefill: use omchat_decode_step_masked (the reference positions such a "
                              "batch with sum(mask) - 1 and a per-row key mask, omchat_arch.py:61-70; DESIGN.md section 7)");
  OM_CHECK(ctx->dec_mode != 2, "omchat_decode_step after omchat_decode_step_masked on the same prefill: the two place the cache rows differently");
  bool same_len = true;
  for (int i = 0; i < b; ++i) {
    OM_CHECK(ctx->h_len[i] >= 1, "decode before prefill");
    Lmax = std::max(Lmax, ctx->h_len[i] + 1);
    same_len = same_len && ctx->h_len[i] == ctx->h_len[0];
  }
  OM_CHECK(Lmax <= c.max_seq, "KV cache full (max_seq)");
  hipStream_t s = (hipStream_t)stream;
  if (b > 1 && b <= 32) TRY(ensure_packed(ctx));      // first batched step (or after a weight reload): build the packed replica
  // a weight reload (omchat_load_tensor) leaves the e4m3 replica stale: re-quantise IN PLACE before streaming it (same device
  // pointers, so a captured decode graph stays valid and replays the fresh bytes)
  if (ctx->fp8_decode && ctx->fp8_stale) TRY(ensure_fp8_weights(ctx));
  ctx->graph_steps++;
  // graph replay needs replay-invariant arguments: single-GPU fused path only; with profiling on, every 8th step runs eagerly
  // so that the HIP-event brackets of the dominant kernel are still recorded inside the timed region
  const bool graph = ctx->graph_on && ctx->tp_size == 1 && b <= 32 && !(ctx->prof_on && ctx->graph_steps % 8 == 0);
  if (!graph) {
    TRY(decode_body(ctx, tokens, b, Lmax, logits, next_tokens, s, true, same_len));
  } else {
    const bool f8 = ctx->fp8_decode && b == 1;
    omchat_ctx::DecodeGraph& g = ctx->graphs[b * 4 + (f8 ? 2 : 0) + ((ctx->fp8_kv && ctx->kv8_valid) ? 1 : 0)];
    hipStream_t gs = ctx->graph_stream;
    OM_HIP(hipEventRecord(ctx->graph_ev_in, s));
    OM_HIP(hipStreamWaitEvent(gs, ctx->graph_ev_in, 0));
    OM_HIP(hipMemcpyAsync(ctx->d_tok_in, tokens, (size_t)b * 4, hipMemcpyDeviceToDevice, gs));
    if (!g.exec || Lmax > g.cap_len) {
      destroy_graph(g);
      const int cap = std::min(c.max_seq, (Lmax + 1024 + 63) / 64 * 64);
      OM_HIP(hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal));
      const int rc = decode_body(ctx, ctx->d_tok_in, b, cap, ctx->tw_logits, ctx->d_tok_out, gs, false);
      hipGraph_t graph_h = nullptr;
      const hipError_t e = hipStreamEndCapture(gs, &graph_h);
      if (rc) { if (graph_h) (void)hipGraphDestroy(graph_h); return rc; }
      OM_HIP(e);
      g.graph = graph_h;
      OM_HIP(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
      g.cap_len = cap;
      ctx->graph_captures++;
    }
    OM_HIP(hipGraphLaunch(g.exec, gs));
    ctx->graph_replays++;
    if (next_tokens) OM_HIP(hipMemcpyAsync(next_tokens, ctx->d_tok_out, (size_t)b * 4, hipMemcpyDeviceToDevice, gs));
    if (logits) OM_HIP(hipMemcpyAsync(logits, ctx->tw_logits, (size_t)b * c.t_vocab * 4, hipMemcpyDeviceToDevice, gs));
    OM_HIP(hipEventRecord(ctx->graph_ev_out, gs));
    OM_HIP(hipStreamWaitEvent(s, ctx->graph_ev_out, 0));
  }
  for (int i = 0; i < b; ++i) ctx->h_len[i] += 1;
  ctx->dec_mode = 1;
  return 0;
}

// Decode step of a PADDED batch exactly as the reference computes it (omchat_arch.py:61-70 as HF generate drives it with `images` passed
// on every step): the new token of EVERY row is appended at the common cache length (S of the padded prefill + the steps so far), rotated
// to positions[i] = sum(mask_i) - 1, and attends the cache slots whose mask byte is non-zero -- the token-level mask padded with ones,
// which hides real prompt slots and exposes padded ones once images expanded the rows differently.  After a right-padded prefill the
// padded slots hold the K / V of causally un-masked padded query rows (what every backend of the reference computes); after a
// left-padded one those of fully masked rows, which the prefill filled as the reference's eager (CPU) attention does.
static int masked_common_checks(omchat_ctx* ctx, int b) {
  const omchat_config& c = ctx->c;
  OM_CHECK(c.t_layers > 0, "context has no decoder");
  OM_CHECK(ctx->pre_S >= 1 && b == ctx->pre_b, "masked decode: the batch of the last prefill, all rows");
  OM_CHECK(ctx->dec_mode != 1, "omchat_decode_step_masked after omchat_decode_step on the same prefill: the two place the cache rows differently");
  OM_CHECK(ctx->tp_size == 1, "masked decode: one GPU (under tensor parallelism right-padded batches take the per-sequence step)");
  OM_CHECK(ctx->pre_S + ctx->masked_steps + 1 <= c.max_seq, "KV cache full (max_seq)");
  if (!ctx->d_mask) {
    ctx->mask_sb = (int64_t)cdiv(c.max_seq, 64) * 64;
    TRY(ctx->alloc((void**)&ctx->d_mask, (size_t)c.max_batch * ctx->mask_sb));
  }
  // replicas first: a failure here must not leave host-to-device copies of caller / stack memory in flight
  if (b > 1 && b <= 32) TRY(ensure_packed(ctx));
  if (ctx->fp8_decode && ctx->fp8_stale) TRY(ensure_fp8_weights(ctx));
  return 0;
}

extern "C" int omchat_decode_step_masked(omchat_ctx* ctx, const int32_t* tokens, int b, const int32_t* positions, const uint8_t* key_mask,
                                         int mask_ld, float* logits, int32_t* next_tokens, void* stream) {
  OM_CHECK(ctx && tokens && positions && key_mask, "null argument");
  const omchat_config& c = ctx->c;
  TRY(masked_common_checks(ctx, b));
  const int Lc = ctx->pre_S + ctx->masked_steps;          // slots every row holds; the new token goes to slot Lc
  OM_CHECK(mask_ld >= Lc + 1, "key_mask rows must cover the cache and the new token (slots + 1 columns)");
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < b; ++i) {
    OM_CHECK(positions[i] >= 0 && positions[i] < c.max_seq, "position outside the RoPE table");
    OM_CHECK(key_mask[(size_t)i * mask_ld + Lc] != 0, "the new token must see itself");
  }
  ctx->mask_on_device = false;      // the device copy holds THIS step's mask, zero-padded: omchat_decode_step_masked_next needs a new begin
  OM_HIP(hipMemsetAsync(ctx->d_mask, 0, (size_t)b * ctx->mask_sb, s));
  OM_HIP(hipMemcpy2DAsync(ctx->d_mask, (size_t)ctx->mask_sb, key_mask, (size_t)mask_ld, (size_t)Lc + 1, (size_t)b, hipMemcpyHostToDevice, s));
  OM_HIP(hipMemcpyAsync(ctx->d_pos, positions, (size_t)b * 4, hipMemcpyHostToDevice, s));
  const int rc = decode_body(ctx, tokens, b, Lc + 1, logits, next_tokens, s, false, true, true);
  OM_HIP(hipStreamSynchronize(s));     // the caller's mask / positions may be reused as soon as this returns
  if (rc) return rc;
  ctx->masked_steps += 1;
  ctx->dec_mode = 2;
  return 0;
}

// The same step without per-step host data (round 5).  In the loop HF generate drives (single_inference.py:53-62 on a padded batch) the decode
// branch of prepare_inputs_labels_for_multimodal (omchat_arch.py:61-70) pads the token-level mask with ONES up to the cache length, and every
// generated token appends another one: over the cache slots the key mask of step k is [token-level mask of the PROMPT | ones], the same for
// every k, and position_ids = sum(mask) - 1 grows by one per step.  So the mask goes to the device once (begin: `mask_cols` columns from the
// caller, ones behind them) with the first step's positions, and every following step (next) takes its positions from the device, where the
// greedy pick of the previous step advanced them: no host buffer, no stream synchronisation, and the caller may enqueue step k + 1 before it
// has read token k (OmChatQwen2ForCausalLM.generate).  Arbitrary per-step masks keep omchat_decode_step_masked.
extern "C" int omchat_masked_decode_begin(omchat_ctx* ctx, int b, const int32_t* positions, const uint8_t* key_mask, int mask_ld, int mask_cols,
                                          void* stream) {
  OM_CHECK(ctx && positions && key_mask, "null argument");
  const omchat_config& c = ctx->c;
  TRY(masked_common_checks(ctx, b));
  OM_CHECK(mask_cols >= 1 && mask_cols <= mask_ld && mask_cols <= c.max_seq, "mask_cols: 1 .. min(mask_ld, max_seq)");
  // Only the slots that exist at the first step carry caller data: [0, Lc) the cache, Lc the new token's own slot.  Everything behind is a
  // generated token's slot and must read as visible whatever a zero-padded caller buffer holds there (zeros at or beyond Lc hid
  // the new token and every later one without an error); a caller mask that hides slot Lc itself is refused like the host-mask entry does.
  const int Lc = ctx->pre_S + ctx->masked_steps;
  const int cols = mask_cols < Lc + 1 ? mask_cols : Lc + 1;
  for (int i = 0; i < b; ++i) {
    OM_CHECK(positions[i] >= 0 && positions[i] < c.max_seq, "position outside the RoPE table");
    OM_CHECK(mask_cols <= Lc || key_mask[(size_t)i * mask_ld + Lc] != 0, "the new token must see itself (key_mask column pre_S + steps)");
  }
  hipStream_t s = (hipStream_t)stream;
  OM_HIP(hipMemsetAsync(ctx->d_mask, 1, (size_t)b * ctx->mask_sb, s));
  OM_HIP(hipMemcpy2DAsync(ctx->d_mask, (size_t)ctx->mask_sb, key_mask, (size_t)mask_ld, (size_t)cols, (size_t)b, hipMemcpyHostToDevice, s));
  OM_HIP(hipMemcpyAsync(ctx->d_pos, positions, (size_t)b * 4, hipMemcpyHostToDevice, s));
  OM_HIP(hipStreamSynchronize(s));     // once per generation: the caller's buffers are free again
  ctx->mask_on_device = true;
  return 0;
}

extern "C" int omchat_decode_step_masked_next(omchat_ctx* ctx, const int32_t* tokens, int b, float* logits, int32_t* next_tokens, void* stream) {
  OM_CHECK(ctx && tokens, "null argument");
  TRY(masked_common_checks(ctx, b));
  OM_CHECK(ctx->mask_on_device, "omchat_decode_step_masked_next: call omchat_masked_decode_begin after the prefill (and after any omchat_decode_step_masked)");
  const int Lc = ctx->pre_S + ctx->masked_steps;
  const int rc = decode_body(ctx, tokens, b, Lc + 1, logits, next_tokens, (hipStream_t)stream, false, true, true);
  if (rc) return rc;
  ctx->masked_steps += 1;
  ctx->dec_mode = 2;
  return 0;
}

extern "C" int omchat_ctx_set_peer(omchat_ctx* ctx, omchat_peer* peer, size_t max_bytes, int all_sizes) {
  OM_CHECK(ctx, "null ctx");
  ctx->peer = peer;
  if (max_bytes) ctx->peer_max = max_bytes;
  ctx->peer_all = all_sizes != 0;
  return 0;
}

// in-place sum over the tensor-parallel group of a caller buffer, through the context's transports (peer kernels / RCCL); the buffer
// must be 16-byte aligned with >= 16 bytes of slack after `count` elements when the byte count is not a multiple of 16
extern "C" int omchat_ctx_allreduce(omchat_ctx* ctx, void* buf, size_t count, int dtype, void* stream) {
  OM_CHECK(ctx && buf, "null argument");
  OM_CHECK(dtype == OMCHAT_F16 || dtype == OMCHAT_BF16 || dtype == OMCHAT_F32, "bad dtype");
  return ctx->allreduce_any(buf, count, dtype, (hipStream_t)stream);
}

extern "C" int omchat_ctx_comm_stats(omchat_ctx* ctx, long* peer_calls, long* rccl_calls) {
  OM_CHECK(ctx, "null ctx");
  if (peer_calls) *peer_calls = ctx->n_ar_peer + ctx->n_fused_norm;
  if (rccl_calls) *rccl_calls = ctx->n_ar_rccl;
  return 0;
}

// sequence-parallel exchanges so far: reduce-scatters and all-gathers over row blocks (model.hip gemm_sp; 0 / 0 = the all-reduce form ran)
extern "C" int omchat_ctx_sp_stats(omchat_ctx* ctx, long* reduce_scatters, long* all_gathers) {
  OM_CHECK(ctx, "null ctx");
  if (reduce_scatters) *reduce_scatters = ctx->n_rs;
  if (all_gathers) *all_gathers = ctx->n_ag;
  return 0;
}

extern "C" int omchat_set_allreduce_hook(omchat_ctx* ctx, omchat_allreduce_fn fn, void* user) {
  OM_CHECK(ctx, "null ctx");
  ctx->hook = fn; ctx->hook_user = user;
  return 0;
}

extern "C" int omchat_prof_enable(omchat_ctx* ctx, int on) {
  OM_CHECK(ctx, "null ctx");
  ctx->prof_on = on != 0;
  return 0;
}

extern "C" int omchat_prof_read(omchat_ctx* ctx, int cat, double* total_ms, long* launches, int reset) {
  OM_CHECK(ctx && cat >= 0 && cat < OMCHAT_PROF_CATS && total_ms && launches, "bad argument");
  omchat_ctx::Prof& p = ctx->prof[cat];
  OM_HIP(hipDeviceSynchronize());
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    float ms = 0.f;
    OM_HIP(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
    p.ms += ms; p.count += 1;
  }
  p.used = 0;
  *total_ms = p.ms; *launches = p.count;
  if (reset) { p.ms = 0; p.count = 0; }
  return 0;
}

// Fused decode launches (fused_decode.hip): how many ran, and the sticky time-out bits of their in-launch hand-offs (0 = every hand-off
// completed; bit 0 = a merge sweep, bit 1 = a row sweep gave up after its wall-clock budget -- the step's results are then wrong and the
// caller must not use them).  Synchronises the device.
extern "C" int omchat_fused_status(omchat_ctx* ctx, long* launches, unsigned* timeout_bits) {
  OM_CHECK(ctx, "null ctx");
  if (launches) *launches = ctx->n_fused_launches + ctx->n_layer_launches;
  if (timeout_bits) {
    *timeout_bits = 0;
    if (ctx->fd_err) OM_HIP(hipMemcpy(timeout_bits, ctx->fd_err, 4, hipMemcpyDeviceToHost));
  }
#if OMCHAT_EXPERIMENTS
  if ((g_ao_oproj & 16) && ctx->dbg_stamps) {
    // measurement: clock stamps (100 MHz) of the LAST decode step, averaged over the layers: every kernel of the six-launch layer from its first wave's
    // start to its last wave's end, and the boundary in front of it (first start - the previous kernel's end)
    const int nl = ctx->c.t_layers;
    std::vector<unsigned long long> st((size_t)nl * 32);
    OM_HIP(hipMemcpy(st.data(), ctx->dbg_stamps, (size_t)nl * 256, hipMemcpyDeviceToHost));
    // order in the layer: qkv (8, 9), attention (10, 11), merge (12, 0), o_proj (1, 3), gate|up (4, 5), down_proj (6, 7)
    static const int S0[6] = {8, 10, 12, 1, 4, 6}, S1[6] = {9, 11, 0, 3, 5, 7};
    static const char* NM[6] = {"qkv", "attention", "merge", "o_proj", "gate|up", "down_proj"};
    double dur[6] = {0, 0, 0, 0, 0, 0}, gap[6] = {0, 0, 0, 0, 0, 0}, seen = 0, xcd_end[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xcd_raw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n = 0, ngap0 = 0, nx = 0, xcd_slowest[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nl; ++i) {
      const unsigned long long* t = &st[(size_t)i * 32];
      bool ok = true;
      for (int k = 0; k < 6; ++k) ok = ok && t[S0[k]] && t[S1[k]];
      if (!ok) continue;
      for (int k = 0; k < 6; ++k) {
        dur[k] += ((double)t[S1[k]] - (double)~t[S0[k]]) / 100.0;
        if (k > 0) gap[k] += ((double)~t[S0[k]] - (double)t[S1[k - 1]]) / 100.0;
      }
      if (i > 0 && st[(size_t)(i - 1) * 32 + 7]) { gap[0] += ((double)~t[8] - (double)st[(size_t)(i - 1) * 32 + 7]) / 100.0; ++ngap0; }
      if (t[2]) seen += ((double)t[2] - (double)t[0]) / 100.0;
      {      // gate|up: the end per XCD relative to the launch's first start, sorted within the layer (which XCD is slow changes from launch to launch)
        double e[8]; bool all = true;
        for (int x = 0; x < 8; ++x) { all = all && t[16 + x]; e[x] = ((double)t[16 + x] - (double)~t[4]) / 100.0; }
        if (all) {
          int slow = 0;
          for (int x = 0; x < 8; ++x) { xcd_raw[x] += e[x]; if (e[x] > e[slow]) slow = x; }
          ++xcd_slowest[slow];
          std::sort(e, e + 8); for (int x = 0; x < 8; ++x) xcd_end[x] += e[x]; ++nx;
        }
      }
      ++n;
    }
    if (n) {
      fprintf(stderr, "[omchat dbg] key 42 = %d, last decode step, mean over %d layers, us (boundary in front | first start -> last end):", g_ao_oproj, n);
      double sd = 0, sg = 0;
      for (int k = 0; k < 6; ++k) {
        const double g = k == 0 ? (ngap0 ? gap[0] / ngap0 : 0.0) : gap[k] / n;
        fprintf(stderr, " %s %.2f | %.2f;", NM[k], g, dur[k] / n);
        sd += dur[k] / n; sg += g;
      }
      fprintf(stderr, " kernels %.2f + boundaries %.2f = %.2f per layer", sd, sg, sd + sg);
      if (seen > 0) fprintf(stderr, "; flags seen %.2f after the merge's end", seen / n);
      if (nx) {
        fprintf(stderr, "; gate|up end per XCD after its first start, fastest to slowest:");
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.2f", xcd_end[x] / nx);
        fprintf(stderr, "; by XCD id 0..7:");
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.2f", xcd_raw[x] / nx);
        fprintf(stderr, "; layers in which XCD x ended last:");
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %d", xcd_slowest[x]);
      }
      fprintf(stderr, "\n");
    }
  }
#endif
  return 0;
}

// Forget the last n decode steps of sequences 0..b-1 (their cache rows stay in memory and are overwritten by the next steps): the drop-in
// generate() enqueues step k + 1 before it has looked at token k on the host, and takes the step back when token k ends the generation.
extern "C" int omchat_kv_rewind(omchat_ctx* ctx, int b, int n, void* stream) {
  OM_CHECK(ctx && b >= 1 && b <= (int)ctx->h_len.size() && n >= 0, "bad argument");
  if (n == 0) return 0;
  if (ctx->dec_mode == 2) {
    OM_CHECK(n <= ctx->masked_steps, "rewind beyond the prefill");
    ctx->masked_steps -= n;
    // the device-resident positions (omchat_decode_step_masked_next) go back with the slots; the host-mask form passes its positions every step
    hipLaunchKernelGGL(add_positions_kernel, dim3(1), dim3(64 > b ? 64 : b), 0, (hipStream_t)stream, ctx->d_pos, b, -n);
    OM_LAUNCH_CHECK();
    return 0;
  }
  std::vector<int> pos(b), len1(b);
  for (int i = 0; i < b; ++i) {
    OM_CHECK(ctx->h_len[i] - n >= 1, "rewind beyond the prefill");
    ctx->h_len[i] -= n;
    pos[i] = ctx->h_len[i]; len1[i] = ctx->h_len[i] + 1;
  }
  hipStream_t s = (hipStream_t)stream;
  OM_HIP(hipMemcpyAsync(ctx->d_pos, pos.data(), (size_t)b * 4, hipMemcpyHostToDevice, s));
  OM_HIP(hipMemcpyAsync(ctx->d_len, len1.data(), (size_t)b * 4, hipMemcpyHostToDevice, s));
  OM_HIP(hipStreamSynchronize(s));     // stack vectors
  return 0;
}

extern "C" int omchat_kv_lengths(omchat_ctx* ctx, int32_t* out, int b) {
  OM_CHECK(ctx && out && b <= (int)ctx->h_len.size(), "bad argument");
  // after a left-padded prefill, and once a masked decode step ran, every row holds the same number of cache slots
  for (int i = 0; i < b; ++i) out[i] = (ctx->left_padded || ctx->dec_mode == 2) ? ctx->pre_S + ctx->masked_steps : ctx->h_len[i];
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// tensor-parallel bootstrap
// ---------------------------------------------------------------------------------------------------------
extern "C" int omchat_comm_unique_id(char id[128]) {
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than 128 bytes");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) { omchat_set_error(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r)); return 3; }
  memset(id, 0, 128);
  memcpy(id, &u, sizeof(u));
  return 0;
}
extern "C" int omchat_comm_init(const char id[128], int rank, int size, void** comm_out) {
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm;
  ncclResult_t r = ncclCommInitRank(&comm, size, u, rank);
  if (r != ncclSuccess) { omchat_set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); return 3; }
  *comm_out = comm;
  return 0;
}
extern "C" int omchat_comm_allreduce(void* comm, void* buf, size_t count, int dtype, void* stream) {
  OM_CHECK(comm && buf, "null argument");
  const ncclDataType_t t = dtype == OMCHAT_F16 ? ncclFloat16 : dtype == OMCHAT_BF16 ? ncclBfloat16 : ncclFloat32;
  OM_CHECK(dtype == OMCHAT_F16 || dtype == OMCHAT_BF16 || dtype == OMCHAT_F32, "bad dtype");
  ncclResult_t r = ncclAllReduce(buf, buf, count, t, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  if (r != ncclSuccess) { omchat_set_error(std::string("ncclAllReduce: ") + ncclGetErrorString(r)); return 3; }
  return 0;
}
extern "C" int omchat_comm_count(void* comm, int* nranks) {
  OM_CHECK(comm && nranks, "null argument");
  ncclResult_t r = ncclCommCount((ncclComm_t)comm, nranks);
  if (r != ncclSuccess) { omchat_set_error(std::string("ncclCommCount: ") + ncclGetErrorString(r)); return 3; }
  return 0;
}
extern "C" void omchat_comm_destroy(void* comm) {
  if (comm) ncclCommDestroy((ncclComm_t)comm);
}
