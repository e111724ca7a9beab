// Batch-1 decode, one GPU: ONE Qwen2 decoder layer as ONE launch (round 4).
//
// Qwen2DecoderLayer.forward for a single new token with a KV cache (transformers modeling_qwen2.py:269-298: input RMSNorm -> q/k/v
// projections with bias -> RoPE -> cache append -> GQA attention -> o_proj -> residual -> post-attention RMSNorm -> SwiGLU MLP -> residual),
// the six launches of the batch-1 decode step (model.hip: qkv GEMV with the norm in registers, split-KV attention, merge, o_proj + residual,
// gate|up GEMV with the norm in registers, down_proj + residual) as one persistent grid: one workgroup per CU, eight waves, every
// dependency between the phases an in-launch hand-off instead of a kernel boundary.
//
// SAME BITS as the six launches: every output element is computed by one wave with the arithmetic and the summation order of the kernel it
// replaces (gemv_rows_norm_kernel / gemv_rows_norm_loop_kernel / gemv_rows_longk_kernel / gemv_rows_kernel, attn_decode_tile,
// attn_merge_kernel for <= 64 partials); only WHO computes an element and WHEN changes.  tests/test_gpu_round4.py compares logits bit for bit.
//
// Why (measured on MI355X, profiles/r04_*): a dependent kernel boundary of this step costs ~4 us (2.4 us between empty kernels + the first
// memory round trip of the next kernel + the tail of the previous one), six of them per layer = a quarter of the 91 us layer; an
// all-to-all hand-off of an activation row inside a launch costs 2.1-2.3 us (tools/experiments/tune_handoff.hip), and -- because a CU's memory pipe
// returns in order -- the polls of a hand-off wait behind whatever that CU has in flight.  So every hand-off here is
//     publish results -> issue the FIRST loads of the next phase's weights -> sweep
// and the sweep's latency runs under weight bytes that had to be loaded anyway.  The K / V tile of the attention is requested at kernel
// entry, behind the qkv weights, and has landed when q arrives.
//
// Work split (G workgroups = CUs, c = blockIdx.x; waves 0..6 = row waves, wave 7 = attention wave):
//   qkv      rows [c r_qkv, (c+1) r_qkv) of the fused weight, RQ rows per row wave; the input RMSNorm shared through LDS (four-wave order)
//   attention unit (kv head, 64-key split) = c on wave 7;  merge unit (head, 16-column group) = c on wave 7
//   o_proj   rows [c r_o, (c+1) r_o), RO rows per row wave; x + attn stays in lane 0's registers as the residual of down_proj
//   gate|up  outputs [c n_gu, (c+1) n_gu) as (gate, up) row pairs, output pairs dealt to the row waves, three register buffers per wave
//   down     rows [c r_o, (c+1) r_o), the activation row (It values) gathered into LDS, passes of 8 x 512 with three register buffers
// Hand-offs: 8-byte {tag, value} granules, ONE sc1 store each, swept with sc1 loads until every tag is this launch's tag
// (cdna_hip_programming.md Guideline 16, R2: the data is the flag; no fence, no counter; placement independent); five buffers (q/k/v
// values, attention partials, merged attention row, x + attn row, activation row), so one tag per launch serves all.  Every spin is
// bounded by the wall clock and reports through a sticky error word; the launch needs all its workgroups resident at once (grid = CUs,
// 84 KB of LDS per workgroup = one per CU) and nothing else of this kind beside it.
#include "kernels.h"
#if OMCHAT_EXPERIMENTS
#include "attn_common.h"
#include "rowdot.h"

namespace {

typedef unsigned long long u64;
constexpr int DL_PG_STRIDE = 132;                 // granules per (head, split): 128 O columns, m, l, 2 pad
constexpr int DL_STAGE = 16384;                   // LDS map (bytes): V image | merge staging | weights | q/k/v rows | red | raw row | row | act
constexpr int DL_FW = DL_STAGE + 16 * 64 * 4;
constexpr int DL_QKV = DL_FW + 256;
constexpr int DL_RED = DL_QKV + 18 * 128 * 2;
constexpr int DL_RAW = 25600;
constexpr int DL_ROW = DL_RAW + 8192;
constexpr int DL_ACT = DL_ROW + 8192;
constexpr int DL_ACT_MAX_CHUNKS = 40;             // activation row up to 40 x 512 elements in LDS
constexpr size_t DL_LDS_REQUEST = 84 * 1024;      // >= DL_ACT + 40 KB, and more than half of the CU's 160 KB: one workgroup per CU
static_assert(DL_RED + 64 <= DL_RAW && DL_ACT + DL_ACT_MAX_CHUNKS * 1024 <= (int)DL_LDS_REQUEST, "LDS map");

struct DecLayerP {
  const void *ln1, *ln2, *wqkv, *bqkv, *wo, *wgu, *wd;
  void *kc, *vc;                 // this layer's cache [kv_heads][cap][128], post-RoPE K / raw V
  void* x;                       // residual stream [H]: read at entry, the layer's output written in place
  int H, qd, kvd, It, q_heads, kv_heads;
  int kv_len;                    // keys after this step; the new token sits at kv_len - 1 (position and cache slot)
  int64_t k_sh;                  // cache head stride (elements)
  const float* rope; int rope_max;
  float eps, c;                  // RMSNorm eps; softmax scale * log2(e)
  u64 *qkv_g, *part_g, *ao_g, *x2_g, *act_g, *done_g;      // done_g [G]: workgroup c has published all its activation values
  unsigned tag;
  unsigned* err; u64 timeout_ticks;
  int r_qkv, r_o, n_gu;          // qkv rows, o_proj / down rows, gate|up outputs per workgroup
  u64* dbg;                      // diagnostic build only: [gridDim.x][32] phase stamps
};

#ifdef OMCHAT_FUSED_STAMPS
#define DL_STAMP(slot) do { if (P.dbg && lane == 0) P.dbg[(size_t)blockIdx.x * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DL_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ void g_store(u64* p, unsigned tag, unsigned v) {
  __hip_atomic_store(p, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 g_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lds_fence_wave() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// workgroup barrier that leaves global loads in flight (LDS traffic of this wave drained first)
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ bool spin_expired(u64 t0, const DecLayerP& P, unsigned code, int lane) {
  if (__builtin_amdgcn_s_memrealtime() - t0 <= P.timeout_ticks) return false;
  if (lane == 0) atomicOr(P.err, code);
  return true;
}

// This wave's share [lo, hi) of a granule buffer into LDS words (dst[i] = value of granule i), PER granules per lane and pass, until every
// tag matches.  HALF: the granule value is ONE 16-bit element (dst is a 16-bit array) instead of a 32-bit word.
template <bool HALF, int PER = 4>
__device__ __forceinline__ void sweep_range(const u64* g, int lo, int hi, void* dst, const DecLayerP& P, unsigned code, int lane) {
  if (lo >= hi) return;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  for (int g0 = lo; g0 < hi; g0 += 64 * PER) {
    for (;;) {
      bool ok = true;
      unsigned v[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = g0 + k * 64 + lane;
        const u64 x = g_load(g + (i < hi ? i : hi - 1));
        v[k] = (unsigned)x; ok &= (unsigned)(x >> 32) == P.tag;
      }
      if (__all(ok)) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int i = g0 + k * 64 + lane;
          if (i < hi) { if constexpr (HALF) ((unsigned short*)dst)[i] = (unsigned short)v[k]; else ((unsigned*)dst)[i] = v[k]; }
        }
        break;
      }
      if (spin_expired(t0, P, code, lane)) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// Qwen2RMSNorm of the row that sits RAW in LDS (16-bit, NCH chunks of 512, zero beyond H) into `row`: the sum of squares in the four-wave
// order of gemv_rows_norm_kernel (waves 0..3 own chunks w and w + 4, per-lane accumulation, butterfly, ((r0 + r1) + r2) + r3), chunk w
// normalised by wave w with the norm weights it loaded earlier.  Two workgroup barriers inside; every wave of the workgroup must call it.
template <typename T, int NCH>
__device__ __forceinline__ void rmsnorm_lds(const char* raw, char* row, float* red, const rw_u32x4 (&nq)[(NCH + 7) / 8], int H, float eps, int lane, int wave) {
  typedef typename V8<T>::type v8;
  constexpr int SC = (NCH + 3) / 4;
  float ss = 0.f;
  if (wave < 4) {
#pragma unroll
    for (int i = 0; i < SC; ++i) {
      const int ch = wave + 4 * i;
      if (ch < NCH) {
        const v8 xv = *reinterpret_cast<const v8*>(raw + (ch * 512 + lane * 8) * 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = tof(xv[j]); ss += v * v; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0 && wave < 4) red[wave] = ss;
  wg_barrier();
  float tot = red[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) tot += red[w];
  const float inv = rsqrtf(tot / (float)H + eps);
#pragma unroll
  for (int i = 0; i < (NCH + 7) / 8; ++i) {
    const int ch = wave + 8 * i;
    if (ch < NCH) {
      const v8 xv = *reinterpret_cast<const v8*>(raw + (ch * 512 + lane * 8) * 2), wv = __builtin_bit_cast(v8, nq[i]);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[j]) * rnd<T>(tof(xv[j]) * inv));
      *reinterpret_cast<v8*>(row + (ch * 512 + lane * 8) * 2) = o;
    }
  }
  wg_barrier();
}

// NCH = chunks of 512 of the hidden size, NCQ = of the attention width (q_heads * 128), RQ / RO = qkv / o_proj + down rows per row wave
template <typename T, int NCH, int NCQ, int RQ, int RO>
__global__ __launch_bounds__(512) void decode_layer_kernel(DecLayerP P) {
  typedef typename V8<T>::type v8;
  typedef typename V8<T>::type frag_t;
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x;
  const int kv_len = P.kv_len, pp = kv_len - 1;
  const int ns = (kv_len + KV_TILE - 1) / KV_TILE;
  const int n_rep = P.q_heads / P.kv_heads;
  const int qkvd = P.qd + 2 * P.kvd;
  float* red = (float*)(smem + DL_RED);
  unsigned* gu_done = (unsigned*)(smem + DL_RED + 32);      // row waves of this workgroup whose gate|up outputs are published and drained
  char* raw = smem + DL_RAW;
  char* row = smem + DL_ROW;
  const bool rowwave = wave < 7;
  constexpr int NQ = (NCH + 7) / 8;                             // norm-weight chunks per wave (chunk = wave + 8 i)

  // =========================================================================================== 1. entry: every load that depends on nothing
  // row waves: the qkv weight rows; all waves: x (raw, into LDS for the shared norm), both norm weights
  rw_u32x4 wq[RQ][NCH];
  int qrow[RQ]; bool qvalid[RQ];
  if (rowwave) {
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int local = wave * RQ + r;
      int n = c * P.r_qkv + local;
      qvalid[r] = local < P.r_qkv && n < qkvd;
      n = n < qkvd ? n : qkvd - 1;
      qrow[r] = n;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        int k = ch * 512 + lane * 8;
        k = k < P.H ? k : 0;
        wq[r][ch] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)P.wqkv + (size_t)n * P.H + k));
      }
    }
  }
  rw_u32x4 xq[NQ], n1q[NQ], n2q[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int ch = wave + 8 * i, k = ch * 512 + lane * 8;
    const rw_u32x4 z = {0u, 0u, 0u, 0u};
    const bool ok = ch < NCH && k < P.H;
    xq[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)P.x + k) : z;
    n1q[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)P.ln1 + k) : z;
    n2q[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)P.ln2 + k) : z;
  }
  __builtin_amdgcn_sched_barrier(0);
  const int kvh = c % P.kv_heads, split = c / P.kv_heads;
  const bool has_tile = split < ns;
  const int key0 = split * KV_TILE;
  const int fc = lane & 15, fg = lane >> 4;

  // =========================================================================================== 2. input RMSNorm (shared) + qkv rows
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int ch = wave + 8 * i;
    if (ch < NCH) *reinterpret_cast<rw_u32x4*>(raw + (ch * 512 + lane * 8) * 2) = xq[i];
  }
  if (threadIdx.x == 0) *gu_done = 0u;
  wg_barrier();                                   // (1) the raw row is in LDS
  rmsnorm_lds<T, NCH>(raw, row, red, n1q, P.H, P.eps, lane, wave);      // barriers (2), (3)
  if (rowwave) {
    rw_u32x4 xr[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) xr[ch] = *reinterpret_cast<const rw_u32x4*>(row + (ch * 512 + lane * 8) * 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      float a = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) a = rw_dot8<T>(wq[r][ch], xr[ch], a);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0 && qvalid[r]) {
        const float y = a + (P.bqkv ? tof(((const T*)P.bqkv)[qrow[r]]) : 0.f);
        g_store(P.qkv_g + qrow[r], P.tag, (unsigned)__builtin_bit_cast(unsigned short, fromf<T>(y)));
      }
    }
    if (wave == 0) DL_STAMP(8);
  }

  // (a convergent no-op: without it the compiler threads the role branch above into the one below as ONE flow region, in which the qkv
  // weight registers count as live across the attention wave's code -- 32 to 56 of them were then spilled right behind their loads, each
  // with a full wait)
  __builtin_amdgcn_wave_barrier();
  // =========================================================================================== 3. o_proj weights requested; attention
  rw_u32x4 wo[RO][NCQ];
  float resid[RO];
  int orow[RO]; bool ovalid[RO];
  if (rowwave) {
    // (only the FIRST row of each wave now: the attention wave's polls and stores share this CU's in-order memory pipe with these loads, and
    // ~50 KB drain in about the time the hand-off takes anyway; the second row follows behind barrier (4))
#pragma unroll
    for (int r = 0; r < RO; ++r) {
      const int local = wave * RO + r;
      int n = c * P.r_o + local;
      ovalid[r] = local < P.r_o && n < P.H;
      n = n < P.H ? n : P.H - 1;
      orow[r] = n;
      if (r == 0) {
#pragma unroll
        for (int ch = 0; ch < NCQ; ++ch) {
          int k = ch * 512 + lane * 8;
          k = k < P.qd ? k : 0;
          wo[r][ch] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)P.wo + (size_t)n * P.qd + k));
        }
      }
      resid[r] = tof(*reinterpret_cast<const T*>(raw + n * 2));      // x[n] (the raw row is still in LDS)
    }
  } else {
    // ---- the K / V tile of unit (kv head, split) = c from the cache and the RoPE row of the new position: requested now -- behind the
    // qkv weights in this CU's memory pipe, ahead of the o_proj weights -- and landed long before q arrives
    frag_t kf[4][4], vreg[16];
    float csv[2][16];
    DL_STAMP(0);
    if (has_tile) {
      const T* Kg = (const T*)P.kc + kvh * P.k_sh;
      const T* Vg = (const T*)P.vc + kvh * P.k_sh;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int key = key0 + kt * 16 + fc;
        const T* src = Kg + (int64_t)(key < pp ? key : (pp > 0 ? pp - 1 : 0)) * 128;       // rows >= pp are replaced after the hand-off
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) kf[kt][ds] = ld8<T>(src + ds * 32 + fg * 8);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = key0 + i * 4 + fg;
        const T* src = Vg + (int64_t)(key < pp ? key : (pp > 0 ? pp - 1 : 0)) * 128;
        vreg[i] = ld8<T>(src + fc * 8);
      }
      const int pt = pp < P.rope_max ? pp : P.rope_max - 1;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const float* cs = P.rope + ((size_t)pt * 64 + ds * 32 + fg * 8) * 2;
#pragma unroll
        for (int j = 0; j < 16; ++j) csv[ds][j] = cs[j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the V rows go to their LDS transpose image as they land (64 registers less across the hand-off; the row of the new token is
    // patched in below, rows beyond it are masked keys: weight exactly 0 whatever finite values they hold), and the RoPE row parks in
    // the (still unused) activation area of LDS, 128 bytes per lane, until q is there
    char* Vs = smem;
    float* cs_park = (float*)(smem + DL_ACT) + lane * 32;
    char* k_park = smem + DL_ACT + 8192 + lane * 16;            // K fragments, lane-linear: fragment (kt, ds) at (kt * 4 + ds) * 1024
    if (has_tile) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) *reinterpret_cast<frag_t*>(k_park + (kt * 4 + ds) * 1024) = kf[kt][ds];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rw_ = i * 4 + fg;
        *reinterpret_cast<frag_t*>(Vs + rw_ * 256 + ((fc ^ ((rw_ & 7) << 1)) << 4)) = vreg[i];
      }
#pragma unroll
      for (int ds = 0; ds < 2; ++ds)
#pragma unroll
        for (int j = 0; j < 16; j += 4) *reinterpret_cast<f32x4*>(cs_park + ds * 16 + j) = (f32x4){csv[ds][j], csv[ds][j + 1], csv[ds][j + 2], csv[ds][j + 3]};
    }
    // ---- hand-off 1: q of this kv head's group, k and v of the new token
    T* qs = (T*)(smem + DL_QKV);                  // [n_rep][128] q, [128] k, [128] v
    const int nq = n_rep * 128;
    if (has_tile) {
      sweep_range<true>(P.qkv_g + (size_t)kvh * nq, 0, nq, qs, P, 1u, lane);
      sweep_range<true>(P.qkv_g + P.qd + (size_t)kvh * 128, 0, 128, qs + nq, P, 1u, lane);
      sweep_range<true>(P.qkv_g + P.qd + P.kvd + (size_t)kvh * 128, 0, 128, qs + nq + 128, P, 1u, lane);
      lds_fence_wave();
      DL_STAMP(1);
      // ---- the tile (attn_decode_tile's arithmetic; K / V came from the cache before q existed, the rows >= pp are patched in now)
      const T* kn = qs + nq;
      const T* vn = qs + nq + 128;
      bool kfresh[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) kfresh[kt] = key0 + kt * 16 + fc >= pp;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rw_ = i * 4 + fg;
        if (key0 + rw_ == pp) {                   // the new token's v: into the image and (append) into the cache
          const frag_t vnew = *reinterpret_cast<const frag_t*>(vn + fc * 8);
          *reinterpret_cast<frag_t*>(Vs + rw_ * 256 + ((fc ^ ((rw_ & 7) << 1)) << 4)) = vnew;
          st8<T>((T*)P.vc + kvh * P.k_sh + (int64_t)pp * 128 + fc * 8, vnew);
        }
      }
      frag_t qf[4], knr[4];                       // q of this lane's head; the new token's k (every row >= pp of the tile reads it)
      {
        const int hh = fc < n_rep ? fc : n_rep - 1;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          qf[ds] = *reinterpret_cast<const frag_t*>(qs + hh * 128 + ds * 32 + fg * 8);
          knr[ds] = *reinterpret_cast<const frag_t*>(kn + ds * 32 + fg * 8);
        }
      }
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        float cs[16];
#ifdef DL_DBG_CS_GLOBAL
        {
          const int pt_ = pp < P.rope_max ? pp : P.rope_max - 1;
          const float* csg = P.rope + ((size_t)pt_ * 64 + ds * 32 + fg * 8) * 2;
#pragma unroll
          for (int j = 0; j < 16; ++j) cs[j] = csg[j];
        }
#else
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(cs_park + ds * 16 + j);
          cs[j] = t4[0]; cs[j + 1] = t4[1]; cs[j + 2] = t4[2]; cs[j + 3] = t4[3];
        }
#endif
        const frag_t lo = qf[ds], hi = qf[ds + 2];
        qf[ds] = rope_chunk<T>(lo, hi, cs, false);
        qf[ds + 2] = rope_chunk<T>(hi, lo, cs, true);
        const frag_t kl = knr[ds], kh = knr[ds + 2];
        knr[ds] = rope_chunk<T>(kl, kh, cs, false);
        knr[ds + 2] = rope_chunk<T>(kh, kl, cs, true);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (key0 + kt * 16 + fc == pp) {
#pragma unroll
          for (int ds = 0; ds < 4; ++ds) st8<T>((T*)P.kc + kvh * P.k_sh + (int64_t)pp * 128 + ds * 32 + fg * 8, knr[ds]);
        }
      f32x4 s[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
#ifdef DL_DBG_K_GLOBAL
          const int key_ = key0 + kt * 16 + fc;
          const frag_t kp = ld8<T>((const T*)P.kc + kvh * P.k_sh + (int64_t)(key_ < pp ? key_ : (pp > 0 ? pp - 1 : 0)) * 128 + ds * 32 + fg * 8);
#else
          const frag_t kp = *reinterpret_cast<const frag_t*>(k_park + (kt * 4 + ds) * 1024);
#endif
          s[kt] = mfma16(kfresh[kt] ? knr[ds] : kp, qf[ds], s[kt]);
        }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = key0 + kt * 16 + 4 * fg + r < kv_len ? s[kt][r] : NEG_BIG;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = max_xor32(max_xor16(mx));
      const float mc = mx * P.c;
      float psum = 0.f;
      frag_t pf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        f32x8 e;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          e[j] = __builtin_amdgcn_exp2f(fmaf(s[2 * ks + (j >> 2)][j & 3], P.c, -mc));
          psum += e[j];
        }
        pf[ks] = __builtin_convertvector(e, frag_t);
      }
      const float l = sum_xor32(sum_xor16(psum));
      lds_fence_wave();
      const int tq = fc >> 2, tp = fc & 3;
      const int vrow_lo = 4 * fg + tq;
      const int vswz = ((vrow_lo & 7) << 1);
      f32x4 o[8];
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dn = 0; dn < 8; ++dn) {
          const int chh = (2 * dn + (tp >> 1)) ^ vswz;
          const char* a0 = Vs + (ks * 32 + vrow_lo) * 256 + (chh << 4) + 8 * (tp & 1);
          const s16x4 lo = tr_read(a0);
          const s16x4 hi = tr_read(a0 + 16 * 256);
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[dn] = mfma16(__builtin_bit_cast(frag_t, cat), pf[ks], o[dn]);
        }
      DL_STAMP(2);
      if (fc < n_rep) {
        u64* base = P.part_g + ((size_t)(kvh * n_rep + fc) * 64 + split) * DL_PG_STRIDE;
#pragma unroll
        for (int dn = 0; dn < 8; ++dn)
#pragma unroll
          for (int r = 0; r < 4; ++r) g_store(base + dn * 16 + fg * 4 + r, P.tag, __float_as_uint(o[dn][r]));
        if (fg == 0) { g_store(base + 128, P.tag, __float_as_uint(mx)); g_store(base + 129, P.tag, __float_as_uint(l)); }
      }
    }
    DL_STAMP(3);
    // ---- hand-off 2 + merge of unit (head, 16-column group) = c: attn_merge_kernel's arithmetic for <= 64 partials, lane = split
    if (c < P.q_heads * 8) {
      const int h = c >> 3, dg = c & 7;
      const int s = lane, sc = s < ns ? s : ns - 1;
      const u64* base = P.part_g + ((size_t)h * 64 + sc) * DL_PG_STRIDE;
      unsigned xv[16], xm = 0, xl = 0;
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const u64 g = g_load(base + dg * 16 + j); xv[j] = (unsigned)g; ok &= (unsigned)(g >> 32) == P.tag; }
        { const u64 g = g_load(base + 128); xm = (unsigned)g; ok &= (unsigned)(g >> 32) == P.tag; }
        { const u64 g = g_load(base + 129); xl = (unsigned)g; ok &= (unsigned)(g >> 32) == P.tag; }
        if (__all(ok || s >= ns)) break;
        if (spin_expired(t0, P, 2u, lane)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      DL_STAMP(4);
      const float m_s = s < ns ? __uint_as_float(xm) : NEG_BIG;
      const float l_s = s < ns ? __uint_as_float(xl) : 0.f;
      const float m = wave_max(m_s);
      const float fw = s < ns ? exp2f((m_s - m) * P.c) : 0.f;
      const float lsum = wave_sum(fw * l_s);
      const float ltot = __shfl(lsum, 0, 64);
      float* stage = (float*)(smem + DL_STAGE);
      float* fws = (float*)(smem + DL_FW);
      fws[s] = fw;
#pragma unroll
      for (int j = 0; j < 16; ++j) stage[j * 64 + s] = s < ns ? __uint_as_float(xv[j]) : 0.f;
      lds_fence_wave();
      float a = 0.f;
      const int d = lane & 15;
      // (16-byte LDS reads, the 64 terms added in split order: the same sum as attn_merge_kernel's)
#pragma unroll
      for (int k = 0; k < 64; k += 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(fws + k), v4 = *reinterpret_cast<const f32x4*>(stage + d * 64 + k);
        a += w4[0] * v4[0]; a += w4[1] * v4[1]; a += w4[2] * v4[2]; a += w4[3] * v4[3];
      }
      const T ov = fromf<T>(a / ltot);
      const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, ov);
      const unsigned hi = (unsigned)__shfl_down((int)bits, 1, 64);
      if (lane < 16 && (lane & 1) == 0) g_store(P.ao_g + ((h * 128 + dg * 16 + lane) >> 1), P.tag, bits | (hi << 16));
    }
    DL_STAMP(5);
  }
  wg_barrier();                                   // (4) the row waves poll from here on; the row buffer is free (qkv rows done)
  if (rowwave) {
#pragma unroll
    for (int r = 1; r < RO; ++r) {
#pragma unroll
      for (int ch = 0; ch < NCQ; ++ch) {
        int k = ch * 512 + lane * 8;
        k = k < P.qd ? k : 0;
        wo[r][ch] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)P.wo + (size_t)orow[r] * P.qd + k));
      }
    }
  }

  // =========================================================================================== 4. hand-off 3: merged attention row; o_proj
  {
    const int n_pairs = P.qd >> 1, per = (n_pairs + 7) >> 3;
    const int lo = wave * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
    sweep_range<false>(P.ao_g, lo, hi, row, P, 4u, lane);
    // zero tail of the attention row (K padded to whole chunks; the weights are clamped there)
    for (int i = n_pairs + (int)threadIdx.x; i < NCQ * 256; i += 512) ((unsigned*)row)[i] = 0u;
  }
  if (wave == 7) DL_STAMP(6);
  wg_barrier();                                   // (5)
  // gate|up: output pairs (2 p, 2 p + 1), p = wave + 7 j, of this workgroup's n_gu outputs; three register buffers (one output each)
  const int gu0 = c * P.n_gu;
  const int n_gu_here = gu0 >= P.It ? 0 : (gu0 + P.n_gu <= P.It ? P.n_gu : P.It - gu0);
  const int my_out = rowwave ? (((n_gu_here + 1) / 2 > wave ? ((n_gu_here + 1) / 2 - 1 - wave) / 7 + 1 : 0) * 2) : 0;      // outputs of this wave (pairs x 2)
  auto out_of = [&](int u) { return gu0 + 2 * (wave + 7 * (u >> 1)) + (u & 1); };       // u-th output of this wave
  auto gu_row = [&](int n, int r) { return 32 * (n >> 4) + (n & 15) + r * 16; };
  rw_u32x4 ga[2][NCH], gb[2][NCH], gc[2][NCH];
  auto load_gu = [&](rw_u32x4 (&w)[2][NCH], int u) {
    int n = out_of(u); n = n < P.It ? n : P.It - 1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const size_t rowi = (size_t)gu_row(n, r);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        int k = ch * 512 + lane * 8;
        k = k < P.H ? k : 0;
        w[r][ch] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)P.wgu + rowi * P.H + k));
      }
    }
  };
  float x2v[RO];
  if (rowwave) {
    rw_u32x4 xr[NCQ];
#pragma unroll
    for (int ch = 0; ch < NCQ; ++ch) xr[ch] = *reinterpret_cast<const rw_u32x4*>(row + (ch * 512 + lane * 8) * 2);
    __builtin_amdgcn_sched_barrier(0);
    unsigned pair = 0;
#pragma unroll
    for (int r = 0; r < RO; ++r) {
      float a = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCQ; ++ch) a = rw_dot8<T>(wo[r][ch], xr[ch], a);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      float y = a + 0.f;
      y = rnd<T>(y);
      y = resid[r] + y;
      const T yt = fromf<T>(y);
      x2v[r] = tof(yt);                           // x + attn of this row: the residual of down_proj, and this row of the next norm's input
      if (lane == 0 && ovalid[r]) {
        const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, yt);
        if constexpr (RO == 2) {
          pair |= bits << (16 * r);
          if (r == 1) g_store(P.x2_g + (orow[0] >> 1), P.tag, pair);
        } else {
          g_store(P.x2_g + orow[r], P.tag, bits);
        }
      }
    }
    if (wave == 0) DL_STAMP(9);
    // the first gate|up output of this wave: requested BEFORE the sweep below, whose polls then run under these bytes
    // (four of the seven row waves: ~56 KB per CU drain in about the time the hand-off takes; the other waves load theirs behind it)
    if (my_out > 0 && wave < 4) load_gu(ga, 0);
  }
  // =========================================================================================== 5. hand-off 4: x + attn row; post-attention norm
  {
    // RO == 2: pair granules (index = row / 2); RO == 1: one value per granule
    if constexpr (RO == 2) {
      const int n_pairs = P.H >> 1, per = (n_pairs + 7) >> 3;
      const int lo = wave * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
      sweep_range<false>(P.x2_g, lo, hi, raw, P, 8u, lane);
    } else {
      const int per = (P.H + 7) >> 3;
      const int lo = wave * per, hi = lo + per < P.H ? lo + per : P.H;
      sweep_range<true>(P.x2_g, lo, hi, raw, P, 8u, lane);
    }
  }
  if (wave == 7) DL_STAMP(7);
  wg_barrier();                                   // (6) the raw x + attn row is in LDS (its zero tail beyond H is still there from the entry)
  rmsnorm_lds<T, NCH>(raw, row, red, n2q, P.H, P.eps, lane, wave);      // barriers (7), (8)

  // =========================================================================================== 6. gate|up outputs; activation row published
  // down_proj passes (8 chunks of 512 per pass, RO rows one after the other): item i = (row i / npass, pass i % npass)
  const int nch_d = (P.It + 511) >> 9, npass = (nch_d + 7) >> 3, n_items = RO * npass;
  rw_u32x4 da[8], db[8], dc[8];
  auto load_d = [&](rw_u32x4 (&w)[8], int item) {
    const int r = item / npass, ps = item - r * npass;
    const size_t rowi = (size_t)(r == 0 ? orow[0] : orow[RO - 1]);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      int k = (ps * 8 + ch) * 512 + lane * 8;
      k = k < P.It ? k : 0;
      w[ch] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)P.wd + rowi * P.It + k));
    }
  };
  if (rowwave) {
    rw_u32x4 xr[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) xr[ch] = *reinterpret_cast<const rw_u32x4*>(row + (ch * 512 + lane * 8) * 2);
    __builtin_amdgcn_sched_barrier(0);
    if (my_out > 0 && wave >= 4) load_gu(ga, 0);
    if (my_out > 1) load_gu(gb, 1);
    if (my_out > 2) load_gu(gc, 2);
    unsigned pairv = 0;
    auto finish = [&](rw_u32x4 (&w)[2][NCH], int u) {
      float acc[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float a = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) a = rw_dot8<T>(w[r][ch], xr[ch], a);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        acc[r] = a;
      }
      const float gt = rnd<T>(acc[0]), up = rnd<T>(acc[1]);
      const T yt = fromf<T>(rnd<T>(silu(gt)) * up);
      const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, yt);
      const int n = out_of(u);
      if (u & 1) {
        if (lane == 0 && n - 1 < P.It) g_store(P.act_g + (n >> 1), P.tag, pairv | (n < P.It ? bits << 16 : 0u));
      } else {
        pairv = bits;
      }
    };
    for (int u = 0; u < my_out; u += 3) {
      finish(ga, u);
      if (u + 3 < my_out) load_gu(ga, u + 3);
      if (u + 1 < my_out) {
        finish(gb, u + 1);
        if (u + 4 < my_out) load_gu(gb, u + 4);
      }
      if (u + 2 < my_out) {
        finish(gc, u + 2);
        if (u + 5 < my_out) load_gu(gc, u + 5);
      }
    }
    if (wave == 0) DL_STAMP(10);
    // this wave's activation values are out and drained -> count it; the seventh wave raises the workgroup's flag (R1: the polling side
    // then looks at ONE word per workgroup instead of re-reading 74 KB of granules per attempt while the slower CUs still stream)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned arrived = 0;
    if (lane == 0) arrived = __hip_atomic_fetch_add(gu_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived == 6u && lane == 0) g_store(P.done_g + c, P.tag, 1u);
    // the first down_proj pass of this wave: requested before the sweep of the activation row
    if (ovalid[0]) load_d(da, 0);
  } else {
    // attention wave: wait until EVERY workgroup has raised its flag (one 8-byte load per lane and attempt for up to 256 workgroups)
    const int G = (int)gridDim.x;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (int g0 = 0; g0 < G; g0 += 64 * 4) {
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = g0 + k * 64 + lane;
          const u64 x = g_load(P.done_g + (i < G ? i : G - 1));
          ok &= (unsigned)(x >> 32) == P.tag;
        }
        if (__all(ok)) break;
        if (spin_expired(t0, P, 32u, lane)) break;
        __builtin_amdgcn_s_sleep(8);
      }
    }
  }
  wg_barrier();                                   // (8b) every activation value of every workgroup is published: sweep once
  // =========================================================================================== 7. hand-off 5: activation row; down_proj
  {
    const int n_pairs = (P.It + 1) >> 1, per = (n_pairs + 7) >> 3;
    const int lo = wave * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
    sweep_range<false, 10>(P.act_g, lo, hi, smem + DL_ACT, P, 16u, lane);
    // zero tail up to the last whole pass (the weights are clamped there)
    for (int i = n_pairs + (int)threadIdx.x; i < npass * 8 * 256; i += 512) ((unsigned*)(smem + DL_ACT))[i] = 0u;
  }
  if (wave == 7) DL_STAMP(11);
  wg_barrier();                                   // (9)
  if (rowwave && ovalid[0]) {
    const T* xs = (const T*)(smem + DL_ACT);
    float acc = 0.f;
    auto dots = [&](rw_u32x4 (&w)[8], int item) {
      const int r = item / npass, ps = item - r * npass;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const rw_u32x4 xr = *reinterpret_cast<const rw_u32x4*>(xs + (ps * 8 + ch) * 512 + lane * 8);
        acc = rw_dot8<T>(w[ch], xr, acc);
      }
      if (ps == npass - 1) {                      // the row is complete: butterfly, residual, store (gemv_rows_longk_kernel's epilogue)
        float a = acc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        const bool ok = r == 0 ? ovalid[0] : ovalid[RO - 1];
        if (lane == 0 && ok) {
          const float y = rnd<T>(a + 0.f);
          ((T*)P.x)[r == 0 ? orow[0] : orow[RO - 1]] = fromf<T>((r == 0 ? x2v[0] : x2v[RO - 1]) + y);
        }
        acc = 0.f;
      }
    };
    const int items = (RO == 2 && !ovalid[RO - 1]) ? npass : n_items;
    if (items > 1) load_d(db, 1);
    for (int i = 0; i < items; i += 3) {
      if (i + 2 < items) load_d(dc, i + 2);
      dots(da, i);
      if (i + 1 < items) {
        if (i + 3 < items) load_d(da, i + 3);
        dots(db, i + 1);
      }
      if (i + 2 < items) {
        if (i + 4 < items) load_d(db, i + 4);
        dots(dc, i + 2);
      }
    }
    if (wave == 0) DL_STAMP(12);
  }
}

struct Geo { int nch, ncq, rq, ro; };

template <typename T, int NCH, int NCQ, int RQ, int RO>
int launch_one(const DecLayerP& P, int grid, hipStream_t s) {
  auto k = decode_layer_kernel<T, NCH, NCQ, RQ, RO>;
  static bool set = false;
  if (!set) { OM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DL_LDS_REQUEST)); set = true; }
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), DL_LDS_REQUEST, s, P);
  OM_LAUNCH_CHECK();
  return 0;
}

// the geometries this launch is built for: Qwen2-7B (3584 wide, 28 / 4 heads, 18944) and the tiny test configurations
template <typename T>
int launch_geo(const DecLayerP& P, const Geo& g, int grid, hipStream_t s) {
  if (g.nch == 7 && g.ncq == 7 && g.rq == 3 && g.ro == 2) return launch_one<T, 7, 7, 3, 2>(P, grid, s);
  if (g.nch == 7 && g.ncq == 1 && g.rq == 1 && g.ro == 2) return launch_one<T, 7, 1, 1, 2>(P, grid, s);      // a TP = 8 rank's widths (round 5 measurement)
  if (g.nch == 1 && g.ncq == 1 && g.rq == 1 && g.ro == 1) return launch_one<T, 1, 1, 1, 1>(P, grid, s);
  if (g.nch == 1 && g.ncq == 2 && g.rq == 1 && g.ro == 1) return launch_one<T, 1, 2, 1, 1>(P, grid, s);
  omchat_set_error("launch_decode_layer: geometry without an instantiation");
  return 1;
}

bool geo_of(const DecodeLayerArgs& a, Geo* g, int G) {
  const int qkvd = a.qd + 2 * a.kvd;
  const int r_qkv = cdiv(qkvd, G), r_o = cdiv(a.H, G);
  g->nch = cdiv(a.H, 512); g->ncq = cdiv(a.qd, 512); g->rq = cdiv(r_qkv, 7); g->ro = cdiv(r_o, 7);
  return (g->nch == 7 && g->ncq == 7 && g->rq == 3 && g->ro == 2) || (g->nch == 7 && g->ncq == 1 && g->rq == 1 && g->ro == 2) ||
         (g->nch == 1 && g->ncq == 1 && g->rq == 1 && g->ro == 1) ||
         (g->nch == 1 && g->ncq == 2 && g->rq == 1 && g->ro == 1);
}

}  // namespace

size_t decode_layer_ws_bytes(int q_heads, int H, int qd, int kvd, int It) {
  return ((size_t)(qd + 2 * kvd) + (size_t)q_heads * 64 * DL_PG_STRIDE + (size_t)q_heads * 64 + (size_t)H + (size_t)(It + 1) / 2 + 1024 + 64) * sizeof(u64);
}

bool decode_layer_ok(const DecodeLayerArgs& a) {
  const int G = device_cus();
  Geo g;
  const int ns = cdiv(a.kv_len, KV_TILE);
  if (!geo_of(a, &g, G)) return false;
  const int n_gu = 2 * cdiv(cdiv(a.It, G), 2);
  return a.kv_len >= 1 && ns <= 64 && (long)a.kv_heads * ns <= G && a.q_heads * 8 <= G && a.q_heads % a.kv_heads == 0 && a.q_heads / a.kv_heads <= 16 &&
         a.qd == a.q_heads * 128 && a.kvd == a.kv_heads * 128 && a.H % 8 == 0 && a.It % 8 == 0 && (a.H % 2) == 0 &&
         cdiv(a.It, 512) <= DL_ACT_MAX_CHUNKS && (long)n_gu * G >= a.It && a.rope && a.ws;
}

int launch_decode_layer(int dtype, const DecodeLayerArgs& a, hipStream_t s) {
  OM_CHECK(decode_layer_ok(a), "geometry outside the one-launch decode layer (batch 1, <= 4096 keys, Qwen2-7B or tiny widths)");
  const int G = device_cus();
  Geo g; geo_of(a, &g, G);
  DecLayerP P;
  P.ln1 = a.ln1; P.ln2 = a.ln2; P.wqkv = a.wqkv; P.bqkv = a.bqkv; P.wo = a.wo; P.wgu = a.wgu; P.wd = a.wd;
  P.kc = a.kc; P.vc = a.vc; P.x = a.x;
  P.H = a.H; P.qd = a.qd; P.kvd = a.kvd; P.It = a.It; P.q_heads = a.q_heads; P.kv_heads = a.kv_heads;
  P.kv_len = a.kv_len; P.k_sh = a.k_sh; P.rope = a.rope; P.rope_max = a.rope_max; P.eps = a.eps; P.c = a.scale * 1.4426950408889634f;
  u64* w = (u64*)a.ws;
  P.qkv_g = w; w += a.qd + 2 * a.kvd;
  P.part_g = w; w += (size_t)a.q_heads * 64 * DL_PG_STRIDE;
  P.ao_g = w; w += (size_t)a.q_heads * 64;
  P.x2_g = w; w += a.H;
  P.act_g = w; w += (size_t)(a.It + 1) / 2;
  P.done_g = w;
  P.tag = a.epoch;
  P.err = a.err; P.timeout_ticks = (u64)a.timeout_ms * 100000ull;
  P.r_qkv = cdiv(a.qd + 2 * a.kvd, G); P.r_o = cdiv(a.H, G); P.n_gu = 2 * cdiv(cdiv(a.It, G), 2);
  P.dbg = (u64*)a.dbg;
  if (dtype == OMCHAT_F16) return launch_geo<f16>(P, g, G, s);
  if (dtype == OMCHAT_BF16) return launch_geo<bf16>(P, g, G, s);
  omchat_set_error("launch_decode_layer: bad dtype");
  return 1;
}

#endif  // OMCHAT_EXPERIMENTS
