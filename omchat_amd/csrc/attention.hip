// Flash attention for gfx950, head_dim 128, "query on the lane" formulation.
//
// Replaces flash_attn_varlen_qkvpacked_func / _naive_attn in the ViT (intern_vit_6b/flash_attention.py:51-54,
// modeling_intern_vit.py:148-152: non-causal MHA, S = 1025) and eager_attention_forward in the Qwen2 decoder
// (modeling_qwen2.py:150-172: causal GQA with KV cache) for prefill AND decode.
//
// Everything is computed transposed so that no lane shuffles and no LDS round trip sit between the two MFMA chains:
//   S^T[key][q] = K[key][:] . Q[q][:]      A = K rows (ds_read_b128 from the K tile), B = Q^T (registers, loaded once)
//       -> lane (g = lane>>4, c = lane&15) holds scores of query c for keys 16*kt + 4*g + r   (MFMA 16x16x32 C/D map)
//   softmax over keys = over the lane's 16 registers + xor-16/xor-32 shuffles; running max / sum per lane (= per query)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]  B = P^T: the score registers themselves, packed to 16 bit (k-slot j of lane
//       group g <-> key 32*ks + 16*(j>>2) + 4*g + (j&3)), A = V^T fetched with ds_read_b64_tr_b16 in the SAME key order
//   -> O^T accumulator has the query on the lane again, so the rescale by alpha needs no cross-lane traffic.
// K tile LDS image: 256-B rows, 16-B chunk' = chunk ^ (row & 15)      (conflict-free ds_read_b128 operand reads)
// V tile LDS image: 256-B rows, 16-B chunk' = chunk ^ ((row & 7) << 1) (conflict-free transposed reads)
//
// Prefill: workgroup = 4 waves x 32 queries (2 query tiles per wave share every K/V fragment), KV tile = 64 keys,
// double-buffered in LDS and filled by LDS-DMA one tile ahead.
// Decode: one wave per (sequence, kv head, 64-key split); the "queries" are the n_rep heads sharing that kv head;
// partial (m, l, O) go to a workspace and a merge kernel normalises.
#include "kernels.h"

#include "attn_common.h"

namespace {

// Prefill kernel.  K/V tiles go HBM -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write): the LDS
// destination is lane-linear (wave instruction i of wave w fills rows 4*(NW*i + w) .. +3), so the bank swizzles are
// applied to the per-lane SOURCE chunk (physical chunk pc of row r holds logical chunk pc ^ f(r)).  Tile t+1 is issued
// right after the barrier that opens iteration t and is waited for (vmcnt(0)) just before the next barrier.
// raw v_max3_f32 / v_max_f32: fmaxf() makes the compiler canonicalise every operand it cannot prove quiet (an extra
// v_max x, x each); the scores are finite by construction
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_xor32_raw(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor16_raw(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Head dim D = 128 (InternViT-6B, Qwen2) or 64 (InternViT-300M): tile rows are 2*D bytes = CPR 16-B chunks, and the two
// swizzles become  K: chunk ^ (D == 128 ? row & 15 : (row >> 1) & 7)   V: chunk ^ (D == 128 ? (row & 7) << 1 : ((row >> 1) & 3) << 1)
// (a 128-B row spans 32 banks, so two consecutive rows already differ and the XOR acts on row >> 1).
template <int D> __device__ __forceinline__ int swz_k(int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); }
template <int D> __device__ __forceinline__ int swz_v(int row) { return D == 128 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1); }

template <typename T, int NW, int NQ, int D>
__global__ __launch_bounds__(NW * 64, 2) void attn_kernel(AttnP p) {
  typedef typename V8<T>::type frag_t;
  constexpr int NT = NW * 64;
  constexpr int RB = D * 2;                // bytes per K / V tile row
  constexpr int CPR = RB / 16;             // 16-B chunks per row
  constexpr int RPI = NT / CPR;            // rows covered by one instruction round of the workgroup
  constexpr int CH = KV_TILE / RPI;        // LDS-DMA instructions per thread per tile (K and V each)
  constexpr int DS = D / 32;               // 32-wide d steps of S^T = K Q^T
  constexpr int DN = D / 16;               // 16-row tiles of O^T
  constexpr int BUF = 2 * KV_TILE * RB;
  __shared__ __attribute__((aligned(256))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fc = lane & 15, fg = lane >> 4;

  const int n_rep = p.q_heads / p.kv_heads;
  const int nqb = gridDim.x;
  const int qb = p.causal ? nqb - 1 - (int)blockIdx.x : (int)blockIdx.x;     // heaviest causal blocks first
  const int hq0 = blockIdx.y, kvh = hq0 / n_rep, b = blockIdx.z;
  const int q0 = qb * (NW * NQ * 16);
  const int kv_len = p.kv_len ? p.kv_len[b] : p.Skv;
  int kmax = kv_len;
  if (p.causal) { const int lim = q0 + NW * NQ * 16 + p.q_pos0; kmax = lim < kmax ? lim : kmax; }
  const int t_end = (kmax + KV_TILE - 1) / KV_TILE;
  const int kv_start = p.kv_start ? p.kv_start[b] : 0;      // left-padded batch: the first kv_start keys are padding
  const int t_begin = kv_start / KV_TILE;

  const T* Kg = (const T*)p.K + b * p.k_sb + kvh * p.k_sh;
  const T* Vg = (const T*)p.V + b * p.v_sb + kvh * p.v_sh;

  // ---- Q fragments (B operand of S^T): lane holds Q[query fc][d = 32*ds + 8*fg + j]
  frag_t qf[NQ][DS];
  int qrow[NQ];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    qrow[qt] = q0 + (wave * NQ + qt) * 16 + fc;
    const int rr = qrow[qt] < p.Sq ? qrow[qt] : p.Sq - 1;
    const T* qp = (const T*)p.Q + b * p.q_sb + hq0 * p.q_sh + rr * p.q_sr;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[qt][ds] = ld8<T>(qp + ds * 32 + fg * 8);
  }

  f32x4 o[NQ][DN];
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    m_run[qt] = M_FLOOR; l_run[qt] = 0.f;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) o[qt][dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging: instruction i of this thread fills LDS (row = i*RPI + tid/16, physical chunk tid%16).  RPI is a multiple
  // of 16, so the swizzled source chunk is the same for every i: one base pointer per operand + wave-uniform row offsets
  static_assert(RPI % 16 == 0 && KV_TILE % RPI == 0, "staging rows per round must keep the swizzle bits of the row");
  const int srow = tid / CPR, spc = tid % CPR;
  const T* const kbase = Kg + (int64_t)srow * p.k_sr + ((spc ^ swz_k<D>(srow)) << 3);
  const T* const vbase = Vg + (int64_t)srow * p.v_sr + ((spc ^ swz_v<D>(srow)) << 3);
  auto issue_tile = [&](int t, int buf) {
    char* const Kw = smem + buf * BUF;
    char* const Vw = Kw + KV_TILE * RB;
    if ((t + 1) * KV_TILE <= kv_len) {                                      // full tile (uniform): scalar row offsets
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int64_t r0 = t * KV_TILE + i * RPI;
        __builtin_amdgcn_global_load_lds((gptr_t)(kbase + r0 * p.k_sr), (lptr_t)(Kw + (i * NT + wave * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vbase + r0 * p.v_sr), (lptr_t)(Vw + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    } else {                                                                // ragged last tile: clamp the row per lane
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        int kr = t * KV_TILE + i * RPI + srow; kr = (kr < kv_len ? kr : kv_len - 1) - srow;     // masked keys must stay finite
        __builtin_amdgcn_global_load_lds((gptr_t)(kbase + (int64_t)kr * p.k_sr), (lptr_t)(Kw + (i * NT + wave * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vbase + (int64_t)kr * p.v_sr), (lptr_t)(Vw + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
  };

  // per-lane LDS read offsets
  // K operand: row = 16*kt + fc, chunk = 4*ds + fg -> phys = chunk ^ fc
  // V^T operand (transposed read): lane fc = 4*tq + tp supplies row 32*ks + 4*fg + tq (+16), chunk 2*dn + (tp>>1), +8*(tp&1)
  const int tq = fc >> 2, tp = fc & 3;
  const int vrow_lo = 4 * fg + tq;                       // (+32*ks, +16 for the second read)
  const int vswz = swz_v<D>(vrow_lo);                    // rows +16 / +32 keep the swizzle

  const bool wave_active = q0 + wave * NQ * 16 < p.Sq;        // wave-uniform
  // tile t lives in buffer t & 1.  Iteration t: wait for own DMA of tile t, barrier (tile t visible to everyone, buffer
  // (t+1)&1 no longer read by anyone), issue tile t+1, then S^T, softmax and PV from buffer t & 1.
  if (t_end > t_begin) issue_tile(t_begin, t_begin & 1);
  for (int t = t_begin; t < t_end; ++t) {
    const int cur = t & 1;
    const char* const Ks = smem + cur * BUF;
    const char* const Vs = Ks + KV_TILE * RB;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < t_end) issue_tile(t + 1, cur ^ 1);
    // a wave whose 16 * NQ queries all lie beyond Sq (the ViT's 1025 = 8 x 128 + 1 rows leave three such waves in the last query
    // block) keeps staging and the barriers but skips the arithmetic
    if (!wave_active) continue;

    // ---- S^T = K Q^T
    f32x4 s[NQ][4];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + (kt * 16 + fc) * RB + (((ds * 4 + fg) ^ swz_k<D>(fc)) << 4));
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) s[qt][kt] = mfma16(kf, qf[qt][ds], s[qt][kt]);
      }

    // ---- mask + online softmax (per lane = per query column).  exp2((s - m) * c) = exp2(s * c - m * c).
    const int key0 = t * KV_TILE + 4 * fg;
    // masking is needed only on the last kv tile (ragged tail) and on tiles that reach the causal diagonal of this block
    const bool need_mask = (t + 1) * KV_TILE > kv_len || (p.causal && (t + 1) * KV_TILE > q0 + p.q_pos0 + 1) || t * KV_TILE < kv_start;
    frag_t pf[NQ][2];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
      if (need_mask) {
        int lim = kv_len;                                   // keys < lim are visible
        if (p.causal) { const int cl = qrow[qt] + p.q_pos0 + 1; lim = cl < lim ? cl : lim; }
        const int rel = lim - key0, rel_lo = kv_start - key0;     // register r of key tile kt is key key0 + 16*kt + r
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qt][kt][r] = (kt * 16 + r < rel && kt * 16 + r >= rel_lo) ? s[qt][kt][r] : NEG_BIG;
      }
      // VALU is the bound of this kernel (MFMA and VALU issue do not overlap on a SIMD: tools/experiments/tune_pipes.hip), so the
      // softmax is written for instruction count: v_max3 chain, v_pk_fma / v_pk_add on register pairs
      float mx = max3(s[qt][0][0], s[qt][0][1], s[qt][0][2]);
      mx = max3(mx, s[qt][0][3], s[qt][1][0]);
      mx = max3(mx, s[qt][1][1], s[qt][1][2]);
      mx = max3(mx, s[qt][1][3], s[qt][2][0]);
      mx = max3(mx, s[qt][2][1], s[qt][2][2]);
      mx = max3(mx, s[qt][2][3], s[qt][3][0]);
      mx = max3(mx, s[qt][3][1], s[qt][3][2]);
      mx = max2(mx, s[qt][3][3]);
      mx = max_xor32_raw(max_xor16_raw(mx));
      // lazy rescale: the reference point m_run only has to keep exp2() in range, not to be the exact running max, so it
      // moves only when some query of the wave exceeds it by more than 2^RESCALE_LOG2 (p <= 2^8: exact in fp32 sums,
      // same relative rounding in the 16-bit P operand); the final O / l is the same quantity either way
      if (__any((mx - m_run[qt]) * p.c > RESCALE_LOG2)) {
        const float m_new = max2(m_run[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * p.c);
        m_run[qt] = m_new;
        l_run[qt] *= alpha;
#pragma unroll
        for (int dn = 0; dn < DN; ++dn) o[qt][dn] *= alpha;
      }
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float nmc = -m_run[qt] * p.c;
      const f32x2 c2 = {p.c, p.c}, nmc2 = {nmc, nmc};
      f32x2 psum2 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        f32x8 e;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const f32x4 sv = s[qt][2 * ks + (h >> 1)];
          const f32x2 x = (h & 1) ? __builtin_shufflevector(sv, sv, 2, 3) : __builtin_shufflevector(sv, sv, 0, 1);
          const f32x2 y = __builtin_elementwise_fma(x, c2, nmc2);
          f32x2 ex;
          ex[0] = __builtin_amdgcn_exp2f(y[0]); ex[1] = __builtin_amdgcn_exp2f(y[1]);
          psum2 += ex;
          e[2 * h] = ex[0]; e[2 * h + 1] = ex[1];
        }
        pf[qt][ks] = __builtin_convertvector(e, frag_t);     // packed f32 -> 16-bit converts
      }
      l_run[qt] += psum2[0] + psum2[1];
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        const int ch = (2 * dn + (tp >> 1)) ^ vswz;
        const char* a0 = Vs + (ks * 32 + vrow_lo) * RB + (ch << 4) + 8 * (tp & 1);
        const s16x4 lo = tr_read(a0);
        const s16x4 hi = tr_read(a0 + 16 * RB);
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        const frag_t vf = __builtin_bit_cast(frag_t, cat);
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) o[qt][dn] = mfma16(vf, pf[qt][ks], o[qt][dn]);
      }
  }

  // ---- finalize.  o[qt][dn][r] = O^T[d = 16*dn + 4*fg + r][query fc]
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    const float l = sum_xor32(sum_xor16(l_run[qt]));
    if (qrow[qt] < p.Sq) {
      const float inv = l > 0.f ? 1.f / l : 0.f;
      T* op = (T*)p.O + b * p.o_sb + hq0 * p.o_sh + qrow[qt] * p.o_sr + fg * 4;
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        typename V8<T>::half_type h4;
#pragma unroll
        for (int r = 0; r < 4; ++r) h4[r] = fromf<T>(o[qt][dn][r] * inv);
        *reinterpret_cast<typename V8<T>::half_type*>(op + dn * 16) = h4;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Prefill kernel, second generation (head_dim 128): 32x32x16 MFMA, one wave = 32 queries of one head.
//
// Why: the 16x16x32 kernel above is VALU-issue bound (a 16x16x32 MFMA blocks the SIMD's vector issue for 8 of its 16 cycles, a
// 32x32x16 for 8 of its 32: MI355X_MICROARCH.md "vector-instruction ISSUE cost"), and every q head re-staged the K / V tiles of its
// kv head.  Here
//   * S^T[32 keys][32 q] = K Q^T with 32x32x16: the lane holds 16 scores of ONE query (its column), the partner lane (l ^ 32) the other
//     16 keys of the 32-key tile: the row max / sum need one v_permlane32_swap each, and half as many MFMA issue slots per FLOP go to
//     the softmax's VALU work;
//   * P^T feeds the PV MFMA as its B operand straight from the score registers (cdna_hip_programming.md §3 "An accumulator tile as the
//     next MFMA's operand": registers 8s..8s+7 are k-step s, element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)); V^T comes
//     through ds_read_b64_tr_b16 in that same key order;
//   * GQA (decoder prefill): the workgroup is the n_rep (7) query heads of ONE kv head x 32 queries, so a K / V tile is staged once
//     for all of them; MHA (ViT): 4 waves = 128 queries of one head.
// K and V tiles (64 keys x 256 B) share ONE LDS image formula, T10 image (b): off(row, ch) = 256 row + 16 (ch ^ (((row & 3) << 2) |
// ((row >> 2) & 3))), conflict-free for the 32-row ds_read_b128 operand read AND for the transposed reads; filled by LDS-DMA with the
// permutation on the source chunk.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int swz_b(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// LDS images of the second-generation kernel.  Head dim 128: 256-byte rows, K and V share image (b) above.  Head dim 64 (InternViT-300M,
// round 3): 128-byte rows, two key rows per 256-byte bank row R = row >> 1, slot = ((row & 1) << 3 | chunk) ^ f(R) with
//   K: f = R & 7          (the 32-row ds_read_b128 operand read: a 16-lane group holds 8 distinct R whose low 3 bits differ, and the two
//                          rows of one R differ in slot bit 3)
//   V: f = (R & 1) << 2   (a transposed read's 32-lane half takes rows r0 .. r0 + 3 (two R) x 4 consecutive chunks: bit 3 = row parity,
//                          bit 2 = R parity, low bits = chunk: 16 distinct slots)
// tile_off<D, IS_V>(row, ch) = byte offset of 16-byte chunk ch of key row `row`; tile_src<D, IS_V>(slot_index) = (row, chunk) stored at a
// lane-linear LDS slot (the LDS-DMA destination is lane-linear, so the permutation is applied to the SOURCE address).
template <int D, bool IS_V> __device__ __forceinline__ int tile_off(int row, int ch) {
  if constexpr (D == 128) return 256 * row + 16 * (ch ^ swz_b(row));
  const int R = row >> 1, f = IS_V ? ((R & 1) << 2) : (R & 7);
  return 256 * R + 16 * ((((row & 1) << 3) | ch) ^ f);
}
template <int D, bool IS_V> __device__ __forceinline__ void tile_src(int piece, int& row, int& ch) {
  if constexpr (D == 128) { row = piece >> 4; ch = (piece & 15) ^ swz_b(row); return; }
  const int R = piece >> 4, f = IS_V ? ((R & 1) << 2) : (R & 7);
  const int vc = (piece & 15) ^ f;
  row = 2 * R + (vc >> 3); ch = vc & 7;
}

#ifndef OMCHAT_TR_ASM
#define OMCHAT_TR_ASM 0      // 1: the transposed V reads of the PV phase as inline assembly (A/B twin of round 4, measured 5-8 % slower: below)
#endif
// KG = 2 (MHA only, round 5): the workgroup is EIGHT waves on the same 128 queries -- two groups of four, each walking one half of the key tiles with
// its own two-stage K / V ring -- and the halves meet through LDS at the end (flash-decoding inside one workgroup: no partials in memory, no
// second launch).  Why: a wave's unit of work is 32 queries x ALL keys, the 3-tile ViT has 2475 such units for 2048 wave slots, so the launch
// lasts two units whatever the block shape; halved units are 4950 on 2048 slots = three halves (1.5 units).  The bits differ from KG = 1 only by
// where the online softmax is cut (one more rescale per query); launch_attn_prefill chooses per shape.
template <typename T, int NW, bool GQA, int D = 128, int KG = 1>
__global__ __launch_bounds__(NW * 64, 2) void attn2_kernel(AttnP p) {
  typedef typename V8<T>::type frag_t;
  static_assert(KG == 1 || (!GQA && NW == 8), "key groups: MHA, eight waves");
  constexpr int NT = NW * 64 / KG;                    // threads that stage one tile (a key group)
  constexpr int RB = 2 * D;                           // bytes per key row
  constexpr int TILE = KV_TILE * RB;                  // bytes of one K (or V) tile
  constexpr int BUF = 2 * TILE;
  constexpr int NPIECE = TILE / 16;                   // 16-byte pieces per tile
  constexpr int ROUNDS = (NPIECE + NT - 1) / NT;
  constexpr int KS = D / 16;                          // k-steps of S^T = K Q^T
  constexpr int DB = D / 32;                          // 32-row blocks of O^T
  extern __shared__ __attribute__((aligned(256))) char smem_all[];      // KG x 2 x BUF bytes (dynamic: 128 KB at D = 128, KG = 2)

  const int tid = threadIdx.x % NT, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int kg = KG == 1 ? 0 : wave_all / (NW / KG);                    // key group of this wave
  const int wave = KG == 1 ? wave_all : wave_all % (NW / KG);           // wave inside its group
  char* const smem = smem_all + kg * 2 * BUF;
  const int qc = lane & 31, hh = lane >> 5;           // query column of this lane, lane half

  const int n_rep = p.q_heads / p.kv_heads;
  // GQA: the kv head is the FASTEST index of the workgroup id, so that the dispatcher hands out the causal query blocks heaviest first across
  // ALL kv heads (round 4).  With the kv head on blockIdx.y the 448 workgroups of a single S = 3584 sequence were dealt head by head: the
  // heaviest blocks of the last head (56 key tiles) started only after a CU had finished one of the light blocks of the first heads -- the
  // launch ended at ~66 tile steps where 56 (the heaviest block alone) are the bound.  It also keeps one kv head per XCD (id % 8 -> id % 4).
  // (the sequence is the next index: id = (block rank * batch + sequence) * kv_heads + kv head; p.nsplit carries the batch size)
  // Head split (round 4, p.tpw = xs): when the launch is at most one workgroup per CU its length is the heaviest block's, not the mean.  The
  // xs heaviest block ranks are then issued as TWO workgroups each, one running the first half of the kv group's query heads and one the
  // rest (the other waves stay idle but still stage their share of every tile): no partial results, no merge.  Ids [0, 2 xs per_rank) are
  // those pairs, the other ranks follow in order.
  const int nb = GQA ? p.nsplit : 1;
  const int per_rank = GQA ? p.kv_heads * nb : 1;
  const int xs = GQA ? p.tpw : 0;
  const int nqb = GQA ? (int)gridDim.x / per_rank - xs : (p.tpw == -1 ? (p.nsplit & 0xffff) : (int)gridDim.x);
  int h_lo = 0, h_hi = GQA ? n_rep : 1, bx, rest;
  if (GQA && (int)blockIdx.x < 2 * xs * per_rank) {
    const int pair = (int)blockIdx.x >> 1;
    bx = pair / per_rank; rest = pair % per_rank;
    if (blockIdx.x & 1) h_lo = (n_rep + 1) / 2; else h_hi = (n_rep + 1) / 2;
  } else if (GQA) {
    const int j = (int)blockIdx.x - 2 * xs * per_rank;
    bx = xs + j / per_rank; rest = j % per_rank;
  } else {
    bx = (int)blockIdx.x; rest = 0;
  }
  // MHA (the ViT), p.tpw = -1: one-dimensional grid in which the query blocks of a (head, sequence) pair sit 8 ids apart, i.e. on ONE XCD under
  // round-robin placement (round 4).  With the query block as blockIdx.x the nine blocks of a head were dealt to all eight XCDs and every
  // XCD's L2 fetched that head's K / V for itself: 355 MB per launch by the FETCH_SIZE counter for 79 MB of q, k, v, o (profiles/r04_ah_pmc...).
  int mha_head = (int)blockIdx.y, mha_b = (int)blockIdx.z;
  if (!GQA && p.tpw == -1) {
    const int nq = p.nsplit & 0xffff, nbat = p.nsplit >> 16, grp = (int)blockIdx.x / (8 * nq), r = (int)blockIdx.x % (8 * nq);
    const int pair = grp * 8 + (r & 7);
    bx = r >> 3;
    if (pair >= p.q_heads * nbat) return;      // (the last group of eight is padded; p.nsplit = query blocks | batch << 16)
    mha_head = pair % p.q_heads; mha_b = pair / p.q_heads;
  }
  const int qb = p.causal ? nqb - 1 - bx : bx;        // heaviest causal blocks first
  const int b = GQA ? rest / p.kv_heads : mha_b;
  constexpr int QW = GQA ? 32 : (NW / KG) * 32;       // queries of the workgroup
  const int kvh = GQA ? rest % p.kv_heads : mha_head / n_rep;
  const int hq = GQA ? kvh * n_rep + wave : mha_head;
  const int q0b = qb * QW;
  const int q0 = q0b + (GQA ? 0 : wave * 32);
  const int kv_len = p.kv_len ? p.kv_len[b] : p.Skv;
  const int kv_start = p.kv_start ? p.kv_start[b] : 0;
  // One key over a whole number of tiles (the ViT: 1025 = 16 x 64 + 1 keys): the odd key does not get a 17th tile step of its own -- 64-key MFMAs, a
  // softmax pass and a barrier for ONE key, 5.9 % of the launch -- it INITIALISES the online softmax instead: m = q . k_last, l = 1, O = v_last (its
  // probability is exp2(0) = 1 exactly, in the accumulator as in the 16-bit P operand), and the loop walks the whole tiles only (round 6, p.peel_last).
  const bool peel = !GQA && KG == 1 && D == 128 && p.peel_last && !p.causal && kv_start == 0 && kv_len > KV_TILE && (kv_len & (KV_TILE - 1)) == 1;
  int kmax = peel ? kv_len - 1 : kv_len;
  if (p.causal) { const int lim = q0b + QW + p.q_pos0; kmax = lim < kmax ? lim : kmax; }
  const int t_end = (kmax + KV_TILE - 1) / KV_TILE;
  const int t_begin = kv_start / KV_TILE;

  const T* Kg = (const T*)p.K + b * p.k_sb + kvh * p.k_sh;
  const T* Vg = (const T*)p.V + b * p.v_sb + kvh * p.v_sh;
  const bool wave_active = (GQA ? (wave >= h_lo && wave < h_hi) : true) && q0 < p.Sq;       // wave-uniform

  // ---- Q fragments (B operand of S^T): lane holds Q[query qc][d = 16 s + 8 hh + j]
  frag_t qf[KS];
  const int qrow = q0 + qc;
  {
    const int rr = qrow < p.Sq ? qrow : p.Sq - 1;
    const T* qp = (const T*)p.Q + b * p.q_sb + (wave_active ? hq : 0) * p.q_sh + (int64_t)rr * p.q_sr;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = ld8<T>(qp + s * 16 + hh * 8);
  }
  // q norm on load (MHA, round 6): the row's finished sum of squares and this lane's slice of w_q are requested with Q
  [[maybe_unused]] float qn_ss = 0.f;
  [[maybe_unused]] frag_t qn_wv[(!GQA && D == 128) ? KS : 1];
  if constexpr (!GQA && D == 128) {
    if (p.qn_sumsq) {
      const int rr = qrow < p.Sq ? qrow : p.Sq - 1;
      qn_ss = p.qn_sumsq[((size_t)b * p.Sq + rr) * p.qn_stride];
      const T* wq = (const T*)p.qn_w + (size_t)hq * D;
#pragma unroll
      for (int s = 0; s < KS; ++s) qn_wv[s] = ld8<T>(wq + s * 16 + hh * 8);
    }
  }

  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = M_FLOOR, l_run = 0.f;

  // ---- staging: piece = i * NT + tid -> lane-linear LDS slot, permuted source (tile_src)
  // A whole tile (every key < kv_len) is addressed as  wave-uniform tile base + a 32-bit lane offset computed once  (the per-tile 64-bit
  // row * stride products were ~60 of the ~200 VALU instructions a key tile costs, and VALU time adds to MFMA time here: profiles/
  // r02_b_pmc_attn.txt); only the last, partial tile clamps its rows (masked keys must stay finite) the long way.
  unsigned koff[ROUNDS], voff[ROUNDS];
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    int row, ch;
    tile_src<D, false>(i * NT + tid, row, ch);
    koff[i] = (unsigned)row * (unsigned)(p.k_sr * 2) + ch * 16;
    tile_src<D, true>(i * NT + tid, row, ch);
    voff[i] = (unsigned)row * (unsigned)(p.v_sr * 2) + ch * 16;
  }
  auto issue_tile = [&](int t, int buf) {
    char* const Kw = smem + buf * BUF;
    char* const Vw = Kw + TILE;
    if ((t + 1) * KV_TILE <= kv_len) {                                     // whole tile (uniform)
      const char* Kt = (const char*)(Kg + (int64_t)t * KV_TILE * p.k_sr);
      const char* Vt = (const char*)(Vg + (int64_t)t * KV_TILE * p.v_sr);
#pragma unroll
      for (int i = 0; i < ROUNDS; ++i) {
        if ((i + 1) * NT <= NPIECE || i * NT + wave * 64 < NPIECE) {       // whole waves only (wave-uniform)
          __builtin_amdgcn_global_load_lds((gptr_t)(Kt + koff[i]), (lptr_t)(Kw + (i * NT + wave * 64) * 16), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gptr_t)(Vt + voff[i]), (lptr_t)(Vw + (i * NT + wave * 64) * 16), 16, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      if ((i + 1) * NT <= NPIECE || i * NT + wave * 64 < NPIECE) {
        int row, ch;
        tile_src<D, false>(i * NT + tid, row, ch);
        int kr = t * KV_TILE + row; kr = kr < kv_len ? kr : kv_len - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(Kg + (int64_t)kr * p.k_sr + ch * 8), (lptr_t)(Kw + (i * NT + wave * 64) * 16), 16, 0, 0);
        tile_src<D, true>(i * NT + tid, row, ch);
        kr = t * KV_TILE + row; kr = kr < kv_len ? kr : kv_len - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(Vg + (int64_t)kr * p.v_sr + ch * 8), (lptr_t)(Vw + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
  };

  // ---- per-lane LDS read offsets
  // K (A operand of S^T), k-step s of key tile kt: row 32 kt + qc, chunk 2 s + hh
  // V^T (A operand of PV), k-step s2 of key tile kt, d block db: 16-lane group g = lane >> 4 (half = g >> 1, d sub-block = g & 1);
  //   lane 4 q + pp of the group supplies row r0 + q, chunk c0 + (pp >> 1), + 8 (pp & 1) bytes, r0 = 32 kt + 16 s2 + 4 half (+ 8), c0 = 4 db + 2 dsub
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg = lane >> 4;
  const int v_row_lo = 4 * (vg >> 1) + vq;            // + 32 kt + 16 s2 (+ 8 for the second read)
  const int v_ch_lo = 2 * (vg & 1) + (vp >> 1);       // + 4 db
  const int v_byte = 8 * (vp & 1);

  // A deeper ring does not help (round 4): a three-stage form of this loop (tile t + 2 in flight, counted vmcnt, transposed reads as inline
  // assembly so that the compiler's vmcnt(0) does not drain it) gave the same bits 10 % SLOWER on every shape (S = 3584: 116.7 vs 105.5 us,
  // 33 k keys 949 vs 1044 TF): the step is bound by the MFMA + softmax chain of the waves on a SIMD (~0.85 of the MFMA pipe busy at two
  // waves per SIMD), not by the tile's arrival.
  // key group kg walks tiles [g_begin, g_end); both groups take the same number of steps (the barriers are the workgroup's)
  const int n_tiles = t_end > t_begin ? t_end - t_begin : 0;
  const int steps = KG == 1 ? n_tiles : (n_tiles + 1) / 2;
  const int g_begin = KG == 1 ? t_begin : t_begin + kg * steps;
  const int g_end = KG == 1 ? t_end : (kg == 0 ? t_begin + steps : t_end);
  if (g_end > g_begin) issue_tile(g_begin, 0);
  // (the q norm's arithmetic sits HERE, behind the first tile's LDS-DMA: its statistic / weight loads were requested with Q, in front of the DMA, so
  // waiting for them does not wait for the tile, and its ~400 VALU instructions run while the tile is in flight)
  if constexpr (!GQA && D == 128) {
    if (p.qn_sumsq) {
      // InternAttention's q norm (modeling_intern_vit.py:143-148) on the fragments just loaded, with vit_qknorm_kernel's rounding points:
      // T(T(w_q * T(q * rstd)) * scale)
      const float inv = rsqrtf(qn_ss / (float)p.qn_dim + p.qn_eps);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        frag_t nq;
#pragma unroll
        for (int j = 0; j < 8; ++j) nq[j] = fromf<T>(rnd<T>(tof(qn_wv[s][j]) * rnd<T>(tof(qf[s][j]) * inv)) * p.qn_scale);
        qf[s] = nq;
      }
    }
  }
  if constexpr (!GQA && KG == 1 && D == 128) {
    if (peel && wave_active) {
      const T* kl = Kg + (int64_t)(kv_len - 1) * p.k_sr + hh * 8;
      const T* vl = Vg + (int64_t)(kv_len - 1) * p.v_sr + 4 * hh;
      float sd = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const frag_t kf = ld8<T>(kl + s * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) sd = __builtin_fmaf(tof(qf[s][j]), tof(kf[j]), sd);
      }
      sd += __shfl_xor(sd, 32);                       // the query's two lanes hold the two halves of every 16-deep k-step
      m_run = sd;
      l_run = hh == 0 ? 1.f : 0.f;                    // (the final sum adds the two lanes)
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const typename V8<T>::half_type v4 = *reinterpret_cast<const typename V8<T>::half_type*>(vl + db * 32 + g * 8);      // O^T rows d = 32 db + 8 g + 4 hh + (0..3)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[db][4 * g + r] = tof(v4[r]);
        }
    }
  }
  for (int it = 0; it < steps; ++it) {
    const int t = g_begin + it;
    const int cur = it & 1;
    const char* const Ks = smem + cur * BUF;
    const char* const Vs = Ks + TILE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < g_end) issue_tile(t + 1, cur ^ 1);
    if (!wave_active || t >= g_end) continue;

    // ---- S^T = K Q^T: two 32-key tiles
    f32x16 sc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
      const int row = kt * 32 + qc;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + tile_off<D, false>(row, 2 * s + hh));
        sc[kt] = mfma32(kf, qf[s], sc[kt]);
      }
    }

    // ---- mask + online softmax (this lane: query qc, keys 32 kt + (r & 3) + 8 (r >> 2) + 4 hh)
    const int key0 = t * KV_TILE + 4 * hh;
    const bool need_mask = (t + 1) * KV_TILE > kv_len || (p.causal && (t + 1) * KV_TILE > q0 + p.q_pos0 + 1) || t * KV_TILE < kv_start;
    if (need_mask) {
      int lim = kv_len;
      if (p.causal) { const int cl = qrow + p.q_pos0 + 1; lim = cl < lim ? cl : lim; }
      const int rel = lim - key0, rel_lo = kv_start - key0;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kt * 32 + (r & 3) + 8 * (r >> 2);
          sc[kt][r] = (kk < rel && kk >= rel_lo) ? sc[kt][r] : NEG_BIG;
        }
    }
    float mx = max3(sc[0][0], sc[0][1], sc[0][2]);
#pragma unroll
    for (int r = 3; r + 1 < 16; r += 2) mx = max3(mx, sc[0][r], sc[0][r + 1]);
    mx = max3(mx, sc[0][15], sc[1][0]);
#pragma unroll
    for (int r = 1; r + 1 < 16; r += 2) mx = max3(mx, sc[1][r], sc[1][r + 1]);
    mx = max2(mx, sc[1][15]);
    mx = max_xor32_raw(mx);
    if (__any((mx - m_run) * p.c > RESCALE_LOG2)) {
      const float m_new = max2(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d) o[d] *= alpha;
    }
    const float nmc = -m_run * p.c;
    frag_t pf[2][2];
    // two scores per instruction (v_pk_fma_f32 / v_pk_add_f32): the softmax's VALU instructions add to the MFMA time on this SIMD
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 c2 = {p.c, p.c}, nmc2 = {nmc, nmc};
    f32x2 psum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        f32x8 e;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const f32x2 sv = {sc[kt][8 * s2 + j], sc[kt][8 * s2 + j + 1]};
          const f32x2 x = __builtin_elementwise_fma(sv, c2, nmc2);
          const f32x2 ev = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          e[j] = ev[0]; e[j + 1] = ev[1];
          psum2 += ev;
        }
        pf[kt][s2] = __builtin_convertvector(e, frag_t);
      }
    l_run += psum2[0] + psum2[1];

    // ---- O^T += V^T P^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int r_lo = kt * 32 + s2 * 16 + v_row_lo, r_hi = r_lo + 8;
        // hipcc puts an s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 it emits while an LDS-DMA is outstanding (the intrinsic has
        // no memory operand to tell the stage being read from the stage being filled), so the PV phase of tile t formally waits for the prefetch
        // of tile t + 1 issued at the top of the iteration.  Round 4 tested whether that wait costs anything: the same reads as inline assembly
        // (OMCHAT_TR_ASM = 1: four reads + their own lgkmcnt wait per pair of d blocks, no vmcnt wait left in the loop but the one at its
        // top) run 5-8 % SLOWER on every shape (ViT 3 tiles 78.2 vs 72.3 us, causal S = 3584 164.4 vs 153.4 us, 33 k keys 973 vs 1018 TF:
        // profiles/r04_y): the K / V tiles come from L2 and have landed by the time the softmax is done, and the compiler's own
        // read-by-read lgkmcnt interleave with the MFMAs is worth more than the removed wait.  (In the decode ring kernel, whose tiles
        // come from HBM, the same wait drained the ring and the assembly form is the one that works.)
        if constexpr (OMCHAT_TR_ASM != 0) {
#pragma unroll
        for (int db = 0; db < DB; db += 2) {
          s16x4 lo0, hi0, lo1, hi1;
          const unsigned a0 = (unsigned)(size_t)(lds_s16x4_ptr)(Vs + tile_off<D, true>(r_lo, 4 * db + v_ch_lo) + v_byte);
          const unsigned a1 = (unsigned)(size_t)(lds_s16x4_ptr)(Vs + tile_off<D, true>(r_hi, 4 * db + v_ch_lo) + v_byte);
          const unsigned a2 = (unsigned)(size_t)(lds_s16x4_ptr)(Vs + tile_off<D, true>(r_lo, 4 * (db + 1) + v_ch_lo) + v_byte);
          const unsigned a3 = (unsigned)(size_t)(lds_s16x4_ptr)(Vs + tile_off<D, true>(r_hi, 4 * (db + 1) + v_ch_lo) + v_byte);
          asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(lo0), "=&v"(hi0), "=&v"(lo1), "=&v"(hi1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 c0 = __builtin_shufflevector(lo0, hi0, 0, 1, 2, 3, 4, 5, 6, 7);
          const s16x8 c1 = __builtin_shufflevector(lo1, hi1, 0, 1, 2, 3, 4, 5, 6, 7);
          o[db] = mfma32(__builtin_bit_cast(frag_t, c0), pf[kt][s2], o[db]);
          o[db + 1] = mfma32(__builtin_bit_cast(frag_t, c1), pf[kt][s2], o[db + 1]);
        }
        } else {
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const s16x4 lo = tr_read(Vs + tile_off<D, true>(r_lo, 4 * db + v_ch_lo) + v_byte);
          const s16x4 hi = tr_read(Vs + tile_off<D, true>(r_hi, 4 * db + v_ch_lo) + v_byte);
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[db] = mfma32(__builtin_bit_cast(frag_t, cat), pf[kt][s2], o[db]);
        }
        }
      }
  }

  if constexpr (KG == 2) {
    // the second key group hands its running state to the first through LDS (the rings are dead: every read is behind the barrier)
    constexpr int ST = DB * 16 + 2;
    float* const mg = reinterpret_cast<float*>(smem_all) + (size_t)(wave * 64 + lane) * ST;
    __syncthreads();
    if (kg == 1) {
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) mg[d * 16 + r] = o[d][r];
      mg[DB * 16] = m_run; mg[DB * 16 + 1] = l_run;
    }
    __syncthreads();
    if (kg == 1) return;
    const float m1 = mg[DB * 16], l1 = mg[DB * 16 + 1];
    const float m = max2(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f((m_run - m) * p.c), a1 = __builtin_amdgcn_exp2f((m1 - m) * p.c);
    l_run = l_run * a0 + l1 * a1;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = o[d][r] * a0 + mg[d * 16 + r] * a1;
  }
  // ---- finalize.  o[db][r] = O^T[d = 32 db + (r & 3) + 8 (r >> 2) + 4 hh][query qc]
  const float l = sum_xor32(l_run);
  if (wave_active && qrow < p.Sq) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    T* op = (T*)p.O + b * p.o_sb + hq * p.o_sh + (int64_t)qrow * p.o_sr + 4 * hh;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename V8<T>::half_type h4;
#pragma unroll
        for (int r = 0; r < 4; ++r) h4[r] = fromf<T>(o[db][4 * g + r] * inv);
        *reinterpret_cast<typename V8<T>::half_type*>(op + db * 32 + g * 8) = h4;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Decode attention, one wave per (64-key split, kv head, sequence).  Same transposed MFMA formulation as the prefill kernel
// with the n_rep query heads of the kv group as the 16 "queries", but latency-shaped: ALL global loads of the tile (K as
// MFMA A fragments straight to registers, V for the LDS transpose image, Q, RoPE table row) are issued at once -- one HBM
// round trip -- then S^T from registers, softmax, V -> LDS, PV through ds_read_b64_tr_b16.  With `rope` set it also
// rotates q in registers and, in the split that owns the new position, rotates k and appends k / v to the cache
// (replaces the separate RoPE + KV-append launch for S = 1).
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool KV8 = false, bool MASKED = false>
__global__ __launch_bounds__(64, 2) void attn_decode_kernel(AttnP p) {
  __shared__ __attribute__((aligned(256))) char Vs[KV_TILE * 256];
  const int lane = threadIdx.x, fc = lane & 15, fg = lane >> 4;
  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int n_rep = p.q_heads / p.kv_heads, hq0 = kvh * n_rep;
  const int kv_len = p.kv_len ? p.kv_len[b] : p.Skv;
#if OMCHAT_EXPERIMENTS
  if (p.dbg && split < 8 && kvh == 0 && lane == 0) atomicMax(p.dbg + 10, ~wall_clock64());      // measurement: first start (stored inverted)
#endif
  float* wsb = p.ws + ((size_t)(b * p.q_heads + hq0 + (fc < n_rep ? fc : 0)) * p.nsplit + split) * WS_STRIDE;
  if (split * KV_TILE >= kv_len) {            // empty split (uniform): neutral partial
    if (fc < n_rep && fg == 0) { wsb[128] = NEG_BIG; wsb[129] = 0.f; }
    return;
  }
  f32x4 o[8];
  float mx, l;
  attn_decode_tile<T, KV8, false, MASKED>(p, split, kvh, b, kv_len, Vs, lane, o, mx, l);
  if (fc < n_rep) {
#pragma unroll
    for (int dn = 0; dn < 8; ++dn) *reinterpret_cast<f32x4*>(wsb + dn * 16 + fg * 4) = o[dn];
    if (fg == 0) { wsb[128] = mx; wsb[129] = l; }
  }
#if OMCHAT_EXPERIMENTS
  if (p.dbg && kvh == 0 && lane == 0) atomicMax(p.dbg + 11, wall_clock64());
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Decode attention for LARGE grids (batched decode): one wave walks p.tpw consecutive 64-key tiles of its (kv head, sequence) with a
// running max / sum, so the partials that are written here and read back by the merge (3.6 KB per 32 KB of K / V with one tile per
// wave) shrink by that factor.  The one-tile kernel above keeps every load of its tile in flight at once (a single sequence has few
// waves: latency is everything); here thousands of waves hide each other's latency, so a tile's K and V loads share ONE register
// block -- K, S^T = K Q^T, then V into the same registers while the softmax runs -- which keeps the running output tile resident
// without spilling.  16-bit cache, fused RoPE + append as in the one-tile kernel.
// ---------------------------------------------------------------------------------------------------------
// KLDS (round 3): the K tile is loaded like the V tile -- whole 256-byte rows, 1 KiB contiguous per wave instruction -- and reaches the
// MFMA A-fragment order through the SAME 16 KiB LDS buffer the V image uses afterwards (K image: chunk' = chunk ^ (row & 15), conflict-free
// ds_read_b128 operand reads).  The register-direct form loads fragment-shaped K (16 rows x 64 B per instruction): half of this kernel's
// bytes then arrive as half cache lines per request, and the load path, not HBM, sets its 5 TB/s (cdna_hip_programming.md section 5,
// "x through LDS in full lines").  Costs 16 ds_write_b128 + 16 ds_read_b128 + one single-wave barrier per tile.
template <typename T, bool KLDS = false>
__global__ __launch_bounds__(64, 2) void attn_decode_multi_kernel(AttnP p) {
  typedef typename V8<T>::type frag_t;
  __shared__ __attribute__((aligned(256))) char Vs[KV_TILE * 256];
  const int lane = threadIdx.x, fc = lane & 15, fg = lane >> 4;
  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int n_rep = p.q_heads / p.kv_heads, hq0 = kvh * n_rep;
  const int kv_len = p.kv_len ? p.kv_len[b] : p.Skv;
  const int key_base = split * KV_TILE * p.tpw;
  float* wsb = p.ws + ((size_t)(b * p.q_heads + hq0 + (fc < n_rep ? fc : 0)) * p.nsplit + split) * WS_STRIDE;
  if (key_base >= kv_len) {                   // empty split (uniform): neutral partial
    if (fc < n_rep && fg == 0) { wsb[128] = NEG_BIG; wsb[129] = 0.f; }
    return;
  }
  const bool fuse = p.rope != nullptr;
  const int pp = kv_len - 1;
  const int pt = pp < p.rope_max ? pp : p.rope_max - 1;
  const T* Kg = (const T*)p.K + b * p.k_sb + kvh * p.k_sh;
  const T* Vg = (const T*)p.V + b * p.v_sb + kvh * p.v_sh;
  const T* kn = fuse ? (const T*)p.k_new + b * p.new_sb + kvh * 128 : nullptr;
  const T* vn = fuse ? (const T*)p.v_new + b * p.new_sb + kvh * 128 : nullptr;

  frag_t qf[4];
  {
    const int hh = fc < n_rep ? fc : n_rep - 1;
    const T* qp = (const T*)p.Q + b * p.q_sb + (hq0 + hh) * p.q_sh;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = ld8<T>(qp + ds * 32 + fg * 8);
    if (fuse) {
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const float* cs = p.rope + ((size_t)pt * 64 + ds * 32 + fg * 8) * 2;
        const frag_t lo = qf[ds], hi = qf[ds + 2];
        qf[ds] = rope_chunk<T>(lo, hi, cs, false);
        qf[ds + 2] = rope_chunk<T>(hi, lo, cs, true);
      }
    }
  }
  f32x4 o[8];
#pragma unroll
  for (int dn = 0; dn < 8; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const int tq = fc >> 2, tp = fc & 3;
  const int vrow_lo = 4 * fg + tq;
  const int vswz = ((vrow_lo & 7) << 1);

  for (int tt = 0; tt < p.tpw; ++tt) {
    const int key0 = key_base + tt * KV_TILE;
    if (key0 >= kv_len) break;                // uniform
    frag_t kv[16];                            // K fragments [kt][ds] first, then the V image rows of the same tile
    f32x4 s[4];
    if constexpr (KLDS) {
      // lane (fg, fc) loads chunk fc of key row key0 + 4 i + fg
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = key0 + i * 4 + fg;
        const bool fresh = fuse && key >= pp;
        const T* src = fresh ? kn : Kg + (int64_t)(key < kv_len ? key : kv_len - 1) * p.k_sr;
        kv[i] = ld8s<T>(src + fc * 8);
      }
      if (fuse && key0 + KV_TILE > pp) {      // only the tile that owns the new position (uniform): rotate the fresh row, append it
        const float* cs = p.rope + ((size_t)pt * 64 + (fc & 7) * 8) * 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (key0 + i * 4 + 3 >= pp) {         // uniform: this register holds a row >= pp in some lane group
            // rotate-half partner of chunk fc is chunk fc ^ 8 of the same row: lane ^ 8
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 own = __builtin_bit_cast(i32x4, kv[i]);
            i32x4 oth;
#pragma unroll
            for (int w = 0; w < 4; ++w) oth[w] = __shfl_xor(own[w], 8, 64);
            const frag_t rot = rope_chunk<T>(kv[i], __builtin_bit_cast(frag_t, oth), cs, fc >= 8);
            if (key0 + i * 4 + fg >= pp) kv[i] = rot;
            if (key0 + i * 4 + fg == pp)
              st8<T>((T*)p.k_cache_w + b * p.k_sb + kvh * p.k_sh + (int64_t)pp * p.k_sr + fc * 8, kv[i]);
          }
        }
      }
      if (tt > 0) __syncthreads();            // the previous tile's transposed V reads are done
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = i * 4 + fg;
        *reinterpret_cast<frag_t*>(Vs + row * 256 + ((fc ^ (row & 15)) << 4)) = kv[i];
      }
      __syncthreads();
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
          kv[kt * 4 + ds] = *reinterpret_cast<const frag_t*>(Vs + (kt * 16 + fc) * 256 + (((ds * 4 + fg) ^ fc) << 4));
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) s[kt] = mfma16(kv[kt * 4 + ds], qf[ds], s[kt]);
      }
    } else {
      bool kfresh[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int key = key0 + kt * 16 + fc;
        kfresh[kt] = fuse && key >= pp;
        const T* src = kfresh[kt] ? kn : Kg + (int64_t)(key < kv_len ? key : kv_len - 1) * p.k_sr;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) kv[kt * 4 + ds] = ld8s<T>(src + ds * 32 + fg * 8);
      }
      if (fuse && key0 + KV_TILE > pp) {      // only the tile that owns the new position (uniform)
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          const float* cs = p.rope + ((size_t)pt * 64 + ds * 32 + fg * 8) * 2;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            if (pp >= key0 + kt * 16 && pp < key0 + kt * 16 + 16) {      // the block that holds row pp (uniform); later blocks are masked
              const frag_t kl = kv[kt * 4 + ds], kh = kv[kt * 4 + ds + 2];
              const frag_t rl = rope_chunk<T>(kl, kh, cs, false), rh = rope_chunk<T>(kh, kl, cs, true);
              if (kfresh[kt]) { kv[kt * 4 + ds] = rl; kv[kt * 4 + ds + 2] = rh; }      // per lane: cached rows of the block are rotated already
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          if (key0 + kt * 16 + fc == pp) {
#pragma unroll
            for (int ds = 0; ds < 4; ++ds)
              st8<T>((T*)p.k_cache_w + b * p.k_sb + kvh * p.k_sh + (int64_t)pp * p.k_sr + ds * 32 + fg * 8, kv[kt * 4 + ds]);
          }
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) s[kt] = mfma16(kv[kt * 4 + ds], qf[ds], s[kt]);
      }
    }
    // ---- V of the same tile into the K registers; the softmax below runs under these loads
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = key0 + i * 4 + fg;
      const bool fresh = fuse && key >= pp;
      const T* src = fresh ? vn : Vg + (int64_t)(key < kv_len ? key : kv_len - 1) * p.v_sr;
      kv[i] = ld8s<T>(src + fc * 8);
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
    if (tt > 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
      l_run *= alpha;
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) o[dn] *= alpha;
      mx = m_new;
    }
    m_run = mx;
    const float mc = mx * p.c;
    float psum = 0.f;
    frag_t pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typedef float f32x8 __attribute__((ext_vector_type(8)));
      f32x8 e;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[j] = __builtin_amdgcn_exp2f(fmaf(s[2 * ks + (j >> 2)][j & 3], p.c, -mc));
        psum += e[j];
      }
      pf[ks] = __builtin_convertvector(e, frag_t);
    }
    l_run += psum;
    if (KLDS || tt > 0) __syncthreads();      // the previous tile's transposed reads (KLDS: this tile's K fragment reads) are done
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = i * 4 + fg;
      *reinterpret_cast<frag_t*>(Vs + row * 256 + ((fc ^ ((row & 7) << 1)) << 4)) = kv[i];
      if (fuse && key0 + row == pp)
        st8<T>((T*)p.v_cache_w + b * p.v_sb + kvh * p.v_sh + (int64_t)pp * p.v_sr + fc * 8, kv[i]);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) {
        const int ch = (2 * dn + (tp >> 1)) ^ vswz;
        const char* a0 = Vs + (ks * 32 + vrow_lo) * 256 + (ch << 4) + 8 * (tp & 1);
        const s16x4 lo = tr_read(a0);
        const s16x4 hi = tr_read(a0 + 16 * 256);
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[dn] = mfma16(__builtin_bit_cast(frag_t, cat), pf[ks], o[dn]);
      }
  }
  const float l = sum_xor32(sum_xor16(l_run));
  if (fc < n_rep) {
#pragma unroll
    for (int dn = 0; dn < 8; ++dn) *reinterpret_cast<f32x4*>(wsb + dn * 16 + fg * 4) = o[dn];
    if (fg == 0) { wsb[128] = m_run; wsb[129] = l; }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Decode attention over the e4m3 KV cache for LONG contexts (round 6; tuning key 47): one wave walks TPW consecutive 64-key tiles with the
// NEXT tile's bytes already on their way.  An e4m3 tile is half the registers of a 16-bit one (K 32 + V 32 VGPRs as raw bytes), so two tiles
// fit next to the running output -- the 16-bit multi-tile kernel above cannot prefetch, this one does -- and the widening to the MFMA's
// 16-bit operands happens where a fragment is consumed.  Against one wave per tile (attn_decode_kernel<KV8>) the launch writes 1 / TPW of
// the partials and the merge reads 1 / TPW of them: at the 33 k keys of BASELINE configs[4] that pair was the worst row of the decode layer
// (13.2 + 6.9 us for 34 MB, profiles/r06_a_*).
//   K: lane (fc, fg) loads bytes [64 h + 16 fg, +16) of key row key0 + 16 kt + fc (8 loads of 16 B per tile, 64 contiguous bytes per row per
//      instruction); the MFMA's k index is a free permutation of d as long as Q is loaded the same way: step 2 h + half, k = 8 fg + j  <->
//      d = 64 h + 16 fg + 8 half + j.
//   V: as in the one-tile kernel (whole 128-byte rows, widened into the transposed-read image).
//   scales: ONE float per lane per array (key0 + lane), redistributed to the score layout (key0 + 16 kt + 4 fg + r) through 512 B of LDS
//      (the front of the V image: this wave's previous transposed reads are behind the hand-over in its LDS order, the image write after it).
// The workgroup is a single wave: its LDS operations execute in program order, so a compiler barrier is all the hand-over needs.
// Same arithmetic as attn_decode_tile<KV8> per tile (score * k_scale in fp32, v_scale folded into P), the running max / sum of the
// multi-tile kernel across tiles.
// ---------------------------------------------------------------------------------------------------------
// NW = 4: four such waves (one per SIMD) share a workgroup and fold their running states in LDS before the partial is written -- a split is
// NW x TPW tiles, so that 33 k keys leave 44 partials per head and the merge is the one-round-trip launch (<= 64 partials) again.
// FUSE (round 6, tuning key 48): the RoPE + append + quantise launch in front of this one (rope_kv_kernel with the e4m3 outputs) folded in,
// as the 16-bit one-tile kernel folds its RoPE + append: every wave rotates q in registers and -- under the tile loads already in flight --
// rotates and quantises the new token's k / v rows (s = absmax / 448, bytes = e4m3_rne(x / s): rope_kv_kernel's arithmetic, the same bytes);
// the wave whose tiles hold position kv_len - 1 stores them (e4m3 rows + scales, and the 16-bit rows the other cache keeps), and every tile
// row at or beyond that position takes the new bytes from registers (those rows of the cache are stale or being written).
// (FUSE: the new rows' bytes and scales live across the tile loop next to two tiles of raw bytes and the running output: ~300 registers, one
// wave per SIMD -- attn_decode_kv8_fuses_rope takes this form only while the launch is one resident round of such waves.)
template <typename T, int TPW, int NW, bool FUSE = false>
__global__ __launch_bounds__(64 * NW, FUSE ? 1 : 2) void attn_decode_kv8_walk_kernel(AttnP p) {
  typedef typename V8<T>::type frag_t;
  __shared__ __attribute__((aligned(256))) char Vs_all[NW][KV_TILE * 256];
  const int lane = threadIdx.x & 63, fc = lane & 15, fg = lane >> 4;
  const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const Vs = Vs_all[wave];
  float* const Ss = reinterpret_cast<float*>(Vs);      // the scale hand-over borrows the image's first 512 bytes: it is over before the image is written
  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int n_rep = p.q_heads / p.kv_heads, hq0 = kvh * n_rep;
  const int kv_len = p.kv_len ? p.kv_len[b] : p.Skv;
  const int key_base = (split * NW + wave) * KV_TILE * TPW;
  float* wsb = p.ws + ((size_t)(b * p.q_heads + hq0 + (fc < n_rep ? fc : 0)) * p.nsplit + split) * WS_STRIDE;
  if (split * NW * KV_TILE * TPW >= kv_len) {                   // empty split (uniform over the workgroup): neutral partial
    if (wave == 0 && fc < n_rep && fg == 0) { wsb[128] = NEG_BIG; wsb[129] = 0.f; }
    return;
  }
  const bool wave_live = key_base < kv_len;      // NW > 1: a wave beyond the sequence contributes the neutral state (its loads are clamped, its scores masked)
  const unsigned char* Kg = (const unsigned char*)p.K + b * p.k_sb + kvh * p.k_sh;
  const unsigned char* Vg = (const unsigned char*)p.V + b * p.v_sb + kvh * p.v_sh;
  const float* ksp = p.k_scale + b * p.scale_sb + kvh * p.scale_sh;
  const float* vsp = p.v_scale + b * p.scale_sb + kvh * p.scale_sh;

  frag_t qf[4];                               // step 2 h + half: d = 64 h + 16 fg + 8 half + j
  {
    const int hh = fc < n_rep ? fc : n_rep - 1;
    const T* qp = (const T*)p.Q + b * p.q_sb + (hq0 + hh) * p.q_sh;
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = ld8<T>(qp + (st >> 1) * 64 + fg * 16 + (st & 1) * 8);
  }
  // FUSE: the new token's raw k (in q's layout) and v (chunk fc) rows and the RoPE table row, requested in front of the tiles
  const int pp = kv_len - 1;
  frag_t knf[4], vnf;
  float csv[2][16];
  if constexpr (FUSE) {
    const int pt = pp < p.rope_max ? pp : p.rope_max - 1;
    const T* kn = (const T*)p.k_new + b * p.new_sb + kvh * 128;
    const T* vn = (const T*)p.v_new + b * p.new_sb + kvh * 128;
#pragma unroll
    for (int st = 0; st < 4; ++st) knf[st] = ld8<T>(kn + (st >> 1) * 64 + fg * 16 + (st & 1) * 8);
    vnf = ld8<T>(vn + fc * 8);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4* cs = reinterpret_cast<const f32x4*>(p.rope + ((size_t)pt * 64 + 16 * fg + 8 * half) * 2);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) { const f32x4 v = cs[q4]; csv[half][4 * q4] = v[0]; csv[half][4 * q4 + 1] = v[1]; csv[half][4 * q4 + 2] = v[2]; csv[half][4 * q4 + 3] = v[3]; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  struct Raw { u32x4 k[4][2]; u32x2 v[16]; float ks, vs; };
  // every load of one tile; rows beyond the sequence are clamped onto its last row (their scores are masked below) -- no branch around loads
  auto issue = [&](Raw& r, int key0) {      // in the order of use: vector memory returns in order
    const int kc = key0 + lane < kv_len ? key0 + lane : kv_len - 1;
    r.ks = ksp[kc]; r.vs = vsp[kc];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int key = key0 + kt * 16 + fc;
      const unsigned char* src = Kg + (int64_t)(key < kv_len ? key : kv_len - 1) * p.k_sr + fg * 16;
      r.k[kt][0] = *reinterpret_cast<const u32x4*>(src);
      r.k[kt][1] = *reinterpret_cast<const u32x4*>(src + 64);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = key0 + i * 4 + fg;
      r.v[i] = *reinterpret_cast<const u32x2*>(Vg + (int64_t)(key < kv_len ? key : kv_len - 1) * p.v_sr + fc * 8);
    }
  };

  f32x4 o[8];
#pragma unroll
  for (int dn = 0; dn < 8; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const int tq = fc >> 2, tp = fc & 3;
  const int vrow_lo = 4 * fg + tq;
  const int vswz = ((vrow_lo & 7) << 1);

  Raw buf[2];
  issue(buf[0], key_base);
  u32x4 kq[2] = {};
  u32x2 vq = {};
  float sk = 1.f, sv = 1.f;
  if constexpr (FUSE) {
    __builtin_amdgcn_sched_barrier(0);
    // rotate-half partner of d = 64 h + 16 fg + 8 half + j is step (h ^ 1, half) of the same lane (q and the new key alike)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const frag_t lo = qf[half], hi = qf[2 + half];
      qf[half] = rope_chunk<T>(lo, hi, csv[half], false);
      qf[2 + half] = rope_chunk<T>(hi, lo, csv[half], true);
    }
    // only a wave whose tiles reach the new position needs the new rows (uniform; no loads inside: nothing is waited for at the join)
    if (key_base + TPW * KV_TILE > pp) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const frag_t kl = knf[half], kh = knf[2 + half];
        knf[half] = rope_chunk<T>(kl, kh, csv[half], false);
        knf[2 + half] = rope_chunk<T>(kh, kl, csv[half], true);
      }
      // quantise the rotated key row (this lane holds 32 of its 128 elements; the row's absmax over the four fg groups) and the value row
      // (8 elements per lane; absmax over the 16 fc lanes).  The 16 fc lanes hold the same key elements and the 4 fg groups the same value
      // elements, so every lane divides only ONE four-byte word per 16-byte piece and the words are gathered by lane shuffles
      // (12 divisions per lane instead of 40 on the launch's critical path).
      float mk = 0.f, mv = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) mk = fmaxf(mk, fabsf(tof(knf[st][j])));
      mk = max_xor32(max_xor16(mk));
#pragma unroll
      for (int j = 0; j < 8; ++j) mv = fmaxf(mv, fabsf(tof(vnf[j])));
#pragma unroll
      for (int o1 = 1; o1 < 16; o1 <<= 1) mv = fmaxf(mv, __shfl_xor(mv, o1, 64));
      sk = mk > 0.f ? mk / 448.0f : 1.0f;
      sv = mv > 0.f ? mv / 448.0f : 1.0f;
      auto pack4 = [](float a0, float a1, float a2, float a3, float sc) {
        const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(a0 / sc, a1 / sc, 0, false);
        const int hi = __builtin_amdgcn_cvt_pk_fp8_f32(a2 / sc, a3 / sc, 0, false);
        return (unsigned)(lo & 0xFFFF) | ((unsigned)(hi & 0xFFFF) << 16);
      };
      const int wsel = fc & 3;      // the word of a key piece this lane quantises: elements 4 wsel .. 4 wsel + 3 of d = 64 h + 16 fg + (0..15)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float e[4];      // (element-wise selects: a lane-dependent index into the fragments would send them through scratch)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a0 = tof(knf[2 * h][j]), a1 = tof(knf[2 * h][4 + j]), a2 = tof(knf[2 * h + 1][j]), a3 = tof(knf[2 * h + 1][4 + j]);
          e[j] = wsel == 0 ? a0 : (wsel == 1 ? a1 : (wsel == 2 ? a2 : a3));
        }
        const unsigned mine = pack4(e[0], e[1], e[2], e[3], sk);
#pragma unroll
        for (int w = 0; w < 4; ++w) kq[h][w] = (unsigned)__shfl((int)mine, (lane & ~3) | w, 64);
      }
      {
        const bool up = fg & 1;      // value word (fg & 1): elements 4 (fg & 1) .. + 3 of d = 8 fc + (0..7)
        const unsigned mine = pack4(tof(up ? vnf[4] : vnf[0]), tof(up ? vnf[5] : vnf[1]), tof(up ? vnf[6] : vnf[2]), tof(up ? vnf[7] : vnf[3]), sv);
        vq[0] = (unsigned)__shfl((int)mine, fc, 64);
        vq[1] = (unsigned)__shfl((int)mine, fc + 16, 64);
      }
      // the wave whose tiles hold position pp appends: e4m3 rows + scales, and the 16-bit rows of the other cache
      if (wave_live && pp >= key_base) {
        const int64_t ko = b * p.k_sb + kvh * p.k_sh + (int64_t)pp * p.k_sr, vo = b * p.v_sb + kvh * p.v_sh + (int64_t)pp * p.v_sr;
        if (fc == 0) {
#pragma unroll
          for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>((unsigned char*)p.K + ko + 64 * h + 16 * fg) = kq[h];
#pragma unroll
          for (int st = 0; st < 4; ++st) st8<T>((T*)p.k_cache_w + ko + (st >> 1) * 64 + fg * 16 + (st & 1) * 8, knf[st]);
        }
        if (fg == 0) {
          *reinterpret_cast<u32x2*>((unsigned char*)p.V + vo + fc * 8) = vq;
          st8<T>((T*)p.v_cache_w + vo + fc * 8, vnf);
        }
        if (lane == 0) {
          ((float*)p.k_scale)[b * p.scale_sb + kvh * p.scale_sh + pp] = sk;
          ((float*)p.v_scale)[b * p.scale_sb + kvh * p.scale_sh + pp] = sv;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int key0 = key_base + tt * KV_TILE;
    Raw& cur = buf[tt & 1];
    if (tt + 1 < TPW) issue(buf[(tt + 1) & 1], key0 + KV_TILE);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSE) {
      if (key0 + KV_TILE > pp) {      // (uniform) rows at or beyond the new position: the new bytes (rows beyond it are masked, but must stay finite)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int h = 0; h < 2; ++h) cur.k[kt][h] = key0 + kt * 16 + fc >= pp ? kq[h] : cur.k[kt][h];
#pragma unroll
        for (int i = 0; i < 16; ++i) cur.v[i] = key0 + i * 4 + fg >= pp ? vq : cur.v[i];
        cur.ks = key0 + lane >= pp ? sk : cur.ks;
        cur.vs = key0 + lane >= pp ? sv : cur.vs;
      }
    }

    // ---- scales into the score layout
    Ss[lane] = cur.ks; Ss[KV_TILE + lane] = cur.vs;
    asm volatile("" ::: "memory");
    f32x4 ksc[4], vsc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) ksc[kt] = *reinterpret_cast<const f32x4*>(Ss + kt * 16 + 4 * fg);
    // ---- S^T = K Q^T, K widened fragment by fragment
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32x4 w = cur.k[kt][h];
        s[kt] = mfma16(widen8<T>((u32x2){w.x, w.y}), qf[2 * h], s[kt]);
        s[kt] = mfma16(widen8<T>((u32x2){w.z, w.w}), qf[2 * h + 1], s[kt]);
      }
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = key0 + kt * 16 + 4 * fg + r < kv_len ? s[kt][r] * ksc[kt][r] : NEG_BIG;
        s[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    // (the value scales are fetched only now: the K tile's registers are free, and they are not live across the S^T MFMAs)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) vsc[kt] = *reinterpret_cast<const f32x4*>(Ss + KV_TILE + kt * 16 + 4 * fg);
    mx = max_xor32(max_xor16(mx));
    if (tt > 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
      l_run *= alpha;
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) o[dn] *= alpha;
      mx = m_new;
    }
    m_run = mx;
    const float mc = mx * p.c;
    float psum = 0.f;
    frag_t pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typedef float f32x8 __attribute__((ext_vector_type(8)));
      f32x8 e;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[j] = __builtin_amdgcn_exp2f(fmaf(s[2 * ks + (j >> 2)][j & 3], p.c, -mc));
        psum += e[j];
        e[j] *= vsc[2 * ks + (j >> 2)][j & 3];      // V = scale * e4m3: the per-key scale folded into P
      }
      pf[ks] = __builtin_convertvector(e, frag_t);
    }
    l_run += psum;
    // ---- V -> transposed-read image (behind the previous tile's reads in this wave's LDS order)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = i * 4 + fg;
      *reinterpret_cast<frag_t*>(Vs + row * 256 + ((fc ^ ((row & 7) << 1)) << 4)) = widen8<T>(cur.v[i]);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) {
        const int ch = (2 * dn + (tp >> 1)) ^ vswz;
        const char* a0 = Vs + (ks * 32 + vrow_lo) * 256 + (ch << 4) + 8 * (tp & 1);
        const s16x4 lo = tr_read(a0);
        const s16x4 hi = tr_read(a0 + 16 * 256);
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[dn] = mfma16(__builtin_bit_cast(frag_t, cat), pf[ks], o[dn]);
      }
    asm volatile("" ::: "memory");
  }
  float l = sum_xor32(sum_xor16(l_run));
  if constexpr (NW > 1) {
    // the waves' states meet in LDS (each wave's own image region is dead: its last transposed reads are behind it in its LDS order):
    // [32 O values + m + l][64 lanes] floats per wave; wave 0 folds them in wave order
    if (!wave_live) {      // every key of this wave was masked: with a NEG_BIG maximum its exponentials were exp2(0) -- drop that state
      m_run = NEG_BIG; l = 0.f;
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float* const mg = reinterpret_cast<float*>(Vs);
    if (wave > 0) {
#pragma unroll
      for (int dn = 0; dn < 8; ++dn)
#pragma unroll
        for (int r = 0; r < 4; ++r) mg[(dn * 4 + r) * 64 + lane] = o[dn][r];
      mg[32 * 64 + lane] = m_run; mg[33 * 64 + lane] = l;
    }
    __syncthreads();
    if (wave > 0) return;
    float mw[NW], lw[NW];
    mw[0] = m_run; lw[0] = l;
    float M = m_run;
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float* const g = reinterpret_cast<const float*>(Vs_all[w]);
      mw[w] = g[32 * 64 + lane]; lw[w] = g[33 * 64 + lane];
      M = fmaxf(M, mw[w]);
    }
    const float a0 = __builtin_amdgcn_exp2f((m_run - M) * p.c);
    l = l * a0;
#pragma unroll
    for (int dn = 0; dn < 8; ++dn) o[dn] *= a0;
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float* const g = reinterpret_cast<const float*>(Vs_all[w]);
      const float aw = __builtin_amdgcn_exp2f((mw[w] - M) * p.c);
      l += lw[w] * aw;
#pragma unroll
      for (int dn = 0; dn < 8; ++dn)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dn][r] += g[(dn * 4 + r) * 64 + lane] * aw;
    }
    m_run = M;
  }
  if (fc < n_rep) {
#pragma unroll
    for (int dn = 0; dn < 8; ++dn) *reinterpret_cast<f32x4*>(wsb + dn * 16 + fg * 4) = o[dn];
    if (fg == 0) { wsb[128] = m_run; wsb[129] = l; }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Batched decode attention with the K / V tiles brought in by LDS-DMA (round 4; tuning key 25).  attn_decode_multi_kernel above walks its
// tiles with TWO exposed memory round trips per tile (K -> S^T -> V -> PV: K and V share one register block because both next to the
// running output do not fit), 7.5 waves per CU at b = 32: by its own timeline the launch is 4 tiles x 2 round trips of ~5 us.  Here the
// tiles never touch the vector registers on their way in: one wave per workgroup owns a two-stage LDS ring of 32-key tiles (stage = K
// 8 KiB + V 8 KiB, written by global_load_lds_dwordx4 in the swizzled images the operand reads want: the swizzle is applied to the
// per-lane SOURCE address, the DMA destination is lane-linear), tile t + 1 (and, once a stage is free, t + 2) is in flight while tile t is
// computed, and four such workgroups per CU keep up to 128 KB per CU on the way -- what the weight-streaming GEMVs hold.  A 32-key tile is
// one K = 32 step of the PV MFMA and two S^T fragments; running max / sum per tile as in the multi-tile kernel; the partials, the merge
// and the fused RoPE + append (the split that owns the new position patches the rotated key row into its LDS image and appends k / v)
// are the same.  p.tpw = 32-key tiles per split.
// ---------------------------------------------------------------------------------------------------------
constexpr int DMA_TILE = 32;
template <typename T, int NST, bool NT = (OMCHAT_KV_NT != 0)>
__global__ __launch_bounds__(64) void attn_decode_dma_kernel(AttnP p) {
  typedef typename V8<T>::type frag_t;
  extern __shared__ __attribute__((aligned(256))) char dsm[];      // NST stages x (K 8 KiB | V 8 KiB)
  const int lane = threadIdx.x, fc = lane & 15, fg = lane >> 4;
  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int n_rep = p.q_heads / p.kv_heads, hq0 = kvh * n_rep;
  const int kv_len = p.kv_len ? p.kv_len[b] : p.Skv;
  const int key_base = split * DMA_TILE * p.tpw;
  float* wsb = p.ws + ((size_t)(b * p.q_heads + hq0 + (fc < n_rep ? fc : 0)) * p.nsplit + split) * WS_STRIDE;
  if (key_base >= kv_len) {                   // empty split (uniform): neutral partial
    if (fc < n_rep && fg == 0) { wsb[128] = NEG_BIG; wsb[129] = 0.f; }
    return;
  }
  const int left = (kv_len - key_base + DMA_TILE - 1) / DMA_TILE;
  const int nt = left < p.tpw ? left : p.tpw;                     // tiles this wave walks (uniform)
  const bool fuse = p.rope != nullptr;
  const int pp = kv_len - 1;
  const int pt = pp < p.rope_max ? pp : p.rope_max - 1;
  const T* Kg = (const T*)p.K + b * p.k_sb + kvh * p.k_sh;
  const T* Vg = (const T*)p.V + b * p.v_sb + kvh * p.v_sh;
  const T* kn = fuse ? (const T*)p.k_new + b * p.new_sb + kvh * 128 : nullptr;
  const T* vn = fuse ? (const T*)p.v_new + b * p.new_sb + kvh * 128 : nullptr;

  // tile t -> stage st: lane (fg, fc) brings LDS chunk position fc of tile row 4 i + fg, i.e. source chunk fc ^ swizzle(row)
  // p.causal (unused by decode) = 1: every wave starts at a different tile of its split and wraps around (experiment: do the lock-step
  // streams of a launch camp on the same HBM channels?)
  const int rot = p.causal ? (b * 5 + kvh * 3 + split * 7) % nt : 0;
  auto issue = [&](int t_, int st) {
    const int t = t_ + rot < nt ? t_ + rot : t_ + rot - nt;
    const int key0 = key_base + t * DMA_TILE;
    char* kb = dsm + st * 16384;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = i * 4 + fg, key = key0 + row;
      const bool fresh = fuse && key >= pp;                         // not in the cache yet (or clamped onto it): the raw new row, finite
      const int64_t kc = key < kv_len ? key : kv_len - 1;
      const T* ks = (fresh ? kn : Kg + kc * p.k_sr) + ((fc ^ (row & 15)) << 3);
      const T* vs = (fresh ? vn : Vg + kc * p.v_sr) + ((fc ^ ((row & 7) << 1)) << 3);
      if (NT) {      // streamed once: non-temporal (the weight-streaming GEMVs load theirs the same way)
        __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(kb + i * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds((gptr_t)vs, (lptr_t)(kb + 8192 + i * 1024), 16, 0, 2);
      } else {
        __builtin_amdgcn_global_load_lds((gptr_t)ks, (lptr_t)(kb + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)vs, (lptr_t)(kb + 8192 + i * 1024), 16, 0, 0);
      }
    }
  };
  // q, the RoPE table row and (in the split that owns the new position) the raw new rows are requested FIRST: vector memory returns in
  // order, so behind the tiles they would not be usable before both stages have landed
  frag_t qf[4];
  const bool owner = fuse && pp >= key_base && pp < key_base + nt * DMA_TILE;      // uniform
  frag_t kown = {}, koth = {}, vnew = {};
  float csq[2][16], cso[16];
  {
    const int hh = fc < n_rep ? fc : n_rep - 1;
    const T* qp = (const T*)p.Q + b * p.q_sb + (hq0 + hh) * p.q_sh;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = ld8<T>(qp + ds * 32 + fg * 8);
    if (fuse) {
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const f32x4* cs = reinterpret_cast<const f32x4*>(p.rope + ((size_t)pt * 64 + ds * 32 + fg * 8) * 2);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) { const f32x4 v = cs[q4]; csq[ds][4 * q4] = v[0]; csq[ds][4 * q4 + 1] = v[1]; csq[ds][4 * q4 + 2] = v[2]; csq[ds][4 * q4 + 3] = v[3]; }
      }
    }
    if (owner) {
      kown = ld8<T>(kn + fc * 8); koth = ld8<T>(kn + (fc ^ 8) * 8); vnew = ld8<T>(vn + fc * 8);
      const f32x4* cs = reinterpret_cast<const f32x4*>(p.rope + ((size_t)pt * 64 + (fc & 7) * 8) * 2);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) { const f32x4 v = cs[q4]; cso[4 * q4] = v[0]; cso[4 * q4 + 1] = v[1]; cso[4 * q4 + 2] = v[2]; cso[4 * q4 + 3] = v[3]; }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NST; ++i)
    if (i < nt) issue(i, i);
  __builtin_amdgcn_sched_barrier(0);
  if (fuse) {
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      const frag_t lo = qf[ds], hi = qf[ds + 2];
      qf[ds] = rope_chunk<T>(lo, hi, csq[ds], false);
      qf[ds + 2] = rope_chunk<T>(hi, lo, csq[ds], true);
    }
  }
  frag_t knew = {};
  if (owner) {      // chunk fc of the new rows in every lane group (only group 0 writes); rotate-half partner = chunk fc ^ 8
    knew = rope_chunk<T>(kown, koth, cso, fc >= 8);
    if (fg == 0) {
      st8<T>((T*)p.k_cache_w + b * p.k_sb + kvh * p.k_sh + (int64_t)pp * p.k_sr + fc * 8, knew);
      st8<T>((T*)p.v_cache_w + b * p.v_sb + kvh * p.v_sh + (int64_t)pp * p.v_sr + fc * 8, vnew);
    }
  }

  f32x4 o[8];
#pragma unroll
  for (int dn = 0; dn < 8; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const int tq = fc >> 2, tp = fc & 3;
  const int vrow_lo = 4 * fg + tq;
  const int vswz = ((vrow_lo & 7) << 1);

  for (int t = 0; t < nt; ++t) {
    const int st = t % NST;
    const int key0 = key_base + (t + rot < nt ? t + rot : t + rot - nt) * DMA_TILE;
    const char* kb = dsm + st * 16384;
    const char* vb = kb + 8192;
    // tile t has landed when at most the 16 DMA instructions of each tile issued behind it are outstanding
    {
      const int behind = nt - 1 - t < NST - 1 ? nt - 1 - t : NST - 1;      // uniform
      if (behind >= 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else if (behind == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (behind == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (owner && pp >= key0 && pp < key0 + DMA_TILE) {             // uniform: the rotated key row replaces the raw one in the image
      const int r = pp - key0;
      if (fg == 0) *reinterpret_cast<frag_t*>(dsm + st * 16384 + r * 256 + ((fc ^ (r & 15)) << 4)) = knew;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    f32x4 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      frag_t kf[4];
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) kf[ds] = *reinterpret_cast<const frag_t*>(kb + (kt * 16 + fc) * 256 + (((ds * 4 + fg) ^ fc) << 4));
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) s[kt] = mfma16(kf[ds], qf[ds], s[kt]);
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = key0 + kt * 16 + 4 * fg + r < kv_len ? s[kt][r] : NEG_BIG;
        s[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = max_xor32(max_xor16(mx));
    if (t > 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
      l_run *= alpha;
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) o[dn] *= alpha;
      mx = m_new;
    }
    m_run = mx;
    const float mc = mx * p.c;
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 e;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      e[j] = __builtin_amdgcn_exp2f(fmaf(s[j >> 2][j & 3], p.c, -mc));
      l_run += e[j];
    }
    const frag_t pf = __builtin_convertvector(e, frag_t);
    // the transposed reads go out as inline assembly: hipcc puts an s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 it emits itself
    // while an LDS-DMA is outstanding (the intrinsic carries no memory operand to disambiguate), which would drain the ring at every tile
    s16x4 vlo[8], vhi[8];
    {
      const unsigned vbase = (unsigned)(size_t)(lds_s16x4_ptr)(vb + vrow_lo * 256 + 8 * (tp & 1));
#pragma unroll
      for (int dn = 0; dn < 8; ++dn) {
        const unsigned addr = vbase + ((unsigned)((2 * dn + (tp >> 1)) ^ vswz) << 4);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[dn]) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(vhi[dn]) : "v"(addr));
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vlo[0]), "+v"(vlo[1]), "+v"(vlo[2]), "+v"(vlo[3]), "+v"(vlo[4]), "+v"(vlo[5]), "+v"(vlo[6]), "+v"(vlo[7]),
                     "+v"(vhi[0]), "+v"(vhi[1]), "+v"(vhi[2]), "+v"(vhi[3]), "+v"(vhi[4]), "+v"(vhi[5]), "+v"(vhi[6]), "+v"(vhi[7]));
    }
#pragma unroll
    for (int dn = 0; dn < 8; ++dn) {
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 cat = __builtin_shufflevector(vlo[dn], vhi[dn], 0, 1, 2, 3, 4, 5, 6, 7);
      o[dn] = mfma16(__builtin_bit_cast(frag_t, cat), pf, o[dn]);
    }
    // every LDS read of this stage has returned (its MFMAs are issued): the stage takes tile t + NST
    if (t + NST < nt) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(t + NST, st);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float l = sum_xor32(sum_xor16(l_run));
  if (fc < n_rep) {
#pragma unroll
    for (int dn = 0; dn < 8; ++dn) *reinterpret_cast<f32x4*>(wsb + dn * 16 + fg * 4) = o[dn];
    if (fg == 0) { wsb[128] = m_run; wsb[129] = l; }
  }
}

// merge split-KV partials: one 128-thread block per (sequence, head).  Phase 1: thread s owns split s (max, weight);
// phase 2: thread d sums its column over the splits with independent (unrolled) loads.
template <typename T>
__global__ __launch_bounds__(128) void attn_merge_kernel(const float* ws, int nsplit, int q_heads, const int* kv_len, int L, float c,
                                                         T* O, int64_t o_sb, int64_t o_sh, int pack_nb, int split_keys,
                                                         unsigned* done_flags = nullptr, unsigned done_epoch = 0, int done_mode = 0,
                                                         unsigned long long* done_dbg = nullptr) {
  __shared__ float fw[1024];
  __shared__ float red[4];
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
#if OMCHAT_EXPERIMENTS
  if (done_dbg && d == 0 && h < 8) atomicMax(done_dbg + 12, ~wall_clock64());      // measurement: first start (stored inverted)
#endif
  const int len = kv_len ? kv_len[b] : L;
  int ns = (len + split_keys - 1) / split_keys;
  ns = ns < nsplit ? ns : nsplit;
  const float* w = ws + (size_t)(b * q_heads + h) * nsplit * WS_STRIDE;
  if (ns <= 64) {
    // short contexts (<= 4096 keys): this launch is pure latency, so every global load is issued before anything is reduced -- one
    // round trip instead of three (split maxima, then weights, then the partial outputs).  Same arithmetic, same summation order.
    // the (max, sum) pair of split d goes out IN FRONT of the 64 partial-output loads (in-order return: behind them the softmax weights
    // could not be formed before everything had arrived); no branch around the loads
    const int sd = d & 63;
    const float m_ld = w[(size_t)(sd < ns ? sd : 0) * WS_STRIDE + 128];
    const float l_ld = w[(size_t)(sd < ns ? sd : 0) * WS_STRIDE + 129];
    __builtin_amdgcn_sched_barrier(0);
    float v[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) v[s] = w[(size_t)(s < ns ? s : 0) * WS_STRIDE + d];
    __builtin_amdgcn_sched_barrier(0);
    if (d < 64) {
      const float m_s = d < ns ? m_ld : NEG_BIG;
      const float l_s = d < ns ? l_ld : 0.f;
      const float m = wave_max(m_s);
      const float f = d < ns ? exp2f((m_s - m) * c) : 0.f;
      const float lsum = wave_sum(f * l_s);
      fw[d] = f;
      if (d == 0) red[0] = lsum;
    }
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < 64; ++s) a += fw[s] * v[s];
#if OMCHAT_EXPERIMENTS
    if (done_flags && (done_mode & 4)) {
      // mode bit 2: the row element as a write-through store (sc0 sc1), drained, instead of a plain store + agent-scope release (buffer_wbl2 sc1)
      const T o = fromf<T>(a / red[0]);
      const size_t idx = pack_nb ? packed_x_index(b, h * 128 + d, pack_nb) : (size_t)(b * o_sb + h * o_sh + d);
      const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(O + idx - d, 0, 256, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, o), os, d * 2, 0, 17);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (d == 0) __hip_atomic_store(done_flags + h, done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done_dbg && d == 0) atomicMax(done_dbg, wall_clock64());
      return;
    }
#endif
    O[pack_nb ? packed_x_index(b, h * 128 + d, pack_nb) : (size_t)(b * o_sb + h * o_sh + d)] = fromf<T>(a / red[0]);
#if OMCHAT_EXPERIMENTS
    if (done_flags) {      // tuning key 42: an out-of-order consumer polls these (gemv_rows_wait_kernel)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (d == 0) __hip_atomic_store(done_flags + h, done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (done_dbg && d == 0) atomicMax(done_dbg, wall_clock64());
#endif
    return;
  }
  float acc = 0.f, ltot = 0.f, m_run = NEG_BIG;
  for (int s0 = 0; s0 < ns; s0 += 1024) {                 // chunks of up to 1024 splits (64k keys)
    const int n = ns - s0 < 1024 ? ns - s0 : 1024;
    // phase 1
    float mloc = NEG_BIG;
    for (int s = d; s < n; s += 128) mloc = fmaxf(mloc, w[(size_t)(s0 + s) * WS_STRIDE + 128]);
    mloc = wave_max(mloc);
    if ((d & 63) == 0) red[d >> 6] = mloc;
    __syncthreads();
    const float m_new = fmaxf(m_run, fmaxf(red[0], red[1]));
    __syncthreads();
    float lloc = 0.f;
    for (int s = d; s < n; s += 128) {
      const float f = exp2f((w[(size_t)(s0 + s) * WS_STRIDE + 128] - m_new) * c);
      fw[s] = f;
      lloc += f * w[(size_t)(s0 + s) * WS_STRIDE + 129];
    }
    lloc = wave_sum(lloc);
    if ((d & 63) == 0) red[2 + (d >> 6)] = lloc;
    __syncthreads();
    const float alpha = exp2f((m_run - m_new) * c);
    ltot = ltot * alpha + red[2] + red[3];
    acc *= alpha;
    m_run = m_new;
    // phase 2
    const float* wd = w + (size_t)s0 * WS_STRIDE + d;
    int s = 0;
    for (; s + 8 <= n; s += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = wd[(size_t)(s + u) * WS_STRIDE];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += fw[s + u] * v[u];
    }
    for (; s < n; ++s) acc += fw[s] * wd[(size_t)s * WS_STRIDE];
    __syncthreads();
  }
  // pack_nb: the consumer is the batched o_proj GEMV, which reads its x operand in MFMA fragment order (common.h)
  O[pack_nb ? packed_x_index(b, h * 128 + d, pack_nb) : (size_t)(b * o_sb + h * o_sh + d)] = fromf<T>(acc / ltot);
}

// 65 .. 1024 partials per head (4 k - 64 k keys at one tile per wave; round 3): 128 NG threads, group g = tid / 128 takes the g-th
// 1 / NG of the splits for thread column d = tid % 128, so that the loads of the launch are issued in one batch (<= 64 partial outputs per
// thread + one (max, sum) pair; two batches beyond 64 NG splits): one or two memory round trips like the <= 64 path above, where the
// general path needs ~ns / 8 + 2 dependent ones (8.8 k keys, BASELINE configs[3]: 8.4 -> 6.7 us per launch, 2.94 -> 2.88 ms per token).
// Fixed summation order: inside a group by split, then the groups left to right.
template <typename T, int NG, int DG>
__global__ __launch_bounds__(128 * NG) void attn_merge_mid_kernel(const float* ws, int nsplit, int q_heads, const int* kv_len, int L, float c,
                                                                  T* O, int64_t o_sb, int64_t o_sh, int pack_nb, int split_keys) {
  // DG > 1: blockIdx.z takes 128 / DG of the head's columns and the 128 NG threads form NG DG split groups instead of NG -- DG times the
  // workgroups for the launches whose bytes matter (33 k keys, configs[4]: 520 partials per head = 7.6 MB read by 28 workgroups took 21 us)
  constexpr int NWV = 2 * NG, COLS = 128 / DG, NGRP = NG * DG;
  __shared__ float fw[128 * NG];
  __shared__ float red[2 * NWV];
  __shared__ float part[NGRP][COLS];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, dl = tid % COLS, d = blockIdx.z * COLS + dl, g = tid / COLS;
  const int len = kv_len ? kv_len[b] : L;
  int ns = (len + split_keys - 1) / split_keys;
  ns = ns < nsplit ? ns : nsplit;
  ns = ns < 128 * NG ? ns : 128 * NG;                // (the launcher only comes here when nsplit <= 128 NG)
  const float* w = ws + (size_t)(b * q_heads + h) * nsplit * WS_STRIDE;
  const int q = (ns + NGRP - 1) / NGRP, s_lo = g * q;     // this thread's splits: s_lo .. min(s_lo + q, ns) - 1, q <= 128
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) v[i] = w[(size_t)(i < q && s_lo + i < ns ? s_lo + i : 0) * WS_STRIDE + d];
  const float m_s = tid < ns ? w[(size_t)tid * WS_STRIDE + 128] : NEG_BIG;
  const float l_s = tid < ns ? w[(size_t)tid * WS_STRIDE + 129] : 0.f;
  const float mw = wave_max(m_s);
  if ((tid & 63) == 0) red[tid >> 6] = mw;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int i = 1; i < NWV; ++i) m = fmaxf(m, red[i]);
  const float f = tid < ns ? exp2f((m_s - m) * c) : 0.f;
  fw[tid] = f;
  const float lw = wave_sum(f * l_s);
  if ((tid & 63) == 0) red[NWV + (tid >> 6)] = lw;
  __syncthreads();
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) a += (i < q && s_lo + i < ns ? fw[s_lo + i] : 0.f) * v[i];
  if (q > 64) {                                      // second batch (uniform)
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = w[(size_t)(64 + i < q && s_lo + 64 + i < ns ? s_lo + 64 + i : 0) * WS_STRIDE + d];
#pragma unroll
    for (int i = 0; i < 64; ++i) a += (64 + i < q && s_lo + 64 + i < ns ? fw[s_lo + 64 + i] : 0.f) * v[i];
  }
  part[g][dl] = a;
  __syncthreads();
  if (g == 0) {
    float ltot = red[NWV], tot = part[0][dl];
#pragma unroll
    for (int i = 1; i < NWV; ++i) ltot += red[NWV + i];
#pragma unroll
    for (int i = 1; i < NGRP; ++i) tot += part[i][dl];
    O[pack_nb ? packed_x_index(b, h * 128 + d, pack_nb) : (size_t)(b * o_sb + h * o_sh + d)] = fromf<T>(tot / ltot);
  }
}

// fp8 KV cache: one wave per cache row (128 elements): s = absmax / 448 (1 for a zero row), bytes = e4m3_rne(x / s)
template <typename T>
__global__ __launch_bounds__(256) void kv_quant_kernel(const T* kc, const T* vc, unsigned char* k8, unsigned char* v8, float* ks, float* vs, int kv_heads,
                                                       int64_t c_sb, int64_t c_sh, int64_t s_sb, int64_t s_sh, const int* pos_lo, int pos0, const int* len,
                                                       int max_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;                        // row index inside [pos_lo, ...) of this (sequence, head)
  const int h = blockIdx.y, b = blockIdx.z;
  const int lo = pos_lo ? pos_lo[b] : pos0;
  const int pos = lo + r;
  if (r >= max_rows || (len && pos >= len[b])) return;        // wave-uniform
  const int64_t off = b * c_sb + h * c_sh + (int64_t)pos * 128 + lane * 2;
  const int64_t soff = b * s_sb + h * s_sh + pos;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const T* src = (which ? vc : kc) + off;
    const float x0 = tof(src[0]), x1 = tof(src[1]);
    float m = fmaxf(fabsf(x0), fabsf(x1));
    m = wave_max(m);
    const float sc = m > 0.f ? m / 448.0f : 1.0f;
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x0 / sc, x1 / sc, 0, false);
    *reinterpret_cast<unsigned short*>((which ? v8 : k8) + off) = (unsigned short)(pk & 0xFFFF);
    if (lane == 0) (which ? vs : ks)[soff] = sc;
  }
}

}  // namespace

int launch_kv_quant(int dtype, const void* kc, const void* vc, void* k8, void* v8, float* ks, float* vs, int b, int kv_heads, int64_t c_sb, int64_t c_sh,
                    int64_t s_sb, int64_t s_sh, const int* pos_lo, int pos0, const int* len, int max_rows, hipStream_t s) {
  if (b <= 0 || max_rows <= 0) return 0;
  dim3 grid(cdiv(max_rows, 4), kv_heads, b);
  if (dtype == OMCHAT_F16) hipLaunchKernelGGL(kv_quant_kernel<f16>, grid, dim3(256), 0, s, (const f16*)kc, (const f16*)vc, (unsigned char*)k8, (unsigned char*)v8,
                                              ks, vs, kv_heads, c_sb, c_sh, s_sb, s_sh, pos_lo, pos0, len, max_rows);
  else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL(kv_quant_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)kc, (const bf16*)vc, (unsigned char*)k8,
                                                    (unsigned char*)v8, ks, vs, kv_heads, c_sb, c_sh, s_sb, s_sh, pos_lo, pos0, len, max_rows);
  else { omchat_set_error("launch_kv_quant: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}

int g_attn_kv8_tpw = 0;     // omchat_op_set_tuning key 47: 64-key tiles per wave of the decode attention over the e4m3 cache (attn_decode_kv8_walk_kernel): 0 = by the launch's size, 1 = one wave per tile (attn_decode_kernel), 2 / 3 / 4 forced, 11 = 3 tiles per wave x 4 waves per workgroup folded in LDS
void attn_set_kv8_tpw(int v) { g_attn_kv8_tpw = v < 0 || v > 15 ? 0 : v; }
int g_attn_tpw = 0;     // omchat_op_set_tuning key 10: key tiles per wave of the decode attention (0 = by grid size; 1, 2, 4 force)
void attn_set_tpw(int v) { g_attn_tpw = v < 0 ? 0 : v; }
int g_merge_dg = 1;       // omchat_op_set_tuning key 21: split-KV merge, column groups per head: 0 = none, 1 = 4 groups beyond 256 partials (default), 2 = also 2 groups for 65..256
void attn_set_merge_dg(int v) { g_merge_dg = v; }
int g_merge_mid_min = 64;   // omchat_op_set_tuning key 19: split-KV merges with more partials per head than this take the 512-thread form
void attn_set_merge_mid_min(int v) { g_merge_mid_min = v < 1 ? 1 : v; }
int g_attn_dma = 1;     // omchat_op_set_tuning key 25: 1 = large-grid batched decode attention takes the LDS-DMA ring form (attn_decode_dma_kernel), 0 = attn_decode_multi_kernel
void attn_set_dma(int v) { g_attn_dma = v; }
int g_attn_dma_slots = 0;     // key 26: resident one-wave workgroups per CU the split count of that form is sized for (low byte; 0 = by the launch's size), ring stages per wave (next byte, 2..4; 16 KiB of LDS each)
int g_attn_dma_stages = 2;
void attn_set_dma_slots(int v) { g_attn_dma_slots = v & 255; const int st = (v >> 8) & 255; g_attn_dma_stages = st < 2 ? 2 : (st > 4 ? 4 : st); }
int g_attn_kg = 0;            // key 36: key groups of the MHA prefill kernel (attn2_kernel KG): 0 = by the launch's fill, 1 / 2 = forced
void attn_set_kg(int v) { g_attn_kg = v; }
// attn2_kernel takes its K / V rings as dynamic LDS (128 KB with two key groups at head dim 128): the attribute is set once per instantiation
template <typename K>
static int launch_attn2(K kern, dim3 grid, int threads, int lds, hipStream_t s, const AttnP& p, PerDeviceOnce* attr_done) {
  if (attr_done->first()) OM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, p);
  return 0;
}
#define OM_LAUNCH_ATTN2(KERN, GRID, THREADS, LDS)                                     \
  do { static PerDeviceOnce done_; const int rc_ = launch_attn2((KERN), (GRID), (THREADS), (LDS), s, p, &done_); if (rc_) return rc_; } while (0)
int g_attn_peel = 1;          // key 46: 1 = MHA prefill attention with one key over whole tiles folds that key into the online softmax's initial state
void attn_set_peel(int v) { g_attn_peel = v; }
int g_attn_mha_xcd = 1;       // key 33: 1 = MHA prefill attention launches keep the query blocks of a head on one XCD (one-dimensional grid)
void attn_set_mha_xcd(int v) { g_attn_mha_xcd = v; }
int g_attn_hsplit = -1;       // key 30: heaviest causal block ranks of a GQA prefill attention launch issued as two head halves (-1 = a quarter of the ranks when the launch is <= one workgroup per CU, 0 = off, n > 0 = n ranks whatever the size)
void attn_set_hsplit(int v) { g_attn_hsplit = v; }
int g_attn_dma_rot = 0;       // key 27 (experiment): rotated tile order per wave
void attn_set_dma_rot(int v) { g_attn_dma_rot = v; }
int g_attn_klds = 0;    // omchat_op_set_tuning key 12: 1 = batched decode attention loads K as whole rows through LDS (measured neutral: 4.47 ms / step either way at b = 32, profiles/r03_c)
void attn_set_klds(int v) { g_attn_klds = v; }
int g_attn_v2 = 1;      // omchat_op_set_tuning key 8: 0 = first-generation 16x16x32 prefill kernel (A/B)
void attn_set_v2(int v) { g_attn_v2 = v; }

namespace {
// one workgroup per (head, sequence), thread = output column: the uniform average of the sequence's V rows, written to every padded query row
template <typename T>
__global__ __launch_bounds__(128) void attn_uniform_rows_kernel(AttnP p) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const int start = p.kv_start[b];
  if (start <= 0) return;
  const int kvh = h / (p.q_heads / p.kv_heads);
  const T* V = (const T*)p.V + b * p.v_sb + kvh * p.v_sh + d;
  const float w = rnd<T>(1.0f / (float)p.Skv);          // softmax in fp32, cast to the activation dtype (modeling_qwen2.py:166-167)
  float acc = 0.f;
  for (int j = 0; j < p.Skv; ++j) acc += w * tof(V[(int64_t)j * p.v_sr]);
  const T o = fromf<T>(acc);
  T* O = (T*)p.O + b * p.o_sb + h * p.o_sh + d;
  for (int i = 0; i < start && i < p.Sq; ++i) O[(int64_t)i * p.o_sr] = o;
}
}  // namespace

int launch_attn_uniform_rows(int dtype, const AttnArgs& a, hipStream_t s) {
  OM_CHECK(a.kv_start && (a.head_dim == 0 || a.head_dim == 128), "uniform rows: left-padded batch (kv_start), head dim 128");
  AttnP p{a.Q, a.K, a.V, a.O, a.q_sb, a.q_sh, a.q_sr, a.k_sb, a.k_sh, a.k_sr, a.v_sb, a.v_sh, a.v_sr, a.o_sb, a.o_sh, a.o_sr,
          a.kv_len, a.kv_start, a.q_heads, a.kv_heads, a.Sq, a.Skv, a.causal, a.q_pos0, 0, a.scale * 1.4426950408889634f, nullptr,
          nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, nullptr, 0};
  const dim3 grid(a.q_heads, a.batch);
  if (dtype == OMCHAT_F16) hipLaunchKernelGGL(attn_uniform_rows_kernel<f16>, grid, dim3(128), 0, s, p);
  else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL(attn_uniform_rows_kernel<bf16>, grid, dim3(128), 0, s, p);
  else { omchat_set_error("launch_attn_uniform_rows: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_attn_prefill(int dtype, const AttnArgs& a, hipStream_t s) {
  OM_CHECK(a.q_heads % a.kv_heads == 0, "q_heads must be a multiple of kv_heads");
  OM_CHECK(a.Sq > 0 && a.Skv > 0 && a.batch > 0, "empty attention");
  OM_CHECK(a.q_sr % 8 == 0 && a.k_sr % 8 == 0 && a.v_sr % 8 == 0 && a.o_sr % 4 == 0, "row strides must keep 16-B alignment");
  AttnP p{a.Q, a.K, a.V, a.O, a.q_sb, a.q_sh, a.q_sr, a.k_sb, a.k_sh, a.k_sr, a.v_sb, a.v_sh, a.v_sr, a.o_sb, a.o_sh, a.o_sr,
          a.kv_len, a.kv_start, a.q_heads, a.kv_heads, a.Sq, a.Skv, a.causal, a.q_pos0, 0, a.scale * 1.4426950408889634f, nullptr,
          nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, nullptr, 0};
  dim3 grid(cdiv(a.Sq, 128), a.q_heads, a.batch);
  p.peel_last = g_attn_peel;
  const int hd = a.head_dim ? a.head_dim : 128;
  OM_CHECK(hd == 128 || hd == 64, "head_dim must be 128 or 64");
  if (a.qn_sumsq) {      // Q half of the ViT's joint-head norm on load (attn2_kernel, MHA, head dim 128)
    OM_CHECK(a.q_heads == a.kv_heads && hd == 128 && g_attn_v2 && !a.causal && a.qn_w && a.qn_stride >= 1 && a.qn_dim > 0,
             "q norm on load: MHA at head dim 128 on the second-generation kernel");
    p.qn_sumsq = a.qn_sumsq; p.qn_stride = a.qn_stride; p.qn_dim = a.qn_dim; p.qn_w = a.qn_w; p.qn_eps = a.qn_eps; p.qn_scale = a.qn_scale;
  }
  // MHA on the second-generation kernel: query blocks of a (head, sequence) pair 8 workgroup ids apart = on one XCD (attn2_kernel; key 33 = 0: the
  // three-dimensional grid)
  dim3 grid_x = grid;
  if (g_attn_mha_xcd && !a.causal && grid.x < 65536 && a.batch < 32768) { p.tpw = -1; p.nsplit = (int)grid.x | (a.batch << 16); grid_x = dim3(cdiv(a.q_heads * a.batch, 8) * 8 * grid.x, 1, 1); }
  constexpr int LDS128 = 2 * 2 * KV_TILE * 256;      // two stages of a K + V tile pair at head dim 128
  // two key groups per workgroup (attn2_kernel KG = 2) when the launch's workgroups fill their resident rounds badly: `wgs` workgroups on
  // CUs x per_cu1 slots in whole rounds against twice the rounds of half length on CUs x per_cu2 slots (3-tile ViT: 675 workgroups = 2 rounds
  // of 512 slots against 3 half rounds of 256; 24 tiles: 11 rounds either way -> one group)
  auto use_kg2 = [&](int per_cu1, int per_cu2) {
    // measured (round 5, profiles/r05_r_attn_two_key_groups.txt): SLOWER -- 3 tiles 73.4 vs 68.0 us, 24 tiles 510.8 vs 436.4 us.  The eight waves share
    // every barrier, so the two waves of a SIMD sit in the same phase (both in their MFMAs, then both in the softmax), where two independent
    // four-wave workgroups drift apart and overlap one's MFMAs with the other's VALU: a step costs 17 % more and eats what the better fill
    // gives.  Kept for the experiments build only (forced with key 36 = 2).
    if (!OMCHAT_EXPERIMENTS || g_attn_kg != 2) return false;
    if (a.causal || a.kv_start || a.q_heads != a.kv_heads) return false;
    if (g_attn_kg) return g_attn_kg == 2;
    const long wgs = (long)grid.x * a.q_heads * a.batch, G = device_cus();
    const long r1 = (wgs + G * per_cu1 - 1) / (G * per_cu1), r2 = (wgs + G * per_cu2 - 1) / (G * per_cu2);
    return 2 * r1 > r2 + (r1 > 4 ? 1 : 0) && cdiv(a.Skv, KV_TILE) >= 4;      // strictly fewer half rounds (and not a marginal gain on long launches)
  };
  if (hd == 64 && g_attn_v2 && a.q_heads == a.kv_heads) {      // InternViT-300M on the second-generation kernel (round 3)
    // (head dim 64: 124 VGPRs, four 4-wave workgroups per CU, or two 8-wave ones with the keys split)
    const bool kg2 = use_kg2(4, 2); (void)kg2;
    constexpr int LDS64 = 2 * 2 * KV_TILE * 128;
#if OMCHAT_EXPERIMENTS
    if (kg2 && dtype == OMCHAT_F16) OM_LAUNCH_ATTN2((attn2_kernel<f16, 8, false, 64, 2>), grid_x, 512, 2 * LDS64);
    else if (kg2 && dtype == OMCHAT_BF16) OM_LAUNCH_ATTN2((attn2_kernel<bf16, 8, false, 64, 2>), grid_x, 512, 2 * LDS64);
    else
#endif
    if (dtype == OMCHAT_F16) OM_LAUNCH_ATTN2((attn2_kernel<f16, 4, false, 64>), grid_x, 256, LDS64);
    else if (dtype == OMCHAT_BF16) OM_LAUNCH_ATTN2((attn2_kernel<bf16, 4, false, 64>), grid_x, 256, LDS64);
    else { omchat_set_error("launch_attn_prefill: bad dtype"); return 1; }
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (hd == 64) {
    if (dtype == OMCHAT_F16) hipLaunchKernelGGL((attn_kernel<f16, 4, 2, 64>), grid, dim3(256), 0, s, p);
    else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL((attn_kernel<bf16, 4, 2, 64>), grid, dim3(256), 0, s, p);
    else { omchat_set_error("launch_attn_prefill: bad dtype"); return 1; }
    OM_LAUNCH_CHECK();
    return 0;
  }
  const int n_rep = a.q_heads / a.kv_heads;
  if (g_attn_v2 && n_rep == 1) {                 // MHA (ViT): 4 waves = 128 queries of one head
    // (248 VGPRs: two 4-wave workgroups per CU, or ONE 8-wave workgroup with the keys split between its wave groups)
    const bool kg2 = use_kg2(2, 1); (void)kg2;
#if OMCHAT_EXPERIMENTS
    if (kg2 && dtype == OMCHAT_F16) OM_LAUNCH_ATTN2((attn2_kernel<f16, 8, false, 128, 2>), grid_x, 512, 2 * LDS128);
    else if (kg2 && dtype == OMCHAT_BF16) OM_LAUNCH_ATTN2((attn2_kernel<bf16, 8, false, 128, 2>), grid_x, 512, 2 * LDS128);
    else
#endif
    if (dtype == OMCHAT_F16) OM_LAUNCH_ATTN2((attn2_kernel<f16, 4, false>), grid_x, 256, LDS128);
    else if (dtype == OMCHAT_BF16) OM_LAUNCH_ATTN2((attn2_kernel<bf16, 4, false>), grid_x, 256, LDS128);
    else { omchat_set_error("launch_attn_prefill: bad dtype"); return 1; }
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (g_attn_v2 && n_rep <= 8) {                 // GQA: the n_rep query heads of one kv head x 32 queries share every K / V tile
    // head split of the heaviest causal blocks (attn2_kernel): only while the launch is at most one workgroup per CU (S <= 2048 for one
    // Qwen2-7B sequence: 61.2 -> 48.7 us at S = 2048 with a quarter of the ranks split; beyond that the pairs' extra CU time costs more than
    // the shorter tail saves: S = 3584 106.8 -> 115.0 us with 8 ranks split)
    const int nqb_ = cdiv(a.Sq, 32), per_rank = a.kv_heads * a.batch;
    int xs = 0;
    if (a.causal && n_rep >= 4 && !a.kv_start && (long)nqb_ * per_rank <= device_cus() && nqb_ >= 16)
      xs = g_attn_hsplit >= 0 ? std::min(g_attn_hsplit, nqb_ / 2) : nqb_ / 4;
    if (g_attn_hsplit > 0 && a.causal && n_rep >= 4 && !a.kv_start && nqb_ >= 16) xs = std::min(g_attn_hsplit, nqb_ / 2);      // forced (tests, A/B)
    const dim3 g2((nqb_ + xs) * per_rank, 1, 1);      // id = (query block rank * batch + sequence) * kv_heads + kv head (attn2_kernel)
    p.nsplit = a.batch; p.tpw = xs;
#define OM_A2(NW_)                                                                                                       \
  do {                                                                                                                   \
    if (dtype == OMCHAT_F16) OM_LAUNCH_ATTN2((attn2_kernel<f16, NW_, true>), g2, NW_ * 64, LDS128);                      \
    else if (dtype == OMCHAT_BF16) OM_LAUNCH_ATTN2((attn2_kernel<bf16, NW_, true>), g2, NW_ * 64, LDS128);               \
    else { omchat_set_error("launch_attn_prefill: bad dtype"); return 1; }                                               \
  } while (0)
    switch (n_rep) { case 2: OM_A2(2); break; case 3: OM_A2(3); break; case 4: OM_A2(4); break; case 5: OM_A2(5); break;
                     case 6: OM_A2(6); break; case 7: OM_A2(7); break; default: OM_A2(8); break; }
#undef OM_A2
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == OMCHAT_F16) hipLaunchKernelGGL((attn_kernel<f16, 4, 2, 128>), grid, dim3(256), 0, s, p);
  else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL((attn_kernel<bf16, 4, 2, 128>), grid, dim3(256), 0, s, p);
  else { omchat_set_error("launch_attn_prefill: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}

size_t attn_decode_ws_bytes(int batch, int q_heads, int max_len) {
  return (size_t)batch * q_heads * cdiv(max_len, KV_TILE) * WS_STRIDE * sizeof(float);
}

// e4m3 cache: which form a decode launch takes (round 6, tuning key 47).  From two one-tile waves per CU on, a wave walks 3 tiles with the next
// tile prefetched (attention + merge, us, one wave per tile / 2 / 3 / 4 tiles per wave: 8.8 k keys 13.9 / 12.2 / 10.1 / 11.8, 16.5 k keys 16.7 /
// 14.5 / 14.3 / 15.3, 33 k keys 21.1 / 18.3 / 17.6 / 18.9, 66 k keys 67.3 / 24.8 / 24.0 / 23.7, b = 4 at 8.8 k keys 23.0 / 19.1 / 15.3 / 16.9;
// 3.7 k keys 8.5 / - / 10.8 / 11.5).  Four such waves per workgroup, folded in LDS (a split = 12 tiles), when that brings the launch under the
// 64 partials of the one-round-trip merge and still fills half the CUs: 33 k keys 17.6 -> 14.9, b = 2 at 33 k keys 24.7 -> 22.5; 16.5 k keys
// 14.3 / 14.6, 8.8 k keys 10.1 -> 14.1 (48 workgroups), 66 k keys 24.0 -> 25.2 (86 partials: the general merge either way) --
// profiles/r06_q_kv8_attn_ab.txt.
static void kv8_plan(int batch, int kv_heads, int L, bool masked, int* tpw, int* nw) {
  *tpw = 1; *nw = 1;
  if (masked) return;      // the masked form exists for the one-tile kernel only (a rare mode)
  const long waves1 = (long)cdiv(L, KV_TILE) * kv_heads * batch;
  if (g_attn_kv8_tpw > 0) { *tpw = g_attn_kv8_tpw & 7; *nw = (g_attn_kv8_tpw & 8) ? 4 : 1; }
  else {
    *tpw = waves1 >= 2L * device_cus() ? 3 : 1;
    const int n4 = cdiv(L, 12 * KV_TILE);
    if (*tpw == 3 && n4 <= 64 && (long)n4 * kv_heads * batch >= device_cus() / 2) *nw = 4;
  }
  if (*tpw < 1 || *tpw > 4) *tpw = 1;
  if (*tpw != 3) *nw = 1;      // the four-wave workgroup is built for three tiles per wave only
}
int g_attn_kv8_fuse = 1;      // omchat_op_set_tuning key 48: 1 = the walking form also rotates q / k and appends + quantises the new token's rows (no rope_kv launch in front), 0 = separate launch
void attn_set_kv8_fuse(int v) { g_attn_kv8_fuse = v; }
// true: launch_attn_decode over the e4m3 cache of this shape takes `rope` / `k_new` / `v_new` / `k16_w` / `v16_w` and does the RoPE + append itself
bool attn_decode_kv8_fuses_rope(int batch, int kv_heads, int L, bool masked) {
  int tpw, nw;
  kv8_plan(batch, kv_heads, L, masked, &tpw, &nw);
  // the folded form holds ~300 registers (one wave per SIMD): only while the launch is one resident round of such waves
  const long waves = (long)cdiv(L, KV_TILE * tpw * nw) * nw * kv_heads * batch;
  return g_attn_kv8_fuse && tpw == 3 && waves <= 4L * device_cus();
}

int launch_attn_decode(int dtype, const AttnDecodeArgs& a, hipStream_t s) {
  OM_CHECK(a.q_heads % a.kv_heads == 0 && a.q_heads / a.kv_heads <= 16, "group size must be <= 16");
  OM_CHECK(a.L > 0 && a.batch > 0, "empty attention");
  // key tiles per wave: one 64-key tile keeps the most waves in flight (single sequences); once the grid is several waves per SIMD deep,
  // 2 or 4 tiles per wave (attn_decode_multi_kernel, 16-bit cache) cut the partials written here and read back by the merge by that factor
  const bool kv8 = a.k_scale != nullptr;
  const long waves1 = (long)cdiv(a.L, KV_TILE) * a.kv_heads * a.batch;
  int tpw = kv8 ? 1 : (waves1 >= 6144 ? 4 : (waves1 >= 3072 ? 2 : 1));
  if (g_attn_tpw > 0 && !kv8) tpw = g_attn_tpw;
  int kv8_nw = 1;
  if (kv8) kv8_plan(a.batch, a.kv_heads, a.L, a.key_mask != nullptr, &tpw, &kv8_nw);
  if (a.key_mask) tpw = 1;      // the masked form exists for the one-tile kernel only (a rare mode)
  // enough keys in the launch (round 4): the LDS-DMA ring form.  One wave per workgroup owns a ring of 32-key tiles; `slots` of them are
  // resident per CU and the split count is chosen so that the launch is one resident round of (nearly) equal splits.  Measured against the
  // kernels above (tools/bench_attn_decode.py, attention + merge, us): b = 4 / 8 / 12 / 32 at 3.7 k keys 12.6 -> 11.9, 19.9 -> 16.3,
  // 27.1 -> 21.2, 52.7 -> 44.0; b = 1 / 4 at 33 k keys 27.1 -> 21.2, 60.2 -> 48.3; a single sequence at 3.7 k keys (464 tiles of 32) stays
  // with the one-tile kernel, whose few waves keep everything in flight at once.  From ~3 tiles per wave on the ring wins.
  const long tiles32 = (long)cdiv(a.L, DMA_TILE) * a.kv_heads * a.batch;
  const bool dma = g_attn_dma && !kv8 && !a.key_mask && a.k_sr == 128 && a.v_sr == 128 && (g_attn_tpw > 1 || (g_attn_tpw == 0 && tiles32 >= 6L * device_cus()));
  // resident waves per CU: four when that leaves every wave 4 .. 8 tiles and there are many (sequence, kv head) streams (b = 8 / 12 at 3.7 k
  // keys: 16.3 / 21.2 us against 17.0 / 22.5 with two); otherwise two -- long splits pipeline better (b = 32: 44.0 vs 46.7 us; b = 2 at 8.8 k
  // keys 13.1 vs 15.3; one sequence at 16 k keys 15.2 vs 17.3)
  int dma_slots = g_attn_dma_slots;
  if (!dma_slots) {
    const int pairs = a.kv_heads * a.batch, tpw4 = cdiv(cdiv(a.L, DMA_TILE), std::max(1, 4 * device_cus() / pairs));
    dma_slots = (pairs >= 16 && tpw4 >= 4 && tpw4 <= 8 && tiles32 < 32L * device_cus()) ? 4 : 2;
  }
  int split_keys = KV_TILE * tpw * kv8_nw;
  if (dma) {
    const int slots = dma_slots * device_cus(), pairs = a.kv_heads * a.batch;
    const int tiles = cdiv(a.L, DMA_TILE);
    const int ns_target = std::max(1, slots / pairs);
    tpw = std::max(2, cdiv(tiles, ns_target));      // >= 2 tiles of 32 keys: never more partials than the workspace holds (one per 64 keys)
    split_keys = DMA_TILE * tpw;
  }
  const int nsplit = cdiv(a.L, split_keys);
  OM_CHECK(a.ws && a.ws_bytes >= attn_decode_ws_bytes(a.batch, a.q_heads, a.L), "workspace too small");
  AttnP p{a.Q, a.K, a.V, nullptr, a.q_sb, a.q_sh, 0, a.k_sb, a.k_sh, a.k_sr, a.v_sb, a.v_sh, a.v_sr, 0, 0, 0,
          a.kv_len, nullptr, a.q_heads, a.kv_heads, 1, a.L, 0, 0, nsplit, a.scale * 1.4426950408889634f, a.ws,
          a.rope, a.pos, a.k_new, a.v_new, a.new_sb, (void*)a.K, (void*)a.V, a.rope_max, a.k_scale, a.v_scale, a.scale_sb, a.scale_sh, tpw,
          a.key_mask, a.mask_sb};
#if OMCHAT_EXPERIMENTS
  p.dbg = a.done_dbg;
#endif
  OM_CHECK(!a.key_mask || (((kv8 && !a.rope) || (!kv8 && a.rope && a.pos)) && !a.kv_len && a.mask_sb % 64 == 0 && a.mask_sb >= a.L),
           "masked decode: fused RoPE with explicit positions (16-bit cache) or rows appended beforehand (e4m3 cache), uniform length, mask rows padded to a multiple of 64");
  const bool kv8_fuse = kv8 && a.rope != nullptr;
  OM_CHECK(!kv8 || a.v_scale, "fp8 KV cache: both scale arrays");
  OM_CHECK(!kv8_fuse || (tpw == 3 && a.k_new && a.v_new && a.k16_w && a.v16_w && !a.key_mask),
           "fp8 KV cache with fused RoPE: the walking form only (attn_decode_kv8_fuses_rope), raw k / v rows and the 16-bit caches to append to");
  OM_CHECK(!a.rope || (a.k_new && a.v_new), "fused RoPE decode needs k_new and v_new (kv_len == null: every sequence holds exactly L keys)");
  OM_CHECK(a.o_pack_nb == 0 || (a.batch <= 16 * a.o_pack_nb && a.o_sh == 128 && a.q_heads % 1 == 0), "packed output: batch <= 16 * NB, head stride 128");
  if (kv8_fuse) { p.k_cache_w = a.k16_w; p.v_cache_w = a.v16_w; }
  if (dma) p.causal = g_attn_dma_rot == 1;
  // LDS request of the ring form: the ring, padded so that exactly g_attn_dma_slots workgroups fit a CU's 160 KiB (the split count is sized
  // for that many; a CU that took more would leave another one short)
  const int dma_lds = std::min(65536, std::max(g_attn_dma_stages * 16384, (160 / dma_slots) * 1024));
  dim3 grid(nsplit, a.kv_heads, a.batch);
  dim3 mgrid(a.q_heads, a.batch);
  if (a.done_flags && (a.batch != 1 || nsplit > 64 || nsplit > g_merge_mid_min)) { omchat_set_error("launch_attn_decode: completion flags exist for the one-round-trip merge only (batch 1, <= 64 splits)"); return 1; }
  if (dtype == OMCHAT_F16) {
    if (a.key_mask && kv8) hipLaunchKernelGGL((attn_decode_kernel<f16, true, true>), grid, dim3(64), 0, s, p);
    else if (a.key_mask) hipLaunchKernelGGL((attn_decode_kernel<f16, false, true>), grid, dim3(64), 0, s, p);
    else if (kv8 && tpw == 2) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<f16, 2, 1>), grid, dim3(64), 0, s, p);
    else if (kv8_fuse && kv8_nw == 4) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<f16, 3, 4, true>), grid, dim3(256), 0, s, p);
    else if (kv8_fuse) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<f16, 3, 1, true>), grid, dim3(64), 0, s, p);
    else if (kv8 && tpw == 3 && kv8_nw == 4) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<f16, 3, 4>), grid, dim3(256), 0, s, p);
    else if (kv8 && tpw == 3) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<f16, 3, 1>), grid, dim3(64), 0, s, p);
    else if (kv8 && tpw == 4) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<f16, 4, 1>), grid, dim3(64), 0, s, p);
    else if (kv8) hipLaunchKernelGGL((attn_decode_kernel<f16, true>), grid, dim3(64), 0, s, p);
    else if (dma && g_attn_dma_stages == 2) hipLaunchKernelGGL((attn_decode_dma_kernel<f16, 2>), grid, dim3(64), dma_lds, s, p);
    else if (dma && g_attn_dma_stages == 3) hipLaunchKernelGGL((attn_decode_dma_kernel<f16, 3>), grid, dim3(64), dma_lds, s, p);
    else if (dma) hipLaunchKernelGGL((attn_decode_dma_kernel<f16, 4>), grid, dim3(64), dma_lds, s, p);
    else if (tpw > 1 && g_attn_klds) hipLaunchKernelGGL((attn_decode_multi_kernel<f16, true>), grid, dim3(64), 0, s, p);
    else if (tpw > 1) hipLaunchKernelGGL((attn_decode_multi_kernel<f16, false>), grid, dim3(64), 0, s, p);
    else hipLaunchKernelGGL((attn_decode_kernel<f16, false>), grid, dim3(64), 0, s, p);
    if (nsplit > g_merge_mid_min && nsplit <= 256) {
      if (g_merge_dg == 2) hipLaunchKernelGGL((attn_merge_mid_kernel<f16, 4, 2>), dim3(a.q_heads, a.batch, 2), dim3(512), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (f16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
      else hipLaunchKernelGGL((attn_merge_mid_kernel<f16, 4, 1>), mgrid, dim3(512), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (f16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
    } else if (nsplit > 256 && nsplit <= 1024) {
      if (g_merge_dg >= 1) hipLaunchKernelGGL((attn_merge_mid_kernel<f16, 8, 4>), dim3(a.q_heads, a.batch, 4), dim3(1024), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (f16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
      else hipLaunchKernelGGL((attn_merge_mid_kernel<f16, 8, 1>), mgrid, dim3(1024), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (f16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
    }
    else hipLaunchKernelGGL(attn_merge_kernel<f16>, mgrid, dim3(128), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (f16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys, a.done_flags, a.done_epoch, a.done_mode, a.done_dbg);
  } else if (dtype == OMCHAT_BF16) {
    if (a.key_mask && kv8) hipLaunchKernelGGL((attn_decode_kernel<bf16, true, true>), grid, dim3(64), 0, s, p);
    else if (a.key_mask) hipLaunchKernelGGL((attn_decode_kernel<bf16, false, true>), grid, dim3(64), 0, s, p);
    else if (kv8 && tpw == 2) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<bf16, 2, 1>), grid, dim3(64), 0, s, p);
    else if (kv8_fuse && kv8_nw == 4) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<bf16, 3, 4, true>), grid, dim3(256), 0, s, p);
    else if (kv8_fuse) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<bf16, 3, 1, true>), grid, dim3(64), 0, s, p);
    else if (kv8 && tpw == 3 && kv8_nw == 4) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<bf16, 3, 4>), grid, dim3(256), 0, s, p);
    else if (kv8 && tpw == 3) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<bf16, 3, 1>), grid, dim3(64), 0, s, p);
    else if (kv8 && tpw == 4) hipLaunchKernelGGL((attn_decode_kv8_walk_kernel<bf16, 4, 1>), grid, dim3(64), 0, s, p);
    else if (kv8) hipLaunchKernelGGL((attn_decode_kernel<bf16, true>), grid, dim3(64), 0, s, p);
    else if (dma && g_attn_dma_stages == 2) hipLaunchKernelGGL((attn_decode_dma_kernel<bf16, 2>), grid, dim3(64), dma_lds, s, p);
    else if (dma && g_attn_dma_stages == 3) hipLaunchKernelGGL((attn_decode_dma_kernel<bf16, 3>), grid, dim3(64), dma_lds, s, p);
    else if (dma) hipLaunchKernelGGL((attn_decode_dma_kernel<bf16, 4>), grid, dim3(64), dma_lds, s, p);
    else if (tpw > 1 && g_attn_klds) hipLaunchKernelGGL((attn_decode_multi_kernel<bf16, true>), grid, dim3(64), 0, s, p);
    else if (tpw > 1) hipLaunchKernelGGL((attn_decode_multi_kernel<bf16, false>), grid, dim3(64), 0, s, p);
    else hipLaunchKernelGGL((attn_decode_kernel<bf16, false>), grid, dim3(64), 0, s, p);
    if (nsplit > g_merge_mid_min && nsplit <= 256) {
      if (g_merge_dg == 2) hipLaunchKernelGGL((attn_merge_mid_kernel<bf16, 4, 2>), dim3(a.q_heads, a.batch, 2), dim3(512), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (bf16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
      else hipLaunchKernelGGL((attn_merge_mid_kernel<bf16, 4, 1>), mgrid, dim3(512), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (bf16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
    } else if (nsplit > 256 && nsplit <= 1024) {
      if (g_merge_dg >= 1) hipLaunchKernelGGL((attn_merge_mid_kernel<bf16, 8, 4>), dim3(a.q_heads, a.batch, 4), dim3(1024), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (bf16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
      else hipLaunchKernelGGL((attn_merge_mid_kernel<bf16, 8, 1>), mgrid, dim3(1024), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (bf16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys);
    }
    else hipLaunchKernelGGL(attn_merge_kernel<bf16>, mgrid, dim3(128), 0, s, a.ws, nsplit, a.q_heads, a.kv_len, a.L, p.c, (bf16*)a.O, a.o_sb, a.o_sh, a.o_pack_nb, split_keys, a.done_flags, a.done_epoch, a.done_mode, a.done_dbg);
  } else { omchat_set_error("launch_attn_decode: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}
