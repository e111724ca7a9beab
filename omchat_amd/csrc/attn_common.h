// Shared device helpers of the attention kernels (attention.hip) and the fused decode launch (fused_decode.hip): the kernel parameter block,
// e4m3 widening, RoPE of one fragment chunk, lane-exchange reductions, the transposed LDS read -- and the body of the one-tile decode
// attention (attn_decode_tile), which both the stand-alone split-KV launch and the fused attention + merge + o_proj launch execute, so that
// the two give the same bits.  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "kernels.h"

namespace {


struct AttnP {
  const void* Q; const void* K; const void* V; void* O;
  int64_t q_sb, q_sh, q_sr, k_sb, k_sh, k_sr, v_sb, v_sh, v_sr, o_sb, o_sh, o_sr;
  const int* kv_len;
  const int* kv_start;
  int q_heads, kv_heads, Sq, Skv, causal, q_pos0, nsplit;
  float c;      // scale * log2(e)
  float* ws;
  // decode with fused RoPE + KV append (one new token per sequence): raw q/k/v of the new token live in the qkv buffer
  const float* rope;      // [max_pos][64][2] or null
  const int* pos;         // [batch] position of the new token (= kv_len - 1)
  const void* k_new; const void* v_new; int64_t new_sb;   // raw k / v rows, batch stride (elements); head stride 128
  void* k_cache_w; void* v_cache_w;                          // writable views of K / V (same strides as K / V)
  int rope_max;
  const float* k_scale; const float* v_scale; int64_t scale_sb, scale_sh;      // fp8 KV cache (decode only)
  int tpw;                // decode: key tiles per wave (attn_decode_multi_kernel): a split is tpw x KV_TILE keys
  // decode of a padded batch as the reference computes it (omchat_decode_step_masked): key j of sequence b is visible iff
  // key_mask[b * mask_sb + j] != 0 (rows zero-padded to mask_sb, a multiple of 64), and the new token is rotated to pos[b] -- not to the
  // slot it is appended at (kv_len - 1)
  const unsigned char* key_mask; int64_t mask_sb;
  // prefill, MHA (the ViT, round 6): the Q half of InternAttention's joint-head q / k RMSNorm (modeling_intern_vit.py:143-148) applied where
  // the kernel loads Q -- q = T(T(w_q * T(q * rstd_q)) * qn_scale) with rstd_q = rsqrt(qn_sumsq[m * qn_stride] / qn_dim + qn_eps),
  // m = batch * Sq + query: the row's sum of squares over all heads' q channels, finished from the per-wave-column partials the qkv GEMM's
  // epilogue left (gemm.hip EPI_NONE_STATS -> launch_stats_finish).  null = Q is used as stored.
  const float* qn_sumsq = nullptr; int qn_stride = 0, qn_dim = 0; const void* qn_w = nullptr; float qn_eps = 0.f, qn_scale = 1.f;
  int peel_last = 0;      // prefill, MHA: a key count of whole tiles + 1 folds the last key into the online softmax's initial state (attn2_kernel)
#if OMCHAT_EXPERIMENTS
  unsigned long long* dbg;      // measurement only: clock stamps of the layer (model.hip dbg_stamps), else null
#endif
};

// 8 e4m3 bytes -> 8 T (exact widening)
template <typename T> __device__ __forceinline__ typename V8<T>::type widen8(u32x2 w);
template <> __device__ __forceinline__ bf16x8 widen8<bf16>(u32x2 w) {
  typedef bf16 v2 __attribute__((ext_vector_type(2)));
  const v2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, 1.0f, true);
  const v2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, 1.0f, true);
  return (bf16x8){a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}
template <> __device__ __forceinline__ f16x8 widen8<f16>(u32x2 w) {
  typedef f16 v2 __attribute__((ext_vector_type(2)));
  const v2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, 1.0f, true);
  const v2 c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, 1.0f, true);
  return (f16x8){a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// RoPE of one 8-element chunk (rotate-half, modeling_qwen2.py:105-135) with the reference's rounding (N11):
// x = own chunk, o = partner chunk 64 elements away, first half gets -partner*sin, second half +partner*sin
template <typename T>
__device__ __forceinline__ typename V8<T>::type rope_chunk(typename V8<T>::type x, typename V8<T>::type o, const float* cs, bool second_half) {
  // No contraction here: the reference rounds each product to the 16-bit type before the add (x * cos, rotate_half(x) * sin, then +).  The
  // compiler narrows this expression to 16-bit fmul / fadd, and under the default -ffp-contract=fast it may then fuse one product into the
  // add (an fma that skips that product's rounding) -- it did so in one kernel and not in another: 1-ulp differences in rotated q / k.
#pragma clang fp contract(off)
  typename V8<T>::type r;
  const float sgn = second_half ? 1.f : -1.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float co = rnd<T>(cs[2 * j]), si = rnd<T>(cs[2 * j + 1]);
    r[j] = fromf<T>(rnd<T>(tof(x[j]) * co) + rnd<T>(sgn * tof(o[j]) * si));
  }
  return r;
}

constexpr int KV_TILE = 64;
constexpr int WS_STRIDE = 132;   // 128 O values + m + l (+2 pad, keeps 16-B alignment)
constexpr float NEG_BIG = -1e30f;      // a masked score
constexpr float M_FLOOR = -1e20f;      // initial running-max reference: far below any real score, far above NEG_BIG, so that a row whose
                                       // keys are ALL masked (padded query rows of a left-padded batch) gets p = exp2(-huge) = 0, l = 0 and an
                                       // output of exactly 0 -- never inf - inf.  Its V rows feed later layers as masked keys: 0 * finite.
constexpr float RESCALE_LOG2 = 8.f;   // prefill: running-max reference moves only on a > 2^8 overshoot

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ s16x4 tr_read(const char* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds_addr));
}

// lane ^ 32 / lane ^ 16 reductions on the VALU (v_permlane32_swap / v_permlane16_swap): swapping a value with itself gives
// {own, partner}; no LDS round trip (ds_bpermute) on the softmax critical path
__device__ __forceinline__ float max_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------------------
// One 64-key tile of the decode attention for the n_rep query heads of one kv head (the body of attn_decode_kernel): ALL global loads of
// the tile (K as MFMA A fragments straight to registers, V for the LDS transpose image, Q, RoPE table row) are issued at once -- one HBM
// round trip -- then S^T from registers, softmax, V -> LDS, PV through ds_read_b64_tr_b16.  With `rope` set it also rotates q in registers
// and, in the split that owns the new position, rotates k and appends k / v to the cache.  Results stay in registers: lane (fc = lane & 15
// = head in the group, fg = lane >> 4) holds O[d = 16 dn + 4 fg + r] in o[dn][r], the tile's score maximum and exponential sum.
// WAVE_ONLY: the caller is ONE wave of a larger workgroup (fused_decode.hip): LDS hand-over inside the wave instead of a workgroup
// barrier, and one raw workgroup barrier right after the loads are issued (the other waves of the workgroup start THEIR loads behind it,
// so that this tile's K / V are first in the CU's in-order memory pipe).
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool KV8, bool WAVE_ONLY, bool MASKED = false>
__device__ __forceinline__ void attn_decode_tile(const AttnP& p, int split, int kvh, int b, int kv_len, char* Vs, int lane, f32x4 (&o)[8], float& mx_out,
                                                 float& l_out) {
  typedef typename V8<T>::type frag_t;
  const int fc = lane & 15, fg = lane >> 4;
  const int n_rep = p.q_heads / p.kv_heads, hq0 = kvh * n_rep;
  const int key0 = split * KV_TILE;
  const bool fuse = WAVE_ONLY ? true : p.rope != nullptr;      // the fused launch always rotates and appends (its launcher checks)
  const int pp = kv_len - 1;                  // position of the token being appended (fuse)
  const int pr = (MASKED && p.pos) ? p.pos[b] : pp;      // RoPE position of the new token (MASKED: given, omchat_arch.py:70 sum(mask) - 1; none with the e4m3 cache: rope_kv ran before)
  const int pt = pr < p.rope_max ? (pr > 0 ? pr : 0) : p.rope_max - 1;
  const T* Kg = (const T*)p.K + b * p.k_sb + kvh * p.k_sh;
  const T* Vg = (const T*)p.V + b * p.v_sb + kvh * p.v_sh;
  const T* kn = fuse ? (const T*)p.k_new + b * p.new_sb + kvh * 128 : nullptr;
  const T* vn = fuse ? (const T*)p.v_new + b * p.new_sb + kvh * 128 : nullptr;

  // ---- q and the RoPE table row of the new position are requested FIRST (round 4): vector memory returns in order, so behind the tile's 32
  // K / V loads they could not be used before the whole tile has arrived and the rotation of q (~270 VALU instructions of this single wave)
  // sat on the critical path behind the HBM round trip; in front of them it runs while the tile is still on its way.  No runtime branch
  // around the table loads (a conditional block would end in a wait for everything issued inside it): without RoPE they read q's own bytes.
  frag_t qf[4];
  {
    const int hh = fc < n_rep ? fc : n_rep - 1;
    const T* qp = (const T*)p.Q + b * p.q_sb + (hq0 + hh) * p.q_sh;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = ld8<T>(qp + ds * 32 + fg * 8);
  }
  // the new token's raw key row, d = 32 ds + 8 fg + j (every tile row >= pp reads it): loaded and rotated ahead of the tile as well, so that
  // the split that owns the new position only SELECTS it once its tile has arrived (its rotation used to be the tail of the launch)
  frag_t knr[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) knr[ds] = ld8<T>((fuse ? kn : (const T*)p.Q) + ds * 32 + fg * 8);
  float csv[2][16];
  {
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      const f32x4* cs = reinterpret_cast<const f32x4*>(fuse ? p.rope + ((size_t)pt * 64 + ds * 32 + fg * 8) * 2 : (const float*)p.Q);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) { const f32x4 v = cs[q4]; csv[ds][4 * q4] = v[0]; csv[ds][4 * q4 + 1] = v[1]; csv[ds][4 * q4 + 2] = v[2]; csv[ds][4 * q4 + 3] = v[3]; }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- issue every load of the tile
  frag_t kf[4][4];                            // A operand of S^T: key = key0 + 16*kt + fc, d = 32*ds + 8*fg + j
  bool kfresh[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = key0 + kt * 16 + fc;
    kfresh[kt] = fuse && key >= pp;           // not in the cache yet (or clamped onto it)
    if constexpr (KV8) {                      // e4m3 cache bytes, widened exactly; the per-key scale multiplies the score below
      // (plain loads: a 128-byte e4m3 row is fetched by four instructions of 32 bytes each, and with non-temporal loads the line can leave
      // L2 between them -- configs[4] decode 2.27 -> 2.35 ms per token, profiles/r04_p)
      const unsigned char* src = (const unsigned char*)p.K + b * p.k_sb + kvh * p.k_sh + (int64_t)(key < kv_len ? key : kv_len - 1) * p.k_sr;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) kf[kt][ds] = widen8<T>(*reinterpret_cast<const u32x2*>(src + ds * 32 + fg * 8));
    } else {
      // (rows >= pp under fused RoPE are replaced by the rotated new key below: their lanes read the last cached row, or row 0 of an empty cache)
      const int kc = fuse ? (key < pp ? key : (pp > 0 ? pp - 1 : 0)) : (key < kv_len ? key : kv_len - 1);
      const T* src = Kg + (int64_t)kc * p.k_sr;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) kf[kt][ds] = ld8s<T>(src + ds * 32 + fg * 8);
    }
  }
  frag_t vreg[16];                            // V^T image source: chunk idx = i*64 + lane -> row = 4*i + fg, ch = fc
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int key = key0 + i * 4 + fg;
    const bool fresh = fuse && key >= pp;
    if constexpr (KV8) {
      const unsigned char* src = (const unsigned char*)p.V + b * p.v_sb + kvh * p.v_sh + (int64_t)(key < kv_len ? key : kv_len - 1) * p.v_sr;
      vreg[i] = widen8<T>(*reinterpret_cast<const u32x2*>(src + fc * 8));
    } else {
      const T* src = fresh ? vn : Vg + (int64_t)(key < kv_len ? key : kv_len - 1) * p.v_sr;
      vreg[i] = ld8s<T>(src + fc * 8);
    }
  }
  // fp8 cache: scales of the keys this lane's score registers hold (key0 + 16 kt + 4 fg + r), clamped like the rows
  f32x4 ksc[4], vsc[4];
  if constexpr (KV8) {
    const float* ksp = p.k_scale + b * p.scale_sb + kvh * p.scale_sh;
    const float* vsp = p.v_scale + b * p.scale_sb + kvh * p.scale_sh;
    if (key0 + KV_TILE <= kv_len) {      // whole tile (wave-uniform): the four keys of a score fragment are one 16-byte load (8 loads per tile instead of 32)
      typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const f32x4u kv = *reinterpret_cast<const f32x4u*>(ksp + key0 + kt * 16 + 4 * fg), vv = *reinterpret_cast<const f32x4u*>(vsp + key0 + kt * 16 + 4 * fg);
        ksc[kt] = (f32x4){kv[0], kv[1], kv[2], kv[3]}; vsc[kt] = (f32x4){vv[0], vv[1], vv[2], vv[3]};
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + kt * 16 + 4 * fg + r, kc = key < kv_len ? key : kv_len - 1;
          ksc[kt][r] = ksp[kc]; vsc[kt][r] = vsp[kc];
        }
    }
  }
  if constexpr (WAVE_ONLY) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();               // raw: the loads above stay in flight; the workgroup's other waves issue theirs behind them
    __builtin_amdgcn_sched_barrier(0);
  }
  if (fuse) {
    // rotate-half partner of d = 32*ds + 8*fg + j is fragment ds ^ 2 of the same lane (q and fresh k alike).  q first, all of it: it needs
    // only the loads issued in front of the tile; the fresh key rows wait for the tile itself
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      const frag_t lo = qf[ds], hi = qf[ds + 2];
      qf[ds] = rope_chunk<T>(lo, hi, csv[ds], false);
      qf[ds + 2] = rope_chunk<T>(hi, lo, csv[ds], true);
    }
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      const frag_t kl = knr[ds], kh = knr[ds + 2];
      knr[ds] = rope_chunk<T>(kl, kh, csv[ds], false);
      knr[ds + 2] = rope_chunk<T>(kh, kl, csv[ds], true);
    }
    __builtin_amdgcn_sched_barrier(0);
    // rows >= pp of the tile (per lane) take the rotated new key; the rows in front of it came from the cache, rotated when they were appended
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) kf[kt][ds] = kfresh[kt] ? knr[ds] : kf[kt][ds];
    // append (N14): the lanes that hold the real row pp write it (4 lanes x 4 chunks for k, 16 lanes x 1 chunk for v)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (key0 + kt * 16 + fc == pp) {
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
          st8<T>((T*)p.k_cache_w + b * p.k_sb + kvh * p.k_sh + (int64_t)pp * p.k_sr + ds * 32 + fg * 8, kf[kt][ds]);
      }
  }

  // ---- S^T = K Q^T from registers
  f32x4 s[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) s[kt] = mfma16(kf[kt][ds], qf[ds], s[kt]);
  }
  // ---- softmax over this split (keys >= kv_len masked; MASKED: also the keys the sequence's mask hides)
  unsigned vis[4];
  if constexpr (MASKED) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) vis[kt] = *reinterpret_cast<const unsigned*>(p.key_mask + b * p.mask_sb + key0 + kt * 16 + 4 * fg);
  }
  float mx = NEG_BIG;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv = s[kt][r];
      if constexpr (KV8) sv *= ksc[kt][r];
      bool seen = key0 + kt * 16 + 4 * fg + r < kv_len;
      if constexpr (MASKED) seen = seen && ((vis[kt] >> (8 * r)) & 0xffu) != 0u;
      const float v = seen ? sv : NEG_BIG;
      s[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = max_xor32(max_xor16(mx));
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
      if constexpr (KV8) e[j] *= vsc[2 * ks + (j >> 2)][j & 3];      // V = scale * e4m3: fold the per-key scale into P
    }
    pf[ks] = __builtin_convertvector(e, frag_t);
  }
  const float l = sum_xor32(sum_xor16(psum));

  // ---- V -> LDS transpose image (chunk' = chunk ^ ((row & 7) << 1)), append the fresh v row
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = i * 4 + fg;
    *reinterpret_cast<frag_t*>(Vs + row * 256 + ((fc ^ ((row & 7) << 1)) << 4)) = vreg[i];
    if (fuse && key0 + row == pp)
      st8<T>((T*)p.v_cache_w + b * p.v_sb + kvh * p.v_sh + (int64_t)pp * p.v_sr + fc * 8, vreg[i]);
  }
  if constexpr (WAVE_ONLY) { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
  else __syncthreads();

  // ---- O^T = V^T P^T
  const int tq = fc >> 2, tp = fc & 3;
  const int vrow_lo = 4 * fg + tq;
  const int vswz = ((vrow_lo & 7) << 1);
#pragma unroll
  for (int dn = 0; dn < 8; ++dn) o[dn] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
  mx_out = mx;
  l_out = l;
}

}  // namespace
