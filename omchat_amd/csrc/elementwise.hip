// HBM-bound row kernels of the hot path: RMSNorm, ViT joint-head q/k norm, RoPE + KV-cache append, patch im2col,
// CLS/pos-embed assembly, splice gather, argmax, synthetic fill.  All 16-bit traffic is 16 B per lane.
#include "kernels.h"
#include <limits.h>

namespace {

constexpr int NORM_THREADS = 256;
constexpr int NORM_MAXC = 8;     // chunks of 8 elements per thread -> H <= 256*8*8 = 16384

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NORM_THREADS / 64; ++w) t += red[w];
  __syncthreads();
  return t;
}

// ---------------------------------------------------------------------------------------------------------
// RMSNorm: InternRMSNorm.forward (modeling_intern_vit.py:39-44) == Qwen2RMSNorm.forward (modeling_qwen2.py:247-252)
// fp32 statistics, y = T( w * T(x * rsqrt(mean(x^2) + eps)) )   (two roundings, N2)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_kernel(const T* x, int ldx, const T* w, T* y, int ldy, int H, float eps, int pack_nb) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  const T* xr = x + (size_t)row * ldx;
  T* yr = y + (size_t)row * ldy;
  const int nchunk = H >> 3;
  v8 xv[NORM_MAXC];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      xv[i] = ld8<T>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = tof(xv[i][j]); ss += f * f; }
    }
  }
  const float inv = rsqrtf(block_sum(ss, red) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      const v8 wv = ld8<T>(w + c * 8);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[j]) * rnd<T>(tof(xv[i][j]) * inv));
      st8<T>(pack_nb ? y + packed_x_index(row, c * 8, pack_nb) : yr + c * 8, o);
    }
  }
}

#if OMCHAT_EXPERIMENTS
// The same RMSNorm with one WAVE per row (round 5; prefill / ViT row counts, H <= 4096): every load of the row and of the weight is issued before the
// first use, the statistics need no LDS and no barrier, and all rows of a 3 k-row launch are resident at once.  A lane plays the four threads
// {lane, lane + 64, lane + 128, lane + 192} of rmsnorm_kernel's workgroup and the partial sums are combined in that kernel's order, so the bits are the same.
// MEASURED: no faster (rocprof, configs[1]: 10.4 us against 9.3 (ViT) / 10.7 (prefill) per launch; q / k norm 18.3 against 18.5; ViT 38.0-38.4 ms either way):
// these launches move 39 / 79 MB of read + write traffic at the chip's copy rate, not at a latency chain's.  Experiments build only (tuning key 40).
constexpr int NORM_WAVE_H = 4096;
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_wave_kernel(const T* x, int ldx, const T* w, T* y, int ldy, int rows, int H, float eps) {
  typedef typename V8<T>::type v8;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (size_t)row * ldx;
  T* yr = y + (size_t)row * ldy;
  const int nchunk = H >> 3;
  v8 xv[2][4], wv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int c = i * 256 + v * 64 + lane; if (c < nchunk) xv[i][v] = ld8<T>(xr + c * 8); }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int c = i * 256 + v * 64 + lane; if (c < nchunk) wv[i][v] = ld8<T>(w + c * 8); }
  float tot = 0.f;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i * 256 + v * 64 + lane;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = tof(xv[i][v][j]); ss += f * f; }
      }
    }
    tot += wave_sum(ss);
  }
  const float inv = rsqrtf(tot / (float)H + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c = i * 256 + v * 64 + lane;
      if (c < nchunk) {
        v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[i][v][j]) * rnd<T>(tof(xv[i][v][j]) * inv));
        st8<T>(yr + c * 8, o);
      }
    }
}

#endif

// ---------------------------------------------------------------------------------------------------------
// fp8 x fp8 prefill (BASELINE configs[4]): the activation operand of the GEMM is e4m3 with one scale per token (row).
//   NORM: y = T(w * T(x * rsqrt(mean(x^2) + eps)))  (the reference's RMSNorm, same roundings), then s = absmax(y) / 448 (1 for a zero row),
//         y8 = e4m3_rne(y / s).      !NORM: plain per-row quantisation of x.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NORM_THREADS / 64; ++w) t = fmaxf(t, red[w]);
  __syncthreads();
  return t;
}

template <typename T, bool NORM>
__global__ __launch_bounds__(NORM_THREADS) void rows_q8_kernel(const T* x, int ldx, const T* w, unsigned char* y8, int ldy, float* scale, int H, float eps) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  const T* xr = x + (size_t)row * ldx;
  const int nchunk = H >> 3;
  float xv[NORM_MAXC][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      const v8 v = ld8<T>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { xv[i][j] = tof(v[j]); ss += xv[i][j] * xv[i][j]; }
    }
  }
  float amax = 0.f;
  if constexpr (NORM) {
    const float inv = rsqrtf(block_sum(ss, red) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nchunk) {
        const v8 wv = ld8<T>(w + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[i][j] = rnd<T>(tof(wv[j]) * rnd<T>(xv[i][j] * inv)); amax = fmaxf(amax, fabsf(xv[i][j])); }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(xv[i][j]));
      }
    }
  }
  amax = block_max(amax, red);
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  if (threadIdx.x == 0) scale[row] = sc;
  unsigned char* yr = y8 + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = xv[i][j] / sc;
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      u32x2 o2 = {(unsigned)lo, (unsigned)hi};
      *reinterpret_cast<u32x2*>(yr + c * 8) = o2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm (InternViT-300M: NORM2FN['layer_norm'] = nn.LayerNorm, intern_vit_300m/modeling_intern_vit.py:61-64,209-210):
// fp32 mean / biased variance over the row, y = T((x - mean) * rsqrt(var + eps) * w + b)   (one rounding, as ATen)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void layernorm_kernel(const T* x, int ldx, const T* w, const T* b, T* y, int ldy, int H, float eps) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  const T* xr = x + (size_t)row * ldx;
  T* yr = y + (size_t)row * ldy;
  const int nchunk = H >> 3;
  float xv[NORM_MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      const v8 v = ld8<T>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { xv[i][j] = tof(v[j]); sum += xv[i][j]; }
    }
  }
  const float mean = block_sum(sum, red) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xv[i][j] - mean; sq += d * d; }
    }
  }
  const float inv = rsqrtf(block_sum(sq, red) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      const v8 wv = ld8<T>(w + c * 8), bv = ld8<T>(b + c * 8);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>((xv[i][j] - mean) * inv * tof(wv[j]) + tof(bv[j]));
      st8<T>(yr + c * 8, o);
    }
  }
}

// decode tail of o_proj / down_proj: add the split-K fp32 slices (fixed order), round like the reference
// (T(linear) then T(residual + .), modeling_qwen2.py:283-296), then the NEXT RMSNorm of the same row, in one pass.
template <typename T, int KS>
__global__ __launch_bounds__(NORM_THREADS) void resid_rmsnorm_kernel(T* x, int ldx, const float* part, int rows, const T* w, T* xn,
                                                                     int ldn, int H, float eps, int pack_nb) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  T* xr = x + (size_t)row * ldx;
  const int nchunk = H >> 3;
  float xv[NORM_MAXC][8];
  float ss = 0.f;
  // the norm weights do not depend on the reduction: their loads join the first (and only) round trip of this latency-bound launch
  v8 wv[NORM_MAXC];
  if (w) {
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nchunk) wv[i] = ld8<T>(w + c * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      const v8 xi = ld8<T>(xr + c * 8);
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      f32x4 p0[KS], p1[KS];            // all slices in flight at once (KS is a compile-time constant)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float* pp = part + ((size_t)s * rows + row) * H + c * 8;
        p0[s] = *reinterpret_cast<const f32x4*>(pp); p1[s] = *reinterpret_cast<const f32x4*>(pp + 4);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)       // fixed summation order: deterministic
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] += p0[s][j]; a[4 + j] += p1[s][j]; }
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rnd<T>(tof(xi[j]) + rnd<T>(a[j]));
        xv[i][j] = v; o[j] = fromf<T>(v); ss += v * v;
      }
      st8<T>(xr + c * 8, o);
    }
  }
  if (!w) return;
  const float inv = rsqrtf(block_sum(ss, red) / (float)H + eps);
  T* nr = xn + (size_t)row * ldn;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[i][j]) * rnd<T>(xv[i][j] * inv));
      st8<T>(pack_nb ? xn + packed_x_index(row, c * 8, pack_nb) : nr + c * 8, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// ViT q/k norm over ALL heads of a token (modeling_intern_vit.py:143-146), in place on the fused qkv row, followed by
// q * head_dim^-0.5 rounded in the storage type exactly where _naive_attn does it (:148).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void vit_qknorm_kernel(T* qkv, int ld, const T* wq, const T* wk, int C, int C_total,
                                                                  float eps, float q_scale, const float* sumsq_in, int part0) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  const int nchunk = C >> 3;
  {
    const int part = blockIdx.y + part0;       // 0 = q, 1 = k: one workgroup each (twice the rows in flight, half the serial chain); part0 = 1: K alone
    T* xr = qkv + (size_t)row * ld + part * C;
    const T* w = part == 0 ? wq : wk;
    v8 xv[NORM_MAXC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nchunk) {
        xv[i] = ld8<T>(xr + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = tof(xv[i][j]); ss += f * f; }
      }
    }
    float tot = sumsq_in ? sumsq_in[row * 2 + part] : block_sum(ss, red);
    const float inv = rsqrtf(tot / (float)C_total + eps);
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nchunk) {
        const v8 wv = ld8<T>(w + c * 8);
        v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = rnd<T>(tof(wv[j]) * rnd<T>(tof(xv[i][j]) * inv));
          if (part == 0) v = v * q_scale;
          o[j] = fromf<T>(v);
        }
        st8<T>(xr + c * 8, o);
      }
    }
  }
}

#if OMCHAT_EXPERIMENTS
// one wave per (row, q | k) -- see rmsnorm_wave_kernel: same bits as vit_qknorm_kernel, all loads up front, no barrier
template <typename T>
__global__ __launch_bounds__(256) void vit_qknorm_wave_kernel(T* qkv, int ld, const T* wq, const T* wk, int rows, int C, int C_total,
                                                              float eps, float q_scale, const float* sumsq_in) {
  typedef typename V8<T>::type v8;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (row >= rows) return;
  const int part = blockIdx.y;
  const int nchunk = C >> 3;
  T* xr = qkv + (size_t)row * ld + part * C;
  const T* w = part == 0 ? wq : wk;
  v8 xv[2][4], wv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int c = i * 256 + v * 64 + lane; if (c < nchunk) xv[i][v] = ld8<T>(xr + c * 8); }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int c = i * 256 + v * 64 + lane; if (c < nchunk) wv[i][v] = ld8<T>(w + c * 8); }
  float tot = 0.f;
  if (sumsq_in) tot = sumsq_in[row * 2 + part];
  else {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = i * 256 + v * 64 + lane;
        if (c < nchunk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = tof(xv[i][v][j]); ss += f * f; }
        }
      }
      tot += wave_sum(ss);
    }
  }
  const float inv = rsqrtf(tot / (float)C_total + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c = i * 256 + v * 64 + lane;
      if (c < nchunk) {
        v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = rnd<T>(tof(wv[i][v][j]) * rnd<T>(tof(xv[i][v][j]) * inv));
          if (part == 0) t = t * q_scale;
          o[j] = fromf<T>(t);
        }
        st8<T>(xr + c * 8, o);
      }
    }
}

#endif

// ---------------------------------------------------------------------------------------------------------
// Round 6: the ViT's norms folded into its GEMMs (model.hip vit_run, fused form).  The sum of squares of a row is left by the PRODUCING GEMM's
// epilogue as one partial per wave-column block ("slot", gemm.hip EPI_*_STATS); these kernels are what is left of the norm launches:
//   stats_finish_kernel   rstd[m] = rsqrt(sum of a row's slots / dim + eps)  -> the next GEMM's row scale (InternRMSNorm statistics, :39-44)
//   row_sumsq_kernel      the statistics of a residual stream no GEMM produced (layer 0: the embeddings)
//   vit_knorm_stats_kernel  k = T(w_k * T(k * rstd_k)) in place with rstd_k from the qkv GEMM's slots (the K half of the joint q / k norm,
//                         modeling_intern_vit.py:143-146; the Q half is applied where the attention kernel loads Q: attention.hip)
// Slots are summed in slot order by one thread (or one fixed tree): deterministic.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fold_cols_kernel(const T* W, const T* n, T* out, long chunks, int cols8) {
  typedef typename V8<T>::type v8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long)gridDim.x * 256) {
    const int c = (int)(i % cols8);
    const v8 w = ld8<T>(W + i * 8), nv = ld8<T>(n + c * 8);
    v8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(w[j]) * tof(nv[j]));
    st8<T>(out + i * 8, o);
  }
}

__global__ __launch_bounds__(256) void stats_finish_kernel(const float* stats, int ld, int slot0, int nslots, int ngroups, int rows, float inv_dim, float eps,
                                                           float* out) {
  // slot-major statistics [slot][ld]: one thread per row walks its slots in slot order, every wave load one contiguous run
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  for (int g = 0; g < ngroups; ++g) {
    const float* p = stats + (size_t)(slot0 + g * nslots) * ld + row;
    float t = 0.f;
    for (int c0 = 0; c0 < nslots; c0 += 16) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = c0 + k < nslots ? p[(size_t)(c0 + k) * ld] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += v[k];
    }
    out[(size_t)row * ngroups + g] = inv_dim > 0.f ? rsqrtf(t * inv_dim + eps) : t;
  }
}

template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void row_sumsq_kernel(const T* x, int ldx, int H, float* stats) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  const T* xr = x + (size_t)row * ldx;
  float ss = 0.f;
  for (int c = threadIdx.x; c < (H >> 3); c += NORM_THREADS) {
    const v8 v = ld8<T>(xr + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = tof(v[j]); ss += f * f; }
  }
  const float t = block_sum(ss, red);
  if (threadIdx.x == 0) stats[row] = t;      // slot 0 of the slot-major statistics
}

// K half of the joint-head q / k norm straight from the qkv GEMM's statistics slots (round 6): wave 0 sums the row's k slots, wave 1 its q slots
// (<= 64 each, one per lane, a fixed shuffle tree), k = T(w_k * T(k * rstd_k)) in place, and the q sum is left in sumsq_q[row] for the attention
// kernel, which applies the Q half where it loads Q.  No finishing launch: the row's loads and both slot loads are requested together.
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void vit_knorm_slots_kernel(T* k, int ld, const T* wk, int C, int C_total, float eps, const float* stats, int stats_ld,
                                                                       int nslots, float* sumsq_q) {
  typedef typename V8<T>::type v8;
  __shared__ float red[2];
  const int row = blockIdx.x;
  const int nchunk = C >> 3;
  T* xr = k + (size_t)row * ld;
  v8 xv[NORM_MAXC], wv[NORM_MAXC];
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) { xv[i] = ld8<T>(xr + c * 8); wv[i] = ld8<T>(wk + c * 8); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 2) {       // q slots [0, nslots), k slots [nslots, 2 nslots)
    const float part = lane < nslots ? stats[(size_t)((wave == 0 ? nslots : 0) + lane) * stats_ld + row] : 0.f;      // slot-major
    const float t = wave_sum(part);
    if (lane == 0) red[wave] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) sumsq_q[row] = red[1];
  const float inv = rsqrtf(red[0] / (float)C_total + eps);
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[i][j]) * rnd<T>(tof(xv[i][j]) * inv));
      st8<T>(xr + c * 8, o);
    }
  }
}

// Sequence-parallel tensor parallelism (round 6, model.hip gemm_sp): the rows of the residual stream a rank OWNS after the reduce-scatter of a
// row-parallel projection.  x = T(x + y) in place (y = the 16-bit sum over the ranks of the projection's partials, bias / layer scale already inside
// them), then xn = RMSNorm(x) * w, or LayerNorm(x) * w + b when b is given, or nothing when w is null -- Qwen2RMSNorm / InternRMSNorm / nn.LayerNorm with
// the rounding points of rmsnorm_kernel / layernorm_kernel.  One workgroup per row.
template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void resid16_norm_kernel(T* x, int ldx, const T* y, int ldy, const T* w, const T* b, T* xn, int ldn, int H, float eps) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  T* xr = x + (size_t)row * ldx;
  const T* yr = y + (size_t)row * ldy;
  const int nchunk = H >> 3;
  float xv[NORM_MAXC][8];
  float ss = 0.f, sm = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      const v8 a = ld8<T>(xr + c * 8), d = ld8<T>(yr + c * 8);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = rnd<T>(tof(a[j]) + tof(d[j])); xv[i][j] = f; o[j] = fromf<T>(f); ss += f * f; sm += f; }
      st8<T>(xr + c * 8, o);
    }
  }
  if (!w) return;
  if (b) {      // LayerNorm (intern_vit_300m NORM2FN): mean, then the variance of the centred values, as layernorm_kernel
    const float mean = block_sum(sm, red) / (float)H;
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = xv[i][j] - mean; sv += d * d; }
      }
    }
    const float inv = rsqrtf(block_sum(sv, red) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NORM_MAXC; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nchunk) {
        const v8 wv = ld8<T>(w + c * 8), bv = ld8<T>(b + c * 8);
        v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fromf<T>((xv[i][j] - mean) * inv * tof(wv[j]) + tof(bv[j]));
        st8<T>(xn + (size_t)row * ldn + c * 8, o);
      }
    }
    return;
  }
  const float inv = rsqrtf(block_sum(ss, red) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NORM_MAXC; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nchunk) {
      const v8 wv = ld8<T>(w + c * 8);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[j]) * rnd<T>(xv[i][j] * inv));
      st8<T>(xn + (size_t)row * ldn + c * 8, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NORM_THREADS) void vit_qk_sumsq_kernel(const T* qkv, int ld, int C, float* out) {
  typedef typename V8<T>::type v8;
  __shared__ float red[NORM_THREADS / 64];
  const int row = blockIdx.x;
  for (int part = 0; part < 2; ++part) {
    const T* xr = qkv + (size_t)row * ld + part * C;
    float ss = 0.f;
    for (int c = threadIdx.x; c < (C >> 3); c += NORM_THREADS) {
      const v8 v = ld8<T>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = tof(v[j]); ss += f * f; }
    }
    const float tot = block_sum(ss, red);
    if (threadIdx.x == 0) out[row * 2 + part] = tot;
  }
}

// ---------------------------------------------------------------------------------------------------------
// RoPE (rotate-half, modeling_qwen2.py:105-135; cos/sin fp32 table cast to T before the multiply, N11) on q (in place)
// and k (written to the cache, N14), raw v copied to the cache.  One 16-lane group per (row, head): lane c owns
// elements [8c, 8c+8) and pairs with the chunk 64 elements away.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(T* qkv, int ld, int rows, int S, int nq, int nkv, const int* pos, int pos0,
                                                      const float* cos_sin, int max_pos, T* kc, T* vc, int64_t c_sb, int64_t c_sh,
                                                      unsigned char* kq8, unsigned char* vq8, float* ks, float* vs, int64_t s_sb, int64_t s_sh, int slot0) {
  typedef typename V8<T>::type v8;
  const int nh = nq + 2 * nkv;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long item = gid >> 4;                       // (row, head)
  const int c = gid & 15;                           // 8-element chunk inside the head
  if (item >= (long)rows * nh) return;
  const int row = (int)(item / nh), h = (int)(item % nh);
  const int bi = row / S;
  const int pr = pos ? pos[row] : pos0 + (row % S);      // RoPE position
  const int pp = slot0 >= 0 ? slot0 + (row % S) : pr;     // cache slot
  T* src = qkv + (size_t)row * ld + h * 128;
  // fp8 KV cache: the 16-lane group holds the whole 128-element row: s = absmax / 448 (1 for a zero row), bytes = e4m3_rne(x / s) --
  // attention.hip: kv_quant_kernel's arithmetic on the 16-bit values just stored, so the same bytes
  auto quant_row = [&](const v8& r, unsigned char* dst8, float* sc_out) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(tof(r[j])));
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float sc = m > 0.f ? m / 448.0f : 1.0f;
    typedef unsigned u32x2q __attribute__((ext_vector_type(2)));
    u32x2q w;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(tof(r[4 * k]) / sc, tof(r[4 * k + 1]) / sc, 0, false);
      const int hi = __builtin_amdgcn_cvt_pk_fp8_f32(tof(r[4 * k + 2]) / sc, tof(r[4 * k + 3]) / sc, 0, false);
      w[k] = (unsigned)(lo & 0xFFFF) | ((unsigned)(hi & 0xFFFF) << 16);
    }
    *reinterpret_cast<u32x2q*>(dst8 + c * 8) = w;
    if (c == 0) *sc_out = sc;
  };
  if (h >= nq + nkv) {                              // v: plain copy into the cache
    const int kvh = h - nq - nkv;
    const v8 vv = ld8<T>(src + c * 8);
    st8<T>(vc + bi * c_sb + kvh * c_sh + (int64_t)pp * 128 + c * 8, vv);
    if (vq8) quant_row(vv, vq8 + bi * c_sb + kvh * c_sh + (int64_t)pp * 128, vs + bi * s_sb + kvh * s_sh + pp);
    return;
  }
  const int pt = pr < max_pos ? pr : max_pos - 1;
  const v8 x = ld8<T>(src + c * 8);
  const v8 o = ld8<T>(src + ((c + 8) & 15) * 8);    // partner chunk (d +- 64)
  const float sgn = c < 8 ? -1.f : 1.f;             // rotate_half: first half gets -x2, second half +x1
  const float* cs = cos_sin + ((size_t)pt * 64 + (c & 7) * 8) * 2;
  v8 r;
  {
    // no contraction: each product is rounded to the 16-bit type before the add, as the reference does (attn_common.h: rope_chunk)
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float co = rnd<T>(cs[2 * j]), si = rnd<T>(cs[2 * j + 1]);
      const float a = rnd<T>(tof(x[j]) * co);
      const float b = rnd<T>(sgn * tof(o[j]) * si);
      r[j] = fromf<T>(a + b);
    }
  }
  if (h < nq) {
    // every lane of the 16-lane group has read both chunks before anyone writes (same wave, program order)
    st8<T>(src + c * 8, r);
  } else {
    const int kvh = h - nq;
    st8<T>(kc + bi * c_sb + kvh * c_sh + (int64_t)pp * 128 + c * 8, r);
    if (kq8) quant_row(r, kq8 + bi * c_sb + kvh * c_sh + (int64_t)pp * 128, ks + bi * s_sb + kvh * s_sh + pp);
  }
}

// ---------------------------------------------------------------------------------------------------------
// patch embed helpers (InternVisionEmbeddings.forward, modeling_intern_vit.py:90-102)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_kernel(const T* px, T* cols, int B, int HW, int patch, int Kpad) {
  const int g = HW / patch;
  const long n = (long)B * g * g * Kpad;
  const int K = 3 * patch * patch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad);
    const long m = i / Kpad;
    T v = (T)0.f;
    if (k < K) {
      const int b = (int)(m / (g * g)), pp = (int)(m % (g * g)), py = pp / g, pxx = pp % g;
      const int ch = k / (patch * patch), kk = k % (patch * patch), ky = kk / patch, kx = kk % patch;
      v = px[(((size_t)b * 3 + ch) * HW + py * patch + ky) * HW + pxx * patch + kx];
    }
    cols[i] = v;
  }
}

template <typename T>
__global__ void vit_assemble_kernel(const T* pe, const T* cls, const T* pos, T* x, int B, int np, int C) {
  typedef typename V8<T>::type v8;
  const int cc = C >> 3;
  const long n = (long)B * (np + 1) * cc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc);
    const long rt = i / cc;
    const int t = (int)(rt % (np + 1)), b = (int)(rt / (np + 1));
    const v8 a = t == 0 ? ld8<T>(cls + c * 8) : ld8<T>(pe + ((size_t)b * np + (t - 1)) * C + c * 8);
    const v8 p = ld8<T>(pos + (size_t)t * C + c * 8);
    v8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(a[j]) + tof(p[j]));
    st8<T>(x + ((size_t)b * (np + 1) + t) * C + c * 8, o);
  }
}

// splice gather (omchat_arch.py:133-158,172-195): pure row copies
template <typename T>
__global__ void gather_rows_kernel(const int* idx, const T* table, const T* feats, T* out, int rows, int H) {
  typedef typename V8<T>::type v8;
  const int cc = H >> 3;
  const long n = (long)rows * cc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc);
    const int r = (int)(i / cc);
    const int id = idx[r];
    v8 v;
    if (id == INT_MIN) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    } else if (id >= 0) {
      v = ld8<T>(table + (size_t)id * H + c * 8);
    } else {
      v = ld8<T>(feats + (size_t)(-1 - id) * H + c * 8);
    }
    st8<T>(out + (size_t)r * H + c * 8, v);
  }
}

template <typename T>
__global__ void copy_rows_kernel(const T* src, int64_t src_ld, T* dst, int64_t dst_ld, int rows, int H, int group, int skip) {
  typedef typename V8<T>::type v8;
  const int cc = H >> 3;
  const long n = (long)rows * cc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cc);
    const long r = i / cc;
    const long sr = (r / group) * (group + skip) + skip + r % group;
    st8<T>(dst + r * dst_ld + c * 8, ld8<T>(src + sr * src_ld + c * 8));
  }
}

// greedy argmax over fp32 logits, first index wins ties (torch.argmax semantics).  Two stages:
// ARG_CHUNKS workgroups per row reduce a slice each, then one wave picks among the slice winners.
constexpr int ARG_CHUNKS = 64;
__device__ __forceinline__ void arg_better(float& best, int& besti, float v, int i) {
  if (v > best || (v == best && i < besti)) { best = v; besti = i; }
}
__global__ __launch_bounds__(256) void argmax_stage1_kernel(const float* logits, int ld, int V, float* pv, int* pi) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const float* row = logits + (size_t)blockIdx.y * ld;
  const int per = (V + ARG_CHUNKS - 1) / ARG_CHUNKS;
  const int lo = blockIdx.x * per, hi = lo + per < V ? lo + per : V;
  float best = -INFINITY;
  int besti = INT_MAX;
  for (int i = lo + threadIdx.x; i < hi; i += 256) arg_better(best, besti, row[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) arg_better(best, besti, __shfl_xor(best, o, 64), __shfl_xor(besti, o, 64));
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) arg_better(best, besti, bv[w], bi[w]);
    pv[blockIdx.y * ARG_CHUNKS + blockIdx.x] = best;
    pi[blockIdx.y * ARG_CHUNKS + blockIdx.x] = besti;
  }
}
__global__ __launch_bounds__(64) void argmax_stage2_kernel(const float* pv, const int* pi, int* out, int* adv_pos, int* adv_len) {
  if (adv_pos && threadIdx.x == 0) { adv_pos[blockIdx.x] += 1; adv_len[blockIdx.x] += 1; }
  float best = pv[blockIdx.x * ARG_CHUNKS + threadIdx.x];
  int besti = pi[blockIdx.x * ARG_CHUNKS + threadIdx.x];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) arg_better(best, besti, __shfl_xor(best, o, 64), __shfl_xor(besti, o, 64));
  if (threadIdx.x == 0) out[blockIdx.x] = besti == INT_MAX ? 0 : besti;
}

// omchat_amd/synth.py::uniform, bit for bit
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
template <typename T>
__global__ void fill_uniform_kernel(T* dst, int64_t n, uint64_t key, float mul, float offset) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t h = splitmix64(key + (uint64_t)i * 0x9E3779B97F4A7C15ull);
    const int iv = (int)(h >> 40) - (1 << 23);
    float v = __fmul_rn((float)iv, mul);
    if (offset != 0.f) v = __fadd_rn(v, offset);
    // round to bf16 (RNE), keep only values exact in fp16
    uint32_t u = __float_as_uint(v);
    u = ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16) << 16;
    float f = __uint_as_float(u);
    if ((float)((f16)f) != f) f = 0.f;
    dst[i] = (T)f;
  }
}

template <typename T>
__global__ void cast_f32_kernel(const T* src, float* dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = tof(src[i]);
}

inline int grid_for(long n, int threads) {
  long g = (n + threads - 1) / threads;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

#define DISPATCH(dtype, CALL)                                        \
  if ((dtype) == OMCHAT_F16) { typedef f16 T; CALL; }                \
  else if ((dtype) == OMCHAT_BF16) { typedef bf16 T; CALL; }         \
  else { omchat_set_error("bad dtype"); return 1; }

// omchat_op_set_tuning key 40 (experiments build): 1 = RMSNorm / ViT q-k norm launches of >= NORM_WAVE_ROWS rows with H <= 4096 take the one-wave-per-row
// kernels (same bits as the workgroup-per-row ones; measured no faster), 0 (default) = always a workgroup per row
static int g_norm_wave = 0;
[[maybe_unused]] constexpr int NORM_WAVE_ROWS = 256;
void norm_set_wave(int v) { g_norm_wave = v; }

int launch_layernorm(int dtype, const void* x, int ldx, const void* w, const void* b, void* y, int ldy, int rows, int H, float eps, hipStream_t s) {
  OM_CHECK(H % 8 == 0 && H <= NORM_THREADS * NORM_MAXC * 8 && ldx % 8 == 0 && ldy % 8 == 0, "H must be a multiple of 8 and <= 16384");
  OM_CHECK(w && b, "LayerNorm needs weight and bias");
  DISPATCH(dtype, hipLaunchKernelGGL(layernorm_kernel<T>, dim3(rows), dim3(NORM_THREADS), 0, s, (const T*)x, ldx, (const T*)w, (const T*)b, (T*)y, ldy, H, eps));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_rmsnorm(int dtype, const void* x, int ldx, const void* w, void* y, int ldy, int rows, int H, float eps, hipStream_t s, int pack_nb) {
  OM_CHECK(H % 8 == 0 && H <= NORM_THREADS * NORM_MAXC * 8 && ldx % 8 == 0 && ldy % 8 == 0, "H % 8, H <= 16384, ld % 8");
  OM_CHECK(pack_nb == 0 || (rows <= 16 * pack_nb && H % 64 == 0 && x != y), "packed output: rows <= 16 * NB, H % 64 == 0, not in place");
  if (rows == 0) return 0;
#if OMCHAT_EXPERIMENTS
  if (g_norm_wave && rows >= NORM_WAVE_ROWS && H <= NORM_WAVE_H && pack_nb == 0) {
    DISPATCH(dtype, hipLaunchKernelGGL(rmsnorm_wave_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)x, ldx, (const T*)w, (T*)y, ldy, rows, H, eps));
    OM_LAUNCH_CHECK();
    return 0;
  }
#endif
  DISPATCH(dtype, hipLaunchKernelGGL(rmsnorm_kernel<T>, dim3(rows), dim3(NORM_THREADS), 0, s, (const T*)x, ldx, (const T*)w, (T*)y, ldy, H, eps, pack_nb));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_rmsnorm_q8(int dtype, const void* x, int ldx, const void* w, void* y8, int ldy, float* scale, int rows, int H, float eps, hipStream_t s) {
  OM_CHECK(H % 8 == 0 && H <= NORM_THREADS * NORM_MAXC * 8 && ldx % 8 == 0 && ldy % 8 == 0, "H % 8, H <= 16384, ld % 8");
  if (rows == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL((rows_q8_kernel<T, true>), dim3(rows), dim3(NORM_THREADS), 0, s, (const T*)x, ldx, (const T*)w, (unsigned char*)y8, ldy, scale, H, eps));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_quant_rows_q8(int dtype, const void* x, int ldx, void* y8, int ldy, float* scale, int rows, int H, hipStream_t s) {
  OM_CHECK(H % 8 == 0 && H <= NORM_THREADS * NORM_MAXC * 8 && ldx % 8 == 0 && ldy % 8 == 0, "H % 8, H <= 16384, ld % 8");
  if (rows == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL((rows_q8_kernel<T, false>), dim3(rows), dim3(NORM_THREADS), 0, s, (const T*)x, ldx, (const T*)nullptr, (unsigned char*)y8, ldy, scale, H, 0.f));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_vit_qknorm(int dtype, void* qkv, int ld, const void* wq, const void* wk, int rows, int C, int C_total, float eps,
                      float q_scale, const float* sumsq_in, hipStream_t s, int only_k) {
  OM_CHECK(C % 8 == 0 && C <= NORM_THREADS * NORM_MAXC * 8 && ld % 8 == 0, "C % 8, C <= 16384, ld % 8");
  OM_CHECK(!only_k || sumsq_in, "the K half alone takes its statistics from sumsq_in");
  if (rows == 0) return 0;
#if OMCHAT_EXPERIMENTS
  if (!only_k && g_norm_wave && rows >= NORM_WAVE_ROWS && C <= NORM_WAVE_H) {
    DISPATCH(dtype, hipLaunchKernelGGL(vit_qknorm_wave_kernel<T>, dim3((rows + 3) / 4, 2), dim3(256), 0, s, (T*)qkv, ld, (const T*)wq, (const T*)wk, rows,
                                       C, C_total, eps, q_scale, sumsq_in));
    OM_LAUNCH_CHECK();
    return 0;
  }
#endif
  DISPATCH(dtype, hipLaunchKernelGGL(vit_qknorm_kernel<T>, dim3(rows, only_k ? 1 : 2), dim3(NORM_THREADS), 0, s, (T*)qkv, ld, (const T*)wq, (const T*)wk,
                                     C, C_total, eps, q_scale, sumsq_in, only_k ? 1 : 0));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_fold_cols(int dtype, const void* W, const void* n, void* out, int rows, int cols, hipStream_t s) {
  OM_CHECK(cols % 8 == 0 && W && n && out, "cols % 8");
  const long chunks = (long)rows * (cols / 8);
  if (chunks == 0) return 0;
  const int grid = (int)std::min<long>((chunks + 255) / 256, 8192);
  DISPATCH(dtype, hipLaunchKernelGGL(fold_cols_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)W, (const T*)n, (T*)out, chunks, cols / 8));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_stats_finish(const float* stats, int ld, int slot0, int nslots, int ngroups, int rows, int dim, float eps, float* out, hipStream_t s) {
  OM_CHECK(stats && out && nslots >= 1 && ngroups >= 1 && slot0 >= 0 && ld >= rows && dim >= 0, "bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(stats_finish_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, stats, ld, slot0, nslots, ngroups, rows, dim > 0 ? 1.0f / (float)dim : 0.f, eps, out);
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_row_sumsq(int dtype, const void* x, int ldx, int rows, int H, float* stats, hipStream_t s) {
  OM_CHECK(H % 8 == 0 && ldx % 8 == 0 && stats, "H % 8, ldx % 8");
  if (rows == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(row_sumsq_kernel<T>, dim3(rows), dim3(NORM_THREADS), 0, s, (const T*)x, ldx, H, stats));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_vit_knorm_slots(int dtype, void* k, int ld, const void* wk, int rows, int C, int C_total, float eps, const float* stats, int stats_ld, int nslots,
                           float* sumsq_q, hipStream_t s) {
  OM_CHECK(C % 8 == 0 && ld % 8 == 0 && C <= NORM_THREADS * NORM_MAXC * 8, "C % 8, ld % 8, C <= 16384");
  OM_CHECK(stats && sumsq_q && nslots >= 1 && nslots <= 64 && stats_ld >= rows, "statistics: slot-major [2 * nslots][stats_ld >= rows], 1..64 q slots followed by as many k slots");
  if (rows == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(vit_knorm_slots_kernel<T>, dim3(rows), dim3(NORM_THREADS), 0, s, (T*)k, ld, (const T*)wk, C, C_total, eps, stats, stats_ld,
                                     nslots, sumsq_q));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_resid16_norm(int dtype, void* x, int ldx, const void* y, int ldy, const void* w, const void* b, void* xn, int ldn, int rows, int H, float eps,
                        hipStream_t s) {
  OM_CHECK(H % 8 == 0 && H <= NORM_THREADS * NORM_MAXC * 8 && ldx % 8 == 0 && ldy % 8 == 0 && (!w || (xn && ldn % 8 == 0)), "H % 8, H <= 16384, ld % 8");
  if (rows <= 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(resid16_norm_kernel<T>, dim3(rows), dim3(NORM_THREADS), 0, s, (T*)x, ldx, (const T*)y, ldy, (const T*)w, (const T*)b,
                                     (T*)xn, ldn, H, eps));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_vit_qk_sumsq(int dtype, const void* qkv, int ld, int rows, int C, float* out, hipStream_t s) {
  OM_CHECK(C % 8 == 0 && ld % 8 == 0, "C % 8, ld % 8");
  if (rows == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(vit_qk_sumsq_kernel<T>, dim3(rows), dim3(NORM_THREADS), 0, s, (const T*)qkv, ld, C, out));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_rope_kv(int dtype, const RopeArgs& a, hipStream_t s) {
  OM_CHECK(a.ld % 8 == 0 && a.cos_sin && a.kcache && a.vcache, "bad args");
  OM_CHECK(!a.k8 || (a.v8 && a.ks && a.vs), "fp8 KV append: k8, v8 and both scale arrays");
  if (a.rows == 0) return 0;
  const long items = (long)a.rows * (a.nq + 2 * a.nkv) * 16;
  DISPATCH(dtype, hipLaunchKernelGGL(rope_kv_kernel<T>, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, s, (T*)a.qkv, a.ld, a.rows, a.S,
                                     a.nq, a.nkv, a.pos, a.pos0, a.cos_sin, a.max_pos, (T*)a.kcache, (T*)a.vcache, a.c_sb, a.c_sh,
                                     (unsigned char*)a.k8, (unsigned char*)a.v8, a.ks, a.vs, a.s_sb, a.s_sh, a.slot0));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_im2col(int dtype, const void* pixels, void* cols, int B, int HW, int patch, int Kpad, hipStream_t s) {
  OM_CHECK(HW % patch == 0 && Kpad >= 3 * patch * patch, "bad geometry");
  const int g = HW / patch;
  const long n = (long)B * g * g * Kpad;
  if (n == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(im2col_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const T*)pixels, (T*)cols, B, HW, patch, Kpad));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_vit_assemble(int dtype, const void* pe, const void* cls, const void* pos, void* x, int B, int np, int C, hipStream_t s) {
  OM_CHECK(C % 8 == 0, "C % 8");
  const long n = (long)B * (np + 1) * (C / 8);
  if (n == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(vit_assemble_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const T*)pe, (const T*)cls, (const T*)pos, (T*)x, B, np, C));
  OM_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Tensor parallelism, fp32 partial sums (tuning key 29): the row-parallel projections write their raw fp32 accumulators (EPI_F32OUT), the
// all-reduce sums them in fp32, and this kernel applies the epilogue ONCE to the sum, with the rounding points of the one-GPU epilogue:
//   EPI_NONE: T(sum + b)    EPI_RESID: T(r + T(sum + b))    EPI_LS_RESID: T(r + T(T(sum + b) * ls))      (gemm.hip: gemm_epilogue)
// so that a TP = N result differs from TP = 1 only in the fp32 summation order of the K range.  (Default path: every rank rounds its own
// partial -- and, in the ViT, applies the layer scale to it -- before the sum: N roundings instead of one.)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void tp_finish_kernel(const float* __restrict__ sum, const T* __restrict__ bias, const T* __restrict__ ls,
                                                        const T* resid, T* out, long total4, int N, int epi) {
  // no contraction: resid + T(T(sum + b) * ls) must keep the rounding of the product (see gemm.hip gemm_epilogue, round 6)
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    const int col = (int)(e % N);
    const f32x4 a = *reinterpret_cast<const f32x4*>(sum + e);
    typename V8<T>::half_type o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = a[j] + (bias ? tof(bias[col + j]) : 0.f);
      if (epi != EPI_NONE) v = rnd<T>(v);
      if (epi == EPI_LS_RESID) v = tof(resid[e + j]) + rnd<T>(v * tof(ls[col + j]));
      if (epi == EPI_RESID) v = tof(resid[e + j]) + v;
      o[j] = fromf<T>(v);
    }
    *reinterpret_cast<typename V8<T>::half_type*>(out + e) = o;
  }
}

int launch_tp_finish(int dtype, const float* sum, const void* bias, const void* ls, const void* resid, void* out, int M, int N, int epi, hipStream_t s) {
  OM_CHECK(sum && out && M > 0 && N > 0 && N % 4 == 0, "tp_finish: null argument or N % 4 != 0");
  OM_CHECK(epi == EPI_NONE || ((epi == EPI_RESID || epi == EPI_LS_RESID) && resid && (epi != EPI_LS_RESID || ls)), "tp_finish: epilogue NONE / RESID / LS_RESID with its operands");
  const long total4 = (long)M * N / 4;
  const int grid = (int)std::min<long>((total4 + 255) / 256, 4096);
  if (dtype == OMCHAT_F16) hipLaunchKernelGGL(tp_finish_kernel<f16>, dim3(grid), dim3(256), 0, s, sum, (const f16*)bias, (const f16*)ls, (const f16*)resid, (f16*)out, total4, N, epi);
  else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL(tp_finish_kernel<bf16>, dim3(grid), dim3(256), 0, s, sum, (const bf16*)bias, (const bf16*)ls, (const bf16*)resid, (bf16*)out, total4, N, epi);
  else { omchat_set_error("tp_finish: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_gather_rows(int dtype, const int* idx, const void* table, const void* feats, void* out, int rows, int H, hipStream_t s) {
  OM_CHECK(H % 8 == 0, "H % 8");
  const long n = (long)rows * (H / 8);
  if (n == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, s, idx, (const T*)table, (const T*)feats, (T*)out, rows, H));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_copy_rows(int dtype, const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int rows, int H, int group, int skip, hipStream_t s) {
  OM_CHECK(H % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0 && group > 0, "H/ld % 8, group > 0");
  const long n = (long)rows * (H / 8);
  if (n == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const T*)src, src_ld, (T*)dst, dst_ld, rows, H, group, skip));
  OM_LAUNCH_CHECK();
  return 0;
}

size_t argmax_scratch_bytes(int b) { return (size_t)b * ARG_CHUNKS * 8; }

int launch_argmax(const float* logits, int ld, int b, int V, int* out, void* scratch, hipStream_t s, int* adv_pos, int* adv_len) {
  if (b == 0) return 0;
  OM_CHECK(scratch, "argmax scratch missing");
  float* pv = (float*)scratch;
  int* pi = (int*)((char*)scratch + (size_t)b * ARG_CHUNKS * 4);
  hipLaunchKernelGGL(argmax_stage1_kernel, dim3(ARG_CHUNKS, b), dim3(256), 0, s, logits, ld, V, pv, pi);
  hipLaunchKernelGGL(argmax_stage2_kernel, dim3(b), dim3(64), 0, s, pv, pi, out, adv_pos, adv_len ? adv_len : adv_pos);
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_resid_rmsnorm(int dtype, void* x, int ldx, const float* part, int ks, const void* w, void* xn, int ldn, int rows, int H, float eps,
                         hipStream_t s, int pack_nb) {
  OM_CHECK(H % 8 == 0 && H <= NORM_THREADS * NORM_MAXC * 8 && ldx % 8 == 0 && ldn % 8 == 0 && ks >= 1, "H % 8, H <= 16384, ld % 8");
  OM_CHECK(pack_nb == 0 || (rows <= 16 * pack_nb && H % 64 == 0), "packed output: rows <= 16 * NB, H % 64 == 0");
  if (rows == 0) return 0;
  OM_CHECK(ks <= 8, "at most 8 K slices");
#define RR(KS_) case KS_: DISPATCH(dtype, hipLaunchKernelGGL((resid_rmsnorm_kernel<T, KS_>), dim3(rows), dim3(NORM_THREADS), 0, s, (T*)x, ldx, part, rows, (const T*)w, (T*)xn, ldn, H, eps, pack_nb)); break;
  switch (ks) { RR(1) RR(2) RR(3) RR(4) RR(5) RR(6) RR(7) RR(8) }
#undef RR
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_fill_uniform(int dtype, void* dst, int64_t n, uint64_t key, float scale, float offset, hipStream_t s) {
  if (n == 0) return 0;
  const float mul = scale / 8388608.0f;
  DISPATCH(dtype, hipLaunchKernelGGL(fill_uniform_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, s, (T*)dst, n, key, mul, offset));
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_cast_f32(int dtype, const void* src, float* dst, int64_t n, hipStream_t s) {
  if (n == 0) return 0;
  DISPATCH(dtype, hipLaunchKernelGGL(cast_f32_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const T*)src, dst, n));
  OM_LAUNCH_CHECK();
  return 0;
}
