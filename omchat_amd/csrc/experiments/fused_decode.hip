// Batch-1 decode, one GPU: split-KV attention + merge + o_proj (+ residual) as ONE launch (round 4).
//
// Stands behind the same reference code as the three launches it replaces -- Qwen2Attention.forward with a KV cache and the residual add
// of Qwen2DecoderLayer.forward (transformers modeling_qwen2.py:150-172,195-234,283-288) -- and gives the SAME BITS as they do: the
// attention tile is the shared body attn_decode_tile (attn_common.h), the merge and the row dot products repeat the arithmetic and the
// summation order of attn_merge_kernel (<= 64 partials) and gemv_rows_kernel<EPI_RESID, 1 row per wave>.
//
// Why: at batch 1 these launches move 7 MB (K / V at 3.6 k keys) + 26 MB (o_proj weights) and take 6.4 + 4.6 + 6.6 us
// (profiles/r03_ai_kernel_stats_configs1...), three dependent kernel boundaries with one memory round trip behind each.  Here one
// workgroup per CU (8 waves) runs the chain with in-launch hand-offs:
//   wave 7      attention tile of unit (kv head, 64-key split) = blockIdx -> publishes (O, m, l) as granules
//               merge of unit (head, 16-column group) = blockIdx: sweeps the <= 64 partials of its columns, publishes 16 merged values
//   waves 0..6  issue the loads of this CU's o_proj rows (2 rows each, 100 KB per CU) at once -- BEHIND wave 7's K / V loads (a raw
//               workgroup barrier orders the issue: the CU's memory pipe returns in order) -- and hold them in registers
//   all waves   sweep the merged attention row (3584 values from 224 CUs) into LDS, then waves 0..6 finish their rows: dot products,
//               x + T(row) in place.
// Hand-off = 8-byte {tag, value} granules written with ONE sc1 (write-through) store each and swept with sc1 loads until every tag equals
// this launch's tag (cdna_hip_programming.md section 6 Guideline 16, form R2: the data is the flag; no fence, no counter; placement
// independent).  Measured on MI355X (tools/experiments/tune_handoff.hip, profiles/r04_a_handoff.txt): 2.1-2.3 us per all-to-all edge of up to 16 KB
// of granules on a quiet chip.  Every spin is bounded by the wall clock (s_memrealtime) and reports through a sticky error word
// (omchat_fused_status); the launch needs all its workgroups resident at once: grid = number of CUs, one workgroup per CU (84 KB of LDS
// requested), nothing else of this kind running beside it.
#include "kernels.h"
#if OMCHAT_EXPERIMENTS
#include "attn_common.h"
#include "rowdot.h"

namespace {

typedef unsigned long long u64;
constexpr int PG_STRIDE = 132;            // granules per (head, split): 128 O columns, m, l, 2 pad
constexpr int FD_STAGE = 16384;           // LDS: [0, 16 K) V transpose image; 16 x 64 fp32 merge staging; 64 weights; the merged row
constexpr int FD_FW = FD_STAGE + 16 * 64 * 4;
constexpr int FD_AO = FD_FW + 256;
constexpr size_t FD_LDS_REQUEST = 84 * 1024;      // more than half of the CU's 160 KB: one workgroup per CU

struct FusedP {
  const void* Wo; int ldw;      // o_proj weight [H][qd] row-major
  void* x;                      // residual stream [H]: read, x + attn written in place
  int H, qd, rows_per_wg;
  u64* part_g;                  // [q_heads][64][PG_STRIDE]
  u64* ao_g;                    // [q_heads * 64]: two merged 16-bit values per granule
  unsigned tag_a, tag_b;
  unsigned* err;
  u64 timeout_ticks;            // s_memrealtime ticks (100 MHz)
  u64* dbg;                     // diagnostic build only (tools/experiments/tune_fused.hip, OMCHAT_FUSED_STAMPS): [gridDim.x][16] phase stamps
};

// phase stamps of the diagnostic build: lane 0 of wave 7 (slots 0..7) and of wave 0 (slots 8..15) store s_memrealtime (100 MHz) into a
// buffer nothing else reads; the product build compiles them away
#ifdef OMCHAT_FUSED_STAMPS
#define FD_STAMP(slot) do { if (f.dbg && lane == 0) f.dbg[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FD_STAMP(slot) do { } while (0)
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
// a spin that has lasted longer than the budget: record it and let the caller go on with whatever it has (wrong data, no hang)
__device__ __forceinline__ bool spin_expired(u64 t0, const FusedP& f, unsigned code, int lane) {
  if (__builtin_amdgcn_s_memrealtime() - t0 <= f.timeout_ticks) return false;
  if (lane == 0) atomicOr(f.err, code);
  return true;
}

template <typename T, int NCH, int RW>
__global__ __launch_bounds__(512) void attn_oproj_fused_kernel(AttnP p, FusedP f) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x;
  const int kv_len = p.kv_len ? p.kv_len[0] : p.Skv;
  const int ns = (kv_len + KV_TILE - 1) / KV_TILE;            // partials per head (<= 64: checked by the launcher)
  const int n_rep = p.q_heads / p.kv_heads;
  unsigned* ao_lds = (unsigned*)(smem + FD_AO);
  const int n_pairs = f.qd >> 1;                              // granules of the merged row

  typedef rw_u32x4 wreg_t;
  wreg_t w[RW][NCH];
  float resid[RW];
  int rown[RW];
  bool rvalid[RW];

  if (wave == 7) {
    FD_STAMP(0);
    // ---- A. attention tile of unit (kv head, split) = c
    const int kvh = c % p.kv_heads, split = c / p.kv_heads;
    if (split < ns) {
      f32x4 o[8];
      float mx, l;
      attn_decode_tile<T, false, true>(p, split, kvh, 0, kv_len, smem, lane, o, mx, l);
      FD_STAMP(1);
      const int fc = lane & 15, fg = lane >> 4;
      if (fc < n_rep) {
        u64* base = f.part_g + ((size_t)(kvh * n_rep + fc) * 64 + split) * PG_STRIDE;
#pragma unroll
        for (int dn = 0; dn < 8; ++dn)
#pragma unroll
          for (int r = 0; r < 4; ++r) g_store(base + dn * 16 + fg * 4 + r, f.tag_a, __float_as_uint(o[dn][r]));
        if (fg == 0) { g_store(base + 128, f.tag_a, __float_as_uint(mx)); g_store(base + 129, f.tag_a, __float_as_uint(l)); }
      }
    } else {
      __builtin_amdgcn_s_barrier();               // the barrier the tile executes after issuing its loads
    }
    FD_STAMP(2);
    // ---- B. merge of unit (head, 16-column group) = c: attn_merge_kernel's arithmetic for ns <= 64, lane = split
    if (c < p.q_heads * 8) {
      const int h = c >> 3, dg = c & 7;
      const int s = lane, sc = s < ns ? s : ns - 1;
      const u64* base = f.part_g + ((size_t)h * 64 + sc) * PG_STRIDE;
      unsigned xv[16], xm = 0, xl = 0;
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const u64 g = g_load(base + dg * 16 + j); xv[j] = (unsigned)g; ok &= (unsigned)(g >> 32) == f.tag_a; }
        { const u64 g = g_load(base + 128); xm = (unsigned)g; ok &= (unsigned)(g >> 32) == f.tag_a; }
        { const u64 g = g_load(base + 129); xl = (unsigned)g; ok &= (unsigned)(g >> 32) == f.tag_a; }
        if (__all(ok || s >= ns)) break;
        if (spin_expired(t0, f, 1u, lane)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      FD_STAMP(3);
      const float m_s = s < ns ? __uint_as_float(xm) : NEG_BIG;
      const float l_s = s < ns ? __uint_as_float(xl) : 0.f;
      const float m = wave_max(m_s);
      const float fw = s < ns ? exp2f((m_s - m) * p.c) : 0.f;
      const float lsum = wave_sum(fw * l_s);
      const float ltot = __shfl(lsum, 0, 64);       // the stand-alone kernel divides by lane 0's sum
      float* stage = (float*)(smem + FD_STAGE);
      float* fws = (float*)(smem + FD_FW);
      fws[s] = fw;
#pragma unroll
      for (int j = 0; j < 16; ++j) stage[j * 64 + s] = s < ns ? __uint_as_float(xv[j]) : 0.f;
      lds_fence_wave();
      float a = 0.f;
      const int d = lane & 15;
#pragma unroll
      for (int k = 0; k < 64; ++k) a += fws[k] * stage[d * 64 + k];
      const T ov = fromf<T>(a / ltot);
      const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, ov);
      const unsigned hi = (unsigned)__shfl_down((int)bits, 1, 64);
      if (lane < 16 && (lane & 1) == 0) g_store(f.ao_g + ((h * 128 + dg * 16 + lane) >> 1), f.tag_b, bits | (hi << 16));
    }
    FD_STAMP(4);
    __builtin_amdgcn_s_barrier();                 // (b) the other waves start polling only now: until here the CU's memory pipe was this wave's
  } else {
    if (wave == 0) FD_STAMP(8);
    // zero tail of the merged row in LDS (K is padded to whole 512-element chunks; the weights are clamped there)
    for (int i = n_pairs + (int)threadIdx.x; i < NCH * 256; i += 448) ao_lds[i] = 0u;
    __builtin_amdgcn_s_barrier();                 // wave 7 has issued its K / V / q loads: ours queue behind them
    // ---- this wave's o_proj rows: every load now
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int local = wave * RW + r;
      int n = c * f.rows_per_wg + local;
      rvalid[r] = local < f.rows_per_wg && n < f.H;
      n = n < f.H ? n : f.H - 1;
      rown[r] = n;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        int k = ch * 512 + lane * 8;
        k = k < f.qd ? k : 0;
        w[r][ch] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)f.Wo + (size_t)n * f.ldw + k));
      }
      resid[r] = tof(((const T*)f.x)[n]);
    }
    if (wave == 0) FD_STAMP(9);
    __builtin_amdgcn_s_barrier();                 // (b)
    if (wave == 0) FD_STAMP(10);
  }
  // ---- C. the merged attention row: every wave sweeps an eighth of the granules into LDS
  {
    const int per = (n_pairs + 7) >> 3;
    const int lo = wave * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
    if (lo < hi) {
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      for (int g0 = lo; g0 < hi; g0 += 64 * 4) {
        for (;;) {
          bool ok = true;
          unsigned v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = g0 + k * 64 + lane;
            const u64 g = g_load(f.ao_g + (i < hi ? i : hi - 1));
            v[k] = (unsigned)g; ok &= (unsigned)(g >> 32) == f.tag_b;
          }
          if (__all(ok)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = g0 + k * 64 + lane; if (i < hi) ao_lds[i] = v[k]; }
            break;
          }
          if (spin_expired(t0, f, 2u, lane)) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
  }
  if (wave == 7) FD_STAMP(5);
  if (wave == 0) FD_STAMP(11);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wave == 7) FD_STAMP(6);
  // ---- D. o_proj rows: gemv_rows_kernel<EPI_RESID, one row per wave>'s arithmetic
  if (wave < 7) {
    rw_u32x4 xr[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) xr[ch] = *reinterpret_cast<const rw_u32x4*>(smem + FD_AO + (ch * 512 + lane * 8) * 2);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      float a = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) a = rw_dot8<T>(w[r][ch], xr[ch], a);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0 && rvalid[r]) {
        float y = a + 0.f;
        y = rnd<T>(y);
        y = resid[r] + y;
        ((T*)f.x)[rown[r]] = fromf<T>(y);
      }
    }
    if (wave == 0) FD_STAMP(12);
  }
}

template <typename T, int NCH>
int launch_nch(const AttnP& p, const FusedP& f, int grid, int rw, hipStream_t s) {
  const size_t lds = FD_LDS_REQUEST;
  if (rw == 1) {
    auto k = attn_oproj_fused_kernel<T, NCH, 1>;
    static bool set = false;
    if (!set) { OM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, p, f);
  } else {
    auto k = attn_oproj_fused_kernel<T, NCH, 2>;
    static bool set = false;
    if (!set) { OM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, p, f);
  }
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int launch_t(const AttnP& p, const FusedP& f, int grid, int rw, hipStream_t s) {
  switch (cdiv(f.qd, 512)) {
    case 1: return launch_nch<T, 1>(p, f, grid, rw, s);
    case 2: return launch_nch<T, 2>(p, f, grid, rw, s);
    case 3: return launch_nch<T, 3>(p, f, grid, rw, s);
    case 4: return launch_nch<T, 4>(p, f, grid, rw, s);
    case 5: return launch_nch<T, 5>(p, f, grid, rw, s);
    case 6: return launch_nch<T, 6>(p, f, grid, rw, s);
    case 7: return launch_nch<T, 7>(p, f, grid, rw, s);
    default: return launch_nch<T, 8>(p, f, grid, rw, s);
  }
}

}  // namespace

size_t fused_decode_ws_bytes(int q_heads) { return ((size_t)q_heads * 64 * PG_STRIDE + (size_t)q_heads * 64) * sizeof(u64) + 64; }

bool attn_oproj_fused_ok(const AttnDecodeArgs& a, int H, int qd) {
  const int G = device_cus();
  const int ns = cdiv(a.L, KV_TILE);
  return a.batch == 1 && !a.k_scale && !a.key_mask && a.o_pack_nb == 0 && ns <= 64 && (long)a.kv_heads * ns <= G && a.q_heads * 8 <= G &&
         a.q_heads % a.kv_heads == 0 && a.q_heads / a.kv_heads <= 16 && qd == a.q_heads * 128 && qd <= 4096 && cdiv(H, G) <= 14 && H >= 1;
}

int launch_attn_oproj_fused(int dtype, const AttnDecodeArgs& a, const FusedDecodeArgs& fa, hipStream_t s) {
  OM_CHECK(attn_oproj_fused_ok(a, fa.H, fa.qd), "geometry outside the fused attention + o_proj launch (batch 1, <= 4096 keys, 16-bit cache)");
  OM_CHECK(a.rope && a.k_new && a.v_new, "the fused launch rotates q / k and appends k / v itself: rope, k_new and v_new are required");
  OM_CHECK(fa.ws && fa.err && fa.Wo && fa.x && fa.ldw % 8 == 0, "fused decode: null argument / unaligned weight rows");
  const int G = device_cus();
  AttnP p{a.Q, a.K, a.V, nullptr, a.q_sb, a.q_sh, 0, a.k_sb, a.k_sh, a.k_sr, a.v_sb, a.v_sh, a.v_sr, 0, 0, 0,
          a.kv_len, nullptr, a.q_heads, a.kv_heads, 1, a.L, 0, 0, cdiv(a.L, KV_TILE), a.scale * 1.4426950408889634f, nullptr,
          a.rope, a.pos, a.k_new, a.v_new, a.new_sb, (void*)a.K, (void*)a.V, a.rope_max, nullptr, nullptr, 0, 0, 1, nullptr, 0};
  FusedP f;
  f.Wo = fa.Wo; f.ldw = fa.ldw; f.x = fa.x; f.H = fa.H; f.qd = fa.qd; f.rows_per_wg = cdiv(fa.H, G);
  f.part_g = (u64*)fa.ws; f.ao_g = f.part_g + (size_t)a.q_heads * 64 * PG_STRIDE;
  f.tag_a = fa.epoch * 2u; f.tag_b = fa.epoch * 2u + 1u;
  f.err = fa.err;
  f.timeout_ticks = (u64)fa.timeout_ms * 100000ull;
  f.dbg = (u64*)fa.dbg;
  const int rw = cdiv(f.rows_per_wg, 7);
  if (dtype == OMCHAT_F16) return launch_t<f16>(p, f, G, rw, s);
  if (dtype == OMCHAT_BF16) return launch_t<bf16>(p, f, G, rw, s);
  omchat_set_error("launch_attn_oproj_fused: bad dtype");
  return 1;
}

#endif  // OMCHAT_EXPERIMENTS
