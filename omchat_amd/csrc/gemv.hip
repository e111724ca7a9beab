// Weight-streaming skinny GEMM for decode:  Y[b,N] = epi(X[b,K] @ W[N,K]^T), b <= 32 (one or two 16-row batch tiles).  HBM-bound: every weight byte
// is read exactly once, straight from global memory into MFMA A-operand registers (no LDS round trip:
// cdna_hip_programming.md §5 "GEMV / M <= 16 decode weights").
//
// Replaces the decode-step `nn.Linear`s of transformers' Qwen2 (q/k/v/o/gate/up/down/lm_head with M = batch).
//
// One workgroup = 8 waves = NTILE*16 weight rows.  Wave w walks 64-wide K chunks w, w+8, w+16, ... : per chunk a
// lane loads 2 x 16 B of its weight row (A operand, row = lane&15, k = 8*(lane>>4)+j) and 2 x 16 B of x (B operand,
// column = batch row = lane&15; lanes >= b feed zeros), then 2 MFMA 16x16x32.  UNROLL chunks are in flight per wave
// (8 waves x UNROLL x 2 KiB per workgroup).  The 8 partial 16x16 fp32 tiles are summed through LDS in a fixed order
// (deterministic, no atomics), and wave 0 applies the epilogue.
#include "kernels.h"
#include "rowdot.h"
#include <type_traits>

namespace {

struct GemvP {
  const void* X; const void* W; void* Y; const void* bias; const void* resid;
  int ldx, ldw, ldy, ldr, b, N, K, out_f32, ksplit;
  const float* w_scale;
  int y_packed;
  const void* norm_w; float norm_eps;      // whole-row batch-1 form: X is the RAW hidden row, normalised in registers (gemv_rows_norm_kernel)
  unsigned* dyn;                           // gemv_rows_norm_dyn_kernel: 8 pool counters + 1 completion counter (one 256-byte line each), zero before the first launch
  void* y_pack;                            // gemv_xs_kernel<EPI_RESID>: packed copy of the result rows
  unsigned long long* dbg;                 // experiments build: clock stamps (measurement of the out-of-order prototype), else null
#if OMCHAT_EXPERIMENTS
  int xskew;                               // tuning key 28 < 256 (experiment): workgroups moved from every odd XCD's share to every even XCD's (gate|up non-loop form)
#endif
};
// stamps: [0] merge end, [1] o_proj first start (stored inverted: max of ~t), [2] o_proj flags seen, [3] o_proj end, [4] gate|up first start (inverted), [5] gate|up end,
// [6] down_proj first start (inverted), [7] down_proj end, [8] qkv first start (inverted), [9] qkv end, [10] attention first start (inverted), [11] attention end, [12] merge first start (inverted), [16 + x] gate|up end on XCD x
#if OMCHAT_EXPERIMENTS
#define OM_DBG_MIN(slot, cond) do { if (p.dbg && (cond) && threadIdx.x == 0) atomicMax(p.dbg + (slot), ~wall_clock64()); } while (0)
#define OM_DBG_MAX(slot, cond) do { if (p.dbg && (cond) && threadIdx.x == 0) atomicMax(p.dbg + (slot), wall_clock64()); } while (0)
#else
#define OM_DBG_MIN(slot, cond) do { } while (0)
#define OM_DBG_MAX(slot, cond) do { } while (0)
#endif

template <typename T, int NTILE, int N, int WAVES, bool NTL, int NB>
__device__ __forceinline__ void gemv_group(f32x4 (&acc)[NTILE][NB], const T* const (&wrow)[NTILE], const T* const (&xrow)[NB], int k0,
                                           const bool (&xvalid)[NB]) {
  typedef typename V8<T>::type frag_t;
  frag_t wf[N][NTILE][2], xf[N][NB][2];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int k = k0 + u * WAVES * 64;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      if constexpr (NTL) {
        wf[u][t][0] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(wrow[t] + k));
        wf[u][t][1] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(wrow[t] + k + 32));
      } else {
        wf[u][t][0] = ld8<T>(wrow[t] + k);
        wf[u][t][1] = ld8<T>(wrow[t] + k + 32);
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      xf[u][nb][0] = ld8<T>(xrow[nb] + k);
      xf[u][nb][1] = ld8<T>(xrow[nb] + k + 32);
    }
  }
  frag_t zero;
#pragma unroll
  for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;
#pragma unroll
  for (int u = 0; u < N; ++u) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (!xvalid[nb]) { xf[u][nb][0] = zero; xf[u][nb][1] = zero; }
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        acc[t][nb] = mfma16(wf[u][t][0], xf[u][nb][0], acc[t][nb]);
        acc[t][nb] = mfma16(wf[u][t][1], xf[u][nb][1], acc[t][nb]);
      }
    }
  }
}

template <typename T, int NTILE, int N, int WAVES, bool NTL, int NB>
__device__ __forceinline__ void gemv_tail(int rem, f32x4 (&acc)[NTILE][NB], const T* const (&wrow)[NTILE], const T* const (&xrow)[NB], int k0,
                                          const bool (&xvalid)[NB]) {
  if constexpr (N > 0) {
    if (rem == N) gemv_group<T, NTILE, N, WAVES, NTL, NB>(acc, wrow, xrow, k0, xvalid);
    else gemv_tail<T, NTILE, N - 1, WAVES, NTL, NB>(rem, acc, wrow, xrow, k0, xvalid);
  }
}

// NB = batch tiles of 16 rows sharing every weight fragment (b <= 16 * NB): the weights are still read exactly once
template <typename T, int NTILE, int EPI, int GV_WAVES = 8, int GV_UNROLL = 4, bool NTL = false, int NB = 1>
__global__ __launch_bounds__(GV_WAVES * 64) void gemv_kernel(GemvP p) {
  typedef typename V8<T>::type frag_t;
  __shared__ float red[GV_WAVES][NTILE * NB][256];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * (NTILE * 16);

  const T* W = (const T*)p.W;
  const T* X = (const T*)p.X;
  const T* wrow[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    int r = n0 + t * 16 + fr; r = r < p.N ? r : p.N - 1;
    wrow[t] = W + (size_t)r * p.ldw + fg * 8;
  }
  bool xvalid[NB];
  const T* xrow[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    xvalid[nb] = nb * 16 + fr < p.b;
    xrow[nb] = X + (size_t)(xvalid[nb] ? nb * 16 + fr : 0) * p.ldx + fg * 8;
  }

  f32x4 acc[NTILE][NB];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // K slice of this workgroup (blockIdx.y of ksplit): chunks [c_lo, c_hi)
  const int nchunk_all = p.K / 64;
  const int c_lo = (int)(((long)nchunk_all * blockIdx.y) / p.ksplit);
  const int nchunk = (int)(((long)nchunk_all * (blockIdx.y + 1)) / p.ksplit);
  // Wave w owns chunks c_lo + w, c_lo + w + WAVES, ...: full groups of GV_UNROLL chunks in flight, then ONE group of
  // exactly the remaining count (compile-time unrolled per count, so the tail is as deep as the body).
  const int first = c_lo + wave;
  const int n_w = first < nchunk ? (nchunk - first + GV_WAVES - 1) / GV_WAVES : 0;
  int g = 0;
  for (; g + GV_UNROLL <= n_w; g += GV_UNROLL)
    gemv_group<T, NTILE, GV_UNROLL, GV_WAVES, NTL, NB>(acc, wrow, xrow, (first + g * GV_WAVES) * 64, xvalid);
  gemv_tail<T, NTILE, GV_UNROLL - 1, GV_WAVES, NTL, NB>(n_w - g, acc, wrow, xrow, (first + g * GV_WAVES) * 64, xvalid);

  // acc[t][nb][r] = Y^T[n = n0 + t*16 + 4*fg + r][batch = 16*nb + fr]
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t * NB + nb][(fg * 4 + r) * 16 + fr] = acc[t][nb][r];
  __syncthreads();
  if (wave != 0) return;

  // wave 0: lane -> 4 (n, batch) pairs per tile; element e = lane + 64*i : n_local = e >> 4, batch = e & 15
  const T* bias = (const T*)p.bias;
  const T* R = (const T*)p.resid;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    float v[NTILE][4];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < GV_WAVES; ++w) s += red[w][t * NB + nb][lane + 64 * i];
        v[t][i] = s;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + 64 * i, nl = e >> 4, bi = nb * 16 + (e & 15);
      if (bi >= p.b) continue;
      if constexpr (EPI == EPI_PARTIAL) {
        // raw fp32 K-slice sums, layout [ksplit][b][ldy]; the consumer (resid_rmsnorm) adds the slices in a fixed order
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const int n = n0 + t * 16 + nl;
          if (n < p.N) ((float*)p.Y)[((size_t)blockIdx.y * p.b + bi) * p.ldy + n] = v[t][i];
        }
      } else if constexpr (EPI == EPI_SWIGLU) {
        // tile 0 = 16 gate rows, tile 1 = the matching 16 up rows
        const int n = (n0 >> 1) + nl;
        if (n0 + 16 + nl < p.N) {
          const float g = rnd<T>(v[0][i]), u = rnd<T>(v[1][i]);
          ((T*)p.Y)[(size_t)bi * p.ldy + n] = fromf<T>(rnd<T>(silu(g)) * u);
        }
      } else {
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const int n = n0 + t * 16 + nl;
          if (n >= p.N) continue;
          float y = v[t][i] + (bias ? tof(bias[n]) : 0.f);
          if (p.out_f32) {
            ((float*)p.Y)[(size_t)bi * p.ldy + n] = y;
          } else {
            y = rnd<T>(y);
            if constexpr (EPI == EPI_RESID) y = tof(R[(size_t)bi * p.ldr + n]) + y;
            ((T*)p.Y)[(size_t)bi * p.ldy + n] = fromf<T>(y);
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Batched decode (2 <= b <= 32) with PACKED operands (common.h: packed_x_index / packed_w_index).  The kernel above loads
// fragment-shaped operands (16 rows x 64 B per wave instruction) for W AND for x; at b = 32 the x loads (L2 hits, as many bytes
// as the weights) make it address-path bound: 3.2 TB/s on gate|up.  Here x arrives packed from its producer (RMSNorm, attention
// merge, SwiGLU epilogue) and W optionally from the packed replica, so every wave load is 1 KiB contiguous; NTILE row tiles share
// each x fragment.  Measured (tools/experiments/tune_gemv32.hip, MI355X, b = 32): gate|up 84 -> 66 us with packed x, -> 53 us with packed W
// too (5.1 TB/s); down_proj 50 -> 28 us; qkv 17.7 -> 12.2; o_proj 13.8 -> 9.0; lm_head 318 -> 207.
// Wave w of a workgroup walks K chunks (64 wide) c_lo + UNROLL * (w + WAVES * i) + u; partial tiles are summed through LDS in a
// fixed order by all waves, which also apply the epilogue.
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// x-stationary form of the packed batched GEMV for the long launches (gate|up, lm_head; K = 64 * 8 * NCH, no split-K).  In gemv_pk_kernel
// every workgroup re-reads the whole packed x (229 KB at K = 3584, NB = 2): as many bytes through the CU's load path as its share of
// the weights (tools/experiments/tune_gemv32.hip: with x served from L1 gate|up takes 47.6 instead of 53.3 us).  Here one persistent workgroup per
// CU keeps x in REGISTERS -- wave w owns K chunks w, w + 8, ... for the whole launch -- and walks 16-row weight tiles with stride
// gridDim.x; the eight partial tiles meet in LDS (parity double buffer: one barrier per tile) and the NEXT tile's weight fragments are
// loaded before the current tile is reduced, so the weight stream never stops.  Measured in the harness: gate|up 53.3 -> 48.6 us,
// lm_head 219 -> 191 us at b = 32.  SwiGLU: the workgroup takes tiles in (gate, up) pairs; the thread that reduces element (row, batch)
// of the gate tile reduces the same element of the up tile and keeps the gate value in a register.
// Same summation order per output as gemv_pk_kernel with WAVES = 8 would give only if its chunk -> wave deal were identical; it is not
// (there: groups of UNROLL consecutive chunks per wave), so results agree to fp32 rounding of the K sum, not bit for bit.
// ---------------------------------------------------------------------------------------------------------
// c_base: first K chunk of this workgroup's slice (the slice is exactly 8 * NCH chunks), slice: its index for the split-K output
// NORM (round 5, the batched twin of the batch-1 norm-in-GEMV forms): X holds the RAW residual rows in the packed layout and the RMSNorm that
// precedes the projection (transformers modeling_qwen2.py:247-252) runs here, on the registers that keep x for the whole launch: per-row sum
// of squares over the wave's chunks, over the four lanes of a row and over the eight waves (LDS), then T(T(x * rstd) * w) with the reference's
// two roundings.  Every workgroup repeats it (x is read by every workgroup anyway); the residual + RMSNorm launch in front of gate|up goes away.
template <typename T, int EPI, int NB, int NCH, bool NORM = false>
__device__ __forceinline__ void gemv_xs_body(const GemvP& p, float (&red)[2][8][NB][256], int c_base, int slice) {
  typedef typename V8<T>::type frag_t;
  constexpr int WAVES = 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nchunk_all = p.K >> 6, n_tiles = p.N >> 4;
  // The slice is dealt to the waves chunk by chunk: wave w owns chunks c_base + w, c_base + w + 8, ...  No guards inside the load groups on
  // purpose: a ragged variant with per-chunk validity tests kept the compiler from issuing a tile's 2 * NCH weight loads back to back and
  // measured 8 % SLOWER than gemv_pk_kernel.
  const T* W = (const T*)p.W;
  frag_t xf[NCH][NB][2];
  auto load_x = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = c_base + wave + WAVES * i;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          frag_t z;
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] = fromf<T>(0.f);
          xf[i][nb][h] = nb * 16 + fr < p.b ? ld8<T>((const T*)p.X + ((size_t)(c * 2 + h) * NB + nb) * 512 + lane * 8) : z;
        }
    }
  };
  // (x in FRONT of the first weight tile: the other order -- HBM requests first, x from L2 behind them -- measured slower, configs[2] decode
  // 4.33-4.35 vs 4.27-4.30 ms per step on one box, round 5)
  load_x();
  if constexpr (NORM) {
    float ss[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float v = tof(xf[i][nb][h][j]); a = fmaf(v, v, a); }
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      ss[nb] = a;
      if (fg == 0) red[1][wave][nb][fr] = a;          // parity 1: finish(0) writes parity 0 and every wave has passed its barrier before parity 1 is reused
    }
    __syncthreads();
    const T* nw = (const T*)p.norm_w;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += red[1][w][nb][fr];      // fixed order
      const float rstd = rsqrtf(tot / (float)p.K + p.norm_eps);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = c_base + wave + WAVES * i;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          __builtin_amdgcn_sched_barrier(0);      // one norm-weight fragment live at a time: hoisting all 2 NCH of them spilled 120 VGPRs
          const frag_t wv = ld8<T>(nw + (size_t)c * 64 + h * 32 + fg * 8);
          frag_t o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = fromf<T>(rnd<T>(tof(xf[i][nb][h][j]) * rstd) * tof(wv[j]));
          xf[i][nb][h] = o;
        }
      }
    }
  }
  auto load_w = [&](frag_t (&wf)[NCH][2], int tile) {
    const T* base = W + ((size_t)tile * nchunk_all + c_base) * 1024 + lane * 8;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) wf[i][h] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(base + (size_t)(wave + WAVES * i) * 1024 + h * 512));
  };
  // j-th tile of this workgroup: SwiGLU walks (gate, up) pairs, the others single tiles
  const int n_units = EPI == EPI_SWIGLU ? n_tiles >> 1 : n_tiles;
  const int my_units = (int)blockIdx.x < n_units ? (n_units - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int my_tiles = EPI == EPI_SWIGLU ? 2 * my_units : my_units;
  auto tile_of = [&](int j) { return EPI == EPI_SWIGLU ? 2 * ((int)blockIdx.x + (j >> 1) * (int)gridDim.x) + (j & 1) : (int)blockIdx.x + j * (int)gridDim.x; };
  const T* bias = (const T*)p.bias;
  float gate = 0.f;
  // plain epilogue with a bias (qkv): the bias element of the output this thread will reduce is requested when the tile's weights are -- loaded
  // after the reduction it was a dependent L2 round trip per tile (round 4).  No branch around the load: without a bias it reads x[0].
  const bool has_bias = EPI != EPI_SWIGLU && EPI != EPI_PARTIAL && bias != nullptr;
  const T* bsrc = has_bias ? bias : (const T*)p.X;
  auto bias_of = [&](int j) { const int e = threadIdx.x & 255, n = tile_of(j) * 16 + (e >> 4); return tof(bsrc[has_bias ? (n < p.N ? n : p.N - 1) : 0]); };
  // EPI_RESID (un-split o_proj of a batched step, round 5): the residual element this thread will add is requested with the tile's weights, like
  // the bias; it rides in the same two registers (a launch has a bias or a residual, never both)
  const bool has_res = EPI == EPI_RESID;
  const T* rsrc = has_res ? (const T*)p.resid : (const T*)p.X;
  auto res_of = [&](int j) {
    const int i = threadIdx.x & (NB * 256 - 1), nb = i >> 8, e = i & 255, n = tile_of(j) * 16 + (e >> 4), bi = nb * 16 + (e & 15);
    return tof(rsrc[has_res ? (size_t)(bi < p.b ? bi : 0) * p.ldr + (n < p.N ? n : p.N - 1) : 0]);
  };
  float bias_a = 0.f, bias_b = 0.f;
  auto finish = [&](frag_t (&wf)[NCH][2], int j, float bias_v) {
    const int tile = tile_of(j), par = j & 1;
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[nb] = mfma16(wf[i][0], xf[i][nb][0], acc[nb]);
        acc[nb] = mfma16(wf[i][1], xf[i][nb][1], acc[nb]);
      }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[par][wave][nb][(fg * 4 + r) * 16 + fr] = acc[nb][r];
    __syncthreads();
    const int i = threadIdx.x;
    if (i < NB * 256) {
      const int nb = i >> 8, e = i & 255, nl = e >> 4, bi = nb * 16 + (e & 15);
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += red[par][w][nb][e];      // fixed order: deterministic
      if constexpr (EPI == EPI_SWIGLU) {
        if ((j & 1) == 0) gate = rnd<T>(v);
        else if (bi < p.b) {
          const int n = (tile >> 1) * 16 + nl;
          const T y = fromf<T>(rnd<T>(silu(gate)) * rnd<T>(v));
          if (p.y_packed) ((T*)p.Y)[packed_x_index(bi, n, NB)] = y;
          else ((T*)p.Y)[(size_t)bi * p.ldy + n] = y;
        }
      } else if constexpr (EPI == EPI_PARTIAL) {
        if (bi < p.b) ((float*)p.Y)[((size_t)slice * p.b + bi) * p.ldy + tile * 16 + nl] = v;      // [ksplit][b][ldy]
      } else if constexpr (EPI == EPI_RESID) {
        if (bi < p.b) {                                   // T(resid + T(acc)) as gemm_epilogue's EPI_RESID; row-major (may alias resid) and packed
          const int n = tile * 16 + nl;
          const T y = fromf<T>(bias_v + rnd<T>(v));
          ((T*)p.Y)[(size_t)bi * p.ldy + n] = y;
          if (p.y_pack) ((T*)p.y_pack)[packed_x_index(bi, n, NB)] = y;
        }
      } else if (bi < p.b) {
        const int n = tile * 16 + nl;
        const float y = v + (has_bias ? bias_v : 0.f);
        if (p.out_f32) ((float*)p.Y)[(size_t)bi * p.ldy + n] = y;
        else ((T*)p.Y)[(size_t)bi * p.ldy + n] = fromf<T>(y);
      }
    }
  };
  frag_t wa[NCH][2], wb[NCH][2];
  auto pre_of = [&](int j) { if constexpr (EPI == EPI_RESID) return res_of(j); else return bias_of(j); };
  if (my_tiles > 0) { load_w(wa, tile_of(0)); bias_a = pre_of(0); }
  for (int j = 0; j < my_tiles; j += 2) {
    if (j + 1 < my_tiles) { load_w(wb, tile_of(j + 1)); bias_b = pre_of(j + 1); }
    finish(wa, j, bias_a);
    if (j + 1 < my_tiles) {
      if (j + 2 < my_tiles) { load_w(wa, tile_of(j + 2)); bias_a = pre_of(j + 2); }
      finish(wb, j + 1, bias_b);
    }
  }
}

template <typename T, int EPI, int NB, int NCH, bool NORM = false>
__global__ __launch_bounds__(512) void gemv_xs_kernel(GemvP p) {
  __shared__ float red[2][8][NB][256];
  gemv_xs_body<T, EPI, NB, NCH, NORM>(p, red, 0, 0);
}

// split-K form (down_proj): the K = 64 * 8 * q chunks are cut into slices that deal evenly to the 8 waves -- n5 slices of 40 chunks
// (5 per wave) then slices of 16 chunks (2 per wave), blockIdx.y = slice -- each written as one fp32 slice [ksplit][b][ldy] for the
// fused residual + RMSNorm kernel.  (q = 37 for K = 18944: 7 x 40 + 1 x 16 = 8 slices.)
template <typename T, int NB>
__global__ __launch_bounds__(512) void gemv_xs_split_kernel(GemvP p, int n5) {
  __shared__ float red[2][8][NB][256];
  const int y = blockIdx.y;
  if (y < n5) gemv_xs_body<T, EPI_PARTIAL, NB, 5>(p, red, y * 40, y);
  else gemv_xs_body<T, EPI_PARTIAL, NB, 2>(p, red, n5 * 40 + (y - n5) * 16, y);
}

template <typename T, int NTILE, int EPI, int WAVES, int UNROLL, int NB, bool WPACK>
__global__ __launch_bounds__(WAVES * 64) void gemv_pk_kernel(GemvP p) {
  typedef typename V8<T>::type frag_t;
  __shared__ float red[WAVES][NTILE * NB][256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int tile0 = blockIdx.x * NTILE;
  const int n_tiles = (p.N + 15) >> 4;
  const int nchunk_all = p.K >> 6;
  const int c_lo = (int)(((long)nchunk_all * blockIdx.y) / p.ksplit), c_hi = (int)(((long)nchunk_all * (blockIdx.y + 1)) / p.ksplit);

  const T* W = (const T*)p.W;
  const T* wbase[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int tl = tile0 + t < n_tiles ? tile0 + t : n_tiles - 1;
    if constexpr (WPACK) wbase[t] = W + (size_t)tl * nchunk_all * 1024 + lane * 8;
    else { int r = tl * 16 + fr; r = r < p.N ? r : p.N - 1; wbase[t] = W + (size_t)r * p.ldw + fg * 8; }
  }
  const T* xbase = (const T*)p.X + lane * 8;
  // lanes whose batch row (16 nb + fr) does not exist load nothing: the x re-read of every workgroup (half of the CU-side traffic at
  // b = 32, tools/experiments/tune_gemv32.hip) then scales with b instead of with the 16-row padding of the packed layout
  bool xrow[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) xrow[nb] = nb * 16 + fr < p.b;

  f32x4 acc[NTILE][NB];
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c0 = c_lo + wave * UNROLL; c0 < c_hi; c0 += WAVES * UNROLL) {
    frag_t wf[UNROLL][NTILE][2], xf[UNROLL][NB][2];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int c = c0 + u < c_hi ? c0 + u : c_hi - 1;          // ragged last group: re-load the last chunk, skip its MFMAs below
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const T* q = WPACK ? wbase[t] + (size_t)c * 1024 + h * 512 : wbase[t] + c * 64 + h * 32;
          wf[u][t][h] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(q));
        }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          frag_t z;
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] = fromf<T>(0.f);
          xf[u][nb][h] = xrow[nb] ? ld8<T>(xbase + ((size_t)(c * 2 + h) * NB + nb) * 512) : z;
        }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (c0 + u < c_hi) {                                      // wave-uniform
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int t = 0; t < NTILE; ++t) {
            acc[t][nb] = mfma16(wf[u][t][0], xf[u][nb][0], acc[t][nb]);
            acc[t][nb] = mfma16(wf[u][t][1], xf[u][nb][1], acc[t][nb]);
          }
      }
    }
  }
  // acc[t][nb][r] = Y^T[n = (tile0 + t)*16 + 4*fg + r][batch = 16*nb + fr]
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t * NB + nb][(fg * 4 + r) * 16 + fr] = acc[t][nb][r];
  __syncthreads();

  const T* bias = (const T*)p.bias;
  for (int i = threadIdx.x; i < NB * 256; i += WAVES * 64) {
    const int nb = i >> 8, e = i & 255, nl = e >> 4, bi = nb * 16 + (e & 15);
    float v[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) s += red[w][t * NB + nb][e];      // fixed order: deterministic
      v[t] = s;
    }
    if (bi >= p.b) continue;
    if constexpr (EPI == EPI_PARTIAL) {
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const int n = (tile0 + t) * 16 + nl;
        if (n < p.N) ((float*)p.Y)[((size_t)blockIdx.y * p.b + bi) * p.ldy + n] = v[t];
      }
    } else if constexpr (EPI == EPI_SWIGLU) {
      // tiles (2q, 2q + 1) = 16 gate rows and the matching 16 up rows of the fused, interleaved weight
#pragma unroll
      for (int t = 0; t < NTILE; t += 2) {
        const int row = (tile0 + t) * 16 + nl;                  // gate row in the fused layout
        if (row + 16 < p.N) {
          const int n = ((tile0 + t) >> 1) * 16 + nl;
          const float g = rnd<T>(v[t]), u = rnd<T>(v[t + 1]);
          const T y = fromf<T>(rnd<T>(silu(g)) * u);
          if (p.y_packed) ((T*)p.Y)[packed_x_index(bi, n, NB)] = y;
          else ((T*)p.Y)[(size_t)bi * p.ldy + n] = y;
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const int n = (tile0 + t) * 16 + nl;
        if (n >= p.N) continue;
        const float y = v[t] + (bias ? tof(bias[n]) : 0.f);
        if (p.out_f32) ((float*)p.Y)[(size_t)bi * p.ldy + n] = y;
        else ((T*)p.Y)[(size_t)bi * p.ldy + n] = fromf<T>(y);
      }
    }
  }
}

// row-major -> packed (model load time / tests)
template <typename T>
__global__ void pack_x_kernel(const T* X, int ldx, int b, int K, T* out, int NB) {
  const int n8 = NB * 16 * (K >> 3);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x) {
    const int row = i / (K >> 3), k = (i % (K >> 3)) * 8;
    typename V8<T>::type v;
    if (row < b) v = ld8<T>(X + (size_t)row * ldx + k);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    }
    st8<T>(out + packed_x_index(row, k, NB), v);
  }
}
template <typename T>
__global__ void pack_w_kernel(const T* W, int ldw, int N, int K, T* out) {
  const long n8 = (long)N * (K >> 3);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    // destination-linear: consecutive threads write consecutive 16-byte pieces of the packed image
    const long piece = i;
    const int lane = (int)(piece & 63), half = (int)((piece >> 6) & 1);
    const long tc = piece >> 7;
    const int chunk = (int)(tc % (K >> 6)), tile = (int)(tc / (K >> 6));
    const int row = tile * 16 + (lane & 15), k = chunk * 64 + half * 32 + (lane >> 4) * 8;
    st8<T>(out + piece * 8, ld8<T>(W + (size_t)row * ldw + k));
  }
}

// ---------------------------------------------------------------------------------------------------------
// b = 1: whole-row streaming + packed dot products.  Measured on MI355X (tools/experiments/tune_rowdot.hip): 6.5-6.8 TB/s on the
// gate|up shape against 5.5 TB/s for the MFMA form above, whose 16-row x 64-B fragment reads reach ~6.0 TB/s at best; here
// every wave instruction reads 1 KiB of ONE row (non-temporal).  x (this workgroup's K slice) sits in registers; a wave
// takes R rows at a time (all R x NCH loads in flight), v_dot2 per 32 bits, butterfly over the wave, lane 0 applies the
// epilogue.  Split-K slices are chunk ranges of 512 elements, <= RW_MAXC chunks each.
// ---------------------------------------------------------------------------------------------------------
int g_gemv_force_mfma = 0;     // tuning knob (omchat_op_set_tuning key 1): A/B the two forms
int g_gemv_no_xs = 0;          // tuning knob (key 11): 1 = batched decode never takes the x-stationary persistent kernel (A/B)
// omchat_op_set_tuning key 34: launch shapes for the SHARD widths of a tensor-parallel rank (round 5; N or K an eighth of the model's):
// bit 0 = x-stationary form for short EPI_NONE outputs (qkv shard: one tile per workgroup), bit 1 = one-tile / one-chunk-per-wave form for
// short-K split-K slices (o_proj / down_proj shards), bit 2 = one (gate, up) pair per wave for the short batch-1 gate|up,
// bit 3 = x-stationary form, one (gate, up) tile pair per workgroup, for a batched gate|up shard of at most one pair per CU
int g_gemv_shard = 15;
constexpr int RW_MAXC = 8;
// weight-only fp8 (OCP e4m3): 8 weights of a lane = 8 bytes.  gfx950's v_cvt_scalef32_pk_{bf16,f16}_fp8 widens two e4m3
// values to a packed 16-bit pair in one instruction (exact: 3 mantissa bits), which then feeds the same v_dot2 as the
// 16-bit kernel with x still packed in registers: 4 converts + 4 dot2 per 8 weights.
typedef unsigned int rw_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ float rw_dot8_fp8(rw_u32x2 w, rw_u32x4 x, float acc);
template <> __device__ __forceinline__ float rw_dot8_fp8<bf16>(rw_u32x2 w, rw_u32x4 x, float acc) {
  typedef bf16 v2 __attribute__((ext_vector_type(2)));
  const unsigned w0 = w.x, w1 = w.y, x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, false), __builtin_bit_cast(v2, x0), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w0, 1.0f, true), __builtin_bit_cast(v2, x1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, false), __builtin_bit_cast(v2, x2), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w1, 1.0f, true), __builtin_bit_cast(v2, x3), acc, false);
  return acc;
}
template <> __device__ __forceinline__ float rw_dot8_fp8<f16>(rw_u32x2 w, rw_u32x4 x, float acc) {
  typedef f16 v2 __attribute__((ext_vector_type(2)));
  const unsigned w0 = w.x, w1 = w.y, x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
  acc = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, false), __builtin_bit_cast(v2, x0), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, true), __builtin_bit_cast(v2, x1), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, false), __builtin_bit_cast(v2, x2), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, true), __builtin_bit_cast(v2, x3), acc, false);
  return acc;
}

// epilogue operands of a one-shot row GEMV, requested right BEHIND the weight rows (they return with them instead of costing a dependent L2
// round trip after the reduction; in front of the weights a cold 2-byte load would hold the whole in-order queue): two 16-bit vectors
// (bias, residual; a dummy valid pointer + index 0 when absent -- no branch around the loads)
template <typename T, int R> struct RwTail { const T* p0; const T* p1; int idx[R]; float v0[R], v1[R]; };
template <typename T, int R> __device__ __forceinline__ void rw_tail_load(RwTail<T, R>& t) {
#pragma unroll
  for (int r = 0; r < R; ++r) { t.v0[r] = tof(t.p0[t.idx[r]]); t.v1[r] = tof(t.p1[t.idx[r]]); }
}
template <typename T, int R, int NCH>
__device__ __forceinline__ void rw_rows_fp8(const unsigned char* W, int ldw, const int (&rows)[R], int k0, int K, int lane, const rw_u32x4 (&xr)[RW_MAXC],
                                            const float* scale, float (&acc)[R], RwTail<T, R>& tail) {
  rw_u32x2 w[R][NCH];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int k = k0 + c * 512 + lane * 8;
      k = k < K ? k : k0;
      w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x2*>(W + (size_t)rows[r] * ldw + k));
    }
  float sc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) sc[r] = scale[rows[r]];
  rw_tail_load<T, R>(tail);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) a = rw_dot8_fp8<T>(w[r][c], xr[c], a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    acc[r] = a * sc[r];
  }
}

template <typename T, int R, int N>
__device__ __forceinline__ void rw_dispatch_fp8(int nch, const unsigned char* W, int ldw, const int (&rows)[R], int k0, int K, int lane,
                                                const rw_u32x4 (&xr)[RW_MAXC], const float* scale, float (&acc)[R], RwTail<T, R>& tail) {
  if constexpr (N > 0) {
    if (nch == N) rw_rows_fp8<T, R, N>(W, ldw, rows, k0, K, lane, xr, scale, acc, tail);
    else rw_dispatch_fp8<T, R, N - 1>(nch, W, ldw, rows, k0, K, lane, xr, scale, acc, tail);
  }
}

// rows r0 .. r0+R-1 (already mapped to weight-row indices by the caller) over NCH chunks starting at element k0
template <typename T, int R, int NCH>
__device__ __forceinline__ void rw_rows(const T* W, int ldw, const int (&rows)[R], int k0, int K, int lane, const rw_u32x4 (&xr)[RW_MAXC],
                                        float (&acc)[R], RwTail<T, R>& tail) {
  rw_u32x4 w[R][NCH];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int k = k0 + c * 512 + lane * 8;
      k = k < K ? k : k0;                       // ragged last chunk: clamp (x is zero there)
      w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>(W + (size_t)rows[r] * ldw + k));
    }
  rw_tail_load<T, R>(tail);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) a = rw_dot8<T>(w[r][c], xr[c], a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    acc[r] = a;
  }
}

template <typename T, int R, int N>
__device__ __forceinline__ void rw_dispatch(int nch, const T* W, int ldw, const int (&rows)[R], int k0, int K, int lane,
                                            const rw_u32x4 (&xr)[RW_MAXC], float (&acc)[R], RwTail<T, R>& tail) {
  if constexpr (N > 0) {
    if (nch == N) rw_rows<T, R, N>(W, ldw, rows, k0, K, lane, xr, acc, tail);
    else rw_dispatch<T, R, N - 1>(nch, W, ldw, rows, k0, K, lane, xr, acc, tail);
  }
}

template <typename T, int EPI, int RR = 4, int WAVES = 4, bool F8 = false>
__global__ __launch_bounds__(WAVES * 64) void gemv_rows_kernel(GemvP p) {
  constexpr int R = EPI == EPI_SWIGLU ? 2 * RR : RR;      // SwiGLU: RR (gate, up) row pairs per group
  constexpr int OUT = RR;                                 // outputs per group
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* W = (const T*)p.W;
  if constexpr (EPI == EPI_RESID && WAVES == 7) OM_DBG_MIN(1, blockIdx.x < 8);
  // K slice of this workgroup row (blockIdx.y): chunk range [c_lo, c_hi)
  const int nch_all = (p.K + 511) / 512;
  const int c_lo = (int)(((long)nch_all * blockIdx.y) / p.ksplit), c_hi = (int)(((long)nch_all * (blockIdx.y + 1)) / p.ksplit);
  const int nch = c_hi - c_lo, k0 = c_lo * 512;
  rw_u32x4 xr[RW_MAXC];
#pragma unroll
  for (int c = 0; c < RW_MAXC; ++c) {
    const int k = k0 + c * 512 + lane * 8;
    rw_u32x4 z = {0u, 0u, 0u, 0u};
    xr[c] = (c < nch && k < p.K) ? *reinterpret_cast<const rw_u32x4*>((const T*)p.X + k) : z;
  }
  const int n_out = EPI == EPI_SWIGLU ? p.N / 2 : p.N;
  const int ngroups = (n_out + OUT - 1) / OUT;
  for (int g = blockIdx.x * WAVES + wave; g < ngroups; g += gridDim.x * WAVES) {
    int rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (EPI == EPI_SWIGLU) {
        int n = g * RR + (r % RR); n = n < n_out ? n : n_out - 1;
        rows[r] = 32 * (n >> 4) + (n & 15) + (r / RR) * 16;       // gate rows, then the matching up rows
      } else {
        const int n = g * R + r;
        rows[r] = n < p.N ? n : p.N - 1;
      }
    }
    float acc[R];
    RwTail<T, R> tail;
    {
      constexpr bool plain = EPI != EPI_SWIGLU && EPI != EPI_PARTIAL;
      const bool hb = plain && p.bias, hr = EPI == EPI_RESID && !p.out_f32;
      tail.p0 = hb ? (const T*)p.bias : (const T*)p.X;
      tail.p1 = hr ? (const T*)p.resid : (const T*)p.X;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = g * OUT + r;
        tail.idx[r] = (hb || hr) ? (n < n_out ? n : n_out - 1) : 0;
      }
      if (!hb) tail.p0 = tail.p1;      // (one of the two may be absent: read the other's element twice rather than element idx of x)
      if (!hr) tail.p1 = tail.p0;
    }
    if constexpr (F8) rw_dispatch_fp8<T, R, RW_MAXC>(nch, (const unsigned char*)p.W, p.ldw, rows, k0, p.K, lane, xr, p.w_scale, acc, tail);
    else rw_dispatch<T, R, RW_MAXC>(nch, W, p.ldw, rows, k0, p.K, lane, xr, acc, tail);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < OUT; ++r) {
        const int n = g * OUT + r;
        if (n >= n_out) continue;
        if constexpr (EPI == EPI_SWIGLU) {
          const float gt = rnd<T>(acc[r]), up = rnd<T>(acc[r + RR]);
          ((T*)p.Y)[n] = fromf<T>(rnd<T>(silu(gt)) * up);
        } else if constexpr (EPI == EPI_PARTIAL) {
          ((float*)p.Y)[(size_t)blockIdx.y * p.ldy + n] = acc[r];        // [ksplit][1][ldy]
        } else {
          float y = acc[r] + (p.bias ? tail.v0[r] : 0.f);
          if (p.out_f32) ((float*)p.Y)[n] = y;
          else {
            y = rnd<T>(y);
            if constexpr (EPI == EPI_RESID) y = tail.v1[r] + y;
            ((T*)p.Y)[n] = fromf<T>(y);
          }
        }
      }
    }
  }
  if constexpr (EPI == EPI_RESID && WAVES == 7) OM_DBG_MAX(3, (blockIdx.x & 7) == 0);
}

// ---------------------------------------------------------------------------------------------------------
// Whole-row GEMV with the RMSNorm that precedes the projection computed in registers (round 3; batch 1, no split-K, K = 512 NCH <= 4096):
//   y = epi(W xn),  xn = T(w * T(x * rsqrt(mean(x^2) + eps)))      (Qwen2RMSNorm's rounding points, modeling_qwen2.py:247-252)
// X is the RAW hidden row.  One group of rows per wave; ALL of the group's weight loads are issued first, then the workgroup's four
// waves normalise the row together through LDS (x and the norm weights are read once per WORKGROUP) while those loads are in flight,
// the dot products last.  Measured forms that did not pay: the chain in front of the weight loads, every wave for itself (gate|up
// 41.6 -> 45.8 us: exactly the launch it removes); every wave for itself with half the rows per wave to make room (48.3 us: x and the norm
// weights were then half as many bytes through the CU as the weights).  All waves use the same sum (fixed order): the same bits.
// It removes the residual + RMSNorm launch in front of the gate|up GEMV of a batch-1 decode step (o_proj then writes x + attn itself).
// ---------------------------------------------------------------------------------------------------------
template <typename T, int EPI, int RR, int NCH, bool F8, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64) void gemv_rows_norm_kernel(GemvP p) {
  typedef typename V8<T>::type v8;
  constexpr int R = EPI == EPI_SWIGLU ? 2 * RR : RR;
  __shared__ __attribute__((aligned(16))) T xs[NCH * 512];
  __shared__ float red[WAVES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_out = EPI == EPI_SWIGLU ? p.N / 2 : p.N;
  int g = blockIdx.x * WAVES + wave;
#if OMCHAT_EXPERIMENTS
  if constexpr (EPI == EPI_SWIGLU && RR == 1 && WAVES == 4) {
    // experiment (tuning key 28): the odd XCDs end this launch 2-3 us after the even ones (clock stamps), and workgroup ids go
    // round the XCDs -- so XCD x gets a contiguous range of pairs, p.xskew workgroups longer on even x and as much shorter on odd x
    if (p.xskew) {
      const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
      const int base = ((n_out + 3) / 4) / 8;
      const int n_x = base + ((x & 1) ? -p.xskew : p.xskew);
      if (j >= n_x) return;
      g = (x * base + ((x & 1) ? p.xskew : 0) + j) * 4 + wave;
    }
  }
#endif
  if constexpr (EPI == EPI_SWIGLU) OM_DBG_MIN(4, blockIdx.x < 8);
  if constexpr (EPI == EPI_NONE) OM_DBG_MIN(8, blockIdx.x < 8);
  // ---- 0. x and the norm weights are requested IN FRONT of the weight rows (round 4).  Vector memory returns in order: behind the weight
  // loads the 2 x 7 KB of x / norm weights (L2 hits) could not be used before the wave's whole first batch of weights had come in from HBM, and
  // the norm -- two barriers and two LDS round trips -- then ran with nothing left in flight behind it.  In front, the norm is done while the
  // weights are still on their way and the dot products start when they land: qkv 7.95-8.33 -> 7.13 us, e4m3 decode 1.83 -> 1.73 ms per token.
  // (Not in the loop form and not in the long-K kernel: there the same order measured SLOWER, gate|up 42.27 -> 43.00 us, down_proj 23.13 ->
  // 23.49 us -- every wave of the chip asks for the same 14 KB first, and the weight requests of a CU queue behind that hot spot; with
  // three buffers in flight the late norm was hidden already; x BETWEEN the first buffer and the other two measured 42.8 us as well.
  // tools/bench_gemv_b1.py, profiles/r04_al.)
  static_assert(WAVES >= 4 && NCH <= 8, "sum-of-squares order: four owner waves, up to two chunks each");
  constexpr int MC = (NCH + WAVES - 1) / WAVES;
  constexpr int SC = (NCH + 3) / 4;
  rw_u32x4 xq[MC], nq[MC], sq[SC];
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const int c = wave + WAVES * i, k = c * 512 + lane * 8;
    const rw_u32x4 z = {0u, 0u, 0u, 0u};
    const bool ok = c < NCH && k < p.K;
    xq[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)p.X + k) : z;
    nq[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)p.norm_w + k) : z;
  }
#pragma unroll
  for (int i = 0; i < SC; ++i) {
    if constexpr (WAVES == 4) {
      sq[i] = xq[i];
    } else {
      const int c = wave + 4 * i, k = c * 512 + lane * 8;
      const rw_u32x4 z = {0u, 0u, 0u, 0u};
      sq[i] = (wave < 4 && c < NCH && k < p.K) ? *reinterpret_cast<const rw_u32x4*>((const T*)p.X + k) : z;
    }
  }
  // ---- 1. this wave's weight rows: every load issued now
  int rows[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if constexpr (EPI == EPI_SWIGLU) {
      int n = g * RR + (r % RR); n = n < n_out ? n : n_out - 1;
      rows[r] = 32 * (n >> 4) + (n & 15) + (r / RR) * 16;       // gate rows, then the matching up rows
    } else {
      const int n = g * R + r;
      rows[r] = n < p.N ? n : p.N - 1;
    }
  }
  typedef typename std::conditional<F8, rw_u32x2, rw_u32x4>::type wreg_t;
  wreg_t w[R][NCH];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int k = c * 512 + lane * 8;
      k = k < p.K ? k : 0;                      // ragged last chunk: clamp (x is zero there)
      if constexpr (F8) w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x2*>((const unsigned char*)p.W + (size_t)rows[r] * p.ldw + k));
      else w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)p.W + (size_t)rows[r] * p.ldw + k));
    }
  // the epilogue's operands (bias, e4m3 row scales) are requested right BEHIND the weight rows: they come back with them instead of costing a
  // dependent L2 round trip after the reduction.  (In FRONT of the weights they measured slower: a cold 2-byte load at the head of the
  // in-order queue holds every weight row behind it -- profiles/r04_al.)  No branch around the loads: a null bias reads the norm weights.
  float e_bias[RR], e_scale[R];
  {
    const T* bp = (EPI != EPI_SWIGLU && p.bias) ? (const T*)p.bias : (const T*)p.norm_w;
    const int nmax = (EPI != EPI_SWIGLU && p.bias) ? n_out - 1 : 0;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int n = g * RR + r;
      e_bias[r] = tof(bp[n < nmax ? n : nmax]);
    }
    if (!(EPI != EPI_SWIGLU && p.bias)) {
#pragma unroll
      for (int r = 0; r < RR; ++r) e_bias[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) e_scale[r] = 1.f;
    if constexpr (F8) {
#pragma unroll
      for (int r = 0; r < R; ++r) e_scale[r] = p.w_scale[rows[r]];
    }
  }
  // ---- 2. the norm, shared by the workgroup: wave w normalises chunks w, w + WAVES, ... (x and the norm weights are read ONCE per
  // workgroup: 2 x 7 KB instead of 7 KB of xn per wave), the normalised row goes to LDS, every wave reads it back.  The plain loads above
  // stay in flight across the two barriers.
  // The sum of squares is ALWAYS taken in the four-wave order -- waves 0..3 own chunks w and w + 4, per-lane accumulation over the two
  // chunks, butterfly, ((r0 + r1) + r2) + r3 -- whatever WAVES is, so that every launch form of a projection (4, 7 or 9 waves, the loop
  // form) gives the same bits on every device (the nine-wave form used to sum its nine per-chunk partials: ADVICE r03).
  __builtin_amdgcn_sched_barrier(0);            // keep every load above the first wait
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < SC; ++i) {
    const v8 xv = __builtin_bit_cast(v8, sq[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = tof(xv[j]); ss += v * v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0 && wave < 4) red[wave] = ss;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float tot = red[0];                           // fixed order: the same bits in every wave and in every launch form
#pragma unroll
  for (int w = 1; w < 4; ++w) tot += red[w];
  const float inv = rsqrtf(tot / (float)p.K + p.norm_eps);
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const int c = wave + WAVES * i;
    if (c < NCH) {
      const v8 xv = __builtin_bit_cast(v8, xq[i]), wv = __builtin_bit_cast(v8, nq[i]);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[j]) * rnd<T>(tof(xv[j]) * inv));
      *reinterpret_cast<v8*>(xs + c * 512 + lane * 8) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rw_u32x4 xr[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) xr[c] = *reinterpret_cast<const rw_u32x4*>(xs + c * 512 + lane * 8);
  __builtin_amdgcn_sched_barrier(0);
  // ---- 3. dot products
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if constexpr (F8) a = rw_dot8_fp8<T>(w[r][c], xr[c], a);
      else a = rw_dot8<T>(w[r][c], xr[c], a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if constexpr (F8) a *= e_scale[r];
    acc[r] = a;
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int n = g * RR + r;
      if (n >= n_out) continue;
      if constexpr (EPI == EPI_SWIGLU) {
        const float gt = rnd<T>(acc[r]), up = rnd<T>(acc[r + RR]);
        ((T*)p.Y)[n] = fromf<T>(rnd<T>(silu(gt)) * up);
      } else {
        const float y = acc[r] + e_bias[r];
        if (p.out_f32) ((float*)p.Y)[n] = y;
        else ((T*)p.Y)[n] = fromf<T>(y);
      }
    }
  }
  if constexpr (EPI == EPI_SWIGLU) { OM_DBG_MAX(5, blockIdx.x + 64 >= gridDim.x); OM_DBG_MAX(16 + (blockIdx.x & 7), blockIdx.x + 512 >= gridDim.x); }      // + the end per XCD
  if constexpr (EPI == EPI_NONE) OM_DBG_MAX(9, (blockIdx.x & 7) == 0);
}

// ---------------------------------------------------------------------------------------------------------
// Loop form of gemv_rows_norm_kernel (round 3): ONE resident round of workgroups (2 per CU), each owning a contiguous range of outputs
// and every wave walking its share one output at a time (a row, or a (gate, up) pair) through a ring of three register buffers -- two
// outputs' weights in flight under the dot products of the third.  Why: the one-shot form deals 12 outputs per workgroup, and gate|up has
// 18944 = 37 x 512 of them: 1579 workgroups on 512 slots leave a fourth, 8 %-full round whose 43 workgroups finish alone on 43 CUs; here
// every workgroup gets 37 outputs (waves 10 / 9 / 9 / 9) and the launch ends everywhere at once.  The norm prologue is the same and runs
// once per workgroup.  Per output the arithmetic and its order are those of the one-shot form: the same bits.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int EPI, int NCH, bool F8>
__global__ __launch_bounds__(256) void gemv_rows_norm_loop_kernel(GemvP p, int per_wg, unsigned skew) {
  typedef typename V8<T>::type v8;
  constexpr int R = EPI == EPI_SWIGLU ? 2 : 1;
  __shared__ __attribute__((aligned(16))) T xs[NCH * 512];
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_out = EPI == EPI_SWIGLU ? p.N / 2 : p.N;
  int o0 = blockIdx.x * per_wg, o1 = o0 + per_wg < n_out ? o0 + per_wg : n_out;
  if constexpr (EPI == EPI_SWIGLU) OM_DBG_MIN(4, blockIdx.x < 8);
  if (skew) {      // experiment (tuning key 28): shares by blockIdx % 8, per_wg + d_l with d_l = nibble l of skew - 8 (the deltas sum to 0)
    const int l = blockIdx.x & 7;
    int pre = 0;
    for (int i = 0; i < l; ++i) pre += per_wg + (int)((skew >> (4 * i)) & 15u) - 8;
    o0 = (blockIdx.x >> 3) * 8 * per_wg + pre;
    o1 = o0 + per_wg + (int)((skew >> (4 * l)) & 15u) - 8;
    o1 = o1 < n_out ? o1 : n_out;
  }
  typedef typename std::conditional<F8, rw_u32x2, rw_u32x4>::type wreg_t;
  auto row_of = [&](int n, int r) { return EPI == EPI_SWIGLU ? 32 * (n >> 4) + (n & 15) + r * 16 : n; };
  auto load_w = [&](wreg_t (&w)[R][NCH], int n) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const size_t row = (size_t)row_of(n, r);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int k = c * 512 + lane * 8;
        k = k < p.K ? k : 0;                    // ragged last chunk: clamp (x is zero there)
        if constexpr (F8) w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x2*>((const unsigned char*)p.W + row * p.ldw + k));
        else w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)p.W + row * p.ldw + k));
      }
    }
  };
  // ---- 1. the first three outputs of this wave: every load issued now
  wreg_t wa[R][NCH], wb[R][NCH], wc[R][NCH];
  const int u0 = o0 + wave;
  if (u0 < o1) load_w(wa, u0);
  if (u0 + 4 < o1) load_w(wb, u0 + 4);
  if (u0 + 8 < o1) load_w(wc, u0 + 8);
  // ---- 2. the norm, shared by the workgroup (gemv_rows_norm_kernel step 2)
  constexpr int MC = (NCH + 3) / 4;
  rw_u32x4 xq[MC], nq[MC];
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const int c = wave + 4 * i, k = c * 512 + lane * 8;
    const rw_u32x4 z = {0u, 0u, 0u, 0u};
    const bool ok = c < NCH && k < p.K;
    xq[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)p.X + k) : z;
    nq[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)p.norm_w + k) : z;
  }
  __builtin_amdgcn_sched_barrier(0);            // keep every load above the first wait
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const v8 xv = __builtin_bit_cast(v8, xq[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = tof(xv[j]); ss += v * v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0) red[wave] = ss;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const float inv = rsqrtf((((red[0] + red[1]) + red[2]) + red[3]) / (float)p.K + p.norm_eps);
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const int c = wave + 4 * i;
    if (c < NCH) {
      const v8 xv = __builtin_bit_cast(v8, xq[i]), wv = __builtin_bit_cast(v8, nq[i]);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[j]) * rnd<T>(tof(xv[j]) * inv));
      *reinterpret_cast<v8*>(xs + c * 512 + lane * 8) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rw_u32x4 xr[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) xr[c] = *reinterpret_cast<const rw_u32x4*>(xs + c * 512 + lane * 8);
  __builtin_amdgcn_sched_barrier(0);
  // ---- 3. dot products, one output at a time; the buffer just consumed is refilled with the output three steps ahead
  auto finish = [&](wreg_t (&w)[R][NCH], int n) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if constexpr (F8) a = rw_dot8_fp8<T>(w[r][c], xr[c], a);
        else a = rw_dot8<T>(w[r][c], xr[c], a);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if constexpr (F8) a *= p.w_scale[row_of(n, r)];
      acc[r] = a;
    }
    if (lane == 0) {
      if constexpr (EPI == EPI_SWIGLU) {
        const float gt = rnd<T>(acc[0]), up = rnd<T>(acc[R - 1]);
        ((T*)p.Y)[n] = fromf<T>(rnd<T>(silu(gt)) * up);
      } else {
        const float y = acc[0] + (p.bias ? tof(((const T*)p.bias)[n]) : 0.f);
        if (p.out_f32) ((float*)p.Y)[n] = y;
        else ((T*)p.Y)[n] = fromf<T>(y);
      }
    }
  };
  for (int u = u0; u < o1; u += 12) {
    finish(wa, u);
    if (u + 12 < o1) load_w(wa, u + 12);
    if (u + 4 < o1) {
      finish(wb, u + 4);
      if (u + 16 < o1) load_w(wb, u + 16);
    }
    if (u + 8 < o1) {
      finish(wc, u + 8);
      if (u + 20 < o1) load_w(wc, u + 20);
    }
  }
  if constexpr (EPI == EPI_SWIGLU) {      // measurement (experiments build): start / end of the launch and the end per XCD (workgroup ids go round the 8 XCDs)
    OM_DBG_MAX(5, (blockIdx.x & 7) == 0);
    OM_DBG_MAX(16 + (blockIdx.x & 7), true);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Dynamic form of the loop kernel (round 4).  The loop form deals every workgroup the same number of outputs and ends when the SLOWEST
// workgroup ends -- and the eight XCDs of this chip do not stream at the same rate: with equal shares the gate|up phase of a CU took
// 32.8 / 39.9 / 34.7 / 37.8 / 33.0 / 38.2 / 35.1 / 38.1 us by blockIdx % 8 (= the XCD group under round-robin dispatch; phase stamps of
// tools/experiments/tune_layer.hip), i.e. the launch waits ~10 % for three of the XCDs.  Here the outputs are cut into chunks of DYN_CH and dealt by
// atomic counters: DYN_POOLS pools (workgroup b starts at pool b % DYN_POOLS, so pool p is drained by one XCD group, p % 8), a wave takes
// chunks from its pool and, when that is empty, from the following pools -- the fast XCDs end up with more chunks.  (Eight pools measured
// 80 us per launch: a counter under streaming load answers ~15 tickets per us, not the 88 of an idle chip.)
// Which wave computes an output changes nothing in its arithmetic: the same bits as the loop form.  Placement independent: the pool index
// is only a shard label.  A chunk id is requested BEFORE the loads of the previous chunk are issued and used when that chunk is half
// done, so the returning atomic never waits behind weight loads that are younger than the ones the wave needs next.
// Counters: one 256-byte line each (nine counters in ONE line took 331 us per launch: the memory-side atomic unit serialises a line):
// p.dyn[64 g] pool g, p.dyn[64 DYN_POOLS] finished waves; the last wave to finish zeroes them for the next launch on the same slot.
// ---------------------------------------------------------------------------------------------------------
constexpr int DYN_CH = 2;
constexpr int DYN_POOLS = 64;      // work counters per launch, one 256-byte line each (a counter then sees ~4 grabs per us)
template <typename T, int EPI, int NCH, bool F8>
__global__ __launch_bounds__(256) void gemv_rows_norm_dyn_kernel(GemvP p) {
  typedef typename V8<T>::type v8;
  constexpr int R = EPI == EPI_SWIGLU ? 2 : 1;
  __shared__ __attribute__((aligned(16))) T xs[NCH * 512];
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_out = EPI == EPI_SWIGLU ? p.N / 2 : p.N;
  const int n_chunks = (n_out + DYN_CH - 1) / DYN_CH, pool_sz = (n_chunks + DYN_POOLS - 1) / DYN_POOLS;
  typedef typename std::conditional<F8, rw_u32x2, rw_u32x4>::type wreg_t;
  auto row_of = [&](int n, int r) { return EPI == EPI_SWIGLU ? 32 * (n >> 4) + (n & 15) + r * 16 : n; };
  auto load_w = [&](wreg_t (&w)[R][NCH], int n) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const size_t row = (size_t)row_of(n, r);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int k = c * 512 + lane * 8;
        k = k < p.K ? k : 0;
        if constexpr (F8) w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x2*>((const unsigned char*)p.W + row * p.ldw + k));
        else w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)p.W + row * p.ldw + k));
      }
    }
  };
  // ---- chunk queue of this wave.  `pool` walks the eight pools starting at the workgroup's own; a raw ticket >= the pool's size means
  // that pool is empty: move on (blocking grabs from then on: the tail of the launch).
  int pool = 0;                                   // pools tried so far (DYN_POOLS = nothing left anywhere)
  const int g0 = blockIdx.x % DYN_POOLS;
  auto pool_count = [&](int g) { const int lo = g * pool_sz; return lo >= n_chunks ? 0 : (lo + pool_sz <= n_chunks ? pool_sz : n_chunks - lo); };
  auto ticket = [&]() -> unsigned {               // one ticket from the current pool (returning atomic; wave-uniform through lane 0)
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(p.dyn + 64 * ((g0 + pool) % DYN_POOLS), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  auto resolve = [&](unsigned t) -> int {         // ticket -> chunk id, or -1 when every pool is empty
    int tk = (int)__builtin_amdgcn_readfirstlane(t);
    for (;;) {
      const int g = (g0 + pool) % DYN_POOLS;
      if (tk < pool_count(g)) return g * pool_sz + tk;
      if (++pool >= DYN_POOLS) return -1;
      tk = (int)__builtin_amdgcn_readfirstlane(ticket());
    }
  };
  // ---- 1. the first two chunks: tickets first, then the first chunk's loads as soon as its id is known
  int cur = resolve(ticket());
  unsigned nxt_t = cur >= 0 ? ticket() : 0u;
  wreg_t wa[R][NCH], wb[R][NCH], wc[R][NCH];
  auto out_ok = [&](int chunk, int i) { return chunk >= 0 && chunk * DYN_CH + i < n_out; };
  if (out_ok(cur, 0)) load_w(wa, cur * DYN_CH);
  if (out_ok(cur, 1)) load_w(wb, cur * DYN_CH + 1);
  // ---- 2. the norm, shared by the workgroup (gemv_rows_norm_kernel step 2; four-wave order)
  constexpr int MC = (NCH + 3) / 4;
  rw_u32x4 xq[MC], nq[MC];
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const int c = wave + 4 * i, k = c * 512 + lane * 8;
    const rw_u32x4 z = {0u, 0u, 0u, 0u};
    const bool ok = c < NCH && k < p.K;
    xq[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)p.X + k) : z;
    nq[i] = ok ? *reinterpret_cast<const rw_u32x4*>((const T*)p.norm_w + k) : z;
  }
  __builtin_amdgcn_sched_barrier(0);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const v8 xv = __builtin_bit_cast(v8, xq[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float v = tof(xv[j]); ss += v * v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0) red[wave] = ss;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const float inv = rsqrtf((((red[0] + red[1]) + red[2]) + red[3]) / (float)p.K + p.norm_eps);
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    const int c = wave + 4 * i;
    if (c < NCH) {
      const v8 xv = __builtin_bit_cast(v8, xq[i]), wv = __builtin_bit_cast(v8, nq[i]);
      v8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fromf<T>(tof(wv[j]) * rnd<T>(tof(xv[j]) * inv));
      *reinterpret_cast<v8*>(xs + c * 512 + lane * 8) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  rw_u32x4 xr[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) xr[c] = *reinterpret_cast<const rw_u32x4*>(xs + c * 512 + lane * 8);
  __builtin_amdgcn_sched_barrier(0);
  auto finish = [&](wreg_t (&w)[R][NCH], int n) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if constexpr (F8) a = rw_dot8_fp8<T>(w[r][c], xr[c], a);
        else a = rw_dot8<T>(w[r][c], xr[c], a);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if constexpr (F8) a *= p.w_scale[row_of(n, r)];
      acc[r] = a;
    }
    if (lane == 0) {
      if constexpr (EPI == EPI_SWIGLU) {
        const float gt = rnd<T>(acc[0]), up = rnd<T>(acc[R - 1]);
        ((T*)p.Y)[n] = fromf<T>(rnd<T>(silu(gt)) * up);
      } else {
        const float y = acc[0] + (p.bias ? tof(((const T*)p.bias)[n]) : 0.f);
        if (p.out_f32) ((float*)p.Y)[n] = y;
        else ((T*)p.Y)[n] = fromf<T>(y);
      }
    }
  };
  // ---- 3. chunks of two outputs through three register buffers: while chunk j is computed, the first output of chunk j + 1 is in flight
  // (its id was requested before chunk j's loads were issued), and the ticket of chunk j + 2 goes out before chunk j + 1's loads
  while (cur >= 0) {
    const int nxt = resolve(nxt_t);               // usually long answered; blocking only when a pool ran dry
    if (nxt >= 0) nxt_t = ticket();
    if (out_ok(nxt, 0)) load_w(wc, nxt * DYN_CH);
    finish(wa, cur * DYN_CH);
    if (out_ok(nxt, 1)) load_w(wa, nxt * DYN_CH + 1);
    if (out_ok(cur, 1)) finish(wb, cur * DYN_CH + 1);
    cur = nxt;
    if (cur < 0) break;
    // the roles of the buffers rotate: (wa, wb, wc) held (cur.0, cur.1, nxt.0); now cur.0 sits in wc and cur.1 in wa
    const int nx2 = resolve(nxt_t);
    if (nx2 >= 0) nxt_t = ticket();
    if (out_ok(nx2, 0)) load_w(wb, nx2 * DYN_CH);
    finish(wc, cur * DYN_CH);
    if (out_ok(nx2, 1)) load_w(wc, nx2 * DYN_CH + 1);
    if (out_ok(cur, 1)) finish(wa, cur * DYN_CH + 1);
    cur = nx2;
    if (cur < 0) break;
    // now cur.0 sits in wb and cur.1 in wc
    const int nx3 = resolve(nxt_t);
    if (nx3 >= 0) nxt_t = ticket();
    if (out_ok(nx3, 0)) load_w(wa, nx3 * DYN_CH);
    finish(wb, cur * DYN_CH);
    if (out_ok(nx3, 1)) load_w(wb, nx3 * DYN_CH + 1);
    if (out_ok(cur, 1)) finish(wc, cur * DYN_CH + 1);
    cur = nx3;                                    // cur.0 in wa, cur.1 in wb again
  }
  // ---- 4. the last wave of the launch zeroes the counters for the next launch on this slot
  if (lane == 0) {
    const unsigned done = __hip_atomic_fetch_add(p.dyn + 64 * DYN_POOLS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x * 4u - 1u) {
      for (int i = 0; i <= DYN_POOLS; ++i) __hip_atomic_store(p.dyn + 64 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Whole-row GEMV for LONG K without split-K (round 3; batch 1: down_proj, K = 18944): y[n] = resid[n] + T(sum_k W[n][k] x[k]), in place.
// The split-K form leaves fp32 slices that a residual + RMSNorm launch must sum; here a workgroup stages x (37 KB) in LDS once, every
// wave streams whole rows in passes of 8 chunks of 512 (weights of pass p + 1 in flight under the dot products of pass p) and lane 0
// adds the residual: x + mlp is complete when the launch ends, and the NEXT projection normalises it in its own registers
// (gemv_rows_norm_kernel).  Rows are dealt to 2 workgroups per CU, row r of a workgroup to wave r % 4.
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool F8>
__global__ __launch_bounds__(512) void gemv_rows_longk_kernel(GemvP p, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char xs_raw[];
  T* xs = (T*)xs_raw;
  // one row per wave, rows_per_wg (<= 8) waves per workgroup: every wave of a workgroup streams the same number of bytes (7 rows per
  // workgroup at N = 3584 on 2 x 256 workgroups; two rows per wave on 4 waves left the fourth wave idle half the time: 26.0 us)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nch = (p.K + 511) >> 9, npass = (nch + 7) >> 3;
  const int n = blockIdx.x * rows_per_wg + wave;
  const bool valid = n < p.N;
  const int row = valid ? n : p.N - 1;
  OM_DBG_MIN(6, blockIdx.x < 8);
  typedef typename std::conditional<F8, rw_u32x2, rw_u32x4>::type wreg_t;
  auto load_w = [&](wreg_t (&w)[8], int pass) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int k = (pass * 8 + c) * 512 + lane * 8;
      k = k < p.K ? k : 0;                      // beyond K: any valid address (x is zero there)
      if constexpr (F8) w[c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x2*>((const unsigned char*)p.W + (size_t)row * p.ldw + k));
      else w[c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)p.W + (size_t)row * p.ldw + k));
    }
  };
  float acc = 0.f;
  auto dots = [&](wreg_t (&w)[8], int pass) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const rw_u32x4 xr = *reinterpret_cast<const rw_u32x4*>(xs + (pass * 8 + c) * 512 + lane * 8);
      if constexpr (F8) acc = rw_dot8_fp8<T>(w[c], xr, acc);
      else acc = rw_dot8<T>(w[c], xr, acc);
    }
  };
  wreg_t wa[8], wb[8], wc[8];
  load_w(wa, 0);
  if (npass > 1) load_w(wb, 1);
  if (npass > 2) load_w(wc, 2);      // all three buffers are on their way before x is staged (round 4: the third used to wait for the barrier below)
  // the epilogue's operands (bias, residual -- which may alias Y: only this wave writes element n --, the e4m3 row scale) right behind the first
  // weight passes: they come back with them instead of costing a dependent L2 round trip after the reduction (no branch around the loads)
  const float e_bias_raw = tof(((const T*)(p.bias ? p.bias : p.X))[p.bias ? row : 0]);
  const float e_res_raw = tof(((const T*)(p.resid ? p.resid : p.X))[p.resid ? row : 0]);
  const float e_bias = p.bias ? e_bias_raw : 0.f, e_res = p.resid ? e_res_raw : 0.f;
  float e_scale = 1.f;
  if constexpr (F8) e_scale = p.w_scale[row];
  // x -> LDS (zero beyond K up to the last whole pass), once per workgroup, under the first passes' weight loads
  for (int i = threadIdx.x; i < npass * 8 * 64; i += blockDim.x) {
    const rw_u32x4 z = {0u, 0u, 0u, 0u};
    const rw_u32x4 v = i * 8 < p.K ? *reinterpret_cast<const rw_u32x4*>((const T*)p.X + i * 8) : z;
    *reinterpret_cast<rw_u32x4*>(xs + i * 8) = v;
  }
  __syncthreads();
  // three register buffers: the weights of passes ps + 1 and ps + 2 are in flight under the dot products of pass ps (24 KB per wave)
  for (int ps = 0; ps < npass; ps += 3) {
    if (ps > 0 && ps + 2 < npass) load_w(wc, ps + 2);
    dots(wa, ps);
    if (ps + 1 < npass) {
      if (ps + 3 < npass) load_w(wa, ps + 3);
      dots(wb, ps + 1);
    }
    if (ps + 2 < npass) {
      if (ps + 4 < npass) load_w(wb, ps + 4);
      dots(wc, ps + 2);
    }
  }
  float a = acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if constexpr (F8) a *= e_scale;
  if (lane == 0 && valid) {
    const float y = rnd<T>(a + e_bias);
    ((T*)p.Y)[n] = fromf<T>(e_res + y);
  }
  OM_DBG_MAX(7, (blockIdx.x & 7) == 0);
}

#if OMCHAT_EXPERIMENTS
// The same rows WITHOUT the LDS stage (round 5, tuning key 39): a wave streams one row and uses every x element once, so the LDS copy only saves L2
// reads of x (37 KB, resident in every XCD's L2).  Here a wave loads its x passes from L2 next to its weight passes, two passes in flight:
// no barrier and no shared state, so the workgroups can be as small as one wave and the dispatcher re-deals them over the XCDs as they retire (what
// made the one-pair-per-wave gate|up launch faster).  Same chunk order and the same dot products as the LDS form: bit-identical.
// MEASURED SLOWER (same box, configs[1] decode ms per token): LDS form 2.605; this form as 1 / 2 / 4 waves per workgroup 2.700 / 2.707 / 2.712 -- the
// x registers halve the waves per SIMD (204 VGPRs) and 3584 rows no longer fit one resident round.  Experiments build only.
template <typename T, bool F8>
__global__ __launch_bounds__(256) void gemv_rows_longk_direct_kernel(GemvP p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nch = (p.K + 511) >> 9, npass = (nch + 7) >> 3;
  const int n = blockIdx.x * (int)(blockDim.x >> 6) + wave;
  if (n >= p.N) return;
  const int row = n;
  typedef typename std::conditional<F8, rw_u32x2, rw_u32x4>::type wreg_t;
  auto load_w = [&](wreg_t (&w)[8], int pass) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int k = (pass * 8 + c) * 512 + lane * 8;
      k = k < p.K ? k : 0;
      if constexpr (F8) w[c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x2*>((const unsigned char*)p.W + (size_t)row * p.ldw + k));
      else w[c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)p.W + (size_t)row * p.ldw + k));
    }
  };
  auto load_x = [&](rw_u32x4 (&x)[8], int pass) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int k = (pass * 8 + c) * 512 + lane * 8;
      const rw_u32x4 z = {0u, 0u, 0u, 0u};
      x[c] = k < p.K ? *reinterpret_cast<const rw_u32x4*>((const T*)p.X + k) : z;
    }
  };
  float acc = 0.f;
  auto dots = [&](wreg_t (&w)[8], rw_u32x4 (&x)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if constexpr (F8) acc = rw_dot8_fp8<T>(w[c], x[c], acc);
      else acc = rw_dot8<T>(w[c], x[c], acc);
    }
  };
  wreg_t wa[8], wb[8];
  rw_u32x4 xa[8], xb[8];
  load_w(wa, 0);
  load_x(xa, 0);
  if (npass > 1) { load_w(wb, 1); load_x(xb, 1); }
  const float e_bias_raw = tof(((const T*)(p.bias ? p.bias : p.X))[p.bias ? row : 0]);
  const float e_res_raw = tof(((const T*)(p.resid ? p.resid : p.X))[p.resid ? row : 0]);
  const float e_bias = p.bias ? e_bias_raw : 0.f, e_res = p.resid ? e_res_raw : 0.f;
  float e_scale = 1.f;
  if constexpr (F8) e_scale = p.w_scale[row];
  // two passes of weights and x in flight per wave (128 registers: three to four waves per SIMD); a buffer pair is refilled right after its dot products
  for (int ps = 0; ps < npass; ps += 2) {
    dots(wa, xa);
    if (ps + 2 < npass) { load_w(wa, ps + 2); load_x(xa, ps + 2); }
    if (ps + 1 < npass) {
      dots(wb, xb);
      if (ps + 3 < npass) { load_w(wb, ps + 3); load_x(xb, ps + 3); }
    }
  }
  float a = acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if constexpr (F8) a *= e_scale;
  if (lane == 0) {
    const float y = rnd<T>(a + e_bias);
    ((T*)p.Y)[n] = fromf<T>(e_res + y);
  }
}

#endif
int g_gemv_longk_direct = 0;   // omchat_op_set_tuning key 39 (experiments build): waves per workgroup (1, 2, 4) of the no-LDS long-K form; 0 = x through LDS (gemv_rows_longk_kernel)

template <typename T>
int launch_rows_longk(const GemvP& p, hipStream_t s) {
#if OMCHAT_EXPERIMENTS
  if (g_gemv_longk_direct > 0) {
    const int wpw = g_gemv_longk_direct > 4 ? 4 : g_gemv_longk_direct;
    if (p.w_scale) hipLaunchKernelGGL((gemv_rows_longk_direct_kernel<T, true>), dim3(cdiv(p.N, wpw)), dim3(64 * wpw), 0, s, p);
    else hipLaunchKernelGGL((gemv_rows_longk_direct_kernel<T, false>), dim3(cdiv(p.N, wpw)), dim3(64 * wpw), 0, s, p);
    return 0;
  }
#endif
  const int n_cu = device_cus();
  int rpw = cdiv(p.N, 2 * n_cu);                // two workgroups per CU
  rpw = rpw < 1 ? 1 : (rpw > 8 ? 8 : rpw);
  const int grid = cdiv(p.N, rpw);
  const int npass = (cdiv(p.K, 512) + 7) / 8;
  const size_t lds = (size_t)npass * 8 * 512 * 2;
  if (p.w_scale) {
    auto k = gemv_rows_longk_kernel<T, true>;
    static PerDeviceOnce set;
    if (set.first()) OM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * rpw), lds, s, p, rpw);
  } else {
    auto k = gemv_rows_longk_kernel<T, false>;
    static PerDeviceOnce set;
    if (set.first()) OM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * rpw), lds, s, p, rpw);
  }
  return 0;
}

int g_gemv_rows_balance = 1;   // omchat_op_set_tuning key 17: 1 = one-row-per-wave launches whose rows deal evenly to 2 workgroups per CU take N / (2 CUs) waves per workgroup (o_proj 7, qkv 9)
unsigned g_gemv_skew = 0;      // omchat_op_set_tuning key 28 (experiment): per-(blockIdx % 8) share deltas of the loop form, eight nibbles (d + 8)
int g_gemv_dyn = 0;            // omchat_op_set_tuning key 24: 1 = the loop form takes its outputs from atomic work counters when the caller provides them (gemv_rows_norm_dyn_kernel)
// omchat_op_set_tuning key 38: (gate, up) pairs per wave of the NON-loop norm form of the batch-1 gate|up GEMV.  Round 5: ONE pair per wave (4736
// workgroups of four waves for Qwen2-7B) beats the loop form's one resident round of 512 workgroups x 37 pairs: decode 2.650 -> 2.597 ms per token on
// one box (3 pairs per wave: 2.671, 2: 2.624).  The XCDs do not deliver the same HBM bandwidth when all stream at once, so a
// static equal split ends on the slowest XCD; thousands of 14 KB-per-wave workgroups are re-balanced by the dispatcher as they retire, the way the
// lm_head launch (9504 workgroups) always was -- it streams at 7.0 TB/s where the loop form reached 6.35.
int g_gemv_gu_rr = 1;
int g_gemv_gu_rr8 = 1;          // the same for the e4m3 replica (key 38, value x 16): 1.700 -> 1.678 -> 1.658 ms per token at 4 / 2 / 1 pairs per wave
int g_gemv_norm_loop = 0;      // omchat_op_set_tuning key 16: loop form (gemv_rows_norm_loop_kernel) of a batch-1 step's bit 0 = gate|up, 1 = qkv, 2 = e4m3 gate|up, 3 = lm_head

template <typename T, int EPI, int RR, bool F8, int NCH>
void launch_rows_norm_n(const GemvP& p, hipStream_t s) {
  const int n_out = EPI == EPI_SWIGLU ? p.N / 2 : p.N;
  if constexpr (NCH >= 4) {        // the decoder's hidden sizes only (K > 1536): keeps the instantiation count down
    // measured (profiles/r03_p): gate|up 44.3 -> 42.8 us (2.778 -> 2.738 ms per token); qkv 9.2 us either way; e4m3 gate|up 0.6 % slower
    const bool want = EPI == EPI_SWIGLU ? (g_gemv_norm_loop & (F8 ? 4 : 1)) : (p.N < 32768 ? (g_gemv_norm_loop & 2) : (g_gemv_norm_loop & 8));
    const int n_cu = device_cus();
    // dynamic form: the outputs dealt by atomic counters instead of equal shares (the XCDs do not stream at the same rate); tuning key 24
#if OMCHAT_EXPERIMENTS
    if (want && p.dyn && g_gemv_dyn && n_out >= 8 * n_cu) {
      hipLaunchKernelGGL((gemv_rows_norm_dyn_kernel<T, EPI, NCH, F8>), dim3(2 * n_cu), dim3(256), 0, s, p);
      return;
    }
#endif
    if (want && n_out >= 8 * n_cu) {
      const int per = cdiv(n_out, 2 * n_cu);
      const unsigned skew = (g_gemv_skew >= 256u && n_out == per * 2 * n_cu && (2 * n_cu) % 8 == 0) ? g_gemv_skew : 0u;
      hipLaunchKernelGGL((gemv_rows_norm_loop_kernel<T, EPI, NCH, F8>), dim3(cdiv(n_out, per)), dim3(256), 0, s, p, per, skew);
      return;
    }
  }
  if constexpr (NCH >= 4 && EPI == EPI_NONE && RR <= 2) {
    // qkv of a batch-1 step: 4608 rows = 9 x 512: one row per wave, nine waves per workgroup, 2 workgroups per CU (every CU streams 18 rows)
    // instead of 576 workgroups of 8 rows on 4 waves (2.25 per CU)
    const int n_cu = device_cus();
    if (g_gemv_rows_balance && n_out == 9 * 2 * n_cu) {
      hipLaunchKernelGGL((gemv_rows_norm_kernel<T, EPI, 1, NCH, F8, 9>), dim3(2 * n_cu), dim3(576), 0, s, p);
      return;
    }
  }
  // (four waves per workgroup: 8 / 16 waves -- fewer repeats of the norm -- measured 2.633 / 2.646 against 2.599 ms per token; down_proj's long-K form
  // with fewer rows per workgroup 2.615-2.765: each workgroup stages the 37 KB x for itself; lm_head with 2 / 1 rows per wave 2.591 against 2.597: noise)
#if OMCHAT_EXPERIMENTS
  if constexpr (EPI == EPI_SWIGLU && RR == 1) {
    const int nwg = cdiv(n_out, 4), xs = (int)(g_gemv_skew < 256u ? g_gemv_skew : 0u);
    if (xs > 0 && nwg % 8 == 0 && n_out % 4 == 0 && xs < nwg / 8) {
      GemvP q = p; q.xskew = xs;
      hipLaunchKernelGGL((gemv_rows_norm_kernel<T, EPI, RR, NCH, F8>), dim3(8 * (nwg / 8 + xs)), dim3(256), 0, s, q);
      return;
    }
  }
#endif
  hipLaunchKernelGGL((gemv_rows_norm_kernel<T, EPI, RR, NCH, F8>), dim3(cdiv(cdiv(n_out, RR), 4)), dim3(256), 0, s, p);
}
template <typename T, int EPI, int RR, bool F8>
void launch_rows_norm(const GemvP& p, hipStream_t s) {
  switch (cdiv(p.K, 512)) {
    case 1: launch_rows_norm_n<T, EPI, RR, F8, 1>(p, s); break;
    case 2: launch_rows_norm_n<T, EPI, RR, F8, 2>(p, s); break;
    case 3: launch_rows_norm_n<T, EPI, RR, F8, 3>(p, s); break;
    case 4: launch_rows_norm_n<T, EPI, RR, F8, 4>(p, s); break;
    case 5: launch_rows_norm_n<T, EPI, RR, F8, 5>(p, s); break;
    case 6: launch_rows_norm_n<T, EPI, RR, F8, 6>(p, s); break;
    case 7: launch_rows_norm_n<T, EPI, RR, F8, 7>(p, s); break;
    default: launch_rows_norm_n<T, EPI, RR, F8, 8>(p, s); break;
  }
}

template <typename T, int EPI, int RR, bool F8>
void launch_rows_r(const GemvP& p, hipStream_t s) {
  const int n_out = EPI == EPI_SWIGLU ? p.N / 2 : p.N;
  if constexpr (RR == 1 && EPI == EPI_RESID) {
    // o_proj of a batch-1 step: 3584 rows as 896 four-wave workgroups are 3.5 per CU (one CU in two streams a fourth workgroup); as
    // 512 seven-wave workgroups every CU streams the same 14 rows (profiles/r03_r)
    const int n_cu = device_cus();
    if (g_gemv_rows_balance && p.ksplit == 1 && n_out == 7 * 2 * n_cu) {
      hipLaunchKernelGGL((gemv_rows_kernel<T, EPI, 1, 7, F8>), dim3(2 * n_cu, 1), dim3(448), 0, s, p);
      return;
    }
  }
  int grid = cdiv(cdiv(n_out, RR), 4);
  grid = grid > 2048 ? 2048 : grid;
  hipLaunchKernelGGL((gemv_rows_kernel<T, EPI, RR, 4, F8>), dim3(grid, p.ksplit), dim3(256), 0, s, p);
}
// rows per wave-group from tools/experiments/tune_rows.hip (MI355X, r01): short outputs (fused qkv) are latency-bound and want the most
// waves (R = 1: 7.8 vs 8.8 us), everything else is flat in R; 4 waves per workgroup beat 8.  fp8 rows are half as many
// bytes: the wide lm_head takes 8 rows per group, the split-K shapes measured flat (o_proj slightly worse) and keep 4.
template <typename T, int EPI>
void launch_rows(const GemvP& p, hipStream_t s) {
  if (p.w_scale) {
    if constexpr (EPI == EPI_SWIGLU || EPI == EPI_NONE) {
      if (p.norm_w) {
        if (EPI == EPI_NONE && p.N < 32768) launch_rows_norm<T, EPI, 2, true>(p, s);
        else if (EPI == EPI_SWIGLU && g_gemv_gu_rr8 == 2) launch_rows_norm<T, EPI, 2, true>(p, s);
        else if (EPI == EPI_SWIGLU && g_gemv_gu_rr8 == 1) launch_rows_norm<T, EPI, 1, true>(p, s);
        else launch_rows_norm<T, EPI, 4, true>(p, s);
        return;
      }
    }
    if ((EPI == EPI_NONE || EPI == EPI_RESID) && p.N < 32768) launch_rows_r<T, EPI, 1, true>(p, s);
    else if (EPI == EPI_NONE) launch_rows_r<T, EPI, 8, true>(p, s);      // lm_head: 80.7 vs 82.3 us
    else launch_rows_r<T, EPI, 4, true>(p, s);
    return;
  }
  if constexpr (EPI == EPI_SWIGLU || EPI == EPI_NONE) {
    if (p.norm_w) {          // the norm shared through LDS; 3 (gate, up) pairs per wave: 6 x 7 chunks of weights + the row fit 256 VGPRs
      if (EPI == EPI_NONE && p.N < 32768) launch_rows_norm<T, EPI, 2, false>(p, s);      // qkv: 8 rows (56 KB) per workgroup against 14 KB of x + norm weights
      else if (EPI == EPI_SWIGLU && g_gemv_gu_rr == 2) launch_rows_norm<T, EPI, 2, false>(p, s);
      else if (EPI == EPI_SWIGLU && g_gemv_gu_rr == 3) launch_rows_norm<T, EPI, 3, false>(p, s);
      else if (EPI == EPI_SWIGLU) launch_rows_norm<T, EPI, 1, false>(p, s);
      else launch_rows_norm<T, EPI, 4, false>(p, s);
      return;
    }
  }
  if ((EPI == EPI_NONE || EPI == EPI_RESID) && p.N < 32768) launch_rows_r<T, EPI, 1, false>(p, s);      // short outputs: one row per wave (latency-bound)
  // a tensor-parallel rank's gate|up shard (2368 pairs at TP = 8): 4 pairs per wave are 148 workgroups on 256 CUs; one pair per wave
  // gives every CU ~9 waves (round 5)
  else if (EPI == EPI_SWIGLU && (g_gemv_shard & 4) && p.N / 2 < 16 * device_cus()) launch_rows_r<T, EPI, 1, false>(p, s);
  else launch_rows_r<T, EPI, 4, false>(p, s);
}

// Launch shapes measured with tools/experiments/tune_gemv.hip on MI355X (r01): plain loads beat non-temporal ones for this access
// shape (gate|up 54.9 -> 49.4 us), split-K slices prefer 4 waves x 8 chunks in flight when a slice is short.
template <typename T>
int launch_t(const GemvArgs& a, hipStream_t s) {
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  GemvP p{a.X, a.W, a.Y, a.bias, a.resid, a.ldx, a.ldw, a.ldy, a.ldr, a.b, a.N, a.K, a.out_f32, ks, a.w_scale, a.y_packed, a.norm_w, a.norm_eps,
          (unsigned*)a.dyn_ctr, a.y_pack, a.dbg};
  if (a.norm_w && a.x_packed && !(a.w_packed && ks == 1 && (a.K >> 6) == 56 && a.epi == EPI_SWIGLU && a.N % 32 == 0 && !g_gemv_no_xs)) {
    omchat_set_error("launch_gemv: the in-register RMSNorm on packed x exists for the x-stationary gate|up form only (packed W, K = 3584, no split-K)");
    return 1;
  }
  if (a.epi == EPI_RESID && a.x_packed && !(a.w_packed && ks == 1 && (a.K >> 6) == 56 && a.N % 16 == 0 && a.N / 16 <= device_cus() && !g_gemv_no_xs)) {
    omchat_set_error("launch_gemv: the batched residual epilogue exists for the x-stationary form only (packed W, K = 3584, N / 16 <= CUs, no split-K)");
    return 1;
  }
  if (a.x_packed) {
    // launch shapes from tools/experiments/tune_gemv32.hip (MI355X): <NTILE, WAVES, UNROLL> per shape class
#define OM_PK(NT_, EPI_, WV_, UN_)                                                                                                  \
  do {                                                                                                                              \
    const dim3 grid(cdiv(cdiv(a.N, 16), NT_), ks);                                                                                  \
    if (a.b > 16) { if (a.w_packed) hipLaunchKernelGGL((gemv_pk_kernel<T, NT_, EPI_, WV_, UN_, 2, true>), grid, dim3(WV_ * 64), 0, s, p);   \
                    else hipLaunchKernelGGL((gemv_pk_kernel<T, NT_, EPI_, WV_, UN_, 2, false>), grid, dim3(WV_ * 64), 0, s, p); }            \
    else { if (a.w_packed) hipLaunchKernelGGL((gemv_pk_kernel<T, NT_, EPI_, WV_, UN_, 1, true>), grid, dim3(WV_ * 64), 0, s, p);            \
           else hipLaunchKernelGGL((gemv_pk_kernel<T, NT_, EPI_, WV_, UN_, 1, false>), grid, dim3(WV_ * 64), 0, s, p); }                     \
  } while (0)
    // long launches with K = 64 * 8 * 7 (= 3584: gate|up, lm_head): x-stationary persistent form, one workgroup per CU, >= 4 units each
    const int n_cu = device_cus();
    // (units = what one workgroup walks: (gate, up) tile pairs or single tiles; fewer than 4 per workgroup leaves the last round too empty)
#if !OMCHAT_EXPERIMENTS
    // the seven-launch batched layer (un-split o_proj + residual, RMSNorm in the gate|up GEMV's registers; round 5) measured SLOWER than the eight
    // launches (configs[2] decode 4.57-4.87 vs 4.28 ms per step: the norm delays the weight stream by ~3 us in every workgroup and the NB = 2
    // form spills 35-120 VGPRs next to the 168 registers that hold x and two weight tiles): compiled with -DOMCHAT_EXPERIMENTS=1 only
    if (a.norm_w || a.epi == EPI_RESID) { omchat_set_error("launch_gemv: the batched norm-in-GEMV / residual forms need a -DOMCHAT_EXPERIMENTS=1 build"); return 1; }
#else
    if (a.w_packed && ks == 1 && !g_gemv_no_xs && (a.K >> 6) == 56 && a.epi == EPI_RESID) {      // un-split o_proj (+ residual): one tile per workgroup
      const dim3 grid(a.N / 16);
      if (a.b > 16) hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_RESID, 2, 7>), grid, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_RESID, 1, 7>), grid, dim3(512), 0, s, p);
      OM_LAUNCH_CHECK();
      return 0;
    }
#endif
    if (a.w_packed && ks == 1 && !g_gemv_no_xs && (a.K >> 6) == 56 && a.N % 32 == 0 && (a.epi == EPI_SWIGLU || a.epi == EPI_NONE) &&
        ((a.epi == EPI_SWIGLU ? a.N / 32 : a.N / 16) >= 4 * n_cu || (a.epi == EPI_NONE && a.N / 16 >= n_cu && a.N / 16 <= 2 * n_cu) ||
         ((g_gemv_shard & 1) && a.epi == EPI_NONE && a.N / 16 < n_cu) || ((g_gemv_shard & 8) && a.epi == EPI_SWIGLU && a.N / 32 <= n_cu))) {
      // the fused qkv projection (288 tiles on 256 CUs) cannot give every CU 4 units: it takes HALF as many workgroups as tiles, two tiles
      // each -- x is read 144 times instead of 288 and the second tile's loads run under the first's reduction (13.3 -> 11.7 us at b = 32)
      // a tensor-parallel rank's qkv shard (768 rows = 48 tiles at TP = 8) is shorter than one tile per CU: one tile per workgroup, 48
      // workgroups that each read x once, instead of gemv_pk_kernel's 24 workgroups of two tiles (round 5: 11.4 us -> see DESIGN.md section 6)
      const dim3 grid((a.epi == EPI_SWIGLU ? a.N / 32 : a.N / 16) >= 4 * n_cu ? n_cu : (a.epi == EPI_SWIGLU ? a.N / 32 : a.N / 16 < n_cu ? a.N / 16 : a.N / 32));
#if OMCHAT_EXPERIMENTS
      if (a.epi == EPI_SWIGLU && a.norm_w) {
        if (a.b > 16) hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_SWIGLU, 2, 7, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_SWIGLU, 1, 7, true>), grid, dim3(512), 0, s, p);
      } else
#endif
      if (a.epi == EPI_SWIGLU) {
        if (a.b > 16) hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_SWIGLU, 2, 7>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_SWIGLU, 1, 7>), grid, dim3(512), 0, s, p);
      } else {
        if (a.b > 16) hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_NONE, 2, 7>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemv_xs_kernel<T, EPI_NONE, 1, 7>), grid, dim3(512), 0, s, p);
      }
      OM_LAUNCH_CHECK();
      return 0;
    }
    // split-K launch whose chunks cut into ks slices of 40 and 16 (down_proj at TP = 1: 296 = 7 x 40 + 16, ks = 8): same form, one slice per blockIdx.y
    if (a.w_packed && !g_gemv_no_xs && a.epi == EPI_PARTIAL && ks > 1 && (a.K >> 6) % 8 == 0 && a.N % 16 == 0) {
      const int q = (a.K >> 6) / 8;                 // chunks per wave over the whole K
      // q = 5 * a5 + 2 * a2 with a5 + a2 == ks
      int a5 = -1;
      for (int t = 0; t <= ks; ++t) if (5 * t + 2 * (ks - t) == q) a5 = t;
      const int n_cu2 = device_cus();
      int gx = std::max(1, n_cu2 / ks);
      const int tiles = a.N / 16;
      // short launches (o_proj: 224 tiles x 2 slices): two tiles per workgroup instead of >= 4 units on every CU (as for qkv)
      if (tiles < 4 * gx && tiles % 2 == 0 && (long)tiles * ks >= n_cu2 && (long)tiles * ks <= 2L * n_cu2) gx = tiles / 2;
      if (a5 >= 0 && (tiles >= 4 * gx || gx == tiles / 2)) {
        const dim3 grid(gx, ks);
        if (a.b > 16) hipLaunchKernelGGL((gemv_xs_split_kernel<T, 2>), grid, dim3(512), 0, s, p, a5);
        else hipLaunchKernelGGL((gemv_xs_split_kernel<T, 1>), grid, dim3(512), 0, s, p, a5);
        OM_LAUNCH_CHECK();
        return 0;
      }
    }
    if (a.norm_w) { omchat_set_error("launch_gemv: packed-x RMSNorm form: gate|up shape outside the x-stationary launch (N / 32 >= 4 CUs)"); return 1; }
    if (a.epi == EPI_SWIGLU) OM_PK(2, EPI_SWIGLU, 8, 4);
    else if (a.epi == EPI_PARTIAL) {
      // short K slices (a tensor-parallel rank's o_proj: K = 512 = 8 chunks; down_proj: 37 chunks in two slices): one chunk per wave per
      // step so that all eight waves of a workgroup load, and one tile per workgroup so that the grid covers the CUs (round 5)
      const int cps = cdiv(a.K >> 6, ks);
      if ((g_gemv_shard & 2) && cps <= 8) OM_PK(1, EPI_PARTIAL, 8, 1);
      else if ((g_gemv_shard & 2) && cps <= 32) OM_PK(1, EPI_PARTIAL, 8, 2);
      else if (a.K / ks >= 1536) OM_PK(4, EPI_PARTIAL, 4, 2);
      else OM_PK(2, EPI_PARTIAL, 8, 4);
    }
    else if (a.N >= 32768) OM_PK(4, EPI_NONE, 4, 4);
    else OM_PK(2, EPI_NONE, 8, 4);
#undef OM_PK
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (a.b == 1 && !a.x_packed && a.epi == EPI_RESID && ks == 1 && !a.out_f32 && a.K > RW_MAXC * 512 && a.K % 8 == 0 && a.K <= 32768 &&
      !a.force_mfma && !g_gemv_force_mfma) {
    const int rc = launch_rows_longk<T>(p, s);      // long K in one piece: x through LDS, the result complete when the launch ends
    if (rc) return rc;
    OM_LAUNCH_CHECK();
    return 0;
  }
  const bool rows_ok = a.b == 1 && cdiv(cdiv(a.K, 512), ks) <= RW_MAXC;
  if (a.norm_w && !(rows_ok && ks == 1 && !a.force_mfma && !g_gemv_force_mfma && (a.epi == EPI_NONE || a.epi == EPI_SWIGLU))) {
    omchat_set_error("launch_gemv: the in-register RMSNorm needs the whole-row batch-1 form without split-K (K <= 4096), epilogue NONE / SWIGLU");
    return 1;
  }
  if (a.w_scale && !rows_ok) { omchat_set_error("launch_gemv: fp8 weights need b == 1 and <= 8 chunks of 512 per K slice"); return 1; }
  if (rows_ok && (a.w_scale || (!a.force_mfma && !g_gemv_force_mfma))) {       // whole-row streaming form
    switch (a.epi) {
      case EPI_PARTIAL: launch_rows<T, EPI_PARTIAL>(p, s); break;
      case EPI_SWIGLU: launch_rows<T, EPI_SWIGLU>(p, s); break;
      case EPI_RESID: launch_rows<T, EPI_RESID>(p, s); break;
      default: launch_rows<T, EPI_NONE>(p, s); break;
    }
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (a.b > 16) {       // two batch tiles per weight fragment: weights still stream once for b <= 32
    if (a.epi == EPI_PARTIAL) hipLaunchKernelGGL((gemv_kernel<T, 1, EPI_PARTIAL, 8, 4, false, 2>), dim3(cdiv(a.N, 16), ks), dim3(512), 0, s, p);
    else if (a.epi == EPI_SWIGLU) hipLaunchKernelGGL((gemv_kernel<T, 2, EPI_SWIGLU, 8, 4, false, 2>), dim3(cdiv(a.N, 32)), dim3(512), 0, s, p);
    else if (a.epi == EPI_RESID) hipLaunchKernelGGL((gemv_kernel<T, 1, EPI_RESID, 8, 4, false, 2>), dim3(cdiv(a.N, 16)), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((gemv_kernel<T, 2, EPI_NONE, 8, 4, false, 2>), dim3(cdiv(a.N, 32)), dim3(512), 0, s, p);
    OM_LAUNCH_CHECK();
    return 0;
  }
  if (a.epi == EPI_PARTIAL) {
    if (a.K / 64 / ks <= 32)
      hipLaunchKernelGGL((gemv_kernel<T, 1, EPI_PARTIAL, 4, 8>), dim3(cdiv(a.N, 16), ks), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((gemv_kernel<T, 1, EPI_PARTIAL, 8, 4>), dim3(cdiv(a.N, 16), ks), dim3(512), 0, s, p);
  } else if (a.epi == EPI_SWIGLU) {
    hipLaunchKernelGGL((gemv_kernel<T, 2, EPI_SWIGLU, 8, 4>), dim3(cdiv(a.N, 32)), dim3(512), 0, s, p);
  } else if (a.epi == EPI_RESID) {
    hipLaunchKernelGGL((gemv_kernel<T, 1, EPI_RESID, 8, 4>), dim3(cdiv(a.N, 16)), dim3(512), 0, s, p);
  } else {
    // wide outputs (lm_head): 2 tiles per workgroup halves the x re-reads
    if (a.N >= 16384)
      hipLaunchKernelGGL((gemv_kernel<T, 2, EPI_NONE, 8, 8>), dim3(cdiv(a.N, 32)), dim3(512), 0, s, p);
    else
      hipLaunchKernelGGL((gemv_kernel<T, 1, EPI_NONE, 8, 8>), dim3(cdiv(a.N, 16)), dim3(512), 0, s, p);
  }
  OM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

void gemv_set_force_mfma(int v) { g_gemv_force_mfma = v; }
int gemv_get_force_mfma() { return g_gemv_force_mfma; }
void gemv_set_no_xs(int v) { g_gemv_no_xs = v; }
void gemv_set_shard_shapes(int v) { g_gemv_shard = v; }
int gemv_get_shard_shapes() { return g_gemv_shard; }
void gemv_set_norm_loop(int v) { g_gemv_norm_loop = v; }
void gemv_set_gu_rr(int v) { if (v >= 16) g_gemv_gu_rr8 = v / 16; else g_gemv_gu_rr = v; }
void gemv_set_dyn(int v) { g_gemv_dyn = v; }
void gemv_set_longk_direct(int v) { g_gemv_longk_direct = v; }
void gemv_set_skew(int v) { g_gemv_skew = (unsigned)v; }
void gemv_set_rows_balance(int v) { g_gemv_rows_balance = v; }

namespace {
// one wave per row: absmax, then e4m3 (round to nearest even) of w / scale, 8 weights per lane per step
template <typename T>
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const T* __restrict__ W, int ldw, int N, int K, unsigned char* __restrict__ W8, int ldq,
                                                             float* __restrict__ scale) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const T* w = W + (size_t)n * ldw;
  float m = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const typename V8<T>::type v = ld8<T>(w + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(tof(v[j])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const float sc = m > 0.f ? m / 448.0f : 1.0f;
  if (lane == 0) scale[n] = sc;
  for (int k = lane * 8; k < K; k += 512) {
    const typename V8<T>::type v = ld8<T>(w + k);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = tof(v[j]) / sc;
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    rw_u32x2 o2 = {(unsigned)lo, (unsigned)hi};
    *reinterpret_cast<rw_u32x2*>(W8 + (size_t)n * ldq + k) = o2;
  }
}
}  // namespace

int launch_quant_fp8_rows(int dtype, const void* W, int ldw, int N, int K, void* W8, int ld8, float* scale, hipStream_t s) {
  OM_CHECK(K % 8 == 0 && ldw % 8 == 0 && ld8 % 8 == 0, "K, ldw, ld8 must be multiples of 8");
  if (dtype == OMCHAT_F16) hipLaunchKernelGGL(quant_fp8_rows_kernel<f16>, dim3(cdiv(N, 4)), dim3(256), 0, s, (const f16*)W, ldw, N, K, (unsigned char*)W8, ld8, scale);
  else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL(quant_fp8_rows_kernel<bf16>, dim3(cdiv(N, 4)), dim3(256), 0, s, (const bf16*)W, ldw, N, K, (unsigned char*)W8, ld8, scale);
  else { omchat_set_error("launch_quant_fp8_rows: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_pack_x(int dtype, const void* X, int ldx, int b, int K, void* out, hipStream_t s) {
  OM_CHECK(b >= 1 && b <= 32 && K % 64 == 0 && ldx % 8 == 0, "pack_x: 1 <= b <= 32, K % 64, ldx % 8");
  const int NB = b > 16 ? 2 : 1;
  const int grid = cdiv(NB * 16 * (K / 8), 256);
  if (dtype == OMCHAT_F16) hipLaunchKernelGGL(pack_x_kernel<f16>, dim3(grid), dim3(256), 0, s, (const f16*)X, ldx, b, K, (f16*)out, NB);
  else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL(pack_x_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)X, ldx, b, K, (bf16*)out, NB);
  else { omchat_set_error("launch_pack_x: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}

int launch_pack_w(int dtype, const void* W, int ldw, int N, int K, void* out, hipStream_t s) {
  OM_CHECK(N % 16 == 0 && K % 64 == 0 && ldw % 8 == 0, "pack_w: N % 16, K % 64, ldw % 8");
  const long n8 = (long)N * (K / 8);
  const int grid = (int)(cdiv64(n8, 256) > 8192 ? 8192 : cdiv64(n8, 256));
  if (dtype == OMCHAT_F16) hipLaunchKernelGGL(pack_w_kernel<f16>, dim3(grid), dim3(256), 0, s, (const f16*)W, ldw, N, K, (f16*)out);
  else if (dtype == OMCHAT_BF16) hipLaunchKernelGGL(pack_w_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)W, ldw, N, K, (bf16*)out);
  else { omchat_set_error("launch_pack_w: bad dtype"); return 1; }
  OM_LAUNCH_CHECK();
  return 0;
}

#if OMCHAT_EXPERIMENTS
// PROTOTYPE (round 5, tuning key 42): the batch-1 o_proj GEMV launched OUT OF ORDER behind the split-KV merge (hipExtAnyOrderLaunch: its workgroups are
// placed as soon as the merge's are) -- every wave issues its weight row first, wave 0 then polls the merge's per-head completion flags (epoch values,
// written after an agent-scope release by each merge workgroup), the workgroup acquires and only then loads the attention row.  One row per wave, K = NCH x 512
// in one piece, residual in the epilogue: the same chunk order and dot products as gemv_rows_kernel<EPI_RESID> -- the same bits.
template <typename T, int NCH>
__global__ __launch_bounds__(448) void gemv_rows_wait_kernel(GemvP p, const unsigned* flags, unsigned epoch, int nflags, unsigned* err, int mode) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x * (int)(blockDim.x >> 6) + wave;
  const bool valid = n < p.N;
  const int row = valid ? n : p.N - 1;
  OM_DBG_MIN(1, blockIdx.x < 8);
  rw_u32x4 w[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) w[c] = __builtin_nontemporal_load(reinterpret_cast<const rw_u32x4*>((const T*)p.W + (size_t)row * p.ldw + c * 512 + lane * 8));
  const float e_res = tof(((const T*)p.resid)[row]);
  __builtin_amdgcn_sched_barrier(0);
  if (wave == 0) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
      const unsigned f = lane < nflags ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
      if (__all(f == epoch)) break;
      if (mode & 8) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(1);      // mode bit 3: ~1 us between polls instead of ~30 ns
      if ((++spins & 255u) == 0u && wall_clock64() - t0 > 200000000ull) { if (lane == 0) atomicOr(err, 4u); break; }      // 2 s of the 100 MHz clock
    }
  }
  OM_DBG_MAX(2, blockIdx.x < 8);
  __syncthreads();
  rw_u32x4 xr[NCH];
  if (mode & 2) {
    // mode bit 1: the attention row by cache-bypassing loads (sc0 sc1) instead of an agent-scope acquire (buffer_inv sc1 in 3584 waves) in front of plain loads
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, NCH * 1024, 0x00020000);
#pragma unroll
    for (int c = 0; c < NCH; ++c) xr[c] = __builtin_bit_cast(rw_u32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, c * 1024 + lane * 16, 0, 17));
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int c = 0; c < NCH; ++c) xr[c] = *reinterpret_cast<const rw_u32x4*>((const T*)p.X + c * 512 + lane * 8);
  }
  float a = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) a = rw_dot8<T>(w[c], xr[c], a);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0 && valid) ((T*)p.Y)[n] = fromf<T>(e_res + rnd<T>(a));
  OM_DBG_MAX(3, (blockIdx.x & 7) == 0);
}

template <typename T>
static int launch_gemv_wait_t(const GemvArgs& a, const unsigned* flags, unsigned epoch, int nflags, unsigned* err, int mode, hipStream_t s) {
  GemvP p{a.X, a.W, a.Y, a.bias, a.resid, a.ldx, a.ldw, a.ldy, a.ldr, a.b, a.N, a.K, 0, 1, nullptr, 0, nullptr, 0.f, nullptr, nullptr, a.dbg};
  int wpw = cdiv(a.N, 2 * device_cus());
  wpw = wpw < 1 ? 1 : (wpw > 7 ? 7 : wpw);
  hipExtLaunchKernelGGL((gemv_rows_wait_kernel<T, 7>), dim3(cdiv(a.N, wpw)), dim3(64 * wpw), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, p, flags, epoch, nflags, err, mode);
  OM_LAUNCH_CHECK();
  return 0;
}
#endif

// experiments build: see gemv_rows_wait_kernel.  Requirements: b = 1, K = 3584, 16-bit weights, EPI_RESID without bias, nflags <= 64.
int launch_gemv_wait(int dtype, const GemvArgs& a, const unsigned* flags, unsigned epoch, int nflags, unsigned* err, int mode, hipStream_t s) {
#if OMCHAT_EXPERIMENTS
  OM_CHECK(a.b == 1 && a.K == 3584 && !a.w_scale && a.epi == EPI_RESID && !a.bias && a.resid && !a.x_packed && nflags >= 1 && nflags <= 64 && flags && err,
           "launch_gemv_wait: batch 1, K = 3584, 16-bit weights, residual epilogue without bias");
  if (dtype == OMCHAT_F16) return launch_gemv_wait_t<f16>(a, flags, epoch, nflags, err, mode, s);
  if (dtype == OMCHAT_BF16) return launch_gemv_wait_t<bf16>(a, flags, epoch, nflags, err, mode, s);
  omchat_set_error("bad dtype");
  return 1;
#else
  (void)dtype; (void)a; (void)flags; (void)epoch; (void)nflags; (void)err; (void)mode; (void)s;
  omchat_set_error("launch_gemv_wait: -DOMCHAT_EXPERIMENTS=1 builds only");
  return 1;
#endif
}

int launch_gemv(int dtype, const GemvArgs& a, hipStream_t s) {
  OM_CHECK(a.b >= 1 && a.b <= 32, "batch must be 1..32 per call");
  OM_CHECK(!a.x_packed || !a.w_scale, "packed x: 16-bit weights");      // (EPI_RESID: x-stationary form only, checked in launch_t)
  OM_CHECK(!a.w_packed || (a.x_packed && a.N % 16 == 0), "packed W needs packed x and N % 16 == 0");
  OM_CHECK(!a.y_packed || (a.x_packed && a.epi == EPI_SWIGLU), "packed y: SwiGLU epilogue of the packed kernel only");
  OM_CHECK(a.K % 64 == 0 && a.ldw % 8 == 0 && a.ldx % 8 == 0, "K % 64, ldw % 8, ldx % 8");
  OM_CHECK(!a.w_scale || a.b == 1, "fp8 weights: batch 1 only");
  OM_CHECK(a.epi == EPI_NONE || a.epi == EPI_RESID || a.epi == EPI_SWIGLU || a.epi == EPI_PARTIAL, "bad epilogue");
  OM_CHECK(a.ksplit <= 1 || a.epi == EPI_PARTIAL, "ksplit > 1 only with EPI_PARTIAL (fp32 slices)");
  OM_CHECK(a.ksplit <= a.K / 64, "ksplit exceeds the number of 64-wide K chunks");
  OM_CHECK(a.epi != EPI_RESID || a.resid, "resid missing");
  OM_CHECK(a.epi != EPI_SWIGLU || a.N % 32 == 0, "SwiGLU needs N % 32 == 0");
  OM_CHECK(!(a.out_f32 && a.epi != EPI_NONE), "fp32 output only with EPI_NONE");
  if (dtype == OMCHAT_F16) return launch_t<f16>(a, s);
  if (dtype == OMCHAT_BF16) return launch_t<bf16>(a, s);
  omchat_set_error("launch_gemv: bad dtype");
  return 1;
}
