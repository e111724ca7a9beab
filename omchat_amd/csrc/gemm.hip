// MFMA GEMM with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] @ W[N,K]^T), fp32 accumulate.
//
// Replaces every ATen/cuBLAS `nn.Linear` on the hot path (modeling_intern_vit.py:124,136,184-185;
// multimodal_projector/builder.py:57-61; transformers Qwen2 q/k/v/o/gate/up/down) plus the elementwise ops the
// reference runs after them (bias, erf-GELU, layer-scale + residual, SiLU*up), with the reference's fp16
// rounding points kept.
//
// Structure: BK = 64, one 128-byte LDS row per tile row, two LDS stages.  Tiles are staged HBM->LDS with
// global_load_lds_dwordx4 (lane-linear destination, so the bank swizzle is applied to the per-lane SOURCE
// address and to the ds_read_b128 address: chunk' = chunk ^ ((row >> 1) & 7), conflict-free for the
// 16x16x32 operand read).  One barrier per K-step: the prefetch of tile t+1 is issued right after the barrier and
// lands under the 64 (256^2 tile) or 32 (128^2 tile) MFMAs per wave of tile t.  Grid is 1-D with a bijective
// XCD remap so that the workgroups sharing an XCD's L2 walk neighbouring tiles (M fastest).
#include "kernels.h"
#include <array>
#include <stdio.h>
#include <stdlib.h>
#include <limits.h>
#include <algorithm>
#include <map>
#include <mutex>

namespace {

int g_gemm_wide_store = 1;   // omchat_op_set_tuning key 37: 1 = 16-byte epilogue stores (two column blocks exchanged between lane pairs), 0 = 8-byte stores
int g_gemm_skip_dead = 1;    // omchat_op_set_tuning key 43: 1 = 64-row blocks / waves that lie wholly beyond M issue no operand reads and no MFMAs (same bits), 0 = full issue

struct GemmP {
  const void* A; const void* W; void* C; const void* bias; const void* ls; const void* resid;
  int lda, ldw, ldc, ldr, M, N, K;
  const float* a_scale; const float* w_scale;      // fp8 x fp8 kernel: per-row scales of A and W (null otherwise)
  int wide_store;                                  // 16-byte epilogue stores (gemm_epilogue)
  int skip_dead;                                   // ragged M: skip the MFMAs of row blocks beyond M (gemm8_segment, gemm_kernel)
  const float* row_scale;                          // [M] or null: the accumulators of row m are multiplied by row_scale[m] first (folded RMSNorm)
  float* stats; int stats_ld;                      // EPI_*_STATS: slot-major [slots][stats_ld >= M] partial sums of squares of the stored outputs, slot = colu / (NR * 16)
  // the row scale finished INSIDE the launch from the producer's statistics slots: rstd[m] = rsqrt(sum_{s < rs_nslots} rs_stats[m][s] * rs_inv_dim + rs_eps),
  // computed once per tile into LDS after the K loop (no finishing launch between the producing GEMM and this one); slot-major: rs_stats[s * rs_ld + m]
  const float* rs_stats; int rs_ld, rs_nslots; float rs_inv_dim, rs_eps;
  int rs_dma;                                      // 1: the tile's slots are fetched by LDS-DMA BEFORE the K loop into the LDS behind the stage buffers
};

// Workgroup -> tile.  (1) bijective XCD remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so
// ids b and b+8 share an L2; give each XCD a contiguous range of logical ids.  (2) inside that range walk the tile grid in
// groups of GROUP_M row tiles x all column tiles, row fastest: the 32 workgroups an XCD runs at once then form a 4 x 8
// block that shares 4 A row-panels and 8 W column-panels per K-step (12 panel slices instead of up to 33).
constexpr int GROUP_M = 4;
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}
__device__ __forceinline__ void tile_coords_id(int wgid, int M, int N, int BM, int BN, int& m0, int& n0) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int per_group = GROUP_M * tiles_n;
  const int g = wgid / per_group, in_g = wgid % per_group;
  const int rows_here = tiles_m - g * GROUP_M < GROUP_M ? tiles_m - g * GROUP_M : GROUP_M;
  m0 = (g * GROUP_M + in_g % rows_here) * BM;
  n0 = (in_g / rows_here) * BN;
}
__device__ __forceinline__ void tile_coords(int M, int N, int BM, int BN, int& m0, int& n0) {
  tile_coords_id(xcd_remap(blockIdx.x, gridDim.x), M, N, BM, BN, m0, n0);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------------------
// Shared epilogue.  Every kernel below issues its MFMAs with the operands SWAPPED (W fragment first): the two 16x16x32 operands have
// the same lane layout, so the accumulator then holds the transposed tile,
//     acc[i][j][r] = C[rowu + i*16 + fr][colu + j*16 + 4*fg + r]        (rowu / colu: wave-uniform corner of the wave tile)
// i.e. a lane owns FOUR CONSECUTIVE COLUMNS of one row per fragment: one 8-byte store (and one 8-byte residual load) per fragment
// instead of four 2-byte ones, bias / layer-scale as 8-byte loads per column block.  VALU does not overlap MFMA on a SIMD, so epilogue
// instructions are paid in full (round 2, row-major accumulators: ~25 VALU + a 2-byte store + a 2-byte load per ELEMENT made a 256^2
// tile's epilogue ~20 % of a K = 3200 main loop and the whole cost of the K = 512 row-parallel shards of TP = 8: 243 TF).
// C and the residual are addressed through buffer resources that start at the wave tile's first row and end after its last valid row:
// the hardware drops out-of-range rows.  Rounding points are unchanged.  N % 4 == 0.
// ---------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ unsigned short bits16(T v) { return __builtin_bit_cast(unsigned short, v); }
template <typename T> __device__ __forceinline__ float from_bits16(unsigned short v) { return tof(__builtin_bit_cast(T, v)); }
template <typename T> __device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  typename V8<T>::half_type h = {fromf<T>(a), fromf<T>(b), fromf<T>(c), fromf<T>(d)};
  return __builtin_bit_cast(u32x2, h);
}
template <typename T> __device__ __forceinline__ f32x4 unpack4(u32x2 w) {
  const typename V8<T>::half_type h = __builtin_bit_cast(typename V8<T>::half_type, w);
  return (f32x4){tof(h[0]), tof(h[1]), tof(h[2]), tof(h[3])};
}
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p) {
  return unpack4<T>(*reinterpret_cast<const u32x2*>(p));
}

template <typename T, int MR, int NR, int EPI_, bool F8 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[MR][NR], int rowu_, int colu_, int fg, int fr, const float* rs_lds = nullptr) {
  // No contraction in the epilogue (round 6): resid + T(T(acc + b) * ls) is an expression over 16-bit values, the compiler narrows it to 16-bit
  // fmul / fadd and -ffp-contract=fast then fuses the product into the add -- an fma that SKIPS the rounding of branch * ls (N8): 6.5 % of the
  // layer-scale + residual outputs were 1 ulp off the reference's rounding points (tools/dbg_r6.py; the statistics variant of the same epilogue,
  // whose extra use of the value blocked the narrowing, matched the fp32 restatement exactly).  GELU's polynomial uses explicit fma builtins.
#pragma clang fp contract(off)
  constexpr bool STATS = EPI_ == EPI_LS_RESID_STATS || EPI_ == EPI_NONE_STATS;
  constexpr int EPI = EPI_ == EPI_LS_RESID_STATS ? EPI_LS_RESID : (EPI_ == EPI_NONE_STATS ? EPI_NONE : EPI_);
  // One statistics slot per WAVE TILE (64 columns on the 256^2 kernels, 112 on the 192 x 224 ones; a two-launch form leaves the slots of its second
  // launch behind those of the first).  A row's statistic is the sum of its slots in slot order: deterministic for a given problem, but the grouping
  // follows the tile kernel tuned for the problem SIZE, so the same tile of pixels in another batch size may see its rstd differ in the last fp32 bit
  // (as the reference's own GEMM library does per shape).  Two forms that made the grouping size-independent were measured and dropped: the 192 x 224
  // family for every M (-1.6 % ViT at 3 tiles, +4 % at 24) and one slot per 16-column block (the 200-slot read per tile cost what the fusion saves).
  const int rowu = __builtin_amdgcn_readfirstlane(rowu_), colu = __builtin_amdgcn_readfirstlane(colu_);      // SGPRs: scalar offsets, scalar resources
  const int rows_valid = p.M - rowu < MR * 16 ? p.M - rowu : MR * 16;
  if (rows_valid <= 0) return;                                                   // wave-uniform
  if (p.row_scale || rs_lds) {        // folded RMSNorm: y = rstd[m] * (x W'^T); wave-uniform branch, one 4-byte load per row fragment
    float rs[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int row = rowu + i * 16 + fr;
      rs[i] = rs_lds ? rs_lds[i * 16 + fr] : p.row_scale[row < p.M ? row : p.M - 1];      // (rs_lds: this wave's rows, finished by tile_row_scale)
    }
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] *= rs[i];
  }
  float ssq[MR];            // STATS: this lane's part of sum_c out[row][c]^2 over the wave tile's valid columns
#pragma unroll
  for (int i = 0; i < MR; ++i) ssq[i] = 0.f;
  if constexpr (F8) {       // fp8 x fp8 operands: the accumulators are sums of unscaled e4m3 products
    float sa[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) { const int row = rowu + i * 16 + fr; sa[i] = p.a_scale[row < p.M ? row : p.M - 1]; }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int col = colu + j * 16 + 4 * fg;
      const f32x4 sw = col < p.N ? *reinterpret_cast<const f32x4*>(p.w_scale + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MR; ++i) acc[i][j] *= sw * sa[i];
    }
  }
  if constexpr (EPI == EPI_F32OUT) {      // raw fp32 accumulators, 16 bytes per fragment row (tensor-parallel partial sums)
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((float*)p.C + (size_t)rowu * p.ldc, 0, rows_valid * p.ldc * 4, 0x00020000);
    const int f_lane = (fr * p.ldc + 4 * fg) * 4;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (colu + j * 16 + 4 * fg < p.N) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), frs, f_lane + (colu + j * 16) * 4 + i * 16 * p.ldc * 4, 0, 0);
      }
    }
    return;
  }
  const T* __restrict__ bias = (const T*)p.bias;
  // byte ranges of the wave tile's rows (< 2^32: at most 128 rows of one matrix row stride each)
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((T*)p.C + (size_t)rowu * p.ldc, 0, rows_valid * p.ldc * 2, 0x00020000);
  const int c_lane = (fr * p.ldc + 4 * fg) * 2;                                  // lane part of the byte offset inside C's range
  if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
    for (int j = 0; j < NR; j += 2) {
      if (colu + j * 16 + 16 + 4 * fg < p.N) {                                   // gate columns in the fused layout (up = +16)
        const int oc = ((colu + j * 16) >> 1) * 2;                               // byte offset of the output column block
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = rnd<T>(acc[i][j][r]), u = rnd<T>(acc[i][j + 1][r]);
            o[r] = rnd<T>(silu(g)) * u;
          }
          __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(o[0], o[1], o[2], o[3]), crs, c_lane + i * 16 * p.ldc * 2 + oc, 0, 0);
        }
      }
    }
  } else {
    const T* __restrict__ ls = (const T*)p.ls;
    const bool has_r = (EPI == EPI_LS_RESID || EPI == EPI_RESID) && p.resid != nullptr;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(has_r ? (T*)p.resid + (size_t)rowu * p.ldr : (T*)p.C, 0,
                                                                        has_r ? rows_valid * p.ldr * 2 : 0, 0x00020000);
    const int r_lane = (fr * p.ldr + 4 * fg) * 2;
    // The loads of the epilogue are issued in large groups BEFORE their first use: bias / layer-scale and the MR residual fragments
    // (8 bytes each) of JG column blocks at a time.  Measured with tools/bench_gemm_k.py
    // (K = 64, i.e. the fixed cost of a launch): the per-column-block form -- load, wait, store, four times in a row -- cost a 256^2 tile
    // 5.6 us alone on the chip and 13.9 us per round with all 256 CUs in their epilogue together: four dependent memory round trips.
    // JG column blocks per group: all of them (224 VGPRs, no spill on the 256^2 tile); the fp8 kernels, whose scale vectors stay live, take
    // two groups
    constexpr int JG = (F8 && MR * NR > 16) ? (NR + 1) / 2 : NR;
#pragma unroll
    for (int j0 = 0; j0 < NR; j0 += JG) {
      f32x4 bv[JG], lsv[JG];
      u32x2 rb[MR][JG];
#pragma unroll
      for (int jj = 0; jj < JG; ++jj) {
        const int j = j0 + jj;
        if (j < NR) {
          const int col = colu + j * 16 + 4 * fg;
          const int cc = col < p.N ? col : 0;                                      // clamped: the value is unused when the block is out of range
          bv[jj] = bias ? load4<T>(bias + cc) : (f32x4){0.f, 0.f, 0.f, 0.f};
          lsv[jj] = (f32x4){1.f, 1.f, 1.f, 1.f};
          if constexpr (EPI == EPI_LS_RESID) lsv[jj] = load4<T>(ls + cc);
          if constexpr (EPI == EPI_LS_RESID || EPI == EPI_RESID) {
            const int rj = r_lane + (colu + j * 16) * 2;
            // (no residual: the resource has zero records and the loads return 0 -- no branch either way; columns beyond N of the last
            // tile read the next row's first bytes or nothing, and are never stored)
#pragma unroll
            for (int i = 0; i < MR; ++i) rb[i][jj] = __builtin_amdgcn_raw_buffer_load_b64(rrs, rj + i * 16 * p.ldr * 2, 0, 0);
          }
        }
      }
      // one output fragment (i, jj) of this lane: four consecutive columns, packed to 8 bytes
      auto out4 = [&](int i, int jj) {
        const int j = j0 + jj;
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_LS_RESID || EPI == EPI_RESID) rv = unpack4<T>(rb[i][jj]);
        float o[4];
        if constexpr (EPI == EPI_GELU) {
          const f32x2 g0 = gelu_erf2((f32x2){rnd<T>(acc[i][j][0] + bv[jj][0]), rnd<T>(acc[i][j][1] + bv[jj][1])});
          const f32x2 g1 = gelu_erf2((f32x2){rnd<T>(acc[i][j][2] + bv[jj][2]), rnd<T>(acc[i][j][3] + bv[jj][3])});
          o[0] = g0[0]; o[1] = g0[1]; o[2] = g1[0]; o[3] = g1[1];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][r] + bv[jj][r];
            if constexpr (EPI != EPI_NONE) v = rnd<T>(v);      // T(acc + b) feeds further fp32 math; alone, the store below is that rounding
            if constexpr (EPI == EPI_LS_RESID) v = rv[r] + rnd<T>(v * lsv[jj][r]);
            if constexpr (EPI == EPI_RESID) v = rv[r] + v;
            o[r] = v;
          }
        }
        if constexpr (STATS) {
          // the statistics are those of the STORED values (the reference's norm reads the 16-bit tensor); columns beyond N do not count
          if (colu + j * 16 + 4 * fg < p.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float f = rnd<T>(o[r]); ssq[i] = __builtin_fmaf(f, f, ssq[i]); }
          }
        }
        return pack4<T>(o[0], o[1], o[2], o[3]);
      };
      // Wide stores (round 5): the PMC passes read 1.6-1.7 x the output bytes in WRITE_SIZE for these epilogues (fc1 133 MB for 78.7 MB): a 128-byte
      // line of C was completed by FOUR 8-byte-per-lane store instructions MR stores apart, and L2 wrote partly filled lines back in between.  Two
      // neighbouring column blocks are exchanged between the lane pairs (fg, fg ^ 1) with v_permlane16_swap after packing -- the even lane keeps
      // block j and takes its partner's four columns of it, the odd lane gets block j + 1 -- so every lane stores 16 bytes = 8 consecutive columns and
      // a row's 64 bytes of the block pair leave in one instruction.  Needs N, ldc % 8 == 0 and a 16-byte aligned C (every hot-path shape).
      const bool wide = (JG % 2 == 0) && (NR % 2 == 0) && p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 && p.wide_store;
      if (wide) {
        // row fragment outer, block pairs inner: the two 64-byte halves of a 128-byte line of C leave in consecutive instructions
#pragma unroll
        for (int i = 0; i < MR; ++i) {
#pragma unroll
          for (int jj = 0; jj < JG; jj += 2) {
            const int j = j0 + jj;
            if (j + 1 < NR && colu + j * 16 < p.N) {              // (a started block pair may end inside either block: per-lane test below)
              // this lane's eight columns start at a multiple of 8: inside the matrix or wholly outside it (N % 8 == 0)
              const bool mine = colu + (j + (fg & 1)) * 16 + 4 * (fg & ~1) < p.N;
              const int cw = (fr * p.ldc + 4 * (fg & ~1)) * 2 + (colu + (j + (fg & 1)) * 16) * 2;
              const u32x2 a = out4(i, jj), b = out4(i, jj + 1);
              const auto s0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
              const auto s1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
              const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
              if (mine) __builtin_amdgcn_raw_buffer_store_b128(w, crs, cw + i * 16 * p.ldc * 2, 0, 0);
            }
          }
        }
      } else {
#pragma unroll
      for (int jj = 0; jj < JG; ++jj) {
        const int j = j0 + jj;
        const int col = colu + j * 16 + 4 * fg;
        if (j < NR && col < p.N) {
          const int cj = c_lane + (colu + j * 16) * 2;
#pragma unroll
          for (int i = 0; i < MR; ++i) __builtin_amdgcn_raw_buffer_store_b64(out4(i, jj), crs, cj + i * 16 * p.ldc * 2, 0, 0);
        }
      }
      }
    }
    if constexpr (STATS) {
      // the four lanes (fr, fg = 0..3) of a row meet by two xor steps (fixed order); lane fg == 0 writes the row's partial of this wave tile
      if (colu < p.N) {
        const int slot = colu / (NR * 16);
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          float v = ssq[i];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          const int row = rowu + i * 16 + fr;
          if (fg == 0 && row < p.M) p.stats[(size_t)slot * p.stats_ld + row] = v;      // slot-major: the 16 rows of a fragment are one 64-byte run
        }
      }
    }
  }
}

// rstd of the tile's BM rows from the producer's statistics slots, into LDS (the stage buffers are dead: every wave is behind its last LDS read).
// Two threads per row take the even / the odd slots -- all of a thread's loads are requested before the first add -- and meet by one lane swap:
// a fixed order.  Returns with a workgroup barrier behind the LDS writes.
// The slots of the tile's BM rows requested by LDS-DMA at the START of the kernel (no registers, nothing to wait for): they land in the LDS behind
// the stage buffers, Ls[slot * BM + r], while the K loop runs -- the K loop's first counted vmcnt wait retires them, they are older than its own
// loads.  (Fetched after the K loop they were a cold round trip at the end of every tile of every round: +8 us on the ViT's fc1, +10 on its qkv.)
template <int BM, int NT>
__device__ __forceinline__ void tile_row_scale_prefetch(const GemmP& p, int m0, float* Ls) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  constexpr int G = (BM + 63) / 64;                    // 64-row groups of the tile
  for (int pc = wave; pc < p.rs_nslots * G; pc += NT / 64) {
    const int sl = pc / G, g = pc - sl * G;
    int r = g * 64 + lane; r = r < BM ? r : BM - 1;   // (a last partial group re-reads the tile's last row: in-bounds, unused)
    int row = m0 + r; row = row < p.M ? row : p.M - 1;
    __builtin_amdgcn_global_load_lds((gptr_t)(p.rs_stats + (size_t)sl * p.rs_ld + row), (lptr_t)(Ls + sl * (G * 64) + g * 64), 4, 0, 0);
  }
}

template <int BM, int NT>
__device__ __forceinline__ void tile_row_scale(const GemmP& p, int m0, float* L, const float* Ls) {
  // slot-major statistics [slot][rs_ld]: a thread walks the slots of ONE row, so every load of a wave is one contiguous 256-byte run (a first form
  // with row-major slots -- 32 rows per load instruction -- cost the qkv GEMM 11 us per launch: profiles/r06_c).  Two threads per row when the
  // workgroup has them: the first / the second half of the slots, each in slot order, halves added last -- ONE fixed order for every tile kernel
  // and for both sources (Ls != null: the slots were prefetched into LDS, Ls[slot * G64 + r]).
  constexpr int TPR = NT >= 2 * BM ? 2 : 1;
  constexpr int G64 = (BM + 63) / 64 * 64;
  const int tid = threadIdx.x, h = tid / BM, r = tid % BM;
  if (h < TPR) {
    int row = m0 + r; row = row < p.M ? row : p.M - 1;
    const int per = (p.rs_nslots + TPR - 1) / TPR, s0 = h * per, s1 = s0 + per < p.rs_nslots ? s0 + per : p.rs_nslots;
    const float* base = p.rs_stats + row;
    float t = 0.f;
    for (int c0 = s0; c0 < s1; c0 += 16) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = c0 + k < s1 ? (Ls ? Ls[(c0 + k) * G64 + r] : base[(size_t)(c0 + k) * p.rs_ld]) : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += v[k];
    }
    L[h * BM + r] = t;
  }
  __syncthreads();
  if (tid < BM) {
    float t = L[tid];
    if (TPR == 2) t += L[BM + tid];
    L[tid] = rsqrtf(t * p.rs_inv_dim + p.rs_eps);
  }
  __syncthreads();
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmP p) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MR = WTM / 16, NR = WTN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_ROUNDS = (BM * 8 + NT - 1) / NT, B_ROUNDS = (BN * 8 + NT - 1) / NT;
  // a last partial round is taken by whole waves only (wave-uniform predicate: BM*8 and BN*8 are multiples of 64)
  static_assert((BM * 8) % 64 == 0 && (BN * 8) % 64 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile geometry");
  typedef typename V8<T>::type frag_t;
  extern __shared__ __attribute__((aligned(256))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int m0, n0;
  tile_coords(p.M, p.N, BM, BN, m0, n0);

  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;

  // ---- per-thread staging sources (row clamped into range: out-of-range rows are computed but never stored)
  const T* a_src[A_ROUNDS];
  const T* b_src[B_ROUNDS];
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) {
    const int lin = i * NT + tid, row = lin >> 3, pc = lin & 7;
    const int c = pc ^ ((row >> 1) & 7);
    int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
    a_src[i] = A + (size_t)gr * p.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) {
    const int lin = i * NT + tid, row = lin >> 3, pc = lin & 7;
    const int c = pc ^ ((row >> 1) & 7);
    int gr = n0 + row; gr = gr < p.N ? gr : p.N - 1;
    b_src[i] = W + (size_t)gr * p.ldw + c * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i)
      if ((i + 1) * NT <= BM * 8 || i * NT + wave * 64 < BM * 8)
        __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * 64), (lptr_t)(base + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i)
      if ((i + 1) * NT <= BN * 8 || i * NT + wave * 64 < BN * 8)
        __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + kt * 64), (lptr_t)(base + A_BYTES + (i * NT + wave * 64) * 16), 16, 0, 0);
  };

  // ---- fragment read offsets (bytes inside a stage)
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = (fr >> 1) & 7;
  int a_off[2], b_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int pc = ((s * 4 + fg) ^ swz) * 16;
    a_off[s] = (wm * WTM + fr) * 128 + pc;
    b_off[s] = A_BYTES + (wn * WTN + fr) * 128 + pc;
  }

  f32x4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / 64;
  float* const rs_ls = reinterpret_cast<float*>(smem + 2 * STAGE);      // behind the two stages (launch_cfg requests the bytes when rs_dma)
  if (p.rs_stats && p.rs_dma) tile_row_scale_prefetch<BM, NT>(p, m0, rs_ls);
  // ragged M (round 6, same rule as gemm8_segment): a wave whose rows all lie beyond M reads no operands and issues no MFMA
  const bool wave_live = !p.skip_dead || m0 + wm * WTM < p.M;
  stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nk) stage((t + 1) & 1, t + 1);
    if (!wave_live) continue;
    const char* base = smem + (t & 1) * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_t af[MR], bf[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const frag_t*>(base + a_off[s] + i * 16 * 128);
#pragma unroll
      for (int j = 0; j < NR; ++j) bf[j] = *reinterpret_cast<const frag_t*>(base + b_off[s] + j * 16 * 128);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = mfma16(bf[j], af[i], acc[i][j]);      // swapped operands: transposed accumulator (see gemm_epilogue)
    }
  }

  const float* rs_lds = nullptr;
  if (p.rs_stats) {        // wave-uniform; the barrier in front: a faster wave must not overwrite a stage another wave still reads
    __syncthreads();
    tile_row_scale<BM, NT>(p, m0, reinterpret_cast<float*>(smem), p.rs_dma ? rs_ls : nullptr);
    rs_lds = reinterpret_cast<const float*>(smem) + wm * WTM;
  }
  gemm_epilogue<T, MR, NR, EPI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, fg, fr, rs_lds);
}


// ---------------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves (2 M x 4 N), 4 phases per K-tile, the two wave groups (wm = 0 / 1) staggered by one
// barrier so that on every SIMD one wave is in its MFMA segment while its partner reads LDS / issues LDS-DMA
// (cdna_hip_programming.md §5 "The 256^2 8-phase template", T3/T4/T5; MI355X_MICROARCH "Two waves per SIMD").
//
// LDS = 2 stage buffers x 4 slots x 16 KiB.  A slot is what ONE phase consumes across all waves:
//   A_mh: tile rows wm*128 + mh*64 + [0,64) for wm = 0,1   (slot row = wm*64 + r)
//   B_nh: tile cols wn*64 + nh*32 + [0,32) for wn = 0..3    (slot row = wn*32 + c)
// Phase p of K-tile t (buffer t&1) computes accumulator quadrant (mh, nh) x K = 64 (16 MFMA):
//   p1 (0,0) reads A_0 + B_0     p2 (0,1) reads B_1     p3 (1,1) reads A_1     p4 (1,0) re-uses the B_0 registers
// so slots die early (A_0, B_0 after p1, B_1 after p2, A_1 after p3) and are refilled in place with the data of one or
// two K-tiles ahead while the rest of the buffer is still being consumed: every slot is issued >= 5 phases before its
// first read (4 half-tiles = 8 LDS-DMA per wave in flight).
// Per phase and wave:  LOAD: ds_reads, 2 x global_load_lds, s_waitcnt vmcnt(8), s_waitcnt lgkmcnt(0) (reads retired
// BEFORE the barrier: WAR safety for the staggered partner)  ->  s_barrier  ->  COMPUTE: 16 MFMA  ->  s_barrier.
// Group 1 runs one barrier behind group 0.  Hazard bookkeeping (phase numbers global, P = 4t + p; group 0 executes
// LOAD(P) before barrier 2P-1, group 1 before barrier 2P): a slot last read in LOAD(X) is retired for both groups at
// barrier 2X; a refill issued in LOAD(Y) starts after barrier 2Y-2, so Y >= X+1 is safe (the schedule keeps Y >= X+2);
// data issued in LOAD(Y) by every wave has landed for everyone after barrier 2(Y+4) (vmcnt(8) at the end of LOAD(Y+4)),
// so it may be read from LOAD(Y+5) on.
// ---------------------------------------------------------------------------------------------------------
// K-tiles [kt0, kt1) of the 256x256 tile at (m0, n0), accumulated into acc (zeroed here).  All 512 threads call it
// together; on return every wave has passed the same number of barriers and no LDS read is outstanding.
// one 16-byte operand fragment pair -> accumulator: 8 16-bit elements (one 16x16x32 MFMA) or 16 e4m3 bytes (two 16x16x32 fp8 MFMAs: the
// lane's bytes 0-7 and 8-15 are two K slots; A and B use the same assignment, so the K order inside a 128-byte row is immaterial)
// fp8 x fp8, block-scaled form (round 5): ONE v_mfma_scale_f32_16x16x128_f8f6f4 takes the 32 e4m3 bytes a lane holds for the whole 128-deep
// K step -- the two 16-byte chunks (s = 0, 1) that the non-scaled form feeds to four 16x16x32 MFMAs -- at twice the non-scaled rate
// (MI355X_MICROARCH.md, Matrix cores: the non-scaled fp8 opcodes run at the bf16 rate; only the block-scaled ones reach the fp8 peak).
// Scales: E8M0 bytes, one per lane and 32-element block; all 127 (= 2^0): the per-token / per-row fp32 scales stay in the epilogue, so the
// products summed here are exactly those of the non-scaled form.  A and B take the same lane -> K assignment, so the order of K inside the
// 128-byte row is immaterial (as above).  -DOMCHAT_F8_SCALED=0 builds the non-scaled twin for A/B.
#ifndef OMCHAT_F8_SCALED
#define OMCHAT_F8_SCALED 1
#endif
template <typename T>
__device__ __forceinline__ f32x4 mma_frag_f8x128(typename V8<T>::type a0, typename V8<T>::type a1, typename V8<T>::type b0, typename V8<T>::type b1, f32x4 c) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  const i32x4 al = __builtin_bit_cast(i32x4, a0), ah = __builtin_bit_cast(i32x4, a1), bl = __builtin_bit_cast(i32x4, b0), bh = __builtin_bit_cast(i32x4, b1);
  const i32x8 a = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
  const i32x8 b = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

template <typename T, bool F8>
__device__ __forceinline__ f32x4 mma_frag(typename V8<T>::type a, typename V8<T>::type b, f32x4 c) {
  if constexpr (F8) {
    typedef long l2 __attribute__((ext_vector_type(2)));
    const l2 a2 = __builtin_bit_cast(l2, a), b2 = __builtin_bit_cast(l2, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[0], b2[0], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[1], b2[1], c, 0, 0, 0);
  } else {
    return mfma16(a, b, c);
  }
}

// Persistent use (gemm8p_kernel): prologue_issued = the staging loads of this tile's first K-tiles were already issued by the previous call
// (with has_next), which recomputes the staging sources for tile (next_m0, next_n0) after its own last barrier -- every LDS read of the
// finished tile is retired there for both wave groups -- and issues that tile's prologue, so that the caller's epilogue (VALU + global
// stores, no LDS) runs under the next tile's first HBM round trip.
template <typename T, bool F8 = false>
__device__ __forceinline__ void gemm8_segment(const GemmP& p, int m0, int n0, int kt0, int kt1, char* smem, f32x4 (&acc)[8][4],
                                              bool prologue_issued = false, bool has_next = false, int next_m0 = 0, int next_n0 = 0) {
  constexpr int SLOT = 128 * 128;                 // 128 rows x 128 B
  constexpr int STAGE = 4 * SLOT;                 // A_0, A_1, B_0, B_1
  constexpr int ES = F8 ? 1 : 2;                  // bytes per operand element: a 128-byte row holds 128 / ES elements of K
  typedef typename V8<T>::type frag_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const char* __restrict__ A = (const char*)p.A;
  const char* __restrict__ W = (const char*)p.W;

  // staging sources (byte pointers): slot piece lin = i*512 + tid -> slot row rho = lin >> 3, physical chunk pc = lin & 7
  const char* a_src[2][2];     // [mh][round]
  const char* b_src[2][2];     // [nh][round]
  auto set_src = [&](int tm0, int tn0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lin = i * 512 + tid, rho = lin >> 3, pc = lin & 7;
      const int c = pc ^ ((rho >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int gr = tm0 + (rho >> 6) * 128 + h * 64 + (rho & 63); gr = gr < p.M ? gr : p.M - 1;
        a_src[h][i] = A + (size_t)gr * p.lda * ES + c * 16;
        int gc = tn0 + (rho >> 5) * 64 + h * 32 + (rho & 31); gc = gc < p.N ? gc : p.N - 1;
        b_src[h][i] = W + (size_t)gc * p.ldw * ES + c * 16;
      }
    }
  };
  set_src(m0, n0);
  // slot order inside a stage: 0 = A_0, 1 = A_1, 2 = B_0, 3 = B_1
  auto issue = [&](int buf, int slot, const char* const (&src)[2], int kt) {
    char* base = smem + buf * STAGE + slot * SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (size_t)(kt0 + kt) * 128), (lptr_t)(base + (i * 512 + wave * 64) * 16), 16, 0, 0);
  };

  const int fr = lane & 15, fg = lane >> 4;
  const int swz = (fr >> 1) & 7;
  int a_off[2], b_off[2];            // per k-step, inside a slot
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int pc = ((s * 4 + fg) ^ swz) * 16;
    a_off[s] = (wm * 64 + fr) * 128 + pc;
    b_off[s] = (wn * 32 + fr) * 128 + pc;
  }

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = kt1 - kt0;
  // Ragged M (round 6): rows beyond M are staged clamped and never stored; their MFMAs were still issued -- the 13th row tile of the 3-tile
  // ViT (M = 3075 = 12 x 256 + 3) ran 64 MFMAs per wave and K-tile for three valid rows.  A 64-row block (wave group wm, half mh) that lies
  // wholly beyond M now skips its operand reads and MFMAs (wave-uniform branches; barriers, staging and the slot schedule are untouched, so
  // the valid rows are computed by exactly the same instructions: same bits).  tuning key 43 = 0 restores the full issue for A/B.
  const int rows_left = p.skip_dead ? p.M - (m0 + wm * 128) : 128;
  const bool live0 = rows_left > 0, live1 = rows_left > 64;
  // prologue: K-tile 0 completely, plus A_0 / B_0 of K-tile 1 (steady state issues them in p3 / p4 of tile t-1)
  auto issue_prologue = [&]() {
    issue(0, 0, a_src[0], 0); issue(0, 2, b_src[0], 0); issue(0, 3, b_src[1], 0); issue(0, 1, a_src[1], 0);
    if (nk > 1) { issue(1, 0, a_src[0], 1); issue(1, 2, b_src[0], 1); }
  };
  if (!prologue_issued) issue_prologue();
  // (persistent use: the epilogue stores of the previous tile were issued after these loads; vmcnt is in order, so the counted wait
  // retires the loads it is meant for and whatever stores are older than the 4 youngest operations)
  if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();      // stagger: group 1 runs one barrier behind

  // Issue schedule (slot X of tile u goes to buffer u&1; its previous content died >= 2 phases earlier because the
  // B_0 fragments are kept in registers from p1 to p4):
  //   p1(t): B_1(t+1)   p2(t): A_1(t+1)   p3(t): A_0(t+2)   p4(t): B_0(t+2)
  // First reads: p1 A_0,B_0   p2 B_1   p3 A_1   -> every slot is issued >= 5 phases before its first read, and
  // s_waitcnt vmcnt(8) at the end of a LOAD segment (4 half-tiles may stay in flight) retires what the NEXT phase reads.
  frag_t af[4][2], bf0[2][2], bf1[2][2];
  for (int t = 0; t < nk; ++t) {
    const char* st = smem + (t & 1) * STAGE;
    const int nb = (t + 1) & 1;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;

#define OM_LOAD_A(MH)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int s = 0; s < 2; ++s)                      \
      af[i][s] = *reinterpret_cast<const frag_t*>(st + (MH) * SLOT + a_off[s] + i * 16 * 128);
#define OM_LOAD_B(BF, NH)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int s = 0; s < 2; ++s)                      \
      BF[j][s] = *reinterpret_cast<const frag_t*>(st + (2 + (NH)) * SLOT + b_off[s] + j * 16 * 128);
#define OM_SYNC_COMPUTE(MH, NH, BF)                                                                                \
  if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_barrier();                                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_setprio(1);                                                                                   \
  if (!((MH) ? live1 : live0)) {                                                                                   \
  } else                                                                                                           \
  if constexpr (F8 && OMCHAT_F8_SCALED) {                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
        acc[(MH) * 4 + i][(NH) * 2 + j] = mma_frag_f8x128<T>(BF[j][0], BF[j][1], af[i][0], af[i][1], acc[(MH) * 4 + i][(NH) * 2 + j]); \
  } else {                                                                                                         \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < 4; ++i)                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
          acc[(MH) * 4 + i][(NH) * 2 + j] = mma_frag<T, F8>(BF[j][s], af[i][s], acc[(MH) * 4 + i][(NH) * 2 + j]);  \
  }                                                                                                                \
  __builtin_amdgcn_s_setprio(0);                                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_barrier();                                                                                    \
  __builtin_amdgcn_sched_barrier(0);

    // phase 1: quadrant (0,0)
    if (live0) { OM_LOAD_B(bf0, 0) OM_LOAD_A(0) }
    if (more1) issue(nb, 3, b_src[1], t + 1);
    OM_SYNC_COMPUTE(0, 0, bf0)
    // phase 2: quadrant (0,1)
    if (live0) { OM_LOAD_B(bf1, 1) }
    if (more1) issue(nb, 1, a_src[1], t + 1);
    OM_SYNC_COMPUTE(0, 1, bf1)
    // phase 3: quadrant (1,1)
    if (live1) { OM_LOAD_A(1) }
    if (more2) issue(t & 1, 0, a_src[0], t + 2);
    OM_SYNC_COMPUTE(1, 1, bf1)
    // phase 4: quadrant (1,0), B_0 fragments still in registers
    if (more2) issue(t & 1, 2, b_src[0], t + 2);
    OM_SYNC_COMPUTE(1, 0, bf0)
#undef OM_LOAD_A
#undef OM_LOAD_B
#undef OM_SYNC_COMPUTE
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();      // group 0 matches group 1's extra barrier
  if (has_next) {                                  // uniform: both groups are past their last LDS read here
    set_src(next_m0, next_n0);
    issue_prologue();
  }
}


template <typename T, int EPI, bool F8 = false>
__global__ __launch_bounds__(512) void gemm8_kernel(GemmP p, int skew) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  int m0, n0;
  tile_coords(p.M, p.N, 256, 256, m0, n0);          // data-parallel: workgroup = one whole tile (logical ids [0, gridDim))
  if (skew > 0) {
    // Start skew between the workgroups of an XCD that share an operand panel (logical ids walk GROUP_M rows, then the next
    // column): tiles that begin in lockstep all miss on the same L2 lines at the same moment and every miss goes to the
    // fabric (TCC_MISS ~ sharers x unique bytes, measured); a delay of a fraction of a K-step lets the followers hit.
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int d = (id & 3) + ((id >> 2) & 3);
    for (int i = 0; i < d * skew; ++i) __builtin_amdgcn_s_sleep(16);
  }
  float* const rs_ls = reinterpret_cast<float*>(smem + 2 * 4 * 128 * 128);      // behind the stage buffers (launch_cfg8 requests the bytes when rs_dma)
  if (p.rs_stats && p.rs_dma) tile_row_scale_prefetch<256, 512>(p, m0, rs_ls);
  f32x4 acc[8][4];
  gemm8_segment<T, F8>(p, m0, n0, 0, p.K / (F8 ? 128 : 64), smem, acc);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fg = lane >> 4;
  const float* rs_lds = nullptr;
  if (p.rs_stats) {        // wave-uniform (gemm8_segment returns with every LDS read of both wave groups retired)
    __syncthreads();
    tile_row_scale<256, 512>(p, m0, reinterpret_cast<float*>(smem), p.rs_dma ? rs_ls : nullptr);
    rs_lds = reinterpret_cast<const float*>(smem) + wm * 128;
  }
  gemm_epilogue<T, 8, 4, EPI, F8>(p, acc, m0 + wm * 128, n0 + wn * 64, fg, fr, rs_lds);
}

// Persistent form for multi-round launches (round 3): one workgroup per CU walks tiles w, w + G, w + 2G, ... and issues the NEXT tile's
// first staging loads before the CURRENT tile's epilogue.  tools/bench_gemm_k.py (K = 64: the fixed cost of a launch) shows what a round
// costs beyond its K loop with one workgroup per tile: ~15 us (EPI_NONE) to ~29 us (layer-scale + residual) per round on the 3-tile ViT
// shapes -- workgroup dispatch, the first HBM round trip of the prologue, the epilogue and its store drain, none of it overlapped because
// all 256 workgroups move in lockstep.  Here the prologue's round trip and the store drain run under the neighbouring tile's work and
// the per-round dispatch disappears.
template <typename T, int EPI, bool F8 = false>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmP p, int tiles) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int G = gridDim.x, w = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fg = lane >> 4;
  // Tile order = the one-workgroup-per-tile launch's: XCD x (workgroups w with w % 8 == x, dealt round-robin by the dispatcher) owns the
  // contiguous logical range xcd_remap gives it and its G / 8 workgroups walk it G / 8 ids per round, so the 32 tiles an XCD runs together
  // form the same 4 x 8 block and consecutive rounds move along the same A row panels (a per-round remap measured +2.8 % on the 259-round
  // prefill gate|up: every round met a cold A panel).  G is a multiple of 8.
  const int xcd = w & 7, per = G >> 3;
  const int q8 = tiles >> 3, r8 = tiles & 7;
  const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int count = q8 + (xcd < r8 ? 1 : 0);
  const int nk = p.K / (F8 ? 128 : 64);
  int idx = w >> 3;
  if (idx >= count) return;
  int m0, n0;
  tile_coords_id(start + idx, p.M, p.N, 256, 256, m0, n0);
  bool issued = false;
  for (; idx < count; idx += per) {
    const bool more = idx + per < count;
    int m1 = 0, n1 = 0;
    if (more) tile_coords_id(start + idx + per, p.M, p.N, 256, 256, m1, n1);
    f32x4 acc[8][4];
    gemm8_segment<T, F8>(p, m0, n0, 0, nk, smem, acc, issued, more, m1, n1);
    gemm_epilogue<T, 8, 4, EPI, F8>(p, acc, m0 + wm * 128, n0 + wn * 64, fg, fr);
    issued = more; m0 = m1; n0 = n1;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Stream-K tail: the tiles of the last, partially filled round (logical ids [tile_first, tile_first + R)) are cut
// into R * KT K-tile iterations and dealt to the G resident workgroups in contiguous, equal ranges, so every CU gets
// the same number of MFMA iterations whatever R is.  A range may START inside a tile: that workgroup stores the fp32
// accumulators of this first segment (one 256 KiB slab per workgroup, lane-linear) and raises its flag at once; the
// workgroup that began the tile (k = 0, necessarily its LAST segment) adds the slabs of the following workgroups in
// ascending order (deterministic) and runs the epilogue -- producers publish early, owners wait late: no chains.  Hand-off = cdna_hip_programming.md Guideline 16: every storing wave drains vmcnt, workgroup barrier, one lane
// agent-scope release + drain + relaxed agent flag store; consumer: one lane polls relaxed (bounded), one agent-scope
// acquire + drain, workgroup barrier, plain loads.  G <= #CUs with one workgroup per CU (128 KiB LDS), so every awaited
// producer is resident and producers never wait: no deadlock; flags are zeroed by a memset
// node before every launch.
// ---------------------------------------------------------------------------------------------------------
struct SkP { int tile_first, R; float* slabs; unsigned* flags; unsigned* err; };

template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm8_sk_kernel(GemmP p, SkP sk) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int KT = p.K / 64;
  const int total = sk.R * KT;                      // <= 256 tiles x K/64: fits int with room for the * G below
  const int G = gridDim.x, w = blockIdx.x;
  auto begin_of = [&](int g) { return (int)((long)g * total / G); };
  int it = begin_of(w);
  const int it_end = begin_of(w + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fg = lane >> 4;
  f32x4 acc[8][4];
  while (it < it_end) {
    const int tl = it / KT;
    const int k0 = it - tl * KT;
    const int k1 = KT - k0 < it_end - it ? KT : k0 + (it_end - it);
    int m0, n0;
    tile_coords_id(sk.tile_first + tl, p.M, p.N, 256, 256, m0, n0);
    gemm8_segment<T>(p, m0, n0, k0, k1, smem, acc);
    if (k0 > 0) {
      // contributor: this range starts inside a tile that a LOWER workgroup began; publish the slab right away (it is
      // the first thing this workgroup computes, so the owner never waits on a chain)
      // buffer stores: one descriptor (SGPRs) + one per-lane offset; the per-slot offset is a scalar, so no address VGPRs
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sk.slabs + (size_t)w * 65536, 0, 65536 * 4, 0x00020000);
      const int voff = (wave * 2048 + lane) * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, voff, (i * 4 + j) * 1024, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(sk.flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (k1 < KT) {
        // owner of a tile it cannot finish (its range ends inside the tile, so this is its LAST segment): add the slabs of
        // the following workgroups, in ascending order, until the tile's K range is covered
        const int tile_end = (tl + 1) * KT;
        int covered = tl * KT + k1;
        for (int c = w + 1; covered < tile_end && c < G; ++c) {
          if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(sk.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > (1u << 24)) { atomicExch(sk.err, 1u); break; }      // bounded: never hang the device
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sk.slabs + (size_t)c * 65536, 0, 65536 * 4, 0x00020000);
          const int voff = (wave * 2048 + lane) * 16;
          // 4 loads in flight at a time: hoisting all 32 would spill the accumulators
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (i * 4 + j) * 1024, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += __builtin_bit_cast(f32x4, v[j]);
            __builtin_amdgcn_sched_barrier(0);
          }
          const int e = begin_of(c + 1);
          covered = e < tile_end ? e : tile_end;
        }
      }
      gemm_epilogue<T, 8, 4, EPI>(p, acc, m0 + wm * 128, n0 + wn * 64, fg, fr);
    }
    it += k1 - k0;
  }
}

int g_gemm_persist = 0;   // omchat_op_set_tuning key 13: 1 = multi-round 256^2 GEMMs take the persistent form (measured neutral: DESIGN.md section 6)
int g_gemm_skew = 0;      // tuning knob (omchat_op_set_tuning), units of s_sleep(16) ~ 1024 cycles
constexpr size_t SK_SLAB_BYTES = 65536 * 4;
constexpr int SK_MAX_WG = 256;

template <typename T, int EPI>
int launch_cfg8(const GemmArgs& a, hipStream_t stream) {
  constexpr int LDS = 2 * 4 * 128 * 128;
  auto kern = gemm8_kernel<T, EPI>;
  auto kern_p = gemm8p_kernel<T, EPI>;
  auto kern_sk = gemm8_sk_kernel<T, EPI>;
  static PerDeviceOnce attr_set;
  const int n_cu = device_cus();
  if (attr_set.first()) {
    OM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      // stage buffers + prefetched row statistics
    OM_HIP(hipFuncSetAttribute((const void*)kern_p, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    OM_HIP(hipFuncSetAttribute((const void*)kern_sk, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  GemmP p{a.A, a.W, a.C, a.bias, a.ls, a.resid, a.lda, a.ldw, a.ldc, a.ldr, a.M, a.N, a.K, nullptr, nullptr, g_gemm_wide_store, g_gemm_skip_dead, a.row_scale, a.stats, a.stats_ld, a.rs_stats, a.rs_ld, a.rs_nslots, a.rs_dim > 0 ? 1.0f / (float)a.rs_dim : 0.f, a.rs_eps, 0};
  const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  if (a.stats_nslots) *a.stats_nslots = cdiv(a.N, 64);      // one slot per 64-column wave tile
  // row statistics through LDS-DMA when the tile's slots fit behind the stage buffers (160 KiB per workgroup): 29 slots x 256 rows = 29 KiB
  const int rs_bytes = a.rs_stats ? a.rs_nslots * 256 * 4 : 0;
  p.rs_dma = rs_bytes > 0 && LDS + rs_bytes <= 160 * 1024;
  const int lds_main = p.rs_dma ? LDS + rs_bytes : LDS;
  const int KT = a.K / 64;
  const int G = n_cu < SK_MAX_WG ? n_cu : SK_MAX_WG;
  int R = tiles % G;
  const bool have_ws = a.sk_ws && a.sk_ws_bytes >= (size_t)G * SK_SLAB_BYTES + 4096;
  // stream-K only pays when the last round is visibly under-filled and every workgroup still gets a few K-tiles
  // measured on MI355X (r01, tools/bench_gemm.py t20 vs t2): the under-filled GEMMs of this model are operand-fetch bound,
  // not balance bound, and the slab exchange costs more than the idle CUs: stream-K only runs when explicitly requested
  const bool use_sk = a.stream_k > 0 && have_ws && !a.rs_stats && R > 0 && R * 10 < G * 9 && (long)R * KT >= 4L * G && KT >= 8;
  if (a.stream_k > 0 && !use_sk && !have_ws) { omchat_set_error("launch_gemm: stream-K requested without workspace"); return 1; }
  if (!use_sk) R = 0;
  const int n_dp = tiles - R;
  if (n_dp > G && g_gemm_persist && R == 0 && G >= 8 && !a.rs_stats) hipLaunchKernelGGL(kern_p, dim3(G & ~7), dim3(512), LDS, stream, p, n_dp);
  else if (n_dp > 0) hipLaunchKernelGGL(kern, dim3(n_dp), dim3(512), lds_main, stream, p, g_gemm_skew);
  if (R > 0) {
    unsigned* flags = reinterpret_cast<unsigned*>((char*)a.sk_ws + (size_t)G * SK_SLAB_BYTES);
    OM_HIP(hipMemsetAsync(flags, 0, 4096, stream));
    SkP sk{n_dp, R, (float*)a.sk_ws, flags, flags + 1000};
    hipLaunchKernelGGL(kern_sk, dim3(G), dim3(512), LDS, stream, p, sk);
  }
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T, int EPI>
int launch_cfg8_f8(const GemmArgs& a, hipStream_t stream) {
  constexpr int LDS = 2 * 4 * 128 * 128;
  auto kern = gemm8_kernel<T, EPI, true>;
  auto kern_p = gemm8p_kernel<T, EPI, true>;
  static PerDeviceOnce attr_set;
  const int n_cu = device_cus();
  if (attr_set.first()) {
    OM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    OM_HIP(hipFuncSetAttribute((const void*)kern_p, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  GemmP p{a.A, a.W, a.C, a.bias, a.ls, a.resid, a.lda, a.ldw, a.ldc, a.ldr, a.M, a.N, a.K, a.a_scale, a.w_scale, g_gemm_wide_store, g_gemm_skip_dead, a.row_scale, a.stats, a.stats_ld, a.rs_stats, a.rs_ld, a.rs_nslots, a.rs_dim > 0 ? 1.0f / (float)a.rs_dim : 0.f, a.rs_eps, 0};
  const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  if (tiles > n_cu && g_gemm_persist && n_cu >= 8 && !a.rs_stats) hipLaunchKernelGGL(kern_p, dim3(n_cu & ~7), dim3(512), LDS, stream, p, tiles);
  else hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, stream, p, 0);
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  constexpr int LDS = 2 * (BM + BN) * 128;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI>;
  static PerDeviceOnce attr_set;
  constexpr int LDS_MAX = LDS + 32 * 1024 <= 160 * 1024 ? LDS + 32 * 1024 : 160 * 1024;
  if (attr_set.first()) OM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  GemmP p{a.A, a.W, a.C, a.bias, a.ls, a.resid, a.lda, a.ldw, a.ldc, a.ldr, a.M, a.N, a.K, nullptr, nullptr, g_gemm_wide_store, g_gemm_skip_dead, a.row_scale, a.stats, a.stats_ld, a.rs_stats, a.rs_ld, a.rs_nslots, a.rs_dim > 0 ? 1.0f / (float)a.rs_dim : 0.f, a.rs_eps, 0};
  const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
  if (a.stats_nslots) *a.stats_nslots = cdiv(a.N, BN / WN);      // one slot per wave tile
  const int rs_bytes = a.rs_stats ? a.rs_nslots * ((BM + 63) / 64 * 64) * 4 : 0;      // row statistics prefetched by LDS-DMA behind the two stages
  p.rs_dma = rs_bytes > 0 && LDS + rs_bytes <= LDS_MAX;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), p.rs_dma ? LDS + rs_bytes : LDS, stream, p);
  OM_LAUNCH_CHECK();
  return 0;
}

// Tile menu.  256 CUs take one 8-wave workgroup each (two for the 128^2 tile), so the cost of a launch is
// rounds(tiles / slots) x per-tile time; the shape that fills the last round best wins (M = 3075 x N = 3200 is
// 169 tiles of 256^2 = 66 % of one round, but 221 tiles of 256x192 = 86 %).  `eff` = measured MFMA efficiency of the
// shape relative to 256^2 (smaller tiles re-read more LDS per MFMA).
struct TileCfg { int id, bm, bn, slots; float eff; };
// Measured (r01): 224/192-row tiles lose (power-limited chip: idle CUs give their budget to the busy ones), so the
// automatic menu is 256^2, 256x192 (N = 3200 shapes: -13 % time) and 128^2 for tiny problems.
static const TileCfg kTiles[] = {{2, 256, 256, 256, 1.00f}, {3, 256, 192, 256, 0.90f}, {1, 128, 128, 512, 0.60f}};

static int pick_tile(int M, int N, int epi) {
  if ((long)cdiv(M, 256) * cdiv(N, 256) >= 3 * 256) return 2;     // many rounds: the tail round is amortised
  int best = 2;
  float best_cost = 1e30f;
  for (const TileCfg& t : kTiles) {
    if (epi == EPI_SWIGLU && (t.bn / 4) % 32 != 0) continue;      // gate|up pairs must not straddle a wave's columns
    const long tiles = (long)cdiv(M, t.bm) * cdiv(N, t.bn);
    const long rounds = (tiles + t.slots - 1) / t.slots;
    const float cost = (float)rounds * (float)(t.bm * t.bn) / t.eff * (t.slots == 512 ? 2.0f : 1.0f) + 0.02f * 65536.f * (float)rounds;
    if (cost < best_cost) { best_cost = cost; best = t.id; }
  }
  return best;
}

template <typename T, int EPI>
int launch_epi(const GemmArgs& a, hipStream_t stream) {
  const int tile = a.force_tile ? a.force_tile : pick_tile(a.M, a.N, EPI);
  switch (tile) {
    case 1: return launch_cfg<T, 128, 128, 2, 2, EPI>(a, stream);
    case 3: return launch_cfg<T, 256, 192, 2, 4, EPI>(a, stream);
    case 4: return launch_cfg<T, 224, 256, 2, 4, EPI>(a, stream);
    case 5: return launch_cfg<T, 192, 256, 2, 4, EPI>(a, stream);
    case 6: return launch_cfg<T, 256, 256, 2, 4, EPI>(a, stream);     // one-barrier-per-K-step structure (kept for A/B)
    case 7: return launch_cfg<T, 256, 192, 4, 3, EPI>(a, stream);     // 12 waves of 64x64: 3 waves per SIMD
    case 8: return launch_cfg<T, 256, 256, 4, 4, EPI>(a, stream);     // 16 waves of 64x64: 4 waves per SIMD
    case 9: return launch_cfg<T, 192, 256, 3, 4, EPI>(a, stream);
    // round 5: the N = 3200 outputs of the 3-tile ViT (proj, fc2; M = 3075) are 17 x 13 = 221 tiles of 192 x 256 -- 86 % of ONE round, so the launch
    // lasts one tile whatever the fill; 192 x 224 tiles are 17 x 15 = 255 tiles (99.6 % of the round) of 7 / 8 the size.  8 waves of 48 x 112
    // (3 x 7 fragments: 10 operand reads per 21 MFMAs) or 12 waves of 32 x 112
    case 10: if constexpr (EPI != EPI_SWIGLU) return launch_cfg<T, 192, 224, 4, 2, EPI>(a, stream); else break;
    case 11: if constexpr (EPI != EPI_SWIGLU) return launch_cfg<T, 192, 224, 6, 2, EPI>(a, stream); else break;
    // round 5: two launches over disjoint column ranges.  M = 3075 x N = 12800 (ViT fc1) is 650 tiles of 256^2 = 2.54 rounds on 256 CUs: the
    // launch lasts THREE tile times.  The leading columns that make whole rounds of 256^2 tiles (39 column tiles x 13 row tiles = 507 <= 512) go to
    // the staggered 256^2 kernel, the remaining 2816 columns to ONE round of 192 x 224 tiles (13 x 17 = 221): ~2.7 tile times.  No extra bytes:
    // both launches read A once more through L2 only; outputs, bias and residual are column slices of the same buffers.
    case 12: case 13:
      if constexpr (EPI != EPI_SWIGLU) {
        const int G = device_cus(), rt = cdiv(a.M, 256), ct = cdiv(a.N, 256);
        const int full = (rt * ct) / G;                                     // whole rounds of 256^2 tiles
        const int ct_main = full >= 1 ? std::min(ct, (full * G) / rt) : 0;
        const int n_main = ct_main * 256, n_tail = a.N - n_main;
        if (n_main > 0 && n_tail > 0 && (long)cdiv(a.M, 192) * cdiv(n_tail, 224) <= G) {
          GemmArgs m = a, t = a;
          m.N = n_main; m.force_tile = 2;
          const size_t es = 2;
          t.N = n_tail; t.force_tile = a.force_tile == 12 ? 10 : 11;
          t.W = (const char*)a.W + (size_t)n_main * a.ldw * es;
          t.C = (char*)a.C + (size_t)n_main * es;
          if (a.bias) t.bias = (const char*)a.bias + (size_t)n_main * es;
          if (a.ls) t.ls = (const char*)a.ls + (size_t)n_main * es;
          if (a.resid) t.resid = (const char*)a.resid + (size_t)n_main * es;
          int ns_main = 0, ns_tail = 0;
          if (a.stats) { t.stats = a.stats + (size_t)(n_main / 64) * a.stats_ld; m.stats_nslots = &ns_main; t.stats_nslots = &ns_tail; }      // the tail's slots behind the main launch's
          int rc = launch_cfg8<T, EPI>(m, stream);
          if (rc) return rc;
          rc = t.force_tile == 10 ? launch_cfg<T, 192, 224, 4, 2, EPI>(t, stream) : launch_cfg<T, 192, 224, 6, 2, EPI>(t, stream);
          if (a.stats_nslots) *a.stats_nslots = ns_main + ns_tail;
          return rc;
        }
      }
      break;
    default: return launch_cfg8<T, EPI>(a, stream);
  }
  return launch_cfg8<T, EPI>(a, stream);      // (a tile that does not apply to this problem / epilogue: the 256^2 kernel)
}

template <typename T>
int launch_t(const GemmArgs& a, hipStream_t stream) {
  switch (a.epi) {
    case EPI_LS_RESID_STATS: return launch_epi<T, EPI_LS_RESID_STATS>(a, stream);
    case EPI_NONE_STATS: {      // the fused qkv output: 64-column wave tiles only, so that no slot straddles the q / k / v thirds (multiples of 64)
      GemmArgs b = a;
      if (b.force_tile != 2 && b.force_tile != 6 && b.force_tile != 8 && b.force_tile != 9 && b.force_tile != 7) b.force_tile = 2;
      return launch_epi<T, EPI_NONE_STATS>(b, stream);
    }
    case EPI_NONE: return launch_epi<T, EPI_NONE>(a, stream);
    case EPI_GELU: return launch_epi<T, EPI_GELU>(a, stream);
    case EPI_LS_RESID: return launch_epi<T, EPI_LS_RESID>(a, stream);
    case EPI_RESID: return launch_epi<T, EPI_RESID>(a, stream);
    case EPI_SWIGLU: return launch_epi<T, EPI_SWIGLU>(a, stream);
    case EPI_F32OUT: return a.force_tile == 1 ? launch_cfg<T, 128, 128, 2, 2, EPI_F32OUT>(a, stream) : launch_cfg8<T, EPI_F32OUT>(a, stream);
  }
  omchat_set_error("launch_gemm: bad epilogue");
  return 1;
}

}  // namespace

void gemm_set_skew(int v) { g_gemm_skew = v; }
void gemm_set_persist(int v) { g_gemm_persist = v; }
void gemm_set_wide_store(int v) { g_gemm_wide_store = v; }
void gemm_set_skip_dead(int v) { g_gemm_skip_dead = v; }

// ---------------------------------------------------------------------------------------------------------
// Tile choice by measurement.  Which kernel wins depends on the shape in ways a fill-the-last-round model does not capture
// (MI355X, bf16, r01: the staggered 8-wave 256^2 kernel wins on long-K and exactly-one-round shapes; the 16-wave 256^2 one-barrier
// kernel -- 4 waves per SIMD -- is 14 % faster on the ViT qkv shape and 15 % on o_proj; the 12-wave 192x256 kernel is 13-17 %
// faster on the N = 3200 shapes).  So the first time a (dtype, epilogue, ceil(M/256), N, K) problem is seen, the candidates are
// timed on the caller's own A / W with a scratch C (the real C may alias the residual input, so it is never written while
// tuning) and the winner is cached for the life of the process.  All kernels accumulate every output element over K in the same
// order, so the choice changes the speed and nothing else.  omchat_op_set_tuning(key 5, 0) falls back to the cost model.
// ---------------------------------------------------------------------------------------------------------
static int g_autotune = getenv("OMCHAT_GEMM_AUTOTUNE") ? atoi(getenv("OMCHAT_GEMM_AUTOTUNE")) : 1;
void gemm_set_autotune(int v) { g_autotune = v; }

static std::mutex g_tune_mu;
static std::map<std::array<int, 5>, int> g_tuned;
static long g_tune_runs = 0;      // first-use measurements performed by this process (0 when every shape came from the cache file)

// Persisted winners (first-use tuning must not run inside a timed or multi-rank region, and costs a live request
// 5 candidates x 4 launches + 384 MB flushes): one line per problem class "dtype*8+epi  ceil(M/256)  N  K  ldc  tile".
// omchat_amd/gemm_tune_gfx950.txt (measured on MI355X, committed) is loaded by the Python binding when the library is loaded;
// classes it does not cover are still measured on first use and can be appended with omchat_gemm_tune_dump.
int gemm_tune_load(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  std::lock_guard<std::mutex> lock(g_tune_mu);
  int n = 0, a, b, c, d, e, t;
  char line[256];
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#') continue;
    if (sscanf(line, "%d %d %d %d %d %d", &a, &b, &c, &d, &e, &t) == 6 && t >= 1 && t <= 13) { g_tuned[{a, b, c, d, e}] = t; ++n; }
  }
  fclose(f);
  return n;
}
int gemm_tune_dump(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  std::lock_guard<std::mutex> lock(g_tune_mu);
  fprintf(f, "# omchat_amd GEMM tile choices measured on this device: dtype*8+epi ceil(M/256) N K ldc tile\n");
  for (auto& kv : g_tuned) fprintf(f, "%d %d %d %d %d %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.first[4], kv.second);
  fclose(f);
  return (int)g_tuned.size();
}
long gemm_tune_runs() { return g_tune_runs; }

// choices that were NOT measured (the cost model's answer for a launch-bound shape, a neighbouring class's winner): remembered for the
// life of the process, never written to the tune file -- g_tuned holds measurements (made here or loaded) only
static std::map<std::array<int, 5>, int> g_guess;

static int tuned_tile(int dtype, const GemmArgs& a, hipStream_t stream) {
  if (a.epi == EPI_LS_RESID_STATS || a.epi == EPI_NONE_STATS) {      // the class (and the measurement) of the base epilogue
    GemmArgs b = a;
    b.epi = a.epi == EPI_NONE_STATS ? EPI_NONE : EPI_LS_RESID;
    b.stats = nullptr;
    return tuned_tile(dtype, b, stream);
  }
  const std::array<int, 5> key{dtype * 8 + a.epi, cdiv(a.M, 256), a.N, a.K, a.ldc};
  std::lock_guard<std::mutex> lock(g_tune_mu);
  auto it = g_tuned.find(key);
  if (it != g_tuned.end()) return it->second;
  it = g_guess.find(key);
  if (it != g_guess.end()) return it->second;
  const int heuristic = pick_tile(a.M, a.N, a.epi);
  if ((double)a.M * a.N * a.K < 4e9) return g_guess[key] = heuristic;            // < 8 GFLOP: launch-bound, nothing to choose
  {
    // a MEASURED class of the same (dtype, epilogue, N, K, ldc) whose row-tile count is within one tile or 1/8: prompts come in every
    // length, and a live request must not pay 5 candidates x 4 launches because its S rounds to a row-tile count nobody has seen.
    // Ties go to the smaller row-tile count (std::map order), so the answer does not depend on the order in which shapes were seen.
    int best_d = INT_MAX, best_t = 0;
    for (auto& kv : g_tuned) {
      const auto& k = kv.first;
      if (k[0] != key[0] || k[2] != key[2] || k[3] != key[3] || k[4] != key[4]) continue;
      const int d = k[1] > key[1] ? k[1] - key[1] : key[1] - k[1];
      if (d < best_d) { best_d = d; best_t = kv.second; }
    }
    if (best_t && best_d <= std::max(1, key[1] / 8)) return g_guess[key] = best_t;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return heuristic;      // cannot time inside a capture
  GemmArgs t = a;
  t.M = a.M < 8192 ? a.M : 8192;                                                 // enough rounds to rank the kernels, bounded scratch
  void* scratch = nullptr;
  if (hipMalloc(&scratch, (size_t)t.M * a.ldc * 2) != hipSuccess) { (void)hipGetLastError(); return g_guess[key] = heuristic; }
  t.C = scratch;
  // In the model every GEMM meets its weights cold (27 GB stream through a 256 MB Infinity Cache between two uses), while
  // back-to-back timing runs would find them cached and rank the kernels differently (o_proj: 102 us warm, 120 us in place).
  // So each timed run is preceded by a write sweep over a buffer larger than the cache.
  constexpr size_t FLUSH_BYTES = (size_t)384 << 20;
  void* flush = nullptr;
  if (hipMalloc(&flush, FLUSH_BYTES) != hipSuccess) { (void)hipGetLastError(); flush = nullptr; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  ++g_tune_runs;
  int cands[9] = {2, 8, 9, 7, 3, 10, 11, 12, 13};
  int best = heuristic;
  float best_ms = 1e30f;
  for (int c : cands) {
    if ((c == 3 || c >= 10) && a.epi == EPI_SWIGLU) continue;           // 48 / 112 columns per wave would split gate|up pairs
    t.force_tile = c;
    bool ok = true;
    auto run = [&]() {
      int rc = 1;
      if (dtype == OMCHAT_F16) rc = launch_t<f16>(t, stream);
      else if (dtype == OMCHAT_BF16) rc = launch_t<bf16>(t, stream);
      ok = ok && rc == 0;
    };
    run();                                                                        // first launch of an instantiation (attribute set-up)
    float ms = 1e30f;
    for (int i = 0; i < 3 && ok; ++i) {
      if (flush) (void)hipMemsetAsync(flush, i, FLUSH_BYTES, stream);
      (void)hipEventRecord(e0, stream);
      run();
      (void)hipEventRecord(e1, stream);
      if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
      float one = 0.f;
      (void)hipEventElapsedTime(&one, e0, e1);
      ms = one < ms ? one : ms;
    }
    if (!ok) continue;
    if (getenv("OMCHAT_TUNE_LOG")) fprintf(stderr, "[omchat tune] M=%d N=%d K=%d epi=%d tile %d: %.1f us\n", a.M, a.N, a.K, a.epi, c, ms * 1e3f);
    if (ms < best_ms) { best_ms = ms; best = c; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(scratch);
  if (flush) (void)hipFree(flush);
  return g_tuned[key] = best;
}
size_t gemm_sk_ws_bytes() { return (size_t)SK_MAX_WG * SK_SLAB_BYTES + 4096; }

template <typename T>
static int launch_f8_t(const GemmArgs& a, hipStream_t stream) {
  switch (a.epi) {
    case EPI_NONE: return launch_cfg8_f8<T, EPI_NONE>(a, stream);
    case EPI_GELU: return launch_cfg8_f8<T, EPI_GELU>(a, stream);
    case EPI_LS_RESID: return launch_cfg8_f8<T, EPI_LS_RESID>(a, stream);
    case EPI_RESID: return launch_cfg8_f8<T, EPI_RESID>(a, stream);
    case EPI_SWIGLU: return launch_cfg8_f8<T, EPI_SWIGLU>(a, stream);
  }
  omchat_set_error("launch_gemm: bad epilogue");
  return 1;
}

int launch_gemm(int dtype, const GemmArgs& a, hipStream_t stream) {
  OM_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "empty problem");
  if (a.f8) {
    OM_CHECK(a.K % 128 == 0 && a.lda % 16 == 0 && a.ldw % 16 == 0 && a.a_scale && a.w_scale, "fp8 GEMM: K % 128, lda / ldw % 16 bytes, both scale vectors");
    OM_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && a.ldc % 4 == 0 && ((uintptr_t)a.C & 7) == 0 &&
             (!a.resid || (a.ldr % 4 == 0 && ((uintptr_t)a.resid & 7) == 0)), "fp8 GEMM: A / W 16-byte, C / residual 8-byte aligned with row strides % 4");
    OM_CHECK(a.epi != EPI_SWIGLU || a.N % 32 == 0, "SwiGLU epilogue needs N % 32 == 0");
    OM_CHECK(a.N % 4 == 0 && ((uintptr_t)a.bias & 7) == 0 && ((uintptr_t)a.ls & 7) == 0 && ((uintptr_t)a.w_scale & 15) == 0,
             "fp8 GEMM: N % 4, bias / layer-scale 8-byte and w_scale 16-byte aligned");
    if (dtype == OMCHAT_F16) return launch_f8_t<f16>(a, stream);
    if (dtype == OMCHAT_BF16) return launch_f8_t<bf16>(a, stream);
    omchat_set_error("launch_gemm: bad dtype");
    return 1;
  }
  OM_CHECK(a.K % 64 == 0, "K must be a multiple of 64");
  OM_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, "lda/ldw must be multiples of 8 elements (16-byte rows)");
  // the epilogue owns four consecutive columns of a row per lane: 8-byte stores to C and 8-byte loads of the residual
  OM_CHECK(a.ldc % 4 == 0 && (!a.resid || a.ldr % 4 == 0) && ((uintptr_t)a.C & 7) == 0 && ((uintptr_t)a.resid & 7) == 0,
           "C / residual: 8-byte aligned base, row stride a multiple of 4 elements (8-byte epilogue accesses)");
  OM_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "A/W must be 16-byte aligned");
  OM_CHECK(a.epi != EPI_SWIGLU || a.N % 32 == 0, "SwiGLU epilogue needs N % 32 == 0");
  OM_CHECK((a.epi != EPI_LS_RESID && a.epi != EPI_LS_RESID_STATS) || a.ls, "layer-scale epilogue needs ls");
  OM_CHECK((a.epi != EPI_LS_RESID_STATS && a.epi != EPI_NONE_STATS) || (a.stats && a.stats_ld >= a.M), "statistics epilogue: stats [slots][stats_ld] (slot-major), stats_ld >= M");
  OM_CHECK(!a.rs_stats || (a.rs_nslots >= 1 && a.rs_ld >= a.M && a.rs_dim > 0 && !a.row_scale), "row statistics: slot-major [rs_nslots][rs_ld >= M], rs_dim > 0, not together with row_scale");
  OM_CHECK(a.N % 4 == 0 && ((uintptr_t)a.bias & 7) == 0 && ((uintptr_t)a.ls & 7) == 0,
           "N must be a multiple of 4 and bias / layer-scale 8-byte aligned (the epilogue owns four consecutive columns per lane)");
  OM_CHECK(dtype == OMCHAT_F16 || dtype == OMCHAT_BF16, "bad dtype");
  GemmArgs b = a;
  if (a.epi == EPI_F32OUT) {      // fp32 output (numerics option of the tensor-parallel path): 256^2 staggered kernel, 128^2 for small problems; no tuning
    OM_CHECK(a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0 && !a.bias && !a.resid, "fp32-output GEMM: C 16-byte aligned, ldc % 4 == 0, no bias / residual");
    b.force_tile = (long)cdiv(a.M, 256) * cdiv(a.N, 256) < 64 ? 1 : 2;
    b.stream_k = -1;
  } else
  if (!a.force_tile && g_autotune && a.stream_k <= 0) b.force_tile = tuned_tile(dtype, a, stream);
  if (dtype == OMCHAT_F16) return launch_t<f16>(b, stream);
  return launch_t<bf16>(b, stream);
}
