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

namespace {

struct GemmP {
  const void* A; const void* W; void* C; const void* bias; const void* ls; const void* resid;
  int lda, ldw, ldc, ldr, M, N, K;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmP p) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MR = WTM / 16, NR = WTN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_ROUNDS = BM * 8 / NT, B_ROUNDS = BN * 8 / NT;
  static_assert(A_ROUNDS * NT == BM * 8 && B_ROUNDS * NT == BN * 8, "tile/threads mismatch");
  typedef typename V8<T>::type frag_t;
  extern __shared__ __attribute__((aligned(256))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile id: bijective XCD remap (blocks b and b+8 share an XCD), M fastest inside an XCD's chunk
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int m0 = (wgid % tiles_m) * BM, n0 = (wgid / tiles_m) * BN;

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
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * 64), (lptr_t)(base + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i)
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
  stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nk) stage((t + 1) & 1, t + 1);
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
        for (int j = 0; j < NR; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
    }
  }

  // ---- epilogue.  acc[i][j][r] = C[row0 + i*16 + 4*fg + r][col0 + j*16 + fr]
  const int row0 = m0 + wm * WTM + fg * 4;
  const int col0 = n0 + wn * WTN + fr;
  const T* __restrict__ bias = (const T*)p.bias;
  T* C = (T*)p.C;
  if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
    for (int j = 0; j < NR; j += 2) {
      const int colg = col0 + j * 16;            // gate column in the fused layout (up = +16)
      if (colg + 16 < p.N) {
        const int oc = ((n0 + wn * WTN + j * 16) >> 1) + fr;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + i * 16 + r;
            if (row < p.M) {
              const float g = rnd<T>(acc[i][j][r]), u = rnd<T>(acc[i][j + 1][r]);
              C[(size_t)row * p.ldc + oc] = fromf<T>(rnd<T>(silu(g)) * u);
            }
          }
      }
    }
  } else {
    const T* __restrict__ ls = (const T*)p.ls;
    const T* R = (const T*)p.resid;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int col = col0 + j * 16;
      if (col < p.N) {
        const float bv = bias ? tof(bias[col]) : 0.f;
        float lsv = 1.f;
        if constexpr (EPI == EPI_LS_RESID) lsv = tof(ls[col]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + i * 16 + r;
            if (row < p.M) {
              float v = rnd<T>(acc[i][j][r] + bv);
              if constexpr (EPI == EPI_GELU) v = gelu_erf(v);
              if constexpr (EPI == EPI_LS_RESID) v = (R ? tof(R[(size_t)row * p.ldr + col]) : 0.f) + rnd<T>(v * lsv);
              if constexpr (EPI == EPI_RESID) v = (R ? tof(R[(size_t)row * p.ldr + col]) : 0.f) + v;
              C[(size_t)row * p.ldc + col] = fromf<T>(v);
            }
          }
      }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  constexpr int LDS = 2 * (BM + BN) * 128;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    OM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  GemmP p{a.A, a.W, a.C, a.bias, a.ls, a.resid, a.lda, a.ldw, a.ldc, a.ldr, a.M, a.N, a.K};
  const int grid = cdiv(a.M, BM) * cdiv(a.N, BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), LDS, stream, p);
  OM_LAUNCH_CHECK();
  return 0;
}

template <typename T, int EPI>
int launch_epi(const GemmArgs& a, hipStream_t stream) {
  int tile = a.force_tile;
  if (tile == 0) {
    const long big = (long)cdiv(a.M, 256) * cdiv(a.N, 256);
    tile = (big >= 128) ? 2 : 1;      // measured (r01): the 256^2 kernel at half-filled CUs still beats the 128^2 one
  }
  if (tile == 2) return launch_cfg<T, 256, 256, 2, 4, EPI>(a, stream);
  return launch_cfg<T, 128, 128, 2, 2, EPI>(a, stream);
}

template <typename T>
int launch_t(const GemmArgs& a, hipStream_t stream) {
  switch (a.epi) {
    case EPI_NONE: return launch_epi<T, EPI_NONE>(a, stream);
    case EPI_GELU: return launch_epi<T, EPI_GELU>(a, stream);
    case EPI_LS_RESID: return launch_epi<T, EPI_LS_RESID>(a, stream);
    case EPI_RESID: return launch_epi<T, EPI_RESID>(a, stream);
    case EPI_SWIGLU: return launch_epi<T, EPI_SWIGLU>(a, stream);
  }
  omchat_set_error("launch_gemm: bad epilogue");
  return 1;
}

}  // namespace

int launch_gemm(int dtype, const GemmArgs& a, hipStream_t stream) {
  OM_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "empty problem");
  OM_CHECK(a.K % 64 == 0, "K must be a multiple of 64");
  OM_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, "lda/ldw must be multiples of 8 elements (16-byte rows)");
  OM_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "A/W must be 16-byte aligned");
  OM_CHECK(a.epi != EPI_SWIGLU || a.N % 32 == 0, "SwiGLU epilogue needs N % 32 == 0");
  OM_CHECK(a.epi != EPI_LS_RESID || a.ls, "layer-scale epilogue needs ls");
  if (dtype == OMCHAT_F16) return launch_t<f16>(a, stream);
  if (dtype == OMCHAT_BF16) return launch_t<bf16>(a, stream);
  omchat_set_error("launch_gemm: bad dtype");
  return 1;
}
