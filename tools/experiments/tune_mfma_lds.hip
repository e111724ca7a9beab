// Micro-benchmark (not product): sustained MFMA rate of a CU when every wave also reads its operands from LDS at the ratio of the
// 256x256 GEMM (wave tile 128x64, K = 64 per step: 24 ds_read_b128 per 64 16x16x32 MFMAs), 2 waves per SIMD, for the two MFMA shapes:
//   16x16x32: 64 MFMAs / step (blocks the SIMD's vector issue 8 of 16 cycles each)      32x32x16: 32 MFMAs / step (8 of 32 cycles)
// Same flops and the same LDS bytes per step.  No global memory traffic in the loop.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_mfma_lds.hip -o /tmp/tune_mfma_lds && /tmp/tune_mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// SHAPE 0: 16x16x32, acc 8x4 tiles of 16x16; SHAPE 1: 32x32x16, acc 4x2 tiles of 32x32.  LDSR: ds_read_b128 per step (24 = the GEMM).
#ifndef RANDOM_DATA
#define RANDOM_DATA 2
#endif
template <int SHAPE, int LDSR>
__global__ __launch_bounds__(512) void k(float* out, int steps) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // pseudo-random bf16 pairs in (-1, 1): operand toggling (and so power / clocks) like real activations, not like constants
  for (int i = threadIdx.x; i < 32768; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // two bf16 values uniform in (-1, 1): random sign, mantissa AND exponent bits (RANDOM_DATA 2), or fixed exponent (RANDOM_DATA 1)
    unsigned lo, hi;
    if (RANDOM_DATA == 2) {
      const float f0 = ((h & 0xffffu) / 32768.0f - 1.0f), f1 = ((h >> 16) / 32768.0f - 1.0f);
      lo = __builtin_bit_cast(unsigned, f0) >> 16; hi = __builtin_bit_cast(unsigned, f1) >> 16;
    } else { lo = 0x3f00u | (h & 0x80ffu); hi = 0x3f00u | ((h >> 16) & 0x80ffu); }
    reinterpret_cast<unsigned*>(smem)[i] = RANDOM_DATA ? (lo | (hi << 16)) : 0x3c003c00u;
  }
  __syncthreads();
  // conflict-free 16-row x 64-B style read: row = lane & 15, chunk = (lane >> 4) ^ swizzle (as the GEMM)
  const int fr = lane & 15, fg = lane >> 4;
  const char* base = smem + (wave * 16 + fr) * 128 + ((fg ^ ((fr >> 1) & 7)) << 4);
  bf16x8 fa[8], fb[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
#pragma unroll
  for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const bf16x8*>(base + 16384 + i * 2048);
  float sink = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[8][4] = {};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // LDSR/2 reads per K half, round-robin into the A / B fragment registers (consumed by the MFMAs below)
#pragma unroll
        for (int r = 0; r < LDSR / 2; ++r) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + ((s * 2 + h) & 3) * 16 + (r % 12) * 2048 + (r / 12) * 64);
          if (r % 12 < 8) fa[r % 12] = v; else fb[r % 12 - 8] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[4][2] = {};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {      // 4 K-quarters of 16: per quarter 4 A frags (32 rows x 16 k) + 2 B frags = 6 reads of 16 B... x4 = 24
#pragma unroll
        for (int r = 0; r < LDSR / 4; ++r) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + ((s * 4 + h) & 3) * 16 + (r % 6) * 2048 + (r / 6) * 64);
          if (r % 6 < 4) fa[r % 6] = v; else fb[r % 6 - 4] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sink += acc[i][j][0] + acc[i][j][15];
  }
  if (sink == 123.456f) out[0] = sink;
}

template <int SHAPE, int LDSR>
void run(const char* name, float* out) {
  const int steps = 2000, blocks = 256;
  auto kern = k<SHAPE, LDSR>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 131072, 0, out, steps);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 131072, 0, out, steps);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 8 * steps * 128.0 * 64 * 64 * 2;
  printf("%-34s %8.1f us  %7.0f TF   LDS %5.2f TB/s\n", name, ms * 1e3, flops / ms / 1e9, (double)blocks * 8 * steps * LDSR * 1024.0 / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 64);
  run<0, 0>("16x16x32, no LDS reads", out);
  run<0, 12>("16x16x32, 12 ds_read_b128/step", out);
  run<0, 24>("16x16x32, 24 ds_read_b128/step", out);
  run<1, 0>("32x32x16, no LDS reads", out);
  run<1, 12>("32x32x16, 12 ds_read_b128/step", out);
  run<1, 24>("32x32x16, 24 ds_read_b128/step", out);
  if (hipGetLastError() != hipSuccess) printf("HIP ERROR\n");
  return 0;
}
