// Micro-benchmark (not product): do MFMA and VALU overlap on a gfx950 SIMD, in one wave and across waves?  Issue cost of
// v_exp_f32 / v_fma_f32 / v_pk_mul_f32 / v_cvt_pk / v_max.   hipcc --offload-arch=gfx950 -O3 tools/tune_pipes.hip -o /tmp/tune_pipes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int waves_mfma) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4] = {};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + threadIdx.x * 1e-3f + j;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < waves_mfma);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= waves_mfma);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
      // same wave: 1 MFMA then 4 VALU, x4 (16 VALU per 4 MFMA)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %3, %3, %3, %2"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      }
    } else {
      if (do_mfma) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %1, %1, %1, %0\n v_fma_f32 %3, %3, %3, %2"
                       : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      }
    }
  }
  float r = 0;
  for (int u = 0; u < 4; ++u) r += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  for (int j = 0; j < 8; ++j) r += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// VALU op cost: OP selects the instruction, 16 independent-ish ops per iteration
template <int OP>
__global__ __launch_bounds__(512) void kv(float* out, int iters) {
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 0.5f + threadIdx.x * 1e-4f + j * 1e-2f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 10) asm volatile("v_max_f32 %0, %0, %0\n v_max_f32 %1, %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 11) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 12) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0\n v_cvt_pk_bf16_f32 %1, %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 13) asm volatile("v_max3_f32 %0, %0, %0, %0\n v_max3_f32 %1, %1, %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 14) asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 15) asm volatile("v_pk_mul_f32 %0, %0, %0 op_sel_hi:[0,1]" : "+v"(*(double*)&v[j]));
      if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&v[j]));
      if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %0" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 4) asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %0" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %1\n v_max3_f32 %1, %1, %0, %0" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&v[j]));
      if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(double*)&v[j]));
      if (OP == 8) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %0" : "+v"(v[j]), "+v"(v[j + 1]));
      if (OP == 9) asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1" : "+v"(v[j]), "+v"(v[j + 1]));
    }
  }
  float r = 0;
  for (int j = 0; j < 16; ++j) r += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4 * 8);
  const int iters = 20000;
  // clock estimate: MFMA-only, 1 wave per SIMD: 4 MFMA x 16 cycles per iteration
  auto cyc = [&](float ms, double per_iter_units) { return ms * 1e-3 * 2.4e9 / iters / per_iter_units; };
  for (int nw : {4, 8}) {   // waves per CU (block) = 1 or 2 per SIMD
    float t0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(256), dim3(nw * 64), 0, 0, out, iters, 0); });
    float t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(256), dim3(nw * 64), 0, 0, out, iters, 0); });
    float t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(256), dim3(nw * 64), 0, 0, out, iters, 0); });
    printf("waves/CU %d: mfma-only %.3f ms (%.1f cyc/MFMA/wave) valu-only %.3f ms (%.1f cyc/VALU/wave) same-wave-interleaved %.3f ms\n", nw, t0,
           cyc(t0, 4), t1, cyc(t1, 16), t2);
  }
  // cross-wave: 8 waves per CU, first 4 MFMA-only (one per SIMD?) -- wave -> SIMD mapping is round-robin (wave & 3)
  float t3 = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, iters, 4); });
  printf("cross-wave (4 MFMA waves + 4 VALU waves per CU): %.3f ms\n", t3);
  const char* names[] = {"v_exp_f32", "v_fma_f32", "v_pk_mul_f32", "cvt_pk(dep)", "v_max(dep)", "v_max3(dep)", "v_pk_fma_f32", "v_pk_add_f32", "v_add(dep)", "v_exp_f16",
                         "v_max_f32", "v_add_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_add_u32", "v_pk_mul opsel"};
  for (int nw : {4, 8, 16}) {
    float tt[16];
#define RUN(i) tt[i] = timeit([&] { hipLaunchKernelGGL(kv<i>, dim3(256), dim3(nw * 64), 0, 0, out, iters); });
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
    printf("-- %d waves per SIMD: SIMD cycles (2.4 GHz) per instruction\n", nw / 4);
    for (int i = 0; i < 16; ++i) {
      const int n = (i == 2 || i == 6 || i == 7 || i == 15) ? 8 : 16;
      printf("%-20s %.3f ms  %.2f cyc/instr\n", names[i], tt[i], cyc(tt[i], n) / (nw / 4));
    }
  }
  return 0;
}
