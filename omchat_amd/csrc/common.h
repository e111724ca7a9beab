// Common device/host helpers for the OmChat gfx950 kernels.  MI355X (CDNA4) only: wave = 64 lanes,
// MFMA 16x16x32 f16/bf16, 160 KiB LDS per CU.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

enum { OMCHAT_F16 = 0, OMCHAT_BF16 = 1 };

template <typename T> struct V8;
template <> struct V8<f16> { typedef f16x8 type; typedef f16x4 half_type; };
template <> struct V8<bf16> { typedef bf16x8 type; typedef bf16x4 half_type; };

// D[16x16] += A[16x32] * B[32x16].  Lane l: A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15],
// D[row 4(l>>4)+r][col l&15]  (cdna_hip_programming.md §3).
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ float tof(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T fromf(float x) { return (T)x; }
// round a float through the storage type (the reference rounds after every op in fp16)
template <typename T> __device__ __forceinline__ float rnd(float x) { return (float)((T)x); }

// Exact (erf) GELU, x * Phi(x), for the GEMM epilogues.  ocml's erff costs ~35 VALU per element (both branches under
// divergence), which made the epilogue of a 256x256 ViT fc1 tile ~20 % of the tile's time (VALU does not overlap MFMA on
// a SIMD).  Phi is evaluated through erfc(z) = t * exp(-z^2 + P9(t)), t = 1 / (1 + z/2)  (Numerical Recipes erfcc,
// fractional error < 1.2e-7 everywhere, so the negative tail keeps its RELATIVE accuracy): ~20 VALU, 2 transcendental.
// Measured against float64 erf: |x| < 6 relative error <= 4.4e-6, three orders below the 16-bit rounding of the output
// (tests/test_host_cpu.py::test_fast_gelu_formula_accuracy restates it in numpy float32).
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.0f));
  float p = 0.17087277f;
  p = fmaf(p, t, -0.82215223f);
  p = fmaf(p, t, 1.48851587f);
  p = fmaf(p, t, -1.13520398f);
  p = fmaf(p, t, 0.27886807f);
  p = fmaf(p, t, -0.18628806f);
  p = fmaf(p, t, 0.09678418f);
  p = fmaf(p, t, 0.37409196f);
  p = fmaf(p, t, 1.00002368f);
  p = fmaf(p, t, -1.26551223f);
  const float e = 0.5f * t * __expf(fmaf(-z, z, p));          // erfc(z) / 2
  return x * (x >= 0.f ? 1.0f - e : e);
}
// Two elements per instruction (v_pk_fma_f32 / v_pk_mul_f32): the same formula on a register pair, for the GEMM epilogue where no MFMA of
// this wave competes for issue slots (the polynomial is 9 of the ~20 VALU of an element; the two transcendentals stay scalar).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 z = ax * 0.70710678118654752440f;
  const f32x2 d = __builtin_elementwise_fma((f32x2){0.5f, 0.5f}, z, (f32x2){1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = {0.17087277f, 0.17087277f};
  p = __builtin_elementwise_fma(p, t, (f32x2){-0.82215223f, -0.82215223f});
  p = __builtin_elementwise_fma(p, t, (f32x2){1.48851587f, 1.48851587f});
  p = __builtin_elementwise_fma(p, t, (f32x2){-1.13520398f, -1.13520398f});
  p = __builtin_elementwise_fma(p, t, (f32x2){0.27886807f, 0.27886807f});
  p = __builtin_elementwise_fma(p, t, (f32x2){-0.18628806f, -0.18628806f});
  p = __builtin_elementwise_fma(p, t, (f32x2){0.09678418f, 0.09678418f});
  p = __builtin_elementwise_fma(p, t, (f32x2){0.37409196f, 0.37409196f});
  p = __builtin_elementwise_fma(p, t, (f32x2){1.00002368f, 1.00002368f});
  p = __builtin_elementwise_fma(p, t, (f32x2){-1.26551223f, -1.26551223f});
  const f32x2 a = __builtin_elementwise_fma(-z, z, p);
  const f32x2 e = (t * 0.5f) * (f32x2){__expf(a[0]), __expf(a[1])};          // erfc(z) / 2
  const f32x2 one = {1.0f, 1.0f};
  const f32x2 phi = {x[0] >= 0.f ? one[0] - e[0] : e[0], x[1] >= 0.f ? one[1] - e[1] : e[1]};
  return x * phi;
}
// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU): the result is rounded to 16 bit next
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-byte global load / store of 8 16-bit elements
template <typename T> __device__ __forceinline__ typename V8<T>::type ld8(const T* p) {
  return *reinterpret_cast<const typename V8<T>::type*>(p);
}
// the same load for data that is read ONCE per launch and never again before it is evicted (the KV cache rows of a decode step):
// non-temporal, so that the stream does not displace anything in L2 / the Infinity Cache on its way through.  Measured on the batched
// decode attention (b = 32, 242 MB of K / V per launch): 52.3 -> 45.5 us per launch (profiles/r04_o...); -DOMCHAT_KV_NT=0 builds the A/B twin
#ifndef OMCHAT_KV_NT
#define OMCHAT_KV_NT 1
#endif
template <typename T> __device__ __forceinline__ typename V8<T>::type ld8s(const T* p) {
#if OMCHAT_KV_NT
  return __builtin_nontemporal_load(reinterpret_cast<const typename V8<T>::type*>(p));
#else
  return *reinterpret_cast<const typename V8<T>::type*>(p);
#endif
}
template <typename T> __device__ __forceinline__ void st8(T* p, typename V8<T>::type v) {
  *reinterpret_cast<typename V8<T>::type*>(p) = v;
}

// ---------------------------------------------------------------------------------------------------------
// Packed ("MFMA fragment order") operand layouts of the batched decode GEMV (gemv.hip): every wave load is 1 KiB contiguous.
//   x  [K/64 chunks][2 halves][NB][64 lanes][8]   lane l of (chunk, half, nb) holds batch row 16*nb + (l & 15), k = 64*chunk + 32*half + 8*(l >> 4) + j
//   W  [N/16 tiles][K/64 chunks][2 halves][64 lanes][8]   lane l holds weight row 16*tile + (l & 15), the same k
// NB = 1 for up to 16 batch rows, 2 for up to 32.  Element index of x[row][k]:
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ size_t packed_x_index(int row, int k, int NB) {
  return ((((size_t)(k >> 6) * 2 + ((k >> 5) & 1)) * NB + (row >> 4)) * 64 + (row & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7);
}
__host__ __device__ __forceinline__ size_t packed_w_index(int row, int k, int K) {
  return ((((size_t)(row >> 4) * (K >> 6) + (k >> 6)) * 2 + ((k >> 5) & 1)) * 64 + (row & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
void omchat_set_error(const std::string& s);
#define OM_CHECK(cond, msg)                                                         \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      omchat_set_error(std::string(__func__) + ": " + (msg));                       \
      return 1;                                                                     \
    }                                                                               \
  } while (0)
#define OM_HIP(call)                                                                \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      omchat_set_error(std::string(__func__) + ": " #call " -> " + hipGetErrorString(e_)); \
      return 2;                                                                     \
    }                                                                               \
  } while (0)
#define OM_LAUNCH_CHECK() OM_HIP(hipGetLastError())

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// compute units of the current device (one process drives one GPU: looked up once; 256 if the query fails)
static inline int device_cus() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
// hipFuncSetAttribute is a per-DEVICE setting: a per-call-site "done" flag is kept per device id, so a process that drives a second GPU (the
// normal deployment is one process per GPU) still raises the dynamic-LDS limit there before its first launch.
struct PerDeviceOnce {
  unsigned long long done = 0;
  bool first() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return false;
    done |= bit;
    return true;
  }
};
