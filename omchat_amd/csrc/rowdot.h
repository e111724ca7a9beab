// Packed dot product of 8 16-bit weights with 8 activations (v_dot2, fp32 accumulate), shared by the whole-row GEMV kernels (gemv.hip)
// and the fused decode launch (fused_decode.hip): the same instruction sequence gives the same bits in both.
#pragma once
#include "common.h"

namespace {

typedef unsigned int rw_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ float rw_dot8(rw_u32x4 w, rw_u32x4 x, float acc);
template <> __device__ __forceinline__ float rw_dot8<bf16>(rw_u32x4 w, rw_u32x4 x, float acc) {
  typedef bf16 v2 __attribute__((ext_vector_type(2)));
  // explicit components: bit_cast of a loop-indexed vector element was observed to read element 0 four times
  const unsigned w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w, x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, w0), __builtin_bit_cast(v2, x0), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, w1), __builtin_bit_cast(v2, x1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, w2), __builtin_bit_cast(v2, x2), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, w3), __builtin_bit_cast(v2, x3), acc, false);
  return acc;
}
template <> __device__ __forceinline__ float rw_dot8<f16>(rw_u32x4 w, rw_u32x4 x, float acc) {
  typedef f16 v2 __attribute__((ext_vector_type(2)));
  const unsigned w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w, x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, w0), __builtin_bit_cast(v2, x0), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, w1), __builtin_bit_cast(v2, x1), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, w2), __builtin_bit_cast(v2, x2), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, w3), __builtin_bit_cast(v2, x3), acc, false);
  return acc;
}

}  // namespace
