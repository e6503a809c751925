// Device helpers shared by the fused encoders (encoder_fused.hip: pull form; encoder_typed.hip: per-bond-type form).
#pragma once

#include "encoder_layout.h"

namespace impnn {
namespace enc {
namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Keeps a quad assembled from scalar results in one register tuple (no instruction is emitted): without
// it the compiler splits the following vector arithmetic back into scalar v_add / v_mul.
__device__ __forceinline__ f32x4 as_tuple(f32x4 v) {
  asm("" : "+v"(v));
  return v;
}
// Activations on whole accumulator quads, written as vector arithmetic so that the multiplies / adds
// around the quarter-rate v_exp_f32 / v_rcp_f32 become packed v_pk_{mul,add,fma}_f32 (two lanes of work
// per instruction).  SCALED: the accumulator carries the mode-1 scale kAcc (folded into the constant).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// A splat constant held in an SGPR pair: packed-f32 instructions cannot encode a 32-bit literal, so with a
// literal the compiler falls back to one scalar multiply per element.
__device__ __forceinline__ f32x4 splat_sgpr(float c) {
  f32x2 v = {c, c};
  asm("" : "+s"(v));
  return __builtin_shufflevector(v, v, 0, 1, 0, 1);
}
template <bool SCALED>
__device__ __forceinline__ f32x4 sigmoid4(f32x4 x) {
  const f32x4 a = x * splat_sgpr(SCALED ? -1.44269504088896f / kAcc : -1.44269504088896f);
  f32x4 e;
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(a[i]);
  const f32x4 d = as_tuple(as_tuple(e) + 1.0f);
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_rcpf(d[i]);
  return as_tuple(r);
}
template <bool SCALED>
__device__ __forceinline__ f32x4 tanh4(f32x4 x) {
  const f32x4 a = x * splat_sgpr(SCALED ? 2.88539008177793f / kAcc : 2.88539008177793f);
  f32x4 e;
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(a[i]);
  const f32x4 d = as_tuple(as_tuple(e) + 1.0f);
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_rcpf(d[i]);
  return 1.0f - 2.0f * as_tuple(r);
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

}  // namespace
}  // namespace enc
}  // namespace impnn
