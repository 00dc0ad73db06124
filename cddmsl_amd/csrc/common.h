// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the CDDMSL hot path.
// Wavefront = 64 lanes everywhere; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CDDMSL_OK 0
#define CDDMSL_ERR_ARG 1
#define CDDMSL_ERR_LAUNCH 2

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// f32 -> bf16 bits, round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32).
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
  return __builtin_bit_cast(float, ((unsigned int)u) << 16);
}
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 (round to nearest even, the same conversion f2bf performs) instead of two converts + shift + or
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_));
}

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

// running max of |x| on the BIT PATTERN (sign cleared): orders like the magnitudes for finite values and puts Inf, then every NaN,
// above them -- so a non-finite input is not dropped the way fmaxf(m, fabsf(NaN)) drops it, and reaches the recorded maximum
__device__ __forceinline__ unsigned absmax_bits(unsigned m, float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x) & 0x7fffffffu;
  return u > m ? u : m;
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = w > v ? w : v; }
  return v;
}

// four f32 -> four OCP e4m3 bytes (v_cvt_pk_fp8_f32, gfx950), clamped to the format's finite range first (e4m3fn has no infinity)
__device__ __forceinline__ unsigned e4m3x4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CDDMSL_OK : CDDMSL_ERR_LAUNCH;
}
