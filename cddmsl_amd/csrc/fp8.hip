// OCP e4m3 support kernels of the fp8 configuration (BASELINE.json configs[4]: "fp8 MFMA conv + fp8 region x text-embedding GEMM"):
//   quantize        x (bf16 or f32) -> e4m3 with a per-tensor scale held ON THE DEVICE (delayed scaling: the scale of step t
//                   comes from max|x| of step t-1, which this kernel records on the way -- no host round trip, one pass)
//   fp8_dot_nt      C[R][N] f32 = alpha * A[R][K] . B[N][K]^T  for a skinny N (<= 32): the region x text-embedding contraction of
//                   FastRCNNOutputLayers.forward (detectron2/modeling/roi_heads/fast_rcnn.py:546-572) on e4m3 operands,
//                   v_mfma_scale_f32_32x32x64_f8f6f4 with f32 accumulation, operands straight from global memory (each byte is
//                   read once: nothing to stage)
// The convolutions themselves run on k_conv_fwd256<fp8e4> (gemm_conv.hip).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
  // v_cvt_pk_fp8_f32: two f32 -> two OCP e4m3 bytes (gfx950), into the low / high half of the destination word.  Inputs are
  // clamped to the format's finite range first (e4m3fn has no infinity: 448 is its largest magnitude).
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

template <bool F32IN>
__global__ __launch_bounds__(256) void k_quantize_fp8(const char* x, char* y, const float* scale, unsigned* amax_bits, long n8) {
  const float s = scale ? scale[0] : 1.f;
  unsigned m = 0u;                                    // max |x| as a bit pattern: Inf / NaN stay visible (common.h absmax_bits)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    if (F32IN) {
      const f32x4 a = ((const f32x4*)x)[2 * i], b = ((const f32x4*)x)[2 * i + 1];
      f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    } else {
      const u32x4 v = ((const u32x4*)x)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f(v[j] & 0xffff); f[2 * j + 1] = bf2f(v[j] >> 16); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { m = absmax_bits(m, f[j]); f[j] *= s; }
    u32x2 o = {pack4_e4m3(f[0], f[1], f[2], f[3]), pack4_e4m3(f[4], f[5], f[6], f[7])};
    ((u32x2*)y)[i] = o;
  }
  if (amax_bits) {                                    // one atomic per block, spread over the slot's 64 words
    __shared__ unsigned sm[4];
    m = wave_max_u(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
      atomicMax(amax_bits + (blockIdx.x & 63), a > b ? a : b);   // non-negative floats order like their bit patterns
    }
  }
}

// one wave per 32 rows of A; lane (r, h) holds row r's bytes [64 s + 32 h, +32) of step s -- the same bytes of B's row r
__global__ __launch_bounds__(64) void k_fp8_dot_nt(const char* a, const char* b, float* c, const float* alpha, int R, int N, int K, int ldc) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const long row = (long)blockIdx.x * 32 + r;
  const bool va = row < R, vb = r < N;
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = 0.f;
  const char* pa = a + row * K + 32 * h;
  const char* pb = b + (long)r * K + 32 * h;
  for (int s = 0; s < K; s += 64) {
    i32x8 fa = {0, 0, 0, 0, 0, 0, 0, 0}, fb = {0, 0, 0, 0, 0, 0, 0, 0};
    if (va) {
      const u32x4 a0 = *(const u32x4*)(pa + s), a1 = *(const u32x4*)(pa + s + 16);
      fa = i32x8{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    }
    if (vb) {
      const u32x4 b0 = *(const u32x4*)(pb + s), b1 = *(const u32x4*)(pb + s + 16);
      fb = i32x8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    }
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  // D[i][j] = sum_k A'[i][k] B'[k][j] with A' = the A rows, B' = the B rows transposed: column j = lane & 31 is B's row (the class),
  // register g walks A's rows (g & 3) + 8 (g >> 2) + 4 h
  const float al = alpha ? alpha[0] : 1.f;
  if (r < N) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const long m = (long)blockIdx.x * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
      if (m < R) c[m * ldc + r] = al * acc[g];
    }
  }
}

}  // namespace

extern "C" int cddmsl_quantize_fp8(const void* x, void* y, const float* scale, float* amax, long numel, int src_dtype, void* stream) {
  if (numel < 0 || (numel & 7) || (src_dtype != 0 && src_dtype != 1)) return CDDMSL_ERR_ARG;
  if (numel == 0) return CDDMSL_OK;
  const long n8 = numel / 8;
  long blocks = (n8 + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;            // 8 resident blocks per CU, grid-stride
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == 0) k_quantize_fp8<false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const char*)x, (char*)y, scale, (unsigned*)amax, n8);
  else k_quantize_fp8<true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>((const char*)x, (char*)y, scale, (unsigned*)amax, n8);
  return launch_status();
}

extern "C" int cddmsl_fp8_dot_nt(const void* a, const void* b, float* c, const float* alpha, int R, int N, int K, int ldc, void* stream) {
  if (R < 0 || N <= 0 || N > 32 || K <= 0 || (K % 64) || ldc < N) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  k_fp8_dot_nt<<<dim3((unsigned)((R + 31) / 32)), dim3(64), 0, (hipStream_t)stream>>>((const char*)a, (const char*)b, c, alpha, R, N, K, ldc);
  return launch_status();
}
