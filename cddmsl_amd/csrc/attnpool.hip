// CLIP attention-pool core for gfx950: query-0-only multi-head attention over the 50 tokens of a region.
//
// Replaces F.multi_head_attention_forward as called by AttentionPool2d.forward
// (detectron2/modeling/backbone/clip_backbone.py:83-107): the reference projects q/k/v for all 50
// tokens and keeps only output token 0; here only token 0's query is formed, k/v come from one fused
// [K*50, 2C] projection GEMM (gemm_conv.hip) and this kernel does the per-head softmax(q0 k^T / sqrt(d)) v
// in-wavefront.  Head dim is 64 = one lane per channel of a head; tokens <= 64 = one lane per token.
//
//   fwd: q0 [K][C], kv [K][T][2C] (k | v)  ->  o [K][C] (T dtype), p [K][H][T] f32 (saved)
//   bwd: dO [K][C] -> dq0 [K][C], dkv [K][T][2C]
#include "common.h"

namespace {

template <typename T> struct IO;
template <> struct IO<__bf16> {
  static constexpr int ES = 2;
  __device__ static __forceinline__ float ld(const char* p, long i) { return bf2f(((const unsigned short*)p)[i]); }
  __device__ static __forceinline__ void st(char* p, long i, float v) { ((unsigned short*)p)[i] = f2bf(v); }
};
template <> struct IO<float> {
  static constexpr int ES = 4;
  __device__ static __forceinline__ float ld(const char* p, long i) { return ((const float*)p)[i]; }
  __device__ static __forceinline__ void st(char* p, long i, float v) { ((float*)p)[i] = v; }
};

// grid = K regions; block = 256 (4 waves); wave w handles heads w, w+4, ...
template <typename T>
__global__ void k_attnpool_fwd(const char* q0, const char* kv, char* o, float* p, int Tn, int H, float scale) {
  const int k = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int C = H * 64;
  __shared__ float sk[4][64][65];   // per-wave k tile [token][d] (+1 pad)
  for (int h = wv; h < H; h += 4) {
    float q = IO<T>::ld(q0, (long)k * C + h * 64 + lane) * scale;   // lane = d
    // stage k[t][h][:] coalesced (lane = d) into LDS, then lane = token reads its row
    for (int t = 0; t < Tn; ++t) sk[wv][t][lane] = IO<T>::ld(kv, ((long)k * Tn + t) * 2 * C + h * 64 + lane);
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
    const int tl = lane < Tn ? lane : Tn - 1;
    for (int d = 0; d < 64; ++d) s += __shfl(q, d, 64) * sk[wv][tl][d];   // shuffles stay outside any lane-divergent branch
    if (lane >= Tn) s = -INFINITY;
    float mx = wave_max(s);
    float e = lane < Tn ? expf(s - mx) : 0.f;
    float den = wave_sum(e);
    float pr = e / den;
    if (lane < Tn) p[((long)k * H + h) * Tn + lane] = pr;
    float acc = 0.f;   // lane = d
    for (int t = 0; t < Tn; ++t)
      acc += __shfl(pr, t, 64) * IO<T>::ld(kv, ((long)k * Tn + t) * 2 * C + C + h * 64 + lane);
    IO<T>::st(o, (long)k * C + h * 64 + lane, acc);
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T>
__global__ void k_attnpool_bwd(const char* dO, const char* q0, const char* kv, const float* p, char* dq0, char* dkv, int Tn,
                               int H, float scale) {
  const int k = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int C = H * 64;
  __shared__ float sv[4][64][65];
  for (int h = wv; h < H; h += 4) {
    float go = IO<T>::ld(dO, (long)k * C + h * 64 + lane);           // lane = d
    float q = IO<T>::ld(q0, (long)k * C + h * 64 + lane);
    float pr = lane < Tn ? p[((long)k * H + h) * Tn + lane] : 0.f;    // lane = token
    for (int t = 0; t < Tn; ++t) sv[wv][t][lane] = IO<T>::ld(kv, ((long)k * Tn + t) * 2 * C + C + h * 64 + lane);
    __builtin_amdgcn_wave_barrier();
    float dp = 0.f;
    for (int d = 0; d < 64; ++d) {
      float g = __shfl(go, d, 64);
      if (lane < Tn) dp += g * sv[wv][lane][d];
    }
    float sumpd = wave_sum(pr * dp);
    float ds = pr * (dp - sumpd);                                     // lane = token, d(score)
    float dq = 0.f;
    for (int t = 0; t < Tn; ++t) {
      float dst = __shfl(ds, t, 64), pt = __shfl(pr, t, 64);
      long base = ((long)k * Tn + t) * 2 * C + h * 64 + lane;
      float kk = IO<T>::ld(kv, base);
      dq += dst * kk;
      IO<T>::st(dkv, base, dst * q * scale);                          // dk[t][d]
      IO<T>::st(dkv, base + C, pt * go);                              // dv[t][d]
    }
    IO<T>::st(dq0, (long)k * C + h * 64 + lane, dq * scale);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int cddmsl_attnpool_core_fwd(const void* q0, const void* kv, void* o, float* p, int K, int T, int H, float scale,
                                        int dtype, void* stream) {
  if (K < 0 || T <= 0 || T > 64 || H <= 0 || (dtype != 0 && dtype != 1)) return CDDMSL_ERR_ARG;
  if (K == 0) return CDDMSL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) k_attnpool_fwd<__bf16><<<dim3(K), dim3(256), 0, st>>>((const char*)q0, (const char*)kv, (char*)o, p, T, H, scale);
  else k_attnpool_fwd<float><<<dim3(K), dim3(256), 0, st>>>((const char*)q0, (const char*)kv, (char*)o, p, T, H, scale);
  return launch_status();
}

extern "C" int cddmsl_attnpool_core_bwd(const void* dO, const void* q0, const void* kv, const float* p, void* dq0, void* dkv,
                                        int K, int T, int H, float scale, int dtype, void* stream) {
  if (K < 0 || T <= 0 || T > 64 || H <= 0 || (dtype != 0 && dtype != 1)) return CDDMSL_ERR_ARG;
  if (K == 0) return CDDMSL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) k_attnpool_bwd<__bf16><<<dim3(K), dim3(256), 0, st>>>((const char*)dO, (const char*)q0, (const char*)kv, p, (char*)dq0, (char*)dkv, T, H, scale);
  else k_attnpool_bwd<float><<<dim3(K), dim3(256), 0, st>>>((const char*)dO, (const char*)q0, (const char*)kv, p, (char*)dq0, (char*)dkv, T, H, scale);
  return launch_status();
}
