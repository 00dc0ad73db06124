// fp32 loss-side kernels of the CDDMSL hot path (gfx950): row L2 normalisation, the region x text
// cosine-logit classifier, and the symmetric contrastive cross-entropy over the similarity matrix.
//
// Reference call sites (paths under detectron2/):
//   FastRCNNOutputLayers.forward  modeling/roi_heads/fast_rcnn.py:546-572   x/||x|| . w_c/||w_c|| / T, bg logit 0
//   v2l_contrastive               modeling/meta_arch/rcnn.py:308-317,458-468 rows / norm, S = A B^T,
//                                 0.5 * (CE(S, arange) + CE(S^T, arange))
// The S = A B^T contraction itself runs on the exact-f32 MFMA GEMM (gemm_conv.hip); these kernels are the
// row-wise pieces around it.  Everything is f32: T = 0.01 multiplies logits by 100, so no bf16 here.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, red[i]);
  return s;
}

// y = x / max(||x||, eps) per row; inv[r] = 1/max(||x||, eps).  eps = 0 -> plain division (rcnn.py:308).
__global__ void k_l2norm_fwd(const float* x, float* y, float* inv, int D, float eps) {
  __shared__ float red[4];
  long r = blockIdx.x;
  float ss = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) { float v = x[r * D + d]; ss += v * v; }
  ss = block_sum(ss, red);
  float iv = 1.0f / fmaxf(sqrtf(ss), eps);
  for (int d = threadIdx.x; d < D; d += blockDim.x) y[r * D + d] = x[r * D + d] * iv;
  if (threadIdx.x == 0) inv[r] = iv;
}
// dx = inv * (dy - y * (y . dy))
__global__ void k_l2norm_bwd(const float* dy, const float* y, const float* inv, float* dx, int D) {
  __shared__ float red[4];
  long r = blockIdx.x;
  float dot = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) dot += dy[r * D + d] * y[r * D + d];
  dot = block_sum(dot, red);
  float iv = inv[r];
  for (int d = threadIdx.x; d < D; d += blockDim.x) dx[r * D + d] = iv * (dy[r * D + d] - y[r * D + d] * dot);
}

// scores[r][c] = (xhat_r . wn_c) / T for c < Kc, scores[r][Kc] = 0 (zero background embedding).
// wn = row-normalised text embeddings [Kc][D].  One wave per row, 4 rows per block.
__global__ void k_cosine_logits_fwd(const float* x, const float* wn, float* scores, float* inv, long R, int D, int Kc,
                                    float invT, float eps) {
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + r * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
  ss = wave_sum(ss);
  float iv = 1.0f / fmaxf(sqrtf(ss), eps);
  for (int c = 0; c < Kc; ++c) {
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += (xr[d] * iv) * wn[(long)c * D + d];
    dot = wave_sum(dot);
    if (lane == 0) scores[r * (Kc + 1) + c] = dot * invT;
  }
  if (lane == 0) { scores[r * (Kc + 1) + Kc] = 0.f; inv[r] = iv; }
}
// dx_r (+)= inv * (g - xhat (xhat . g)),  g = sum_c ds[r][c] * invT * wn_c ; accumulate = add into dx
__global__ void k_cosine_logits_bwd(const float* ds, const float* x, const float* wn, const float* inv, float* dx, long R,
                                    int D, int Kc, float invT, int accumulate) {
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + r * D;
  float iv = inv[r];
  float dot = 0.f;
  if (D <= 64 * 16) {                     // the row's g values stay in registers (same sums, formed once instead of twice)
    float gv[16], xv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = lane + 64 * i;
      gv[i] = 0.f; xv[i] = 0.f;
      if (d < D) {
        float g = 0.f;
        for (int c = 0; c < Kc; ++c) g += ds[r * (Kc + 1) + c] * invT * wn[(long)c * D + d];
        gv[i] = g; xv[i] = xr[d] * iv;
        dot += g * xv[i];
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = lane + 64 * i;
      if (d < D) {
        float v = iv * (gv[i] - xv[i] * dot);
        dx[r * D + d] = accumulate ? dx[r * D + d] + v : v;
      }
    }
    return;
  }
  for (int d = lane; d < D; d += 64) {
    float g = 0.f;
    for (int c = 0; c < Kc; ++c) g += ds[r * (Kc + 1) + c] * invT * wn[(long)c * D + d];
    dot += g * (xr[d] * iv);
  }
  dot = wave_sum(dot);
  for (int d = lane; d < D; d += 64) {
    float g = 0.f;
    for (int c = 0; c < Kc; ++c) g += ds[r * (Kc + 1) + c] * invT * wn[(long)c * D + d];
    float v = iv * (g - (xr[d] * iv) * dot);
    dx[r * D + d] = accumulate ? dx[r * D + d] + v : v;
  }
}

// Symmetric CE over S [n][n]: rlse[i] = logsumexp_j S[i][j], clse[j] = logsumexp_i S[i][j].
__global__ void k_lse_rows(const float* S, float* rlse, int n, int ld, int transposed) {
  __shared__ float red[4];
  int i = blockIdx.x;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < n; j += blockDim.x) mx = fmaxf(mx, transposed ? S[(long)j * ld + i] : S[(long)i * ld + j]);
  mx = block_max(mx, red);
  float se = 0.f;
  for (int j = threadIdx.x; j < n; j += blockDim.x) se += expf((transposed ? S[(long)j * ld + i] : S[(long)i * ld + j]) - mx);
  se = block_sum(se, red);
  if (threadIdx.x == 0) rlse[i] = mx + logf(se);
}
// loss = 0.5 * ( mean_i(rlse[i] - S[i][i]) + mean_j(clse[j] - S[j][j]) ); dS = g/(2n) * (softmax_row + softmax_col - 2 I)
__global__ void k_contrastive_loss(const float* S, const float* rlse, const float* clse, float* loss, int n, int ld) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (rlse[i] - S[(long)i * ld + i]) + (clse[i] - S[(long)i * ld + i]);
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = 0.5f * s / (float)n;
}
__global__ void k_contrastive_grad(const float* S, const float* rlse, const float* clse, const float* gloss, float* dS, int n, int ld) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  int i = idx / n, j = idx % n;
  float s = S[(long)i * ld + j];
  float g = expf(s - rlse[i]) + expf(s - clse[j]) - (i == j ? 2.f : 0.f);
  dS[(long)i * n + j] = g * gloss[0] * 0.5f / (float)n;
}

}  // namespace

extern "C" int cddmsl_l2norm_fwd(const float* x, float* y, float* inv, long R, int D, float eps, void* stream) {
  if (R < 0 || D <= 0) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  k_l2norm_fwd<<<dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream>>>(x, y, inv, D, eps);
  return launch_status();
}
extern "C" int cddmsl_l2norm_bwd(const float* dy, const float* y, const float* inv, float* dx, long R, int D, void* stream) {
  if (R < 0 || D <= 0) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  k_l2norm_bwd<<<dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream>>>(dy, y, inv, dx, D);
  return launch_status();
}
extern "C" int cddmsl_cosine_logits_fwd(const float* x, const float* wn, float* scores, float* inv, long R, int D, int Kc,
                                        float temperature, float eps, void* stream) {
  if (R < 0 || D <= 0 || Kc <= 0 || !(temperature > 0.f)) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  k_cosine_logits_fwd<<<dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(x, wn, scores, inv, R, D, Kc, 1.0f / temperature, eps);
  return launch_status();
}
extern "C" int cddmsl_cosine_logits_bwd(const float* ds, const float* x, const float* wn, const float* inv, float* dx, long R,
                                        int D, int Kc, float temperature, int accumulate, void* stream) {
  if (R < 0 || D <= 0 || Kc <= 0 || !(temperature > 0.f)) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  k_cosine_logits_bwd<<<dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(ds, x, wn, inv, dx, R, D, Kc, 1.0f / temperature, accumulate);
  return launch_status();
}
// S [n][ld] f32 -> loss (1 float), rlse/clse (n floats each, saved for backward)
extern "C" int cddmsl_contrastive_fwd(const float* S, float* rlse, float* clse, float* loss, int n, int ld, void* stream) {
  if (n <= 0 || ld < n) return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  k_lse_rows<<<dim3(n), dim3(256), 0, st>>>(S, rlse, n, ld, 0);
  k_lse_rows<<<dim3(n), dim3(256), 0, st>>>(S, clse, n, ld, 1);
  k_contrastive_loss<<<dim3(1), dim3(256), 0, st>>>(S, rlse, clse, loss, n, ld);
  return launch_status();
}
extern "C" int cddmsl_contrastive_bwd(const float* S, const float* rlse, const float* clse, const float* gloss, float* dS, int n,
                                      int ld, void* stream) {
  if (n <= 0 || ld < n) return CDDMSL_ERR_ARG;
  long tot = (long)n * n;
  k_contrastive_grad<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(S, rlse, clse, gloss, dS, n, ld);
  return launch_status();
}

// ================================================================================================
// LayerNorm (mapper, clipcap.py:97-100 pre-LN) and the detection losses as fused fp32 kernels
//   layernorm        nn.LayerNorm(768), eps 1e-5, frozen affine: f32 residual stream in, T (GEMM operand) out
//   focal CE         FastRCNNOutputLayers.focal_loss  fast_rcnn.py:624-644  mean_i CE_i (1-p_t)^gamma w_i
//   box L1           box_reg_loss                     fast_rcnn.py:646-689  sum over fg rows / R
//   BCE-with-logits  RPN.losses                       rpn.py:405-427        sum over sampled anchors
// ================================================================================================
namespace {

template <typename T> struct OutT;
template <> struct OutT<__bf16> { __device__ static __forceinline__ void st(void* p, long i, float v) { ((unsigned short*)p)[i] = f2bf(v); }
                                  __device__ static __forceinline__ float ld(const void* p, long i) { return bf2f(((const unsigned short*)p)[i]); } };
template <> struct OutT<float> { __device__ static __forceinline__ void st(void* p, long i, float v) { ((float*)p)[i] = v; }
                                 __device__ static __forceinline__ float ld(const void* p, long i) { return ((const float*)p)[i]; } };
// four consecutive elements (index in units of 4 elements)
__device__ __forceinline__ void st4(__bf16*, void* p, long i4, float4 v) { ((uint2*)p)[i4] = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w)); }
__device__ __forceinline__ void st4(float*, void* p, long i4, float4 v) { ((float4*)p)[i4] = v; }
__device__ __forceinline__ float4 ld4(__bf16*, const void* p, long i4) {
  const uint2 u = ((const uint2*)p)[i4];
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4(float*, const void* p, long i4) { return ((const float4*)p)[i4]; }

// Rows of NSEG x 256 columns (the mapper's 768): one wave per row, the row held in registers -- every operand is read once, in
// 16-byte (f32) / 8-byte (bf16) pieces.  Same expressions as the generic kernels below; only the order of the row sums differs.
template <typename T, int NSEG>
__global__ __launch_bounds__(256) void k_layernorm_fwd_v(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, long R, float eps) {
  constexpr int D = NSEG * 256;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float4* xr = (const float4*)(x + r * D);
  float4 v[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) v[s] = xr[s * 64 + lane];
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < NSEG; ++s) sum += (v[s].x + v[s].y) + (v[s].z + v[s].w);
  const float mu = wave_sum(sum) / (float)D;
  float var = 0.f;
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    const float a = v[s].x - mu, b = v[s].y - mu, c = v[s].z - mu, d = v[s].w - mu;
    var += (a * a + b * b) + (c * c + d * d);
  }
  const float rs = rsqrtf(wave_sum(var) / (float)D + eps);
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    const float4 g = ((const float4*)gamma)[s * 64 + lane], b = ((const float4*)beta)[s * 64 + lane];
    float4 o;
    o.x = (v[s].x - mu) * rs * g.x + b.x; o.y = (v[s].y - mu) * rs * g.y + b.y;
    o.z = (v[s].z - mu) * rs * g.z + b.z; o.w = (v[s].w - mu) * rs * g.w + b.w;
    st4((T*)nullptr, y, r * (D / 4) + s * 64 + lane, o);
  }
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}
template <typename T, int NSEG>
__global__ __launch_bounds__(256) void k_layernorm_bwd_v(const void* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* dx, long R, int accumulate) {
  constexpr int D = NSEG * 256;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float4* xr = (const float4*)(x + r * D);
  float4* dxr = (float4*)(dx + r * D);
  const float mu = mean[r], rs = rstd[r];
  float4 g[NSEG], xh[NSEG], o[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    const float4 d4 = ld4((T*)nullptr, dy, r * (D / 4) + s * 64 + lane), ga = ((const float4*)gamma)[s * 64 + lane], xv = xr[s * 64 + lane];
    if (accumulate) o[s] = dxr[s * 64 + lane];
    g[s] = make_float4(d4.x * ga.x, d4.y * ga.y, d4.z * ga.z, d4.w * ga.w);
    xh[s] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
  }
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    sg += (g[s].x + g[s].y) + (g[s].z + g[s].w);
    sgx += (g[s].x * xh[s].x + g[s].y * xh[s].y) + (g[s].z * xh[s].z + g[s].w * xh[s].w);
  }
  sg = wave_sum(sg) / (float)D; sgx = wave_sum(sgx) / (float)D;
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    float4 v;
    v.x = rs * (g[s].x - sg - xh[s].x * sgx); v.y = rs * (g[s].y - sg - xh[s].y * sgx);
    v.z = rs * (g[s].z - sg - xh[s].z * sgx); v.w = rs * (g[s].w - sg - xh[s].w * sgx);
    if (accumulate) { v.x = o[s].x + v.x; v.y = o[s].y + v.y; v.z = o[s].z + v.z; v.w = o[s].w + v.w; }
    dxr[s * 64 + lane] = v;
  }
}
template <typename T> bool layernorm_fwd_v(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                           long R, int D, float eps, hipStream_t st) {
  const dim3 grid((unsigned)((R + 3) / 4)), blk(256);
  switch (D) {
    case 256: k_layernorm_fwd_v<T, 1><<<grid, blk, 0, st>>>(x, gamma, beta, y, mean, rstd, R, eps); return true;
    case 512: k_layernorm_fwd_v<T, 2><<<grid, blk, 0, st>>>(x, gamma, beta, y, mean, rstd, R, eps); return true;
    case 768: k_layernorm_fwd_v<T, 3><<<grid, blk, 0, st>>>(x, gamma, beta, y, mean, rstd, R, eps); return true;
    case 1024: k_layernorm_fwd_v<T, 4><<<grid, blk, 0, st>>>(x, gamma, beta, y, mean, rstd, R, eps); return true;
  }
  return false;
}
template <typename T> bool layernorm_bwd_v(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                           float* dx, long R, int D, int accumulate, hipStream_t st) {
  const dim3 grid((unsigned)((R + 3) / 4)), blk(256);
  switch (D) {
    case 256: k_layernorm_bwd_v<T, 1><<<grid, blk, 0, st>>>(dy, x, gamma, mean, rstd, dx, R, accumulate); return true;
    case 512: k_layernorm_bwd_v<T, 2><<<grid, blk, 0, st>>>(dy, x, gamma, mean, rstd, dx, R, accumulate); return true;
    case 768: k_layernorm_bwd_v<T, 3><<<grid, blk, 0, st>>>(dy, x, gamma, mean, rstd, dx, R, accumulate); return true;
    case 1024: k_layernorm_bwd_v<T, 4><<<grid, blk, 0, st>>>(dy, x, gamma, mean, rstd, dx, R, accumulate); return true;
  }
  return false;
}
__host__ inline bool aligned16(const void* a, const void* b, const void* c, const void* d, const void* e = nullptr) {
  return ((((size_t)a) | ((size_t)b) | ((size_t)c) | ((size_t)d) | ((size_t)e)) & 15) == 0;
}

// one wave per row
template <typename T>
__global__ void k_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                long R, int D, float eps) {
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + r * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += xr[d];
  float mu = wave_sum(s) / (float)D;
  float v = 0.f;
  for (int d = lane; d < D; d += 64) { float c = xr[d] - mu; v += c * c; }
  float rs = rsqrtf(wave_sum(v) / (float)D + eps);
  for (int d = lane; d < D; d += 64) OutT<T>::st(y, r * D + d, (xr[d] - mu) * rs * gamma[d] + beta[d]);
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma   (affine frozen: no dgamma/dbeta)
template <typename T>
__global__ void k_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                float* dx, long R, int D, int accumulate) {
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + r * D;
  float mu = mean[r], rs = rstd[r];
  float sg = 0.f, sgx = 0.f;
  for (int d = lane; d < D; d += 64) {
    float g = OutT<T>::ld(dy, r * D + d) * gamma[d];
    float xh = (xr[d] - mu) * rs;
    sg += g; sgx += g * xh;
  }
  sg = wave_sum(sg) / (float)D; sgx = wave_sum(sgx) / (float)D;
  for (int d = lane; d < D; d += 64) {
    float g = OutT<T>::ld(dy, r * D + d) * gamma[d];
    float xh = (xr[d] - mu) * rs;
    float v = rs * (g - sg - xh * sgx);
    dx[r * D + d] = accumulate ? dx[r * D + d] + v : v;
  }
}

// focal-scaled, background-weighted CE over [R][C] logits (C <= 64): one lane per class, 4 rows per block.
// row_loss[r] = CE * (1-pt)^gamma * w ; dlogits (mean reduction folded in by the caller's upstream gradient).
__global__ void k_focal_ce_fwd(const float* logits, const long* target, float* row_loss, float* probs, long R, int C,
                               float gamma, int bg_class, float bg_weight) {
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= R) return;
  float z = lane < C ? logits[r * C + lane] : -INFINITY;
  float mx = wave_max(z);
  float e = lane < C ? expf(z - mx) : 0.f;
  float se = wave_sum(e);
  float p = e / se;
  if (lane < C) probs[r * C + lane] = p;
  int t = (int)target[r];
  float zt = __shfl(z, t, 64), pt = __shfl(p, t, 64);
  if (lane == 0) {
    float ce = (mx + logf(se)) - zt;
    float w = (t == bg_class) ? bg_weight : 1.f;
    float mod = gamma > 0.f ? powf(fmaxf(1.f - pt, 0.f), gamma) : 1.f;
    row_loss[r] = ce * mod * w;
  }
}
// d(row_loss)/d(logit_c) = w * [ mod * (p_c - 1[c==t]) + ce * d(mod)/dz_c ],  d(mod)/dz_c = -gamma (1-pt)^(gamma-1) pt (1[c==t] - p_c)
__global__ void k_focal_ce_bwd(const float* logits, const long* target, const float* probs, const float* gscale, float* dlogits,
                               long R, int C, float gamma, int bg_class, float bg_weight) {
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (r >= R) return;
  int t = (int)target[r];
  float p = lane < C ? probs[r * C + lane] : 0.f;
  float pt = __shfl(p, t, 64);
  float w = (t == bg_class) ? bg_weight : 1.f;
  float one = lane == t ? 1.f : 0.f;
  float ce = -logf(fmaxf(pt, 1e-38f));
  float omp = fmaxf(1.f - pt, 0.f);
  float mod = gamma > 0.f ? powf(omp, gamma) : 1.f;
  float dmod = (gamma > 0.f && omp > 0.f) ? -gamma * powf(omp, gamma - 1.f) * pt * (one - p) : 0.f;
  float g = w * (mod * (p - one) + ce * dmod) * gscale[0];
  if (lane < C) dlogits[r * C + lane] = g;
}

}  // namespace

extern "C" int cddmsl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    long R, int D, float eps, int dtype, void* stream) {
  if (R < 0 || D <= 0 || (dtype != 0 && dtype != 1)) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  dim3 grid((unsigned)((R + 3) / 4));
  if (aligned16(x, gamma, beta, y) &&
      (dtype == 0 ? layernorm_fwd_v<__bf16>(x, gamma, beta, y, mean, rstd, R, D, eps, (hipStream_t)stream)
                  : layernorm_fwd_v<float>(x, gamma, beta, y, mean, rstd, R, D, eps, (hipStream_t)stream)))
    return launch_status();
  if (dtype == 0) k_layernorm_fwd<__bf16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, gamma, beta, y, mean, rstd, R, D, eps);
  else k_layernorm_fwd<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, gamma, beta, y, mean, rstd, R, D, eps);
  return launch_status();
}
extern "C" int cddmsl_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                    float* dx, long R, int D, int accumulate, int dtype, void* stream) {
  if (R < 0 || D <= 0 || (dtype != 0 && dtype != 1)) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  dim3 grid((unsigned)((R + 3) / 4));
  if (aligned16(dy, x, gamma, dx) &&
      (dtype == 0 ? layernorm_bwd_v<__bf16>(dy, x, gamma, mean, rstd, dx, R, D, accumulate, (hipStream_t)stream)
                  : layernorm_bwd_v<float>(dy, x, gamma, mean, rstd, dx, R, D, accumulate, (hipStream_t)stream)))
    return launch_status();
  if (dtype == 0) k_layernorm_bwd<__bf16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(dy, x, gamma, mean, rstd, dx, R, D, accumulate);
  else k_layernorm_bwd<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>(dy, x, gamma, mean, rstd, dx, R, D, accumulate);
  return launch_status();
}
extern "C" int cddmsl_focal_ce_fwd(const float* logits, const long* target, float* row_loss, float* probs, long R, int C,
                                   float gamma, int bg_class, float bg_weight, void* stream) {
  if (R < 0 || C <= 0 || C > 64) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  k_focal_ce_fwd<<<dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(logits, target, row_loss, probs, R, C, gamma, bg_class, bg_weight);
  return launch_status();
}
extern "C" int cddmsl_focal_ce_bwd(const float* logits, const long* target, const float* probs, const float* gscale,
                                   float* dlogits, long R, int C, float gamma, int bg_class, float bg_weight, void* stream) {
  if (R < 0 || C <= 0 || C > 64) return CDDMSL_ERR_ARG;
  if (R == 0) return CDDMSL_OK;
  k_focal_ce_bwd<<<dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(logits, target, probs, gscale, dlogits, R, C, gamma, bg_class, bg_weight);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------------
// Losses over SAMPLED rows, index lists known (no dense pass, no boolean masks):
//   RPN  (modeling/proposal_generator/rpn.py:365-429, box_regression.py:229-270; smooth-L1 beta 0 = L1):
//        loss_cls = sum over the sampled anchors of BCE-with-logits(logit, label) * inv_norm      (label 1 = positive, 0 = negative)
//        loss_loc = sum over the positives of |delta - get_deltas(anchor, matched gt box)| * inv_norm
//   box head (modeling/roi_heads/fast_rcnn.py:646-689): sum over the foreground rows of |delta[class-specific 4] - get_deltas(proposal, gt)| * inv_norm
// One block: a few thousand rows at most, summed in a fixed order (deterministic).  The backward scatters into caller-zeroed
// gradient tensors: d/dlogit = (sigmoid(x) - y) g inv_norm, d/ddelta = sign(delta - target) g inv_norm (0 at equality, as torch.abs).
namespace {
struct BoxW { float wx, wy, ww, wh; };
// Box2BoxTransform.get_deltas (box_regression.py:42-75), same expression order as cddmsl_amd/modeling/rpn.py::get_deltas
__device__ __forceinline__ void get_deltas4(const float* s, const float* t, BoxW w, float* d) {
  const float sw = s[2] - s[0], sh = s[3] - s[1];
  const float sx = s[0] + 0.5f * sw, sy = s[1] + 0.5f * sh;
  const float tw = t[2] - t[0], th = t[3] - t[1];
  const float tx = t[0] + 0.5f * tw, ty = t[1] + 0.5f * th;
  d[0] = w.wx * (tx - sx) / sw; d[1] = w.wy * (ty - sy) / sh;
  d[2] = w.ww * logf(tw / sw); d[3] = w.wh * logf(th / sh);
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void k_rpn_losses(const float* logits, const float* deltas, const long* pos, int npos, const long* neg, int nneg,
                                                     const long* midx, const float* gt, const long* gt_off, const float* anchors, long A, BoxW w,
                                                     float inv_norm, float* out2, const float* gout2, float* dlogits, float* ddeltas) {
  __shared__ float red[4];
  const bool bwd = gout2 != nullptr;
  const float gc = bwd ? gout2[0] * inv_norm : 0.f, gl = bwd ? gout2[1] * inv_norm : 0.f;
  float cls = 0.f, loc = 0.f;
  for (int i = threadIdx.x; i < npos + nneg; i += blockDim.x) {
    const bool is_pos = i < npos;
    const long r = is_pos ? pos[i] : neg[i - npos];
    const float x = logits[r], y = is_pos ? 1.f : 0.f;
    if (!bwd) cls += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));          // binary_cross_entropy_with_logits
    else dlogits[r] = (1.f / (1.f + expf(-x)) - y) * gc;
    if (is_pos) {
      const long img = r / A, a = r - img * A;
      float d[4];
      get_deltas4(anchors + 4 * a, gt + 4 * (midx[r] + gt_off[img]), w, d);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float e = deltas[4 * r + c] - d[c];
        if (!bwd) loc += fabsf(e); else ddeltas[4 * r + c] = sgn(e) * gl;
      }
    }
  }
  if (!bwd) {
    cls = block_sum(cls, red);
    loc = block_sum(loc, red);
    if (threadIdx.x == 0) { out2[0] = cls * inv_norm; out2[1] = loc * inv_norm; }
  }
}

__global__ __launch_bounds__(256) void k_box_l1(const float* deltas, int ld, const long* fg, int nfg, const long* cls, const float* src, const float* tgt,
                                                BoxW w, float inv_norm, float* out1, const float* gout1, float* ddeltas) {
  __shared__ float red[4];
  const bool bwd = gout1 != nullptr;
  const float g = bwd ? gout1[0] * inv_norm : 0.f;
  float acc = 0.f;
  for (int i = threadIdx.x; i < nfg; i += blockDim.x) {
    const long r = fg[i], c0 = cls ? 4 * cls[r] : 0;
    float d[4];
    get_deltas4(src + 4 * r, tgt + 4 * r, w, d);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float e = deltas[r * ld + c0 + c] - d[c];
      if (!bwd) acc += fabsf(e); else ddeltas[r * ld + c0 + c] = sgn(e) * g;
    }
  }
  if (!bwd) {
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out1[0] = acc * inv_norm;
  }
}
}  // namespace

// logits [N*A] f32, deltas [N*A][4] f32, pos / neg: global anchor indices (image * A + anchor) of the sampled anchors, midx [N*A] the
// matched gt index per anchor (within its image), gt [G][4] all images' boxes, gt_off [N] first row of each image's boxes,
// anchors [A][4].  out2 = (loss_rpn_cls, loss_rpn_loc).  Backward (gout2 given): dlogits / ddeltas must be zero-filled by the caller.
extern "C" int cddmsl_rpn_losses(const float* logits, const float* deltas, const long* pos, int npos, const long* neg, int nneg, const long* midx,
                                 const float* gt, const long* gt_off, const float* anchors, long A, float wx, float wy, float ww, float wh,
                                 float inv_norm, float* out2, const float* gout2, float* dlogits, float* ddeltas, void* stream) {
  if (npos < 0 || nneg < 0 || A <= 0 || (gout2 ? (!dlogits || !ddeltas) : !out2)) return CDDMSL_ERR_ARG;
  hipLaunchKernelGGL(k_rpn_losses, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, deltas, pos, npos, neg, nneg, midx, gt, gt_off, anchors, A,
                     BoxW{wx, wy, ww, wh}, inv_norm, out2, gout2, dlogits, ddeltas);
  return launch_status();
}

// deltas [R][ld] f32, fg [nfg] row indices, cls [R] (nullable: class-agnostic, column 0) the class whose 4 columns a row trains,
// src / tgt [R][4] proposal and ground-truth boxes.  out1 = sum |delta - target| * inv_norm.  Backward: ddeltas zero-filled by the caller.
extern "C" int cddmsl_box_l1(const float* deltas, int ld, const long* fg, int nfg, const long* cls, const float* src, const float* tgt, float wx,
                             float wy, float ww, float wh, float inv_norm, float* out1, const float* gout1, float* ddeltas, void* stream) {
  if (nfg < 0 || ld < 4 || (gout1 ? !ddeltas : !out1)) return CDDMSL_ERR_ARG;
  hipLaunchKernelGGL(k_box_l1, dim3(1), dim3(256), 0, (hipStream_t)stream, deltas, ld, fg, nfg, cls, src, tgt, BoxW{wx, wy, ww, wh}, inv_norm, out1,
                     gout1, ddeltas);
  return launch_status();
}
