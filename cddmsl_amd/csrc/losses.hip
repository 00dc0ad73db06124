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
