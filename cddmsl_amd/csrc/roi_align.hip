// RoIAlign forward / backward for gfx950, channels-last.
//
// Replaces torchvision.ops.roi_align as called from the reference:
//   ROIAlign.forward   detectron2/layers/roi_align.py:49-65  (aligned=True, sampling_ratio=0, scale 1/16)
//   ROIPooler.forward  detectron2/modeling/poolers.py:190-229
// Algorithm = torchvision's: adaptive sample grid ceil(roi/pooled), bilinear
// with the [-1, H] validity window, clamp at 0, top edge snap to H-1.
//
// Layout: feature map NHWC [N][H][W][C], rois [K][5] f32 (batch, x0, y0, x1, y1), output [K][ph][pw][C]
// (= K images of ph x pw for the RoI layer4 convs).  16 B per lane along C everywhere.
//
// Backward avoids global atomics (the reference's CUDA path scatter-adds): bilinear weights are
// separable, so sum_samples w(y,x) = Ay[py][bi] * Ax[px][bj] with per-RoI, per-axis tables
// (k_roi_tables).  The gather kernel then walks, for every feature pixel, the RoIs of its image whose
// footprint covers it and accumulates Ay*Ax*dY in registers -- deterministic, written once.
#include "common.h"

namespace {

template <typename T> struct Vec;
template <> struct Vec<__bf16> {
  static constexpr int ES = 2, VEC = 8;
  __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f(v[j] & 0xffff); f[2 * j + 1] = bf2f(v[j] >> 16); }
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack2bf(f[2 * j], f[2 * j + 1]);
    return v;
  }
};
template <> struct Vec<float> {
  static constexpr int ES = 4, VEC = 4;
  __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
    const f32x4 x = __builtin_bit_cast(f32x4, v);
    f[0] = x[0]; f[1] = x[1]; f[2] = x[2]; f[3] = x[3];
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    f32x4 x = {f[0], f[1], f[2], f[3]};
    return __builtin_bit_cast(u32x4, x);
  }
};

struct RoiGeom {
  int b, gh, gw;
  float x0, y0, bh, bw;
};
__device__ __forceinline__ RoiGeom roi_geom(const float* r, float scale, int ph, int pw, int sampling_ratio, int aligned) {
  RoiGeom g;
  g.b = (int)r[0];
  float off = aligned ? 0.5f : 0.0f;
  g.x0 = r[1] * scale - off; g.y0 = r[2] * scale - off;
  float x1 = r[3] * scale - off, y1 = r[4] * scale - off;
  float rw = x1 - g.x0, rh = y1 - g.y0;
  if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
  g.bh = rh / (float)ph; g.bw = rw / (float)pw;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
  return g;
}
// one axis of the bilinear tap: returns false when the sample is outside [-1, L]
__device__ __forceinline__ bool axis_tap(float v, int L, int& lo, int& hi, float& wl, float& wh) {
  if (v < -1.0f || v > (float)L) return false;
  if (v <= 0.f) v = 0.f;
  lo = (int)v;
  if (lo >= L - 1) { hi = lo = L - 1; v = (float)lo; } else hi = lo + 1;
  wh = v - (float)lo;
  wl = 1.0f - wh;
  return true;
}

// grid: K*ph blocks (one RoI bin ROW each: the pw bins of a row share their y taps, and 14x fewer, longer blocks than one
// block per bin); block: min(256, cch rounded) threads, each 16 B of channels (loops if C is larger).
// The bilinear taps of a bin row -- gh y samples and pw*gw x samples, the same for every channel -- are computed ONCE per
// block into LDS (one sample per thread): with one 16-byte channel chunk per thread (C = 1024) the per-thread tap
// arithmetic (~50 vector instructions per sample) was most of the kernel (VALU-bound at 3x its HBM time).  Same tap
// expressions, same accumulation order: bit-identical output.
constexpr int ROI_MAXS = 256;     // samples per axis held in LDS (adaptive sampling grids beyond that take the inline path)
// RP = bin rows (and columns) per step: 1, or 2 when the 2x2-average-pooled map is written as well (yp): the first block of the
// RoI head's layer4 pools its input for the downsample path (clip_backbone.py:45-52), and reading the 3.3 GB map back just
// for that costs more than computing four neighbouring bins in one thread.  The pooled value is formed from the four ROUNDED
// outputs in avgpool2_fwd's order, so it is bit-identical to pooling the stored map.
// ``y`` may be absent (RP = 2: only the pooled map is wanted); ``esc`` / ``ebi`` (per channel) and ``relu``: y = relu?(esc * v + ebi)
// applied to the pooled-over-samples value before it is rounded -- the FrozenBN + ReLU of a 1x1 convolution that was applied to
// the feature map BEFORE the pooling (both are linear: see cddmsl_roi_align_forward_affine).
template <typename T, int RP>
__global__ void k_roi_align_fwd(const char* x, const float* rois, char* y, char* yp, int* dbg_grid, int N, int H, int W, int cch,
                                int ph, int pw, float scale, int sampling_ratio, int aligned, const float* esc, const float* ebi,
                                int relu, char* y8, const float* q8, unsigned* amax8) {
  // y8 (bf16 instantiation, fp8 configuration): a second, OCP e4m3 copy of y = sat(y * q8[0]) for the convolution that consumes
  // the crops, max|y| recorded in amax8[block & 63] -- see cddmsl_conv_fwd_q8
  const float q8s = (y8 && q8) ? q8[0] : 1.f;
  unsigned am8 = 0u;                                 // (bit pattern of max |y|: common.h absmax_bits)
  const int nrb = ph / RP;
  const int i0 = (blockIdx.x % nrb) * RP, k = blockIdx.x / nrb;
  RoiGeom g = roi_geom(rois + 5 * (long)k, scale, ph, pw, sampling_ratio, aligned);
  if (dbg_grid && i0 == 0 && threadIdx.x == 0) { dbg_grid[2 * k] = g.gh; dbg_grid[2 * k + 1] = g.gw; }
  float count = (float)max(g.gh * g.gw, 1);
  const int icount = max(g.gh * g.gw, 1);
  const bool pow2 = (icount & (icount - 1)) == 0;
  const float inv_count = 1.0f / count;
  const u32x4* xb = (const u32x4*)x + (long)g.b * H * W * cch;
  constexpr int VEC = Vec<T>::VEC;
  __shared__ int s_lo[2][ROI_MAXS], s_hi[2][ROI_MAXS];
  __shared__ float s_wl[2][ROI_MAXS], s_wh[2][ROI_MAXS];
  const int ny = RP * g.gh, nx = pw * g.gw;
  const bool tabled = ny <= ROI_MAXS && nx <= ROI_MAXS;
  if (tabled) {
    for (int s = threadIdx.x; s < ny + nx; s += blockDim.x) {
      int lo = -1, hi = -1; float wl = 0.f, wh = 0.f;
      if (s < ny) {
        const int rr = s / g.gh, iy = s - rr * g.gh;
        const float yy = g.y0 + (float)(i0 + rr) * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
        if (!axis_tap(yy, H, lo, hi, wl, wh)) lo = -1;
        s_lo[0][s] = lo; s_hi[0][s] = hi; s_wl[0][s] = wl; s_wh[0][s] = wh;
      } else {
        const int sx = s - ny, j = sx / g.gw, ix = sx - j * g.gw;
        const float xx = g.x0 + (float)j * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
        if (!axis_tap(xx, W, lo, hi, wl, wh)) lo = -1;
        s_lo[1][sx] = lo; s_hi[1][sx] = hi; s_wl[1][sx] = wl; s_wh[1][sx] = wh;
      }
    }
    __syncthreads();
  }
  for (int j0 = 0; j0 < pw; j0 += RP) {
  for (int c = threadIdx.x; c < cch; c += blockDim.x) {
    u32x4 outp[RP][RP];
#pragma unroll
    for (int rr = 0; rr < RP; ++rr)
#pragma unroll
    for (int cc = 0; cc < RP; ++cc) {
    const int i = i0 + rr, j = j0 + cc;
    const long bin = ((long)k * ph + i) * pw + j;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g.b >= 0 && g.b < N) {
      for (int iy = 0; iy < g.gh; ++iy) {
        int yl, yh; float wyl, wyh;
        if (tabled) {
          yl = s_lo[0][rr * g.gh + iy];
          if (yl < 0) continue;
          yh = s_hi[0][rr * g.gh + iy]; wyl = s_wl[0][rr * g.gh + iy]; wyh = s_wh[0][rr * g.gh + iy];
        } else {
          float yy = g.y0 + (float)i * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
          if (!axis_tap(yy, H, yl, yh, wyl, wyh)) continue;
        }
        for (int ix = 0; ix < g.gw; ++ix) {
          int xl, xh; float wxl, wxh;
          if (tabled) {
            const int sx = j * g.gw + ix;
            xl = s_lo[1][sx];
            if (xl < 0) continue;
            xh = s_hi[1][sx]; wxl = s_wl[1][sx]; wxh = s_wh[1][sx];
          } else {
            float xx = g.x0 + (float)j * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
            if (!axis_tap(xx, W, xl, xh, wxl, wxh)) continue;
          }
          float v1[8], v2[8], v3[8], v4[8];
          Vec<T>::unpack(xb[((long)yl * W + xl) * cch + c], v1);
          Vec<T>::unpack(xb[((long)yl * W + xh) * cch + c], v2);
          Vec<T>::unpack(xb[((long)yh * W + xl) * cch + c], v3);
          Vec<T>::unpack(xb[((long)yh * W + xh) * cch + c], v4);
          float w1 = wyl * wxl, w2 = wyl * wxh, w3 = wyh * wxl, w4 = wyh * wxh;
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            if (sizeof(T) == 2) {
              // throughput (bf16) instantiation: the four taps as one multiply + three fused multiply-adds (5 instead of 8
              // vector-ALU operations per element -- this loop is what bounds the kernel); the result is rounded to bf16
              // anyway.  The exact-f32 instantiation keeps the reference's expression (roi_align_kernel: w1*v1 + ... + w4*v4).
              float tq = w1 * v1[q];
              tq = __builtin_fmaf(w2, v2[q], tq);
              tq = __builtin_fmaf(w3, v3[q], tq);
              tq = __builtin_fmaf(w4, v4[q], tq);
              acc[q] += tq;
            } else {
              acc[q] += w1 * v1[q] + w2 * v2[q] + w3 * v3[q] + w4 * v4[q];
            }
          }
        }
      }
    }
    if (pow2) {                                   // (x / 2^k == x * 2^-k exactly: skips 8 IEEE divisions per chunk)
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] *= inv_count;
    } else {
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] /= count;
    }
    if (esc) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const float sq = esc[c * VEC + q], bq = ebi[c * VEC + q];
        acc[q] = sizeof(T) == 2 ? __builtin_fmaf(acc[q], sq, bq) : acc[q] * sq + bq;       // (as the conv epilogues: gemm_conv.hip affine<T>)
        if (relu) acc[q] = fmaxf(acc[q], 0.f);
      }
    }
    outp[rr][cc] = Vec<T>::pack(acc);
    if (y) ((u32x4*)y)[bin * cch + c] = outp[rr][cc];
    if (sizeof(T) == 2 && y8) {
      float f8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { am8 = absmax_bits(am8, acc[q]); f8[q] = __builtin_amdgcn_fmed3f(acc[q] * q8s, -448.f, 448.f); }
      int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f8[0], f8[1], 0, false), w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f8[4], f8[5], 0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f8[2], f8[3], w0, true); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f8[6], f8[7], w1, true);
      const u32x2 o8 = {(unsigned)w0, (unsigned)w1};
      ((u32x2*)y8)[bin * cch + c] = o8;
    }
    }
    if (RP == 2) {
      float a0[8], a1[8], a2[8], a3[8], o[8];
      Vec<T>::unpack(outp[0][0], a0); Vec<T>::unpack(outp[0][RP - 1], a1);
      Vec<T>::unpack(outp[RP - 1][0], a2); Vec<T>::unpack(outp[RP - 1][RP - 1], a3);
#pragma unroll
      for (int q = 0; q < VEC; ++q) o[q] = ((a0[q] + a1[q]) + (a2[q] + a3[q])) * 0.25f;
      ((u32x4*)yp)[(((long)k * (ph / 2) + i0 / 2) * (pw / 2) + j0 / 2) * cch + c] = Vec<T>::pack(o);
    }
  }
  }
  if (sizeof(T) == 2 && y8 && amax8) {              // block-uniform condition
    __shared__ unsigned sm8[4];
    am8 = wave_max_u(am8);
    if ((threadIdx.x & 63) == 0) sm8[threadIdx.x >> 6] = am8;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned m = sm8[0];
      for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) m = sm8[wv] > m ? sm8[wv] : m;
      atomicMax(amax8 + (blockIdx.x & 63), m);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Throughput (bf16) forward: every feature pixel a bin row needs is loaded ONCE.
//
// k_roi_align_fwd above walks (bin, sample, 4 taps): 4 * gh * gw loads of a pixel's channel chunk per bin.  On the training
// workload the sampled RoIs average 8.8 x 9.4 feature pixels (tools: scratch profile of round 3), i.e. bins of ~0.65 pixel: the
// 14 bins of a row re-read the same ~11 pixel columns 56+ times, and tap loads + per-sample table reads were ~90 % of the kernel
// (round-2 ablation).  Bilinear weights are separable and the row weights do not depend on the x sample, so for one bin row
//     R[px]      = sum over the row's y samples of (wy_lo x[y_lo][px] + wy_hi x[y_hi][px])     (duplicate rows merged)
//     bin j      = 1/count * sum over its x samples of (wx_lo R[x_lo] + wx_hi R[x_hi])
// and x_lo never decreases along the row: the wave keeps R[cur], R[cur+1] in registers and slides.  One wave = one bin row (RP = 1)
// or one POOLED row = two bin rows folded at half weight (RP = 2: the 2x2-average-pooled crops only) x 64 channel chunks; all
// addresses and weights are wave-uniform (LDS tables, readfirstlane).  Same sums in another order: the bf16 results differ from
// k_roi_align_fwd's by rounding; the exact-f32 instantiation keeps the reference's expression order.
constexpr int ROW_MAXY = 32;       // merged feature rows per bin row pair held in LDS
template <int RP>
__global__ __launch_bounds__(256) void k_roi_align_fwd_rows(const char* x, const float* rois, char* y, int N, int H, int W, int cch, int ph, int pw,
                                                            float scale, int sampling_ratio, int aligned, const float* esc, const float* ebi, int relu) {
  const int nrb = ph / RP;
  const int i0 = (blockIdx.x % nrb) * RP, k = blockIdx.x / nrb;
  const RoiGeom g = roi_geom(rois + 5 * (long)k, scale, ph, pw, sampling_ratio, aligned);
  __shared__ int s_xl[ROI_MAXS], s_py[ROW_MAXY], s_n[2];
  __shared__ float s_xwl[ROI_MAXS], s_xwh[ROI_MAXS], s_wy[ROW_MAXY];
  __shared__ int s_ylo[ROW_MAXY], s_yhi[ROW_MAXY];
  __shared__ float s_ywl[ROW_MAXY], s_ywh[ROW_MAXY];
  const int ny = RP * g.gh, nx = pw * g.gw;                    // y samples of this (pooled) row; x samples of the whole row
  const int opw = pw / RP;                                     // outputs per row
  const int c = threadIdx.x;                                   // channel chunk of this lane
  const bool inb = g.b >= 0 && g.b < N;
  if (nx > ROI_MAXS || 2 * ny > ROW_MAXY) {                    // a sampling grid beyond the tables (block-uniform): bins evaluated tap by tap
    if (c >= cch) return;
    const u32x4* xb = (const u32x4*)x + (long)(inb ? g.b : 0) * H * W * cch + c;
    const float inv = 1.0f / ((float)max(g.gh * g.gw, 1) * (float)(RP * RP));
    for (int jo = 0; jo < opw; ++jo) {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int rr = 0; rr < RP && inb; ++rr)
        for (int cc = 0; cc < RP; ++cc)
          for (int iy = 0; iy < g.gh; ++iy) {
            int yl, yh; float wyl, wyh;
            const float yy = g.y0 + (float)(i0 + rr) * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
            if (!axis_tap(yy, H, yl, yh, wyl, wyh)) continue;
            for (int ix = 0; ix < g.gw; ++ix) {
              int xl, xh; float wxl, wxh;
              const float xx = g.x0 + (float)(jo * RP + cc) * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
              if (!axis_tap(xx, W, xl, xh, wxl, wxh)) continue;
              float v1[8], v2[8], v3[8], v4[8];
              Vec<__bf16>::unpack(xb[((long)yl * W + xl) * cch], v1); Vec<__bf16>::unpack(xb[((long)yl * W + xh) * cch], v2);
              Vec<__bf16>::unpack(xb[((long)yh * W + xl) * cch], v3); Vec<__bf16>::unpack(xb[((long)yh * W + xh) * cch], v4);
#pragma unroll
              for (int q = 0; q < 8; ++q) acc[q] += wyl * (wxl * v1[q] + wxh * v2[q]) + wyh * (wxl * v3[q] + wxh * v4[q]);
            }
          }
      float o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        o[q] = acc[q] * inv;
        if (esc) { o[q] = __builtin_fmaf(o[q], esc[c * 8 + q], ebi[c * 8 + q]); if (relu) o[q] = fmaxf(o[q], 0.f); }
      }
      ((u32x4*)y)[(((long)k * nrb + i0 / RP) * opw + jo) * cch + c] = Vec<__bf16>::pack(o);
    }
    return;
  }
  for (int s = threadIdx.x; s < ny + nx; s += blockDim.x) {
    int lo = -1, hi = -1; float wl = 0.f, wh = 0.f;
    if (s < ny) {
      const int rr = s / g.gh, iy = s - rr * g.gh;
      const float yy = g.y0 + (float)(i0 + rr) * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      if (!axis_tap(yy, H, lo, hi, wl, wh)) lo = -1;
      s_ylo[s] = lo; s_yhi[s] = hi; s_ywl[s] = wl; s_ywh[s] = wh;
    } else {
      const int sx = s - ny, j = sx / g.gw, ix = sx - j * g.gw;
      const float xx = g.x0 + (float)j * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
      if (!axis_tap(xx, W, lo, hi, wl, wh)) lo = -1;
      s_xl[sx] = lo; s_xwl[sx] = wl; s_xwh[sx] = wh;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {                                      // merge the y taps by feature row (samples ascend: rows repeat back to back)
    int n = 0;
    for (int s = 0; s < ny; ++s) {
      if (s_ylo[s] < 0) continue;
      const int r2[2] = {s_ylo[s], s_yhi[s]};
      const float w2[2] = {s_ywl[s], s_ywh[s]};
      for (int e = 0; e < 2; ++e) {
        int f = -1;
        for (int q = 0; q < n; ++q) if (s_py[q] == r2[e]) f = q;
        if (f < 0) { f = n++; s_py[f] = r2[e]; s_wy[f] = 0.f; }
        s_wy[f] += w2[e];
      }
    }
    int first = -1;
    for (int s = 0; s < nx && first < 0; ++s) if (s_xl[s] >= 0) first = s_xl[s];
    s_n[0] = n; s_n[1] = first;
  }
  __syncthreads();
  // The x tables leave LDS once, spread over the lanes of every wave (entry s in lane s & 63 of register s >> 6), and the walk below reads
  // them with v_readlane: the former walk took one LDS read (+ readfirstlane) per sample and table, a full LDS latency inside every
  // iteration of a 28-56-iteration serial loop.  (Loaded by EVERY lane, before the lanes without a channel chunk leave: v_readlane
  // reads a lane's register whether or not the lane is still active.)
  const int lane = threadIdx.x & 63;
  int xlr[ROI_MAXS / 64], xwlr[ROI_MAXS / 64], xwhr[ROI_MAXS / 64];
#pragma unroll
  for (int q = 0; q < ROI_MAXS / 64; ++q) {
    const int idx = q * 64 + lane;
    const bool in = idx < nx;
    xlr[q] = in ? s_xl[idx] : -1;
    xwlr[q] = in ? __builtin_bit_cast(int, s_xwl[idx]) : 0;
    xwhr[q] = in ? __builtin_bit_cast(int, s_xwh[idx]) : 0;
  }
  if (c >= cch) return;
  const int nrow = __builtin_amdgcn_readfirstlane(s_n[0]);
  const int first = __builtin_amdgcn_readfirstlane(s_n[1]);
  const float inv = 1.0f / ((float)max(g.gh * g.gw, 1) * (float)(RP * RP));
  float sq[8], bq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { sq[q] = esc ? esc[c * 8 + q] : 1.f; bq[q] = ebi ? ebi[c * 8 + q] : 0.f; }
  const u32x4* xb = (const u32x4*)x + (long)(inb ? g.b : 0) * H * W * cch + c;
  auto rload = [&](int px, float* R) {                         // R = sum over the merged rows of wy * x[py][px]   (wave-uniform px)
#pragma unroll
    for (int q = 0; q < 8; ++q) R[q] = 0.f;
    for (int e = 0; e < nrow; ++e) {
      const int py = __builtin_amdgcn_readfirstlane(s_py[e]);
      const float wy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_wy[e])));
      float v[8];
      Vec<__bf16>::unpack(xb[((long)py * W + px) * cch], v);
#pragma unroll
      for (int q = 0; q < 8; ++q) R[q] = __builtin_fmaf(wy, v[q], R[q]);
    }
  };
  float R0[8], R1[8], acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { R0[q] = R1[q] = acc[q] = 0.f; }
  int cur = first;
  const bool any = inb && nrow > 0 && first >= 0;
  // the column BEHIND the window is requested one slide ahead (raw rows in registers, up to PF merged rows): its latency passes
  // under the bins in between instead of stalling the slide that needs it
  constexpr int PF = 4;
  const bool ahead = nrow <= PF;
  int pyr[PF]; float wyr[PF];
#pragma unroll
  for (int e = 0; e < PF; ++e) {
    pyr[e] = __builtin_amdgcn_readfirstlane(s_py[e < nrow ? e : 0]);
    wyr[e] = e < nrow ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_wy[e]))) : 0.f;
  }
  u32x4 raw[PF];
  auto issue = [&](int px) {
#pragma unroll
    for (int e = 0; e < PF; ++e) if (e < nrow) raw[e] = xb[((long)pyr[e] * W + px) * cch];
  };
  auto finish = [&](float* R) {
#pragma unroll
    for (int q = 0; q < 8; ++q) R[q] = 0.f;
#pragma unroll
    for (int e = 0; e < PF; ++e) if (e < nrow) {
      float v[8];
      Vec<__bf16>::unpack(raw[e], v);
#pragma unroll
      for (int q = 0; q < 8; ++q) R[q] = __builtin_fmaf(wyr[e], v[q], R[q]);
    }
  };
  if (any) {
    rload(cur, R0); rload(min(cur + 1, W - 1), R1);
    if (ahead) issue(min(cur + 2, W - 1));
  }
  const int spo = RP * g.gw;                                   // x samples per output
  u32x4* yo = (u32x4*)y + ((long)k * nrb + i0 / RP) * opw * cch + c;
  if (nx == 0) {                                               // an empty box has a 0 x 0 sampling grid: every bin pools to zero
    float o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { o[q] = 0.f; if (esc) { o[q] = bq[q]; if (relu) o[q] = fmaxf(o[q], 0.f); } }
    for (int jo = 0; jo < opw; ++jo) yo[(long)jo * cch] = Vec<__bf16>::pack(o);
    return;
  }
  int left = spo, j = 0;
#pragma unroll
  for (int q = 0; q < ROI_MAXS / 64; ++q) {
  const int lim = min(64, nx - q * 64);
  for (int l = 0; l < lim; ++l) {
    const int xl = __builtin_amdgcn_readlane(xlr[q], l);
    if (any && xl >= 0) {
      while (cur < xl) {                                       // slide: x_lo never decreases along the row
        ++cur;
#pragma unroll
        for (int q = 0; q < 8; ++q) R0[q] = R1[q];
        if (ahead) { finish(R1); issue(min(cur + 2, W - 1)); }
        else rload(min(cur + 1, W - 1), R1);
      }
      const float wl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xwlr[q], l));
      const float wh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xwhr[q], l));
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_fmaf(wl, R0[q], __builtin_fmaf(wh, R1[q], acc[q]));
    }
    if (--left == 0) {
      float o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        o[q] = acc[q] * inv;
        if (esc) { o[q] = __builtin_fmaf(o[q], sq[q], bq[q]); if (relu) o[q] = fmaxf(o[q], 0.f); }
        acc[q] = 0.f;
      }
      // (non-temporal: the crops are read by a later launch; stored this way they stop evicting the feature map every RoI re-reads --
      // 1.42 -> 1.36 ms per step and -0.3 ms on the step, same-box A/B)
      __builtin_nontemporal_store(Vec<__bf16>::pack(o), &yo[(long)j * cch]);
      ++j; left = spo;
    }
  }
  }
}

// Per-RoI separable weight tables: ay[k][py][bi] = (1/gh) sum_{iy in bin bi} wy(py; y_iy), likewise ax (1/gw).
// (count = max(gh*gw,1) = gh*gw whenever any sample exists.)  Also the footprint box fp[k] = (ylo,yhi,xlo,xhi)
// inclusive, empty when ylo > yhi.
// ``fold`` = 2: the gradient arrives on the 2x2-AVERAGE-POOLED map (ph x pw there; the RoIAlign grid is 2 ph x 2 pw): pooling is
// separable as well (0.5 per axis), so table column i collects the two bins 2i, 2i+1 at half weight and the same gather kernel
// walks ph x pw pooled bins -- it reads a quarter of the gradient bytes and never sees the full-resolution tensor.
__global__ void k_roi_tables(const float* rois, float* ay, float* ax, int* fp, int K, int H, int W, int ph, int pw,
                             float scale, int sampling_ratio, int aligned, int fold) {
  int k = blockIdx.x;
  RoiGeom g = roi_geom(rois + 5 * (long)k, scale, ph * fold, pw * fold, sampling_ratio, aligned);
  float* ayk = ay + (long)k * H * ph;
  float* axk = ax + (long)k * W * pw;
  for (int i = threadIdx.x; i < H * ph; i += blockDim.x) ayk[i] = 0.f;
  for (int i = threadIdx.x; i < W * pw; i += blockDim.x) axk[i] = 0.f;
  __shared__ int lim[4];
  if (threadIdx.x == 0) { lim[0] = H; lim[1] = -1; lim[2] = W; lim[3] = -1; }
  __syncthreads();
  const float fw = 1.0f / (float)fold;
  // one thread per table column per axis (fold bins each); columns of an axis are disjoint
  if (threadIdx.x < ph) {
    int lo_min = H, hi_max = -1;
    for (int i = threadIdx.x * fold; i < (threadIdx.x + 1) * fold; ++i)
    for (int iy = 0; iy < g.gh; ++iy) {
      float yy = g.y0 + (float)i * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
      int yl, yh; float wl, wh;
      if (!axis_tap(yy, H, yl, yh, wl, wh)) continue;
      ayk[yl * ph + threadIdx.x] += fw * (wl / (float)g.gh);
      ayk[yh * ph + threadIdx.x] += fw * (wh / (float)g.gh);
      lo_min = min(lo_min, yl); hi_max = max(hi_max, yh);
    }
    atomicMin(&lim[0], lo_min); atomicMax(&lim[1], hi_max);
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + pw) {
    const int jc = threadIdx.x - 64;
    int lo_min = W, hi_max = -1;
    for (int j = jc * fold; j < (jc + 1) * fold; ++j)
    for (int ix = 0; ix < g.gw; ++ix) {
      float xx = g.x0 + (float)j * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
      int xl, xh; float wl, wh;
      if (!axis_tap(xx, W, xl, xh, wl, wh)) continue;
      axk[xl * pw + jc] += fw * (wl / (float)g.gw);
      axk[xh * pw + jc] += fw * (wh / (float)g.gw);
      lo_min = min(lo_min, xl); hi_max = max(hi_max, xh);
    }
    atomicMin(&lim[2], lo_min); atomicMax(&lim[3], hi_max);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool ok = g.b >= 0;
    fp[4 * k] = ok ? lim[0] : 1; fp[4 * k + 1] = ok ? lim[1] : 0; fp[4 * k + 2] = lim[2]; fp[4 * k + 3] = lim[3];
  }
}

// gather backward: block = one TS x TS (2x2) TILE of feature pixels of one image; threads over channel chunks.  A dY bin whose
// bilinear support touches several pixels of the tile is loaded once and applied to all of them (the per-pixel version
// re-read every bin ~4x: its traffic, not the 3.3 GB of dY, set the time).  Per pixel the terms are added in the same
// (roi, bin row, bin column) order with the same products as before, so results are bit-identical.
// roi_start[n] .. roi_start[n+1] = the (contiguous) RoIs of image n (rois are grouped by image, as
// convert_boxes_to_pooler_format poolers.py:68-95 emits them).
constexpr int MAXP = 16;
template <typename T, int NC, int TS>   // NC = channel chunks (16 B) per thread; TS x TS = pixels per tile
__global__ __launch_bounds__(256) void k_roi_align_bwd(const char* dy, const float* ay, const float* ax, const int* fp, const int* roi_start,
                                char* dx, int H, int W, int cch, int ph, int pw) {
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const long tile = blockIdx.x;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / ((long)tiles_x * tiles_y);
  const int py0 = TS * ty, px0 = TS * tx;
  const int k0 = roi_start[n], k1 = roi_start[n + 1];
  constexpr int VEC = Vec<T>::VEC;
  float acc[NC][TS * TS][8];                        // [channel chunk of this thread][pixel of the tile][element]
#pragma unroll
  for (int u = 0; u < NC; ++u)
#pragma unroll
    for (int q4 = 0; q4 < TS * TS; ++q4)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[u][q4][q] = 0.f;
  // The image's RoIs are tested against the tile 64 at a time -- lane l takes RoI kb + l's footprint, a ballot gives the ones that reach
  // the tile (the same mask in every wave of the block) -- instead of one scalar load + wait + branch per RoI (512 of them per tile, a
  // fifth of the kernel's time); the hits are then visited in ascending order, as before.
  const int lane_ = threadIdx.x & 63;
  for (int kb = k0; kb < k1; kb += 64) {
    bool hit = false;
    if (kb + lane_ < k1) {
      const int* f = fp + 4 * (kb + lane_);
      hit = !(py0 + TS - 1 < f[0] || py0 > f[1] || px0 + TS - 1 < f[2] || px0 > f[3]);
    }
    unsigned long long hits = __ballot(hit);
  while (hits) {
    const int k = kb + __builtin_ctzll(hits);
    hits &= hits - 1;
    const float* ayr = ay + ((long)k * H + py0) * ph;
    const float* axr = ax + ((long)k * W + px0) * pw;
    // Lane l of every wave holds the weights of bin row l / bin column l for the tile's TS pixel rows / columns (2 + 2 loads
    // per thread; the first version loaded all 4*14 weights in every thread and walked all 14 x 14 bins testing for zeros:
    // the kernel was bound by that control flow, not by the 3.3 GB of dY).  A ballot gives the bins that reach the tile;
    // the loops below visit only those, in the same ascending (bin row, bin column) order, reading the block-uniform
    // weights back with v_readlane.
    const int lane = threadIdx.x & 63;
    float wyv[TS], wxv[TS];
    bool nzy = false, nzx = false;
#pragma unroll
    for (int a = 0; a < TS; ++a) {
      wyv[a] = (lane < ph && py0 + a < H) ? ayr[a * ph + lane] : 0.f;
      wxv[a] = (lane < pw && px0 + a < W) ? axr[a * pw + lane] : 0.f;
      nzy |= wyv[a] != 0.f; nzx |= wxv[a] != 0.f;
    }
    unsigned long long ym = __ballot(nzy);
    const unsigned long long xm0 = __ballot(nzx);
    if (xm0 == 0ull) continue;
    while (ym) {
      const int i = __builtin_ctzll(ym);
      ym &= ym - 1;
      float wyi[TS];
#pragma unroll
      for (int a = 0; a < TS; ++a) wyi[a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyv[a]), i));
      unsigned long long xm = xm0;
      while (xm) {
        const int j = __builtin_ctzll(xm);
        xm &= xm - 1;
        float wxj[TS];
#pragma unroll
        for (int a = 0; a < TS; ++a) wxj[a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wxv[a]), j));
        const u32x4* src = (const u32x4*)dy + (((long)k * ph + i) * pw + j) * cch;
        float v[NC][8];
#pragma unroll
        for (int u = 0; u < NC; ++u) {
          const int c = threadIdx.x + u * blockDim.x;
          if (c < cch) Vec<T>::unpack(src[c], v[u]);
        }
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
          for (int b = 0; b < TS; ++b) {
            const float w = wyi[a] * wxj[b];
            if (w == 0.f) continue;                 // block-uniform: pixels this bin does not reach cost one scalar branch
#pragma unroll
            for (int u = 0; u < NC; ++u)
#pragma unroll
              for (int q = 0; q < VEC; ++q) acc[u][a * TS + b][q] += w * v[u][q];
          }
      }
    }
  }
  }
#pragma unroll
  for (int u = 0; u < NC; ++u) {
    const int c = threadIdx.x + u * blockDim.x;
    if (c < cch) {
#pragma unroll
      for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b)
          if (py0 + a < H && px0 + b < W) ((u32x4*)dx)[(((long)n * H + py0 + a) * W + px0 + b) * cch + c] = Vec<T>::pack(acc[u][a * TS + b]);
    }
  }
}

}  // namespace

static int roi_align_forward_impl(const void* x, const float* rois, void* y, void* y_pooled, int* dbg_grid, const float* esc,
                                  const float* ebi, int relu, int N, int C, int H, int W, int K, int ph, int pw, float spatial_scale,
                                  int sampling_ratio, int aligned, int dtype, void* stream, void* y8 = nullptr, const float* q8 = nullptr,
                                  float* amax8 = nullptr) {
  int es = dtype == 0 ? 2 : 4;
  if ((dtype != 0 && dtype != 1) || (C * es) % 16 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || K < 0 || N < 0)
    return CDDMSL_ERR_ARG;
  if ((esc == nullptr) != (ebi == nullptr)) return CDDMSL_ERR_ARG;
  if (K == 0) return CDDMSL_OK;   // empty inputs return correctly-shaped empties (poolers.py:221-224)
  if (!y && !y_pooled) return CDDMSL_ERR_ARG;
  if (y8 && (dtype != 0 || !y)) return CDDMSL_ERR_ARG;
  int cch = C * es / 16;
  if (y_pooled && ((ph & 1) || (pw & 1))) return CDDMSL_ERR_ARG;      // the pooled copy needs whole 2x2 groups of bins
  long grid = (long)K * (y_pooled ? ph / 2 : ph);
  if (grid > 0x7fffffffL) return CDDMSL_ERR_ARG;
  int threads = cch >= 256 ? 256 : ((cch + 63) / 64) * 64;
  hipStream_t st = (hipStream_t)stream;
  // throughput path: one output (the crops, or only their 2x2-pooled map), bf16, every channel chunk in one block, tables that fit
  const char* rows_env = getenv("CDDMSL_ROI_ROWS");            // "0": the tap-by-tap kernel for everything (A/B switch, read per launch)
  const bool rows_off = rows_env && rows_env[0] == '0';
  if (dtype == 0 && !y8 && !dbg_grid && ((y != nullptr) != (y_pooled != nullptr)) && cch <= 256 && !rows_off) {
    if (y) k_roi_align_fwd_rows<1><<<dim3((unsigned)((long)K * ph)), dim3(threads), 0, st>>>((const char*)x, rois, (char*)y, N, H, W, cch, ph, pw, spatial_scale, sampling_ratio, aligned, esc, ebi, relu);
    else k_roi_align_fwd_rows<2><<<dim3((unsigned)((long)K * (ph / 2))), dim3(threads), 0, st>>>((const char*)x, rois, (char*)y_pooled, N, H, W, cch, ph, pw, spatial_scale, sampling_ratio, aligned, esc, ebi, relu);
    return launch_status();
  }
#define CDDMSL_RAF(TT, RPP) k_roi_align_fwd<TT, RPP><<<dim3((unsigned)grid), dim3(threads), 0, st>>>((const char*)x, rois, (char*)y, (char*)y_pooled, dbg_grid, N, H, W, cch, ph, pw, spatial_scale, sampling_ratio, aligned, esc, ebi, relu, (char*)y8, q8, (unsigned*)amax8)
  if (dtype == 0) { if (y_pooled) CDDMSL_RAF(__bf16, 2); else CDDMSL_RAF(__bf16, 1); }
  else { if (y_pooled) CDDMSL_RAF(float, 2); else CDDMSL_RAF(float, 1); }
#undef CDDMSL_RAF
  return launch_status();
}

extern "C" int cddmsl_roi_align_forward(const void* x, const float* rois, void* y, void* y_pooled, int* dbg_grid, int N, int C,
                                        int H, int W, int K, int ph, int pw, float spatial_scale, int sampling_ratio,
                                        int aligned, int dtype, void* stream) {
  if (!y && K > 0) return CDDMSL_ERR_ARG;
  return roi_align_forward_impl(x, rois, y, y_pooled, dbg_grid, nullptr, nullptr, 0, N, C, H, W, K, ph, pw, spatial_scale,
                                sampling_ratio, aligned, dtype, stream);
}

// RoIAlign of a map that already went through a 1x1 convolution: y = relu?(scale[c] * roi_align(x)[..., c] + bias[c]), and / or
// (y NULL allowed) only the 2x2-average-pooled map y_pooled.  Used by the RoI head's first bottleneck, whose conv1 and
// downsample AvgPool2d are moved across the pooling (cddmsl_amd/layers.py RoIStageFn): RoIAlign is linear over pixels, a 1x1
// convolution over channels, so roi_align(x) W = roi_align(x W); the FrozenBN affine and the ReLU follow here.
extern "C" int cddmsl_roi_align_forward_affine(const void* x, const float* rois, void* y, void* y_pooled, const float* scale,
                                               const float* bias, int relu, int N, int C, int H, int W, int K, int ph, int pw,
                                               float spatial_scale, int sampling_ratio, int aligned, int dtype, void* y8,
                                               const float* q8, float* amax8, void* stream) {
  return roi_align_forward_impl(x, rois, y, y_pooled, nullptr, scale, bias, relu, N, C, H, W, K, ph, pw, spatial_scale,
                                sampling_ratio, aligned, dtype, stream, y8, q8, amax8);
}

static int roi_align_backward_impl(const void* dy, const float* rois, const int* roi_start, void* dx, float* ws_ay,
                                   float* ws_ax, int* ws_fp, int N, int C, int H, int W, int K, int ph, int pw,
                                   float spatial_scale, int sampling_ratio, int aligned, int dtype, int fold, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((dtype != 0 && dtype != 1) || (C * es) % 16 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || K < 0 || N < 0)
    return CDDMSL_ERR_ARG;
  if (ph > MAXP || pw > MAXP || ph > 64 || pw > 64) return CDDMSL_ERR_ARG;
  int cch = C * es / 16;
  if (cch > 1024) return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) return CDDMSL_OK;
  if (K > 0) k_roi_tables<<<dim3((unsigned)K), dim3(128), 0, st>>>(rois, ws_ay, ws_ax, ws_fp, K, H, W, ph, pw, spatial_scale, sampling_ratio, aligned, fold);
  int threads = cch >= 256 ? 256 : ((cch + 63) / 64) * 64;
  const int ncpt = (cch + threads - 1) / threads;
  const int ts = 2;                                 // 2x2 tiles (4x4 tiles measured slower both before (6.7 vs 3.0 ms) and after the ballot rewrite (2.75 vs 1.36 ms))
  long grid = (long)N * ((H + ts - 1) / ts) * ((W + ts - 1) / ts);
  if (grid > 0x7fffffffL) return CDDMSL_ERR_ARG;
#define CDDMSL_RAB(TT, NCC, TSS) k_roi_align_bwd<TT, NCC, TSS><<<dim3((unsigned)grid), dim3(threads), 0, st>>>((const char*)dy, ws_ay, ws_ax, ws_fp, roi_start, (char*)dx, H, W, cch, ph, pw)
  if (dtype == 0) { if (ncpt <= 1) CDDMSL_RAB(__bf16, 1, 2); else if (ncpt == 2) CDDMSL_RAB(__bf16, 2, 2); else CDDMSL_RAB(__bf16, 4, 2); }
  else { if (ncpt <= 1) CDDMSL_RAB(float, 1, 2); else if (ncpt == 2) CDDMSL_RAB(float, 2, 2); else CDDMSL_RAB(float, 4, 2); }
#undef CDDMSL_RAB
  return launch_status();
}

// ws_ay: K*H*ph floats, ws_ax: K*W*pw floats, ws_fp: 4*K ints; roi_start: N+1 ints (device)
extern "C" int cddmsl_roi_align_backward(const void* dy, const float* rois, const int* roi_start, void* dx, float* ws_ay,
                                         float* ws_ax, int* ws_fp, int N, int C, int H, int W, int K, int ph, int pw,
                                         float spatial_scale, int sampling_ratio, int aligned, int dtype, void* stream) {
  return roi_align_backward_impl(dy, rois, roi_start, dx, ws_ay, ws_ax, ws_fp, N, C, H, W, K, ph, pw, spatial_scale,
                                 sampling_ratio, aligned, dtype, 1, stream);
}

// Backward of avgpool2(roi_align(x)): dy is the gradient of the POOLED map [K][ph][pw][C] (RoIAlign grid 2 ph x 2 pw);
// equal to cddmsl_roi_align_backward applied to the AvgPool2d(2) backward of dy, without forming that tensor.
extern "C" int cddmsl_roi_align_backward_pooled(const void* dy, const float* rois, const int* roi_start, void* dx, float* ws_ay,
                                                float* ws_ax, int* ws_fp, int N, int C, int H, int W, int K, int ph, int pw,
                                                float spatial_scale, int sampling_ratio, int aligned, int dtype, void* stream) {
  return roi_align_backward_impl(dy, rois, roi_start, dx, ws_ay, ws_ax, ws_fp, N, C, H, W, K, ph, pw, spatial_scale,
                                 sampling_ratio, aligned, dtype, 2, stream);
}

// ------------------------------------------------------------------------------------------------------------------------
// torchvision.ops.roi_align as the reference calls it (layers/roi_align.py:58-65): input [N][C][H][W], rois [K][5] in ANY order
// -> output [K][C][ph][pw]; backward: grad [K][C][ph][pw] -> grad_input [N][C][H][W].  The kernels above are channels-last and
// the gather backward walks the RoIs of one image: these entry points re-lay the operands in caller-provided scratch (channels
// zero-padded to whole 16-byte chunks, RoIs ranked stably by image for the backward) and call them -- same arithmetic, so the
// results equal the channels-last entry points' element for element.
namespace {
template <typename T>
__global__ void k_nchw_to_nhwc(const T* x, T* y, long N, int C, long HW, int Cp) {       // y [N][HW][Cp], pad channels zero
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * HW * Cp) return;
  const int c = (int)(i % Cp);
  const long p = (i / Cp) % HW, n = i / ((long)Cp * HW);
  y[i] = c < C ? x[(n * C + c) * HW + p] : (T)0.f;
}
// out [K][C][HW] <- in [perm ? perm[k] : k][HW][Cp]   (perm: position of RoI k in the image-ranked order)
template <typename T>
__global__ void k_nhwc_to_nchw(const T* x, T* y, long K, int C, long HW, int Cp, const int* perm) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * C * HW) return;
  const long p = i % HW, k = i / ((long)C * HW);
  const int c = (int)((i / HW) % C);
  const long ks = perm ? perm[k] : k;
  y[i] = x[(ks * HW + p) * Cp + c];
}
// in [K][C][HW] (rows in the caller's order) -> out [rank[k]][HW][Cp]
template <typename T>
__global__ void k_nchw_rows_to_ranked_nhwc(const T* x, T* y, long K, int C, long HW, int Cp, const int* rank) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * HW * Cp) return;
  const int c = (int)(i % Cp);
  const long p = (i / Cp) % HW, k = i / ((long)Cp * HW);
  y[((long)rank[k] * HW + p) * Cp + c] = c < C ? x[(k * C + c) * HW + p] : (T)0.f;
}
// stable rank of every RoI by image index (ties keep the caller's order), the ranked RoI list and roi_start [N+1].
// K is a few thousand: one thread per RoI counting its predecessors is a fraction of the gather's time.
__global__ void k_rank_rois(const float* rois, int* rank, float* sorted, int* roi_start, int K, int N) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k <= N) {                                    // roi_start[n] = number of RoIs of images < n (out-of-range images sort last)
    int cnt = 0;
    for (int j = 0; j < K; ++j) { const int b = (int)rois[5 * (long)j]; cnt += (b >= 0 && b < k) ? 1 : 0; }
    roi_start[k] = cnt;
  }
  if (k >= K) return;
  const int bk = (int)rois[5 * (long)k];
  const long key = (bk >= 0 && bk < N) ? bk : N;
  int r = 0;
  for (int j = 0; j < K; ++j) {
    const int bj = (int)rois[5 * (long)j];
    const long kj = (bj >= 0 && bj < N) ? bj : N;
    r += (kj < key || (kj == key && j < k)) ? 1 : 0;
  }
  rank[k] = r;
#pragma unroll
  for (int q = 0; q < 5; ++q) sorted[5 * (long)r + q] = rois[5 * (long)k + q];
}
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace

extern "C" int cddmsl_roi_align_nchw_anyorder(const void* input, const float* rois, void* output, int N, int C, int H, int W, int K,
                                              int ph, int pw, float spatial_scale, int sampling_ratio, int aligned, int dtype,
                                              void* temp, size_t* temp_bytes, void* stream) {
  if ((dtype != 0 && dtype != 1) || !temp_bytes || N < 0 || C <= 0 || H <= 0 || W <= 0 || K < 0 || ph <= 0 || pw <= 0) return CDDMSL_ERR_ARG;
  const int es = dtype == 0 ? 2 : 4, per = 16 / es;
  const int Cp = (C + per - 1) / per * per;
  const size_t o_x = 0, o_y = al256((size_t)N * H * W * Cp * es), total = o_y + al256((size_t)K * ph * pw * Cp * es);
  if (!temp) { *temp_bytes = total; return CDDMSL_OK; }
  if (*temp_bytes < total) return CDDMSL_ERR_ARG;
  if (K == 0 || N == 0) return CDDMSL_OK;          // correctly-shaped empties (poolers.py:221-224): nothing to write
  hipStream_t st = (hipStream_t)stream;
  char* t = (char*)temp;
  const long nx = (long)N * H * W * Cp, ny = (long)K * C * ph * pw;
  if ((nx + 255) / 256 > 0x7fffffffL || (ny + 255) / 256 > 0x7fffffffL) return CDDMSL_ERR_ARG;
  if (dtype == 0) k_nchw_to_nhwc<__bf16><<<dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st>>>((const __bf16*)input, (__bf16*)(t + o_x), N, C, (long)H * W, Cp);
  else k_nchw_to_nhwc<float><<<dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st>>>((const float*)input, (float*)(t + o_x), N, C, (long)H * W, Cp);
  int rc = cddmsl_roi_align_forward(t + o_x, rois, t + o_y, nullptr, nullptr, N, Cp, H, W, K, ph, pw, spatial_scale, sampling_ratio, aligned, dtype, stream);
  if (rc != CDDMSL_OK) return rc;
  if (dtype == 0) k_nhwc_to_nchw<__bf16><<<dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, st>>>((const __bf16*)(t + o_y), (__bf16*)output, K, C, (long)ph * pw, Cp, nullptr);
  else k_nhwc_to_nchw<float><<<dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, st>>>((const float*)(t + o_y), (float*)output, K, C, (long)ph * pw, Cp, nullptr);
  return launch_status();
}

extern "C" int cddmsl_roi_align_backward_nchw_anyorder(const void* grad, const float* rois, void* grad_input, int N, int C, int H, int W,
                                                       int K, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                                                       int dtype, void* temp, size_t* temp_bytes, void* stream) {
  if ((dtype != 0 && dtype != 1) || !temp_bytes || N < 0 || C <= 0 || H <= 0 || W <= 0 || K < 0 || ph <= 0 || pw <= 0) return CDDMSL_ERR_ARG;
  const int es = dtype == 0 ? 2 : 4, per = 16 / es;
  const int Cp = (C + per - 1) / per * per;
  const size_t o_dy = 0, o_dx = o_dy + al256((size_t)K * ph * pw * Cp * es), o_ay = o_dx + al256((size_t)N * H * W * Cp * es),
               o_ax = o_ay + al256((size_t)K * H * ph * 4), o_fp = o_ax + al256((size_t)K * W * pw * 4), o_rk = o_fp + al256((size_t)K * 16),
               o_rs = o_rk + al256((size_t)K * 4), o_st = o_rs + al256((size_t)K * 20), total = o_st + al256((size_t)(N + 1) * 4);
  if (!temp) { *temp_bytes = total; return CDDMSL_OK; }
  if (*temp_bytes < total) return CDDMSL_ERR_ARG;
  if (N == 0) return CDDMSL_OK;
  hipStream_t st = (hipStream_t)stream;
  char* t = (char*)temp;
  const long nd = (long)K * ph * pw * Cp, nx = (long)N * C * H * W;
  if ((nd + 255) / 256 > 0x7fffffffL || (nx + 255) / 256 > 0x7fffffffL) return CDDMSL_ERR_ARG;
  const int nrk = (K > N + 1 ? K : N + 1);
  k_rank_rois<<<dim3((unsigned)((nrk + 127) / 128)), dim3(128), 0, st>>>(rois, (int*)(t + o_rk), (float*)(t + o_rs), (int*)(t + o_st), K, N);
  if (K > 0) {
    if (dtype == 0) k_nchw_rows_to_ranked_nhwc<__bf16><<<dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st>>>((const __bf16*)grad, (__bf16*)(t + o_dy), K, C, (long)ph * pw, Cp, (const int*)(t + o_rk));
    else k_nchw_rows_to_ranked_nhwc<float><<<dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st>>>((const float*)grad, (float*)(t + o_dy), K, C, (long)ph * pw, Cp, (const int*)(t + o_rk));
  }
  // RoIs naming an image outside [0, N) rank last, past roi_start[N]: the gather never visits them (the forward writes zeros there)
  int rc = cddmsl_roi_align_backward(t + o_dy, (const float*)(t + o_rs), (const int*)(t + o_st), t + o_dx, (float*)(t + o_ay), (float*)(t + o_ax),
                                     (int*)(t + o_fp), N, Cp, H, W, K, ph, pw, spatial_scale, sampling_ratio, aligned, dtype, stream);
  if (rc != CDDMSL_OK) return rc;
  if (dtype == 0) k_nhwc_to_nchw<__bf16><<<dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st>>>((const __bf16*)(t + o_dx), (__bf16*)grad_input, N, C, (long)H * W, Cp, nullptr);
  else k_nhwc_to_nchw<float><<<dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st>>>((const float*)(t + o_dx), (float*)grad_input, N, C, (long)H * W, Cp, nullptr);
  return launch_status();
}
