// HBM-bound elementwise / pooling kernels of the CDDMSL hot path (gfx950, wave64).
//
//   preprocess      GeneralizedRCNN.preprocess_image  detectron2/modeling/meta_arch/rcnn.py:758-768
//                   + ImageList.from_tensors zero pad  detectron2/structures/image_list.py:72-124
//                   u8 CHW image -> (x/255 - mean)/std -> NHWC T, channel-padded, zero outside the image
//   preprocess224   preprocess_image_train            rcnn.py:161-179 (x/255 -> pad -> bicubic short side
//                   224, align_corners=False, no antialias -> center crop 224 -> normalise)
//   avgpool2        nn.AvgPool2d(2) (floor)           clip_backbone.py:36,46,147 forward/backward
//   attn tokens     AttentionPool2d token build        clip_backbone.py:84-86 forward/backward
//   sgd_clip        per-parameter grad-norm clip + SGD detectron2/solver/build.py:59-67,104,113-130
// All kernels read/write 16 B per lane where the layout allows (channels contiguous).
#include "common.h"
#include <stdlib.h>

namespace {

template <typename T> struct Elt;
template <> struct Elt<__bf16> {
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
  __device__ static __forceinline__ float ld(const char* p) { return bf2f(*(const unsigned short*)p); }
  __device__ static __forceinline__ void st(char* p, float v) { *(unsigned short*)p = f2bf(v); }
};
template <> struct Elt<float> {
  static constexpr int ES = 4, VEC = 4;
  __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
    const f32x4 x = __builtin_bit_cast(f32x4, v);
    f[0] = x[0]; f[1] = x[1]; f[2] = x[2]; f[3] = x[3];
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    f32x4 x = {f[0], f[1], f[2], f[3]};
    return __builtin_bit_cast(u32x4, x);
  }
  __device__ static __forceinline__ float ld(const char* p) { return *(const float*)p; }
  __device__ static __forceinline__ void st(char* p, float v) { *(float*)p = v; }
};

// ---------------------------------------------------------------- preprocess (full resolution)
template <typename T>
__device__ __forceinline__ void preprocess_px(const unsigned char* img, char* out, int n, int h, int w, int Hp, int Wp, int Cp,
                                              float m0, float m1, float m2, float s0, float s1, float s2, float div) {
  // one thread per padded output pixel of image n; out[n][y][x][0..Cp)
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Hp * Wp) return;
  const unsigned ii = (unsigned)i;                 // (Hp * Wp < 2^31: a 32-bit division)
  int x = ii % (unsigned)Wp, y = ii / (unsigned)Wp;
  float v[3] = {0.f, 0.f, 0.f};
  if (y < h && x < w) {
    long o = (long)y * w + x, pl = (long)h * w;
    v[0] = ((float)img[o] / div - m0) / s0;
    v[1] = ((float)img[pl + o] / div - m1) / s1;
    v[2] = ((float)img[2 * pl + o] / div - m2) / s2;
  }
  char* dst = out + (((long)n * Hp + y) * Wp + x) * Cp * Elt<T>::ES;
  if (Cp == Elt<T>::VEC) {                         // a pixel is one 16-byte chunk (the layout the stem conv reads): one store
    float o[8] = {v[0], v[1], v[2], 0.f, 0.f, 0.f, 0.f, 0.f};
    *(u32x4*)dst = Elt<T>::pack(o);
    return;
  }
  for (int c = 0; c < Cp; ++c) Elt<T>::st(dst + c * Elt<T>::ES, c < 3 ? v[c] : 0.f);
}
template <typename T>
__global__ void k_preprocess(const unsigned char* img, char* out, int n, int h, int w, int Hp, int Wp, int Cp,
                             float m0, float m1, float m2, float s0, float s1, float s2, float div) {
  preprocess_px<T>(img, out, n, h, w, Hp, Wp, Cp, m0, m1, m2, s0, s1, s2, div);
}
// up to PRE_MAX images of a batch in ONE launch (blockIdx.y = image): the images are separate tensors of different sizes, so their
// pointers and sizes ride in the kernel arguments -- 32 launches of ~8 us per step were 0.51 ms for work that streams in ~0.15
constexpr int PRE_MAX = 32;
struct PreBatch { const unsigned char* img[PRE_MAX]; int h[PRE_MAX]; int w[PRE_MAX]; };
template <typename T>
__global__ void k_preprocess_batch(PreBatch b, char* out, int n0, int Hp, int Wp, int Cp,
                                   float m0, float m1, float m2, float s0, float s1, float s2, float div) {
  const int j = blockIdx.y;
  preprocess_px<T>(b.img[j], out, n0 + j, b.h[j], b.w[j], Hp, Wp, Cp, m0, m1, m2, s0, s1, s2, div);
}

// ---------------------------------------------------------------- 224 bicubic preprocess
// ATen upsample_bicubic2d (align_corners=False): src = (dst+0.5)*scale-0.5, A=-0.75, taps clamped.
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

template <typename T>
__device__ __forceinline__ void preprocess224_px(const unsigned char* img, char* out, int n, int h, int w, int Hp, int Wp,
                                                 int RH, int RW, int top, int left, int S, int Cp,
                                                 float m0, float m1, float m2, float s0, float s1, float s2) {
  // (h,w): this image; (Hp,Wp): padded batch size (zero outside the image); (RH,RW): resized padded size;
  // crop window (top,left,S,S).  One thread per output pixel.
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * S) return;
  int ox = i % S, oy = i / S;
  float sy = (float)Hp / (float)RH, sx = (float)Wp / (float)RW;
  float fy = ((float)(oy + top) + 0.5f) * sy - 0.5f, fx = ((float)(ox + left) + 0.5f) * sx - 0.5f;
  int iy = (int)floorf(fy), ix = (int)floorf(fx);
  float ty = fy - iy, tx = fx - ix;
  const float A = -0.75f;
  float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
  float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
  float acc[3] = {0.f, 0.f, 0.f};
  long pl = (long)h * w;
  for (int a = 0; a < 4; ++a) {
    int yy = min(max(iy - 1 + a, 0), Hp - 1);
    float r[3] = {0.f, 0.f, 0.f};
    for (int b = 0; b < 4; ++b) {
      int xx = min(max(ix - 1 + b, 0), Wp - 1);
      if (yy < h && xx < w) {
        long o = (long)yy * w + xx;
        r[0] += wx[b] * ((float)img[o] / 255.0f);
        r[1] += wx[b] * ((float)img[pl + o] / 255.0f);
        r[2] += wx[b] * ((float)img[2 * pl + o] / 255.0f);
      }
    }
    acc[0] += wy[a] * r[0]; acc[1] += wy[a] * r[1]; acc[2] += wy[a] * r[2];
  }
  float v[3] = {(acc[0] - m0) / s0, (acc[1] - m1) / s1, (acc[2] - m2) / s2};
  char* dst = out + (((long)n * S + oy) * S + ox) * Cp * Elt<T>::ES;
  if (Cp == Elt<T>::VEC) {
    float o[8] = {v[0], v[1], v[2], 0.f, 0.f, 0.f, 0.f, 0.f};
    *(u32x4*)dst = Elt<T>::pack(o);
    return;
  }
  for (int c = 0; c < Cp; ++c) Elt<T>::st(dst + c * Elt<T>::ES, c < 3 ? v[c] : 0.f);
}
template <typename T>
__global__ void k_preprocess224(const unsigned char* img, char* out, int n, int h, int w, int Hp, int Wp,
                                int RH, int RW, int top, int left, int S, int Cp,
                                float m0, float m1, float m2, float s0, float s1, float s2) {
  preprocess224_px<T>(img, out, n, h, w, Hp, Wp, RH, RW, top, left, S, Cp, m0, m1, m2, s0, s1, s2);
}
template <typename T>
__global__ void k_preprocess224_batch(PreBatch b, char* out, int n0, int Hp, int Wp, int RH, int RW, int top, int left, int S, int Cp,
                                      float m0, float m1, float m2, float s0, float s1, float s2) {
  const int j = blockIdx.y;
  preprocess224_px<T>(b.img[j], out, n0 + j, b.h[j], b.w[j], Hp, Wp, RH, RW, top, left, S, Cp, m0, m1, m2, s0, s1, s2);
}

// ---------------------------------------------------------------- avgpool 2x2 (NHWC)
template <typename T>
__global__ void k_avgpool2_fwd(const char* x, char* y, int N, int H, int W, int cch) {
  // cch = 16-B chunks per pixel; one thread per output chunk
  int Ho = H / 2, Wo = W / 2;
  long total = (long)N * Ho * Wo * cch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cch; long q = i / cch;
    int ox = q % Wo; q /= Wo;
    int oy = q % Ho; int n = q / Ho;
    const u32x4* b = (const u32x4*)x + (((long)n * H + 2 * oy) * W + 2 * ox) * cch + c;
    float a0[8], a1[8], a2[8], a3[8], o[8];
    Elt<T>::unpack(b[0], a0); Elt<T>::unpack(b[cch], a1);
    Elt<T>::unpack(b[(long)W * cch], a2); Elt<T>::unpack(b[(long)W * cch + cch], a3);
#pragma unroll
    for (int j = 0; j < Elt<T>::VEC; ++j) o[j] = ((a0[j] + a1[j]) + (a2[j] + a3[j])) * 0.25f;
    ((u32x4*)y)[i] = Elt<T>::pack(o);
  }
}

// dx[n][y][x] = (dy[n][y/2][x/2] / 4) (+ add[n][y][x]) masked by (mask > 0); zero on the floor-dropped row/col
template <typename T>
__global__ void k_avgpool2_bwd(const char* dy, const char* mask, const char* add, char* dx, int N, int H, int W, int cch) {
  int Ho = H / 2, Wo = W / 2;
  long total = (long)N * H * W * cch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cch; long q = i / cch;
    int xx = q % W; q /= W;
    int yy = q % H; int n = q / H;
    float g[8], m[8], a[8], o[8];
    bool in = (yy >> 1) < Ho && (xx >> 1) < Wo;
    if (in) Elt<T>::unpack(((const u32x4*)dy)[(((long)n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * cch + c], g);
    if (mask) Elt<T>::unpack(((const u32x4*)mask)[i], m);
    if (add) Elt<T>::unpack(((const u32x4*)add)[i], a);
#pragma unroll
    for (int j = 0; j < Elt<T>::VEC; ++j) {
      float v = in ? g[j] * 0.25f : 0.f;
      if (add) v += a[j];
      if (mask && !(m[j] > 0.f)) v = 0.f;
      o[j] = v;
    }
    ((u32x4*)dx)[i] = Elt<T>::pack(o);
  }
}

// fp8 configuration: the same pass (bf16) also writes the e4m3 copy of dx for the GEMMs that consume it -- y8 = sat(dx * q8[0]), from
// the f32 value -- and max-es |dx| into amax8[block & 63] (delayed scaling; see fp8.hip)
__global__ __launch_bounds__(256) void k_avgpool2_bwd_q8(const char* dy, const char* mask, const char* add, char* dx, int N, int H, int W, int cch,
                                                         char* y8, const float* q8, unsigned* amax8) {
  using T = __bf16;
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)N * H * W * cch;
  const float s = q8[0];
  unsigned mx = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cch; long q = i / cch;
    int xx = q % W; q /= W;
    int yy = q % H; int n = q / H;
    float g[8], m[8], a[8], o[8];
    bool in = (yy >> 1) < Ho && (xx >> 1) < Wo;
    if (in) Elt<T>::unpack(((const u32x4*)dy)[(((long)n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * cch + c], g);
    if (mask) Elt<T>::unpack(((const u32x4*)mask)[i], m);
    if (add) Elt<T>::unpack(((const u32x4*)add)[i], a);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = in ? g[j] * 0.25f : 0.f;
      if (add) v += a[j];
      if (mask && !(m[j] > 0.f)) v = 0.f;
      o[j] = v;
      mx = absmax_bits(mx, v);
    }
    ((u32x4*)dx)[i] = Elt<T>::pack(o);
    const u32x2 o8 = {e4m3x4(o[0] * s, o[1] * s, o[2] * s, o[3] * s), e4m3x4(o[4] * s, o[5] * s, o[6] * s, o[7] * s)};
    ((u32x2*)y8)[i] = o8;
  }
  __shared__ unsigned sm[4];
  mx = wave_max_u(mx);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
    atomicMax(amax8 + (blockIdx.x & 63), a > b ? a : b);
  }
}

// ---------------------------------------------------------------- stock ResNet pieces (config #1: detectron2 R50-C4)
// F.max_pool2d(x, 3, stride 2, padding 1)  (BasicStem, modeling/backbone/resnet.py:355-358), NHWC, forward only (stem frozen)
template <typename T>
__global__ void k_maxpool3s2(const char* x, char* y, int N, int H, int W, int Ho, int Wo, int cch) {
  long total = (long)N * Ho * Wo * cch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cch; long q = i / cch;
    int ox = q % Wo; q /= Wo;
    int oy = q % Ho; int n = q / Ho;
    float m[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      int iy = 2 * oy - 1 + dy;
      if (iy < 0 || iy >= H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        int ix = 2 * ox - 1 + dx;
        if (ix < 0 || ix >= W) continue;
        Elt<T>::unpack(((const u32x4*)x)[(((long)n * H + iy) * W + ix) * cch + c], v);
#pragma unroll
        for (int j = 0; j < Elt<T>::VEC; ++j) m[j] = fmaxf(m[j], v[j]);
      }
    }
    ((u32x4*)y)[i] = Elt<T>::pack(m);
  }
}
// backward of a stride-2 1x1 convolution's input gather: dx[n][y][x] = t[n][y/2][x/2] on even (y,x), 0 elsewhere,
// (+ add) then masked by (mask > 0)   (BottleneckBlock conv1 / shortcut with STRIDE_IN_1X1, resnet.py:100-210)
template <typename T>
__global__ void k_upsample_zero2(const char* t, const char* mask, const char* add, char* dx, int N, int H, int W, int Ho, int Wo, int cch) {
  long total = (long)N * H * W * cch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cch; long q = i / cch;
    int xx = q % W; q /= W;
    int yy = q % H; int n = q / H;
    float g[8], m[8], a[8], o[8];
    bool in = !(yy & 1) && !(xx & 1) && (yy >> 1) < Ho && (xx >> 1) < Wo;
    if (in) Elt<T>::unpack(((const u32x4*)t)[(((long)n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * cch + c], g);
    if (mask) Elt<T>::unpack(((const u32x4*)mask)[i], m);
    if (add) Elt<T>::unpack(((const u32x4*)add)[i], a);
#pragma unroll
    for (int j = 0; j < Elt<T>::VEC; ++j) {
      float v = in ? g[j] : 0.f;
      if (add) v += a[j];
      if (mask && !(m[j] > 0.f)) v = 0.f;
      o[j] = v;
    }
    ((u32x4*)dx)[i] = Elt<T>::pack(o);
  }
}
// x [K][P][C] -> mean over P (f32 out); backward broadcasts dy/P  (Res5ROIHeads box_features.mean(dim=[2,3]), roi_heads.py:487)
template <typename T>
__global__ void k_meanpool_fwd(const char* x, float* y, long K, int P, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * C) return;
  long k = i / C; int c = i % C;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += Elt<T>::ld(x + ((k * P + p) * C + c) * Elt<T>::ES);
  y[i] = s / (float)P;
}
template <typename T>
__global__ void k_meanpool_bwd(const float* dy, char* dx, long K, int P, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * P * C) return;
  int c = i % C; long k = i / ((long)P * C);
  Elt<T>::st(dx + i * Elt<T>::ES, dy[k * C + c] / (float)P);
}

// ---------------------------------------------------------------- attention-pool tokens
// x [K][P][C] (P = 49 pixels) -> tok [K][TP][C] (TP >= P+1 token rows per region in memory; rows P+1.. are zero):
// tok0 = mean_p x + pos[0], tok_{i+1} = x_i + pos[i+1]
// mbits (nullable, P <= 64): one 64-bit word per region and column, bit p = (x[k][p][col] > 0) -- the ReLU mask of the pooled map
// in the form the backward's fused epilogue reads (cddmsl_attnpool_dx): 8 bytes per column instead of P elements
template <typename T>
__global__ void k_attn_tokens_fwd(const char* x, const float* pos, char* tok, unsigned long long* mbits, int K, int P, int TP, int cch) {
  long total = (long)K * cch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = i % cch; long k = i / cch;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8], o[8];
    unsigned long long mb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int VEC = Elt<T>::VEC;
    for (int p = 0; p < P; ++p) {
      Elt<T>::unpack(((const u32x4*)x)[(k * P + p) * cch + c], v);
      const float* pe = pos + (long)(p + 1) * cch * VEC + c * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s[j] += v[j]; o[j] = v[j] + pe[j]; mb[j] |= (unsigned long long)(v[j] > 0.f) << p; }
      ((u32x4*)tok)[(k * TP + p + 1) * cch + c] = Elt<T>::pack(o);
    }
    if (mbits) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) mbits[(k * cch + c) * VEC + j] = mb[j];
    }
    const float* pe = pos + c * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = s[j] / (float)P + pe[j];
    ((u32x4*)tok)[(k * TP) * cch + c] = Elt<T>::pack(o);
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int p = P + 1; p < TP; ++p) ((u32x4*)tok)[(k * TP + p) * cch + c] = z;
  }
}
// dx[k][p] = dtok[k][p+1] + dtok[k][0]/P, zeroed where relu_mask[k][p] <= 0 (relu_mask = the pooled map itself when it is the
// output of a ReLU: the ReLU backward of the residual stage in front of the pool rides in this pass), and -- gpos given --
// gpos[t][:] += sum_k dtok[k][t][:] for t <= P (the positional embedding's gradient: dtok is read once for both).
// grid: x = 256-chunk column blocks, y = token row t, z = slab of regions; a thread owns one 16-byte column chunk of one
// token row, walks its slab's regions and ends with one atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void k_attn_tokens_bwd(const char* __restrict__ dtok, const char* __restrict__ relu_mask, char* __restrict__ dx, float* __restrict__ gpos, int K,
                                                         int P, int TP, int cch, int slab) {
  constexpr int V = Elt<T>::VEC;
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int t = blockIdx.y;
  if (c >= cch || (t == 0 && !gpos)) return;
  const int k0 = blockIdx.z * slab, k1 = min(K, k0 + slab);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  const u32x4* src = (const u32x4*)dtok;
#pragma unroll 4
  for (int k = k0; k < k1; ++k) {
    float a[8];
    Elt<T>::unpack(src[((long)k * TP + t) * cch + c], a);
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] += a[j];
    if (t > 0 && dx) {
      float b[8], m[8], o[8];
      const long oi = ((long)k * P + t - 1) * cch + c;
      Elt<T>::unpack(src[(long)k * TP * cch + c], b);
      if (relu_mask) Elt<T>::unpack(((const u32x4*)relu_mask)[oi], m);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        o[j] = a[j] + b[j] / (float)P;
        if (relu_mask && !(m[j] > 0.f)) o[j] = 0.f;
      }
      ((u32x4*)dx)[oi] = Elt<T>::pack(o);
    }
  }
  if (gpos) {
#pragma unroll
    for (int j = 0; j < V; ++j) atomicAdd(gpos + (long)t * cch * V + (long)c * V + j, acc[j]);
  }
}

// Softmax glue of the reassociated attention pool (layers.py::AttnPoolFn), one thread per (region, head) row of <= TP scores:
// forward  p = softmax(S[:P1] * scale) (f32, saved) and its transposed compute-dtype copy pT[k][t][h] (zero for t >= P1);
// backward ds = p * (dP - sum_t p dP) * scale, written as dsT[k][t][h] and, stacked under p, as pds[k][h | H+h][t] (zero pads).
template <typename T>
__global__ void k_attnpool_softmax_fwd(const float* S, float* p, char* pT, long rows, int H, int P1, int TP, float scale) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;     // r = k * H + h
  if (r >= rows) return;
  const long k = r / H; const int h = (int)(r - k * H);
  const float* s = S + r * TP;
  float mx = -INFINITY;
  for (int t = 0; t < P1; ++t) mx = fmaxf(mx, s[t] * scale);
  float sum = 0.f;
  for (int t = 0; t < P1; ++t) sum += expf(s[t] * scale - mx);
  const float inv = 1.f / sum;
  for (int t = 0; t < TP; ++t) {
    float v = 0.f;
    if (t < P1) { v = expf(s[t] * scale - mx) * inv; p[r * P1 + t] = v; }
    Elt<T>::st(pT + ((k * TP + t) * H + h) * Elt<T>::ES, v);
  }
}
template <typename T>
__global__ void k_attnpool_softmax_bwd(const float* p, const float* dP, char* dsT, char* pds, long rows, int H, int P1, int TP,
                                       float scale) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const long k = r / H; const int h = (int)(r - k * H);
  const float* pr = p + r * P1;
  const float* dp = dP + r * TP;
  float dot = 0.f;
  for (int t = 0; t < P1; ++t) dot += pr[t] * dp[t];
  char* prow = pds + ((k * 2 * H + h) * TP) * Elt<T>::ES;
  char* drow = pds + ((k * 2 * H + H + h) * TP) * Elt<T>::ES;
  for (int t = 0; t < TP; ++t) {
    float pv = 0.f, ds = 0.f;
    if (t < P1) { pv = pr[t]; ds = pv * (dp[t] - dot) * scale; }
    Elt<T>::st(dsT + ((k * TP + t) * H + h) * Elt<T>::ES, ds);
    Elt<T>::st(prow + t * Elt<T>::ES, pv);
    Elt<T>::st(drow + t * Elt<T>::ES, ds);
  }
}

// The same with one WAVE per region (H <= 32 heads, TP <= 64 token slots): lane t holds column t of every head's row, so the
// score rows are read coalesced, the row reductions are wave reductions, and the transposed outputs leave as H consecutive
// elements per lane (the one-thread-per-row form above walks 224-byte-strided rows: 1.7 ms/step against 0.1 here).
template <typename T>
__global__ __launch_bounds__(256) void k_attnpool_softmax_fwd_w(const float* __restrict__ S, float* __restrict__ p, char* __restrict__ pT,
                                                                long K, int H, int P1, int TP, float scale) {
  const long k = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = threadIdx.x & 63;
  if (k >= K) return;
  float v[32];
#pragma unroll
  for (int h = 0; h < 32; ++h) {
    v[h] = 0.f;
    if (h < H) {
      const bool in = t < P1;
      const float x = in ? S[(k * H + h) * TP + t] * scale : -INFINITY;
      const float mx = wave_max(x);
      const float e = in ? expf(x - mx) : 0.f;
      const float sum = wave_sum(e);
      v[h] = e * (1.f / sum);
      if (in) p[(k * H + h) * P1 + t] = v[h];
    }
  }
  if (t < TP) {
    char* dst = pT + ((k * TP + t) * H) * Elt<T>::ES;
    if (H % Elt<T>::VEC == 0) {                    // H consecutive elements of this lane: 16-byte stores
#pragma unroll
      for (int g = 0; g < 32 / Elt<T>::VEC; ++g)
        if (g * Elt<T>::VEC < H) ((u32x4*)dst)[g] = Elt<T>::pack(v + g * Elt<T>::VEC);
    } else {
#pragma unroll
      for (int h = 0; h < 32; ++h)
        if (h < H) Elt<T>::st(dst + h * Elt<T>::ES, v[h]);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_attnpool_softmax_bwd_w(const float* __restrict__ p, const float* __restrict__ dP,
                                                                char* __restrict__ dsT, char* __restrict__ pds, long K, int H, int P1,
                                                                int TP, float scale) {
  const long k = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = threadIdx.x & 63;
  if (k >= K) return;
  float dsv[32];
#pragma unroll
  for (int h = 0; h < 32; ++h) {
    dsv[h] = 0.f;
    if (h < H) {
      const bool in = t < P1;
      const float pv = in ? p[(k * H + h) * P1 + t] : 0.f;
      const float dp = in ? dP[(k * H + h) * TP + t] : 0.f;
      const float dot = wave_sum(pv * dp);
      dsv[h] = in ? pv * (dp - dot) * scale : 0.f;
      if (t < TP) {
        Elt<T>::st(pds + ((k * 2 * H + h) * TP + t) * Elt<T>::ES, pv);
        Elt<T>::st(pds + ((k * 2 * H + H + h) * TP + t) * Elt<T>::ES, dsv[h]);
      }
    }
  }
  if (t < TP) {
    char* dst = dsT + ((k * TP + t) * H) * Elt<T>::ES;
    if (H % Elt<T>::VEC == 0) {
#pragma unroll
      for (int g = 0; g < 32 / Elt<T>::VEC; ++g)
        if (g * Elt<T>::VEC < H) ((u32x4*)dst)[g] = Elt<T>::pack(dsv + g * Elt<T>::VEC);
    } else {
#pragma unroll
      for (int h = 0; h < 32; ++h)
        if (h < H) Elt<T>::st(dst + h * Elt<T>::ES, dsv[h]);
    }
  }
}

// ReLU backward: dx = g where y > 0 else 0 (optionally g given in f32 while y/dx are T)
template <typename T>
__global__ void k_relu_bwd(const char* g, const char* y, char* dx, long nchunks, int g_f32) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float a[8], m[8], o[8];
    Elt<T>::unpack(((const u32x4*)y)[i], m);
    if (g_f32 && Elt<T>::VEC == 8) {
      Elt<float>::unpack(((const u32x4*)g)[2 * i], a);
      Elt<float>::unpack(((const u32x4*)g)[2 * i + 1], a + 4);
    } else Elt<T>::unpack(((const u32x4*)g)[i], a);
#pragma unroll
    for (int j = 0; j < Elt<T>::VEC; ++j) o[j] = m[j] > 0.f ? a[j] : 0.f;
    ((u32x4*)dx)[i] = Elt<T>::pack(o);
  }
}

// column sums: out[(r % period)][c] (f32) += sum_r x[r][c]  (bias grads; positional-embedding grads with period 50).
// grid.x = 64-column blocks, grid.y = row slabs, grid.z = residue (period); each thread walks the rows of its residue
// inside the slab and issues ONE atomic.
template <typename T>
__global__ void k_colsum(const char* x, float* out, long rows, int cols, int period, int slab) {
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int rlane = threadIdx.x >> 6;           // 4 row lanes
  int res = blockIdx.z;
  long j0 = (long)blockIdx.y * slab;      // index within the residue class: row = res + period * j
  long nj = (rows - res + period - 1) / period;
  float s = 0.f;
  if (c < cols)
    for (long j = j0 + rlane; j < min(nj, j0 + slab); j += 4) s += Elt<T>::ld(x + ((res + period * j) * cols + c) * Elt<T>::ES);
  __shared__ float red[4][64];
  red[rlane][threadIdx.x & 63] = s;
  __syncthreads();
  if (rlane == 0 && c < cols) atomicAdd(out + (long)res * cols + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the same with 16-byte loads (rows that are whole chunks): a wave reads 1 KiB of a row per instruction instead of 128 B
template <typename T>
__global__ void k_colsum_vec(const char* x, float* out, long rows, int cols, int period, int slab) {
  constexpr int VEC = Elt<T>::VEC;
  const int cch = cols / VEC;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);   // chunk column
  const int rlane = threadIdx.x >> 6;
  const int res = blockIdx.z;
  const long j0 = (long)blockIdx.y * slab;
  const long nj = (rows - res + period - 1) / period;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cch)
    for (long j = j0 + rlane; j < min(nj, j0 + slab); j += 4) {
      float v[8];
      Elt<T>::unpack(((const u32x4*)x)[(res + period * j) * cch + c], v);
#pragma unroll
      for (int q = 0; q < VEC; ++q) s[q] += v[q];
    }
  __shared__ float red[4][64][8];
#pragma unroll
  for (int q = 0; q < VEC; ++q) red[rlane][threadIdx.x & 63][q] = s[q];
  __syncthreads();
  if (rlane == 0 && c < cch) {
#pragma unroll
    for (int q = 0; q < VEC; ++q)
      atomicAdd(out + (long)res * cols + c * VEC + q, red[0][threadIdx.x][q] + red[1][threadIdx.x][q] + red[2][threadIdx.x][q] + red[3][threadIdx.x][q]);
  }
}

// ---------------------------------------------------------------- fused clip + SGD (multi-tensor, 2 passes)
struct SgdItem { float* p; const float* g; float* m; long n; };
constexpr int SGD_MAX = 96;
struct SgdBatch { SgdItem it[SGD_MAX]; int count; };

__global__ void k_sqnorm(SgdBatch b, float* norms) {
  // grid.y = tensor, grid.x = slab; 16-byte accesses when the gradient view is 16-byte aligned (views into the flat gradient
  // buffer start wherever the tensors before them end), scalar otherwise and for the last n % 4 elements
  const SgdItem& t = b.it[blockIdx.y];
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  float s = 0.f;
  long done = 0;
  if ((((size_t)t.g) & 15) == 0) {
    const long n4 = t.n >> 2;
    for (long i = tid; i < n4; i += nth) {
      const float4 g = ((const float4*)t.g)[i];
      s += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
    }
    done = n4 << 2;
  }
  for (long i = done + tid; i < t.n; i += nth) {
    float g = t.g[i];
    s += g * g;
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(norms + blockIdx.y, red[0] + red[1] + red[2] + red[3]);
}
__device__ __forceinline__ void sgd_one(float& p, float g, float& m, float coef, float lr, float momentum, float wd, int first_step) {
  g = g * coef + wd * p;
  m = first_step ? g : momentum * m + g;
  p = p - lr * m;
}
__global__ void k_sgd(SgdBatch b, const float* norms, float lr, float momentum, float wd, float clip, int first_step) {
  const SgdItem& t = b.it[blockIdx.y];
  float nrm = sqrtf(norms[blockIdx.y]);
  float coef = fminf(clip / (nrm + 1e-6f), 1.0f);   // torch.nn.utils.clip_grad_norm_
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  long done = 0;
  if (((((size_t)t.g) | ((size_t)t.p) | ((size_t)t.m)) & 15) == 0) {
    const long n4 = t.n >> 2;
    for (long i = tid; i < n4; i += nth) {
      float4 p = ((float4*)t.p)[i], m = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : ((float4*)t.m)[i];
      const float4 g = ((const float4*)t.g)[i];
      sgd_one(p.x, g.x, m.x, coef, lr, momentum, wd, first_step);
      sgd_one(p.y, g.y, m.y, coef, lr, momentum, wd, first_step);
      sgd_one(p.z, g.z, m.z, coef, lr, momentum, wd, first_step);
      sgd_one(p.w, g.w, m.w, coef, lr, momentum, wd, first_step);
      ((float4*)t.m)[i] = m;
      ((float4*)t.p)[i] = p;
    }
    done = n4 << 2;
  }
  for (long i = done + tid; i < t.n; i += nth) {
    float p = t.p[i], m = first_step ? 0.f : t.m[i];
    sgd_one(p, t.g[i], m, coef, lr, momentum, wd, first_step);
    t.m[i] = m;
    t.p[i] = p;
  }
}

inline unsigned gsz(long n, int per = 256, long cap = -1) {
  // default cap on the grid of the grid-stride streaming kernels.  tools/hbm_probe.hip: a plain 16-bytes-per-thread copy runs 6.3 TB/s as a
  // ONE-SHOT grid (what ATen's elementwise kernels do) against 4.8-5.0 with 8-32 blocks per CU looping; on these kernels (more index
  // arithmetic per element) 65536 blocks instead of 8192: avgpool2_bwd 1.02 -> 0.89, avgpool2_fwd 1.29 -> 1.25, relu_bwd 0.31 -> 0.29 ms/step
  static const long env_cap = getenv("CDDMSL_GRID_CAP") ? atol(getenv("CDDMSL_GRID_CAP")) : 65536;
  if (cap < 0) cap = env_cap;
  long g = (n + per - 1) / per;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define DISPATCH(dtype, KERNEL, ...)                                   \
  do {                                                                 \
    if ((dtype) == 0) { KERNEL<__bf16> __VA_ARGS__; }                  \
    else if ((dtype) == 1) { KERNEL<float> __VA_ARGS__; }              \
    else return CDDMSL_ERR_ARG;                                        \
  } while (0)

extern "C" int cddmsl_preprocess(const unsigned char* img, void* out, int n, int h, int w, int Hp, int Wp, int Cp,
                                 const float* mean3, const float* std3, int div255, int dtype, void* stream) {
  if (h <= 0 || w <= 0 || h > Hp || w > Wp || Cp < 3) return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  long px = (long)Hp * Wp;
  DISPATCH(dtype, k_preprocess, <<<dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st>>>(
      img, (char*)out, n, h, w, Hp, Wp, Cp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], div255 ? 255.0f : 1.0f));
  return launch_status();
}

extern "C" int cddmsl_preprocess224(const unsigned char* img, void* out, int n, int h, int w, int Hp, int Wp, int RH,
                                    int RW, int top, int left, int S, int Cp, const float* mean3, const float* std3,
                                    int dtype, void* stream) {
  if (h <= 0 || w <= 0 || h > Hp || w > Wp || Cp < 3 || S <= 0 || top < 0 || left < 0 || top + S > RH || left + S > RW)
    return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype, k_preprocess224, <<<dim3((unsigned)((S * S + 255) / 256)), dim3(256), 0, st>>>(
      img, (char*)out, n, h, w, Hp, Wp, RH, RW, top, left, S, Cp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]));
  return launch_status();
}

// the same for a whole batch: imgs / hs / ws are HOST arrays of N device pointers / heights / widths; image j goes to out[j]
extern "C" int cddmsl_preprocess_batch(const unsigned char* const* imgs, const int* hs, const int* ws, int N, void* out, int Hp, int Wp,
                                       int Cp, const float* mean3, const float* std3, int div255, int dtype, void* stream) {
  if (N < 0 || Cp < 3 || !imgs || !hs || !ws) return CDDMSL_ERR_ARG;
  for (int j = 0; j < N; ++j) if (hs[j] <= 0 || ws[j] <= 0 || hs[j] > Hp || ws[j] > Wp || !imgs[j]) return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long px = (long)Hp * Wp;
  for (int n0 = 0; n0 < N; n0 += PRE_MAX) {
    const int c = N - n0 < PRE_MAX ? N - n0 : PRE_MAX;
    PreBatch b;
    for (int j = 0; j < PRE_MAX; ++j) { const int q = j < c ? j : 0; b.img[j] = imgs[n0 + q]; b.h[j] = hs[n0 + q]; b.w[j] = ws[n0 + q]; }
    DISPATCH(dtype, k_preprocess_batch, <<<dim3((unsigned)((px + 255) / 256), (unsigned)c), dim3(256), 0, st>>>(
        b, (char*)out, n0, Hp, Wp, Cp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], div255 ? 255.0f : 1.0f));
  }
  return launch_status();
}
extern "C" int cddmsl_preprocess224_batch(const unsigned char* const* imgs, const int* hs, const int* ws, int N, void* out, int Hp, int Wp,
                                          int RH, int RW, int top, int left, int S, int Cp, const float* mean3, const float* std3,
                                          int dtype, void* stream) {
  if (N < 0 || Cp < 3 || !imgs || !hs || !ws || S <= 0 || top < 0 || left < 0 || top + S > RH || left + S > RW) return CDDMSL_ERR_ARG;
  for (int j = 0; j < N; ++j) if (hs[j] <= 0 || ws[j] <= 0 || hs[j] > Hp || ws[j] > Wp || !imgs[j]) return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int n0 = 0; n0 < N; n0 += PRE_MAX) {
    const int c = N - n0 < PRE_MAX ? N - n0 : PRE_MAX;
    PreBatch b;
    for (int j = 0; j < PRE_MAX; ++j) { const int q = j < c ? j : 0; b.img[j] = imgs[n0 + q]; b.h[j] = hs[n0 + q]; b.w[j] = ws[n0 + q]; }
    DISPATCH(dtype, k_preprocess224_batch, <<<dim3((unsigned)((S * S + 255) / 256), (unsigned)c), dim3(256), 0, st>>>(
        b, (char*)out, n0, Hp, Wp, RH, RW, top, left, S, Cp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]));
  }
  return launch_status();
}

extern "C" int cddmsl_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((C * es) % 16 || H < 2 || W < 2) return CDDMSL_ERR_ARG;
  int cch = C * es / 16;
  long total = (long)N * (H / 2) * (W / 2) * cch;
  if (total == 0) return CDDMSL_OK;
  DISPATCH(dtype, k_avgpool2_fwd, <<<dim3(gsz(total)), dim3(256), 0, (hipStream_t)stream>>>((const char*)x, (char*)y, N, H, W, cch));
  return launch_status();
}

extern "C" int cddmsl_avgpool2_bwd(const void* dy, const void* mask, const void* add, void* dx, int N, int H, int W,
                                   int C, int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((C * es) % 16 || H < 2 || W < 2) return CDDMSL_ERR_ARG;
  int cch = C * es / 16;
  long total = (long)N * H * W * cch;
  if (total == 0) return CDDMSL_OK;
  DISPATCH(dtype, k_avgpool2_bwd, <<<dim3(gsz(total)), dim3(256), 0, (hipStream_t)stream>>>(
      (const char*)dy, (const char*)mask, (const char*)add, (char*)dx, N, H, W, cch));
  return launch_status();
}

extern "C" int cddmsl_avgpool2_bwd_q8(const void* dy, const void* mask, const void* add, void* dx, int N, int H, int W, int C, void* y8,
                                      const float* q8, float* amax8, void* stream) {
  if ((C * 2) % 16 || !y8 || !q8 || !amax8) return CDDMSL_ERR_ARG;
  const int cch = C * 2 / 16;
  const long total = (long)N * H * W * cch;
  if (total == 0) return CDDMSL_OK;
  // (8192 blocks, not the streaming kernels' 65536: every block ends with a reduction and an atomic -- measured 1.17 vs 1.21 ms; the
  // quantiser itself, same structure, 0.82 vs 1.07 ms at 2048 vs 65536 blocks)
  hipLaunchKernelGGL(k_avgpool2_bwd_q8, dim3(gsz(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const char*)dy, (const char*)mask, (const char*)add,
                     (char*)dx, N, H, W, cch, (char*)y8, q8, (unsigned*)amax8);
  return launch_status();
}

extern "C" int cddmsl_maxpool3s2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((C * es) % 16 || H < 1 || W < 1) return CDDMSL_ERR_ARG;
  int cch = C * es / 16, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  long total = (long)N * Ho * Wo * cch;
  if (total == 0) return CDDMSL_OK;
  DISPATCH(dtype, k_maxpool3s2, <<<dim3(gsz(total)), dim3(256), 0, (hipStream_t)stream>>>((const char*)x, (char*)y, N, H, W, Ho, Wo, cch));
  return launch_status();
}
extern "C" int cddmsl_upsample_zero2(const void* t, const void* mask, const void* add, void* dx, int N, int H, int W, int C,
                                     int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((C * es) % 16 || H < 1 || W < 1) return CDDMSL_ERR_ARG;
  int cch = C * es / 16, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  long total = (long)N * H * W * cch;
  if (total == 0) return CDDMSL_OK;
  DISPATCH(dtype, k_upsample_zero2, <<<dim3(gsz(total)), dim3(256), 0, (hipStream_t)stream>>>(
      (const char*)t, (const char*)mask, (const char*)add, (char*)dx, N, H, W, Ho, Wo, cch));
  return launch_status();
}
extern "C" int cddmsl_meanpool_fwd(const void* x, float* y, long K, int P, int C, int dtype, void* stream) {
  if (K < 0 || P <= 0 || C <= 0) return CDDMSL_ERR_ARG;
  if (K == 0) return CDDMSL_OK;
  long n = K * C;
  DISPATCH(dtype, k_meanpool_fwd, <<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>((const char*)x, y, K, P, C));
  return launch_status();
}
extern "C" int cddmsl_meanpool_bwd(const float* dy, void* dx, long K, int P, int C, int dtype, void* stream) {
  if (K < 0 || P <= 0 || C <= 0) return CDDMSL_ERR_ARG;
  if (K == 0) return CDDMSL_OK;
  long n = K * P * C;
  DISPATCH(dtype, k_meanpool_bwd, <<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(dy, (char*)dx, K, P, C));
  return launch_status();
}

extern "C" int cddmsl_attn_tokens_fwd(const void* x, const float* pos, void* tok, int K, int P, int TP, int C, int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((C * es) % 16 || P <= 0 || TP < P + 1) return CDDMSL_ERR_ARG;
  int cch = C * es / 16;
  long total = (long)K * cch;
  if (total == 0) return CDDMSL_OK;
  DISPATCH(dtype, k_attn_tokens_fwd, <<<dim3(gsz(total, 64)), dim3(64), 0, (hipStream_t)stream>>>((const char*)x, pos, (char*)tok, nullptr, K, P, TP, cch));
  return launch_status();
}

// the same, also writing the pooled map's sign bits (one 64-bit word per region and column, bit p = x[k][p][col] > 0; P <= 64)
extern "C" int cddmsl_attn_tokens_fwd_mask(const void* x, const float* pos, void* tok, unsigned long long* mbits, int K, int P, int TP, int C,
                                           int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((C * es) % 16 || P <= 0 || P > 64 || TP < P + 1 || K < 0 || !mbits) return CDDMSL_ERR_ARG;
  int cch = C * es / 16;
  long total = (long)K * cch;
  if (total == 0) return CDDMSL_OK;
  DISPATCH(dtype, k_attn_tokens_fwd, <<<dim3(gsz(total, 64)), dim3(64), 0, (hipStream_t)stream>>>((const char*)x, pos, (char*)tok, mbits, K, P, TP, cch));
  return launch_status();
}

extern "C" int cddmsl_attn_tokens_bwd(const void* dtok, const void* relu_mask, void* dx, float* gpos, int K, int P, int TP, int C,
                                      int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if ((C * es) % 16 || P <= 0 || TP < P + 1 || K < 0 || (!dx && !gpos)) return CDDMSL_ERR_ARG;
  int cch = C * es / 16;
  if (K == 0) return CDDMSL_OK;
  int slab = 128;                                          // (64-128 regions per block measured best: 16 -> +0.6 ms, 512 -> +0.5 ms)
  while ((K + slab - 1) / slab > 65535) slab *= 2;
  dim3 grid((unsigned)((cch + 255) / 256), (unsigned)(P + 1), (unsigned)((K + slab - 1) / slab));
  DISPATCH(dtype, k_attn_tokens_bwd, <<<grid, dim3(256), 0, (hipStream_t)stream>>>((const char*)dtok, (const char*)relu_mask, (char*)dx, gpos, K, P, TP, cch, slab));
  return launch_status();
}

extern "C" int cddmsl_attnpool_softmax_fwd(const float* S, float* p, void* pT, long K, int H, int P1, int TP, float scale, int dtype,
                                           void* stream) {
  if (K < 0 || H <= 0 || P1 <= 0 || TP < P1) return CDDMSL_ERR_ARG;
  const long rows = K * H;
  if (rows == 0) return CDDMSL_OK;
  if (H <= 32 && TP <= 64) {
    DISPATCH(dtype, k_attnpool_softmax_fwd_w, <<<dim3((unsigned)((K + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(S, p, (char*)pT, K, H, P1, TP, scale));
    return launch_status();
  }
  DISPATCH(dtype, k_attnpool_softmax_fwd, <<<dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(S, p, (char*)pT, rows, H, P1, TP, scale));
  return launch_status();
}
extern "C" int cddmsl_attnpool_softmax_bwd(const float* p, const float* dP, void* dsT, void* pds, long K, int H, int P1, int TP,
                                           float scale, int dtype, void* stream) {
  if (K < 0 || H <= 0 || P1 <= 0 || TP < P1) return CDDMSL_ERR_ARG;
  const long rows = K * H;
  if (rows == 0) return CDDMSL_OK;
  if (H <= 32 && TP <= 64) {
    DISPATCH(dtype, k_attnpool_softmax_bwd_w, <<<dim3((unsigned)((K + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(p, dP, (char*)dsT, (char*)pds, K, H, P1, TP, scale));
    return launch_status();
  }
  DISPATCH(dtype, k_attnpool_softmax_bwd, <<<dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(p, dP, (char*)dsT, (char*)pds, rows, H, P1, TP, scale));
  return launch_status();
}

extern "C" int cddmsl_relu_bwd(const void* g, const void* y, void* dx, long numel, int g_f32, int dtype, void* stream) {
  int vec = dtype == 0 ? 8 : 4;
  if (numel < 0 || numel % vec) return CDDMSL_ERR_ARG;
  if (numel == 0) return CDDMSL_OK;
  long nch = numel / vec;
  DISPATCH(dtype, k_relu_bwd, <<<dim3(gsz(nch)), dim3(256), 0, (hipStream_t)stream>>>((const char*)g, (const char*)y, (char*)dx, nch, g_f32));
  return launch_status();
}

// out (f32, caller-zeroed or accumulated into) [period][cols] += column sums of x [rows][cols]
extern "C" int cddmsl_colsum(const void* x, float* out, long rows, int cols, int period, int dtype, void* stream) {
  if (rows < 0 || cols <= 0 || period <= 0) return CDDMSL_ERR_ARG;
  if (rows == 0) return CDDMSL_OK;
  if (period > 65535) return CDDMSL_ERR_ARG;
  long nj = (rows + period - 1) / period;
  int slab = 256;
  while ((nj + slab - 1) / slab > 2048) slab *= 2;      // keep the grid modest; rows per thread = slab / 4
  const int vec = dtype == 0 ? 8 : 4;
  if (cols % vec == 0) {                                  // whole 16-byte chunks per row: vector loads
    dim3 gridv((unsigned)((cols / vec + 63) / 64), (unsigned)((nj + slab - 1) / slab), (unsigned)period);
    DISPATCH(dtype, k_colsum_vec, <<<gridv, dim3(256), 0, (hipStream_t)stream>>>((const char*)x, out, rows, cols, period, slab));
    return launch_status();
  }
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((nj + slab - 1) / slab), (unsigned)period);
  DISPATCH(dtype, k_colsum, <<<grid, dim3(256), 0, (hipStream_t)stream>>>((const char*)x, out, rows, cols, period, slab));
  return launch_status();
}

// Multi-tensor step: ptr arrays live on the host; tensors are batched SGD_MAX per launch pair.
extern "C" int cddmsl_sgd_clip_step(float** params, const float** grads, float** moms, const long* sizes, int count,
                                    float* norm_ws, float lr, float momentum, float wd, float clip, int first_step,
                                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (count < 0) return CDDMSL_ERR_ARG;
  for (int base = 0; base < count; base += SGD_MAX) {
    SgdBatch b;
    b.count = count - base < SGD_MAX ? count - base : SGD_MAX;
    long mx = 0;
    for (int i = 0; i < b.count; ++i) {
      b.it[i].p = params[base + i]; b.it[i].g = grads[base + i]; b.it[i].m = moms[base + i]; b.it[i].n = sizes[base + i];
      if (sizes[base + i] > mx) mx = sizes[base + i];
    }
    if (hipMemsetAsync(norm_ws + base, 0, sizeof(float) * b.count, st) != hipSuccess) return CDDMSL_ERR_LAUNCH;
    unsigned gx = gsz(mx, 256 * 8, 256);
    k_sqnorm<<<dim3(gx, b.count), dim3(256), 0, st>>>(b, norm_ws + base);
    k_sgd<<<dim3(gx, b.count), dim3(256), 0, st>>>(b, norm_ws + base, lr, momentum, wd, clip, first_step);
  }
  return launch_status();
}

// ABI version of include/cddmsl_hip.h (bumped when an entry point's signature changes); 2 = round 2 (bring-up probe removed)
extern "C" int cddmsl_abi_version() { return 3; }
