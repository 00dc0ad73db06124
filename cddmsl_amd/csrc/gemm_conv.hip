// Implicit-GEMM convolution / linear kernels for gfx950 (MI355X, CDNA4), wave64 + MFMA.
//
// Replaces on the hot path: ATen conv2d / F.linear as called from the reference's
//   ModifiedResNet / Bottleneck   detectron2/modeling/backbone/clip_backbone.py:57-70,193-219
//   FrozenBatchNorm2d             detectron2/layers/batch_norm.py:45-66   (fused epilogue)
//   StandardRPNHead               detectron2/modeling/proposal_generator/rpn.py:158-177
//   AttentionPool2d projections   clip_backbone.py:83-107
//   TransformerMapper linears     detectron2/modeling/backbone/clipcap/clipcap.py:39-163
//
// Data layout: activations NHWC (channels contiguous), weights [Cout][KH][KW][Cin] (= torch
// channels_last OIHW).  Everything is addressed in 16-byte "chunks" along the contraction axis, so
// one kernel template serves bf16 (8 elem/chunk, v_mfma_f32_32x32x16_bf16) and exact f32
// (4 elem/chunk, v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) -- the f32 instantiation is the
// parity path, the bf16 one the throughput path.
//
//   conv_fwd  : Y[m,n] = epi( sum_k A[m,k] * W[n,k] ),  A gathered on the fly from NHWC input
//               (m = (img,oy,ox), k = (ky,kx,c)); epilogue = per-n scale/bias (FrozenBN), residual
//               add, ReLU, ReLU-backward mask, f32 or T store.  dgrad runs through the same kernel
//               with flipped/transposed weights (see weight_prep).
//   conv_wgrad: dW[n,k] += scale[n] * sum_m dY[m,n] * A[m,k]  (reduction over m = pixels), both
//               operands staged row-major and read transposed (ds_read_b64_tr_b16), split over m
//               with f32 atomics.
//
// Tile: 128x128 per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 2x2 MFMA 32x32 tiles),
// K-tile = 8 chunks (128 B per row).  LDS rows are 128 B with chunk ^= (row>>1)&7 so that every
// ds_read_b128 lane group hits 16 distinct 16-B slots (bank = (addr/4)%64).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, KCH = 8;  // KCH chunks of 16 B per K-tile row

// n / d for 0 <= n < 2^31 with a host-precomputed multiplier (round-up method): q = (umulhi(n, mul) + n) >> shr
struct FastDiv {
  unsigned mul, shr, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.shr = s;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.shr; }

// FrozenBN / bias in the epilogues: v * scale + bias.  The exact-f32 instantiations round twice like the reference's separate
// multiply and add (this file is built with -ffp-contract=off); the bf16 ones use one fused multiply-add -- one vector-ALU
// operation less per output element in every epilogue, identical where scale = 1 and bias = 0 (input-gradient launches).
template <typename T> __device__ __forceinline__ float affine(float v, float sc, float bi) {
  return sizeof(T) <= 2 ? __builtin_fmaf(v, sc, bi) : v * sc + bi;
}

__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);       // e4m3fn has no infinity: saturate
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// Cache-policy bits (buffer instruction aux: 1 = sc0, 2 = nt, 16 = sc1) of the 256x256 kernel's output stores and of its residual /
// ReLU-mask loads.  The outputs are written once and read by a LATER launch, by which time they have left the caches anyway: stored
// non-temporal they stop evicting the operand tiles the other workgroups are re-reading -- measured on the training step, same box,
// rebuilt library (scratch A/B): nt stores -0.55 .. -0.95 ms per step, sc0|nt the same, sc0 alone nothing,
// nt|sc1 +0.4 ms, nt on the residual / mask loads nothing on top.
#ifndef CDDMSL_STORE_AUX
#define CDDMSL_STORE_AUX 2
#endif
#ifndef CDDMSL_LOAD_AUX
#define CDDMSL_LOAD_AUX 0
#endif

struct ConvArgs {
  const char* x;
  const char* w;
  char* y;
  const float* scale;
  const float* bias;
  const char* residual;
  const char* relu_mask;
  int Nimg, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int ldy, ldr, ldm;
  int relu, out_f32, pool;
  int res_f32;     // residual rows are f32 although T is bf16 (f32 output only): the mapper's f32 residual stream
  int res_pool;    // residual is [Nimg][Ho/2][Wo/2][ldr]: row m adds 0.25 * residual[pooled pixel of m] (AvgPool2d(2) backward fused)
  int M, Kc, cpp;  // rows, total K chunks, chunks per pixel
  FastDiv dWo, dHo, dcpp, dKW;
  int xrs, wrs;    // row strides in 16-byte chunks: A pixel -> pixel (default cpp), B row -> row (default Kc)
  long bx, bw, by; // byte strides of the batch axis (gridDim.y); 0 for plain convolutions
  // fp8 configuration: a second, OCP e4m3 copy of the (bf16) output for the convolution that consumes it -- y8[m][n] =
  // sat(y * q8[0]) -- written by the 256x256 kernel's epilogue, which also max-es |y| into amax8[blockIdx & 63] (delayed scaling)
  char* y8 = nullptr;
  const float* q8 = nullptr;
  unsigned* amax8 = nullptr;
  // split-K form of the 256x256 kernel (the tail of a launch whose tile count leaves the last round of workgroups nearly empty):
  // block b computes K-tiles [kper * (b % ksplits), ...) of logical tile tile0 + b / ksplits and stores its raw accumulators, in
  // fragment order, at partial[b]; k_conv_split_reduce sums a tile's splits and applies the epilogue.  tile_limit: the main
  // launch stops at this logical tile (persistent form; the one-tile grid is simply shorter).
  float* partial = nullptr;
  int tile0 = 0, ksplits = 1, kper = 0, tile_limit = 0;
  int nt_out = 1;  // bf16 output stored non-temporal (CDDMSL_STORE_AUX): outputs too large to be found in the caches by their consumer
#ifdef CDDMSL_STAMPS
  unsigned long long* stamps;   // diagnostic build only (scratch/k256.hip): per-wave cycle sums of the phase segments
#endif
#ifdef CDDMSL_TILE_STAMPS
  unsigned long long* tstamps = nullptr;  // diagnostic build only (tools/tile_stamps.py): per wave, 100 MHz s_memrealtime stamps at entry / loop start / loop end / exit
#endif
};

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
  static constexpr int ES = 2;
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ float load(const char* p) { return bf2f(*(const unsigned short*)p); }
  __device__ static __forceinline__ void store(char* p, float v) { *(unsigned short*)p = f2bf(v); }
};
template <> struct Mma<float> {
  static constexpr int ES = 4;
  // lane half h holds k = 4h + j in element j; MFMA step j contracts k in {j, 4 + j}: the same
  // permutation on both operands, so the sum over the chunk pair is exact f32 fma accumulation.
  __device__ static __forceinline__ void step(f32x16& acc, const u32x4& a, const u32x4& b) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc, 0, 0, 0);
  }
  __device__ static __forceinline__ float load(const char* p) { return *(const float*)p; }
  __device__ static __forceinline__ void store(char* p, float v) { *(float*)p = v; }
};

// OCP e4m3 operands (BASELINE.json configs[4]): a 16-byte chunk holds 16 elements, one block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 (E8M0 scales fixed at 2^0: per-tensor scales are folded into the epilogue's per-channel scale by
// the host) consumes TWO chunks per lane = 64 elements of K: twice the K per matrix-pipe cycle of the bf16 form.  The epilogue
// (output, residual, ReLU mask) stays bf16: ES below is the epilogue's element size.  Lane half h of a step holds chunks
// (2s' + h) for s' in the step's pair -- the same bytes for A and B, which is all a dot product needs.
struct fp8e4 { unsigned char v; };
typedef __attribute__((ext_vector_type(8))) int i32x8;
template <> struct Mma<fp8e4> {
  static constexpr int ES = 2;
  __device__ static __forceinline__ void step2(f32x16& acc, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
    const i32x8 a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const i32x8 b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
};
// one 64x32 quadrant of a K-tile (4 chunk pairs per row): two row tiles x the tile's k-steps
template <typename T> struct MmaQuad {
  __device__ static __forceinline__ void run(f32x16& c0, f32x16& c1, const u32x4 (*fa)[4], const u32x4* fb) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Mma<T>::step(c0, fa[0][ks], fb[ks]);
      Mma<T>::step(c1, fa[1][ks], fb[ks]);
    }
  }
};
template <> struct MmaQuad<fp8e4> {
  __device__ static __forceinline__ void run(f32x16& c0, f32x16& c1, const u32x4 (*fa)[4], const u32x4* fb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Mma<fp8e4>::step2(c0, fa[0][2 * s], fa[0][2 * s + 1], fb[2 * s], fb[2 * s + 1]);
      Mma<fp8e4>::step2(c1, fa[1][2 * s], fa[1][2 * s + 1], fb[2 * s], fb[2 * s + 1]);
    }
  }
};

// 8 consecutive elements (16 B of bf16 / 32 B of f32) -> floats
template <typename T> __device__ __forceinline__ void load8(const char* p, float* f);
template <> __device__ __forceinline__ void load8<__bf16>(const char* p, float* f) {
  const u32x4 v = *(const u32x4*)p;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f(v[j] & 0xffff); f[2 * j + 1] = bf2f(v[j] >> 16); }
}
template <> __device__ __forceinline__ void load8<float>(const char* p, float* f) {
  const f32x4 a = ((const f32x4*)p)[0], b = ((const f32x4*)p)[1];
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

// XCD-aware tile order (8 XCDs, blocks dealt round-robin): give each XCD a contiguous run of logical tiles so the
// tiles that share an A row-panel / weight panel hit the same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, rem = nblk & 7, x = bid & 7, j = bid >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * KCH + (chunk ^ ((row >> 1) & 7)); }

// average of 4 packed chunks (2x2 avg-pool fused into the A loader)
template <typename T> __device__ __forceinline__ u32x4 avg4(const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& d);
template <> __device__ __forceinline__ u32x4 avg4<float>(const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& d) {
  const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
  const f32x4 fc = __builtin_bit_cast(f32x4, c), fd = __builtin_bit_cast(f32x4, d);
  f32x4 r = ((fa + fb) + (fc + fd)) * 0.25f;
  return __builtin_bit_cast(u32x4, r);
}
template <> __device__ __forceinline__ u32x4 avg4<__bf16>(const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& d) {
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float lo = (bf2f(a[j] & 0xffff) + bf2f(b[j] & 0xffff)) + (bf2f(c[j] & 0xffff) + bf2f(d[j] & 0xffff));
    float hi = (bf2f(a[j] >> 16) + bf2f(b[j] >> 16)) + (bf2f(c[j] >> 16) + bf2f(d[j] >> 16));
    r[j] = pack2bf(lo * 0.25f, hi * 0.25f);
  }
  return r;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void k_conv_fwd_reg(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][BM * KCH];
  const int t = threadIdx.x;
  const int ntn = (p.Cout + BN - 1) / BN;
  const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int cc = t & 7, rb = t >> 3;

  // per-thread A-row geometry (4 rows, fixed across K-tiles)
  long pix[4];
  int iy0[4], ix0[4];
  bool vm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + rb + 32 * i;
    vm[i] = m < p.M;
    int mm = vm[i] ? m : 0;
    int ox = mm % p.Wo, tq = mm / p.Wo;
    int oy = tq % p.Ho, img = tq / p.Ho;
    int s = p.pool ? 2 * p.stride : p.stride;
    iy0[i] = oy * s - p.pad;
    ix0[i] = ox * s - p.pad;
    pix[i] = ((long)img * p.Hi + iy0[i]) * p.Wi + ix0[i];
  }
  const int nkt = (p.Kc + KCH - 1) / KCH;
  u32x4 ra[4], rbv[4];
  const u32x4 zero = {0u, 0u, 0u, 0u};

  auto gload = [&](int kt) {
    int kc = kt * KCH + cc;
    bool vk = kc < p.Kc;
    int pp = vk ? kc / p.cpp : 0;
    int coff = vk ? kc - pp * p.cpp : 0;
    int ky = pp / p.KW, kx = pp - ky * p.KW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = iy0[i] + ky, ix = ix0[i] + kx;
      if (!p.pool) {
        bool ok = vk && vm[i] && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const u32x4* src = (const u32x4*)(p.x + ((pix[i] + (long)ky * p.Wi + kx) * p.cpp + coff) * 16);
        ra[i] = ok ? *src : zero;
      } else {  // 1x1 conv over a 2x2 average-pooled input (floor semantics: Ho = Hi/2)
        bool ok = vk && vm[i];
        if (ok) {
          const char* b0 = p.x + (pix[i] * p.cpp + coff) * 16;
          long rs = (long)p.Wi * p.cpp * 16, cs = (long)p.cpp * 16;
          ra[i] = avg4<T>(*(const u32x4*)b0, *(const u32x4*)(b0 + cs), *(const u32x4*)(b0 + rs), *(const u32x4*)(b0 + rs + cs));
        } else ra[i] = zero;
      }
      int n = n0 + rb + 32 * i;
      bool okb = vk && n < p.Cout;
      rbv[i] = okb ? *(const u32x4*)(p.w + ((long)n * p.Kc + kc) * 16) : zero;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lane = t & 63, wv = t >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;

  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = rb + 32 * i;
      lds[0][swz(row, cc)] = ra[i];
      lds[1][swz(row, cc)] = rbv[i];
    }
    __syncthreads();
    if (kt + 1 < nkt) gload(kt + 1);  // next tile's HBM loads fly under this tile's MFMAs
#pragma unroll
    for (int ks = 0; ks < KCH / 2; ++ks) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = lds[0][swz(wm * 64 + i * 32 + r, 2 * ks + h)];
        fb[i] = lds[1][swz(wn * 64 + i * 32 + r, 2 * ks + h)];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) Mma<T>::step(acc[a][b], fa[a], fb[b]);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------------------------------------
  // epilogue.  C/D map of the 32x32 MFMA: col = lane&31 (n), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (m).
  // Vector path (all leading dims multiples of 8): each wave transposes its accumulators through LDS
  // (32 rows x 64 cols f32 per pass, 32-byte column groups XOR-swizzled by row) so that every lane owns
  // 8 consecutive channels of one pixel: residual / mask are read and y is written 16-32 B per lane,
  // whole 128-B lines per 8 lanes -- the scalar path issued 64 two-byte stores per lane instead.
  const bool vec_ok = (p.Cout % 8 == 0) && (p.ldy % 8 == 0) && (!p.residual || p.ldr % 8 == 0) && (!p.relu_mask || p.ldm % 8 == 0);
  if (vec_ok) {
    float* ep = (float*)&lds[0][0] + wv * 2048;      // 8 KB per wave; the K-loop's last barrier already passed
    const int cg = lane & 7, rr = lane >> 3;
    const int n = n0 + wn * 64 + cg * 8;
    float sc[8], bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = (p.scale && n + j < p.Cout) ? p.scale[n + j] : 1.f;
      bi[j] = (p.bias && n + j < p.Cout) ? p.bias[n + j] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          int row = (g & 3) + 8 * (g >> 2) + 4 * h, col = b * 32 + r;
          ep[row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7))] = acc[a][b][g];
        }
      __syncthreads();
      if (n < p.Cout) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int row = rr + 8 * i;
          int m = m0 + wm * 64 + a * 32 + row;
          if (m >= p.M) continue;
          const f32x4* src = (const f32x4*)(ep + row * 64 + ((cg ^ (row & 7)) << 3));
          f32x4 v0 = src[0], v1 = src[1];
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = affine<T>(v[j], sc[j], bi[j]);
          if (p.residual) {
            float rv[8];
            load8<T>(p.residual + ((long)m * p.ldr + n) * Mma<T>::ES, rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
          }
          if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if (p.relu_mask) {
            float mv[8];
            load8<T>(p.relu_mask + ((long)m * p.ldm + n) * Mma<T>::ES, mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(mv[j] > 0.f)) v[j] = 0.f;
          }
          if (p.out_f32 || Mma<T>::ES == 4) {
            f32x4* dst = (f32x4*)(p.y + ((long)m * p.ldy + n) * 4);
            f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            dst[0] = o0; dst[1] = o1;
          } else {
            u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            *(u32x4*)(p.y + ((long)m * p.ldy + n) * 2) = o;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    int n = n0 + wn * 64 + b * 32 + r;
    if (n >= p.Cout) continue;
    float sc = p.scale ? p.scale[n] : 1.f;
    float bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int m = m0 + wm * 64 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = affine<T>(acc[a][b][g], sc, bi);
        if (p.residual) v += Mma<T>::load(p.residual + ((long)m * p.ldr + n) * Mma<T>::ES);
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.relu_mask && !(Mma<T>::load(p.relu_mask + ((long)m * p.ldm + n) * Mma<T>::ES) > 0.f)) v = 0.f;
        if (p.out_f32) *(float*)(p.y + ((long)m * p.ldy + n) * 4) = v;
        else Mma<T>::store(p.y + ((long)m * p.ldy + n) * Mma<T>::ES, v);
      }
    }
  }
}

// LDS-DMA variant (pool == 0): A and B tiles go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
// instruction = 8 rows x 128 B), double-buffered, with a counted vmcnt so the next tile's 8 DMAs per thread stay in
// flight across the barrier.  The VGPR -> LDS write path (ds_write_b128 ~13 cycles per wave-instruction) was the
// bottleneck of the register-staged loop at two blocks per CU.  LDS destination is lane-linear, so the XOR swizzle
// is applied to the per-lane SOURCE chunk; out-of-image / tail lanes read a 16-byte zero page.
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// row of the 2x2-average-pooled tensor that output pixel m falls into (res_pool), or -1 on an odd size's last row / column
__device__ __forceinline__ long pooled_row(const ConvArgs& p, int m) {
  const unsigned tq = fdiv((unsigned)m, p.dWo), ox = m - tq * p.Wo;
  const unsigned img = fdiv(tq, p.dHo), oy = tq - img * p.Ho;
  const unsigned hp = p.Ho >> 1, wp = p.Wo >> 1;
  return ((oy >> 1) < hp && (ox >> 1) < wp) ? ((long)img * hp + (oy >> 1)) * wp + (ox >> 1) : -1L;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void k_conv_fwd(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][2][BM * KCH];   // [buffer][A|B]
  const int t = threadIdx.x;
  p.x += (long)blockIdx.y * p.bx; p.w += (long)blockIdx.y * p.bw; p.y += (long)blockIdx.y * p.by;   // batched GEMM
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);     // wave index in an SGPR: LDS-DMA bases become scalar
  const int ntn = (p.Cout + BN - 1) / BN;
  const int lbid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = lbid % ntn, tile_m = lbid / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int cc = t & 7, rb = t >> 3;
  const int cl = cc ^ ((rb >> 1) & 7);        // logical K chunk this lane fetches (rows rb+32i share (row>>1)&7)

  // per-thread row geometry, fixed across K-tiles: byte offset of the (ky=kx=0) tap of each of the 4 rows
  long rowoff[4], wrow[4];
  int iy0[4], ix0[4];
  bool vm[4], vn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + rb + 32 * i;
    vm[i] = m < p.M;
    unsigned mm = vm[i] ? m : 0;
    unsigned tq = fdiv(mm, p.dWo), ox = mm - tq * p.Wo;
    unsigned img = fdiv(tq, p.dHo), oy = tq - img * p.Ho;
    iy0[i] = (int)oy * p.stride - p.pad;
    ix0[i] = (int)ox * p.stride - p.pad;
    rowoff[i] = ((((long)img * p.Hi + iy0[i]) * p.Wi + ix0[i]) * p.xrs) * 16;
    int n = n0 + rb + 32 * i;
    vn[i] = n < p.Cout;
    wrow[i] = (long)(vn[i] ? n : 0) * p.wrs * 16;
  }
  const int nkt = (p.Kc + KCH - 1) / KCH;
  const char* zp = (const char*)g_zero_page;
  const bool taps = !(p.KH == 1 && p.KW == 1 && p.pad == 0);   // 1x1 / linear: every tap is inside the image
  // running K position of this lane's logical chunk: kc -> (ky, kx, coff); advanced by KCH per tile without divisions
  int kc = cl;
  int pp0 = (int)fdiv((unsigned)kc, p.dcpp);
  int coff = kc - pp0 * p.cpp;
  int ky = (int)fdiv((unsigned)pp0, p.dKW), kx = pp0 - ky * p.KW;

  // Fast path (cpp % 8 == 0, i.e. a K-tile never straddles two filter taps -- every layer past the stem): per-row source
  // pointers are kept in registers and advanced by 128 B per tile (2 VALU each); tap validity / base addresses are
  // recomputed only when the (wave-uniform) tap changes.  The general path recomputes everything per tile.
  const bool fast = (p.cpp & 7) == 0;
  const char* pa[4];
  const char* pb[4];
  int inca[4], incb[4];
  int tiles_left_in_tap = 0;                    // K-tiles before (ky,kx) advances (fast path)
  auto retap = [&]() {                          // (re)build the A pointers for the current (ky, kx, coff)
    const int delta = ((ky * p.Wi + kx) * p.xrs + coff) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = vm[i];
      if (taps) {
        int iy = iy0[i] + ky, ix = ix0[i] + kx;
        ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      }
      pa[i] = ok ? p.x + rowoff[i] + delta : zp;
      inca[i] = ok ? KCH * 16 : 0;
    }
  };
  if (fast) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pb[i] = vn[i] ? p.w + wrow[i] + (long)kc * 16 : zp;
      incb[i] = vn[i] ? KCH * 16 : 0;
    }
    retap();
    tiles_left_in_tap = p.cpp >> 3;
  }

  auto stage = [&](int buf) {
    if (fast) {
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(pa[i], &lds[buf][0][(8 * wvu + 32 * i) * KCH]);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(pb[i], &lds[buf][1][(8 * wvu + 32 * i) * KCH]);
#pragma unroll
      for (int i = 0; i < 4; ++i) pb[i] += incb[i];
      if (--tiles_left_in_tap == 0) {          // wave-uniform: next tile starts a new filter tap
        tiles_left_in_tap = p.cpp >> 3;
        coff = cl;
        if (++kx == p.KW) { kx = 0; ++ky; }
        retap();
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[i] += inca[i];
      }
      return;
    }
    const bool vk = kc < p.Kc;
    const int delta = ((ky * p.Wi + kx) * p.xrs + coff) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = vk && vm[i];
      if (taps) {
        int iy = iy0[i] + ky, ix = ix0[i] + kx;
        ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      }
      const char* src = ok ? p.x + rowoff[i] + delta : zp;
      glds16(src, &lds[buf][0][(8 * wvu + 32 * i) * KCH]);      // wave-uniform base; lane l lands at +16*l
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* src = (vk && vn[i]) ? p.w + wrow[i] + (long)kc * 16 : zp;
      glds16(src, &lds[buf][1][(8 * wvu + 32 * i) * KCH]);
    }
    kc += KCH;
    coff += KCH;
    while (coff >= p.cpp) {
      coff -= p.cpp;
      if (++kx == p.KW) { kx = 0; ++ky; }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lane = t & 63, wv = t >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;

  // Epilogue operands (residual, ReLU mask) are fetched NOW, under the whole K loop, instead of inside the epilogue where
  // their HBM latency (~2 us per pass) was fully exposed on the small-K / wide-N layers (conv3 + residual, dgrad + mask).
  const bool vec_ok = (p.Cout % 8 == 0) && (p.ldy % 8 == 0) && (!p.residual || p.ldr % 8 == 0) && (!p.relu_mask || p.ldm % 8 == 0);
  constexpr bool PRE = Mma<T>::ES == 2;
  u32x4 rres[2][4], rmsk[2][4];
  const bool rf32 = PRE && p.res_f32;             // f32 residual rows on the bf16 kernel: read in the epilogue
  if (PRE && vec_ok && ((p.residual && !rf32) || p.relu_mask)) {
    const int n = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int m = m0 + wm * 64 + a * 32 + (lane >> 3) + 8 * i;
        bool ok = m < p.M && n < p.Cout;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (p.res_pool) {
          const long po = pooled_row(p, m);
          rres[a][i] = (ok && po >= 0) ? *(const u32x4*)(p.residual + (po * p.ldr + n) * 2) : z;
        } else
        rres[a][i] = (ok && p.residual && !rf32) ? *(const u32x4*)(p.residual + ((long)m * p.ldr + n) * 2) : z;
        rmsk[a][i] = (ok && p.relu_mask) ? *(const u32x4*)(p.relu_mask + ((long)m * p.ldm + n) * 2) : z;
      }
  }

  stage(0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) {
      stage(cur ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this tile's 8 DMAs done; the next tile's 8 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // fragments of k-step ks+1 are read while the MFMAs of k-step ks run (two named register sets, static indices)
    u32x4 fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[0][i] = lds[cur][0][swz(wm * 64 + i * 32 + r, h)];
      fb[0][i] = lds[cur][1][swz(wn * 64 + i * 32 + r, h)];
    }
#pragma unroll
    for (int ks = 0; ks < KCH / 2; ++ks) {
      if (ks + 1 < KCH / 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[(ks + 1) & 1][i] = lds[cur][0][swz(wm * 64 + i * 32 + r, 2 * (ks + 1) + h)];
          fb[(ks + 1) & 1][i] = lds[cur][1][swz(wn * 64 + i * 32 + r, 2 * (ks + 1) + h)];
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) Mma<T>::step(acc[a][b], fa[ks & 1][a], fb[ks & 1][b]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // everyone is done reading buffer `cur`
  }

  // ---------------------------------------------------------------------------------------------
  // epilogue.  C/D map of the 32x32 MFMA: col = lane&31 (n), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (m).
  // Vector path (all leading dims multiples of 8): each wave transposes its accumulators through LDS
  // (32 rows x 64 cols f32 per pass, 32-byte column groups XOR-swizzled by row) so that every lane owns
  // 8 consecutive channels of one pixel: residual / mask are read and y is written 16-32 B per lane,
  // whole 128-B lines per 8 lanes -- the scalar path issued 64 two-byte stores per lane instead.
  if (vec_ok) {
    float* ep = (float*)&lds[0][0][0] + wv * 2048;      // 8 KB per wave; the K-loop's last barrier already passed
    const int cg = lane & 7, rr = lane >> 3;
    const int n = n0 + wn * 64 + cg * 8;
    float sc[8], bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = (p.scale && n + j < p.Cout) ? p.scale[n + j] : 1.f;
      bi[j] = (p.bias && n + j < p.Cout) ? p.bias[n + j] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          int row = (g & 3) + 8 * (g >> 2) + 4 * h, col = b * 32 + r;
          ep[row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7))] = acc[a][b][g];
        }
      __syncthreads();
      if (n < p.Cout) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int row = rr + 8 * i;
          int m = m0 + wm * 64 + a * 32 + row;
          if (m >= p.M) continue;
          const f32x4* src = (const f32x4*)(ep + row * 64 + ((cg ^ (row & 7)) << 3));
          f32x4 v0 = src[0], v1 = src[1];
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = affine<T>(v[j], sc[j], bi[j]);
          if (p.residual) {
            float rv[8];
            if (rf32) load8<float>(p.residual + ((long)m * p.ldr + n) * 4, rv);
            else if (PRE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { rv[2 * j] = bf2f(rres[a][i][j] & 0xffff); rv[2 * j + 1] = bf2f(rres[a][i][j] >> 16); }
            } else if (p.res_pool) {
              const long po = pooled_row(p, m);
#pragma unroll
              for (int j = 0; j < 8; ++j) rv[j] = 0.f;
              if (po >= 0) load8<T>(p.residual + (po * p.ldr + n) * Mma<T>::ES, rv);
            } else load8<T>(p.residual + ((long)m * p.ldr + n) * Mma<T>::ES, rv);
            if (p.res_pool) {
#pragma unroll
              for (int j = 0; j < 8; ++j) rv[j] *= 0.25f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
          }
          if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if (p.relu_mask) {
            float mv[8];
            if (PRE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { mv[2 * j] = bf2f(rmsk[a][i][j] & 0xffff); mv[2 * j + 1] = bf2f(rmsk[a][i][j] >> 16); }
            } else load8<T>(p.relu_mask + ((long)m * p.ldm + n) * Mma<T>::ES, mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(mv[j] > 0.f)) v[j] = 0.f;
          }
          if (p.out_f32 || Mma<T>::ES == 4) {
            f32x4* dst = (f32x4*)(p.y + ((long)m * p.ldy + n) * 4);
            f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            dst[0] = o0; dst[1] = o1;
          } else {
            u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            *(u32x4*)(p.y + ((long)m * p.ldy + n) * 2) = o;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    int n = n0 + wn * 64 + b * 32 + r;
    if (n >= p.Cout) continue;
    float sc = p.scale ? p.scale[n] : 1.f;
    float bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int m = m0 + wm * 64 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = affine<T>(acc[a][b][g], sc, bi);
        if (p.residual) v += Mma<T>::load(p.residual + ((long)m * p.ldr + n) * Mma<T>::ES);
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.relu_mask && !(Mma<T>::load(p.relu_mask + ((long)m * p.ldm + n) * Mma<T>::ES) > 0.f)) v = 0.f;
        if (p.out_f32) *(float*)(p.y + ((long)m * p.ldy + n) * 4) = v;
        else Mma<T>::store(p.y + ((long)m * p.ldy + n) * Mma<T>::ES, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad:  dW[n][k] += scale[n] * sum_m dY[m][n] * A[m][k]
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const char* x;    // NHWC input of the forward conv
  const char* dy;   // [M][ldd] T
  float* dw;        // [Cout][K] f32 (K = KH*KW*Cin), accumulated with atomics
  const float* scale;
  int Nimg, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ldd, pool;
  int M, Kc, cpp, K, ncc;  // ncc = chunks per dY row that exist (Cout*ES/16)
  int mtiles_per_split;
  FastDiv dWo, dHo;
  int xrs;            // x row stride in chunks (default cpp)
  int ldo;            // output row stride in elements (default K)
  int direct;         // 0: f32 atomicAdd (split reductions); 1: plain f32 store; 2: plain T store (single split only)
  long bx, bd, bo;    // batch (gridDim.y) byte strides of x, dy, out
  // k_gemm_tn_small MODE 3 (cddmsl_attnpool_dx): per-batch row vector added to every output row, bit masks of the rows to keep,
  // f32 accumulator of the unmasked rows
  const float* g0 = nullptr;
  const unsigned long long* mbits = nullptr;
  float* gpos = nullptr;
  float* ws = nullptr;  // split reductions through a workspace: block (split, tile) stores its accumulators, in fragment order, at
                        // ws[(split * ntiles + tile) * tile_floats ...]; k_wgrad_reduce sums the splits into dw (see cddmsl_set_workspace)
#ifdef CDDMSL_TILE_STAMPS
  unsigned long long* tstamps = nullptr;   // diagnostic build only (tools/tile_stamps.py)
#endif
};

constexpr int WM = 64;                 // m rows per reduction tile
constexpr int WROW = 16 + 4;           // LDS row = 16 data chunks (256 B) + 4 pad chunks (64 B)

template <typename T> struct TrFrag;
template <> struct TrFrag<__bf16> {
  // Reads the MFMA operand fragment (8 reduction elements for column `col`) out of a row-major
  // [m][col] LDS image with two ds_read_b64_tr_b16: group of 16 lanes <-> 16 columns, lane 4q+p
  // supplies row q, columns 4p..4p+3, and receives its own column's 4 rows.
  __device__ static __forceinline__ u32x4 read(const u32x4* base, int mrow0, int col0, int lane) {
    int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    int hh = g >> 1;                      // lane half = k-group of the MFMA operand
    int col = col0 + 16 * (g & 1) + 4 * pp;
    const char* b = (const char*)base;
    const char* a0 = b + (long)(mrow0 + 8 * hh + q) * (WROW * 16) + col * 2;
    const char* a1 = a0 + 4 * (WROW * 16);
    i16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)a0);
    i16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)a1);
    u32x2 p0 = __builtin_bit_cast(u32x2, v0), p1 = __builtin_bit_cast(u32x2, v1);
    u32x4 r = {p0[0], p0[1], p1[0], p1[1]};
    return r;
  }
  static constexpr int MSTEP = 16;  // reduction elements per Mma step
};
template <> struct TrFrag<float> {
  // f32: Mma<float>::step contracts k = 4h + j; element j of lane (r, h) = image[mrow0 + 4h + j][col0 + r]
  __device__ static __forceinline__ u32x4 read(const u32x4* base, int mrow0, int col0, int lane) {
    int r = lane & 31, hh = lane >> 5;
    const char* b = (const char*)base;
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *(const unsigned int*)(b + (long)(mrow0 + 4 * hh + j) * (WROW * 16) + (col0 + r) * 4);
    return v;
  }
  static constexpr int MSTEP = 8;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(WgradArgs p) {
  constexpr int ES = Mma<T>::ES;
  constexpr int TC = 128 * ES / 16;   // chunks per 128-element tile row (bf16: 16, f32: 32)
  constexpr int COLS = 256 / ES;      // columns held per LDS image row (bf16: 128, f32: 64)
  // f32 tiles are 64 columns wide (256 B rows) so both dtypes share the 256 B + pad row geometry.
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][WM * WROW];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int ntn = (p.Cout + COLS - 1) / COLS, ntk = (p.K + COLS - 1) / COLS;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = bid % ntk; bid /= ntk;
  const int tile_n = bid % ntn; bid /= ntn;
  const int split = bid;
  const int n0 = tile_n * COLS, k0 = tile_k * COLS;
  const int mt0 = split * p.mtiles_per_split;
  const int total_mt = (p.M + WM - 1) / WM;
  const int mt1 = min(mt0 + p.mtiles_per_split, total_mt);
  (void)TC;

  // staging map: 64 rows x 16 chunks = 1024 chunks per operand; thread -> chunk cc = t&15, rows t>>4 + 16 i
  const int cc = t & 15, rb = t >> 4;
  const int kc = k0 * ES / 16 + cc;          // global K chunk of the A-operand (im2col) column
  const bool vk = kc < p.Kc;
  const int pp = vk ? kc / p.cpp : 0, coff = vk ? kc - pp * p.cpp : 0;
  const int ky = pp / p.KW, kx = pp - ky * p.KW;
  const int nc = n0 * ES / 16 + cc;          // chunk along dY's channel axis
  const bool vn = nc < p.ncc;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  u32x4 rx[4], rd[4];

  auto gload = [&](int mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = mt * WM + rb + 16 * i;
      bool vmm = m < p.M;
      unsigned mm = vmm ? m : 0;
      unsigned tq = fdiv(mm, p.dWo);
      int ox = (int)(mm - tq * p.Wo);
      unsigned img = fdiv(tq, p.dHo);
      int oy = (int)(tq - img * p.Ho);
      if (!p.pool) {
        int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
        bool ok = vmm && vk && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        rx[i] = ok ? *(const u32x4*)(p.x + ((((long)img * p.Hi + iy) * p.Wi + ix) * p.cpp + coff) * 16) : zero;
      } else {
        if (vmm && vk) {
          const char* b0 = p.x + ((((long)img * p.Hi + 2 * oy) * p.Wi + 2 * ox) * p.cpp + coff) * 16;
          long rs = (long)p.Wi * p.cpp * 16, cs = (long)p.cpp * 16;
          rx[i] = avg4<T>(*(const u32x4*)b0, *(const u32x4*)(b0 + cs), *(const u32x4*)(b0 + rs), *(const u32x4*)(b0 + rs + cs));
        } else rx[i] = zero;
      }
      rd[i] = (vmm && vn) ? *(const u32x4*)(p.dy + ((long)m * p.ldd) * ES + (long)nc * 16) : zero;
    }
  };

  // wave tiling of the COLS x COLS output tile: 2x2 waves
  constexpr int WT = COLS / 2;       // per-wave extent (bf16: 64, f32: 32)
  constexpr int NT = WT / 32;        // 32x32 tiles per wave per dim (bf16: 2, f32: 1)
  const int wn = wv >> 1, wk = wv & 1;
  f32x16 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (mt0 < mt1) gload(mt0);
  for (int mt = mt0; mt < mt1; ++mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = rb + 16 * i;
      lds[0][row * WROW + cc] = rd[i];
      lds[1][row * WROW + cc] = rx[i];
    }
    __syncthreads();
    if (mt + 1 < mt1) gload(mt + 1);
#pragma unroll
    for (int ms = 0; ms < WM; ms += TrFrag<T>::MSTEP) {
      u32x4 fa[NT], fb[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        fa[i] = TrFrag<T>::read(lds[0], ms, wn * WT + i * 32, lane);
        fb[i] = TrFrag<T>::read(lds[1], ms, wk * WT + i * 32, lane);
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) Mma<T>::step(acc[a][b], fa[a], fb[b]);
    }
    __syncthreads();
  }

  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      int k = k0 + wk * WT + b * 32 + r;
      if (k >= p.K) continue;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int n = n0 + wn * WT + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (n >= p.Cout) continue;
        float v = acc[a][b][g] * (p.scale ? p.scale[n] : 1.f);
        atomicAdd(p.dw + (long)n * p.K + k, v);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad, LDS-DMA variant for stride-1 "same" convolutions (1x1/pad 0 and 3x3/pad 1: every trainable layer).
// Output pixel index == input pixel index, so both operands are walked with pointer increments (64 rows per tile);
// tiles go global -> LDS by global_load_lds (no VGPR -> LDS write pass, the limiter of the register-staged kernel
// at two blocks per CU), double-buffered with a counted vmcnt.  LDS rows are plain 256 B; the 16-byte chunk index is
// XOR-swizzled by f(row) = ((row&3)<<2) | ((row>>2)&3) -- applied to the SOURCE chunk a lane fetches and to the
// transposed reads -- which keeps ds_read_b64_tr_b16 conflict-free without row padding (padding is impossible with
// lane-linear DMA writes).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int fsw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// Fragment addressing is split into a per-lane part computed ONCE (fsw of rows ms + 8hh + q does not depend on the
// 16-row step ms, since ms % 16 == 0) and a compile-time row offset ms * 256 that folds into the DS immediate.
template <typename T> struct TrFragS;
template <> struct TrFragS<__bf16> {
  struct Off { int o0, o1; };
  __device__ static __forceinline__ Off prep(int col0, int lane) {
    int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    int hh = g >> 1;
    int ch = ((col0 + 16 * (g & 1)) >> 3) + (pp >> 1);       // logical 16-byte chunk of columns 4pp..4pp+3
    int r0 = 8 * hh + q, r1 = r0 + 4;
    Off o;
    o.o0 = r0 * 256 + 16 * (ch ^ fsw(r0)) + 8 * (pp & 1);
    o.o1 = r1 * 256 + 16 * (ch ^ fsw(r1)) + 8 * (pp & 1);
    return o;
  }
  // Inline asm on purpose: through the builtin, hipcc treats the transposed read as "may alias the in-flight LDS-DMA"
  // and drains vmcnt(0) before it (no overlap with the next tile's DMA).  The asm reads are ordered by the caller's
  // counted vmcnt + barrier before, and by an explicit lgkmcnt(0) + sched_barrier after (tr_wait()).
  __device__ static __forceinline__ u32x4 read(const u32x4* base, int ms, const Off& o) {
    const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)((const char*)base) + ms * 256;
    u32x2 p0, p1;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(p0) : "v"(a + o.o0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(p1) : "v"(a + o.o1));
    u32x4 r = {p0[0], p0[1], p1[0], p1[1]};
    return r;
  }
  static constexpr int MSTEP = 16;
};
template <> struct TrFragS<float> {
  struct Off { int o[4]; };
  __device__ static __forceinline__ Off prep(int col0, int lane) {
    int r = lane & 31, hh = lane >> 5;
    int byte = (col0 + r) * 4;
    Off o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int row = 4 * hh + j;                                  // + ms (multiple of 8): fsw(row + 8k) flips bit 1 of (row>>2)&3
      o.o[j] = row * 256 + (byte & 15) + 16 * (byte >> 4);   // swizzle applied in read (depends on ms & 8)
    }
    return o;
  }
  __device__ static __forceinline__ u32x4 read(const u32x4* base, int ms, const Off& o) {
    const char* b = (const char*)base;
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int off = o.o[j] + ms * 256;
      int row = off >> 8, ch = (off >> 4) & 15;
      v[j] = *(const unsigned int*)(b + (off & ~0xf0) + 16 * (ch ^ fsw(row)));
    }
    return v;
  }
  static constexpr int MSTEP = 8;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_dma(WgradArgs p) {
  constexpr int ES = Mma<T>::ES;
  constexpr int COLS = 256 / ES;
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][2][WM * 16];   // [buffer][dY | X][64 rows x 16 chunks]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  p.x += (long)blockIdx.y * p.bx; p.dy += (long)blockIdx.y * p.bd;
  char* outp = (char*)p.dw + (long)blockIdx.y * p.bo;
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntn = (p.Cout + COLS - 1) / COLS, ntk = (p.K + COLS - 1) / COLS;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = bid % ntk; bid /= ntk;
  const int tile_n = bid % ntn; bid /= ntn;
  const int split = bid;
  const int n0 = tile_n * COLS, k0 = tile_k * COLS;
  const int mt0 = split * p.mtiles_per_split;
  const int total_mt = (p.M + WM - 1) / WM;
  const int mt1 = min(mt0 + p.mtiles_per_split, total_mt);

  const int cc = t & 15, rb = t >> 4;          // physical chunk / row of this lane's DMA slots (rows rb + 16 i)
  const int cl = cc ^ fsw(rb);                 // logical chunk it fetches (fsw(rb + 16 i) == fsw(rb))
  const int kc = k0 * ES / 16 + cl;
  const bool vk = kc < p.Kc;
  const int pp = vk ? kc / p.cpp : 0, coff = vk ? kc - pp * p.cpp : 0;
  const int ky = pp / p.KW, kx = pp - ky * p.KW;
  const int nc = n0 * ES / 16 + cl;
  const bool vn = nc < p.ncc;
  const bool taps = !(p.KH == 1 && p.KW == 1);
  const char* zp = (const char*)g_zero_page;

  int m[4];
  const char* pd[4];
  const char* px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[i] = mt0 * WM + rb + 16 * i;
    pd[i] = p.dy + ((long)m[i] * p.ldd) * ES + (long)nc * 16;
    px[i] = p.x + (((long)m[i] + (long)(ky - p.pad) * p.Wi + (kx - p.pad)) * p.xrs) * 16 + (long)coff * 16;
  }
  const long dstep = (long)WM * p.ldd * ES, xstep = (long)WM * p.xrs * 16;

  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool vmm = m[i] < p.M;
      glds16((vmm && vn) ? pd[i] : zp, &lds[buf][0][(4 * wvu + 16 * i) * 16]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = (m[i] < p.M) & vk;
      if (taps) {                                  // branch-free: every lane does the (cheap) divisions
        unsigned mm = min((unsigned)m[i], (unsigned)(p.M - 1));
        unsigned tq = fdiv(mm, p.dWo);
        unsigned ox = mm - tq * p.Wo;
        unsigned oy = tq - fdiv(tq, p.dHo) * p.Ho;
        ok = ok & ((unsigned)((int)oy - p.pad + ky) < (unsigned)p.Hi) & ((unsigned)((int)ox - p.pad + kx) < (unsigned)p.Wi);
      }
      glds16(ok ? px[i] : zp, &lds[buf][1][(4 * wvu + 16 * i) * 16]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { m[i] += WM; pd[i] += dstep; px[i] += xstep; }
  };

  constexpr int WT = COLS / 2;
  constexpr int NT = WT / 32;
  const int wn = wv >> 1, wk = wv & 1;
  f32x16 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  typename TrFragS<T>::Off offa[NT], offb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    offa[i] = TrFragS<T>::prep(wn * WT + i * 32, lane);
    offb[i] = TrFragS<T>::prep(wk * WT + i * 32, lane);
  }

  if (mt0 < mt1) stage(0);
  for (int mt = mt0; mt < mt1; ++mt) {
    const int cur = (mt - mt0) & 1;
    if (mt + 1 < mt1) {
      stage(cur ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ms = 0; ms < WM; ms += TrFragS<T>::MSTEP) {
      u32x4 fa[NT], fb[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        fa[i] = TrFragS<T>::read(lds[cur][0], ms, offa[i]);
        fb[i] = TrFragS<T>::read(lds[cur][1], ms, offb[i]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // asm reads are invisible to hipcc's own waitcnt pass
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) Mma<T>::step(acc[a][b], fa[a], fb[b]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  const int r = lane & 31, h = lane >> 5;
  if (p.ws) {          // split reduction through the workspace: 16-byte stores in fragment order, 1 KiB per wave instruction
    f32x4* dst = (f32x4*)p.ws + ((long)(split * ntn + tile_n) * ntk + tile_k) * (4 * NT * NT * 4 * 64) + (wv * NT * NT * 4) * 64 + lane;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = {acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
          dst[((a * NT + b) * 4 + g4) * 64] = v;
        }
    return;
  }
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      int k = k0 + wk * WT + b * 32 + r;
      if (k >= p.K) continue;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int n = n0 + wn * WT + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (n >= p.Cout) continue;
        float v = acc[a][b][g] * (p.scale ? p.scale[n] : 1.f);
        long o = (long)n * p.ldo + k;
        if (p.direct == 0) atomicAdd((float*)outp + o, v);
        else if (p.direct == 1) ((float*)outp)[o] = v;
        else Mma<T>::store(outp + o * ES, v);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming batched TN GEMM for SHORT reductions (the attention pool's per-region products: 56 or 64 reduction rows,
// thousands of regions): out_b[n][k] = sum_m A_b[m][n] B_b[m][k].  One block = one (n-tile, k-tile) column of outputs
// for a RUN of batches: the double-buffered LDS-DMA pipeline of k_conv_wgrad_dma keeps running across batch boundaries
// (the next batch's tile is in flight while this one is reduced and stored), where one-block-per-batch launches paid a
// full global-memory latency per 32 KiB tile.  Same LDS images, swizzle and transposed reads as k_conv_wgrad_dma.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_stream(WgradArgs p, int nbatch, int bpb) {
  constexpr int ES = Mma<T>::ES;
  constexpr int COLS = 256 / ES;
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][2][WM * 16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntk = (p.K + COLS - 1) / COLS;
  const int tile_k = blockIdx.x % ntk, tile_n = blockIdx.x / ntk;
  const int n0 = tile_n * COLS, k0 = tile_k * COLS;
  const int b0 = blockIdx.y * bpb, nb = min(bpb, nbatch - b0);
  const int nmt = (p.M + WM - 1) / WM;
  const int nit = nb * nmt;

  const int cc = t & 15, rb = t >> 4;
  const int cl = cc ^ fsw(rb);
  const int kc = k0 * ES / 16 + cl, nc = n0 * ES / 16 + cl;
  const bool vk = kc < p.Kc, vn = nc < p.ncc;
  const char* zp = (const char*)g_zero_page;
  int sb = 0, smt = 0;                               // (batch, reduction tile) of the next tile to stage
  auto stage = [&](int buf) {
    const char* db = p.dy + (long)(b0 + sb) * p.bd + (long)nc * 16;
    const char* xb = p.x + (long)(b0 + sb) * p.bx + (long)kc * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = smt * WM + rb + 16 * i;
      const bool vm = m < p.M;
      glds16((vm && vn) ? db + ((long)m * p.ldd) * ES : zp, &lds[buf][0][(4 * wvu + 16 * i) * 16]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = smt * WM + rb + 16 * i;
      const bool vm = m < p.M;
      glds16((vm && vk) ? xb + ((long)m * p.xrs) * 16 : zp, &lds[buf][1][(4 * wvu + 16 * i) * 16]);
    }
    if (++smt == nmt) { smt = 0; ++sb; }
  };

  constexpr int WT = COLS / 2;
  constexpr int NT = WT / 32;
  const int wn = wv >> 1, wk = wv & 1;
  f32x16 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  typename TrFragS<T>::Off offa[NT], offb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    offa[i] = TrFragS<T>::prep(wn * WT + i * 32, lane);
    offb[i] = TrFragS<T>::prep(wk * WT + i * 32, lane);
  }
  const int r = lane & 31, h = lane >> 5;

  if (nit > 0) stage(0);
  int cb = 0, cmt = 0;                               // (batch, reduction tile) being reduced
  for (int it = 0; it < nit; ++it) {
    const int cur = it & 1;
    if (it + 1 < nit) {
      stage(cur ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ms = 0; ms < WM; ms += TrFragS<T>::MSTEP) {
      u32x4 fa[NT], fb[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        fa[i] = TrFragS<T>::read(lds[cur][0], ms, offa[i]);
        fb[i] = TrFragS<T>::read(lds[cur][1], ms, offb[i]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) Mma<T>::step(acc[a][b], fa[a], fb[b]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (++cmt == nmt) {                              // batch complete: store its tile (the next batch's DMA is already in flight)
      char* outp = (char*)p.dw + (long)(b0 + cb) * p.bo;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int k = k0 + wk * WT + b * 32 + r;
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int n = n0 + wn * WT + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
            if (k < p.K && n < p.Cout) {
              const float v = acc[a][b][g];
              const long o = (long)n * p.ldo + k;
              if (p.direct == 0) atomicAdd((float*)outp + o, v);
              else if (p.direct == 1) ((float*)outp)[o] = v;
              else Mma<T>::store(outp + o * ES, v);
            }
            acc[a][b][g] = 0.f;
          }
        }
      cmt = 0; ++cb;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Compact streaming TN GEMM for the attention pool's per-region products when the whole reduction is ONE tile
// (M <= 64 rows) and the output is narrow (N <= 64): out_b[n][k] = sum_m A_b[m][n] B_b[m][k], bf16.
// k_gemm_tn_stream spends a 16 KiB LDS image per stage on an A operand of 2-7 KiB and keeps one stage in flight per
// block; these products are pure streaming (2.9 GB per call), so what matters is bytes in flight.  Here a stage is a
// compact A image (64 rows x 128 B) + the B image (64 rows x 256 B) = 24 KiB, three stages form a ring (two in flight,
// counted vmcnt(12)), two blocks fit a CU.  Block = one 128-column k-tile for a run of batches; wave w owns columns
// 32w..32w+31 for all (one or two) 32-row n-tiles.  Both operands are read transposed (ds_read_b64_tr_b16).
// vmcnt counts stores too and retires in issue order, so the wait for stage `it` has to allow for the previous item's
// output stores that sit between the DMAs: N = 8*NG is a template parameter and K % 128 == 0 so that this count (4*NG
// store instructions per wave and item) is a compile-time constant.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// MODE 3 (the attention pool's input gradient, cddmsl_attnpool_dx; P = pixels per region): the product's rows are the token
// gradients  dtok[t] = sum_h (p[h][t] dZ[h] + dS[h][t] U[h]),  t = 0 the mean token.  Instead of storing them (and reading them back in
// a second kernel) the epilogue writes the map's gradient directly,  dx[t-1] = dtok[t] + (dtok[0] + g0) / P  for t = 1..P, zeroed where
// bit t-1 of the column's mask word is clear (the pooled map is a ReLU output: its sign bits, one 64-bit word per region and column,
// written by cddmsl_attn_tokens_fwd), and keeps the UNMASKED column sums of dtok[t] (t = 0 includes g0: the query path's gradient of
// the mean token) in registers over the block's run of regions: the positional embedding's gradient, one atomic per element at the end.
// Two more loads per wave and item (g0, mask word), issued in front of the item's stage so that the counted waits stay exact.
template <int NG, int P>
constexpr int tn_small_stores3() {
  int n = 0;
  for (int a = 0; a < (NG + 3) / 4; ++a)
    for (int g = 0; g < 16; ++g) {
      if (a * 4 + (g >> 2) >= NG) continue;
      const int n0 = a * 32 + (g & 3) + 8 * (g >> 2);
      if ((n0 >= 1 && n0 <= P) || (n0 + 4 >= 1 && n0 + 4 <= P)) ++n;
    }
  return n;
}
template <int NG, int MODE, int P = 0>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_small(WgradArgs p, int nbatch, int bpb) {
  constexpr int NTN = (NG + 3) / 4, NSTORE = MODE == 3 ? tn_small_stores3<NG, P>() : 4 * NG, NML = MODE == 3 ? 2 : 0;
  constexpr int STAGE = 8192 + 16384, NST = 3;
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const int t = threadIdx.x, lane = t & 63;
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int k0 = blockIdx.x * 128;
  const int b0 = blockIdx.y * bpb, nit = min(bpb, nbatch - b0);
  const char* zp = (const char*)g_zero_page;
  // DMA slots: A image chunk q = i*256 + t -> row q>>3, chunk q&7 (plain); B image chunk q -> row q>>4, slot q&15 (chunk ^= fsw(row))
  const int ar = t >> 3, ac = t & 7;
  const bool va = ac * 8 < p.Cout;
  const int xr = t >> 4, xc = (t & 15) ^ fsw(t >> 4);
  const int kc = (k0 >> 3) + xc;
  const bool vk = kc < p.Kc;
  int sb = 0;
  auto stage = [&](int slot) {
    char* base = lds + slot * STAGE;
    const char* db = p.dy + (long)(b0 + sb) * p.bd + ac * 16;
    const char* xb = p.x + (long)(b0 + sb) * p.bx + (long)kc * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = ar + 32 * i;
      glds16((va && m < p.M) ? db + ((long)m * p.ldd) * 2 : zp, base + (i * 256 + wvu * 64) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = xr + 16 * i;
      glds16((vk && m < p.M) ? xb + ((long)m * p.xrs) * 16 : zp, base + 8192 + (i * 256 + wvu * 64) * 16);
    }
    ++sb;
  };
  f32x16 acc[NTN];
#pragma unroll
  for (int a = 0; a < NTN; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  // transposed-read addresses inside a stage
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  unsigned aoff[NTN][2], xoff[2];
  {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3, hh = g >> 1;
#pragma unroll
    for (int a = 0; a < NTN; ++a) {
      const int col = a * 32 + 16 * (g & 1) + 4 * pp;
      aoff[a][0] = (8 * hh + q) * 128 + col * 2;
      aoff[a][1] = aoff[a][0] + 4 * 128;
    }
    const TrFragS<__bf16>::Off o = TrFragS<__bf16>::prep(wvu * 32, lane);
    xoff[0] = 8192 + o.o0; xoff[1] = 8192 + o.o1;
  }
  const int r = lane & 31, h = lane >> 5;
  f32x16 gacc[MODE == 3 ? NTN : 1];
  if (MODE == 3) {
#pragma unroll
    for (int a = 0; a < NTN; ++a)
#pragma unroll
      for (int g = 0; g < 16; ++g) gacc[a][g] = 0.f;
  }
  if (nit > 0) stage(0);
  if (nit > 1) stage(1);
  // One item.  KIND (compile time): 2 = two more items follow (stage it+2 is issued here), 1 = one more follows, 0 = the last.
  // The three kinds run as three pieces of straight-line code (loop, tail, tail), so that MODE 3's two per-item loads -- plain
  // loads the compiler waits for by itself, counting the DMA instructions issued behind them -- are not merged across paths with
  // different numbers of younger operations (a merged path waits for vmcnt(0), i.e. for the DMAs just issued).
  auto item = [&](int it, auto KIND) {
    constexpr int kind = decltype(KIND)::value;
    const int slot = it % NST;
    float g0v = 0.f;
    u32x2 mbv = {~0u, ~0u};
    if (MODE == 3) {                                 // this item's row vector and mask word: requested BEFORE stage it+2
      // (inline asm + a counted wait below: hipcc's own wait for a plain load issued in front of LDS-DMA instructions is vmcnt(0))
      const long ci = (long)(b0 + it) * p.K + k0 + wvu * 32 + r;
      const char* gp = (const char*)(p.g0 + ci);
      const char* mp = (const char*)(p.mbits + ci);
      asm volatile("global_load_dword %0, %1, off" : "=v"(g0v) : "v"(gp) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(mbv) : "v"(mp) : "memory");
    }
    // younger than stage `it`: stage it+1 (6 DMAs), the stores of item it-1, [the two loads above,] stage it+2 (6 DMAs)
    if (kind == 2) {
      stage((it + 2) % NST);
      if (it) wait_vm<12 + NML + NSTORE>(); else wait_vm<12 + NML>();
    } else if (kind == 1) {
      if (it) wait_vm<6 + NML + NSTORE>(); else wait_vm<6 + NML>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    const unsigned sbase = lbase + slot * STAGE;
#define CDDMSL_TRS(DST, A0, A1, IMM)                                                             \
  { u32x2 q0_, q1_;                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q0_) : "v"(A0), "i"(IMM));         \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q1_) : "v"(A1), "i"(IMM));         \
    DST = u32x4{q0_[0], q0_[1], q1_[0], q1_[1]}; }
    u32x4 fa[NTN][4], fb[4];
    const unsigned x0 = sbase + xoff[0], x1 = sbase + xoff[1];
    CDDMSL_TRS(fb[0], x0, x1, 0) CDDMSL_TRS(fb[1], x0, x1, 4096) CDDMSL_TRS(fb[2], x0, x1, 8192) CDDMSL_TRS(fb[3], x0, x1, 12288)
#pragma unroll
    for (int a = 0; a < NTN; ++a) {
      const unsigned a0 = sbase + aoff[a][0], a1 = sbase + aoff[a][1];
      CDDMSL_TRS(fa[a][0], a0, a1, 0) CDDMSL_TRS(fa[a][1], a0, a1, 2048) CDDMSL_TRS(fa[a][2], a0, a1, 4096) CDDMSL_TRS(fa[a][3], a0, a1, 6144)
    }
#undef CDDMSL_TRS
    if (NTN == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fa[0][0]), "+v"(fa[0][1]),
                               "+v"(fa[0][2]), "+v"(fa[0][3]) :: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]),
                      "+v"(fa[0][3]), "+v"(fa[NTN - 1][0]), "+v"(fa[NTN - 1][1]), "+v"(fa[NTN - 1][2]), "+v"(fa[NTN - 1][3]) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int a = 0; a < NTN; ++a) Mma<__bf16>::step(acc[a], fa[a][ms], fb[ms]);
    __builtin_amdgcn_s_barrier();                    // every wave is done with this slot before the next iteration restages it
    char* outp = (char*)p.dw + (long)(b0 + it) * p.bo;
    const int k = k0 + wvu * 32 + r;
    if (MODE == 3) {
      // the two loads are older than stage it+2's six DMAs (kind 2); the registers are named by the wait so that no use precedes it
      if (kind == 2) asm volatile("s_waitcnt vmcnt(6)" : "+v"(g0v), "+v"(mbv) :: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0v), "+v"(mbv) :: "memory");
      // row 0 (the mean token) of this column sits in lane r (< 32), register 0: v_permlane32_swap hands the lower half-wave's values
      // to the upper one (not a DS instruction: a compiler-visible LDS operation here would make hipcc wait for the LDS-DMAs in flight)
      const unsigned a00 = __builtin_bit_cast(unsigned, acc[0][0]);
      const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(a00, a00, false, false)[0]) + g0v;
      const float base = t0 * (1.0f / (float)(P > 0 ? P : 1));
      const unsigned mlo = mbv[0], mhi = mbv[1];
#pragma unroll
      for (int a = 0; a < NTN; ++a)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int n0 = a * 32 + (g & 3) + 8 * (g >> 2);
          const bool exists = a * 4 + (g >> 2) < NG;
          const bool lo_ok = n0 >= 1 && n0 <= P, hi_ok = n0 + 4 >= 1 && n0 + 4 <= P;       // (compile-time after unrolling)
          if (exists) {
            const int n = n0 + 4 * h;
            const float v = acc[a][g];
            gacc[a][g] += (n == 0) ? t0 : v;
            if (lo_ok || hi_ok) {
              if (h ? hi_ok : lo_ok) {
                const unsigned word = (n - 1) < 32 ? mlo : mhi;
                const bool keep = (word >> ((n - 1) & 31)) & 1u;
                Mma<__bf16>::store(outp + ((long)(n - 1) * p.ldo + k) * 2, keep ? v + base : 0.f);
              }
            }
          }
          acc[a][g] = 0.f;
        }
      return;
    }
#pragma unroll
    for (int a = 0; a < NTN; ++a)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (a * 4 + (g >> 2) < NG) {                 // rows 8*(4a + g/4) .. +7 exist: exactly NSTORE stores per item
          const int n = a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
          const float v = acc[a][g];
          const long o = (long)n * p.ldo + k;
          if (MODE == 0) atomicAdd((float*)outp + o, v);
          else if (MODE == 1) ((float*)outp)[o] = v;
          else Mma<__bf16>::store(outp + o * 2, v);
        }
        acc[a][g] = 0.f;
      }
  };
  {
    int it = 0;
    for (; it + 2 < nit; ++it) item(it, std::integral_constant<int, 2>{});
    if (it + 1 < nit) { item(it, std::integral_constant<int, 1>{}); ++it; }
    if (it < nit) item(it, std::integral_constant<int, 0>{});
  }
  if (MODE == 3 && p.gpos) {
    const int k = k0 + wvu * 32 + r;
#pragma unroll
    for (int a = 0; a < NTN; ++a)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int n = a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (a * 4 + (g >> 2) < NG && n <= P) atomicAdd(p.gpos + (long)n * p.K + k, gacc[a][g]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad / TN GEMM on the 256x256 ping-pong structure of k_conv_fwd256 (bf16): output tile 256 n x 256 k, reduction
// tiles of 64 m rows, 8 waves (2 over n x 4 over k; 128 n x 64 k per wave), the two wave groups one barrier apart.
// Operands stay row-major in LDS ([64 rows][256 B] images, chunk ^= fsw(row)) and are read transposed
// (ds_read_b64_tr_b16).  A half-tile = one 16 KiB image: dY half h = the 64 columns {wn*128 + h*64 ..} of both wave
// rows, X half j = the 32 columns {wk*64 + j*32 ..} of all four wave columns.  Phases, restaging distance and the
// counted vmcnt are those of k_conv_fwd256.  Sources are buffer-addressed: per-lane offset constant, the m walk in soffset;
// rows past M and out-of-image filter taps set bit 31 of the lane offset (-> zeros).  Tap validity is recomputed per
// reduction tile for the lane's two rows (2 fdiv each) in the phase that stages the first X half.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_wgrad256(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2 * 2 * 2 * 64 * 16];   // byte = buf<<16 | ab<<15 | half<<14 | row*256 + slot*16
  const int t = threadIdx.x, lane = t & 63;
#ifdef CDDMSL_TILE_STAMPS
  const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long ts_loop = ts_entry;
#endif
  p.x += (long)blockIdx.y * p.bx; p.dy += (long)blockIdx.y * p.bd;
  char* outp = (char*)p.dw + (long)blockIdx.y * p.bo;
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wvu >> 2, wk = wvu & 3;
  const int ntn = p.Cout >> 8, ntk = p.K >> 8;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = bid % ntk; bid /= ntk;
  const int tile_n = bid % ntn; bid /= ntn;
  const int n0 = tile_n * 256, k0 = tile_k * 256;
  const int mt0 = bid * p.mtiles_per_split;
  const int nmt = min(p.mtiles_per_split, (p.M + WM - 1) / WM - mt0);
  const bool taps = !(p.KH == 1 && p.KW == 1);

  // ---- staging: thread -> LDS slot (row i*32 + (t>>4), slot t&15), logical chunk cl of that slot
  const int rb = t >> 4, cl = (t & 15) ^ fsw(rb);
  const int gd = (cl >> 3) * 16 + (cl & 7);                 // dY chunk within the 256-column tile (+ 8 per half: immediate)
  const int gx = (cl >> 2) * 8 + (cl & 3);                  // X chunk within the 256-column tile (+ 4 per half: immediate)
  const int kc = (k0 >> 3) + gx;
  const int pp = kc / p.cpp, coff = kc - pp * p.cpp;
  const int ky = pp / p.KW, kx = pp - ky * p.KW;
  unsigned vd[2], vx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    vd[i] = (unsigned)((i * 32 + rb) * p.ldd * 2 + gd * 16);
    vx[i] = (unsigned)((((i * 32 + rb) + ky * p.Wi + kx) * p.xrs + coff) * 16);
  }
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + ((long)mt0 * WM * p.ldd + n0) * 2), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + ((long)mt0 * WM - p.pad * p.Wi - p.pad) * p.xrs * 16), 0, 0x80000000u, 0x00020000);
  const unsigned dstep = (unsigned)(WM * p.ldd * 2), xstep = (unsigned)(WM * p.xrs * 16);
  const int mrow = mt0 * WM + rb;                            // + T*64 + i*32

  char* const L = (char*)lds;
  auto stageD = [&](auto H, int buf, int T) {                // dY half h of reduction tile T (relative to mt0)
    constexpr int h = decltype(H)::value;
    char* dst = L + (buf << 16) + (h << 14) + wvu * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned inv = (mrow + T * WM + i * 32 < p.M) ? 0u : 0x80000000u;
      // the instruction's immediate offset moves BOTH the global and the LDS address: take it back out of the LDS base (M0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(dst + i * 8192 - h * 128), 16,
                                               (int)(vd[i] | inv), (int)(T * dstep), h * 128, 0);
    }
  };
  unsigned xinv[2];
  auto validX = [&](int T) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mrow + T * WM + i * 32;
      bool ok = m < p.M;
      if (taps) {
        const unsigned mm = min((unsigned)m, (unsigned)(p.M - 1));
        const unsigned tq = fdiv(mm, p.dWo);
        const unsigned ox = mm - tq * p.Wo;
        const unsigned oy = tq - fdiv(tq, p.dHo) * p.Ho;
        ok = ok & ((unsigned)((int)oy - p.pad + ky) < (unsigned)p.Hi) & ((unsigned)((int)ox - p.pad + kx) < (unsigned)p.Wi);
      }
      xinv[i] = ok ? 0u : 0x80000000u;
    }
  };
  auto stageX = [&](auto J, int buf, int T) {
    constexpr int j = decltype(J)::value;
    char* dst = L + (buf << 16) + (1 << 15) + (j << 14) + wvu * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + i * 8192 - j * 64), 16,
                                               (int)(vx[i] | xinv[i]), (int)(T * xstep), j * 64, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- transposed fragment reads: per-lane byte addresses (absolute LDS), buffer bit toggled by XOR, half / 16-row step as immediates
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)L;
  unsigned adA[2][2], adB[2];
  {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const TrFragS<__bf16>::Off o = TrFragS<__bf16>::prep(wn * 64 + a * 32, lane);
      adA[a][0] = lbase + o.o0; adA[a][1] = lbase + o.o1;
    }
    const TrFragS<__bf16>::Off o = TrFragS<__bf16>::prep(wk * 32, lane);
    adB[0] = lbase + (1u << 15) + o.o0; adB[1] = lbase + (1u << 15) + o.o1;
  }
  u32x4 fa0[2][4], fa1[2][4], fb0[4], fb1[4];
#define CDDMSL_TR2(DST, A0, A1, IMM)                                                             \
  { u32x2 q0_, q1_;                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q0_) : "v"(A0), "i"(IMM));         \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q1_) : "v"(A1), "i"(IMM));         \
    DST = u32x4{q0_[0], q0_[1], q1_[0], q1_[1]}; }
#define CDDMSL_READ_A(HALF, FA)                                                                  \
  _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                \
    CDDMSL_TR2(FA[a][0], adA[a][0], adA[a][1], ((HALF) << 14) + 0 * 4096)                        \
    CDDMSL_TR2(FA[a][1], adA[a][0], adA[a][1], ((HALF) << 14) + 1 * 4096)                        \
    CDDMSL_TR2(FA[a][2], adA[a][0], adA[a][1], ((HALF) << 14) + 2 * 4096)                        \
    CDDMSL_TR2(FA[a][3], adA[a][0], adA[a][1], ((HALF) << 14) + 3 * 4096) }
#define CDDMSL_READ_B(HALF, FB)                                                                  \
  CDDMSL_TR2(FB[0], adB[0], adB[1], ((HALF) << 14) + 0 * 4096)                                   \
  CDDMSL_TR2(FB[1], adB[0], adB[1], ((HALF) << 14) + 1 * 4096)                                   \
  CDDMSL_TR2(FB[2], adB[0], adB[1], ((HALF) << 14) + 2 * 4096)                                   \
  CDDMSL_TR2(FB[3], adB[0], adB[1], ((HALF) << 14) + 3 * 4096)
#define CDDMSL_FLIP_A() { adA[0][0] ^= 1u << 16; adA[0][1] ^= 1u << 16; adA[1][0] ^= 1u << 16; adA[1][1] ^= 1u << 16; }
#define CDDMSL_FLIP_B() { adB[0] ^= 1u << 16; adB[1] ^= 1u << 16; }
#define CDDMSL_MMA_QUAD(I, J, FA, FB)                                               \
  _Pragma("unroll") for (int ms = 0; ms < 4; ++ms) {                                \
    Mma<__bf16>::step(acc[2 * (I)][J], FA[0][ms], FB[ms]);                          \
    Mma<__bf16>::step(acc[2 * (I) + 1][J], FA[1][ms], FB[ms]);                      \
  }
// The transposed reads are inline asm (see TrFragS): the wait that retires them names the fragments as read-write
// operands, so no MFMA that consumes them can be placed above it; the empty statement after a quadrant's MFMAs names its
// accumulators, so those MFMAs cannot sink below the phase's closing barrier (register-only instructions are otherwise
// free to cross barriers and sched_barrier alike).
#define CDDMSL_WAIT4(F) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) :: "memory");
#define CDDMSL_WAIT8(F) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[0][2]), "+v"(F[0][3]), \
                                     "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[1][2]), "+v"(F[1][3]) :: "memory");
#define CDDMSL_PHASE_SYNC_IN(WAIT)                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                \
  __builtin_amdgcn_s_barrier();                                                     \
  WAIT                                                                              \
  __builtin_amdgcn_sched_barrier(0);                                                \
  __builtin_amdgcn_s_setprio(1);
#define CDDMSL_PHASE_SYNC_OUT(I, J)                                                 \
  asm volatile("" : "+v"(acc[2 * (I)][J]), "+v"(acc[2 * (I) + 1][J]));              \
  __builtin_amdgcn_s_setprio(0);                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                \
  __builtin_amdgcn_s_barrier();                                                     \
  __builtin_amdgcn_sched_barrier(0);

  if (nmt > 0) {
    // prologue: tile 0 complete, tile 1 without its dY half 1 (staged by phase 1 of tile 0); dY half 0 of tile 0 is read ahead
    validX(0);
    stageD(I0{}, 0, 0); stageD(I1{}, 0, 0); stageX(I0{}, 0, 0); stageX(I1{}, 0, 0);
    if (nmt > 1) {
      validX(1);
      stageD(I0{}, 1, 1); stageX(I0{}, 1, 1); stageX(I1{}, 1, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CDDMSL_READ_A(0, fa0)
    CDDMSL_WAIT8(fa0)
    if (wn == 1) __builtin_amdgcn_s_barrier();     // group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);
#ifdef CDDMSL_TILE_STAMPS
    ts_loop = __builtin_amdgcn_s_memrealtime();
#endif

    for (int kt = 0; kt < nmt; ++kt) {
      const int d = kt & 1;
      const bool more1 = kt + 1 < nmt, more2 = kt + 2 < nmt;
      // phase 1
      CDDMSL_READ_B(0, fb0)
      if (more1) stageD(I1{}, d ^ 1, kt + 1);
      CDDMSL_PHASE_SYNC_IN(CDDMSL_WAIT4(fb0))
      CDDMSL_MMA_QUAD(0, 0, fa0, fb0);
      CDDMSL_PHASE_SYNC_OUT(0, 0)
      // phase 2
      CDDMSL_READ_B(1, fb1)
      CDDMSL_FLIP_B()
      if (more2) { stageD(I0{}, d, kt + 2); validX(kt + 2); }     // (the tap tests of the X stages of phases 3 and 4: this phase has the lighter load part)
      CDDMSL_PHASE_SYNC_IN(CDDMSL_WAIT4(fb1))
      CDDMSL_MMA_QUAD(0, 1, fa0, fb1);
      CDDMSL_PHASE_SYNC_OUT(0, 1)
      // phase 3
      CDDMSL_READ_A(1, fa1)
      CDDMSL_FLIP_A()
      if (more2) {
        stageX(I0{}, d, kt + 2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      CDDMSL_PHASE_SYNC_IN(CDDMSL_WAIT8(fa1))
      CDDMSL_MMA_QUAD(1, 1, fa1, fb1);
      CDDMSL_PHASE_SYNC_OUT(1, 1)
      // phase 4
      if (more1) { CDDMSL_READ_A(0, fa0) }
      if (more2) stageX(I1{}, d, kt + 2);
      CDDMSL_PHASE_SYNC_IN(CDDMSL_WAIT8(fa0))
      CDDMSL_MMA_QUAD(1, 0, fa1, fb0);
      CDDMSL_PHASE_SYNC_OUT(1, 0)
    }
    if (wn == 0) __builtin_amdgcn_s_barrier();
  }
#undef CDDMSL_TR2
#undef CDDMSL_WAIT4
#undef CDDMSL_WAIT8
#undef CDDMSL_READ_A
#undef CDDMSL_READ_B
#undef CDDMSL_FLIP_A
#undef CDDMSL_FLIP_B
#undef CDDMSL_MMA_QUAD
#undef CDDMSL_PHASE_SYNC_IN
#undef CDDMSL_PHASE_SYNC_OUT

#ifdef CDDMSL_TILE_STAMPS
  const unsigned long long ts_epi = __builtin_amdgcn_s_memrealtime();
#endif
  const int r = lane & 31, h = lane >> 5;
  if (p.ws) {          // split reduction through the workspace (see k_conv_wgrad_dma): 32 x 16 bytes per lane instead of 128 atomics
    f32x4* dst = (f32x4*)p.ws + ((long)(bid * ntn + tile_n) * ntk + tile_k) * (8 * 32 * 64) + (wvu * 32) * 64 + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = {acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
          dst[((a * 2 + b) * 4 + g4) * 64] = v;
        }
  } else
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int k = k0 + wk * 64 + b * 32 + r;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int n = n0 + wn * 128 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        const float v = acc[a][b][g] * (p.scale ? p.scale[n] : 1.f);
        const long o = (long)n * p.ldo + k;
        if (p.direct == 0) atomicAdd((float*)outp + o, v);
        else if (p.direct == 1) ((float*)outp)[o] = v;
        else Mma<__bf16>::store(outp + o * 2, v);
      }
    }
#ifdef CDDMSL_TILE_STAMPS
  if (p.tstamps && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = p.tstamps + ((long)blockIdx.x * 8 + wvu) * 4;
    o[0] = ts_entry; o[1] = ts_loop; o[2] = ts_epi; o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// fp8 configuration (BASELINE.json configs[4]): the weight gradient of a "same" convolution on the e4m3 copies of BOTH operands
// (the activation's copy its producer wrote for the forward convolution, the output gradient's copy made for the input-gradient
// convolution), v_mfma_scale_f32_32x32x64_f8f6f4.  Output tile 256 n x 256 k, 8 waves of 128 n x 64 k as in k_wgrad256 (same
// accumulator layout: the same epilogue and k_wgrad_reduce).  Reduction tiles of 64 pixels = ONE MFMA step: an image is
// [64 pixels][256 channels] bytes, 16 KiB, 16-byte chunk c of row r stored at chunk c ^ ((r & 7) << 1); fragments are read with
// ds_read_b64_tr_b8 (tools/tr_b8_probe.hip: per 16 lanes a block of 8 rows x 16 columns of bytes, lane 2q+p supplies row q columns
// 8p.., lane i receives column i) -- lane half h takes pixels 32h..32h+31 of the tile in 4 reads, for both operands alike, which is
// all a dot product needs; a 32-lane half touches 8 rows x 32 contiguous bytes whose chunk pairs the XOR spreads over all 64 banks.
// Loop: a ring of 4 LDS buffers filled by LDS-DMA two tiles ahead (counted vmcnt), ONE barrier per tile, fragment reads issued between
// the MFMAs one half tile ahead (register plan below).  The same loop on bf16 operands (32-pixel tiles, ds_read_b64_tr_b16) was built and
// measured against k_wgrad256's ping-pong phases: 19.3-20.1 vs 17.7 ms per step for the same launches -- the bf16 kernel keeps its phases.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_wgrad256_f8(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[4 * 2 * 64 * 16];   // byte = ring<<15 | ab<<14 | row*256 + slot*16
  const int t = threadIdx.x, lane = t & 63;
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wvu >> 2, wk = wvu & 3;
  const int ntn = p.Cout >> 8, ntk = p.K >> 8;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = bid % ntk; bid /= ntk;
  const int tile_n = bid % ntn; bid /= ntn;
  const int n0 = tile_n * 256, k0 = tile_k * 256;
  const int mt0 = bid * p.mtiles_per_split;
  const int nmt_real = min(p.mtiles_per_split, (p.M + WM - 1) / WM - mt0);
  const int nmt = (nmt_real + 1) & ~1;                        // an even number of tiles (one straight-line loop body of two): the extra one is all zeros
  const int mlim = min(p.M, (mt0 + nmt_real) * WM);           // rows at or past this are not this block's
  const bool taps = !(p.KH == 1 && p.KW == 1);

  // ---- staging: wave instruction i of a thread fills rows (i*8 + wave)*4 .. +3 of an image, lane -> (row lane>>4, slot lane&15)
  const int rq = lane >> 4, r8 = (wvu & 1) * 4 + rq;
  const int cl = (lane & 15) ^ (r8 << 1);                      // logical chunk of this lane's slot
  const int pp = k0 / p.Cin, coff = (k0 - pp * p.Cin) >> 4;    // the tile's filter tap and first chunk within the pixel (256 | Cin)
  const int ky = pp / p.KW, kx = pp - ky * p.KW;
  unsigned vd[2], vx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + wvu * 4 + rq;
    vd[i] = (unsigned)(row * p.ldd + cl * 16);
    vx[i] = (unsigned)(((row + ky * p.Wi + kx) * p.xrs + coff + cl) * 16);
  }
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + ((long)mt0 * WM * p.ldd + n0)), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + ((long)mt0 * WM - p.pad * p.Wi - p.pad) * p.xrs * 16), 0, 0x80000000u, 0x00020000);
  const unsigned dstep = (unsigned)(WM * p.ldd), xstep = (unsigned)(WM * p.xrs * 16);
  const int mrow = mt0 * WM + wvu * 4 + rq;                    // + T*64 + i*32
  char* const L = (char*)lds;
  auto stage = [&](int T) {
    char* dst = L + ((T & 3) << 15) + wvu * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mrow + T * WM + i * 32;
      bool ok = m < mlim;
      const unsigned dinv = ok ? 0u : 0x80000000u;
      if (taps) {
        const unsigned mm = min((unsigned)m, (unsigned)(p.M - 1));
        const unsigned tq = fdiv(mm, p.dWo);
        const unsigned ox = mm - tq * p.Wo;
        const unsigned oy = tq - fdiv(tq, p.dHo) * p.Ho;
        ok = ok & ((unsigned)((int)oy - p.pad + ky) < (unsigned)p.Hi) & ((unsigned)((int)ox - p.pad + kx) < (unsigned)p.Wi);
      }
      const unsigned xinv = ok ? 0u : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(dst + i * 8192), 16,
                                               (int)(vd[i] | dinv), (int)(T * dstep), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + (1 << 14) + i * 8192), 16,
                                               (int)(vx[i] | xinv), (int)(T * xstep), 0, 0);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- transposed fragment reads: per-lane byte addresses of read 0 in ring buffer 0 (+ 2048 per read: immediate)
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)L;
  unsigned adA[4], adB[2];
  {
    const int q = (lane & 15) >> 1, gb = (lane >> 4) & 1;
    const unsigned rowb = (unsigned)((32 * (lane >> 5) + q) * 256 + 8 * (lane & 1));
#pragma unroll
    for (int a = 0; a < 4; ++a) adA[a] = lbase + rowb + (unsigned)((((wn * 8 + a * 2 + gb) ^ (q << 1)) & 15) << 4);
#pragma unroll
    for (int b = 0; b < 2; ++b) adB[b] = lbase + (1u << 14) + rowb + (unsigned)((((wk * 4 + b * 2 + gb) ^ (q << 1)) & 15) << 4);
  }
  // Register plan (accumulators 128): dY fragments ONE set of 4 (32 registers), X fragments two sets of 2 (32).  A tile's 8 MFMAs run as
  // two groups: G0 = dY fragments 0,1 (while fragments 2,3 of the same tile are read), G1 = fragments 2,3 (while fragments 0,1 and the
  // X fragments of the NEXT tile are read into the registers G0 has released / the other X set).
  u32x2 fa[4][4], fb[2][2][4];                                 // [fragment][read], [register set][fragment][read]
#define CDDMSL_TR8(DST, AD, R) asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(DST) : "v"(AD), "i"((R) * 2048));
#define CDDMSL_TR8x4(F, AD) CDDMSL_TR8(F[0], AD, 0) CDDMSL_TR8(F[1], AD, 1) CDDMSL_TR8(F[2], AD, 2) CDDMSL_TR8(F[3], AD, 3)
// the waits name the registers the retired reads wrote: no MFMA that consumes them (and no copy of them) can be placed above
#define CDDMSL_F8_WAIT_TOP(S)                                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]),                          \
               "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]),                                                   \
               "+v"(fb[S][0][0]), "+v"(fb[S][0][1]), "+v"(fb[S][0][2]), "+v"(fb[S][0][3]),                                       \
               "+v"(fb[S][1][0]), "+v"(fb[S][1][1]), "+v"(fb[S][1][2]), "+v"(fb[S][1][3]) :: "memory");
#define CDDMSL_F8_WAIT_MID()                                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[2][2]), "+v"(fa[2][3]),                          \
               "+v"(fa[3][0]), "+v"(fa[3][1]), "+v"(fa[3][2]), "+v"(fa[3][3]) :: "memory");
  auto mma = [&](auto SC, auto AC, auto BC) {
    constexpr int S = decltype(SC)::value, a = decltype(AC)::value, b = decltype(BC)::value;
    const i32x8 va = {(int)fa[a][0][0], (int)fa[a][0][1], (int)fa[a][1][0], (int)fa[a][1][1],
                      (int)fa[a][2][0], (int)fa[a][2][1], (int)fa[a][3][0], (int)fa[a][3][1]};
    const i32x8 vb = {(int)fb[S][b][0][0], (int)fb[S][b][0][1], (int)fb[S][b][1][0], (int)fb[S][b][1][1],
                      (int)fb[S][b][2][0], (int)fb[S][b][2][1], (int)fb[S][b][3][0], (int)fb[S][b][3][1]};
    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[a][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    // an MFMA is a register-only instruction: free to sink below later reads, waits and barriers (it did: all 8 of a tile ended up
    // behind the NEXT tile's barrier).  The empty volatile statement names its result, which orders it among the volatile reads / waits.
    asm volatile("" : "+v"(acc[a][b]));
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  auto body = [&](int kt, auto SC) {
    constexpr int S = decltype(SC)::value, N = S ^ 1;
    using IS = std::integral_constant<int, S>;
    if (kt + 1 < nmt) {                                        // this thread's part of tile kt+1 has landed (tile kt+2 may be in flight)
      if (kt + 2 < nmt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                              // ... and everyone's; every wave is past its reads of tile kt-1
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 3 < nmt) stage(kt + 3);                           // into the buffer tile kt-1 occupied
    CDDMSL_F8_WAIT_TOP(S)
    __builtin_amdgcn_sched_barrier(0);
    {                                                          // G0, reading dY fragments 2, 3 of this tile
      const unsigned ro = (unsigned)(kt & 3) << 15;
      const unsigned a2 = adA[2] + ro, a3 = adA[3] + ro;
      CDDMSL_TR8x4(fa[2], a2)
      mma(IS{}, I0{}, I0{}); mma(IS{}, I0{}, I1{});
      CDDMSL_TR8x4(fa[3], a3)
      mma(IS{}, I1{}, I0{}); mma(IS{}, I1{}, I1{});
    }
    CDDMSL_F8_WAIT_MID()
    __builtin_amdgcn_sched_barrier(0);
    {                                                          // G1, reading the next tile's dY fragments 0, 1 and X fragments
      // (behind the last tile: a buffer of the ring that holds an older tile -- read and never used)
      const unsigned ro = (unsigned)((kt + 1) & 3) << 15;
      const unsigned a0 = adA[0] + ro, a1 = adA[1] + ro, b0 = adB[0] + ro, b1 = adB[1] + ro;
      // (one fragment per MFMA: issuing all four up front measured slower, 5.14 vs 4.98 ms for the RoI head's three launches)
      CDDMSL_TR8x4(fa[0], a0)
      mma(IS{}, I2{}, I0{});
      CDDMSL_TR8x4(fa[1], a1)
      mma(IS{}, I2{}, I1{});
      CDDMSL_TR8x4(fb[N][0], b0)
      mma(IS{}, I3{}, I0{});
      CDDMSL_TR8x4(fb[N][1], b1)
      mma(IS{}, I3{}, I1{});
    }
  };
  if (nmt > 0) {
    stage(0);
    if (nmt > 1) stage(1);
    if (nmt > 2) stage(2);
    if (nmt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nmt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CDDMSL_TR8x4(fa[0], adA[0]) CDDMSL_TR8x4(fa[1], adA[1]) CDDMSL_TR8x4(fb[0][0], adB[0]) CDDMSL_TR8x4(fb[0][1], adB[1])
    for (int kt = 0; kt < nmt; kt += 2) {
      body(kt, I0{});
      body(kt + 1, I1{});
    }
  }
#undef CDDMSL_TR8x4
#undef CDDMSL_F8_WAIT_TOP
#undef CDDMSL_F8_WAIT_MID
#undef CDDMSL_TR8

  const int r = lane & 31, h = lane >> 5;
  if (p.ws) {          // split reduction through the workspace, accumulators in fragment order (k_wgrad256's layout: k_wgrad_reduce<8, 32>)
    f32x4* dst = (f32x4*)p.ws + ((long)(bid * ntn + tile_n) * ntk + tile_k) * (8 * 32 * 64) + (wvu * 32) * 64 + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = {acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
          dst[((a * 2 + b) * 4 + g4) * 64] = v;
        }
  } else
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int k = k0 + wk * 64 + b * 32 + r;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int n = n0 + wn * 128 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        atomicAdd(p.dw + (long)n * p.ldo + k, acc[a][b][g] * (p.scale ? p.scale[n] : 1.f));
      }
    }
}

template <typename T>
__device__ __forceinline__ void weight_prep_body(const float* w, const float* scale, char* wf, char* wd, int Cout, int KH, int KW, int Cin,
                                                 long first, long stride) {
  __shared__ float tile[32][33];
  const int taps = KH * KW, nit = (Cin + 31) / 32, nct = (Cout + 31) / 32;
  const long ntiles = (long)nct * taps * nit;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 256 threads: 8 rows x 32 columns per sweep
  constexpr int ES = Mma<T>::ES;
  for (long tl = first; tl < ntiles; tl += stride) {
    const int it = (int)(tl % nit); const long q = tl / nit;
    const int tap = (int)(q % taps), ct = (int)(q / taps);
    const int ky = tap / KW, kx = tap % KW;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = ct * 32 + ty + 8 * r, ci = it * 32 + tx;
      float v = 0.f;
      if (co < Cout && ci < Cin) {
        const long i = ((long)co * taps + tap) * Cin + ci;
        v = w[i];
        if (wf) Mma<T>::store(wf + i * ES, v);
        if (scale) v *= scale[co];
      }
      tile[ty + 8 * r][tx] = v;
    }
    __syncthreads();
    if (wd) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = it * 32 + ty + 8 * r, co = ct * 32 + tx;
        if (ci < Cin && co < Cout) {
          const long j = (((long)ci * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)) * Cout + co;
          Mma<T>::store(wd + j * ES, tile[tx][ty + 8 * r]);
        }
      }
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_weight_prep(const float* w, const float* scale, char* wf, char* wd, int Cout, int KH, int KW, int Cin) {
  weight_prep_body<T>(w, scale, wf, wd, Cout, KH, KW, Cin, (long)blockIdx.x, (long)gridDim.x);
}
// every trainable weight of the step in ONE launch: blockIdx.y = table row {w, scale, wf, wd, Cout, KH, KW, Cin} (8 x int64)
template <typename T>
__global__ __launch_bounds__(256) void k_weight_prep_multi(const long long* table) {
  const long long* e = table + 8 * (long)blockIdx.y;
  weight_prep_body<T>((const float*)e[0], (const float*)e[1], (char*)e[2], (char*)e[3], (int)e[4], (int)e[5], (int)e[6], (int)e[7],
                      (long)blockIdx.x, (long)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// 3x3 convolutions with FEW input channels (the CLIP stem: 3(->8 padded)->32 stride 2, 32->32, 32->64; one or four
// 16-byte chunks per pixel), pad 1, FrozenBN + ReLU epilogue.  These layers are pure streaming work: K = 72 or 288,
// N = 32 or 64, M = millions of pixels.  On the tile kernel a block ran 2-5 K-tiles behind one DMA latency each and
// reached ~1 TB/s; here NOTHING goes through LDS: the whole weight matrix lives in registers as MFMA B fragments
// (<= 18 k-steps x NT tiles), each wave walks 32-pixel tiles with a grid stride, and a lane's A fragment of a k-step is
// ONE 16-byte global load (8 consecutive channels of one filter tap of its pixel; the 9 taps of neighbouring pixels hit
// L1/L2).  All loads of a tile are issued before its MFMAs.  Output: lanes = consecutive channels (64 contiguous bytes
// per pixel and tile).  Same accumulation order as k_conv_fwd (chunk pairs in K order) -> bit-identical results.
// ------------------------------------------------------------------------------------------------
template <typename T, int CPP, int NT, int NTAP = 9>     // NTAP = 1: the same streaming structure for a 1x1 layer with 32 / 64 output channels
__global__ __launch_bounds__(256) void k_conv3x3_small(ConvArgs p) {
  constexpr int KC = NTAP * CPP, KS = (KC + 1) / 2;    // 16-byte chunks of a weight row; k-steps of two chunks
  constexpr int PAD = NTAP == 9 ? 1 : 0;
  constexpr int ES = Mma<T>::ES;
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  // weights as MFMA B fragments: fragment (nt, ks) of a lane = chunk 2ks+hh of weight row nt*32 + r (zero past the row end).
  // One chunk per pixel (K = 72): 5 k-steps, kept in registers.  Four chunks per pixel (K = 288): 18 k-steps x NT tiles would
  // take up to 144 VGPRs and leave one wave per SIMD -- they sit in LDS in fragment order (lane-linear 16-byte reads).
  constexpr bool WLDS = CPP > 1;
  __shared__ __attribute__((aligned(16))) u32x4 wl[WLDS ? NT * KS * 64 : 1];
  u32x4 bw[WLDS ? 1 : NT][WLDS ? 1 : KS];
  {
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (WLDS && ((nt * KS + ks) & 3) != wv) continue;       // the four waves fill the image cooperatively
        const int q = 2 * ks + hh;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q < KC) v = *(const u32x4*)(p.w + ((long)(nt * 32 + r) * KC + q) * 16);
        if (WLDS) wl[(nt * KS + ks) * 64 + lane] = v;
        else bw[WLDS ? 0 : nt][WLDS ? 0 : ks] = v;
      }
    if (WLDS) __syncthreads();
  }
  float sc[NT], bi[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { sc[nt] = p.scale ? p.scale[nt * 32 + r] : 1.f; bi[nt] = p.bias ? p.bias[nt * 32 + r] : 0.f; }
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ntiles = (p.M + 31) >> 5;
  // Addressing.  An ablation of this kernel (round 2: loads, MFMAs and stores switched off one at a time) showed 40-50 % of its time
  // to be the per-tile INDEX ARITHMETIC alone -- per k-step a tap select, two range tests and a 64-bit address, per output element a
  // 64-bit address and a row test: ~530 vector-ALU instructions per 32-pixel tile against 36-72 MFMAs.  Now: buffer addressing
  // (32-bit lane offset from the tensor base; the k-step's tap / chunk offset is wave-uniform and rides in soffset), tap validity as
  // a 9-bit mask per tile whose bit sets bit 31 of the lane offset (out of range -> zeros), and buffer stores with the row offset in
  // soffset; rows past M fall outside num_records.  (One chunk per pixel, CPP = 1: a k-step's two chunks straddle taps -- lane halves
  // differ by more than a constant -- and that instantiation keeps its direct loads.)
  // (base one row and one pixel BEFORE the tensor: the lane offset of a pixel's tap (0,0) is then never negative -- the range check
  // sees the lane offset alone -- and the bytes in front of the tensor are only ever addressed by taps the mask removes)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - (long)(p.Wi + 1) * (CPP * 16)), 0, 0x7fffffff, 0x00020000);
  const long ybytes = (long)p.M * p.ldy * ES, mbytes = (long)p.M * p.ldm * ES;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)(ybytes > 0x7fffffffL ? 0x7fffffffL : ybytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.relu_mask, 0, p.relu_mask ? (int)(mbytes > 0x7fffffffL ? 0x7fffffffL : mbytes) : 0, 0x00020000);
  const float relu_floor = p.relu ? 0.f : -__builtin_inff();
  __shared__ __attribute__((aligned(16))) u32x4 ost[ES == 2 ? 4 * 32 * NT * 4 : 1];    // per wave: one output tile, 32 rows x NT * 64 bytes
  const bool ost_ok = !p.relu_mask && p.ldy == NT * 32 && (long)p.M * NT * 64 < 0x7fffffffL;
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int m = tile * 32 + r;
    const bool vm = m < p.M;
    const unsigned mm = vm ? m : 0;
    const unsigned tq = fdiv(mm, p.dWo), ox = mm - tq * p.Wo;
    const unsigned img = fdiv(tq, p.dHo), oy = tq - img * p.Ho;
    const int iy0 = (int)oy * p.stride - PAD, ix0 = (int)ox * p.stride - PAD;
    // byte offset of tap (0,0), chunk 0 of this lane's pixel (may be "negative": wraps, and is then masked by the tap test)
    const unsigned lbase = (unsigned)((((int)img * p.Hi + iy0 + 1) * p.Wi + ix0 + 1) * (CPP * 16)) + (CPP > 1 ? hh * 16 : 0);
    unsigned bad = vm ? 0u : 0x1ffu;                // bit (3 ky + kx): that tap of this pixel is outside the image
    if (NTAP == 9) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          if ((unsigned)(iy0 + ky) >= (unsigned)p.Hi || (unsigned)(ix0 + kx) >= (unsigned)p.Wi) bad |= 1u << (3 * ky + kx);
    }
    u32x4 a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int q0 = 2 * ks, q1 = 2 * ks + 1;
      const int t0 = q0 / CPP, c0 = q0 % CPP, t1 = q1 / CPP, c1 = q1 % CPP;
      if (CPP > 1) {                                // both chunks in tap t0, the upper half-wave one chunk further (in lbase)
        const unsigned v = lbase | (__builtin_amdgcn_ubfe(bad, (unsigned)t0, 1u) << 31);
        a[ks] = __builtin_amdgcn_raw_buffer_load_b128(rx, v, ((t0 / 3) * p.Wi + (t0 % 3)) * (CPP * 16) + c0 * 16, 0);
      } else {                                      // (direct loads: measured faster than the buffer form here, 120 vs 144 us on the first stem layer)
        const int ky = hh ? t1 / 3 : t0 / 3, kx = hh ? t1 % 3 : t0 % 3, cc = hh ? c1 : c0;
        const bool inq = hh ? (q1 < KC) : (q0 < KC);
        const int iy = iy0 + ky, ix = ix0 + kx;
        const bool ok = vm && inq && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *(const u32x4*)(p.x + ((((long)img * p.Hi + iy0) * p.Wi + ix0) + (long)ky * p.Wi + kx) * (CPP * 16) + cc * 16);
        a[ks] = v;
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;
    int wlane = lane;
    asm volatile("" : "+v"(wlane));            // opaque per tile: keeps the fragment reads in the loop (hoisted, they are 144 VGPRs again)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (WLDS) Mma<T>::step(acc[nt], a[ks], wl[(nt * KS + ks) * 64 + wlane]);
        else Mma<T>::step(acc[nt], a[ks], bw[WLDS ? 0 : nt][WLDS ? 0 : ks]);
      }
    // bf16 rows of exactly the tile's NT * 32 channels, no mask (every forward launch of the stem / layer1): the tile is 32 * NT * 64
    // contiguous bytes of y.  It goes through this wave's LDS slot (2-byte writes in the accumulator layout, 16-byte reads in memory
    // order; one wave's DS operations execute in order, no barrier) and leaves as 2 * NT wave-wide 1 KiB stores instead of 16 * NT
    // stores of 2 bytes per lane (two 64-byte pieces per instruction).
    if (ES == 2 && ost_ok) {
      char* ob = (char*)ost + (threadIdx.x >> 6) * (32 * NT * 64);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int row = (g & 3) + 8 * (g >> 2) + 4 * hh;
          float v = affine<T>(acc[nt][g], sc[nt], bi[nt]);
          asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "s"(relu_floor));
          *(unsigned short*)(ob + row * (NT * 64) + (nt * 32 + r) * 2) = f2bf(v);
        }
#pragma unroll
      for (int i = 0; i < 2 * NT; ++i) {
        const u32x4 o = *(const u32x4*)(ob + (i * 64 + lane) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, (unsigned)((i * 64 + lane) * 16), tile * (32 * NT * 64), 0);   // (rows past M: outside num_records)
      }
      continue;
    }
    // element (row (g&3) + 8(g>>2) + 4hh of the tile, channel 32 nt + r): lane offset once, the row in soffset
    const unsigned vy = (unsigned)(((tile * 32 + 4 * hh) * p.ldy + r) * ES), vmk = (unsigned)(((tile * 32 + 4 * hh) * p.ldm + r) * ES);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = (g & 3) + 8 * (g >> 2);
        float v = affine<T>(acc[nt][g], sc[nt], bi[nt]);
        asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "s"(relu_floor));
        if (p.relu_mask) {
          float mv;
          if (ES == 2) mv = bf2f((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rm, vmk, (row * p.ldm + nt * 32) * ES, 0));
          else mv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, vmk, (row * p.ldm + nt * 32) * ES, 0));
          if (!(mv > 0.f)) v = 0.f;
        }
        if (ES == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)f2bf(v), ry, vy, (row * p.ldy + nt * 32) * ES, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vy, (row * p.ldy + nt * 32) * ES, 0);
      }
  }
}

static thread_local int g_last_kernel = 0;   // which kernel the last conv/GEMM entry point of this thread launched (cddmsl_last_kernel)
static thread_local int g_plan_only = 0;     // cddmsl_plan_only(1): entry points choose their kernel (g_last_kernel) and return without launching
static thread_local int g_batch = 1;   // set by the batched entry point around its launch
static inline int g_batch_peek() { return g_batch; }

// ------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4; 128x64 per wave), two wave groups ping-ponging on each SIMD.
//
// The 128x128 kernel above tops out near 1 PFLOP/s: both of its blocks on a CU stall at the same two barriers per
// K-tile.  Here each SIMD hosts one wave of group 0 (wr = 0) and one of group 1 (wr = 1), group 1 running ONE barrier
// behind: between two consecutive barriers one group issues LDS reads + LDS-DMA for its next quadrant while the other
// runs that quadrant's MFMAs, so the matrix pipe always has a wave feeding it.  A K-tile (8 chunks) is four phases,
// one 64x32 quadrant of the wave's 128x64 output each:
//     phase 1  read B0            MFMA q00   stage A1 of the OTHER buffer with tile kt+1 (read in the previous phase 3)
//     phase 2  read B1            MFMA q01   stage A0 of this buffer with tile kt+2   (read in the previous phase 4)
//     phase 3  read A1            MFMA q11   stage B0 (read in phase 1), s_waitcnt vmcnt(4)
//     phase 4  read A0 of kt+1    MFMA q10   stage B1 (read in phase 2)
// Each half-tile (128 rows x 128 B: sub-tile i of both row groups / sub-tile j of all four column groups) is restaged
// TWO phases after its last read: a phase's reads are retired (lgkmcnt(0), placed after the barrier so the LDS latency
// overlaps the wait for the other group's MFMAs) before its MFMAs, i.e. before the barrier that opens the next phase,
// which every wave passes before the phase after that issues its DMA -- for both groups despite the one-barrier stagger.
// The phase-3 wait leaves the two youngest half-tiles (4 DMAs per thread) in flight and retires every older one: all
// of tile kt+1, whose first read (A0, phase 4) comes after that phase's barrier.
// LDS: [2 buffers][A|B][2 halves][128 rows x 8 chunks] = 128 KiB, swizzled like the 128x128 kernel.
// Sources are buffer-addressed (buffer_load_dwordx4 ... offen lds): per-block base in SGPRs, a per-lane byte offset that
// is constant over the K loop, and the running K / filter-tap position in the wave-uniform soffset -- the loop carries
// no per-lane address arithmetic.  Filter-tap validity is a per-row bit mask (KH*KW <= 31 bits): an out-of-image tap or
// an out-of-range row sets bit 31 of the lane's offset, which the range check turns into zeros written to LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, (int)voff, (int)soff, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Epilogue of a wave's 128 x 64 accumulator tile (acc[4][2] of 32x32 blocks; rows m0 + 128 wr ..., columns n0 + 64 wc ...), shared by
// the 256x256 kernel and the 256x128 two-workgroup kernel.
// Per-wave LDS transpose (32 rows x 64 cols f32 per pass; DS ops of one wave execute in order, so no barrier
// is needed); a lane then owns 8 consecutive columns of a row: 16-byte residual / mask loads and y stores.  All of them are
// buffer-addressed -- rows past M fall outside num_records (loads give 0, stores are dropped), an absent residual / mask is
// a zero-sized buffer, per-lane offsets are computed once and the row / pass position is the wave-uniform soffset -- so the
// passes are straight-line code without per-row exec masks, zero fills or 64-bit address arithmetic.
// The transposition is double-buffered in the wave's private 16 KiB of the (now idle) operand ring: pass a+1's 32 scratch
// writes are issued right behind pass a's 8 scratch reads (all four rows at once), so the write drain and the read latency
// are each paid once per pass and overlap the arithmetic and stores of the pass before -- tools/tile_stamps.py measured the
// former row-at-a-time form (read two chunks, wait, compute, store, scheduling barrier) at 7.4 us per tile without and
// 11 us with a residual, against ~1 us of vector-ALU work.
// `ep`: 16 KiB of LDS private to the wave.
// ------------------------------------------------------------------------------------------------
// PRE: the rows of the first two passes of ONE operand (the residual if there is one, else the ReLU mask) were requested by the caller
// -- the 256x256 kernel issues them in phase 3 of the tile's last K-tile, into the registers the A0 fragments no longer need -- and
// arrive in `pre`.
// `sb` (optional): this lane's 8 scale and 8 bias values, requested by the caller ahead of time (phase 4 of the last K-tile).
template <typename T, bool RPOOL, int EPI, bool PRE = false>
__device__ __forceinline__ void tile_epilogue(const ConvArgs& p, f32x16 (&acc)[4][2], float* ep, int wr, int wc, int lane, int m0, int n0,
                                              const u32x4 (*pre)[4] = nullptr, const f32x4* sb = nullptr) {
  constexpr int ES = Mma<T>::ES;
  const int r32 = lane & 31, hh = lane >> 5;
  const int cg = lane & 7, rr = lane >> 3;
  const int n = n0 + wc * 64 + cg * 8;
  float sc[8], bi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (sb) { sc[j] = sb[j >> 2][j & 3]; bi[j] = sb[2 + (j >> 2)][j & 3]; }
    else { sc[j] = p.scale ? p.scale[n + j] : 1.f; bi[j] = p.bias ? p.bias[n + j] : 0.f; }
  }
  const bool has_res = EPI < 0 ? p.residual != nullptr : (EPI & 1) != 0;
  const bool has_msk = EPI < 0 ? p.relu_mask != nullptr : (EPI & 2) != 0;
  const bool f32out = EPI < 0 && (ES == 4 || p.out_f32 != 0);
  const int eso = f32out ? 4 : 2;
  const float relu_floor = p.relu ? 0.f : -__builtin_inff();
  const long rows = p.M - m0;
  auto mk = [&](const char* base, long ld, int es) {
    long bytes = rows * ld * es;
    if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)m0 * ld * es), 0, base ? (int)bytes : 0, 0x00020000);
  };
  const bool rf32 = EPI < 0 && ES == 2 && p.res_f32;         // f32 residual rows on the bf16 kernel (the mapper's f32 residual stream)
  const int esr = rf32 ? 4 : ES;
  // (pooled residual: addressed from the tensor base -- the pooled pixel of a row is not linear in the row)
  const __amdgpu_buffer_rsrc_t ry = mk(p.y, p.ldy, eso), rmsk = mk(p.relu_mask, p.ldm, ES);
  // RPOOL is a template parameter, not a run-time branch: with the pooled path compiled into the one kernel every launch
  // ran 5 % slower (more uniform branches per epilogue row), although three launches per step use it.
  const __amdgpu_buffer_rsrc_t rres = RPOOL ? __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, 0x7fffffff, 0x00020000)
                                            : mk(p.residual, p.ldr, esr);
  auto pooled_off = [&](int m) -> unsigned {      // byte offset of this lane's 8 columns in the pooled row of output pixel m
    const unsigned tq = fdiv((unsigned)m, p.dWo), ox = m - tq * p.Wo;
    const unsigned img = fdiv(tq, p.dHo), oy = tq - img * p.Ho;
    const unsigned hp = p.Ho >> 1, wp = p.Wo >> 1;
    const bool in = m < p.M && (oy >> 1) < hp && (ox >> 1) < wp;      // an odd size's last row / column has no pooled pixel
    return in ? (unsigned)((((img * hp + (oy >> 1)) * wp + (ox >> 1)) * (unsigned)p.ldr + (unsigned)n) * ES) : 0x80000000u;
  };
  const unsigned vy = (unsigned)(((wr * 128 + rr) * p.ldy + n) * eso);
  const unsigned vr = (unsigned)(((wr * 128 + rr) * p.ldr + n) * esr), vm = (unsigned)(((wr * 128 + rr) * p.ldm + n) * ES);
  const bool emit8 = EPI < 0 && ES == 2 && p.y8 != nullptr;
  const __amdgpu_buffer_rsrc_t ry8 = mk(p.y8, p.ldy, 1);
  const unsigned vy8 = (unsigned)((wr * 128 + rr) * p.ldy + n);
  const float q8s = emit8 && p.q8 ? p.q8[0] : 1.f;
  unsigned am8 = 0u;                                 // max |y| as a bit pattern (common.h absmax_bits): Inf / NaN are recorded, not dropped
  // bf16: residual / mask rows are fetched TWO passes ahead (two register sets, static indices): with one block per CU
  // nothing else hides their HBM latency.  The f32 parity instantiation (twice the registers per row) one pass ahead.
  constexpr int DEPTH = ES == 2 ? 2 : 1;
  // transposition writes: the swizzled chunk (col>>3) ^ (row&7) splits into a lane part ((r32>>3) ^ (hh<<2)) XOR a
  // compile-time part ((b<<2) ^ (g&3)): eight per-lane base addresses, the row of a register is an immediate offset
  char* wbase[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    wbase[c] = (char*)ep + (hh * 4 * 64 + ((((r32 >> 3) ^ (hh << 2)) ^ c) << 3) + (r32 & 7)) * 4;
  u32x4 rresb[DEPTH][4][ES / 2], rmskb[DEPTH][4][ES / 2];
  auto fetch = [&](int a, u32x4 (*rres_)[ES / 2], u32x4 (*rmsk_)[ES / 2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < ES / 2; ++q) {       // (an absent operand is not requested at all: even a zero-sized buffer returns its zeros through the vector memory path)
        if (RPOOL) rres_[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rres, pooled_off(m0 + wr * 128 + a * 32 + rr + 8 * i), q * 16, 0);
        else if (has_res && !rf32) rres_[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rres, vr, (a * 32 + 8 * i) * p.ldr * ES + q * 16, CDDMSL_LOAD_AUX);
        if (has_msk) rmsk_[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rmsk, vm, (a * 32 + 8 * i) * p.ldm * ES + q * 16, CDDMSL_LOAD_AUX);
      }
  };
  if (PRE && DEPTH == 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (EPI & 1) {
          rresb[d % DEPTH][i][0] = pre[d][i];
          if (EPI & 2) rmskb[d % DEPTH][i][0] = __builtin_amdgcn_raw_buffer_load_b128(rmsk, vm, (d * 32 + 8 * i) * p.ldm * ES, CDDMSL_LOAD_AUX);
        } else rmskb[d % DEPTH][i][0] = pre[d][i];
      }
  } else if (DEPTH == 2 && !rf32) { fetch(0, rresb[0], rmskb[0]); fetch(1, rresb[DEPTH - 1], rmskb[DEPTH - 1]); }
  auto put = [&](auto A) {                          // accumulator rows 32a..32a+31 -> transposition buffer a & 1
    constexpr int a = decltype(A)::value;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 16; ++g)                  // element (row rg + 4hh, col 32b + r32) -> ep[row*64 + ((col>>3 ^ row&7) << 3 | col&7)]
        *(float*)(wbase[(b << 2) ^ (g & 3)] + ((g & 3) + 8 * (g >> 2)) * 256 + (a & 1) * 8192) = acc[a][b][g];
  };
  auto pass = [&](auto A) {
    constexpr int a = decltype(A)::value;
    if (DEPTH == 1) fetch(a, rresb[0], rmskb[0]);
    if (rf32) {                                   // this pass's 4 rows x 32 B, in the two bf16 register sets taken together
#pragma unroll
      for (int f = 0; f < 8; ++f)
        rresb[(f >> 2) % DEPTH][f & 3][0] = __builtin_amdgcn_raw_buffer_load_b128(rres, vr, (a * 32 + 8 * (f >> 1)) * p.ldr * 4 + (f & 1) * 16, 0);
    }
    u32x4 (*rres_)[ES / 2] = rresb[a % DEPTH];
    u32x4 (*rmsk_)[ES / 2] = rmskb[a % DEPTH];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this pass's scratch writes have landed
    f32x4 val[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rr + 8 * i;
      const f32x4* src = (const f32x4*)(ep + (a & 1) * 2048 + row * 64 + ((cg ^ (row & 7)) << 3));
      val[i][0] = src[0]; val[i][1] = src[1];
    }
    // (compiler fence: the next pass's float stores must stay behind these f32x4 loads -- type-based alias analysis treats them
    // as unrelated; they go to the OTHER buffer, but a hoisted store of pass a+2 would not)
    asm volatile("" ::: "memory");
    if constexpr (a + 1 < 4) put(std::integral_constant<int, a + 1>{});
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v0 = val[i][0], v1 = val[i][1];
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = affine<T>(v[j], sc[j], bi[j]);
      if (rf32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] += __builtin_bit_cast(f32x4, rresb[((2 * i) >> 2) % DEPTH][(2 * i) & 3][0])[j];
          v[4 + j] += __builtin_bit_cast(f32x4, rresb[((2 * i + 1) >> 2) % DEPTH][(2 * i + 1) & 3][0])[j];
        }
      } else if (RPOOL) {                         // (x 0.25 is exact: the same value avgpool2_bwd would have stored)
        if (ES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[2 * j] += 0.25f * bf2f(rres_[i][0][j] & 0xffff); v[2 * j + 1] += 0.25f * bf2f(rres_[i][0][j] >> 16); }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] += 0.25f * __builtin_bit_cast(f32x4, rres_[i][0])[j]; v[4 + j] += 0.25f * __builtin_bit_cast(f32x4, rres_[i][ES / 2 - 1])[j];
          }
        }
      } else if (has_res) {
        if (ES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[2 * j] += bf2f(rres_[i][0][j] & 0xffff); v[2 * j + 1] += bf2f(rres_[i][0][j] >> 16); }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] += __builtin_bit_cast(f32x4, rres_[i][0])[j]; v[4 + j] += __builtin_bit_cast(f32x4, rres_[i][ES / 2 - 1])[j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) asm("v_max_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "s"(relu_floor));   // (fmaxf adds a canonicalising op per element)
      if (has_msk) {
        float mv[8];
        if (ES == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { mv[2 * j] = bf2f(rmsk_[i][0][j] & 0xffff); mv[2 * j + 1] = bf2f(rmsk_[i][0][j] >> 16); }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mv[j] = __builtin_bit_cast(f32x4, rmsk_[i][0])[j]; mv[4 + j] = __builtin_bit_cast(f32x4, rmsk_[i][ES / 2 - 1])[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (!(mv[j] > 0.f)) v[j] = 0.f;
      }
      const unsigned so = (unsigned)((a * 32 + 8 * i) * p.ldy * eso);
      // Store-data hazard (observed, gfx950): hipcc may put a vector-ALU write to the FIRST data register of a
      // buffer_store_dwordx4 ... soffset offen directly behind the store (it did: v_mul_hi_u32 of the next row's pooled-pixel
      // division), and lanes 12-15 of every 16 then stored that instruction's result instead of the output.  Nothing may
      // WRITE the data registers for a few cycles: a wait behind every store, and a use of the data behind the wait, which
      // keeps the registers allocated until then (the rows of a pass are otherwise free to interleave).
      if (emit8) {                                  // the e4m3 copy for the consuming convolution (fp8 configuration)
#pragma unroll
        for (int j = 0; j < 8; ++j) am8 = absmax_bits(am8, v[j]);
        const u32x2 o8 = {pack4_e4m3(v[0] * q8s, v[1] * q8s, v[2] * q8s, v[3] * q8s), pack4_e4m3(v[4] * q8s, v[5] * q8s, v[6] * q8s, v[7] * q8s)};
        __builtin_amdgcn_raw_buffer_store_b64(o8, ry8, vy8, (unsigned)((a * 32 + 8 * i) * p.ldy), CDDMSL_STORE_AUX);
        asm volatile("s_nop 4" ::: "memory");
        asm volatile("" :: "v"(o8));
      }
      if (f32out) {
        const u32x4 o0 = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
        const u32x4 o1 = {__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(o0, ry, vy, so, CDDMSL_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(o1, ry, vy, so + 16, CDDMSL_STORE_AUX);
        asm volatile("s_nop 4" ::: "memory");
        asm volatile("" :: "v"(o0), "v"(o1));
      } else {
        const u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(o, ry, vy, so, CDDMSL_STORE_AUX);
        else __builtin_amdgcn_raw_buffer_store_b128(o, ry, vy, so, 0);
        asm volatile("s_nop 4" ::: "memory");
        asm volatile("" :: "v"(o));
      }
    }
    if (DEPTH == 2 && a + 2 < 4 && !rf32) fetch(a + 2, rresb[a % DEPTH], rmskb[a % DEPTH]);
  };
  put(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
  pass(std::integral_constant<int, 2>{});
  pass(std::integral_constant<int, 3>{});
  if (emit8 && p.amax8) {                           // (rows past M contribute their bias-only values: an over-estimate at worst)
    am8 = wave_max_u(am8);
    if (lane == 0) atomicMax(p.amax8 + (blockIdx.x & 63), am8);
  }
}

// EPI: which optional epilogue operands exist, as a COMPILE-TIME fact (bit 0 residual rows, bit 1 ReLU-mask rows; bf16 output, no
// e4m3 copy, no f32 residual stream) or -1 = decided at run time (every other combination, and the exact-f32 instantiations).
// With run-time flags every row of a pass is a chain of uniform branches: hipcc then neither interleaves the rows nor counts
// its vmcnt waits across them -- passes 2 and 3 waited vmcnt(0) for their residual rows, i.e. for the previous pass's stores.
// PERSIST: one workgroup per CU walks the tiles  first + i * gridDim.x  (the XCD-contiguous order xcd_remap gives the one-tile grid)
// one after the other -- nothing is carried from tile to tile (tools/tile_stamps.py: ~2.4 us pass between a workgroup's end and
// its successor's first instruction on the CU, and ~1 us of the start-up is kernel-argument and index arithmetic).
template <typename T, bool TAPS, bool RPOOL = false, int EPI = -1, bool PERSIST = false, bool SPLITK = false>
__global__ __launch_bounds__(512) void k_conv_fwd256(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2 * 2 * 2 * 128 * KCH];   // byte address = buf<<16 | ab<<15 | half<<14 | row*128 + slot*16
  const int t_in = threadIdx.x;
  p.x += (long)blockIdx.y * p.bx; p.w += (long)blockIdx.y * p.bw; p.y += (long)blockIdx.y * p.by;
  const int ntn = p.Cout >> 8;
  const int ntiles = PERSIST ? (p.tile_limit ? p.tile_limit : ntn * ((p.M + 255) >> 8)) : 0;
  int lbid = PERSIST ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3))
           : SPLITK ? p.tile0 + (int)(blockIdx.x / (unsigned)p.ksplits) : xcd_remap(blockIdx.x, gridDim.x);
  const int kt0 = SPLITK ? (int)(blockIdx.x % (unsigned)p.ksplits) * p.kper : 0;      // first K-tile of this block's share
  if (PERSIST && lbid >= ntiles) return;
  for (;;) {
  int t = t_in;
  if (PERSIST) asm volatile("" : "+v"(t));       // per-lane values are recomputed per tile, not carried through the main loop
  const int lane = t & 63;
#ifdef CDDMSL_TILE_STAMPS
  const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wvu >> 2, wc = wvu & 3;
  const int tile_n = lbid % ntn, tile_m = lbid / ntn;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int cl = (t & 7) ^ ((t >> 4) & 7);      // logical K chunk of this lane's LDS slot (slot ^ ((row>>1)&7))
  const int nkt = SPLITK ? min(p.kper, (p.Kc >> 3) - kt0) : (p.Kc >> 3);
  const int tpt = p.cpp >> 3;                   // K-tiles per filter tap

  // ---- staging state.  Sources are addressed as buffer base (per block, SGPRs) + per-lane byte offset (constant over
  // the K loop) + a wave-uniform running offset in the instruction's soffset: no per-lane pointer arithmetic in the loop.
  // A lane whose row is outside M, or whose current filter tap falls outside the image, sets bit 31 of its offset:
  // beyond num_records, the load then writes zeros into LDS.
  //   A half h, piece i -> tile row i*128 + h*64 + (t>>3);   B half j, piece i -> tile col (2i + (t>>8))*64 + j*32 + ((t>>3)&31)
  auto rowoff = [&](int m, int& iy0, int& ix0) {
    const unsigned tq = fdiv((unsigned)m, p.dWo), ox = m - tq * p.Wo;
    const unsigned img = fdiv(tq, p.dHo), oy = tq - img * p.Ho;
    iy0 = (int)oy * p.stride - p.pad; ix0 = (int)ox * p.stride - p.pad;
    return (((long)img * p.Hi + iy0) * p.Wi + ix0) * p.xrs * 16;
  };
  int iyb, ixb;
  const long base_a = rowoff(m0, iyb, ixb);      // rows of one tile ascend from here (2*pad <= K-1, checked by the host)
  const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + base_a), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long)n0 * p.wrs * 16), 0, 0x7fffffff, 0x00020000);
  unsigned va[2][2], vinv[2][2], vb[2][2];
  int iy0[2][2], ix0[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + i * 128 + h * 64 + (t >> 3);
      const bool vm = m < p.M;
      const long ro = rowoff(vm ? m : m0, iy0[h][i], ix0[h][i]);
      va[h][i] = (unsigned)(ro - base_a) + cl * 16;
      // rows past M: every tap invalid (TAPS) / bit 31 of the offset (no taps); also parks iy0 outside the image for the loops below
      if (!vm) { iy0[h][i] = -(1 << 20); if (!TAPS) va[h][i] |= 0x80000000u; }
      vinv[h][i] = 0;
      vb[h][i] = (unsigned)(((2 * i + (t >> 8)) * 64 + h * 32 + ((t >> 3) & 31)) * p.wrs + cl) * 16;
    }
  if (TAPS) {
    // tap (ky, kx) of a row is invalid when its input row OR its input column falls outside the image: one pass over the filter
    // columns builds the row's column mask, one over the filter rows places it (or an all-ones group) -- KH + KW iterations with
    // the lane's four rows side by side, where the former KH x KW loop per row took ~4 us of a 3x3 tile's start-up
    // (tools/tile_stamps.py: 5.0 us from kernel entry to the first operand request, 1.2 us for a 1x1 layer).
    unsigned xm[2][2] = {{0, 0}, {0, 0}};
    for (int kx = 0; kx < p.KW; ++kx)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) xm[h][i] |= ((unsigned)(ix0[h][i] + kx) >= (unsigned)p.Wi ? 1u : 0u) << kx;
    const unsigned full = (1u << p.KW) - 1u;
    for (int ky = 0; ky < p.KH; ++ky)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) vinv[h][i] |= ((unsigned)(iy0[h][i] + ky) >= (unsigned)p.Hi ? full : xm[h][i]) << (ky * p.KW);
  }
  const int step_col = (p.xrs - (p.cpp - KCH)) * 16;                               // next tap in the same filter row
  const int step_row = ((p.Wi - (p.KW - 1)) * p.xrs - (p.cpp - KCH)) * 16;         // first tap of the next filter row
  int left[2] = {tpt, tpt}, tap[2] = {0, 0}, kxs[2] = {0, 0};
  unsigned soa[2] = {0, 0}, sob[2] = {0, 0};
  if (SPLITK) {                                 // the streams start at K-tile kt0: inside filter tap kt0 / tpt
    const int tap0 = TAPS ? kt0 / tpt : 0, within = TAPS ? kt0 - tap0 * tpt : kt0;
    const int ky0 = TAPS ? tap0 / p.KW : 0, kx0 = tap0 - ky0 * p.KW;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      left[h] = tpt - (TAPS ? within : 0); tap[h] = tap0; kxs[h] = kx0;
      soa[h] = (unsigned)((ky0 * p.Wi + kx0) * p.xrs * 16 + within * KCH * 16);
      sob[h] = (unsigned)(kt0 * KCH * 16);
    }
  }

  char* const L = (char*)lds;
  auto stageA = [&](auto H, int buf) {
    constexpr int h = decltype(H)::value;
    char* dst = L + (buf << 16) + (h << 14) + wvu * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned v = va[h][i];
      if (TAPS) v |= __builtin_amdgcn_ubfe(vinv[h][i], (unsigned)tap[h], 1u) << 31;
      blds16(ra_rsrc, v, soa[h], dst + i * 8192);
    }
    if (TAPS) {
      int step = KCH * 16;
      if (--left[h] == 0) {
        left[h] = tpt; ++tap[h];
        if (++kxs[h] == p.KW) { kxs[h] = 0; step = step_row; } else step = step_col;
      }
      soa[h] += step;
    } else soa[h] += KCH * 16;
  };
  auto stageB = [&](auto J, int buf) {
    constexpr int j = decltype(J)::value;
    char* dst = L + (buf << 16) + (1 << 15) + (j << 14) + wvu * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) blds16(rb_rsrc, vb[j][i], sob[j], dst + i * 8192);
    sob[j] += KCH * 16;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- fragment reads: per-lane byte addresses per k-step, buffer bit (1<<16) toggled by XOR; half / row-tile offsets are immediates
  const int r32 = lane & 31, hh = lane >> 5, sw = (r32 >> 1) & 7;
  unsigned ada[4], adb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    ada[ks] = (unsigned)(((wr * 64 + r32) * KCH + ((2 * ks + hh) ^ sw)) * 16);
    adb[ks] = (unsigned)(((wc * 32 + r32) * KCH + ((2 * ks + hh) ^ sw)) * 16 + (1 << 15));
  }
  u32x4 fa0[2][4], fa1[2][4], fb0[4], fb1[4];
  auto readA = [&](auto I, u32x4 (*fa)[4]) {
    constexpr int i = decltype(I)::value;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[rt][ks] = *(const u32x4*)(L + ada[ks] + ((i << 14) + rt * 32 * KCH * 16));
  };
  auto readB = [&](auto J, u32x4* fb) {
    constexpr int j = decltype(J)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = *(const u32x4*)(L + adb[ks] + (j << 14));
  };
  auto flipA = [&]() {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ada[ks] ^= 1u << 16;
  };
  auto flipB = [&]() {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) adb[ks] ^= 1u << 16;
  };
#define CDDMSL_MMA_QUAD(I, J, FA, FB) MmaQuad<T>::run(acc[2 * (I)][J], acc[2 * (I) + 1][J], FA, FB)
#ifdef CDDMSL_STAMPS   // segment sums: 0 load section, 1 first barrier + LDS wait, 2 MFMAs, 3 second barrier
  unsigned long long st_sum[4] = {0, 0, 0, 0}, st_last = 0;
#define CDDMSL_STAMP(K) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long tt_ = __builtin_readcyclecounter(); \
                          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_sum[K] += tt_ - st_last; st_last = tt_; }
#else
#define CDDMSL_STAMP(K)
#endif
#define CDDMSL_PHASE_SYNC_IN()                                                      \
  CDDMSL_STAMP(0)                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                \
  __builtin_amdgcn_s_barrier();                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
  CDDMSL_STAMP(1)                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                \
  __builtin_amdgcn_s_setprio(1);
#define CDDMSL_PHASE_SYNC_OUT(I, J)                                                 \
  asm volatile("" : "+v"(acc[2 * (I)][J]), "+v"(acc[2 * (I) + 1][J]));   /* the MFMAs above cannot sink below the barrier */ \
  __builtin_amdgcn_s_setprio(0);                                                    \
  CDDMSL_STAMP(2)                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                \
  __builtin_amdgcn_s_barrier();                                                     \
  CDDMSL_STAMP(3)                                                                   \
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: tile 0 complete, tile 1 without its A1 half (staged by phase 1 of tile 0); A0 of tile 0 is read ahead
  stageA(I0{}, 0); stageA(I1{}, 0); stageB(I0{}, 0); stageB(I1{}, 0);
  if (nkt > 1) {
    stageA(I0{}, 1); stageB(I0{}, 1); stageB(I1{}, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  readA(I0{}, fa0);
  if (wr == 1) __builtin_amdgcn_s_barrier();     // group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);
#ifdef CDDMSL_STAMPS
  st_last = __builtin_readcyclecounter();
#endif
#ifdef CDDMSL_TILE_STAMPS
  const unsigned long long ts_loop = __builtin_amdgcn_s_memrealtime();
#endif

  // One K-tile = four phases.  LAST (compile time): the tile's final K-tile, peeled out of the loop -- nothing is left to stage and no
  // wait is due (it was retired by the K-tile before it, or by the prologue).
  // (Tried here and measured slower, +1 ms of kernel time per step: starting the epilogue's residual / mask rows on their way from HBM
  // with one dword load per 128-byte line -- a lane per row -- during this last K-tile.  The epilogue's first pass does wait ~2 us
  // for its operand rows, but 64 single-line requests per instruction cost the load path more than the wait.)
  // the epilogue's first operand rows ride in the A0 fragments' registers from phase 3 of the last K-tile on (see tile_epilogue PRE)
  constexpr bool PREF = EPI > 0 && !RPOOL && Mma<T>::ES == 2 && !SPLITK;
  u32x4 pre[2][4];
  f32x4 sb[4];
  auto ktile = [&](int kt, auto LAST) {
    constexpr bool last = decltype(LAST)::value;
    const int d = kt & 1;
    const bool more1 = !last, more2 = !last && kt + 2 < nkt;
    // phase 1
    readB(I0{}, fb0);
    if (more1) stageA(I1{}, d ^ 1);
    CDDMSL_PHASE_SYNC_IN();
    CDDMSL_MMA_QUAD(0, 0, fa0, fb0);
    CDDMSL_PHASE_SYNC_OUT(0, 0);
    // phase 2
    readB(I1{}, fb1);
    flipB();
    if (more2) stageA(I0{}, d);
    CDDMSL_PHASE_SYNC_IN();
    CDDMSL_MMA_QUAD(0, 1, fa0, fb1);
    CDDMSL_PHASE_SYNC_OUT(0, 1);
    // phase 3: the wait retires everything but the two youngest half-tiles, i.e. all of tile kt+1 (other buffer)
    readA(I1{}, fa1);
    flipA();
    if (more2) {
      stageB(I0{}, d);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (more1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (PREF) {
      const bool is_res = (EPI & 1) != 0;
      const char* base = is_res ? p.residual : p.relu_mask;
      const int ld = is_res ? p.ldr : p.ldm;
      long bytes = ((long)p.M - m0) * ld * 2;
      if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)m0 * ld * 2), 0, (int)bytes, 0x00020000);
      const unsigned vo = (unsigned)(((wr * 128 + (lane >> 3)) * ld + n0 + wc * 64 + (lane & 7) * 8) * 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[a][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (a * 32 + 8 * i) * ld * 2, 0);
    }
    CDDMSL_PHASE_SYNC_IN();
    CDDMSL_MMA_QUAD(1, 1, fa1, fb1);
    CDDMSL_PHASE_SYNC_OUT(1, 1);
    // phase 4
    if (more1) readA(I0{}, fa0);
    if (more2) stageB(I1{}, d);
    if (last && !SPLITK) {                          // the epilogue's scale / bias values (fb1 is dead from here on)
      const int n = n0 + wc * 64 + (lane & 7) * 8;
      const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
      sb[0] = p.scale ? *(const f32x4*)(p.scale + n) : one; sb[1] = p.scale ? *(const f32x4*)(p.scale + n + 4) : one;
      sb[2] = p.bias ? *(const f32x4*)(p.bias + n) : zero; sb[3] = p.bias ? *(const f32x4*)(p.bias + n + 4) : zero;
    }
    CDDMSL_PHASE_SYNC_IN();
    CDDMSL_MMA_QUAD(1, 0, fa1, fb0);
    CDDMSL_PHASE_SYNC_OUT(1, 0);
  };
  for (int kt = 0; kt + 1 < nkt; ++kt) ktile(kt, std::false_type{});
  ktile(nkt - 1, std::true_type{});
  if (wr == 0) __builtin_amdgcn_s_barrier();     // re-align the two groups (every wave has now passed all reads)
#ifdef CDDMSL_TILE_STAMPS
  const unsigned long long ts_epi = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef CDDMSL_STAMPS
  if (p.stamps && lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) p.stamps[((long)blockIdx.x * 8 + wvu) * 4 + k] = st_sum[k];
  }
#endif
#undef CDDMSL_MMA_QUAD
#undef CDDMSL_PHASE_SYNC_IN
#undef CDDMSL_PHASE_SYNC_OUT
#undef CDDMSL_STAMP

  if (SPLITK) {                                   // raw accumulators, fragment order: 32 x 16 bytes per lane, 1 KiB per wave instruction
    f32x4* dst = (f32x4*)p.partial + (long)blockIdx.x * (8 * 32 * 64) + (wvu * 32) * 64 + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 v = {acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
          dst[((a * 2 + b) * 4 + g4) * 64] = v;
        }
    return;
  }
  tile_epilogue<T, RPOOL, EPI, PREF>(p, acc, (float*)lds + wvu * 4096, wr, wc, lane, m0, n0, pre, sb);
#ifdef CDDMSL_TILE_STAMPS
  if (p.tstamps && lane == 0) {
    if (!PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (exit stamp = the wave's stores have left)
    unsigned long long* o = p.tstamps + ((long)(PERSIST ? lbid : (int)blockIdx.x) * 8 + wvu) * 4;
    o[0] = ts_entry; o[1] = ts_loop; o[2] = ts_epi; o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  if (!PERSIST) break;
  lbid += gridDim.x;
  if (lbid >= ntiles) break;
  __builtin_amdgcn_s_barrier();                    // the next tile's operand DMA overwrites the other waves' transposition scratch
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 128 tile, 4 waves (2 x 2; 128 x 64 per wave -- the 256x256 kernel's per-wave tile and operand reuse), TWO workgroups per CU.
//
// The 256x256 kernel owns its CU: its tile's start-up (operands' first trip from HBM), main loop and epilogue traffic run one
// after the other, which leaves the layers whose epilogue moves as many bytes as their main loop takes time at about half of
// either roofline (tools/tile_stamps.py).  Here two independent workgroups share the CU's matrix pipes and memory path: while
// one drains its tile the other multiplies.  No ping-pong between wave groups (each SIMD hosts one wave of each workgroup, not
// synchronised with each other); instead each wave pipelines itself: the LDS reads of K-tile kt+1 (12 x 16 bytes per lane) and
// the LDS-DMA of K-tile kt+3 are issued in front of K-tile kt's 16 MFMAs, one barrier per K-tile.
// K-tile = 4 chunks (32 bf16): LDS ring of 3 stages x (256 + 128 rows x 64 B) = 72 KiB per workgroup; 64-byte rows, chunk
// ^= (row >> 1) & 3 on the source side of the DMA and on the ds_read_b128 side (8 consecutive rows cover the 8 bank groups).
// One operand stream (all six DMAs of a K-tile share the filter-tap state).  Epilogue: tile_epilogue, scratch = the idle ring.
// ------------------------------------------------------------------------------------------------
template <typename T, bool TAPS, int EPI>
__global__ __launch_bounds__(256, 2) void k_conv_fwd2(ConvArgs p) {
  constexpr int SA = 256 * 64, SB = 128 * 64, SS = SA + SB, STAGES = 3;
  __shared__ __attribute__((aligned(16))) u32x4 lds[STAGES * SS / 16];
  const int t = threadIdx.x, lane = t & 63;
#ifdef CDDMSL_TILE_STAMPS
  const unsigned long long ts_entry = __builtin_amdgcn_s_memrealtime();
#endif
  p.x += (long)blockIdx.y * p.bx; p.w += (long)blockIdx.y * p.bw; p.y += (long)blockIdx.y * p.by;
  const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wvu >> 1, wc = wvu & 1;
  const int ntn = p.Cout >> 7;
  const int lbid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = lbid % ntn, tile_m = lbid / ntn;
  const int m0 = tile_m * 256, n0 = tile_n * 128;
  const int cl = (t & 3) ^ ((t >> 3) & 3);      // logical K chunk of this lane's LDS slot (slot ^ ((row>>1)&3)); rows (t>>2) + 64 i
  const int nkt = p.Kc >> 2;
  const int tpt = p.cpp >> 2;                   // K-tiles per filter tap

  auto rowoff = [&](int m, int& iy0, int& ix0) {
    const unsigned tq = fdiv((unsigned)m, p.dWo), ox = m - tq * p.Wo;
    const unsigned img = fdiv(tq, p.dHo), oy = tq - img * p.Ho;
    iy0 = (int)oy * p.stride - p.pad; ix0 = (int)ox * p.stride - p.pad;
    return (((long)img * p.Hi + iy0) * p.Wi + ix0) * p.xrs * 16;
  };
  int iyb, ixb;
  const long base_a = rowoff(m0, iyb, ixb);
  const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + base_a), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long)n0 * p.wrs * 16), 0, 0x7fffffff, 0x00020000);
  unsigned va[4], vinv[4], vb[2];
  int iy0[4], ix0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + (t >> 2) + 64 * i;
    const bool vm = m < p.M;
    const long ro = rowoff(vm ? m : m0, iy0[i], ix0[i]);
    va[i] = (unsigned)(ro - base_a) + cl * 16;
    if (!vm) { iy0[i] = -(1 << 20); if (!TAPS) va[i] |= 0x80000000u; }
    vinv[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) vb[i] = (unsigned)((((t >> 2) + 64 * i) * p.wrs + cl) * 16);
  if (TAPS) {                                   // (as in k_conv_fwd256: column mask, then one step per filter row)
    unsigned xm[4] = {0, 0, 0, 0};
    for (int kx = 0; kx < p.KW; ++kx)
#pragma unroll
      for (int i = 0; i < 4; ++i) xm[i] |= ((unsigned)(ix0[i] + kx) >= (unsigned)p.Wi ? 1u : 0u) << kx;
    const unsigned full = (1u << p.KW) - 1u;
    for (int ky = 0; ky < p.KH; ++ky)
#pragma unroll
      for (int i = 0; i < 4; ++i) vinv[i] |= ((unsigned)(iy0[i] + ky) >= (unsigned)p.Hi ? full : xm[i]) << (ky * p.KW);
  }
  const int step_col = (p.xrs - (p.cpp - 4)) * 16;
  const int step_row = ((p.Wi - (p.KW - 1)) * p.xrs - (p.cpp - 4)) * 16;
  int left = tpt, tap = 0, kxs = 0;
  unsigned soa = 0, sob = 0;

  char* const L = (char*)lds;
  auto stage = [&](int st) {                    // the next K-tile of the operand stream -> ring stage st
    char* dst = L + st * SS + wvu * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned v = va[i];
      if (TAPS) v |= __builtin_amdgcn_ubfe(vinv[i], (unsigned)tap, 1u) << 31;
      blds16(ra_rsrc, v, soa, dst + i * 4096);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) blds16(rb_rsrc, vb[i], sob, dst + SA + i * 4096);
    sob += 64;
    if (TAPS) {
      int step = 64;
      if (--left == 0) {
        left = tpt; ++tap;
        if (++kxs == p.KW) { kxs = 0; step = step_row; } else step = step_col;
      }
      soa += step;
    } else soa += 64;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int r32 = lane & 31, hh = lane >> 5, sw = (r32 >> 1) & 3;
  unsigned ada[2], adb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    ada[ks] = (unsigned)(((wr * 128 + r32) * 4 + ((2 * ks + hh) ^ sw)) * 16);
    adb[ks] = (unsigned)(SA + ((wc * 64 + r32) * 4 + ((2 * ks + hh) ^ sw)) * 16);
  }
  u32x4 fa[2][4][2], fb[2][2][2];               // [register set][32-row / 32-column tile][k-step]
  auto readf = [&](auto SET, int st) {
    constexpr int set = decltype(SET)::value;
    const char* base = L + st * SS;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[set][rt][ks] = *(const u32x4*)(base + ada[ks] + rt * 32 * 64);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[set][ct][ks] = *(const u32x4*)(base + adb[ks] + ct * 32 * 64);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- prologue: K-tiles 0..2 requested, K-tile 0 landed and read
  stage(0);
  if (nkt > 1) stage(1);
  if (nkt > 2) stage(2);
  if (nkt > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nkt > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  readf(S0{}, 0);
#ifdef CDDMSL_TILE_STAMPS
  const unsigned long long ts_loop = __builtin_amdgcn_s_memrealtime();
#endif
  int st_next = 1, st_free = 0;                 // ring stage of K-tile kt+1 / stage K-tile kt+3 goes to (= K-tile kt's)

  auto ktile = [&](int kt, auto SET) {
    constexpr int set = decltype(SET)::value;
    // K-tile kt+1 has landed (this lane's share; the barrier makes it everyone's), K-tile kt's fragments have been read
    if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0), as the builtin: hipcc's own wait insertion then knows the fragments
                                                 // of K-tile kt are in, and does not put a lgkmcnt(0) -- which would also wait for
                                                 // K-tile kt+1's reads -- in front of the MFMAs
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nkt) readf(std::integral_constant<int, set ^ 1>{}, st_next);
    if (kt + 3 < nkt) stage(st_free);
    st_next = st_next == STAGES - 1 ? 0 : st_next + 1;
    st_free = st_free == STAGES - 1 ? 0 : st_free + 1;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) Mma<T>::step(acc[rt][ct], fa[set][rt][ks], fb[set][ct][ks]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    ktile(kt, S0{});
    if (kt + 1 < nkt) ktile(kt + 1, S1{});
  }
#ifdef CDDMSL_TILE_STAMPS
  asm volatile("" : "+v"(acc[3][1]));
  const unsigned long long ts_epi = __builtin_amdgcn_s_memrealtime();
#endif
  // every wave has passed the last K-tile's barrier with its reads retired and no DMA in flight: the ring is free
  tile_epilogue<T, false, EPI>(p, acc, (float*)lds + wvu * 4096, wr, wc, lane, m0, n0);
#ifdef CDDMSL_TILE_STAMPS
  if (p.tstamps && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = p.tstamps + ((long)blockIdx.x * 4 + wvu) * 4;
    o[0] = ts_entry; o[1] = ts_loop; o[2] = ts_epi; o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// Shapes the 256x256 kernel takes: whole 256-column tiles, K-tiles inside one filter tap, vector epilogue, <= 32 taps,
// and enough tiles to fill the chip.  Environment CDDMSL_GEMM256 (read per launch, so one process can A/B): 0 = always the
// 128x128 kernel, 2 = the 256x256 kernel wherever it is legal, unset/1 = the heuristic below.
static bool gemm256_legal(const ConvArgs& a) {
  const bool vec_ok = (a.ldy % 8 == 0) && (!a.residual || a.ldr % 8 == 0) && (!a.relu_mask || a.ldm % 8 == 0);
  if (a.pool || (a.cpp & 7) || (a.Cout & 255) || !vec_ok || a.KH * a.KW > 31) return false;
  if (2 * a.pad > a.KH - 1 || 2 * a.pad > a.KW - 1) return false;   // rows of a tile must ascend in memory (per-block buffer base)
  return true;
}
static bool use_gemm256(const ConvArgs& a) {
  const char* e = getenv("CDDMSL_GEMM256");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return false;
  if (!gemm256_legal(a)) return false;
  if (mode == 2) return true;                                   // forced (tests)
  // (per-shape A/B inside the training step: 196 tiles (M 25088, N 512) run 1.3-1.5x faster here, 100 tiles and fewer slower)
  const long tiles = (long)((a.M + 255) / 256) * (a.Cout / 256) * g_batch_peek();
  return tiles >= 160;
}


// Sum of a tile's K-splits + the epilogue (bf16 output; scale / bias, residual, ReLU, ReLU mask): one thread per 16-byte slot of
// the fragment-ordered partials = four consecutive rows of one output column.  Only ever a handful of tiles per launch.
__global__ __launch_bounds__(256) void k_conv_split_reduce(ConvArgs p) {
  const int tile_rel = blockIdx.x >> 6, q = (blockIdx.x & 63) * 256 + threadIdx.x;
  const f32x4* src = (const f32x4*)p.partial + (long)tile_rel * p.ksplits * 16384 + q;
  f32x4 sum = src[0];
  for (int s = 1; s < p.ksplits; ++s) sum += src[(long)s * 16384];
  const int wvu = q >> 11, j = (q >> 6) & 31, lane = q & 63, r32 = lane & 31, hh = lane >> 5;
  const int a = j >> 3, b = (j >> 2) & 1, g4 = j & 3, wr = wvu >> 2, wc = wvu & 3;
  const int ntn = p.Cout >> 8, lbid = p.tile0 + tile_rel;
  const int tile_n = lbid % ntn, tile_m = lbid / ntn;
  const int n = tile_n * 256 + wc * 64 + b * 32 + r32;
  const int mrow = tile_m * 256 + wr * 128 + a * 32 + 8 * g4 + 4 * hh;
  const float sc = p.scale ? p.scale[n] : 1.f, bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long m = mrow + e;
    if (m >= p.M) break;
    float v = __builtin_fmaf(sum[e], sc, bi);
    if (p.residual) v += bf2f(*(const unsigned short*)(p.residual + (m * p.ldr + n) * 2));
    if (p.relu) v = fmaxf(v, 0.f);
    if (p.relu_mask && !(bf2f(*(const unsigned short*)(p.relu_mask + (m * p.ldm + n) * 2)) > 0.f)) v = 0.f;
    *(unsigned short*)(p.y + (m * p.ldy + n) * 2) = f2bf(v);
  }
}

static void* g_ws = nullptr;       // device workspace for split reductions (cddmsl_set_workspace); process-wide: one device per process
static long g_ws_bytes = 0;

// Workgroups of the persistent form of the 256x256 kernel: one per CU (a multiple of 8, dealt round-robin over the XCDs), or 0 = use
// the one-tile-per-workgroup grid (CDDMSL_PERSIST=0; read per launch, so one process can A/B).
static int persistent_blocks_raw() {
  static int ncu = -1;
  if (ncu < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    ncu = (n / 8) * 8;
  }
  return ncu;
}
static int persistent_blocks() {
  const char* e = getenv("CDDMSL_PERSIST");
  return (e ? atoi(e) : 1) ? persistent_blocks_raw() : 0;
}

// The 256x128 two-workgroup kernel: whole 128-column tiles, K-tiles of 4 chunks inside one filter tap, vector epilogue.
// CDDMSL_FWD2 (read per launch): 0 = never, 2 = wherever legal (tests, A/B), unset / 1 = the heuristic.
static bool fwd2_legal(const ConvArgs& a) {
  const bool vec_ok = (a.ldy % 8 == 0) && (!a.residual || a.ldr % 8 == 0) && (!a.relu_mask || a.ldm % 8 == 0);
  if (a.pool || a.res_pool || (a.cpp & 3) || (a.Cout & 127) || !vec_ok || a.KH * a.KW > 31) return false;
  if (2 * a.pad > a.KH - 1 || 2 * a.pad > a.KW - 1) return false;
  return true;
}
static bool use_fwd2(const ConvArgs& a) {
  const char* e = getenv("CDDMSL_FWD2");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0 || !fwd2_legal(a)) return false;
  if (mode == 2) return true;
  // What the 256x256 kernel does not take (Cout = 128, 384, ...; too few 256x256 tiles), when there are enough 256x128 tiles to
  // give every CU work: per shape (two dispatches in one process) 1.14-1.31x the 128x128 kernel on the 128-channel 3x3 layers,
  // 1.04-1.23x on their 1x1 layers; against the 256x256 kernel it loses (x0.72-0.99) on everything but K = 128.
  if (use_gemm256(a) || g_batch_peek() != 1) return false;
  const char* et = getenv("CDDMSL_FWD2_MIN");                   // (A/B knob)
  return (long)(a.Cout / 128) * ((a.M + 255) / 256) >= (et ? atol(et) : 256);
}
template <typename T> void launch_fwd2(const ConvArgs& a, dim3 grid, hipStream_t st) {
  const bool taps = !(a.KH == 1 && a.KW == 1 && a.pad == 0);
  if (sizeof(T) == 4 || a.out_f32 || a.res_f32 || a.y8) {
    if (taps) hipLaunchKernelGGL((k_conv_fwd2<T, true, -1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_fwd2<T, false, -1>), grid, dim3(256), 0, st, a);
    return;
  }
  if constexpr (sizeof(T) == 2) {
    const int epi = (a.residual ? 1 : 0) | (a.relu_mask ? 2 : 0);
#define CDDMSL_L2(TP, E) hipLaunchKernelGGL((k_conv_fwd2<T, TP, E>), grid, dim3(256), 0, st, a)
    if (taps) { switch (epi) { case 0: CDDMSL_L2(true, 0); break; case 1: CDDMSL_L2(true, 1); break; case 2: CDDMSL_L2(true, 2); break; default: CDDMSL_L2(true, 3); } }
    else { switch (epi) { case 0: CDDMSL_L2(false, 0); break; case 1: CDDMSL_L2(false, 1); break; case 2: CDDMSL_L2(false, 2); break; default: CDDMSL_L2(false, 3); } }
#undef CDDMSL_L2
  }
}

template <typename T, bool TAPS> void launch256_main(const ConvArgs& a, dim3 grid, hipStream_t st, int epi);
// the 256x256 kernel's epilogue variant (template parameter EPI): compile-time operand set for bf16 outputs, run-time flags otherwise
constexpr int TAIL_FRAC_DEFAULT = 8;
template <typename T, bool TAPS> void launch256(const ConvArgs& a, dim3 grid, hipStream_t st) {
  if (sizeof(T) == 4 || a.out_f32 || a.res_f32 || a.y8) { hipLaunchKernelGGL((k_conv_fwd256<T, TAPS, false, -1>), grid, dim3(512), 0, st, a); return; }
  const int epi = (a.residual ? 1 : 0) | (a.relu_mask ? 2 : 0);
  // Tail of a badly quantised launch.  16 x 50 x 83 pixels are 260 row panels: a 256-column layer of res4 is 260 tiles for 256 CUs --
  // two rounds of workgroups, the second with 4 of them (15 + 16 such launches per step, ~55 and ~30 us each wasted).  When the last
  // round would be less than an eighth full, the main launch stops at the last full round and the leftover tiles are computed
  // split along K (every CU takes a slice; raw accumulators to the workspace) and finished by k_conv_split_reduce.
  if constexpr (std::is_same<T, __bf16>::value) {
    const int ncu = persistent_blocks_raw();
    const int tiles = (int)grid.x, rem = ncu > 0 ? tiles % ncu : 0, nktot = a.Kc >> 3;
    const char* et = getenv("CDDMSL_TAIL_SPLIT");
    const char* ef = getenv("CDDMSL_TAIL_FRAC");                 // the last round counts as "nearly empty" below 1 / FRAC of the CUs
    const int frac = ef ? (atoi(ef) > 0 ? atoi(ef) : 8) : TAIL_FRAC_DEFAULT;
    if (!(et && atoi(et) == 0) && grid.y == 1 && tiles > ncu && rem > 0 && rem * frac <= ncu && nktot >= 16 && g_ws) {
      int S = ncu / rem;
      if (S > nktot / 2) S = nktot / 2;
      { const char* es = getenv("CDDMSL_TAIL_MAXS"); if (es && S > atoi(es)) S = atoi(es); }      // (A/B knob)
      const int kper = (nktot + S - 1) / S;
      S = (nktot + kper - 1) / kper;
      if (S >= 2 && (long)rem * S * 65536 * 4 <= g_ws_bytes) {
        ConvArgs m = a, t = a;
        m.tile_limit = tiles - rem;
        launch256_main<T, TAPS>(m, dim3((unsigned)(tiles - rem), 1), st, epi);
        t.partial = (float*)g_ws; t.tile0 = tiles - rem; t.ksplits = S; t.kper = kper;
        hipLaunchKernelGGL((k_conv_fwd256<T, TAPS, false, -1, false, true>), dim3((unsigned)(rem * S)), dim3(512), 0, st, t);
        hipLaunchKernelGGL(k_conv_split_reduce, dim3((unsigned)(rem * 64)), dim3(256), 0, st, t);
        return;
      }
    }
  }
  launch256_main<T, TAPS>(a, grid, st, epi);
}
template <typename T, bool TAPS> void launch256_main(const ConvArgs& a, dim3 grid, hipStream_t st, int epi) {
  // Persistent form (bf16, no taps) for SHORT reductions only: per shape, two builds in one process, K <= 512 layers gain 4-6 %
  // (the ~2.4 us between workgroups is 10-20 % of such a tile), K >= 2048 layers lose 2-4 % against the hardware's dynamic
  // dispatch; in the step k_conv_fwd256 50.9 -> 50.4 ms.
  if constexpr (std::is_same<T, __bf16>::value && !TAPS) {
    const int nb = persistent_blocks();
    const char* emk = getenv("CDDMSL_PERSIST_MAXKT");             // (A/B knob) longest reduction, in K-tiles, that takes the persistent form
    if (nb > 0 && grid.y == 1 && (int)grid.x > nb && (a.Kc >> 3) <= (emk ? atoi(emk) : 8)) {
      switch (epi) {
        case 0: hipLaunchKernelGGL((k_conv_fwd256<T, false, false, 0, true>), dim3(nb), dim3(512), 0, st, a); break;
        case 1: hipLaunchKernelGGL((k_conv_fwd256<T, false, false, 1, true>), dim3(nb), dim3(512), 0, st, a); break;
        case 2: hipLaunchKernelGGL((k_conv_fwd256<T, false, false, 2, true>), dim3(nb), dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL((k_conv_fwd256<T, false, false, 3, true>), dim3(nb), dim3(512), 0, st, a); break;
      }
      return;
    }
  }
  switch (epi) {
    case 0: hipLaunchKernelGGL((k_conv_fwd256<T, TAPS, false, 0>), grid, dim3(512), 0, st, a); break;
    case 1: hipLaunchKernelGGL((k_conv_fwd256<T, TAPS, false, 1>), grid, dim3(512), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_conv_fwd256<T, TAPS, false, 2>), grid, dim3(512), 0, st, a); break;
    default: hipLaunchKernelGGL((k_conv_fwd256<T, TAPS, false, 3>), grid, dim3(512), 0, st, a); break;
  }
}
template <> void launch256<float, false>(const ConvArgs& a, dim3 grid, hipStream_t st) { hipLaunchKernelGGL((k_conv_fwd256<float, false, false, -1>), grid, dim3(512), 0, st, a); }
template <> void launch256<float, true>(const ConvArgs& a, dim3 grid, hipStream_t st) { hipLaunchKernelGGL((k_conv_fwd256<float, true, false, -1>), grid, dim3(512), 0, st, a); }

template <typename T> int conv_fwd_launch(ConvArgs& a, hipStream_t st) {
  int ntn = (a.Cout + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
  long grid = (long)ntn * ntm;
  if (grid <= 0) return CDDMSL_OK;
  if (grid > 0x7fffffffL) return CDDMSL_ERR_ARG;
  // few-channel 3x3 layers (the CLIP stem): streaming register-weight kernel
  // (8 chunks per pixel = the 64 -> 64 layers of res2 in bf16, forward and -- with the ReLU mask -- input gradient)
  if (!a.pool && a.KH == 3 && a.KW == 3 && a.pad == 1 && (a.cpp == 1 || a.cpp == 4 || (a.cpp == 8 && a.Cout == 64)) &&
      (a.Cout == 32 || a.Cout == 64) && !a.residual && !a.out_f32 && g_batch == 1 && a.xrs == a.cpp && a.wrs == a.Kc) {
    g_last_kernel = 8;
    if (g_plan_only) return CDDMSL_OK;
    // grid-stride over 32-pixel tiles.  Register weights (one chunk per pixel): 8 blocks of 4 waves per CU.  LDS weights (up to
    // 72 KiB per block, two blocks fit a CU): exactly the resident blocks, so that the weight image is filled once per CU slot
    // (2048 blocks refilled it every 4 tiles: 2.9 -> 2.7 ms/step)
    const int nb = a.cpp > 1 ? 256 * 2 : 256 * 8;
    if (a.cpp == 1 && a.Cout == 32) hipLaunchKernelGGL((k_conv3x3_small<T, 1, 1>), dim3(nb), dim3(256), 0, st, a);
    else if (a.cpp == 1) hipLaunchKernelGGL((k_conv3x3_small<T, 1, 2>), dim3(nb), dim3(256), 0, st, a);
    else if (a.cpp == 8) hipLaunchKernelGGL((k_conv3x3_small<T, 8, 2>), dim3(nb), dim3(256), 0, st, a);
    else if (a.Cout == 32) hipLaunchKernelGGL((k_conv3x3_small<T, 4, 1>), dim3(nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv3x3_small<T, 4, 2>), dim3(nb), dim3(256), 0, st, a);
    return launch_status();
  }
  // ... and the 1x1 layers of res2 with 64 output channels (64 -> 64, 256 -> 64: one 128-column tile of the GEMM kernels would be half
  // empty): the same kernel with one tap, bf16.  CDDMSL_SMALL_1X1=0: the 128x128 GEMM kernel (A/B)
  if (sizeof(T) == 2 && !a.pool && a.KH == 1 && a.KW == 1 && a.pad == 0 && a.stride == 1 && a.Cout == 64 && (a.cpp == 8 || a.cpp == 32) &&
      !a.residual && !a.out_f32 && !a.y8 && g_batch == 1 && a.xrs == a.cpp && a.wrs == a.Kc && a.ldy == 64 &&
      !(getenv("CDDMSL_SMALL_1X1") && atoi(getenv("CDDMSL_SMALL_1X1")) == 0)) {
    g_last_kernel = 8;
    if (g_plan_only) return CDDMSL_OK;
    if (a.cpp == 8) hipLaunchKernelGGL((k_conv3x3_small<T, 8, 2, 1>), dim3(512), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv3x3_small<T, 32, 2, 1>), dim3(512), dim3(256), 0, st, a);
    return launch_status();
  }
  if (use_fwd2(a)) {
    grid = (long)(a.Cout / 128) * ((a.M + 255) / 256);
    g_last_kernel = 11;
    if (g_plan_only) return CDDMSL_OK;
    launch_fwd2<T>(a, dim3((unsigned)grid, (unsigned)g_batch), st);
    return launch_status();
  }
  if (use_gemm256(a)) {
    grid = (long)(a.Cout / 256) * ((a.M + 255) / 256);
    g_last_kernel = 3;
    if (g_plan_only) return CDDMSL_OK;
    const dim3 g256((unsigned)grid, (unsigned)g_batch);
    if (a.res_pool) hipLaunchKernelGGL((k_conv_fwd256<T, false, true>), g256, dim3(512), 0, st, a);
    else if (a.KH == 1 && a.KW == 1 && a.pad == 0) launch256<T, false>(a, g256, st);
    else launch256<T, true>(a, g256, st);
  } else if (a.pool) { g_last_kernel = 2; if (g_plan_only) return CDDMSL_OK; hipLaunchKernelGGL(k_conv_fwd_reg<T>, dim3((unsigned)grid), dim3(256), 0, st, a); }
  else { g_last_kernel = 1; if (g_plan_only) return CDDMSL_OK; hipLaunchKernelGGL(k_conv_fwd<T>, dim3((unsigned)grid, (unsigned)g_batch), dim3(256), 0, st, a); }
  return launch_status();
}

}  // namespace

#ifdef CDDMSL_TILE_STAMPS
static unsigned long long* g_tile_stamps = nullptr;
extern "C" void cddmsl_debug_tile_stamps(unsigned long long* p) { g_tile_stamps = p; }
#endif
extern "C" int cddmsl_last_kernel(void) { return g_last_kernel; }
extern "C" int cddmsl_plan_only(int on) { const int was = g_plan_only; g_plan_only = on; return was; }

// outputs up to CDDMSL_NT_MIN_MB MiB are stored with the default policy (their consumer may still find them in the 256 MiB last-level cache), larger
// ones non-temporal (same-box A/B of the training step, 3 runs each: never 102.38 ms, always 101.60, above 128 MiB 101.66; on another box 100 MiB was
// 0.4 ms ahead of always)
static int nt_out_for(long out_bytes) {
  static const long nt_min_mb = getenv("CDDMSL_NT_MIN_MB") ? atol(getenv("CDDMSL_NT_MIN_MB")) : 128;
  return out_bytes > (nt_min_mb << 20) ? 1 : 0;
}

static int conv_fwd_impl(const void* x, const void* w, void* y, const float* scale, const float* bias,
                         const void* residual, const void* relu_mask, int Nimg, int Hi, int Wi, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int pool, int ldy, int ldr, int ldm,
                         int relu, int out_f32, int dtype, void* y8, const float* q8, float* amax8, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if (dtype != 0 && dtype != 1) return CDDMSL_ERR_ARG;
  if (Nimg < 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return CDDMSL_ERR_ARG;
  if ((Cin * es) % 16 != 0) return CDDMSL_ERR_ARG;            // channel rows must be whole 16-B chunks
  if (pool && (KH != 1 || KW != 1 || pad != 0 || stride != 1)) return CDDMSL_ERR_ARG;
  ConvArgs a;
  a.x = (const char*)x; a.w = (const char*)w; a.y = (char*)y; a.scale = scale; a.bias = bias;
  a.residual = (const char*)residual; a.relu_mask = (const char*)relu_mask;
  a.Nimg = Nimg; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  if (pool) { a.Ho = Hi / 2; a.Wo = Wi / 2; }
  else { a.Ho = (Hi + 2 * pad - KH) / stride + 1; a.Wo = (Wi + 2 * pad - KW) / stride + 1; }
  if (a.Ho <= 0 || a.Wo <= 0) return CDDMSL_ERR_ARG;
  // out_f32 bit 1: the residual rows are f32 (bf16 kernels, f32 output, no ReLU mask) -- the mapper's f32 residual stream
  if ((out_f32 & 2) && (!(out_f32 & 1) || !residual || relu_mask || pool)) return CDDMSL_ERR_ARG;
  a.ldy = ldy; a.ldr = ldr; a.ldm = ldm; a.relu = relu; a.out_f32 = out_f32 & 1; a.pool = pool;
  a.res_f32 = (dtype == 0 && (out_f32 & 2)) ? 1 : 0;
  // out_f32 bit 2: the residual is a 2x2-average-pooled gradient (the downsample path's input gradient at pooled resolution);
  // buffer-addressed from the tensor base, so the pooled tensor must stay below 2 GiB
  a.res_pool = (out_f32 & 4) ? 1 : 0;
  if (a.res_pool && (!residual || (out_f32 & 2) || pool || KH != 1 || KW != 1 || pad != 0 || (long)Nimg * (a.Ho / 2) * (a.Wo / 2) * ldr * es >= (1L << 31) ||
                     (Cout & 7) || (ldy & 7) || (ldr & 7) || (relu_mask && (ldm & 7)))) return CDDMSL_ERR_ARG;
  if (a.res_f32 && ((Cout & 7) || (ldy & 7) || (ldr & 7))) return CDDMSL_ERR_ARG;                // (vector epilogue only)
  long M = (long)Nimg * a.Ho * a.Wo;
  if (M > 0x7fffff00L) return CDDMSL_ERR_ARG;
  a.M = (int)M; a.cpp = Cin * es / 16; a.Kc = KH * KW * a.cpp;
  a.dWo = make_fastdiv((unsigned)a.Wo); a.dHo = make_fastdiv((unsigned)a.Ho);
  a.dcpp = make_fastdiv((unsigned)a.cpp); a.dKW = make_fastdiv((unsigned)KW);
  a.xrs = a.cpp; a.wrs = a.Kc; a.bx = a.bw = a.by = 0;
#ifdef CDDMSL_TILE_STAMPS
  a.tstamps = g_tile_stamps;
#endif
  if (a.M == 0) return CDDMSL_OK;
  a.nt_out = nt_out_for((long)a.M * Cout * 2);
  if (y8) {       // e4m3 second output: bf16 launches of the 256x256 kernel only (its epilogue writes it)
    if (dtype != 0 || (out_f32 & 1) || ldy != Cout || !use_gemm256(a)) return CDDMSL_ERR_ARG;
    a.y8 = (char*)y8; a.q8 = q8; a.amax8 = (unsigned*)amax8;
  }
  return dtype == 0 ? conv_fwd_launch<__bf16>(a, (hipStream_t)stream) : conv_fwd_launch<float>(a, (hipStream_t)stream);
}

extern "C" int cddmsl_conv_fwd(const void* x, const void* w, void* y, const float* scale, const float* bias,
                               const void* residual, const void* relu_mask, int Nimg, int Hi, int Wi, int Cin,
                               int Cout, int KH, int KW, int stride, int pad, int pool, int ldy, int ldr, int ldm,
                               int relu, int out_f32, int dtype, void* stream) {
  return conv_fwd_impl(x, w, y, scale, bias, residual, relu_mask, Nimg, Hi, Wi, Cin, Cout, KH, KW, stride, pad, pool, ldy, ldr, ldm,
                       relu, out_f32, dtype, nullptr, nullptr, nullptr, stream);
}

// cddmsl_conv_fwd (bf16) that ALSO writes y8 [M][Cout] = OCP e4m3 of sat(y * q8[0]) and max-es |y| into amax8[0..63] (64 floats,
// spread by block to keep the atomics off one address): the producer side of the fp8 configuration -- the convolution that
// consumes y reads y8 instead of a separate quantisation pass.  Only launches the 256x256 kernel takes (CDDMSL_ERR_ARG otherwise:
// the caller asks cddmsl_conv_fwd_q8_ok first).
extern "C" int cddmsl_conv_fwd_q8(const void* x, const void* w, void* y, const float* scale, const float* bias,
                                  const void* residual, const void* relu_mask, int Nimg, int Hi, int Wi, int Cin,
                                  int Cout, int KH, int KW, int stride, int pad, int relu, void* y8, const float* q8, float* amax8,
                                  void* stream) {
  if (!y8) return CDDMSL_ERR_ARG;
  return conv_fwd_impl(x, w, y, scale, bias, residual, relu_mask, Nimg, Hi, Wi, Cin, Cout, KH, KW, stride, pad, 0, Cout, Cout, Cout,
                       relu, 0, 0, y8, q8, amax8, stream);
}

// e4m3 x e4m3 -> bf16 (or f32) on the 256x256 kernel: x [Nimg][Hi][Wi][Cin] and w [Cout][KH][KW][Cin] hold OCP e4m3 bytes (Cin a
// multiple of 128), everything else as cddmsl_conv_fwd with dtype 0: scale / bias f32 per output channel (the caller folds the two
// per-tensor dequantisation factors into ``scale``), residual / relu_mask / y bf16 (y f32 with out_f32).  Only shapes the 256x256
// kernel takes (Cout % 256 == 0, <= 31 taps, "same" padding at most); anything else is CDDMSL_ERR_ARG -- there is no fallback.
extern "C" int cddmsl_conv_fwd_fp8(const void* x, const void* w, void* y, const float* scale, const float* bias,
                                   const void* residual, const void* relu_mask, int Nimg, int Hi, int Wi, int Cin, int Cout, int KH,
                                   int KW, int pad, int relu, int out_f32, void* y8, const float* q8, float* amax8, void* stream) {
  if (Nimg < 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || (Cin % 128) != 0 || (out_f32 & ~1)) return CDDMSL_ERR_ARG;
  ConvArgs a;
  a.x = (const char*)x; a.w = (const char*)w; a.y = (char*)y; a.scale = scale; a.bias = bias;
  a.residual = (const char*)residual; a.relu_mask = (const char*)relu_mask;
  a.Nimg = Nimg; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = 1; a.pad = pad;
  a.Ho = Hi + 2 * pad - KH + 1; a.Wo = Wi + 2 * pad - KW + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return CDDMSL_ERR_ARG;
  a.ldy = a.ldr = a.ldm = Cout; a.relu = relu; a.out_f32 = out_f32; a.pool = 0; a.res_f32 = 0; a.res_pool = 0;
  long M = (long)Nimg * a.Ho * a.Wo;
  if (M > 0x7fffff00L) return CDDMSL_ERR_ARG;
  a.M = (int)M; a.cpp = Cin / 16; a.Kc = KH * KW * a.cpp;
  a.nt_out = nt_out_for(M * Cout * 2);
  a.dWo = make_fastdiv((unsigned)a.Wo); a.dHo = make_fastdiv((unsigned)a.Ho);
  a.dcpp = make_fastdiv((unsigned)a.cpp); a.dKW = make_fastdiv((unsigned)KW);
  a.xrs = a.cpp; a.wrs = a.Kc; a.bx = a.bw = a.by = 0;
  if (a.M == 0) return CDDMSL_OK;
  if (!gemm256_legal(a)) return CDDMSL_ERR_ARG;
  if (y8) {
    if (out_f32) return CDDMSL_ERR_ARG;
    a.y8 = (char*)y8; a.q8 = q8; a.amax8 = (unsigned*)amax8;
  }
  const long grid = (long)(a.Cout / 256) * ((a.M + 255) / 256);
  if (grid > 0x7fffffffL) return CDDMSL_ERR_ARG;
  g_last_kernel = 10;
  if (g_plan_only) return CDDMSL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (KH == 1 && KW == 1 && pad == 0) launch256<fp8e4, false>(a, dim3((unsigned)grid, 1), st);
  else launch256<fp8e4, true>(a, dim3((unsigned)grid, 1), st);
  return launch_status();
}

// The 256x256 wgrad kernel takes bf16 "same" problems with whole 256-wide output tiles and 8-chunk-aligned pixels
// (so the two X halves of a lane share one filter tap).  CDDMSL_GEMM256 as for the forward kernel (0 = never, 2 = always).
static bool wgrad256_ok(const WgradArgs& a, int batch) {
  const char* e = getenv("CDDMSL_GEMM256");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return false;
  if ((a.Cout & 255) || (a.K & 255) || (a.cpp & 7) || (a.ldd & 7)) return false;
  if (mode == 2) return true;
  // long reductions only: each block ends with 64 Ki scalar atomics, which a short m range cannot amortise
  // (threshold from per-shape A/B inside the training step: 9342 (M 66400, 256 x 2304) and 14112 (M 25088, 512 x 4608) run
  // 1.6x faster here than on the 128x128 kernel, 8300 (M 265600, 512 x 256) and everything below run slower)
  const char* et = getenv("CDDMSL_WGRAD256_MIN");               // (A/B knob)
  return (long)(a.Cout / 256) * (a.K / 256) * batch * ((a.M + WM - 1) / WM) >= (et ? atol(et) : 4000);
}
// buffer addressing: lane offset + soffset must stay below 2 GiB inside one block's reduction range
static bool wgrad256_span_ok(const WgradArgs& a) {
  const long rowb = (long)(a.ldd * 2 > a.xrs * 16 ? a.ldd * 2 : a.xrs * 16);
  return ((long)a.mtiles_per_split * WM + WM + 2L * a.Wi + 2) * rowb + (1L << 20) < (1L << 31);
}

// ------------------------------------------------------------------------------------------------
// Split reductions of the weight-gradient kernels without atomics.  64 Ki f32 atomics per 256x256 block take 50-68 us whatever
// the order (tools/tile_stamps.py: 1.3 TB/s of atomic payload chip-wide; 45 % of a layer3 launch of k_wgrad256, ~a third of a
// k_conv_wgrad_dma launch).  With a workspace registered, every block stores its accumulators as they lie in the registers
// (fragment order: 1 KiB per wave instruction), and this kernel sums a tile's splits -- one thread per 16-byte slot -- and adds the
// result, scaled, to dw: plain read-modify-write, nothing else touches dw on the stream meanwhile, and the sum is deterministic.
// WAVES x FR = waves per block x 16-byte slots per lane: 8 x 32 (k_wgrad256: 256x256 tile), 4 x 16 (k_conv_wgrad_dma, bf16: 128x128).
// ------------------------------------------------------------------------------------------------
template <int WAVES, int FR>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const f32x4* ws, float* dw, const float* scale, int ntn, int ntk, int splits, int Cout, int K, int ldo) {
  constexpr int SLOTS = WAVES * FR * 64;                      // 16-byte slots per tile
  constexpr int TN = WAVES == 8 ? 256 : 128;                  // tile edge
  const int tile = blockIdx.x / (SLOTS / 256), q = (blockIdx.x % (SLOTS / 256)) * 256 + threadIdx.x;
  const int tile_k = tile % ntk, tile_n = tile / ntk;
  const long ntiles = (long)ntn * ntk;
  const f32x4* src = ws + (long)tile * SLOTS + q;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= splits; s += 4) {                            // four loads in flight
    const f32x4 a = src[(s + 0) * ntiles * SLOTS], b = src[(s + 1) * ntiles * SLOTS], c = src[(s + 2) * ntiles * SLOTS], d = src[(s + 3) * ntiles * SLOTS];
    sum += (a + b) + (c + d);
  }
  for (; s < splits; ++s) sum += src[s * ntiles * SLOTS];
  const int wv = q / (FR * 64), j = (q / 64) % FR, lane = q & 63, r = lane & 31, h = lane >> 5, g4 = j & 3;
  int a, b, wn, wk;
  if (WAVES == 8) { a = j >> 3; b = (j >> 2) & 1; wn = wv >> 2; wk = wv & 3; }
  else { a = j >> 3; b = (j >> 2) & 1; wn = wv >> 1; wk = wv & 1; }
  const int n = tile_n * TN + wn * (WAVES == 8 ? 128 : 64) + a * 32 + 8 * g4 + 4 * h;
  const int k = tile_k * TN + wk * 64 + b * 32 + r;
  if (k >= K) return;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < Cout) dw[(long)(n + e) * ldo + k] += sum[e] * (scale ? scale[n + e] : 1.f);
}

extern "C" int cddmsl_set_workspace(void* ptr, long bytes) {
  if (bytes < 0 || (ptr == nullptr && bytes != 0) || ((size_t)ptr & 15)) return CDDMSL_ERR_ARG;
  g_ws = ptr; g_ws_bytes = bytes;
  return CDDMSL_OK;
}
// whether a split reduction of `blocks` tiles of `tile_floats` goes through the workspace (CDDMSL_WGRAD_WS=0: atomics, for A/B)
static bool use_workspace(long blocks, long tile_floats) {
  const char* e = getenv("CDDMSL_WGRAD_WS");
  if (e && atoi(e) == 0) return false;
  return g_ws != nullptr && blocks * tile_floats * 4 <= g_ws_bytes;
}

extern "C" int cddmsl_conv_wgrad(const void* x, const void* dy, float* dw, const float* scale, int Nimg, int Hi,
                                 int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int pool, int ldd,
                                 int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if (dtype != 0 && dtype != 1) return CDDMSL_ERR_ARG;
  if (Nimg < 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return CDDMSL_ERR_ARG;
  if ((Cin * es) % 16 != 0 || (Cout * es) % 16 != 0 || (ldd * es) % 16 != 0) return CDDMSL_ERR_ARG;
  if (pool && (KH != 1 || KW != 1 || pad != 0 || stride != 1)) return CDDMSL_ERR_ARG;
  WgradArgs a;
  a.x = (const char*)x; a.dy = (const char*)dy; a.dw = dw; a.scale = scale;
  a.Nimg = Nimg; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  a.ldd = ldd; a.pool = pool;
  if (pool) { a.Ho = Hi / 2; a.Wo = Wi / 2; }
  else { a.Ho = (Hi + 2 * pad - KH) / stride + 1; a.Wo = (Wi + 2 * pad - KW) / stride + 1; }
  if (a.Ho <= 0 || a.Wo <= 0) return CDDMSL_ERR_ARG;
  long M = (long)Nimg * a.Ho * a.Wo;
  if (M > 0x7fffff00L) return CDDMSL_ERR_ARG;
  a.M = (int)M; a.cpp = Cin * es / 16; a.Kc = KH * KW * a.cpp; a.K = KH * KW * Cin; a.ncc = Cout * es / 16;
  a.dWo = make_fastdiv((unsigned)a.Wo); a.dHo = make_fastdiv((unsigned)a.Ho);
  a.xrs = a.cpp; a.ldo = a.K; a.direct = 0; a.bx = a.bd = a.bo = 0;
#ifdef CDDMSL_TILE_STAMPS
  a.tstamps = g_tile_stamps;
#endif
  if (a.M == 0) return CDDMSL_OK;
  int cols = 256 / es;
  long tiles = (long)((Cout + cols - 1) / cols) * ((a.K + cols - 1) / cols);
  int total_mt = (a.M + WM - 1) / WM;
  // Split count: one round of blocks (2 per CU) for 1x1 layers, two for filters with taps, and at least 8 m-tiles per block.
  // Every block ends with 16 Ki f32 atomics; with 2048+ blocks the atomic traffic at L2, not the reduction, set the
  // time of the short backbone layers (measured: 150 -> 67 us at M = 66 400, N = 1024, K = 256).
  const char* eb = getenv("CDDMSL_WGRAD_BLOCKS");                  // tuning knob (A/B runs): target number of blocks
  const long target = eb ? atol(eb) : ((KH == 1 && KW == 1) ? 512 : 1024);
  long maxs = (total_mt + 7) / 8;
  // ... and a WHOLE number of 512-block rounds (two resident blocks per CU): with 9 output tiles (128 x 1152) a target of 1024
  // gave 114 splits = 1026 blocks, i.e. a third round for two blocks.  Candidates: the largest split count that stays inside
  // r rounds, r = the target's rounds and one more; the one whose last round is fullest wins (fewer rounds on ties).
  long splits = 1;
  {
    const long r0 = (target + 511) / 512;
    double best = -1.0;
    for (long r = r0; r <= r0 + 1; ++r) {
      long c = (512 * r) / tiles;
      if (c < 1) c = 1;
      if (c > maxs) c = maxs;
      const int mps = (int)((total_mt + c - 1) / c);
      const long real = (total_mt + mps - 1) / mps, blocks = tiles * real, rounds = (blocks + 511) / 512;
      const double eff = (double)blocks / (512.0 * rounds);
      if (eff > best + 0.03) { best = eff; splits = c; }
      if (c == maxs) break;
    }
    if (eb) { splits = (target + tiles - 1) / tiles; if (splits > maxs) splits = maxs; }
  }
  if (splits < 1) splits = 1;
  a.mtiles_per_split = (int)((total_mt + splits - 1) / splits);
  splits = (total_mt + a.mtiles_per_split - 1) / a.mtiles_per_split;
  long grid = tiles * splits;
  if (grid > 0x7fffffffL) return CDDMSL_ERR_ARG;
  const bool same = !pool && stride == 1 && a.Ho == Hi && a.Wo == Wi;   // LDS-DMA kernel: output pixel == input pixel
  if (same && dtype == 0 && wgrad256_ok(a, 1)) {
    // 256x256 ping-pong kernel: ONE block per CU, so the grid should be a whole number of 256-block rounds: take the split
    // count whose grid fills its last round best (fewest rounds on ties: every block ends with 64 Ki atomics), with at
    // least 16 reduction tiles per block.  Measured on N = 2048, K = 512: 256 blocks 683 us vs 640 blocks 894 us.
    long tiles2 = (long)(Cout / 256) * (a.K / 256);
    long maxs2 = (total_mt + 15) / 16;
    long sp = 1;
    double best = -1.0;
    const char* eb2 = getenv("CDDMSL_WGRAD256_BLOCKS");            // tuning knob (A/B runs): force ~this many blocks
    for (int r = 1; r <= 6 && !eb2; ++r) {
      long c = (256L * r) / tiles2;
      if (c < 1) continue;
      if (c > maxs2) c = maxs2;
      const long blocks = tiles2 * c, rounds = (blocks + 255) / 256;
      const double eff = (double)blocks / (256.0 * rounds);
      if (eff > best + 0.02) { best = eff; sp = c; }
      if (c == maxs2) break;
    }
    if (eb2) { sp = (atol(eb2) + tiles2 - 1) / tiles2; if (sp > maxs2) sp = maxs2; }
    if (sp < 1) sp = 1;
    int keep = a.mtiles_per_split;
    a.mtiles_per_split = (int)((total_mt + sp - 1) / sp);
    sp = (total_mt + a.mtiles_per_split - 1) / a.mtiles_per_split;
    if (wgrad256_span_ok(a)) {
      g_last_kernel = 6;
      if (g_plan_only) return CDDMSL_OK;
      if (sp > 1 && use_workspace(tiles2 * sp, 65536)) {
        a.ws = (float*)g_ws;
        hipLaunchKernelGGL(k_wgrad256, dim3((unsigned)(tiles2 * sp)), dim3(512), 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL((k_wgrad_reduce<8, 32>), dim3((unsigned)(tiles2 * 64)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g_ws, dw, scale,
                           Cout / 256, a.K / 256, (int)sp, Cout, a.K, a.ldo);
        return launch_status();
      }
      hipLaunchKernelGGL(k_wgrad256, dim3((unsigned)(tiles2 * sp)), dim3(512), 0, (hipStream_t)stream, a);
      return launch_status();
    }
    a.mtiles_per_split = keep;
  }
  g_last_kernel = same ? 5 : 4;
  if (g_plan_only) return CDDMSL_OK;
  if (same && dtype == 0 && splits > 1 && use_workspace(grid, 16384)) {
    a.ws = (float*)g_ws;
    hipLaunchKernelGGL(k_conv_wgrad_dma<__bf16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL((k_wgrad_reduce<4, 16>), dim3((unsigned)(tiles * 16)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g_ws, dw, scale,
                       (Cout + 127) / 128, (a.K + 127) / 128, (int)splits, Cout, a.K, a.ldo);
  } else if (same) {
    if (dtype == 0) hipLaunchKernelGGL(k_conv_wgrad_dma<__bf16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_conv_wgrad_dma<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    if (dtype == 0) hipLaunchKernelGGL(k_conv_wgrad<__bf16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_conv_wgrad<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  }
  return launch_status();
}

// fp8 configuration: dW[Cout][KH*KW*Cin] (f32) += scale[n] * sum_m dy8[m][n] * im2col(x8)[m][k], both operands OCP e4m3 bytes (NHWC, dy
// rows ldd bytes apart); ``scale`` must carry the two dequantisation factors (x the FrozenBN scale).  "Same" convolutions only
// (stride 1, 2 * pad == KH - 1), Cout, Cin multiples of 256.  cddmsl_conv_wgrad_fp8_ok tells whether a shape is taken.
extern "C" int cddmsl_conv_wgrad_fp8_ok(int Cin, int Cout, int KH, int KW, int pad, int ldd) {
  return (Cin % 256 == 0 && Cout % 256 == 0 && KH == KW && (KH & 1) && 2 * pad == KH - 1 && ldd % 16 == 0) ? 1 : 0;
}
extern "C" int cddmsl_conv_wgrad_fp8(const void* x8, const void* dy8, float* dw, const float* scale, int Nimg, int Hi, int Wi, int Cin,
                                     int Cout, int KH, int KW, int pad, int ldd, void* stream) {
  if (Nimg < 0 || Hi <= 0 || Wi <= 0 || !cddmsl_conv_wgrad_fp8_ok(Cin, Cout, KH, KW, pad, ldd) || ldd < Cout) return CDDMSL_ERR_ARG;
  WgradArgs a;
  a.x = (const char*)x8; a.dy = (const char*)dy8; a.dw = dw; a.scale = scale;
  a.Nimg = Nimg; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = 1; a.pad = pad;
  a.ldd = ldd; a.pool = 0; a.Ho = Hi; a.Wo = Wi;
  const long M = (long)Nimg * Hi * Wi;
  if (M > 0x7fffff00L) return CDDMSL_ERR_ARG;
  a.M = (int)M; a.cpp = Cin / 16; a.Kc = KH * KW * a.cpp; a.K = KH * KW * Cin; a.ncc = Cout / 16;
  a.dWo = make_fastdiv((unsigned)a.Wo); a.dHo = make_fastdiv((unsigned)a.Ho);
  a.xrs = a.cpp; a.ldo = a.K; a.direct = 0; a.bx = a.bd = a.bo = 0;
  if (a.M == 0) return CDDMSL_OK;
  const int total_mt = (a.M + WM - 1) / WM;
  // one block per CU: the split count whose grid fills its last 256-block round best, at least 16 reduction tiles per block
  const long tiles = (long)(Cout / 256) * (a.K / 256), maxs = (total_mt + 15) / 16;
  long sp = 1;
  double best = -1.0;
  for (int r = 1; r <= 6; ++r) {
    long c = (256L * r) / tiles;
    if (c < 1) continue;
    if (c > maxs) c = maxs;
    const long blocks = tiles * c, rounds = (blocks + 255) / 256;
    const double eff = (double)blocks / (256.0 * rounds);
    if (eff > best + 0.02) { best = eff; sp = c; }
    if (c == maxs) break;
  }
  a.mtiles_per_split = (int)((total_mt + sp - 1) / sp);
  sp = (total_mt + a.mtiles_per_split - 1) / a.mtiles_per_split;
  // buffer addressing: lane offset + soffset stay below 2 GiB inside one block's reduction range
  const long rowb = ldd > a.xrs * 16 ? ldd : a.xrs * 16;
  if (((long)a.mtiles_per_split * WM + WM + 2L * Wi + 2) * rowb + (1L << 20) >= (1L << 31)) return CDDMSL_ERR_ARG;
  g_last_kernel = 12;
  if (g_plan_only) return CDDMSL_OK;
  if (sp > 1 && use_workspace(tiles * sp, 65536)) {
    a.ws = (float*)g_ws;
    hipLaunchKernelGGL(k_wgrad256_f8, dim3((unsigned)(tiles * sp)), dim3(512), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL((k_wgrad_reduce<8, 32>), dim3((unsigned)(tiles * 64)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g_ws, dw, scale,
                       Cout / 256, a.K / 256, (int)sp, Cout, a.K, a.ldo);
    return launch_status();
  }
  hipLaunchKernelGGL(k_wgrad256_f8, dim3((unsigned)(tiles * sp)), dim3(512), 0, (hipStream_t)stream, a);
  return launch_status();
}

// Batched "NT" GEMM on the conv kernel: for b in [0,batch): C_b[m][n] = sum_k A_b[m][k] * B_b[n][k] (+ bias[n]),
// A_b = a + b*sa, rows lda apart; B_b = w + b*sw, rows ldb apart; C_b = c + b*sc, rows ldc apart (strides in ELEMENTS).
// Used by the reassociated attention pool (per-head and per-region products).
extern "C" int cddmsl_gemm_nt_batched(const void* a, const void* w, void* c, const float* bias, int M, int N, int K, int lda,
                                      int ldb, int ldc, int batch, long sa, long sw, long sc, int out_f32, int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if (dtype != 0 && dtype != 1) return CDDMSL_ERR_ARG;
  if (M < 0 || N <= 0 || K <= 0 || batch < 0 || batch > 65535) return CDDMSL_ERR_ARG;
  if ((K * es) % 16 || (lda * es) % 16 || (ldb * es) % 16 || (sa * es) % 16 || (sw * es) % 16) return CDDMSL_ERR_ARG;
  if (M == 0 || batch == 0) return CDDMSL_OK;
  ConvArgs p;
  p.x = (const char*)a; p.w = (const char*)w; p.y = (char*)c; p.scale = nullptr; p.bias = bias; p.residual = nullptr; p.relu_mask = nullptr;
  p.Nimg = 1; p.Hi = 1; p.Wi = M; p.Cin = K; p.Ho = 1; p.Wo = M; p.Cout = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
  p.ldy = ldc; p.ldr = 0; p.ldm = 0; p.relu = 0; p.out_f32 = out_f32; p.pool = 0; p.res_f32 = 0; p.res_pool = 0;
  p.M = M; p.cpp = K * es / 16; p.Kc = p.cpp;
  p.dWo = make_fastdiv((unsigned)M); p.dHo = make_fastdiv(1u); p.dcpp = make_fastdiv((unsigned)p.cpp); p.dKW = make_fastdiv(1u);
  p.xrs = lda * es / 16; p.wrs = ldb * es / 16;
  p.bx = sa * es; p.bw = sw * es; p.by = sc * (out_f32 ? 4 : es);
  g_batch = batch;
  int st = dtype == 0 ? conv_fwd_launch<__bf16>(p, (hipStream_t)stream) : conv_fwd_launch<float>(p, (hipStream_t)stream);
  g_batch = 1;
  return st;
}

// Batched "TN" GEMM on the LDS-DMA wgrad kernel: out_b[n][k] (+)= sum_m A_b[m][n] * B_b[m][k]; A rows lda apart (n contiguous),
// B rows ldb apart (k contiguous), out rows ldo apart.  mode 0: f32 atomic accumulate (large M is split over blocks),
// 1: f32 store, 2: `dtype` store (modes 1/2 need M <= 64*8 so one block owns a tile... enforced: single split).
extern "C" int cddmsl_gemm_tn_batched(const void* a, const void* b, void* out, int M, int N, int K, int lda, int ldb, int ldo,
                                      int batch, long sa, long sb, long so, int mode, int dtype, void* stream) {
  int es = dtype == 0 ? 2 : 4;
  if (dtype != 0 && dtype != 1) return CDDMSL_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535 || mode < 0 || mode > 2) return CDDMSL_ERR_ARG;
  if ((K * es) % 16 || (N * es) % 16 || (lda * es) % 16 || (ldb * es) % 16 || (sa * es) % 16 || (sb * es) % 16) return CDDMSL_ERR_ARG;
  WgradArgs p;
  p.x = (const char*)b; p.dy = (const char*)a; p.dw = (float*)out; p.scale = nullptr;
  p.Nimg = 1; p.Hi = 1; p.Wi = M; p.Cin = K; p.Ho = 1; p.Wo = M; p.Cout = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
  p.ldd = lda; p.pool = 0; p.M = M; p.cpp = K * es / 16; p.Kc = p.cpp; p.K = K; p.ncc = N * es / 16;
  p.dWo = make_fastdiv((unsigned)M); p.dHo = make_fastdiv(1u);
  p.xrs = ldb * es / 16; p.ldo = ldo; p.direct = mode;
  p.bx = sb * es; p.bd = sa * es; p.bo = so * (mode == 2 ? es : 4);
  int cols = 256 / es;
  long tiles = (long)((N + cols - 1) / cols) * ((K + cols - 1) / cols);
  int total_mt = (M + WM - 1) / WM;
  long splits = 1;
  if (mode == 0) {
    long want = (2048 + tiles * batch - 1) / (tiles * batch);
    long maxs = (total_mt + 7) / 8;
    splits = want < 1 ? 1 : (want > maxs ? maxs : want);
    if (splits < 1) splits = 1;
  }
  if (dtype == 0 && total_mt == 1 && N <= 64 && batch >= 64 && K % 128 == 0 && N % 8 == 0) {
    // one reduction tile, narrow output (the attention pool's per-region products): compact three-stage ring
    const long kt = K / 128;
    long bpb = (kt * batch + 4095) / 4096;
    if (bpb < 8) bpb = 8;
    if (bpb > batch) bpb = batch;
    const unsigned gy = (unsigned)((batch + bpb - 1) / bpb);
    g_last_kernel = 9;
    if (g_plan_only) return CDDMSL_OK;
#define CDDMSL_TNS(NG)                                                                                                               \
  case NG:                                                                                                                           \
    if (mode == 0) hipLaunchKernelGGL((k_gemm_tn_small<NG, 0>), dim3((unsigned)kt, gy), dim3(256), 0, (hipStream_t)stream, p, batch, (int)bpb);      \
    else if (mode == 1) hipLaunchKernelGGL((k_gemm_tn_small<NG, 1>), dim3((unsigned)kt, gy), dim3(256), 0, (hipStream_t)stream, p, batch, (int)bpb); \
    else hipLaunchKernelGGL((k_gemm_tn_small<NG, 2>), dim3((unsigned)kt, gy), dim3(256), 0, (hipStream_t)stream, p, batch, (int)bpb);                \
    break;
    switch (N / 8) { CDDMSL_TNS(1) CDDMSL_TNS(2) CDDMSL_TNS(3) CDDMSL_TNS(4) CDDMSL_TNS(5) CDDMSL_TNS(6) CDDMSL_TNS(7) CDDMSL_TNS(8) }
#undef CDDMSL_TNS
    return launch_status();
  }
  if (total_mt <= 4 && batch >= 64) {
    // short reductions over many batches: stream runs of batches through one block (k_gemm_tn_stream)
    long bpb = (tiles * batch + 4095) / 4096, minb = (8 + total_mt - 1) / total_mt;
    if (bpb < minb) bpb = minb;
    if (bpb > batch) bpb = batch;
    const unsigned gy = (unsigned)((batch + bpb - 1) / bpb);
    g_last_kernel = 7;
    if (g_plan_only) return CDDMSL_OK;
    if (dtype == 0) hipLaunchKernelGGL(k_gemm_tn_stream<__bf16>, dim3((unsigned)tiles, gy), dim3(256), 0, (hipStream_t)stream, p, batch, (int)bpb);
    else hipLaunchKernelGGL(k_gemm_tn_stream<float>, dim3((unsigned)tiles, gy), dim3(256), 0, (hipStream_t)stream, p, batch, (int)bpb);
    return launch_status();
  }
  p.mtiles_per_split = (int)((total_mt + splits - 1) / splits);
  splits = (total_mt + p.mtiles_per_split - 1) / p.mtiles_per_split;
  long grid = tiles * splits;
  if (grid > 0x7fffffffL) return CDDMSL_ERR_ARG;
  g_last_kernel = 5;
  if (g_plan_only) return CDDMSL_OK;
  if (dtype == 0) hipLaunchKernelGGL(k_conv_wgrad_dma<__bf16>, dim3((unsigned)grid, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(k_conv_wgrad_dma<float>, dim3((unsigned)grid, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status();
}

// dx [K][P][C] (bf16) and gpos [P+1][C] (f32, accumulated; nullable) of the CLIP attention pool from  pds [K][2H][TP] = [p ; ds]
// and  zu [K][2H][C] = [dZ ; U]  (cddmsl_amd/layers.py AttnPoolFn.backward; clip_backbone.py:83-107):  one batched TN product whose
// epilogue finishes the token gradients (see k_gemm_tn_small MODE 3).  g0 [K][C] f32 = the query path's gradient of the mean token;
// mbits [K][C] = sign bits of the pooled map per column (bit t = pixel t kept), all ones when the map is not a ReLU output.
extern "C" int cddmsl_attnpool_dx(const void* pds, const void* zu, const float* g0, const unsigned long long* mbits, void* dx, float* gpos,
                                  int K, int H2, int P, int TP, int C, int dtype, void* stream) {
  if (dtype != 0 || K < 0 || P != 49 || TP != 56 || H2 <= 0 || H2 > 64 || (H2 & 7) || C <= 0 || (C & 127) || !g0 || !mbits || !dx) return CDDMSL_ERR_ARG;
  if (K == 0) return CDDMSL_OK;
  if (K > 65535 * 8) return CDDMSL_ERR_ARG;
  WgradArgs p;
  p.x = (const char*)zu; p.dy = (const char*)pds; p.dw = (float*)dx; p.scale = nullptr;
  p.Nimg = 1; p.Hi = 1; p.Wi = H2; p.Cin = C; p.Ho = 1; p.Wo = H2; p.Cout = TP; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
  p.ldd = TP; p.pool = 0; p.M = H2; p.cpp = C * 2 / 16; p.Kc = p.cpp; p.K = C; p.ncc = TP * 2 / 16;
  p.dWo = make_fastdiv((unsigned)H2); p.dHo = make_fastdiv(1u);
  p.xrs = C * 2 / 16; p.ldo = C; p.direct = 3;
  p.bx = (long)H2 * C * 2; p.bd = (long)H2 * TP * 2; p.bo = (long)P * C * 2;
  p.g0 = g0; p.mbits = mbits; p.gpos = gpos;
  const long kt = C / 128;
  long bpb = (kt * K + 4095) / 4096;
  if (bpb < 8) bpb = 8;
  if (bpb > K) bpb = K;
  const unsigned gy = (unsigned)((K + bpb - 1) / bpb);
  hipLaunchKernelGGL((k_gemm_tn_small<7, 3, 49>), dim3((unsigned)kt, gy), dim3(256), 0, (hipStream_t)stream, p, K, (int)bpb);
  return launch_status();
}

extern "C" int cddmsl_weight_prep_multi(const long long* table, int count, int dtype, void* stream) {
  if ((dtype != 0 && dtype != 1) || count < 0 || count > 65535) return CDDMSL_ERR_ARG;
  if (count == 0) return CDDMSL_OK;
  if (dtype == 0) hipLaunchKernelGGL(k_weight_prep_multi<__bf16>, dim3(256, (unsigned)count), dim3(256), 0, (hipStream_t)stream, table);
  else hipLaunchKernelGGL(k_weight_prep_multi<float>, dim3(256, (unsigned)count), dim3(256), 0, (hipStream_t)stream, table);
  return launch_status();
}

extern "C" int cddmsl_weight_prep(const float* w, const float* scale, void* w_fwd, void* w_dgrad, int Cout, int KH,
                                  int KW, int Cin, int dtype, void* stream) {
  if (dtype != 0 && dtype != 1) return CDDMSL_ERR_ARG;
  long n = (long)Cout * KH * KW * Cin;
  if (n <= 0) return n == 0 ? CDDMSL_OK : CDDMSL_ERR_ARG;
  const long tiles = (long)((Cout + 31) / 32) * KH * KW * ((Cin + 31) / 32);         // 32 x 32 tiles per tap
  unsigned grid = (unsigned)(tiles > 4096 ? 4096 : tiles);
  if (dtype == 0) hipLaunchKernelGGL(k_weight_prep<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, scale, (char*)w_fwd, (char*)w_dgrad, Cout, KH, KW, Cin);
  else hipLaunchKernelGGL(k_weight_prep<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, scale, (char*)w_fwd, (char*)w_dgrad, Cout, KH, KW, Cin);
  return launch_status();
}
