// Fused multi-head attention for SHORT sequences (the ClipCap TransformerMapper: 80 tokens, 8 heads of 96 --
// detectron2/modeling/backbone/clipcap/clipcap.py:39-83, MultiHeadAttention.forward: softmax(QK^T * scale) V with no mask).
// bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 softmax.  One workgroup = one (sequence, head); three waves, wave w owns
// query rows [32w, 32w+32) (tokens padded to 96 with zero rows, head dim = 96 = 3 MFMA tiles).
//
// Operands live as row-major [96][96] bf16 images in LDS (row stride 208 B) when some product contracts them along their
// ROWS (read with ds_read_b64_tr_b16, the transposing read); an operand that is only ever contracted along its COLUMNS is
// loaded from global memory straight into MFMA fragments (Q, K in the forward; V in the backward) -- so no transposed copy
// of anything is ever made and the LDS footprint allows 4 (forward) / 2 (backward) workgroups per CU:
//     S  = Q K^T      (cols, cols)        O  = P V        (cols, rows)
//     dP = dO V^T     (cols, cols)        dV = P^T dO     (rows, rows)
//     dQ = dS K       (cols, rows)        dK = dS^T Q     (rows, rows)
// The backward recomputes S and P from Q, K (cheaper than saving them), dS = P o (dP - rowsum(dP o P)) * scale.
#include "common.h"

namespace {

constexpr int AT = 96;        // padded tokens = head dim
constexpr int ARS = 208;      // image row stride in bytes (192 data + 16 pad: b128 reads of 16 rows hit 16 distinct slots)
constexpr int IMG = AT * ARS; // 19968 B

struct AttnArgs {
  const char *q, *k, *v, *dout;
  char *o, *dq, *dk, *dv;
  int nseq, t, heads, ldq, ldk, ldv, ldo;   // row strides in elements
  float scale;
};

// fragment of an operand whose contraction index runs along the image columns: lane (r, hh) <- row row0 + r, elements 16 kk + 8 hh ..+8
__device__ __forceinline__ u32x4 frag_cols(const char* img, int row0, int kk, int lane) {
  return *(const u32x4*)(img + (row0 + (lane & 31)) * ARS + (kk * 16 + (lane >> 5) * 8) * 2);
}
// fragment of an operand whose contraction index runs along the image rows: lane (c, hh) <- column col0 + c, rows 16 kk + 8 hh ..+8
__device__ __forceinline__ u32x4 frag_rows(const char* img, int col0, int kk, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3, hh = g >> 1;
  const int col = col0 + 16 * (g & 1) + 4 * pp;
  const char* a0 = img + (kk * 16 + 8 * hh + q) * ARS + col * 2;
  const i16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)a0);
  const i16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(a0 + 4 * ARS));
  const u32x2 p0 = __builtin_bit_cast(u32x2, v0), p1 = __builtin_bit_cast(u32x2, v1);
  return u32x4{p0[0], p0[1], p1[0], p1[1]};
}
// the same fragment straight from global memory (rows >= t read as zero): operand rows are [t][ld] elements, 16-byte aligned
__device__ __forceinline__ u32x4 gfrag_cols(const char* g, int ld, int t, int row0, int kk, int lane) {
  const int row = row0 + (lane & 31);
  u32x4 v = {0u, 0u, 0u, 0u};
  if (row < t) v = *(const u32x4*)(g + ((long)row * ld + kk * 16 + (lane >> 5) * 8) * 2);
  return v;
}
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
// All-reduce over the 32 lanes that share lane>>5.  __shfl_xor lowers to ds_bpermute_b32 -- an LDS round trip per step, 5 per
// reduction, 240 in the backward kernel, each behind an s_waitcnt -- so the steps inside a 16-lane row are DPP moves (vector ALU, no
// wait: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror; once every lane of a group holds the group's value the
// mirrors pair whole groups) and only the last one crosses rows: ds_swizzle, bit mode, lane ^ 16.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float swz16(float v) {     // the value of lane ^ 16 (and_mask 0x1f, or_mask 0, xor_mask 0x10)
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));
}
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
  return v + swz16(v);
}
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(v, swz16(v));
}

// global [t rows][96 cols] (row stride ld elements) -> LDS image, rows t..95 zeroed.  In two steps, so that a kernel can REQUEST every
// chunk of all its images before it waits for the first: written as one loop (load, LDS write, next chunk) hipcc kept the loop and
// put s_waitcnt vmcnt(0) in front of every write -- 6 (forward) / 18 (backward) serialised memory round trips per workgroup, most of
// its lifetime at two workgroups per CU (found in the ISA; the backward kernel went 236 -> ~100 us per launch).
__device__ __forceinline__ void stage_load(u32x4 (&v)[6], const char* g, int t, int ld, int tid) {
#pragma unroll
  for (int it = 0; it < 6; ++it) {                 // 96 rows x 12 chunks = 6 per thread of the 192
    const int c = tid + it * 192, row = c / 12, ch = c - row * 12;
    // (branch-free: a padding row reads the last valid row and is zeroed afterwards -- under `if (row < t)` hipcc put a wait behind
    // every load of the forward kernel)
    const u32x4 x = *(const u32x4*)(g + ((long)(row < t ? row : t - 1) * ld) * 2 + ch * 16);
    const unsigned keep = row < t ? 0xffffffffu : 0u;
    v[it] = u32x4{x[0] & keep, x[1] & keep, x[2] & keep, x[3] & keep};
  }
}
__device__ __forceinline__ void stage_store(char* img, const u32x4 (&v)[6], int tid) {
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int c = tid + it * 192, row = c / 12, ch = c - row * 12;
    *(u32x4*)(img + row * ARS + ch * 16) = v[it];
  }
}
// C tile (32 x 32, rows row0.., cols col0..) -> bf16 image
__device__ __forceinline__ void put_tile(char* img, int row0, int col0, const f32x16& a, int lane) {
  const int c = col0 + (lane & 31), hh = lane >> 5;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int r = row0 + (g & 3) + 8 * (g >> 2) + 4 * hh;
    *(unsigned short*)(img + r * ARS + c * 2) = f2bf(a[g]);
  }
}
// C tile -> global bf16 rows < t
__device__ __forceinline__ void store_tile(char* gp, int ld, int t, int row0, int col0, const f32x16& a, int lane) {
  const int c = col0 + (lane & 31), hh = lane >> 5;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int r = row0 + (g & 3) + 8 * (g >> 2) + 4 * hh;
    if (r < t) *(unsigned short*)(gp + ((long)r * ld + c) * 2) = f2bf(a[g]);
  }
}

// this wave's 32 image rows (row0.., those < t) -> global, 16 bytes per lane (the tiles were put there by the same wave:
// DS operations of one wave execute in order, no barrier needed)
__device__ __forceinline__ void rows_out(const char* img, char* gp, int ld, int t, int row0, int lane) {
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int c = it * 64 + lane, row = row0 + c / 12, ch = c % 12;
    if (row < t) *(u32x4*)(gp + ((long)row * ld) * 2 + ch * 16) = *(const u32x4*)(img + row * ARS + ch * 16);
  }
}

// S = scale * Q K^T for this wave's 32 query rows, masked softmax over the t valid columns -> P (f32, C layout)
__device__ __forceinline__ void scores_softmax(const char* Qi, const char* Ki, int i0, int t, float scale, int lane, f32x16 (&P)[3]) {
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) P[jt][r] = 0.f;
  // (k-step outer, column tile inner: three independent accumulation chains in flight instead of six dependent MFMAs in a row)
#pragma unroll
  for (int kk = 0; kk < 6; ++kk) {
    const u32x4 qa = frag_cols(Qi, i0, kk, lane);
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) mma(P[jt], qa, frag_cols(Ki, jt * 32, kk, lane));
  }
  const int c = lane & 31;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float s[3], m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      s[jt] = (jt * 32 + c < t) ? P[jt][g] * scale : -INFINITY;
      m = fmaxf(m, s[jt]);
    }
    m = half_max(m);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) { s[jt] = __expf(s[jt] - m); sum += s[jt]; }
    sum = half_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) P[jt][g] = s[jt] * inv;
  }
}

// the same with Q and K fragments already in registers (loaded from global: an operand that is only ever contracted along
// its columns needs no LDS image)
__device__ __forceinline__ void scores_softmax_regs(const u32x4 (&qf)[6], const u32x4 (&kf)[3][6], int t, float scale, int lane, f32x16 (&P)[3]) {
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) P[jt][r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 6; ++kk)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) mma(P[jt], qf[kk], kf[jt][kk]);
  const int c = lane & 31;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float s[3], m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      s[jt] = (jt * 32 + c < t) ? P[jt][g] * scale : -INFINITY;
      m = fmaxf(m, s[jt]);
    }
    m = half_max(m);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) { s[jt] = __expf(s[jt] - m); sum += s[jt]; }
    sum = half_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) P[jt][g] = s[jt] * inv;
  }
}

__global__ __launch_bounds__(192) void k_attn_small_fwd(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * IMG];      // V (read along its rows for P V) and P: 39 KiB -> 4 workgroups per CU
  char *Vi = lds, *Pi = lds + IMG;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int s = blockIdx.x / p.heads, h = blockIdx.x - s * p.heads;
  const long row0 = (long)s * p.t;
  const int i0 = 32 * w;
  const char* qg = p.q + (row0 * p.ldq + h * AT) * 2;
  const char* kg = p.k + (row0 * p.ldk + h * AT) * 2;
  u32x4 qf[6], kf[3][6];
  {
    // V's six chunks in one round trip, written to LDS before the Q / K fragments are requested: with all 30 loads in flight at once
    // the kernel needed 200 registers (24 + 96 of data, 60 of addresses) and lost two of its four workgroups per CU (89 -> 100 us
    // per launch); as two round trips it keeps them
    u32x4 sv[6];
    stage_load(sv, p.v + (row0 * p.ldv + h * AT) * 2, p.t, p.ldv, tid);
    stage_store(Vi, sv, tid);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int kk = 0; kk < 6; ++kk) qf[kk] = gfrag_cols(qg, p.ldq, p.t, i0, kk, lane);
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) kf[jt][kk] = gfrag_cols(kg, p.ldk, p.t, jt * 32, kk, lane);
  f32x16 P[3];
  scores_softmax_regs(qf, kf, p.t, p.scale, lane, P);
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) put_tile(Pi, i0, jt * 32, P[jt], lane);   // rows of this wave only: in-order DS, no barrier
  __syncthreads();                                                         // V image complete
  char* op = p.o + (row0 * p.ldo + h * AT) * 2;
  f32x16 acc[3];
#pragma unroll
  for (int ct = 0; ct < 3; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 6; ++kk) {
    const u32x4 pa = frag_cols(Pi, i0, kk, lane);
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) mma(acc[ct], pa, frag_rows(Vi, ct * 32, kk, lane));
  }
  // O goes out through this wave's own rows of the P image (no longer needed): whole 16-byte chunks per lane
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) put_tile(Pi, i0, ct * 32, acc[ct], lane);
  rows_out(Pi, op, p.ldo, p.t, i0, lane);
}

__global__ __launch_bounds__(192, 2) void k_attn_small_bwd(AttnArgs p) {
  // Images: Q, K, dO (each also contracted along its rows) and ONE image that holds P for dV and then dS for dQ / dK; V is
  // only ever contracted along its columns (dP = dO V^T) and stays in registers.  78 KiB -> 2 workgroups per CU.
  __shared__ __attribute__((aligned(16))) char lds[4 * IMG];
  char *Qi = lds, *Ki = lds + IMG, *Di = lds + 2 * IMG, *Xi = lds + 3 * IMG;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int s = blockIdx.x / p.heads, h = blockIdx.x - s * p.heads;
  const long row0 = (long)s * p.t;
  const int i0 = 32 * w;
  const char* vg = p.v + (row0 * p.ldv + h * AT) * 2;
  u32x4 vf[3][6];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) vf[jt][kk] = gfrag_cols(vg, p.ldv, p.t, jt * 32, kk, lane);
  {
    u32x4 sq[6], sk[6], sd[6];
    stage_load(sq, p.q + (row0 * p.ldq + h * AT) * 2, p.t, p.ldq, tid);
    stage_load(sk, p.k + (row0 * p.ldk + h * AT) * 2, p.t, p.ldk, tid);
    stage_load(sd, p.dout + (row0 * p.ldo + h * AT) * 2, p.t, p.ldo, tid);
    stage_store(Qi, sq, tid);
    stage_store(Ki, sk, tid);
    stage_store(Di, sd, tid);
  }
  __syncthreads();
  f32x16 P[3], dP[3];
  scores_softmax(Qi, Ki, i0, p.t, p.scale, lane, P);
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dP[jt][r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 6; ++kk) {
    const u32x4 da = frag_cols(Di, i0, kk, lane);
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) mma(dP[jt], da, vf[jt][kk]);
  }
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float rs = 0.f;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) rs += dP[jt][g] * P[jt][g];
    rs = half_sum(rs);
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) dP[jt][g] = P[jt][g] * (dP[jt][g] - rs) * p.scale;    // dS (masked columns: P = 0)
  }
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) put_tile(Xi, i0, jt * 32, P[jt], lane);
  __syncthreads();                                  // dV needs every wave's rows of P
  char* dqp = p.dq + (row0 * p.ldq + h * AT) * 2;
  char* dkp = p.dk + (row0 * p.ldk + h * AT) * 2;
  char* dvp = p.dv + (row0 * p.ldv + h * AT) * 2;
  f32x16 av[3];
#pragma unroll
  for (int ct = 0; ct < 3; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) av[ct][r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 6; ++kk) {
    const u32x4 pa = frag_rows(Xi, i0, kk, lane);
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) mma(av[ct], pa, frag_rows(Di, ct * 32, kk, lane));   // dV[j] = sum_i P[i][j] dO[i]   (j0 = i0)
  }
  __syncthreads();                                  // every wave is done reading P (the image now takes dS) and dO
  // dV leaves right away through the dO image (dead from here on; own rows): its 48 accumulators are free before dQ / dK
  // need theirs -- 96 instead of 144 accumulation registers, which is what lets two workgroups share a CU
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) put_tile(Di, i0, ct * 32, av[ct], lane);
  rows_out(Di, dvp, p.ldv, p.t, i0, lane);
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) put_tile(Xi, i0, jt * 32, dP[jt], lane);
  __syncthreads();
  f32x16 aq[3], ak[3];
#pragma unroll
  for (int ct = 0; ct < 3; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { aq[ct][r] = 0.f; ak[ct][r] = 0.f; }
#pragma unroll
  for (int kk = 0; kk < 6; ++kk) {
    const u32x4 sc_ = frag_cols(Xi, i0, kk, lane), sr_ = frag_rows(Xi, i0, kk, lane);    // (read once per k-step, not once per column tile)
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      mma(aq[ct], sc_, frag_rows(Ki, ct * 32, kk, lane));     // dQ[i] = sum_j dS[i][j] K[j]
      mma(ak[ct], sr_, frag_rows(Qi, ct * 32, kk, lane));     // dK[j] = sum_i dS[i][j] Q[i]
    }
  }
  __syncthreads();                                  // all reads of Q, K, dO done: their images carry the results out (own rows)
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    put_tile(Qi, i0, ct * 32, aq[ct], lane);
    put_tile(Ki, i0, ct * 32, ak[ct], lane);
  }
  rows_out(Qi, dqp, p.ldq, p.t, i0, lane);
  rows_out(Ki, dkp, p.ldk, p.t, i0, lane);
}

int check(const AttnArgs& a, int dh, int dtype) {
  if (dtype != 0 || dh != AT) return CDDMSL_ERR_ARG;                 // bf16, head dim 96 (the mapper's geometry)
  if (a.nseq < 0 || a.t <= 0 || a.t > AT || a.heads <= 0) return CDDMSL_ERR_ARG;
  if ((a.ldq | a.ldk | a.ldv | a.ldo) & 7) return CDDMSL_ERR_ARG;   // 16-byte rows
  if ((long)a.nseq * a.heads > 0x7fffffffL) return CDDMSL_ERR_ARG;
  return CDDMSL_OK;
}

}  // namespace

extern "C" int cddmsl_attn_small_fwd(const void* q, const void* k, const void* v, void* o, int nseq, int t, int heads, int dh,
                                     int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream) {
  AttnArgs a{(const char*)q, (const char*)k, (const char*)v, nullptr, (char*)o, nullptr, nullptr, nullptr,
             nseq, t, heads, ldq, ldk, ldv, ldo, scale};
  if (int e = check(a, dh, dtype)) return e;
  if (nseq == 0) return CDDMSL_OK;
  hipLaunchKernelGGL(k_attn_small_fwd, dim3((unsigned)(nseq * heads)), dim3(192), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int cddmsl_attn_small_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, void* dk, void* dv,
                                     int nseq, int t, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                     int dtype, void* stream) {
  AttnArgs a{(const char*)q, (const char*)k, (const char*)v, (const char*)dout, nullptr, (char*)dq, (char*)dk, (char*)dv,
             nseq, t, heads, ldq, ldk, ldv, ldo, scale};
  if (int e = check(a, dh, dtype)) return e;
  if (nseq == 0) return CDDMSL_OK;
  hipLaunchKernelGGL(k_attn_small_bwd, dim3((unsigned)(nseq * heads)), dim3(192), 0, (hipStream_t)stream, a);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------------
// The mapper's LAST layer for the last token only (``v2l`` keeps one of the 80 mapped tokens, clipcap.py:714-719; the layer is
// clipcap.py:59-83 with a single query row): per (sequence, head)  p = softmax(q . K^T * scale) over the t keys, o = p . V.
// One wave per (sequence, head); f32 arithmetic on bf16 operands.  q [n][ldq], kv [n*t][ldkv] with K of head h at column h*dh and
// V at column voff + h*dh (the to_keys_values projection as it comes), o [n][ldo]; p [n][heads][t] f32 is kept for the backward:
//     dp = do . V^T,  ds = p o (dp - sum p dp) * scale,  dq = ds . K,  dK[j] = ds[j] q,  dV[j] = p[j] do.
namespace {
constexpr int ALT = 128;       // most keys per sequence
constexpr int ALD = 128;       // largest head dim

struct LastArgs {
  const char *q, *kv, *dout;
  float* p;
  char *o, *dq, *dkv;
  int n, t, heads, dh, ldq, ldkv, voff, ldo;
  float scale;
};

__device__ __forceinline__ float ldbf(const char* base, long idx) { return bf2f(((const unsigned short*)base)[idx]); }

__global__ __launch_bounds__(64) void k_attn_last_fwd(LastArgs a) {
  __shared__ float sq[ALD], sp[ALT];
  const int lane = threadIdx.x, s = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const char* kvb = a.kv + ((long)s * a.t * a.ldkv + h * a.dh) * 2;
  for (int c = lane; c < a.dh; c += 64) sq[c] = ldbf(a.q, (long)s * a.ldq + h * a.dh + c);
  __syncthreads();
  float sc[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = lane + 64 * u;
    if (j < a.t) {
      const u32x4* row = (const u32x4*)(kvb + (long)j * a.ldkv * 2);
      float acc = 0.f;
      for (int c8 = 0; c8 < a.dh / 8; ++c8) {
        const u32x4 v = row[c8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = __builtin_fmaf(sq[c8 * 8 + 2 * e], bf2f(v[e] & 0xffff), acc);
          acc = __builtin_fmaf(sq[c8 * 8 + 2 * e + 1], bf2f(v[e] >> 16), acc);
        }
      }
      sc[u] = acc * a.scale;
    }
  }
  const float m = wave_max(fmaxf(sc[0], sc[1]));
  float e0 = (lane < a.t) ? __expf(sc[0] - m) : 0.f, e1 = (lane + 64 < a.t) ? __expf(sc[1] - m) : 0.f;
  const float inv = 1.0f / wave_sum(e0 + e1);
  e0 *= inv; e1 *= inv;
  float* pp = a.p + ((long)s * a.heads + h) * a.t;
  if (lane < a.t) { sp[lane] = e0; pp[lane] = e0; }
  if (lane + 64 < a.t) { sp[lane + 64] = e1; pp[lane + 64] = e1; }
  __syncthreads();
  if (2 * lane < a.dh) {
    const char* vb = kvb + a.voff * 2 + lane * 4;
    float o0 = 0.f, o1 = 0.f;
    for (int j = 0; j < a.t; ++j) {
      const unsigned v = *(const unsigned*)(vb + (long)j * a.ldkv * 2);
      o0 = __builtin_fmaf(sp[j], bf2f(v & 0xffff), o0);
      o1 = __builtin_fmaf(sp[j], bf2f(v >> 16), o1);
    }
    *(unsigned*)(a.o + ((long)s * a.ldo + h * a.dh + 2 * lane) * 2) = pack2bf(o0, o1);
  }
}

__global__ __launch_bounds__(64) void k_attn_last_bwd(LastArgs a) {
  __shared__ float sq[ALD], sdo[ALD], sp[ALT], sds[ALT];
  const int lane = threadIdx.x, s = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const char* kvb = a.kv + ((long)s * a.t * a.ldkv + h * a.dh) * 2;
  char* dkvb = a.dkv + ((long)s * a.t * a.ldkv + h * a.dh) * 2;
  const float* pp = a.p + ((long)s * a.heads + h) * a.t;
  for (int c = lane; c < a.dh; c += 64) {
    sq[c] = ldbf(a.q, (long)s * a.ldq + h * a.dh + c);
    sdo[c] = ldbf(a.dout, (long)s * a.ldo + h * a.dh + c);
  }
  for (int j = lane; j < a.t; j += 64) sp[j] = pp[j];
  __syncthreads();
  float dp[2] = {0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = lane + 64 * u;
    if (j < a.t) {
      const u32x4* row = (const u32x4*)(kvb + a.voff * 2 + (long)j * a.ldkv * 2);
      float acc = 0.f;
      for (int c8 = 0; c8 < a.dh / 8; ++c8) {
        const u32x4 v = row[c8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = __builtin_fmaf(sdo[c8 * 8 + 2 * e], bf2f(v[e] & 0xffff), acc);
          acc = __builtin_fmaf(sdo[c8 * 8 + 2 * e + 1], bf2f(v[e] >> 16), acc);
        }
      }
      dp[u] = acc;
    }
  }
  const float p0 = lane < a.t ? sp[lane] : 0.f, p1 = lane + 64 < a.t ? sp[lane + 64] : 0.f;
  const float dot = wave_sum(p0 * dp[0] + p1 * dp[1]);
  if (lane < a.t) sds[lane] = p0 * (dp[0] - dot) * a.scale;
  if (lane + 64 < a.t) sds[lane + 64] = p1 * (dp[1] - dot) * a.scale;
  __syncthreads();
  if (2 * lane < a.dh) {
    const float q0 = sq[2 * lane], q1 = sq[2 * lane + 1], g0 = sdo[2 * lane], g1 = sdo[2 * lane + 1];
    float dq0 = 0.f, dq1 = 0.f;
    for (int j = 0; j < a.t; ++j) {
      const long ro = (long)j * a.ldkv * 2 + lane * 4;
      const unsigned kk = *(const unsigned*)(kvb + ro);
      const float ds = sds[j], pj = sp[j];
      dq0 = __builtin_fmaf(ds, bf2f(kk & 0xffff), dq0);
      dq1 = __builtin_fmaf(ds, bf2f(kk >> 16), dq1);
      *(unsigned*)(dkvb + ro) = pack2bf(ds * q0, ds * q1);
      *(unsigned*)(dkvb + a.voff * 2 + ro) = pack2bf(pj * g0, pj * g1);
    }
    *(unsigned*)(a.dq + ((long)s * a.ldq + h * a.dh + 2 * lane) * 2) = pack2bf(dq0, dq1);
  }
}

int check_last(const LastArgs& a, int dtype) {
  if (dtype != 0 || a.n < 0 || a.t <= 0 || a.t > ALT || a.heads <= 0 || a.dh <= 0 || a.dh > ALD || (a.dh & 7)) return CDDMSL_ERR_ARG;
  if ((a.ldq | a.ldkv | a.voff | a.ldo) & 7) return CDDMSL_ERR_ARG;      // 16-byte rows / column blocks
  if ((long)a.n * a.heads > 0x7fffffffL) return CDDMSL_ERR_ARG;
  return CDDMSL_OK;
}
}  // namespace

extern "C" int cddmsl_attn_last_fwd(const void* q, const void* kv, void* o, float* p, int n, int t, int heads, int dh, int ldq, int ldkv,
                                    int voff, int ldo, float scale, int dtype, void* stream) {
  LastArgs a{(const char*)q, (const char*)kv, nullptr, p, (char*)o, nullptr, nullptr, n, t, heads, dh, ldq, ldkv, voff, ldo, scale};
  if (int e = check_last(a, dtype)) return e;
  if (n == 0) return CDDMSL_OK;
  hipLaunchKernelGGL(k_attn_last_fwd, dim3((unsigned)(n * heads)), dim3(64), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int cddmsl_attn_last_bwd(const void* q, const void* kv, const void* dout, const float* p, void* dq, void* dkv, int n, int t,
                                    int heads, int dh, int ldq, int ldkv, int voff, int ldo, float scale, int dtype, void* stream) {
  LastArgs a{(const char*)q, (const char*)kv, (const char*)dout, (float*)p, nullptr, (char*)dq, (char*)dkv, n, t, heads, dh, ldq, ldkv, voff, ldo, scale};
  if (int e = check_last(a, dtype)) return e;
  if (n == 0) return CDDMSL_OK;
  hipLaunchKernelGGL(k_attn_last_bwd, dim3((unsigned)(n * heads)), dim3(64), 0, (hipStream_t)stream, a);
  return launch_status();
}
