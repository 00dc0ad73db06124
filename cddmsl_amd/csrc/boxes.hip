// Integer/index stages of the RPN and ROI heads on gfx950: anchor decode, sort/top-k, bitmask NMS,
// fused IoU + matcher.  These are HBM/latency-bound byte and index kernels (no MFMA).
//
// Reference call sites replaced (paths under detectron2/):
//   DefaultAnchorGenerator + Box2BoxTransform.apply_deltas + Boxes.clip/nonempty
//        modeling/anchor_generator.py:161-228, modeling/box_regression.py:77-115,
//        modeling/proposal_generator/rpn.py:514-533, structures/boxes.py:193-225
//   find_top_rpn_proposals (sort desc, top-k, NMS, top-k)   modeling/proposal_generator/proposal_utils.py:22-130
//   batched_nms -> torchvision.ops.nms                      layers/nms.py:19-39
//   pairwise_iou + Matcher (+ low-quality matches)          structures/boxes.py:322-367, modeling/matcher.py:61-126
// Bit-exactness: IoU and NMS use only f32 add/sub/mul/div in the reference's association order with
// contraction off (no a*b+c patterns exist), IEEE division; indices are therefore identical to the CPU path.
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

// ---------------------------------------------------------------- decode sorted top-k anchors
// For image n and rank r < topk: a = order[n][r] (index into the H*W*A anchor list, hw-major, a-minor),
// box = apply_deltas(deltas[n][a], anchor(a)), clipped to (img_h, img_w); valid = finite && nonempty.
__global__ void k_rpn_decode(const int* order, const float* deltas, const float* cell, const int* img_hw, float* boxes,
                             unsigned char* valid, int N, int total, int topk, int A, int Wf, float stride, float offset,
                             float wx, float wy, float ww, float wh, float clampv, float min_size) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * topk) return;
  int n = i / topk;
  int a = order[(long)n * total + (i - (long)n * topk)];
  int ca = a % A, loc = a / A;
  float sx = offset * stride + (float)(loc % Wf) * stride, sy = offset * stride + (float)(loc / Wf) * stride;
  float ax0 = sx + cell[4 * ca], ay0 = sy + cell[4 * ca + 1], ax1 = sx + cell[4 * ca + 2], ay1 = sy + cell[4 * ca + 3];
  const float* d = deltas + ((long)n * total + a) * 4;
  float w = ax1 - ax0, h = ay1 - ay0;
  float cx = ax0 + 0.5f * w, cy = ay0 + 0.5f * h;
  float dx = d[0] / wx, dy = d[1] / wy, dw = fminf(d[2] / ww, clampv), dh = fminf(d[3] / wh, clampv);
  float pcx = dx * w + cx, pcy = dy * h + cy;
  float pw = expf(dw) * w, phh = expf(dh) * h;
  float x0 = pcx - 0.5f * pw, y0 = pcy - 0.5f * phh, x1 = pcx + 0.5f * pw, y1 = pcy + 0.5f * phh;
  bool fin = isfinite(x0) && isfinite(y0) && isfinite(x1) && isfinite(y1);
  float ih = (float)img_hw[2 * n], iw = (float)img_hw[2 * n + 1];
  x0 = fminf(fmaxf(x0, 0.f), iw); x1 = fminf(fmaxf(x1, 0.f), iw);
  y0 = fminf(fmaxf(y0, 0.f), ih); y1 = fminf(fmaxf(y1, 0.f), ih);
  float* o = boxes + i * 4;
  o[0] = x0; o[1] = y0; o[2] = x1; o[3] = y1;
  valid[i] = fin ? (((x1 - x0) > min_size && (y1 - y0) > min_size) ? 1 : 0) : 2;  // 2 = non-finite
}

// all anchors decoded (no ordering): used for parity checks and by the matcher path
__global__ void k_anchors(const float* cell, float* out, int total, int A, int Wf, float stride, float offset) {
  int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= total) return;
  int ca = a % A, loc = a / A;
  float sx = offset * stride + (float)(loc % Wf) * stride, sy = offset * stride + (float)(loc / Wf) * stride;
  out[4 * a] = sx + cell[4 * ca]; out[4 * a + 1] = sy + cell[4 * ca + 1];
  out[4 * a + 2] = sx + cell[4 * ca + 2]; out[4 * a + 3] = sy + cell[4 * ca + 3];
}


// ---------------------------------------------------------------- bitmask NMS
// boxes [N][n][4] in descending-score order; valid [N][n].  mask[N][n][nw] (nw = ceil(n/64)) bit j of word w of
// row i set when j = 64w+bit > i and IoU(i,j) > thr.  64-wide wavefront <-> 64-bit mask word.
__global__ void k_nms_mask(const float* boxes, unsigned long long* mask, int n, int nw, float thr) {
  int img = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;  // only j > i matters
  const float* b = boxes + (long)img * n * 4;
  __shared__ float cbx[64 * 4], car[64];
  int t = threadIdx.x;
  int cj = cb * 64 + t;
  if (cj < n) {
    cbx[4 * t] = b[4 * cj]; cbx[4 * t + 1] = b[4 * cj + 1]; cbx[4 * t + 2] = b[4 * cj + 2]; cbx[4 * t + 3] = b[4 * cj + 3];
    car[t] = (b[4 * cj + 2] - b[4 * cj]) * (b[4 * cj + 3] - b[4 * cj + 1]);
  }
  __syncthreads();
  int i = rb * 64 + t;
  if (i >= n) return;
  float x0 = b[4 * i], y0 = b[4 * i + 1], x1 = b[4 * i + 2], y1 = b[4 * i + 3];
  float ai = (x1 - x0) * (y1 - y0);
  unsigned long long bits = 0;
  int jn = min(64, n - cb * 64);
  for (int j = (rb == cb ? t + 1 : 0); j < jn; ++j) {
    float xx1 = fmaxf(x0, cbx[4 * j]), yy1 = fmaxf(y0, cbx[4 * j + 1]);
    float xx2 = fminf(x1, cbx[4 * j + 2]), yy2 = fminf(y1, cbx[4 * j + 3]);
    float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    float inter = w * h;
    float u = ai + car[j] - inter;
    // inter / u > thr decided without the division: r = inter - thr*u is exact in sign (one fma).  The rounded quotient can
    // only disagree with the real one when the real quotient lies within half an ulp above thr, i.e. 0 < r < ~6e-8 u: those
    // (rare) lanes take the division, so the bit is the one `inter / u > thr` gives in every case.
    float r = fmaf(-thr, u, inter);
    bool over = r > 0.f && u > 0.f;                 // (u <= 0: degenerate boxes, the quotient is NaN or <= 0)
    if (over && r < 2.5e-7f * u) over = inter / u > thr;
    if (over) bits |= 1ull << j;
  }
  mask[((long)img * n + i) * nw + cb] = bits;
}
// One workgroup per image.  The dependent chain of greedy NMS is resolved 64 candidates (one mask word) at a time, and only the
// part that really is sequential runs sequentially: wave 0 takes the chunk's 64 DIAGONAL mask words (lane b = candidate b,
// bits = the later candidates of the same chunk it suppresses) and walks the 64 candidates with scalar operations on one
// 64-bit "removed" word (v_readlane for a kept candidate's diagonal word), which yields the chunk's keep mask; then all four
// waves OR the mask rows of the KEPT candidates into the removed words of the later chunks, straight from global memory
// (one thread per word, the rows of a word's column are independent loads).  The first version staged all 64 rows of a chunk
// in LDS (96 KiB) and updated every later word inside the sequential walk: 1.75 ms for 16 images x 12000 candidates.
constexpr int NMS_MAXW = 192;   // mask words per row: up to 12288 candidates (pre-NMS top-k is 12000)
__global__ __launch_bounds__(256) void k_nms_scan(const unsigned long long* mask, const unsigned char* valid, int* keep,
                                                   int* nkeep, int n, int nw, int max_keep) {
  __shared__ unsigned long long rem[NMS_MAXW];
  __shared__ unsigned long long s_km;
  __shared__ int s_cnt;
  const int img = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const unsigned long long* m = mask + (long)img * n * nw;
  const unsigned char* v = valid + (long)img * n;
  for (int ww = t; ww < nw; ww += 256) {          // candidates that are not valid (or do not exist) start out removed
    unsigned long long bits = 0;
    for (int b = 0; b < 64; ++b) {
      const int i = ww * 64 + b;
      if (i >= n || v[i] != 1) bits |= 1ull << b;
    }
    rem[ww] = bits;
  }
  if (t == 0) { s_cnt = 0; s_km = 0; }
  int cnt = 0;                                    // (kept so far; wave 0's copy is the authoritative one)
  unsigned long long dnext = (wv == 0 && lane < n) ? m[(long)lane * nw] : 0ull;
  for (int c = 0; c < nw; ++c) {
    __syncthreads();                              // rem[c] is final: every earlier chunk has been applied
    if (wv == 0) {
      const int i = c * 64 + lane;
      const unsigned long long d = dnext;         // this chunk's diagonal word, requested one chunk ago (it depends on no decision)
      dnext = (c + 1 < nw && i + 64 < n) ? m[(long)(i + 64) * nw + c + 1] : 0ull;
      const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
      // the chunk's "removed" word in scalar registers; the walk jumps from one surviving candidate to the next (s_ff1 on the word's
      // complement) instead of testing all 64 bit positions in turn -- as many iterations as candidates kept, ~10 per chunk, not 64
      const unsigned long long r0 = rem[c];
      unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(r0 >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)r0);
      unsigned long long km = 0, todo = ~0ull;
      const int cnt0 = cnt;
      while (cnt < max_keep) {
        const unsigned long long a = ~cur & todo;
        if (!a) break;
        const int b = __builtin_ctzll(a);
        todo = b == 63 ? 0ull : (~0ull << (b + 1));
        km |= 1ull << b;
        ++cnt;
        cur |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, b) << 32) | (unsigned)__builtin_amdgcn_readlane(dlo, b);
      }
      if ((km >> lane) & 1ull) keep[(long)img * max_keep + cnt0 + __builtin_popcountll(km & ((1ull << lane) - 1ull))] = i;
      if (lane == 0) { s_km = km; s_cnt = cnt; }
    }
    __syncthreads();
    const unsigned long long kmv = s_km;
    if (s_cnt >= max_keep) break;
    // the kept rows of this chunk, OR-ed into the words behind it: the row list is block-uniform (scalar registers), and the
    // rows are requested sixteen at a time -- one memory round trip per sixteen kept rows instead of one per row (a repeated
    // row pads the last group: OR is idempotent)
    unsigned long long km = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kmv >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kmv);
    if (km) {
      for (int ww = c + 1 + t; ww < nw; ww += 256) {
        unsigned long long acc = rem[ww], bits = km;
        const unsigned long long* col = m + (long)c * 64 * nw + ww;
        while (bits) {
          int b[16];
          b[0] = __builtin_ctzll(bits);
          bits &= bits - 1;
#pragma unroll
          for (int u = 1; u < 16; ++u) {
            b[u] = bits ? __builtin_ctzll(bits) : b[0];
            bits &= bits - 1;                     // (0 & anything stays 0)
          }
          unsigned long long r[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) r[u] = col[(long)b[u] * nw];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc |= r[u];
        }
        rem[ww] = acc;
      }
    }
  }
  __syncthreads();
  if (t == 0) nkeep[img] = s_cnt;
}

// ---------------------------------------------------------------- fused IoU + Matcher
__device__ __forceinline__ float iou_pair(const float* g, float x0, float y0, float x1, float y1, float ap) {
  float w = fminf(g[2], x1) - fmaxf(g[0], x0);
  float h = fminf(g[3], y1) - fmaxf(g[1], y0);
  w = fmaxf(w, 0.f); h = fmaxf(h, 0.f);
  float inter = w * h;
  float ag = (g[2] - g[0]) * (g[3] - g[1]);
  return inter > 0.f ? inter / (ag + ap - inter) : 0.f;
}
// pass 1: per prediction j: max/argmax over gt (first max), threshold label; per gt: atomic max over j.
// gt [G][4] for this image, preds [P][4].  thresholds: up to 2 cut points t0 <= t1 with labels l0,l1,l2.
__global__ void k_match1(const float* gt, int G, const float* preds, int P, long* matches, signed char* labels,
                         unsigned int* best_gt, int nthr, float t0, float t1, int l0, int l1, int l2) {
  extern __shared__ float sg[];
  for (int i = threadIdx.x; i < G * 4; i += blockDim.x) sg[i] = gt[i];
  __syncthreads();
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = j < P;
  int jj = live ? j : 0;
  float x0 = preds[4 * jj], y0 = preds[4 * jj + 1], x1 = preds[4 * jj + 2], y1 = preds[4 * jj + 3];
  float ap = (x1 - x0) * (y1 - y0);
  float best = -1.f; int arg = 0;
  for (int g = 0; g < G; ++g) {
    float v = iou_pair(sg + 4 * g, x0, y0, x1, y1, ap);
    if (v > best) { best = v; arg = g; }
    if (best_gt) {   // per-gt max over predictions: wave max first, one atomic per wave (v >= 0: uint order == float order)
      float wm = wave_max(live ? v : 0.f);
      if ((threadIdx.x & 63) == 0 && wm > 0.f) atomicMax(best_gt + g, __float_as_uint(wm));    // (0 = the initial value)
    }
  }
  if (!live) return;
  int lab;
  if (nthr == 1) lab = best < t0 ? l0 : l1;
  else lab = best < t0 ? l0 : (best < t1 ? l1 : l2);
  matches[j] = arg;
  labels[j] = (signed char)lab;
}
// pass 2 (allow_low_quality_matches): label 1 for every prediction attaining some gt's row max (ties included)
__global__ void k_match2(const float* gt, int G, const float* preds, int P, signed char* labels, const unsigned int* best_gt) {
  extern __shared__ float sg[];
  for (int i = threadIdx.x; i < G * 4; i += blockDim.x) sg[i] = gt[i];
  __syncthreads();
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  float x0 = preds[4 * j], y0 = preds[4 * j + 1], x1 = preds[4 * j + 2], y1 = preds[4 * j + 3];
  float ap = (x1 - x0) * (y1 - y0);
  bool hit = false;
  for (int g = 0; g < G; ++g) {
    float v = iou_pair(sg + 4 * g, x0, y0, x1, y1, ap);
    hit |= (__float_as_uint(v) == best_gt[g]);
  }
  if (hit) labels[j] = 1;
}

constexpr int MATCH_PT = 4;   // predictions per thread of k_match1_batched
// batched over images (blockIdx.y = image): gt rows gt_off[n]..gt_off[n+1]; predictions either shared by all images
// (pred_off == nullptr: the anchors; outputs laid out [N][P]) or concatenated with offsets pred_off (proposals)
__global__ void k_match1_batched(const float* gt, const int* gt_off, const float* preds, const int* pred_off, int P, long* matches,
                                 signed char* labels, unsigned int* best_gt, int nthr, float t0, float t1, int l0, int l1, int l2) {
  // A block takes MATCH_PT consecutive runs of blockDim predictions; a box's row maximum leaves the block as ONE atomic (per-thread
  // maximum over its predictions, wave maximum, LDS across the waves).  Same-address atomics serialise in L2: with one per wave
  // and box the launch over 16 x 62 250 anchors x 3 boxes took 170 us for ~50 000 atomics on 48 words; now ~3 000.
  extern __shared__ float sg[];
  __shared__ float wmax[4];
  const int n = blockIdx.y, g0 = gt_off[n], G = gt_off[n + 1] - g0;
  const int p0 = pred_off ? pred_off[n] : 0, Pn = pred_off ? pred_off[n + 1] - p0 : P;
  const long o0 = pred_off ? p0 : (long)n * P;
  const int base = blockIdx.x * blockDim.x * MATCH_PT;
  if (base >= Pn) return;                                    // whole block past this image's predictions (uniform)
  for (int i = threadIdx.x; i < G * 4; i += blockDim.x) sg[i] = gt[4 * (long)g0 + i];
  __syncthreads();
  float x0[MATCH_PT], y0[MATCH_PT], x1[MATCH_PT], y1[MATCH_PT], ap[MATCH_PT], best[MATCH_PT];
  int arg[MATCH_PT];
  bool live[MATCH_PT];
#pragma unroll
  for (int u = 0; u < MATCH_PT; ++u) {
    const int j = base + u * blockDim.x + threadIdx.x;
    live[u] = j < Pn;
    const long jj = live[u] ? p0 + j : p0;
    x0[u] = preds[4 * jj]; y0[u] = preds[4 * jj + 1]; x1[u] = preds[4 * jj + 2]; y1[u] = preds[4 * jj + 3];
    ap[u] = (x1[u] - x0[u]) * (y1[u] - y0[u]);
    best[u] = -1.f; arg[u] = 0;
  }
  for (int g = 0; g < G; ++g) {
    float tm = 0.f;
#pragma unroll
    for (int u = 0; u < MATCH_PT; ++u) {
      const float v = iou_pair(sg + 4 * g, x0[u], y0[u], x1[u], y1[u], ap[u]);
      if (v > best[u]) { best[u] = v; arg[u] = g; }
      tm = fmaxf(tm, live[u] ? v : 0.f);
    }
    if (best_gt) {                                           // (block-uniform)
      tm = wave_max(tm);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = tm;
      __syncthreads();
      if (threadIdx.x == 0) {
        float m = wmax[0];
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) m = fmaxf(m, wmax[wv]);
        if (m > 0.f) atomicMax(best_gt + g0 + g, __float_as_uint(m));      // (v >= 0: uint order == float order; 0 is the initial value)
      }
    }
  }
#pragma unroll
  for (int u = 0; u < MATCH_PT; ++u) {
    if (!live[u]) continue;
    const int j = base + u * blockDim.x + threadIdx.x;
    int lab;
    if (nthr == 1) lab = best[u] < t0 ? l0 : l1;
    else lab = best[u] < t0 ? l0 : (best[u] < t1 ? l1 : l2);
    matches[o0 + j] = arg[u];
    labels[o0 + j] = (signed char)lab;
  }
}
__global__ void k_match2_batched(const float* gt, const int* gt_off, const float* preds, const int* pred_off, int P,
                                 signed char* labels, const unsigned int* best_gt) {
  extern __shared__ float sg[];
  const int n = blockIdx.y, g0 = gt_off[n], G = gt_off[n + 1] - g0;
  const int p0 = pred_off ? pred_off[n] : 0, Pn = pred_off ? pred_off[n + 1] - p0 : P;
  const long o0 = pred_off ? p0 : (long)n * P;
  if (blockIdx.x * blockDim.x >= Pn) return;
  for (int i = threadIdx.x; i < G * 4; i += blockDim.x) sg[i] = gt[4 * (long)g0 + i];
  __syncthreads();
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Pn) return;
  long jj = p0 + j;
  float x0 = preds[4 * jj], y0 = preds[4 * jj + 1], x1 = preds[4 * jj + 2], y1 = preds[4 * jj + 3];
  float ap = (x1 - x0) * (y1 - y0);
  bool hit = false;
  for (int g = 0; g < G; ++g) {
    float v = iou_pair(sg + 4 * g, x0, y0, x1, y1, ap);
    hit |= (__float_as_uint(v) == best_gt[g0 + g]);
  }
  if (hit) labels[o0 + j] = 1;
}

}  // namespace

extern "C" int cddmsl_anchors(const float* cell, float* out, int Hf, int Wf, int A, float stride, float offset, void* stream) {
  int total = Hf * Wf * A;
  if (total <= 0) return total == 0 ? CDDMSL_OK : CDDMSL_ERR_ARG;
  k_anchors<<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(cell, out, total, A, Wf, stride, offset);
  return launch_status();
}

// Stable descending sort of every row of logits [N][total] -> sorted keys + order (int32 index within the image).
// ONE device-wide radix sort over 64-bit composite keys (image index in the high word, the float's descending-order bit
// pattern in the low word; radix sort is stable, so equal scores keep the lower index first) instead of rocPRIM's segmented
// sort, which handles 16 long segments with little parallelism (1.4 ms for 16 x 62 250 keys; this: ~0.2 ms).  Only the
// 32 + ceil(log2 N) significant key bits are sorted.  Call with temp == NULL to get the workspace size in *temp_bytes.
__global__ void k_sort_make_keys(const float* keys, unsigned long long* k64, int* idx, int N, int total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * total) return;
  const unsigned img = (unsigned)(i / total);
  const unsigned f = __float_as_uint(keys[i]);
  const unsigned asc = f ^ ((f >> 31) ? 0xffffffffu : 0x80000000u);      // unsigned order == float order
  k64[i] = ((unsigned long long)img << 32) | (unsigned)(~asc);              // ascending composite == (image, descending score)
  idx[i] = (int)(i - (long)img * total);
}
__global__ void k_sort_unmake_keys(const unsigned long long* k64, float* keys, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned asc = ~(unsigned)k64[i];
  keys[i] = __uint_as_float(asc ^ ((asc >> 31) ? 0x80000000u : 0xffffffffu));
}
extern "C" int cddmsl_sort_desc(const float* keys_in, float* keys_out, int* idx_scratch, int* order_out, const int* offsets,
                                int N, int total, void* temp, size_t* temp_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)offsets;                                   // (row offsets of the segmented form; rows are dense [N][total])
  if (N <= 0 || total <= 0 || (long)N * total > 0x7fffffffL) return CDDMSL_ERR_ARG;
  const long n = (long)N * total;
  int end_bit = 32;
  while ((1L << (end_bit - 32)) < N) ++end_bit;
  size_t rp = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, rp, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int*)nullptr,
                                           (int*)nullptr, (size_t)n, 0, (unsigned)end_bit, st);
  if (e != hipSuccess) return CDDMSL_ERR_LAUNCH;
  const size_t rp_al = (rp + 255) & ~(size_t)255, kb = ((size_t)n * 8 + 255) & ~(size_t)255;
  if (!temp) { *temp_bytes = rp_al + 2 * kb; return CDDMSL_OK; }
  if (*temp_bytes < rp_al + 2 * kb) return CDDMSL_ERR_ARG;
  unsigned long long* k_in = (unsigned long long*)((char*)temp + rp_al);
  unsigned long long* k_out = (unsigned long long*)((char*)temp + rp_al + kb);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  k_sort_make_keys<<<dim3(blocks), dim3(256), 0, st>>>(keys_in, k_in, idx_scratch, N, total);
  e = rocprim::radix_sort_pairs(temp, rp, k_in, k_out, idx_scratch, order_out, (size_t)n, 0, (unsigned)end_bit, st);
  if (e != hipSuccess) return CDDMSL_ERR_LAUNCH;
  k_sort_unmake_keys<<<dim3(blocks), dim3(256), 0, st>>>(k_out, keys_out, n);
  return launch_status();
}

extern "C" int cddmsl_rpn_decode(const int* order, const float* deltas, const float* cell, const int* img_hw, float* boxes,
                                 unsigned char* valid, int N, int Hf, int Wf, int A, int topk, float stride, float offset,
                                 float wx, float wy, float ww, float wh, float scale_clamp, float min_size, void* stream) {
  int total = Hf * Wf * A;
  if (N <= 0 || total <= 0 || topk <= 0 || topk > total) return CDDMSL_ERR_ARG;
  long n = (long)N * topk;
  k_rpn_decode<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      order, deltas, cell, img_hw, boxes, valid, N, total, topk, A, Wf, stride, offset, wx, wy, ww, wh, scale_clamp, min_size);
  return launch_status();
}

// boxes [N][n][4] score-descending, valid [N][n] (1 = candidate).  mask_ws: N*n*ceil(n/64) u64.
// keep [N][max_keep] (positions into the sorted list, score order), nkeep [N].
extern "C" int cddmsl_nms(const float* boxes, const unsigned char* valid, unsigned long long* mask_ws, int* keep, int* nkeep,
                          int N, int n, float thr, int max_keep, void* stream) {
  if (N <= 0 || n < 0 || max_keep <= 0 || n > 64 * NMS_MAXW) return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return hipMemsetAsync(nkeep, 0, sizeof(int) * N, st) == hipSuccess ? CDDMSL_OK : CDDMSL_ERR_LAUNCH;
  int nw = (n + 63) / 64;
  k_nms_mask<<<dim3(nw, nw, N), dim3(64), 0, st>>>(boxes, mask_ws, n, nw, thr);
  k_nms_scan<<<dim3(N), dim3(256), 0, st>>>(mask_ws, valid, keep, nkeep, n, nw, max_keep);
  return launch_status();
}

// torchvision.ops.nms(boxes [K,4], scores [K], iou_threshold) -> kept indices (int64, descending score) as the reference calls it
// (layers/nms.py:6-7,30,35): candidates in ANY order.  Composition of the stages above on caller-provided scratch: stable radix
// sort of the scores, gather of the boxes, 64-bit wave-mask NMS, map of the kept positions back to input indices.
// temp == NULL: *temp_bytes receives the scratch size.  keep: K int64 (first *nkeep valid, the rest -1), nkeep: 1 int (device).
__global__ void k_nms_gather(const float* boxes, const int* order, float* sorted, unsigned char* valid, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  const f32x4 b = ((const f32x4*)boxes)[order[i]];
  ((f32x4*)sorted)[i] = b;
  valid[i] = 1;
}
__global__ void k_nms_unmap(const int* keep32, const int* nkeep, const int* order, long* keep, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  keep[i] = i < nkeep[0] ? (long)order[keep32[i]] : -1L;
}
extern "C" int cddmsl_nms_anyorder(const float* boxes, const float* scores, long* keep, int* nkeep, int K, float iou_threshold,
                                   void* temp, size_t* temp_bytes, void* stream) {
  if (K < 0 || K > 64 * NMS_MAXW || !temp_bytes) return CDDMSL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (K == 0) { if (temp) return hipMemsetAsync(nkeep, 0, sizeof(int), st) == hipSuccess ? CDDMSL_OK : CDDMSL_ERR_LAUNCH; *temp_bytes = 0; return CDDMSL_OK; }
  size_t sort_bytes = 0;
  int rc = cddmsl_sort_desc(nullptr, nullptr, nullptr, nullptr, nullptr, 1, K, nullptr, &sort_bytes, stream);
  if (rc != CDDMSL_OK) return rc;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const int nw = (K + 63) / 64;
  const size_t o_keys = al(sort_bytes), o_idx = o_keys + al((size_t)K * 4), o_ord = o_idx + al((size_t)K * 4), o_box = o_ord + al((size_t)K * 4),
               o_val = o_box + al((size_t)K * 16), o_msk = o_val + al((size_t)K), o_k32 = o_msk + al((size_t)K * nw * 8), total = o_k32 + al((size_t)K * 4);
  if (!temp) { *temp_bytes = total; return CDDMSL_OK; }
  if (*temp_bytes < total) return CDDMSL_ERR_ARG;
  char* t = (char*)temp;
  size_t sb = sort_bytes;
  rc = cddmsl_sort_desc(scores, (float*)(t + o_keys), (int*)(t + o_idx), (int*)(t + o_ord), nullptr, 1, K, t, &sb, stream);
  if (rc != CDDMSL_OK) return rc;
  const unsigned blocks = (unsigned)((K + 255) / 256);
  k_nms_gather<<<dim3(blocks), dim3(256), 0, st>>>(boxes, (const int*)(t + o_ord), (float*)(t + o_box), (unsigned char*)(t + o_val), K);
  rc = cddmsl_nms((const float*)(t + o_box), (const unsigned char*)(t + o_val), (unsigned long long*)(t + o_msk), (int*)(t + o_k32), nkeep, 1, K,
                  iou_threshold, K, stream);
  if (rc != CDDMSL_OK) return rc;
  k_nms_unmap<<<dim3(blocks), dim3(256), 0, st>>>((const int*)(t + o_k32), nkeep, (const int*)(t + o_ord), keep, K);
  return launch_status();
}

// One image: gt [G][4], preds [P][4] -> matches int64 [P], labels int8 [P].  best_ws: G uints (zeroed here).
extern "C" int cddmsl_iou_match(const float* gt, int G, const float* preds, int P, long* matches, signed char* labels,
                                unsigned int* best_ws, int nthr, float t0, float t1, int l0, int l1, int l2,
                                int allow_low_quality, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (G < 0 || P < 0 || nthr < 1 || nthr > 2 || G > 4096) return CDDMSL_ERR_ARG;
  if (P == 0) return CDDMSL_OK;
  if (G == 0) {  // matcher.py:75-86: all predictions get labels[0], match 0
    if (hipMemsetAsync(matches, 0, sizeof(long) * P, st) != hipSuccess) return CDDMSL_ERR_LAUNCH;
    if (hipMemsetAsync(labels, l0 & 0xff, P, st) != hipSuccess) return CDDMSL_ERR_LAUNCH;
    return CDDMSL_OK;
  }
  unsigned int* bw = allow_low_quality ? best_ws : nullptr;
  if (bw && hipMemsetAsync(bw, 0, sizeof(unsigned int) * G, st) != hipSuccess) return CDDMSL_ERR_LAUNCH;
  dim3 grid((P + 255) / 256), block(256);
  size_t sh = sizeof(float) * 4 * G;
  k_match1<<<grid, block, sh, st>>>(gt, G, preds, P, matches, labels, bw, nthr, t0, t1, l0, l1, l2);
  if (allow_low_quality) k_match2<<<grid, block, sh, st>>>(gt, G, preds, P, labels, bw);
  return launch_status();
}

// All images in one launch (pair): gt [sum G][4] with offsets gt_off [N+1] (device ints); preds [P][4] shared by all images
// (pred_off == NULL; outputs [N][P]) or [sum P][4] with offsets pred_off [N+1] (outputs [sum P]); P = the largest per-image
// prediction count, maxG the largest per-image box count, totalG = sum G.  An image without boxes gets labels l0, match 0.
extern "C" int cddmsl_iou_match_batched(const float* gt, const int* gt_off, const float* preds, const int* pred_off, long* matches,
                                        signed char* labels, unsigned int* best_ws, int N, int P, int maxG, int totalG, int nthr,
                                        float t0, float t1, int l0, int l1, int l2, int allow_low_quality, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (N < 0 || P < 0 || maxG < 0 || totalG < 0 || nthr < 1 || nthr > 2 || maxG > 4096) return CDDMSL_ERR_ARG;
  if (N == 0 || P == 0) return CDDMSL_OK;
  unsigned int* bw = (allow_low_quality && totalG > 0) ? best_ws : nullptr;
  if (bw && hipMemsetAsync(bw, 0, sizeof(unsigned int) * totalG, st) != hipSuccess) return CDDMSL_ERR_LAUNCH;
  dim3 grid((P + 255) / 256, N), block(256);
  size_t sh = sizeof(float) * 4 * (maxG > 0 ? maxG : 1);
  k_match1_batched<<<dim3((P + 256 * MATCH_PT - 1) / (256 * MATCH_PT), N), block, sh, st>>>(gt, gt_off, preds, pred_off, P, matches, labels, bw, nthr, t0, t1, l0, l1, l2);
  if (bw) k_match2_batched<<<grid, block, sh, st>>>(gt, gt_off, preds, pred_off, P, labels, bw);
  return launch_status();
}
