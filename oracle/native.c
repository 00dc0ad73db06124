/*
 * ORACLE (test infrastructure, CPU only) -- never imported by the product package.
 *
 * C restatement of the third-party arithmetic the reference's hot path calls but does not
 * vendor (torchvision is absent from /root/reference and from this image):
 *   - torchvision.ops.roi_align  forward/backward   (call site: detectron2/layers/roi_align.py:58-65)
 *   - torchvision.ops.nms                           (call site: detectron2/layers/nms.py:19-39)
 * following the published algorithm summarised in SURVEY.md Appendix C, and pinned by the
 * reference's own known-answer tests (tests/layers/test_roi_align.py:14-47,111-128).
 * NMS numeric parity is "unpinned" by reference fixtures (tests/layers/test_nms.py only checks
 * self-consistency); it is pinned indirectly through the proposals test.
 *
 * Layout: input NCHW float32 contiguous, rois [K,5] = (batch_idx, x0, y0, x1, y1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float bilinear(const float* d, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
  float ly = y - yl, lx = x - xl, hy = 1.0f - ly, hx = 1.0f - lx;
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return w1 * d[yl * W + xl] + w2 * d[yl * W + xh] + w3 * d[yh * W + xl] + w4 * d[yh * W + xh];
}

void oracle_roi_align_forward(const float* in, const float* rois, float* out, int N, int C, int H, int W,
                              int K, int ph, int pw, float scale, int sampling_ratio, int aligned) {
  (void)N;
#pragma omp parallel for schedule(dynamic, 1)
  for (int k = 0; k < K; ++k) {
    const float* r = rois + 5 * k;
    int b = (int)r[0];
    float off = aligned ? 0.5f : 0.0f;
    float x0 = r[1] * scale - off, y0 = r[2] * scale - off;
    float x1 = r[3] * scale - off, y1 = r[4] * scale - off;
    float rw = x1 - x0, rh = y1 - y0;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    float bh = rh / ph, bw = rw / pw;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / ph);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / pw);
    float count = (float)(gh * gw > 1 ? gh * gw : 1);
    for (int c = 0; c < C; ++c) {
      const float* d = in + ((size_t)b * C + c) * H * W;
      float* o = out + ((size_t)k * C + c) * ph * pw;
      for (int i = 0; i < ph; ++i)
        for (int j = 0; j < pw; ++j) {
          float acc = 0.0f;
          for (int iy = 0; iy < gh; ++iy) {
            float y = y0 + i * bh + (iy + 0.5f) * bh / gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = x0 + j * bw + (ix + 0.5f) * bw / gw;
              acc += bilinear(d, H, W, y, x);
            }
          }
          o[i * pw + j] = acc / count;
        }
    }
  }
}

/* grad_in must be zeroed by the caller. Parallel over channels: each (b,c) plane has one writer. */
void oracle_roi_align_backward(const float* gout, const float* rois, float* gin, int N, int C, int H, int W,
                               int K, int ph, int pw, float scale, int sampling_ratio, int aligned) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < C; ++c) {
    for (int k = 0; k < K; ++k) {
      const float* r = rois + 5 * k;
      int b = (int)r[0];
      if (b < 0 || b >= N) continue;
      float off = aligned ? 0.5f : 0.0f;
      float x0 = r[1] * scale - off, y0 = r[2] * scale - off;
      float x1 = r[3] * scale - off, y1 = r[4] * scale - off;
      float rw = x1 - x0, rh = y1 - y0;
      if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
      float bh = rh / ph, bw = rw / pw;
      int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / ph);
      int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / pw);
      float count = (float)(gh * gw > 1 ? gh * gw : 1);
      float* d = gin + ((size_t)b * C + c) * H * W;
      const float* go = gout + ((size_t)k * C + c) * ph * pw;
      for (int i = 0; i < ph; ++i)
        for (int j = 0; j < pw; ++j) {
          float g = go[i * pw + j] / count;
          for (int iy = 0; iy < gh; ++iy) {
            float y = y0 + i * bh + (iy + 0.5f) * bh / gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = x0 + j * bw + (ix + 0.5f) * bw / gw;
              if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
              float yy = y <= 0 ? 0 : y, xx = x <= 0 ? 0 : x;
              int yl = (int)yy, xl = (int)xx, yh, xh;
              if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
              if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
              float ly = yy - yl, lx = xx - xl, hy = 1.0f - ly, hx = 1.0f - lx;
              d[yl * W + xl] += g * hy * hx;
              d[yl * W + xh] += g * hy * lx;
              d[yh * W + xl] += g * ly * hx;
              d[yh * W + xh] += g * ly * lx;
            }
          }
        }
    }
  }
}

/* Greedy NMS over boxes already sorted by descending score (order[] gives original indices).
 * Suppress j when inter/(area_i+area_j-inter) > thr.  Returns number kept; keep[] holds original
 * indices in score order. */
int oracle_nms_sorted(const float* boxes, const int64_t* order, int n, float thr, int64_t* keep) {
  unsigned char* dead = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int nk = 0;
  for (int a = 0; a < n; ++a) {
    if (dead[a]) continue;
    int64_t i = order[a];
    keep[nk++] = i;
    const float* bi = boxes + 4 * i;
    float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
    for (int c = a + 1; c < n; ++c) {
      if (dead[c]) continue;
      const float* bj = boxes + 4 * order[c];
      float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
      float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
      float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
      float inter = w * h;
      float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
      float iou = inter / (ai + aj - inter);
      if (iou > thr) dead[c] = 1;
    }
  }
  free(dead);
  return nk;
}
