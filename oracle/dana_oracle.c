/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked, imported or called by the product path
 * (dual-awareness-attention-for-few-shot-object-detection_amd/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, as the checker.
 *
 * Plain-C (gcc, -O2 -ffp-contract=off) restatement of the reference's native operators and box
 * arithmetic for the DAnA forward path. Each function cites the reference lines it follows
 * (paths relative to the reference repository). Pinned against the reference itself: see
 * tests/golden/make_golden.py (run in the build container, imports the reference's Python and a
 * scratch build of its CPU `_C`) and tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ---- greedy NMS: lib/model/csrc/cpu/nms_cpu.cpp:5-65 (suppress on ovr >= thr, :60);
 * inclusive=0 gives the CUDA variant's `>` (lib/model/csrc/cuda/nms.cu:60).
 * boxes[n][4] are visited in the given `order` (descending score, computed by the caller as the
 * reference does with scores.sort, nms_cpu.cpp:24). Writes suppressed[n] (0/1); returns #kept.
 * keep_out (optional) receives kept ORIGINAL indices ascending (nms_cpu.cpp:64). */
int oracle_nms(const float* boxes, const int64_t* order, int n, float thr, int inclusive, uint8_t* suppressed,
               int64_t* keep_out) {
  float* areas = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    const float* b = boxes + 4 * i;
    areas[i] = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  }
  memset(suppressed, 0, n);
  for (int _i = 0; _i < n; ++_i) {
    int64_t i = order[_i];
    if (suppressed[i]) continue;
    float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    float iarea = areas[i];
    for (int _j = _i + 1; _j < n; ++_j) {
      int64_t j = order[_j];
      if (suppressed[j]) continue;
      float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
      float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
      float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (inclusive ? (ovr >= thr) : (ovr > thr)) suppressed[j] = 1;
    }
  }
  int k = 0;
  for (int i = 0; i < n; ++i)
    if (!suppressed[i]) {
      if (keep_out) keep_out[k] = i;
      ++k;
    }
  free(areas);
  return k;
}

/* ---- RoIAlign forward, NCHW: lib/model/csrc/cpu/ROIAlign_cpu.cpp:17-219
 * (pre_calc_for_bilinear_interpolate + ROIAlignForward_cpu_kernel); identical maths to the CUDA
 * kernel lib/model/csrc/cuda/ROIAlign_cuda.cu:15-122. */
void oracle_roi_align_forward(const float* in, const float* rois, float* out, int C, int H, int W, int R, float scale,
                              int PH, int PW, int sampling_ratio) {
  for (int n = 0; n < R; ++n) {
    const float* roi = rois + 5 * n;
    int b = (int)roi[0];
    float sw = roi[1] * scale, sh = roi[2] * scale, ew = roi[3] * scale, eh = roi[4] * scale;
    float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    float count = (float)(gh * gw);
    for (int c = 0; c < C; ++c) {
      const float* plane = in + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float acc = 0.f;
          for (int iy = 0; iy < gh; ++iy) {
            float yy = sh + ph * bh + ((float)iy + .5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float xx = sw + pw * bw + ((float)ix + .5f) * bw / (float)gw;
              float x = xx, y = yy;
              if (y < -1.0 || y > H || x < -1.0 || x > W) continue; /* all-zero weights */
              if (y <= 0) y = 0;
              if (x <= 0) x = 0;
              int yl = (int)y, xl = (int)x, yh, xh;
              if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
              if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
              float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
              float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
              acc += w1 * plane[yl * W + xl] + w2 * plane[yl * W + xh] + w3 * plane[yh * W + xl] +
                     w4 * plane[yh * W + xh];
            }
          }
          out[(((size_t)n * C + c) * PH + ph) * PW + pw] = acc / count;
        }
    }
  }
}

/* ---- RoIPool forward, NCHW: lib/model/csrc/cuda/ROIPool_cuda.cu:16-77 (no CPU version exists in
 * the reference; restated from the CUDA kernel, unpinned by execution). */
void oracle_roi_pool_forward(const float* in, const float* rois, float* out, int* argmax, int C, int H, int W, int R,
                             float scale, int PH, int PW) {
  for (int n = 0; n < R; ++n) {
    const float* roi = rois + 5 * n;
    int b = (int)roi[0];
    int rsw = (int)roundf(roi[1] * scale), rsh = (int)roundf(roi[2] * scale);
    int rew = (int)roundf(roi[3] * scale), reh = (int)roundf(roi[4] * scale);
    int rw = rew - rsw + 1 > 1 ? rew - rsw + 1 : 1, rh = reh - rsh + 1 > 1 ? reh - rsh + 1 : 1;
    float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    for (int c = 0; c < C; ++c) {
      const float* plane = in + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          int hs = (int)floorf((float)ph * bh), ws = (int)floorf((float)pw * bw);
          int he = (int)ceilf((float)(ph + 1) * bh), we = (int)ceilf((float)(pw + 1) * bw);
          hs = hs + rsh < 0 ? 0 : (hs + rsh > H ? H : hs + rsh);
          he = he + rsh < 0 ? 0 : (he + rsh > H ? H : he + rsh);
          ws = ws + rsw < 0 ? 0 : (ws + rsw > W ? W : ws + rsw);
          we = we + rsw < 0 ? 0 : (we + rsw > W ? W : we + rsw);
          int empty = (he <= hs) || (we <= ws);
          float m = empty ? 0.f : -FLT_MAX;
          int mi = -1;
          for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w)
              if (plane[h * W + w] > m) { m = plane[h * W + w]; mi = h * W + w; }
          size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
          out[o] = m;
          argmax[o] = mi;
        }
    }
  }
}

/* ---- anchors + bbox_transform_inv + clip_boxes:
 * lib/model/rpn/proposal_layer.py:80-93 (shift grid), lib/model/rpn/bbox_transform.py:77-103, :125-133.
 * deltas[K*A][4] in (h, w, a) order, base_anchors[A][4]; out[K*A][4]. expf is the libm one. */
void oracle_decode_clip(const float* base_anchors, const float* deltas, float* out, int A, int H, int W,
                        int feat_stride, float im_h, float im_w) {
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w)
      for (int a = 0; a < A; ++a) {
        size_t i = ((size_t)(h * W + w)) * A + a;
        float sx = (float)(w * feat_stride), sy = (float)(h * feat_stride);
        float x1 = base_anchors[4 * a] + sx, y1 = base_anchors[4 * a + 1] + sy;
        float x2 = base_anchors[4 * a + 2] + sx, y2 = base_anchors[4 * a + 3] + sy;
        float widths = x2 - x1 + 1.0f, heights = y2 - y1 + 1.0f;
        float cx = x1 + 0.5f * widths, cy = y1 + 0.5f * heights;
        const float* d = deltas + 4 * i;
        float pcx = d[0] * widths + cx, pcy = d[1] * heights + cy;
        float pw = expf(d[2]) * widths, ph = expf(d[3]) * heights;
        float o[4] = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph};
        o[0] = fminf(fmaxf(o[0], 0.f), im_w - 1.f);
        o[1] = fminf(fmaxf(o[1], 0.f), im_h - 1.f);
        o[2] = fminf(fmaxf(o[2], 0.f), im_w - 1.f);
        o[3] = fminf(fmaxf(o[3], 0.f), im_h - 1.f);
        memcpy(out + 4 * i, o, sizeof(o));
      }
}

/* ---- RoIAlign backward, NCHW: lib/model/csrc/cuda/ROIAlign_cuda.cu:125-254
 * (bilinear_interpolate_gradient + RoIAlignBackwardFeature). The reference scatters with float
 * atomicAdd in an unspecified order; this restatement visits (n, c, ph, pw, iy, ix) in index order
 * and accumulates in DOUBLE, so it is the order-free value any atomics order rounds towards.
 * gin[B][C][H][W] must be zeroed by the caller (ROIAlign_cuda.cu:318 at::zeros). */
void oracle_roi_align_backward(const float* gout, const float* rois, double* gin, int C, int H, int W, int R,
                               float scale, int PH, int PW, int sampling_ratio) {
  for (int n = 0; n < R; ++n) {
    const float* r = rois + 5 * n;
    const int b = (int)r[0];
    const float roi_start_w = r[1] * scale, roi_start_h = r[2] * scale;
    const float roi_end_w = r[3] * scale, roi_end_h = r[4] * scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
    const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
    const float bin_size_h = roi_height / (float)PH, bin_size_w = roi_width / (float)PW;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / PH);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / PW);
    const float count = (float)(grid_h * grid_w);
    for (int c = 0; c < C; ++c) {
      double* g = gin + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          const float top = gout[(((size_t)n * C + c) * PH + ph) * PW + pw];
          for (int iy = 0; iy < grid_h; ++iy) {
            float y = roi_start_h + ph * bin_size_h + (float)(iy + .5f) * bin_size_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ++ix) {
              float x = roi_start_w + pw * bin_size_w + (float)(ix + .5f) * bin_size_w / (float)grid_w;
              float yy = y;
              if (yy < -1.0 || yy > H || x < -1.0 || x > W) continue; /* :137-142: weights 0, indices -1 */
              if (yy <= 0) yy = 0;
              if (x <= 0) x = 0;
              int y_low = (int)yy, x_low = (int)x, y_high, x_high;
              if (y_low >= H - 1) {
                y_high = y_low = H - 1;
                yy = (float)y_low;
              } else
                y_high = y_low + 1;
              if (x_low >= W - 1) {
                x_high = x_low = W - 1;
                x = (float)x_low;
              } else
                x_high = x_low + 1;
              const float ly = yy - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
              const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
              g[y_low * W + x_low] += (double)(top * w1 / count);
              g[y_low * W + x_high] += (double)(top * w2 / count);
              g[y_high * W + x_low] += (double)(top * w3 / count);
              g[y_high * W + x_high] += (double)(top * w4 / count);
            }
          }
        }
    }
  }
}
