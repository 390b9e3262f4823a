// Episode input pipeline: what the reference's cv2 / numpy loaders do per image on the CPU
// (8 DataLoader workers, train.py:45-57) as three HBM-bound kernels, so that raw uint8 frames go in and the
// model's NCHW fp32 holders (train.py:61-66) come out without a host round trip.
//
//   prep_image_kernel      minibatch.py:70-84 + blob.py:35-52: RGB->BGR, optional horizontal flip, fp32, minus
//                          cfg.PIXEL_MEANS, cv2.resize(fx = fy = im_scale, INTER_LINEAR)  -> [oh][ow][3] fp32
//   crop_resize_pad_kernel fs_loader.py:118-139: crop a box out of a prepared image, cv2.resize it so that its longer
//                          side is `target`, transpose to CHW, zero-pad to [3][target][target]
//   crop_pad_chw_kernel    fs_loader.py:186-280,318: crop window + zero padding + permute(2,0,1) into the batch holder
//
// cv2.resize(INTER_LINEAR) on fp32, restated (OpenCV resize.cpp, float path): sample position
// f = (d + 0.5) * scale - 0.5 evaluated in double and rounded to float, s = floor(f), f -= s; s < 0 -> (s, f) = (0, 0);
// s >= n - 1 -> (s, f) = (n - 1, 0); horizontal pass D = S[s] * (1 - f) + S[s + 1] * f, then the vertical pass with
// the same form. Compiled with -ffp-contract=off so that the arithmetic is exactly this.
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

struct Tap {
  int s0, s1;
  float a0, a1;
};
__device__ __forceinline__ Tap linear_tap(int d, double scale, int n) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    s = 0;
    f = 0.f;
  }
  if (s >= n - 1) {
    s = n - 1;
    f = 0.f;
  }
  Tap t;
  t.s0 = s;
  t.s1 = min(s + 1, n - 1);
  t.a0 = 1.f - f;
  t.a1 = f;
  return t;
}

__global__ void __launch_bounds__(256)
prep_image_kernel(const unsigned char* __restrict__ im, int h, int w, long row_stride, int flipped, float m0, float m1,
                  float m2, double scale_x, double scale_y, float* __restrict__ out, int oh, int ow) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)oh * ow) return;
  const int x = (int)(i % ow), y = (int)(i / ow);
  const Tap tx = linear_tap(x, scale_x, w), ty = linear_tap(y, scale_y, h);
  const int x0 = flipped ? w - 1 - tx.s0 : tx.s0, x1 = flipped ? w - 1 - tx.s1 : tx.s1;
  const unsigned char* r0 = im + (long)ty.s0 * row_stride;
  const unsigned char* r1 = im + (long)ty.s1 * row_stride;
  const float mean[3] = {m0, m1, m2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // output channel c (BGR) reads input channel 2 - c (RGB)
    const int ci = 2 - c;
    const float t00 = (float)r0[x0 * 3 + ci] - mean[c], t01 = (float)r0[x1 * 3 + ci] - mean[c];
    const float t10 = (float)r1[x0 * 3 + ci] - mean[c], t11 = (float)r1[x1 * 3 + ci] - mean[c];
    const float h0 = t00 * tx.a0 + t01 * tx.a1;
    const float h1 = t10 * tx.a0 + t11 * tx.a1;
    out[i * 3 + c] = h0 * ty.a0 + h1 * ty.a1;
  }
}

__global__ void __launch_bounds__(256)
crop_resize_pad_kernel(const float* __restrict__ im, int w, int x0, int y0, int cw, int ch, int rw, int rh, int target,
                       double scale_x, double scale_y, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)target * target) return;
  const int x = (int)(i % target), y = (int)(i / target);
  float v[3] = {0.f, 0.f, 0.f};
  if (x < rw && y < rh) {
    const Tap tx = linear_tap(x, scale_x, cw), ty = linear_tap(y, scale_y, ch);
    const float* r0 = im + ((long)(y0 + ty.s0) * w + x0) * 3;
    const float* r1 = im + ((long)(y0 + ty.s1) * w + x0) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float h0 = r0[tx.s0 * 3 + c] * tx.a0 + r0[tx.s1 * 3 + c] * tx.a1;
      const float h1 = r1[tx.s0 * 3 + c] * tx.a0 + r1[tx.s1 * 3 + c] * tx.a1;
      v[c] = h0 * ty.a0 + h1 * ty.a1;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[(long)c * target * target + i] = v[c];
}

__global__ void __launch_bounds__(256)
crop_pad_chw_kernel(const float* __restrict__ im, int w, int y_s, int x_s, int ch, int cw, float* __restrict__ out, int oh,
                    int ow) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)oh * ow) return;
  const int x = (int)(i % ow), y = (int)(i / ow);
  const bool in = x < cw && y < ch;
  const float* p = im + ((long)(y_s + y) * w + x_s + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) out[(long)c * oh * ow + i] = in ? p[c] : 0.f;
}

}  // namespace

extern "C" {

int dana_prep_image(const unsigned char* rgb_hwc, int height, int width, long row_stride_bytes, int flipped,
                    const float* pixel_means_bgr, double im_scale, float* out_hwc, int out_h, int out_w,
                    dana_stream_t stream) {
  DANA_CHECK_ARG(height > 0 && width > 0 && out_h > 0 && out_w > 0 && im_scale > 0.0, "dana_prep_image: bad shape");
  DANA_CHECK_ARG(rgb_hwc && pixel_means_bgr && out_hwc, "dana_prep_image: null pointer");
  if (row_stride_bytes <= 0) row_stride_bytes = (long)width * 3;
  // cv2.resize(fx, fy): the sampling scale is 1 / fx (not width / out_w)
  const double s = 1.0 / im_scale;
  const long total = (long)out_h * out_w;
  prep_image_kernel<<<dana_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
      rgb_hwc, height, width, row_stride_bytes, flipped, pixel_means_bgr[0], pixel_means_bgr[1], pixel_means_bgr[2], s, s,
      out_hwc, out_h, out_w);
  DANA_CHECK_LAUNCH("dana_prep_image");
  return DANA_OK;
}

int dana_crop_resize_pad(const float* im_hwc, int height, int width, int x_min, int y_min, int x_max, int y_max,
                         int resized_w, int resized_h, int target, float* out_chw, dana_stream_t stream) {
  DANA_CHECK_ARG(height > 0 && width > 0 && target > 0 && resized_w > 0 && resized_h > 0 && resized_w <= target &&
                     resized_h <= target,
                 "dana_crop_resize_pad: bad shape");
  DANA_CHECK_ARG(x_min >= 0 && y_min >= 0 && x_max >= x_min && y_max >= y_min && x_max < width && y_max < height,
                 "dana_crop_resize_pad: crop box outside the image");
  DANA_CHECK_ARG(im_hwc && out_chw, "dana_crop_resize_pad: null pointer");
  const int cw = x_max - x_min + 1, ch = y_max - y_min + 1;  // the slice [min : max + 1] (fs_loader.py:125)
  // cv2.resize(dsize): the sampling scale is src / dst
  const long total = (long)target * target;
  crop_resize_pad_kernel<<<dana_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
      im_hwc, width, x_min, y_min, cw, ch, resized_w, resized_h, target, (double)cw / resized_w, (double)ch / resized_h,
      out_chw);
  DANA_CHECK_LAUNCH("dana_crop_resize_pad");
  return DANA_OK;
}

int dana_crop_pad_chw(const float* im_hwc, int height, int width, int y_start, int x_start, int crop_h, int crop_w,
                      float* out_chw, int out_h, int out_w, dana_stream_t stream) {
  DANA_CHECK_ARG(height > 0 && width > 0 && out_h > 0 && out_w > 0 && crop_h >= 0 && crop_w >= 0 && y_start >= 0 &&
                     x_start >= 0 && y_start + crop_h <= height && x_start + crop_w <= width,
                 "dana_crop_pad_chw: bad shape");
  DANA_CHECK_ARG(im_hwc && out_chw, "dana_crop_pad_chw: null pointer");
  const long total = (long)out_h * out_w;
  crop_pad_chw_kernel<<<dana_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
      im_hwc, width, y_start, x_start, crop_h < out_h ? crop_h : out_h, crop_w < out_w ? crop_w : out_w, out_chw, out_h,
      out_w);
  DANA_CHECK_LAUNCH("dana_crop_pad_chw");
  return DANA_OK;
}

}  // extern "C"
