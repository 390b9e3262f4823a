"""TEST INFRASTRUCTURE ONLY (imported by tests/): CPU restatement of the reference's per-image input pipeline.

* `cv_resize_linear`: cv2.resize(..., interpolation=cv2.INTER_LINEAR) on float32 arrays as OpenCV's float path
  computes it (sample position (d + 0.5) * scale - 0.5 in double -> float, floor, clamp to [0, n - 1], horizontal
  pass then vertical pass in float32). cv2 is the reference's third-party dependency (opencv-python, unpinned in
  its requirements) and is NOT installed in this image: **parity unpinned against cv2 itself**; the restatement is
  checked against hand-computed known-answer vectors of OpenCV's documented sampling rule (pixel-centre convention,
  edge clamps, pass order, fx / fy form) and cross-checked against an independent implementation of the same rule,
  torch's F.interpolate(mode="bilinear", align_corners=False) (tests/test_oracle_golden.py).
* `prep_im_for_blob`: lib/model/utils/blob.py:35-52 on top of it (RGB->BGR + flip as minibatch.py:76-81).
* `support_crop`: roi_data_layer/fs_loader.py:118-139.
* `crop_pad_chw`: fs_loader.py:186-280,318."""
import numpy as np


def _taps(n_dst, scale, n_src):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= n_src - 1
    s[hi], f[hi] = n_src - 1, 0.0
    return s, np.minimum(s + 1, n_src - 1), (np.float32(1.0) - f).astype(np.float32), f.astype(np.float32)


def cv_resize_linear(src, dsize=None, fx=None, fy=None):
    """src float32 [h][w][c]; dsize = (width, height) like cv2, or fx / fy"""
    src = np.asarray(src, dtype=np.float32)
    h, w = src.shape[:2]
    if dsize is None:
        ow, oh = int(np.rint(w * fx)), int(np.rint(h * fy))
        sx, sy = 1.0 / fx, 1.0 / fy
    else:
        ow, oh = dsize
        sx, sy = w / float(ow), h / float(oh)
    x0, x1, a0, a1 = _taps(ow, sx, w)
    y0, y1, b0, b1 = _taps(oh, sy, h)
    hor = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]
    return (hor[y0] * b0[:, None, None] + hor[y1] * b1[:, None, None]).astype(np.float32)


def prep_im_for_blob(im_rgb_u8, pixel_means, target_size, flipped=False):
    im = im_rgb_u8[:, :, ::-1]
    if flipped:
        im = im[:, ::-1, :]
    im = im.astype(np.float32) - np.asarray(pixel_means, dtype=np.float32).reshape(1, 1, 3)
    im_scale = float(target_size) / float(min(im.shape[0], im.shape[1]))
    return cv_resize_linear(im, fx=im_scale, fy=im_scale), im_scale


def support_crop(im, box, target_size):
    x_min, y_min, x_max, y_max = [int(v) for v in box]
    box_h, box_w = y_max - y_min, x_max - x_min
    crop = im[y_min:y_max + 1, x_min:x_max + 1, :]
    if box_h > box_w:
        crop = cv_resize_linear(crop, dsize=(int(box_w * (float(target_size) / float(box_h))), target_size))
    else:
        crop = cv_resize_linear(crop, dsize=(target_size, int(box_h * (float(target_size) / float(box_w)))))
    out = np.zeros((3, target_size, target_size), dtype=np.float32)
    out[:, :crop.shape[0], :crop.shape[1]] = np.transpose(crop, (2, 0, 1))
    return out


def crop_pad_chw(im, y_s, x_s, crop_h, crop_w, out_h, out_w):
    out = np.zeros((out_h, out_w, 3), dtype=np.float32)
    c = im[y_s:y_s + crop_h, x_s:x_s + crop_w][:out_h, :out_w]
    out[:c.shape[0], :c.shape[1]] = c
    return np.ascontiguousarray(np.transpose(out, (2, 0, 1)))
