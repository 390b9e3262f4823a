"""Episode input pipeline on the device.

Mirrors what the reference's loaders do per image on the CPU with cv2 / numpy:
  * `prep_im_for_blob`   lib/model/utils/blob.py:35-52 (called from roi_data_layer/minibatch.py:70-84)
  * `support_crop`       roi_data_layer/fs_loader.py:118-139 (crop the support box, fit it into target x target)
  * `crop_pad_chw`       fs_loader.py:186-280,318 (query crop window, ratio padding, permute(2, 0, 1))
  * `EpisodeHolders`     train.py:61-66,125-129 (persistent device holders the model is fed from)
Raw uint8 RGB frames are uploaded once; everything else happens in HBM through libdana_hip.so. The data-dependent
host decisions (which crop window, which supports: np.random / random in fs_loader.py) stay on the host, as they
are in the reference."""
import ctypes

import numpy as np
import torch

from ._lib import lib
from .config import cfg
from .ops import _chk, _p, _stream


def cv_round(x):
    """cv2's saturate_cast<int>(double): round half to even (cvRound)"""
    return int(np.rint(x))


def prep_im_for_blob(im_rgb_u8, pixel_means=None, target_size=600, max_size=1000, flipped=False):
    """im_rgb_u8: uint8 [h][w][3] RGB on the device (what `imread` returns, minibatch.py:70) -> (fp32 [oh][ow][3] BGR,
    mean-subtracted, resized so that the SHORTER side is target_size; im_scale). max_size is ignored exactly as in
    blob.py:45-47 (the cap is commented out there)."""
    if im_rgb_u8.dtype != torch.uint8 or im_rgb_u8.dim() != 3 or im_rgb_u8.size(2) != 3 or not im_rgb_u8.is_cuda:
        raise RuntimeError("prep_im_for_blob: need a uint8 [h][w][3] HIP tensor")
    im = im_rgb_u8.contiguous()
    h, w = im.size(0), im.size(1)
    means = np.asarray(cfg.PIXEL_MEANS if pixel_means is None else pixel_means, dtype=np.float32).reshape(-1)
    im_scale = float(target_size) / float(min(h, w))
    oh, ow = cv_round(h * im_scale), cv_round(w * im_scale)
    out = torch.empty((oh, ow, 3), dtype=torch.float32, device=im.device)
    f3 = ctypes.c_float * 3
    lib().call("dana_prep_image", im.data_ptr(), h, w, w * 3, int(bool(flipped)),
               ctypes.cast(f3(*[float(m) for m in means[:3]]), ctypes.c_void_p), im_scale, _p(out), oh, ow, _stream())
    return out, im_scale


def support_crop(im_hwc, box_scaled, target_size=320, out=None):
    """fs_loader.py:118-139. im_hwc: prepared fp32 [h][w][3]; box_scaled: (x_min, y_min, x_max, y_max) already
    multiplied by the image's scale and cast to int16 by the caller. -> fp32 [3][target][target]"""
    _chk(im_hwc, "im_hwc")
    h, w = im_hwc.size(0), im_hwc.size(1)
    x_min, y_min, x_max, y_max = [int(v) for v in box_scaled]
    box_h, box_w = y_max - y_min, x_max - x_min
    if box_h > box_w:
        rw, rh = int(box_w * (float(target_size) / float(box_h))), target_size
    else:
        rw, rh = target_size, int(box_h * (float(target_size) / float(box_w)))
    if out is None:
        out = torch.empty((3, target_size, target_size), dtype=torch.float32, device=im_hwc.device)
    lib().call("dana_crop_resize_pad", _p(im_hwc), h, w, x_min, y_min, min(x_max, w - 1), min(y_max, h - 1), rw, rh,
               target_size, _p(out), _stream())
    return out


def crop_pad_chw(im_hwc, y_start, x_start, crop_h, crop_w, out_h, out_w, out=None):
    """fs_loader.py:186-280,318: data[:, y_s:y_s+crop_h, x_s:x_s+crop_w, :] zero-padded to [out_h][out_w], as CHW"""
    _chk(im_hwc, "im_hwc")
    if out is None:
        out = torch.empty((3, out_h, out_w), dtype=torch.float32, device=im_hwc.device)
    lib().call("dana_crop_pad_chw", _p(im_hwc), im_hwc.size(0), im_hwc.size(1), y_start, x_start, crop_h, crop_w,
               _p(out), out_h, out_w, _stream())
    return out


def pad_plan(data_h, data_w, ratio):
    """the padding branch of fs_loader.py:257-280 for an image that needs no crop: -> (out_h, out_w, copy_h, copy_w)"""
    if ratio < 1:
        return int(np.ceil(data_w / ratio)), data_w, data_h, data_w
    if ratio > 1:
        return data_h, int(np.ceil(data_h * ratio)), data_h, data_w
    t = min(data_h, data_w)
    return t, t, t, t


class EpisodeHolders:
    """train.py:61-66: the five persistent device tensors (im_data, im_info, gt_boxes, num_boxes, support_ims) one
    batch of episodes is assembled in; `put_*` write one episode's slice with the kernels above."""

    def __init__(self, batch, way, shot, height, width, device, support_size=320, max_num_box=None):
        nb = cfg.MAX_NUM_GT_BOXES if max_num_box is None else max_num_box
        self.im_data = torch.zeros((batch, 3, height, width), dtype=torch.float32, device=device)
        self.im_info = torch.zeros((batch, 3), dtype=torch.float32, device=device)
        self.gt_boxes = torch.zeros((batch, nb, 5), dtype=torch.float32, device=device)
        self.num_boxes = torch.zeros((batch,), dtype=torch.long, device=device)
        self.support_ims = torch.zeros((batch, way * shot, 3, support_size, support_size), dtype=torch.float32,
                                       device=device)
        self.support_size = support_size

    def put_query(self, b, im_hwc, im_scale, y_start=0, x_start=0, crop_h=None, crop_w=None):
        H, W = self.im_data.size(2), self.im_data.size(3)
        crop_h = im_hwc.size(0) - y_start if crop_h is None else crop_h
        crop_w = im_hwc.size(1) - x_start if crop_w is None else crop_w
        crop_pad_chw(im_hwc, y_start, x_start, min(crop_h, H), min(crop_w, W), H, W, out=self.im_data[b])
        self.im_info[b] = torch.tensor([float(H), float(W), float(im_scale)], device=self.im_info.device)

    def put_support(self, b, slot, im_hwc, box_scaled):
        support_crop(im_hwc, box_scaled, self.support_size, out=self.support_ims[b, slot])

    def put_boxes(self, b, boxes):
        """boxes: float [n][5] (already scaled / shifted / clamped / filtered like fs_loader.py:282-313)"""
        n = min(int(boxes.shape[0]), self.gt_boxes.size(1))
        self.gt_boxes[b].zero_()
        if n:
            self.gt_boxes[b, :n] = torch.as_tensor(boxes[:n], dtype=torch.float32).to(self.gt_boxes.device)
        self.num_boxes[b] = n

    def tensors(self):
        return self.im_data, self.im_info, self.gt_boxes, self.num_boxes, self.support_ims
