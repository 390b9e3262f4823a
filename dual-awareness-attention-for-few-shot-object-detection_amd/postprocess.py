"""Inference post-processing on device: what inference.py:106-140 does between the
model call and `all_boxes[j][i] = cls_dets` -- de-normalise the regression deltas, decode them on the rois,
clip, rescale to the original image, threshold the fg score, sort, NMS (utils.py:312-317) -- as ONE C call
(decode + device sort + on-device NMS) and one small D2H read for the variable-length result."""
import ctypes

import torch

from . import ops
from ._lib import lib
from .config import cfg


def detections(rois, cls_prob, bbox_pred, im_info, thresh=0.05, nms_inclusive=False):
    """rois [1,R,5], cls_prob [R,2], bbox_pred [R,4], im_info [1,3] (device tensors of one image, as returned
    by the eval forward) -> cls_dets [K,5] = (x1,y1,x2,y2,score), descending score (what utils.NMS returns)."""
    rois = ops._chk(rois.reshape(-1, 5).contiguous(), "rois")
    cls_prob = ops._chk(cls_prob.reshape(-1, 2).contiguous(), "cls_prob")
    bbox_pred = ops._chk(bbox_pred.reshape(-1, 4).contiguous(), "bbox_pred")
    im_info = ops._chk(im_info.reshape(-1)[:3].float().contiguous(), "im_info")
    R = rois.size(0)
    dev = rois.device
    dets = torch.empty((R, 5), dtype=torch.float32, device=dev)
    ibuf = torch.empty((R + 2,), dtype=torch.int32, device=dev)  # keep positions | meta[2]
    ws = ops._ws(lib().query("dana_detect_postprocess_workspace_bytes", R), dev)
    f4 = ctypes.c_float * 4
    lib().call("dana_detect_postprocess", ops._p(rois), ops._p(cls_prob), ops._p(bbox_pred), ops._p(im_info), R,
               ctypes.cast(f4(*cfg.TRAIN.BBOX_NORMALIZE_STDS), ctypes.c_void_p),
               ctypes.cast(f4(*cfg.TRAIN.BBOX_NORMALIZE_MEANS), ctypes.c_void_p),
               int(bool(cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED)), float(thresh), float(cfg.TEST.NMS),
               int(bool(nms_inclusive)), ops._p(dets), ibuf.data_ptr(), ibuf.data_ptr() + 4 * R, ops._p(ws),
               ws.numel(), ops._stream())
    host = ibuf.cpu()
    n_valid, n_keep = int(host[R]), int(host[R + 1])
    keep = host[:n_keep]
    keep = keep[keep < n_valid].long().to(dev)
    return dets[keep]
