"""ORACLE (test infrastructure): ctypes access to oracle/liboracle.so (dana_oracle.c, plain C)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build():
    src = os.path.join(_HERE, "dana_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _l():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def nms(dets, scores, thr, inclusive=True):
    """lib/model/csrc/cpu/nms_cpu.cpp:5-65 -> kept original indices (int64, ascending).
    inclusive=True is the reference CPU op (>=); False the reference CUDA op (>)."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = dets.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores, kind="stable").astype(np.int64)
    sup = np.zeros(n, dtype=np.uint8)
    keep = np.zeros(n, dtype=np.int64)
    f = _l().oracle_nms
    f.restype = ctypes.c_int
    k = f(_fp(dets), _fp(order), ctypes.c_int(n), ctypes.c_float(thr), ctypes.c_int(int(inclusive)), _fp(sup),
          _fp(keep))
    return keep[:k].copy()


def roi_align_forward(inp, rois, scale, ph, pw, sampling_ratio):
    """lib/model/csrc/cpu/ROIAlign_cpu.cpp:113-219; NCHW in, [R,C,PH,PW] out."""
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = inp.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ph, pw), dtype=np.float32)
    if R:
        _l().oracle_roi_align_forward(_fp(inp), _fp(rois), _fp(out), C, H, W, R, ctypes.c_float(scale), ph, pw,
                                      sampling_ratio)
    return out


def roi_pool_forward(inp, rois, scale, ph, pw):
    """lib/model/csrc/cuda/ROIPool_cuda.cu:16-77 restated on the CPU."""
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = inp.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ph, pw), dtype=np.float32)
    arg = np.zeros((R, C, ph, pw), dtype=np.int32)
    if R:
        _l().oracle_roi_pool_forward(_fp(inp), _fp(rois), _fp(out), _fp(arg), C, H, W, R, ctypes.c_float(scale), ph,
                                     pw)
    return out, arg


def decode_clip(base_anchors, deltas, H, W, feat_stride, im_h, im_w):
    """anchors + bbox_transform_inv + clip_boxes for ONE image; deltas [H*W*A, 4] in (h, w, a) order."""
    base_anchors = np.ascontiguousarray(base_anchors, dtype=np.float32)
    deltas = np.ascontiguousarray(deltas, dtype=np.float32)
    A = base_anchors.shape[0]
    out = np.zeros((H * W * A, 4), dtype=np.float32)
    _l().oracle_decode_clip(_fp(base_anchors), _fp(deltas), _fp(out), A, H, W, feat_stride, ctypes.c_float(im_h),
                            ctypes.c_float(im_w))
    return out


def roi_align_backward(gout, rois, scale, ph, pw, batch, channels, height, width, sampling_ratio):
    """lib/model/csrc/cuda/ROIAlign_cuda.cu:125-254 (the reference has no CPU backward, ROIAlign.h:44): every per-sample
    contribution exactly as the CUDA kernel forms it (fp32), summed in float64 -> [B,C,H,W] float64."""
    gout = np.ascontiguousarray(gout, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    gin = np.zeros((batch, channels, height, width), dtype=np.float64)
    if rois.shape[0]:
        _l().oracle_roi_align_backward(_fp(gout), _fp(rois), _fp(gin), channels, height, width, rois.shape[0],
                                       ctypes.c_float(scale), ph, pw, sampling_ratio)
    return gin
