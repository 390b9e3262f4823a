"""Drop-in for the reference's pybind module ``model._C`` (lib/model/csrc/vision.cpp:7-13): the same
five names and signatures, served by libdana_hip.so. `lib/model/roi_layers/*.py`-style callers
(``from model import _C``) work unchanged when this module is installed as ``model._C``
(see INTEGRATION.md). CPU tensors raise, like a reference build without its CPU kernels would for
roi_pool / roi_align_backward ("Not implemented on the CPU", ROIAlign.h:44, ROIPool.h:22).

Two bindings of the same C ABI (include/dana_hip.h):
  * ``torch.ops.dana.*`` -- PyTorch custom operators registered with TORCH_LIBRARY by csrc/torch_ops.cpp
    (libdana_torch_ops.so, built by `make`); used when that library loads. `torch.ops.dana.roi_align` /
    `roi_pool` carry autograd formulas.
  * ctypes (`ops.py`, prototypes parsed from the header) -- always available; `BINDING` says which one is live."""
import os

import torch

from . import ops

_TOPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdana_torch_ops.so")
BINDING = "ctypes"
if os.path.exists(_TOPS) and os.environ.get("DANA_NO_TORCH_OPS", "0") == "0":
    try:
        from ._lib import lib as _lib
        _lib()  # libdana_hip.so first (the shim links against it)
        torch.ops.load_library(_TOPS)
        BINDING = "torch.ops.dana"
    except (OSError, RuntimeError) as _e:  # pragma: no cover
        BINDING = "ctypes (torch.ops shim failed to load: %s)" % _e


def nms(dets, scores, threshold):
    """nms.h:10-28 -> int64 kept indices, ascending (IoU > threshold suppresses, nms.cu:60)."""
    if BINDING == "torch.ops.dana":
        return torch.ops.dana.nms(dets, scores, float(threshold))
    return ops.nms(dets, scores, float(threshold), inclusive=False)


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    if BINDING == "torch.ops.dana":
        return torch.ops.dana.roi_align_forward(input, rois, float(spatial_scale), pooled_height, pooled_width,
                                                sampling_ratio)
    return ops.roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio)


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width,
                       sampling_ratio):
    if BINDING == "torch.ops.dana":
        return torch.ops.dana.roi_align_backward(grad, rois, float(spatial_scale), pooled_height, pooled_width,
                                                 batch_size, channels, height, width, sampling_ratio)
    return ops.roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                                  height, width, sampling_ratio)


def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    if BINDING == "torch.ops.dana":
        return torch.ops.dana.roi_pool_forward(input, rois, float(spatial_scale), pooled_height, pooled_width)
    return ops.roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width)


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                      height, width):
    if BINDING == "torch.ops.dana":
        return torch.ops.dana.roi_pool_backward(grad, input, rois, argmax, float(spatial_scale), pooled_height,
                                                pooled_width, batch_size, channels, height, width)
    return ops.roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size,
                                 channels, height, width)
