"""Drop-in for the reference's pybind module ``model._C`` (lib/model/csrc/vision.cpp:7-13): the same
five names and signatures, served by libdana_hip.so. `lib/model/roi_layers/*.py`-style callers
(``from model import _C``) work unchanged when this module is installed as ``model._C``
(see INTEGRATION.md). CPU tensors raise, like a reference build without its CPU kernels would for
roi_pool / roi_align_backward ("Not implemented on the CPU", ROIAlign.h:44, ROIPool.h:22)."""
from . import ops


def nms(dets, scores, threshold):
    """nms.h:10-28 -> int64 kept indices, ascending (IoU > threshold suppresses, nms.cu:60)."""
    return ops.nms(dets, scores, float(threshold), inclusive=False)


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    return ops.roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio)


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width,
                       sampling_ratio):
    return ops.roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                                  height, width, sampling_ratio)


def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    return ops.roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width)


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                      height, width):
    return ops.roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size,
                                 channels, height, width)
