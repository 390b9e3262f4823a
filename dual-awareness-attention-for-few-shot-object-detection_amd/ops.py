"""Tensor-level wrappers over the libdana_hip.so C ABI (include/dana_hip.h).

PyTorch is used here only for device memory and streams: every function hands raw device pointers
and the current HIP stream to a hand-written gfx950 kernel. Non-CUDA tensors are rejected -- there
is no CPU fallback on the product path.
"""
import time as _time

import torch

from ._lib import lib, DanaError  # noqa: F401

HOST_WAIT = [0.0]  # seconds the host spent blocked in the training forward's one D2H read (bench.py, tools/program_hostprof.py)

NCHW, NHWC = 0, 1
EPI_RELU, CONV_STEM7, W_SPLIT3, A_SPLIT3 = 1, 2, 256, 512

# bench.py sets this to a list to time every MFMA contraction launch with HIP events recorded on the
# stream the kernel is launched on; entries are (tag, algorithmic_flops, start_event, end_event).
PROFILE = None
PROF_ROLE = None  # set (while profiling) around launches that play another role than their kernel's name says: "dgrad"


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, tag, flops, nbytes=0.0, executed=None):
    """tag: (format, args) -- formatted only when profiling; flops: ALGORITHMIC flops (2 per multiply-add of the direct
    formulation); nbytes: algorithmic HBM bytes of the launch (every operand and result touched exactly once);
    executed: multiply-add flops the matrix cores actually run (differs for the Winograd-domain launches: 4x fewer)"""
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        name = tag[0] % tag[1]
        if PROF_ROLE is not None and not name.startswith(PROF_ROLE):
            name = PROF_ROLE + ":" + name  # e.g. "dgrad:wino3x3 ...": a data gradient that runs on a forward kernel
        PROFILE.append((name, flops, e0, e1, nbytes, flops if executed is None else executed))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
_get_cur = getattr(torch._C, "_cuda_getCurrentStream", None)
_set_cur = getattr(torch._C, "_cuda_setStream", None)


# Stream bookkeeping without torch.cuda's Python layers. torch.cuda.current_stream(), `with torch.cuda.stream(s)` and
# Event.record() each resolve the device index through several wrappers and an os.environ look-up: ~8 us per call and
# ~1 800 calls per eager training iteration (tools/hostprof_step.py: >1 ms of its host time in the forward alone). The
# helpers below go to the same C entry points directly; on a torch build without them they are the torch.cuda calls.
def cur_stream():
    """torch.cuda.current_stream() of the current device"""
    if _get_cur is None or _raw_device is None:
        return torch.cuda.current_stream()
    sd = _get_cur(_raw_device())
    return torch.cuda.Stream(stream_id=sd[0], device_index=sd[1], device_type=sd[2])


class on_stream:
    """`with torch.cuda.stream(st)` for a stream of the CURRENT device"""
    __slots__ = ("st", "prev")

    def __init__(self, st):
        self.st = st

    def __enter__(self):
        st = self.st
        if _get_cur is None or _set_cur is None or _raw_device is None or st.device_index != _raw_device():
            self.prev = torch.cuda.stream(st)
            self.prev.__enter__()
            return
        self.prev = _get_cur(st.device_index)
        _set_cur(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)

    def __exit__(self, *exc):
        prev = self.prev
        if isinstance(prev, tuple):
            _set_cur(stream_id=prev[0], device_index=prev[1], device_type=prev[2])
        else:
            prev.__exit__(*exc)


def record_event(stream=None, timing=False):
    """a fresh event recorded on `stream` (default: the current one)"""
    e = torch.cuda.Event(enable_timing=timing)
    e.record(stream if stream is not None else cur_stream())
    return e


def _stream():
    """the caller's current HIP stream as a raw handle. cur_stream() builds a Stream object through
    several layers of Python (device index resolution, environment look-ups): 7.8 us per call, 1.4 ms of the eager
    step's ~3 ms of host time (tools/hostprof.py); the two C entry points below take 0.3 us."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return cur_stream().cuda_stream


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor: this build has no CPU path" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def _p(t):
    return None if t is None else t.data_ptr()


_PINNED = {"ring": [], "next": 0}
GAP_EVENTS = None  # debug: list that receives (event in front of, event behind) the training forward's host round trip


def _h2d_int32(arr, device):
    """numpy int32 -> device tensor through a small ring of reusable PINNED staging buffers: a true asynchronous DMA on
    the current stream. (A pageable source goes through the runtime's shared staging pool, whose recycling drains the
    stream every few MB -- measured as periodic 10-50 ms stalls of the training iteration.)"""
    n = int(arr.size)
    ring = _PINNED["ring"]
    if len(ring) < 4:
        ring.append([torch.empty((max(n, 1 << 16),), dtype=torch.int32, pin_memory=True), None])
    i = _PINNED["next"] % len(ring)
    _PINNED["next"] += 1
    buf, ev = ring[i]
    if ev is not None:
        ev.synchronize()  # the copy that last used this buffer (4 uploads ago) has long finished
    if buf.numel() < n:
        buf = ring[i][0] = torch.empty((n * 2,), dtype=torch.int32, pin_memory=True)
    buf[:n].numpy()[...] = arr.reshape(-1)
    out = buf[:n].to(device, non_blocking=True)
    ev = record_event()
    ring[i][1] = ev
    return out


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class W3:
    """a weight after dana_split_weight: three bf16 planes per K-step, [batch][kp/16][3][n][16], of the same fp32 values (exact split), for
    the DANA_W_SPLIT3 flag of the contraction entry points. `t` owns the bytes."""
    __slots__ = ("t", "n", "k", "kp", "batch")

    def __init__(self, t, n, k, kp, batch):
        self.t, self.n, self.k, self.kp, self.batch = t, n, k, kp, batch


def split_weight(w, n, k, batch=1, ldw=0, batch_w=0):
    """fp32 rows [batch][n][k] -> W3 (None when the f32-MFMA kernel is selected: it takes fp32 weights only)"""
    if get_mfma_mode() == 0:
        return None
    _chk(w, "w")
    kp = (k + 15) // 16 * 16
    out = torch.empty(int(lib().query("dana_split_weight_bytes", n, k, batch)), dtype=torch.uint8, device=w.device)
    lib().call("dana_split_weight", _p(w), ldw or k, n, k, batch, batch_w or n * k, _p(out), _stream())
    return W3(out, n, k, kp, batch)


def _wf(weight):
    """(pointer, flag) of a weight argument that is either fp32 rows or a W3"""
    if isinstance(weight, W3):
        return _p(weight.t), W_SPLIT3
    return _p(_chk(weight, "weight")), 0


# ------------------------------------------------------------------------------------------------
# native operators of the reference's `model._C`
# ------------------------------------------------------------------------------------------------
def roi_align_forward(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio):
    """NCHW contract of lib/model/csrc/ROIAlign.h:11 (-> [R, C, PH, PW])."""
    input = _chk(input.contiguous(), "input")
    rois = _chk(rois.contiguous(), "rois")
    B, C, H, W = input.shape
    R = rois.shape[0]
    out = torch.empty((R, C, pooled_h, pooled_w), dtype=input.dtype, device=input.device)
    lib().call("dana_roi_align_forward", _p(input), _p(rois), _p(out), B, C, H, W, R, float(spatial_scale),
               pooled_h, pooled_w, sampling_ratio, NCHW, 0, 0, None, None, 0, _stream())
    return out


def roi_align_forward_nhwc(feat, B, H, W, C, pix_stride, rois, spatial_scale, pooled, sampling_ratio, pe=None,
                           out=None, out_pe=None, out_stride=0, out_pe_stride=0):
    """NHWC fast path: feat is a flat buffer whose pixel (b,y,x) starts at ((b*H+y)*W+x)*pix_stride.
    Returns pooled [R, P*P, C] (and pooled+pe written with row stride out_pe_stride when pe given)."""
    _chk(feat, "feat")
    rois = _chk(rois.contiguous(), "rois")
    R = rois.shape[0]
    P2 = pooled * pooled
    if out is None:
        out = torch.empty((R, P2, C), dtype=torch.float32, device=feat.device)
        out_stride = C
    if pe is not None and out_pe is None:
        out_pe = torch.empty((R, P2, C), dtype=torch.float32, device=feat.device)
        out_pe_stride = C
    lib().call("dana_roi_align_forward", _p(feat), _p(rois), _p(out), B, C, H, W, R, float(spatial_scale), pooled,
               pooled, sampling_ratio, NHWC, pix_stride, out_stride, _p(out_pe) if pe is not None else None,
               _p(pe), out_pe_stride, _stream())
    return out, out_pe


def roi_align_backward(grad, rois, spatial_scale, pooled_h, pooled_w, batch, channels, height, width,
                       sampling_ratio, layout=NCHW):
    grad = _chk(grad.contiguous(), "grad")
    rois = _chk(rois.contiguous(), "rois")
    shape = (batch, channels, height, width) if layout == NCHW else (batch, height, width, channels)
    gin = torch.empty(shape, dtype=grad.dtype, device=grad.device)
    lib().call("dana_roi_align_backward", _p(grad), _p(rois), _p(gin), batch, channels, height, width,
               rois.shape[0], float(spatial_scale), pooled_h, pooled_w, sampling_ratio, layout, _stream())
    return gin


def roi_pool_forward(input, rois, spatial_scale, pooled_h, pooled_w):
    input = _chk(input.contiguous(), "input")
    rois = _chk(rois.contiguous(), "rois")
    B, C, H, W = input.shape
    R = rois.shape[0]
    out = torch.empty((R, C, pooled_h, pooled_w), dtype=input.dtype, device=input.device)
    argmax = torch.empty((R, C, pooled_h, pooled_w), dtype=torch.int32, device=input.device)
    lib().call("dana_roi_pool_forward", _p(input), _p(rois), _p(out), _p(argmax), B, C, H, W, R,
               float(spatial_scale), pooled_h, pooled_w, _stream())
    return out, argmax


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_h, pooled_w, batch, channels, height,
                      width):
    grad = _chk(grad.contiguous(), "grad")
    rois = _chk(rois.contiguous(), "rois")
    argmax = _chk(argmax.contiguous(), "argmax", torch.int32)
    gin = torch.empty((batch, channels, height, width), dtype=grad.dtype, device=grad.device)
    lib().call("dana_roi_pool_backward", _p(grad), _p(argmax), _p(rois), _p(gin), batch, channels, height, width,
               rois.shape[0], pooled_h, pooled_w, _stream())
    return gin


def sort_desc(scores):
    """scores [B, n] -> (order int32 [B, n], sorted scores) ; stable, descending."""
    scores = _chk(scores.contiguous(), "scores")
    B, n = scores.shape
    order = torch.empty((B, n), dtype=torch.int32, device=scores.device)
    sorted_scores = torch.empty_like(scores)
    nbytes = lib().query("dana_sort_desc_workspace_bytes", B, n)
    ws = _ws(nbytes, scores.device)
    lib().call("dana_sort_desc", _p(scores), B, n, _p(order), _p(sorted_scores), _p(ws), ws.numel(), _stream())
    return order, sorted_scores


def topk_desc(scores, topn):
    """scores [B, n] -> (order int32 [B, m], sorted scores [B, m]), m = min(topn, n): the first topn entries of a stable
    descending sort of every row (proposal_layer.py:135-150), one hand-written launch (dana_topk_desc)."""
    scores = _chk(scores.contiguous(), "scores")
    B, n = scores.shape
    m = min(int(topn), n)
    order = torch.empty((B, m), dtype=torch.int32, device=scores.device)
    sorted_scores = torch.empty((B, m), dtype=torch.float32, device=scores.device)
    nbytes = lib().query("dana_topk_desc_workspace_bytes", B, n, int(topn))
    ws = _ws(nbytes, scores.device)
    lib().call("dana_topk_desc", _p(scores), B, n, int(topn), _p(order), m, _p(sorted_scores), _p(ws), ws.numel(), _stream())
    return order, sorted_scores


def nms_sorted(boxes, thr, inclusive=False, max_keep=0):
    """boxes [P, n, 4] already in descending-score order -> (keep int32 [P, mk], num_keep int32 [P])."""
    boxes = _chk(boxes.contiguous(), "boxes")
    P, n, _ = boxes.shape
    mk = n if (max_keep <= 0 or max_keep > n) else max_keep
    keep = torch.empty((P, max(mk, 1)), dtype=torch.int32, device=boxes.device)
    num = torch.empty((P,), dtype=torch.int32, device=boxes.device)
    nbytes = lib().query("dana_nms_workspace_bytes", n, P)
    ws = _ws(nbytes, boxes.device)
    lib().call("dana_nms", _p(boxes), n, P, float(thr), int(bool(inclusive)), mk, _p(keep), max(mk, 1), _p(num),
               _p(ws), ws.numel(), _stream())
    return keep, num


def nms(dets, scores, threshold, inclusive=False):
    """Reference contract (lib/model/csrc/nms.h:10): dets [N,4], scores [N] -> kept ORIGINAL indices,
    int64, ascending. The variable-length result needs one D2H read of the count (as the reference's
    own host scan does); the fused proposal path (proposal_layer) has no host sync."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device="cpu")  # nms.h:17-18
    dets = _chk(dets.contiguous(), "dets")
    scores = _chk(scores.contiguous(), "scores")
    order, _ = sort_desc(scores.view(1, -1))
    boxes = dets[order[0].long()].contiguous().view(1, -1, 4)
    keep, num = nms_sorted(boxes, threshold, inclusive)
    k = int(num[0].item())
    if k == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    kept = order[0][keep[0, :k].long()]
    # ascending ORIGINAL indices (nms.cu:125-131 returns them sorted): this library's own sort on the negated indices
    # (exact in fp32 below 2^24 boxes) -- no torch / rocPRIM sort kernel on the operator's path either
    if dets.size(0) >= (1 << 24):
        return torch.sort(kept.long())[0]
    _, neg_sorted = sort_desc((-kept.float()).view(1, -1))
    return (-neg_sorted[0]).long()


def proposal_layer(cls, cls_strides, cls_is_prob, bbox, bbox_strides, im_info, base_anchors, B, A, H, W,
                   feat_stride, pre_nms_topn, post_nms_topn, nms_thresh, nms_inclusive=False):
    """_ProposalLayer.forward (lib/model/rpn/proposal_layer.py:49-190) in one C call -> rois [B, post, 5]."""
    _chk(cls, "cls")
    _chk(bbox, "bbox")
    im_info = _chk(im_info.contiguous(), "im_info")
    base_anchors = _chk(base_anchors.contiguous(), "base_anchors")
    rois = torch.empty((B, post_nms_topn, 5), dtype=torch.float32, device=cls.device)
    nbytes = lib().query("dana_proposal_layer_workspace_bytes", B, A, H, W, pre_nms_topn, post_nms_topn)
    ws = _ws(nbytes, cls.device)
    lib().call("dana_proposal_layer", _p(cls), cls_strides[0], cls_strides[1], cls_strides[2], int(cls_is_prob),
               _p(bbox), bbox_strides[0], bbox_strides[1], bbox_strides[2], _p(im_info), _p(base_anchors), B, A, H,
               W, feat_stride, pre_nms_topn, post_nms_topn, float(nms_thresh), int(bool(nms_inclusive)), _p(rois),
               _p(ws), ws.numel(), _stream())
    return rois


def rpn_decode(cls, cls_strides, cls_is_prob, bbox, bbox_strides, im_info, base_anchors, B, A, H, W, feat_stride):
    _chk(cls, "cls")
    _chk(bbox, "bbox")
    n = H * W * A
    proposals = torch.empty((B, n, 4), dtype=torch.float32, device=cls.device)
    scores = torch.empty((B, n), dtype=torch.float32, device=cls.device)
    lib().call("dana_rpn_decode", _p(cls), cls_strides[0], cls_strides[1], cls_strides[2], int(cls_is_prob),
               _p(bbox), bbox_strides[0], bbox_strides[1], bbox_strides[2], _p(_chk(im_info.contiguous(), "im_info")),
               _p(_chk(base_anchors.contiguous(), "base_anchors")), B, A, H, W, feat_stride, _p(proposals),
               _p(scores), _stream())
    return proposals, scores


def proposal_target_prepare(rois, gt_boxes, fg_thresh, bg_hi, bg_lo, counts=None):
    """first half of _ProposalTargetLayer (proposal_target_layer_cascade.py:113-141): IoU, assignment, ordered fg / bg
    candidate lists and their counts [B,2] -- everything the sampling needs, nothing random. -> handle"""
    rois = _chk(rois.contiguous(), "rois")
    gt_boxes = _chk(gt_boxes.contiguous(), "gt_boxes")
    B, n_rois, _ = rois.shape
    n_gt = gt_boxes.shape[1]
    n_all = n_rois + n_gt
    dev = rois.device
    max_ov = torch.empty((B, n_all), dtype=torch.float32, device=dev)
    ibuf = torch.empty((3, B, n_all), dtype=torch.int32, device=dev)  # assign, fg_list, bg_list
    if counts is None:
        counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
    lib().call("dana_proposal_target_prepare", _p(rois), _p(gt_boxes), B, n_rois, n_gt, float(fg_thresh), float(bg_hi),
               float(bg_lo), _p(max_ov), _p(ibuf[0]), _p(ibuf[1]), _p(ibuf[2]), _p(counts), _stream())
    return dict(rois=rois, gt_boxes=gt_boxes, B=B, n_rois=n_rois, n_gt=n_gt, n_all=n_all, ibuf=ibuf, max_ov=max_ov,
                counts=counts)


def proposal_target_draw(cnt, B, R, fg_rois_per_image):
    """the reference's np.random draws (proposal_target_layer_cascade.py:143-175) from the host copy of the counts
    -> (picks int32 [B, R], fg_taken int32 [B])"""
    import numpy as np
    picks = np.zeros((B, R), dtype=np.int32)
    taken = np.zeros((B,), dtype=np.int32)
    for i in range(B):
        nf, nb = int(cnt[i, 0]), int(cnt[i, 1])
        if nf > 0 and nb > 0:
            fg_n = min(fg_rois_per_image, nf)
            picks[i, :fg_n] = np.random.permutation(nf)[:fg_n]
            picks[i, fg_n:] = np.floor(np.random.rand(R - fg_n) * nb)
        elif nf > 0:
            fg_n = R
            picks[i] = np.floor(np.random.rand(R) * nf)
        elif nb > 0:
            fg_n = 0
            picks[i] = np.floor(np.random.rand(R) * nb)
        else:
            raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
        taken[i] = fg_n
    return picks, taken


def proposal_target_sample_device(h, R, fg_rois_per_image, seed, offset, counter=None):
    """device Philox draws (no host sync, a different random stream); counter: uint64 device call counter (hipGraphs)"""
    B = h["B"]
    host = torch.empty((B * R + B,), dtype=torch.int32, device=h["rois"].device)
    if counter is None:
        lib().call("dana_proposal_target_sample", _p(h["counts"]), B, h["n_all"], R, fg_rois_per_image, int(seed),
                   int(offset), _p(host), host.data_ptr() + B * R * 4, _stream())
    else:
        lib().call("dana_proposal_target_sample_ctr", _p(h["counts"]), B, h["n_all"], R, fg_rois_per_image, int(seed),
                   int(offset), _p(counter), _p(host), host.data_ptr() + B * R * 4, _stream())
    return host


def proposal_target_finish(h, picks_dev_ptr, taken_dev_ptr, R, means, stds, inside_w, normalize=True):
    """second half (proposal_target_layer_cascade.py:176-213): gather the sampled rois, labels, targets, weights"""
    import ctypes
    dev = h["rois"].device
    B = h["B"]
    rois_out = torch.empty((B, R, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((B, R), dtype=torch.float32, device=dev)
    tgt = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    w_in = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    w_out = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    f4 = ctypes.c_float * 4
    ibuf = h["ibuf"]
    lib().call("dana_proposal_target_gather", _p(h["rois"]), _p(h["gt_boxes"]), B, h["n_rois"], h["n_gt"], _p(ibuf[0]),
               _p(ibuf[1]), _p(ibuf[2]), picks_dev_ptr, taken_dev_ptr, R,
               ctypes.cast(f4(*means), ctypes.c_void_p), ctypes.cast(f4(*stds), ctypes.c_void_p),
               ctypes.cast(f4(*inside_w), ctypes.c_void_p), int(bool(normalize)), _p(rois_out), _p(labels), _p(tgt),
               _p(w_in), _p(w_out), _stream())
    return rois_out, labels, tgt, w_in, w_out


def proposal_target_layer(rois, gt_boxes, rois_per_image, fg_rois_per_image, fg_thresh, bg_hi, bg_lo, means, stds,
                          inside_w, normalize=True, device_rng=None):
    """_ProposalTargetLayer.forward (proposal_target_layer_cascade.py:33-213): two HIP launches around one
    D2H read of the fg/bg counts; the sampling draws np.random exactly like the reference (:143-175).
    device_rng = (seed, offset): draw on the device instead (Philox), no host sync, a different random stream."""
    import numpy as np
    h = proposal_target_prepare(rois, gt_boxes, fg_thresh, bg_hi, bg_lo)
    B, R = h["B"], rois_per_image
    if device_rng is not None:
        host = proposal_target_sample_device(h, R, fg_rois_per_image, device_rng[0], device_rng[1])
    else:
        cnt = h["counts"].cpu().numpy()  # the one host sync: np.random needs the counts
        picks, taken = proposal_target_draw(cnt, B, R, fg_rois_per_image)
        host = _h2d_int32(np.concatenate([picks.reshape(-1), taken]), rois.device)
    return proposal_target_finish(h, host.data_ptr(), host.data_ptr() + B * R * 4, R, means, stds, inside_w, normalize)


def anchor_target_prepare(gt_boxes, im_info, base_anchors, feat_h, feat_w, feat_stride, negative_overlap,
                          positive_overlap, counts=None):
    """_AnchorTargetLayer up to the un-subsampled labels (anchor_target_layer.py:48-136): one C call (IoU / labels /
    ordered lists: three multi-workgroup launches) -> handle with
    labels, max overlaps, argmax, ordered fg / bg lists and their counts [B,2] on the device"""
    gt_boxes = _chk(gt_boxes.contiguous(), "gt_boxes")
    im_info = _chk(im_info.contiguous(), "im_info")
    base_anchors = _chk(base_anchors.contiguous(), "base_anchors")
    B, n_gt, _ = gt_boxes.shape
    A = base_anchors.shape[0]
    total = feat_h * feat_w * A
    dev = gt_boxes.device
    labels = torch.empty((B, total), dtype=torch.float32, device=dev)
    max_ov = torch.empty((B, total), dtype=torch.float32, device=dev)
    ibuf = torch.empty((3, B, total), dtype=torch.int32, device=dev)  # argmax, fg_list, bg_list
    if counts is None:
        counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
    lib().call("dana_anchor_target_prepare", _p(gt_boxes), _p(im_info), _p(base_anchors), B, A, feat_h, feat_w,
               feat_stride, n_gt, float(negative_overlap), float(positive_overlap), _p(labels), _p(max_ov), _p(ibuf[0]),
               _p(ibuf[1]), _p(ibuf[2]), _p(counts), _stream())
    return dict(labels=labels, max_ov=max_ov, argmax=ibuf[0], ibuf=ibuf, gt_boxes=gt_boxes, base_anchors=base_anchors,
                B=B, A=A, H=feat_h, W=feat_w, stride=feat_stride, n_gt=n_gt, total=total, counts=counts, num_examples=1)


def anchor_target_draw(cnt, B, rpn_batchsize, num_fg):
    """the reference's np.random.permutation draws (anchor_target_layer.py:137-156) from the host copy of the counts
    -> (pairs int32 [n, 2] = (image << 1 | is_bg, position in that image's list) to disable, num_examples)"""
    import numpy as np
    which, pos = [], []
    num_examples = 0
    for i in range(B):
        nf, nb = int(cnt[i, 0]), int(cnt[i, 1])
        if nf > num_fg:
            perm = np.random.permutation(nf)[:nf - num_fg]
            which.append(np.full(perm.shape, 2 * i, dtype=np.int32))
            pos.append(perm.astype(np.int32))
        fg_after = min(nf, num_fg)
        num_bg = rpn_batchsize - fg_after
        if nb > num_bg:
            perm = np.random.permutation(nb)[:nb - num_bg]
            which.append(np.full(perm.shape, 2 * i + 1, dtype=np.int32))
            pos.append(perm.astype(np.int32))
        num_examples = fg_after + min(nb, num_bg)  # the LAST image's count is used for all (:156)
    if which:
        pairs = np.stack([np.concatenate(which), np.concatenate(pos)], 1).astype(np.int32)
    else:
        pairs = np.zeros((0, 2), dtype=np.int32)
    return pairs, max(num_examples, 1)


def anchor_target_subsample_device(h, rpn_batchsize, num_fg, seed, offset, counter=None):
    """device Philox subsampling (no host sync); fills h['inv_ne_dev']"""
    inv_ne = torch.empty((1,), dtype=torch.float32, device=h["labels"].device)
    ibuf = h["ibuf"]
    if counter is None:
        lib().call("dana_anchor_target_subsample", _p(h["labels"]), _p(ibuf[1]), _p(ibuf[2]), _p(h["counts"]), h["B"],
                   h["total"], int(rpn_batchsize), num_fg, int(seed), int(offset), _p(inv_ne), _stream())
    else:
        lib().call("dana_anchor_target_subsample_ctr", _p(h["labels"]), _p(ibuf[1]), _p(ibuf[2]), _p(h["counts"]), h["B"],
                   h["total"], int(rpn_batchsize), num_fg, int(seed), int(offset), _p(counter), _p(inv_ne), _stream())
    h["inv_ne_dev"] = inv_ne
    return h


# ---- the ONE host round trip of the training forward -----------------------------------------------------------------
# Both target layers stop at their fg / bg counts (4 * B int32 in one buffer). The host reads them once, draws the
# reference's np.random stream (anchor layer first, as dana.py:166-185 calls them) and sends everything back in ONE
# pinned upload laid out as int32 words:
#   [0] number of anchors to disable   [1] 1 / num_examples (float bits)   [2 .. 2+B*R) picks   [.. +B) fg_taken
#   then (which, pos) pairs, 8-byte aligned. Every consumer reads it through device pointers, so the launches after
#   the draw have step-independent parameters (a captured hipGraph replays them with each step's draws).
def draw_layout(B, R, total):
    """word offsets of the draw buffer: [picks B*R | fg_taken B | (pad) | header: n pairs, 1/num_examples | pairs 2 x n]. The
    proposal part (a few hundred words) and the anchor part (header + up to 2 * B * total words) are two contiguous
    regions: they are uploaded separately, the big one EARLY (draw_and_upload)."""
    hdr = (B * R + B + 1) // 2 * 2
    cap = 2 * B * total  # every fg and every bg anchor of every image could be disabled
    return dict(picks=0, taken=B * R, hdr=hdr, pairs=hdr + 2, cap=cap, words=hdr + 2 + 2 * cap)


def _pinned_upload(arr, dst, stream=None):
    """numpy int32 -> dst (device int32 view) through the ring of reusable PINNED staging buffers: a true asynchronous
    DMA on `stream` (default: the current one)"""
    n = int(arr.size)
    ring = _PINNED["ring"]
    if len(ring) < 6:
        ring.append([torch.empty((max(n, 1 << 16),), dtype=torch.int32, pin_memory=True), None])
    i = _PINNED["next"] % len(ring)
    _PINNED["next"] += 1
    buf, ev = ring[i]
    if ev is not None:
        ev.synchronize()  # the copy that last used this buffer (several uploads ago) has long finished
    if buf.numel() < n:
        buf = ring[i][0] = torch.empty((n * 2,), dtype=torch.int32, pin_memory=True)
    buf[:n].numpy()[...] = arr.reshape(-1)
    with on_stream(stream if stream is not None else cur_stream()):
        dst[:n].copy_(buf[:n], non_blocking=True)
        ev = record_event()
    ring[i][1] = ev
    return ev


def draw_and_upload(req, device, static=None):
    """The training forward's one host round trip: counts -> the reference's np.random draws -> device buffer in
    draw_layout order (`static`: a preallocated int32 device buffer of draw_layout()['words'] words -- hipGraph mode, the
    consumers' pointers are baked into the graph; None: a fresh tensor).

    req: anchor / proposal counts [B,2] on the device, the stream the anchor half ran on, B, R, fg_per, rpn_batchsize,
    num_fg, total. The anchor counts are final long before the proposals are: they are read behind THEIR stream only, the
    anchor layer's draws (permutations over every bg anchor, ~1 ms of host time) and the staging + upload of their result
    -- the list of anchors to disable, ~3 MB at 600 x 1000 -- run while the GPU still works on the trunk, on a copy stream
    of their own. The proposal counts are the one read that waits for the caller's stream; behind it only the proposal
    draws (~20-50 us) and a 2 KB upload stand between the GPU and the rest of the step (round 2 assembled and uploaded
    everything as ONE array after the second read). Measured (tools/sync_gap.py): the GPU idles 0.10 ms at this round
    trip in the eager forward -- and 0.63 ms when the two halves are hipGraphs (the runtime's end-of-graph hand-over, not
    host work: the host spends 70 us there; polling instead of a blocking read changes nothing), which is why the
    three-graph replay of the host-RNG forward loses to eager issue while the one-graph replay of the device-RNG forward
    wins (bench.py: device_rng_one_graph)."""
    import numpy as np
    B, R = req["B"], req["R"]
    lay = draw_layout(B, R, req["total"])
    gap = GAP_EVENTS
    if gap is not None:  # (tools/sync_gap.py: GPU time between the last kernel in front of the round trip and the first behind it)
        e_in = torch.cuda.Event(enable_timing=True)
        e_in.record()
    t0 = _time.perf_counter()
    with on_stream(req["anchor_stream"]):
        cnt_a = req["anchor_counts"].cpu().numpy()
    HOST_WAIT[0] += _time.perf_counter() - t0  # (time the host spent BLOCKED on the GPU, for host-enqueue accounting)
    pairs, num_examples = anchor_target_draw(cnt_a, B, req["rpn_batchsize"], req["num_fg"])
    n = int(pairs.shape[0])
    part_a = np.empty((2 + 2 * n,), dtype=np.int32)
    part_a[0] = n
    part_a[1:2] = np.array([1.0 / num_examples], dtype=np.float32).view(np.int32)
    part_a[2:] = pairs.reshape(-1)
    cs = _PINNED.get("copy_stream")
    if cs is None or cs.device != torch.device(device):
        cs = _PINNED["copy_stream"] = torch.cuda.Stream(device=device)
    cur = cur_stream()
    if static is None:
        # a fresh buffer FROM THE COPY STREAM'S POOL: the upload below starts at once, and a block of the caller's pool
        # may still be in use by a kernel that is queued on the caller's stream (released in stream order only) when
        # that stream lags behind the host -- two processes on one GPU, a long kernel in front (regression:
        # tests/test_gpu_backward.py::test_two_stream_trunk_..._recycled_blocks)
        with on_stream(cs):
            static = torch.empty((lay["pairs"] + 2 * n,), dtype=torch.int32, device=device)
        static.record_stream(cur)
    else:
        # (a static buffer's last reader -- the previous iteration's graph -- finished before this iteration's counts
        # could be read)
        static.record_stream(cs)
    ev_a = _pinned_upload(part_a, static[lay["hdr"]:], stream=cs)
    t0 = _time.perf_counter()
    cnt_p = req["proposal_counts"].cpu().numpy()  # host sync: np.random needs the counts
    HOST_WAIT[0] += _time.perf_counter() - t0
    picks, taken = proposal_target_draw(cnt_p, B, R, req["fg_per"])
    part_p = np.empty((B * R + B,), dtype=np.int32)
    part_p[:B * R] = picks.reshape(-1)
    part_p[B * R:] = taken
    _pinned_upload(part_p, static)
    cur.wait_event(ev_a)
    if gap is not None:
        e_out = torch.cuda.Event(enable_timing=True)
        e_out.record()
        gap.append((e_in, e_out))
    return static


def anchor_target_apply_draws(h, drawn, lay):
    """labels[list[pos]] = -1 for the drawn (which, pos) pairs; 1/num_examples stays in device memory (drawn[1])"""
    ibuf = h["ibuf"]
    base = drawn.data_ptr()
    lib().call("dana_anchor_target_disable_dev", _p(h["labels"]), _p(ibuf[1]), _p(ibuf[2]), base + 4 * lay["hdr"],
               base + 4 * lay["pairs"], lay["cap"], h["total"], _stream())
    h["inv_ne_dev"] = drawn[lay["hdr"] + 1:lay["hdr"] + 2].view(torch.float32)
    h["_drawn"] = drawn  # keeps the buffer alive as long as the handle
    return h


def anchor_target_assign(gt_boxes, im_info, base_anchors, feat_h, feat_w, feat_stride, negative_overlap,
                         positive_overlap, rpn_batchsize, fg_fraction, device_rng=None):
    """_AnchorTargetLayer (anchor_target_layer.py:48-193) up to the sampled labels: one C call, one D2H
    read of the fg/bg counts, the reference's np.random.permutation draws (:137-156), one scatter launch.
    Returns a dict consumed by rpn_losses() / anchor_target_outputs()."""
    import numpy as np
    h = anchor_target_prepare(gt_boxes, im_info, base_anchors, feat_h, feat_w, feat_stride, negative_overlap,
                              positive_overlap)
    num_fg = int(fg_fraction * rpn_batchsize)
    if device_rng is not None:  # subsample on the device (Philox): no host sync, a different random stream
        return anchor_target_subsample_device(h, rpn_batchsize, num_fg, device_rng[0], device_rng[1])
    cnt = h["counts"].cpu().numpy()
    pairs, num_examples = anchor_target_draw(cnt, h["B"], rpn_batchsize, num_fg)
    n = int(pairs.shape[0])
    if n:
        host = _h2d_int32(np.concatenate([pairs[:, 0], pairs[:, 1]]), h["labels"].device)
        lib().call("dana_anchor_target_disable", _p(h["labels"]), _p(h["ibuf"][1]), _p(h["ibuf"][2]), _p(host),
                   host.data_ptr() + 4 * n, n, h["total"], _stream())
    h["num_examples"] = num_examples
    return h


def counter_add_(counter, inc):
    lib().call("dana_counter_add", _p(counter), int(inc), _stream())
    return counter


def anchor_target_outputs(h, inside_weight=1.0):
    """the four `_AnchorTargetLayer` outputs in the reference's layouts, from anchor_target_assign()'s handle"""
    B, A, H, W = h["B"], h["A"], h["H"], h["W"]
    dev = h["labels"].device
    labels = torch.empty((B, 1, A * H, W), dtype=torch.float32, device=dev)
    tgt = torch.empty((B, 4 * A, H, W), dtype=torch.float32, device=dev)
    w_in, w_out = torch.empty_like(tgt), torch.empty_like(tgt)
    lib().call("dana_anchor_target_outputs", _p(h["labels"]), _p(h["max_ov"]), _p(h["argmax"]), _p(h["gt_boxes"]),
               _p(h["base_anchors"]), B, A, H, W, h["stride"], h["n_gt"], float(inside_weight),
               1.0 / h["num_examples"], _p(h.get("inv_ne_dev")), _p(labels), _p(tgt), _p(w_in), _p(w_out), _stream())
    return labels, tgt, w_in, w_out


def rpn_losses(heads, head_row_stride, h, sigma=3.0, inside_weight=1.0):
    """(rpn_loss_cls, rpn_loss_bbox) of rpn.py:97-115, fused over heads[B*H*W][2A | 4A]; -> float32[3] tensor (cls, box, #labeled)"""
    _chk(heads, "heads")
    out = torch.empty((3,), dtype=torch.float32, device=heads.device)  # cls, box, number of labeled anchors
    ws = _ws(lib().query("dana_rpn_loss_workspace_bytes"), heads.device)
    lib().call("dana_rpn_loss", _p(heads), head_row_stride, _p(h["labels"]), _p(h["argmax"]), _p(h["gt_boxes"]),
               _p(h["base_anchors"]), h["B"], h["A"], h["H"], h["W"], h["stride"], h["n_gt"], float(sigma),
               float(inside_weight), 1.0 / h["num_examples"], _p(h.get("inv_ne_dev")), _p(out), _p(ws), ws.numel(),
               _stream())
    return out


def rcnn_losses(score_pos, score_neg, labels, bbox_pred, bbox_targets, inside_ws, outside_ws, sigma=1.0,
                with_grad=False):
    """(RCNN_loss_cls, RCNN_loss_bbox) of dana.py:199-217 in one fused pass, no host sync -> (float32[3] tensor
    (cls, box, rows kept), seeds) with seeds = (d cls / d score_pos, d cls / d score_neg, d box / d bbox_pred) or None"""
    n = score_pos.shape[0]
    dev = score_pos.device
    out = torch.empty((3,), dtype=torch.float32, device=dev)
    seeds = None
    if with_grad:
        seeds = (torch.empty((n, 2), dtype=torch.float32, device=dev), torch.empty((n, 2), dtype=torch.float32, device=dev),
                 torch.empty((n, 4), dtype=torch.float32, device=dev))
    ws = _ws(lib().query("dana_rcnn_loss_workspace_bytes", n), dev)
    lib().call("dana_rcnn_loss", _p(_chk(score_pos, "score_pos")), _p(_chk(score_neg, "score_neg")),
               _p(_chk(labels, "labels")), _p(_chk(bbox_pred, "bbox_pred")), _p(_chk(bbox_targets, "bbox_targets")),
               _p(_chk(inside_ws, "inside_ws")), _p(_chk(outside_ws, "outside_ws")), n, float(sigma), _p(out),
               _p(seeds[0]) if seeds else None, _p(seeds[1]) if seeds else None, _p(seeds[2]) if seeds else None,
               _p(ws), ws.numel(), _stream())
    return out, seeds


def plain_rcnn_losses(scores, labels, bbox_pred, bbox_targets, inside_ws, outside_ws, sigma=1.0, with_grad=False):
    """F.cross_entropy(scores, labels) and _smooth_l1_loss(bbox_pred, ...) of faster_rcnn.py:93-98 in one launch ->
    (float32[2] tensor, (d cls / d scores, d box / d bbox_pred) or None)"""
    n, c = scores.shape
    dev = scores.device
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    seeds = (torch.empty((n, c), dtype=torch.float32, device=dev), torch.empty((n, 4), dtype=torch.float32, device=dev)) \
        if with_grad else None
    labels = labels.contiguous()
    if labels.dtype != torch.int64:
        raise TypeError("plain_rcnn_losses: labels must be int64")
    lib().call("dana_plain_rcnn_loss", _p(_chk(scores, "scores")), _p(labels), _p(_chk(bbox_pred.contiguous(), "bbox_pred")),
               _p(_chk(bbox_targets.contiguous(), "bbox_targets")), _p(_chk(inside_ws.contiguous(), "inside_ws")),
               _p(_chk(outside_ws.contiguous(), "outside_ws")), n, c, float(sigma), _p(out),
               _p(seeds[0]) if seeds else None, _p(seeds[1]) if seeds else None, _stream())
    return out, seeds


# ------------------------------------------------------------------------------------------------
# dense contractions
# ------------------------------------------------------------------------------------------------
def conv2d_nhwc(x, batch, in_h, in_w, cin, weight, cout, kh, kw, stride, pad, scale=None, shift=None,
                residual=None, relu=False, in_stride=0, out=None, out_stride=0, res_stride=0, stem=False):
    """x: flat NHWC buffer; weight packed [cout][kh][kw][cin]. Returns (out, OH, OW)."""
    _chk(x, "x")
    wp, wfl = _wf(weight)
    oh = (in_h + 2 * pad - kh) // stride + 1
    ow = (in_w + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((batch * oh * ow, cout), dtype=torch.float32, device=x.device)
        out_stride = cout
    flags = (EPI_RELU if relu else 0) | (CONV_STEM7 if stem else 0) | wfl
    e0 = _prof_begin()
    lib().call("dana_conv2d_nhwc", _p(x), wp, _p(out), _p(scale), _p(shift), _p(residual), batch, in_h,
               in_w, cin, cout, kh, kw, stride, pad, in_stride, out_stride, res_stride, flags, _stream())
    # algorithmic flops: the stem counts its 3 real channels x 49 real taps, not the padded K=224
    _prof_end(e0, ("conv%dx%d M=%d N=%d K=%d s%d", (kh, kw, batch * oh * ow, cout, kh * kw * cin, stride)),
              2.0 * batch * oh * ow * cout * kh * kw * (3 if stem else cin),
              4.0 * (batch * in_h * in_w * cin // (stride * stride if kh == 1 else 1) + cout * kh * kw * cin
                     + batch * oh * ow * cout * (2 if residual is not None else 1)))
    return out, oh, ow


def bottleneck_tail(x, batch, h, w, cin, w2, scale2, shift2, w3, scale3, shift3, cout, residual=None, relu=True, in_stride=0,
                    out=None, out_stride=0, res_stride=0):
    """conv2 (3x3 / 1 / pad 1, cin -> 64) + bn2 + ReLU + conv3 (1x1, 64 -> cout) + bn3 + residual + ReLU of a Bottleneck
    (resnet.py:92-100) as ONE launch; w2 / w3 are W3 split planes. Returns (out, h, w)."""
    _chk(x, "x")
    if not (isinstance(w2, W3) and isinstance(w3, W3)):
        raise TypeError("bottleneck_tail: both weights must be split planes (ops.split_weight)")
    if out is None:
        out = torch.empty((batch * h * w, cout), dtype=torch.float32, device=x.device)
        out_stride = cout
    e0 = _prof_begin()
    lib().call("dana_bottleneck_tail_nhwc", _p(x), _p(w2.t), _p(scale2), _p(shift2), _p(w3.t), _p(scale3), _p(shift3),
               _p(residual), _p(out), batch, h, w, cin, 64, cout, in_stride, out_stride, res_stride, EPI_RELU if relu else 0,
               _stream())
    m = batch * h * w
    _prof_end(e0, ("conv3x3+1x1 M=%d N=%d K=%d+64 s1", (m, cout, 9 * cin)), 2.0 * m * 64 * (9 * cin + cout),
              4.0 * (m * cin + 64 * 9 * cin + 64 * cout + m * cout * (2 if residual is not None else 1)))
    return out, h, w


def conv2d_nhwc_dual(x, n0, h0, w0, n1, h1, w1, cin, weight, cout, kh, kw, stride, pad, scale=None, shift=None,
                     res0=None, res1=None, relu=False, in_stride=0, out0=None, out1=None, out0_stride=0,
                     out1_stride=0, res0_stride=0, res1_stride=0, stem=False):
    """One launch over two image groups (query batch + support batch); x holds group 0's pixels then group 1's.
    Without out0/out1 the result is ONE merged buffer [M0 + M1][cout]. Returns (out0, out1, (oh0, ow0), (oh1, ow1))."""
    _chk(x, "x")
    wp, wfl = _wf(weight)
    oh0, ow0 = (h0 + 2 * pad - kh) // stride + 1, (w0 + 2 * pad - kw) // stride + 1
    oh1, ow1 = (h1 + 2 * pad - kh) // stride + 1, (w1 + 2 * pad - kw) // stride + 1
    m0, m1 = n0 * oh0 * ow0, n1 * oh1 * ow1
    if out0 is None:
        merged = torch.empty((m0 + m1, cout), dtype=torch.float32, device=x.device)
        out0, out1 = merged, merged[m0:]
        out0_stride = out1_stride = cout
    flags = (EPI_RELU if relu else 0) | (CONV_STEM7 if stem else 0) | wfl
    e0 = _prof_begin()
    lib().call("dana_conv2d_nhwc_dual", _p(x), wp, _p(out0), _p(out1), _p(scale), _p(shift), _p(res0),
               _p(res1), n0, h0, w0, n1, h1, w1, cin, cout, kh, kw, stride, pad, in_stride, out0_stride, out1_stride,
               res0_stride, res1_stride, flags, _stream())
    _prof_end(e0, ("conv%dx%d M=%d N=%d K=%d s%d", (kh, kw, m0 + m1, cout, kh * kw * cin, stride)),
              2.0 * (m0 + m1) * cout * kh * kw * (3 if stem else cin))
    return out0, out1, (oh0, ow0), (oh1, ow1)


def pack_cat2_weight(w0, s0, b0, k0, w1, s1, b1, k1, cout):
    """-> (w_cat [cout][k0+k1] with the two frozen-BN scales folded into the rows, shift = b0 + b1)"""
    w_cat = torch.empty((cout, k0 + k1), dtype=torch.float32, device=w0.device)
    shift = torch.empty((cout,), dtype=torch.float32, device=w0.device)
    lib().call("dana_pack_cat2_weight", _p(_chk(w0, "w0")), _p(s0), _p(b0), k0, _p(_chk(w1, "w1")), _p(s1), _p(b1), k1, cout,
               _p(w_cat), _p(shift), _stream())
    return w_cat, shift


def conv1x1_cat2(a0, k0, a1, k1, batch, h1, w1, stride1, w_cat, shift, cout, relu=True, a0_stride=0, a1_stride=0,
                 out=None, out_stride=0):
    """relu(a0 . w_cat[:, :k0]^T + a1[strided pixels] . w_cat[:, k0:]^T + shift): a bottleneck's expand conv and its
    downsample conv as one contraction (include/dana_hip.h: dana_conv1x1_cat2_nhwc). -> (out, oh, ow)"""
    _chk(a0, "a0")
    _chk(a1, "a1")
    oh, ow = (h1 - 1) // stride1 + 1, (w1 - 1) // stride1 + 1
    if out is None:
        out = torch.empty((batch * oh * ow, cout), dtype=torch.float32, device=a0.device)
        out_stride = cout
    e0 = _prof_begin()
    wp, wfl = _wf(w_cat)
    lib().call("dana_conv1x1_cat2_nhwc", _p(a0), a0_stride, k0, _p(a1), a1_stride, k1, batch, h1, w1, stride1, wp,
               _p(out), None, _p(shift), None, out_stride, 0, cout, (EPI_RELU if relu else 0) | wfl, _stream())
    m = batch * oh * ow
    _prof_end(e0, ("conv1x1cat M=%d N=%d K=%d s%d", (m, cout, k0 + k1, stride1)), 2.0 * m * cout * (k0 + k1),
              4.0 * (m * (k0 + k1) + cout * (k0 + k1) + m * cout))
    return out, oh, ow


def conv1x1_cat2_dual(a0, k0, a1, k1, n0, h0, w0, n1, h1, w1, stride1, w_cat, shift, cout, relu=True, out0=None,
                      out1=None, out0_stride=0, out1_stride=0):
    """conv1x1_cat2 over two image groups (query batch + support batch) in one launch: a0 = [M0 + M1][k0] on the output
    grids, a1 = group 0's [n0][h0][w0][k1] pixels then group 1's. Without out0/out1 the result is ONE merged buffer.
    -> (out0, out1, (oh0, ow0), (oh1, ow1))"""
    _chk(a0, "a0")
    _chk(a1, "a1")
    oh0, ow0 = (h0 - 1) // stride1 + 1, (w0 - 1) // stride1 + 1
    oh1, ow1 = (h1 - 1) // stride1 + 1, (w1 - 1) // stride1 + 1
    m0, m1 = n0 * oh0 * ow0, n1 * oh1 * ow1
    if out0 is None:
        merged = torch.empty((m0 + m1, cout), dtype=torch.float32, device=a0.device)
        out0, out1 = merged, merged[m0:]
        out0_stride = out1_stride = cout
    e0 = _prof_begin()
    wp, wfl = _wf(w_cat)
    lib().call("dana_conv1x1_cat2_nhwc_dual", _p(a0), 0, k0, _p(a1), 0, k1, n0, h0, w0, n1, h1, w1, stride1, wp,
               _p(out0), _p(out1), None, _p(shift), out0_stride, out1_stride, cout, (EPI_RELU if relu else 0) | wfl,
               _stream())
    m = m0 + m1
    _prof_end(e0, ("conv1x1cat M=%d N=%d K=%d s%d", (m, cout, k0 + k1, stride1)), 2.0 * m * cout * (k0 + k1),
              4.0 * (m * (k0 + k1) + cout * (k0 + k1) + m * cout))
    return out0, out1, (oh0, ow0), (oh1, ow1)


def conv3x3_winograd_dual(x, n0, h0, w0, n1, h1, w1, cin, u, cout, scale=None, shift=None, relu=False, mask=None, out=None):
    """stride-1 pad-1 3x3 conv through Winograd F(4x4,3x3) over two image groups with ONE batched plane GEMM (x: group
    0's pixels, then group 1's) -> merged [M0 + M1][cout]. mask [M0 + M1][cout]: the ReLU adjoint of a data gradient
    (zero where mask <= 0), as in conv3x3_winograd(mask=...)."""
    _chk(x, "x")
    up, wfl = _wf(u)
    if (u.batch if isinstance(u, W3) else u.size(0)) != 36:
        raise ValueError("conv3x3_winograd_dual: F(4x4,3x3) filters only")
    m0, m1 = n0 * h0 * w0, n1 * h1 * w1
    if out is None:
        out = torch.empty((m0 + m1, cout), dtype=torch.float32, device=x.device)
    ws = _ws(lib().query("dana_conv3x3_winograd4_dual_workspace_bytes", n0, h0, w0, n1, h1, w1, cin, cout), x.device)
    e0 = _prof_begin()
    lib().call("dana_conv3x3_winograd4_nhwc_dual_masked", _p(x), up, _p(out), _p(out[m0:]), _p(scale), _p(shift),
               _p(mask), _p(mask[m0:]) if mask is not None else None, n0, h0, w0, n1, h1, w1, cin, cout, 0, 0, 0, 0, 0,
               (EPI_RELU if relu else 0) | wfl, _p(ws), ws.numel(), _stream())
    tiles = n0 * ((h0 + 3) // 4) * ((w0 + 3) // 4) + n1 * ((h1 + 3) // 4) * ((w1 + 3) // 4)
    _prof_end(e0, ("wino3x3 M=%d N=%d K=%d s1", (m0 + m1, cout, 9 * cin)), 2.0 * (m0 + m1) * cout * 9 * cin,
              4.0 * 36 * (tiles * (cin + cout) + cout * cin), executed=2.0 * 36 * tiles * cin * cout)
    return out


def conv3x3_winograd_dual_dgrad(grad_out, n0, h0, w0, n1, h1, w1, cout, ud, cin, mask=None, out=None):
    """data gradient of a stride-1 pad-1 3x3 conv over two image groups (the forward kernel on the flipped / transposed
    filters ud, as conv2d_dgrad does per group), labelled as a data gradient in the per-launch profile"""
    global PROF_ROLE
    if PROFILE is not None and PROF_ROLE is None:
        PROF_ROLE = "dgrad"
        try:
            return conv3x3_winograd_dual(grad_out, n0, h0, w0, n1, h1, w1, cout, ud, cin, mask=mask, out=out)
        finally:
            PROF_ROLE = None
    return conv3x3_winograd_dual(grad_out, n0, h0, w0, n1, h1, w1, cout, ud, cin, mask=mask, out=out)


def winograd_filter_transform(w_packed, cout, cin, tile=2):
    """U of Winograd F(tile x tile, 3x3): [16][cout][cin] (tile 2) or [36][cout][cin] (tile 4)"""
    _chk(w_packed, "w_packed")
    planes = (tile + 2) * (tile + 2)
    u = torch.empty((planes, cout, cin), dtype=torch.float32, device=w_packed.device)
    lib().call("dana_winograd_filter_transform" if tile == 2 else "dana_winograd4_filter_transform", _p(w_packed), _p(u),
               cout, cin, _stream())
    return u


def conv3x3_winograd(x, batch, h, w, cin, u, cout, scale=None, shift=None, relu=False, in_stride=0, out=None,
                     out_stride=0, mask=None, mask_stride=0, keep_v=None):
    """stride-1 pad-1 3x3 conv through Winograd F(2x2,3x3) or F(4x4,3x3), chosen by u (winograd_filter_transform).
    keep_v: a list that receives the launch's workspace (F(4x4) only): its first 36*tiles*cin floats are the input's
    transform V, which conv3x3_wgrad_winograd(v=...) takes instead of transforming the input again."""
    _chk(x, "x")
    up, wfl = _wf(u)
    m = 2 if (u.batch if isinstance(u, W3) else u.size(0)) == 16 else 4
    if wfl and m != 4:
        raise ValueError("conv3x3_winograd: split filters with F(4x4,3x3) only")
    if out is None:
        out = torch.empty((batch * h * w, cout), dtype=torch.float32, device=x.device)
        out_stride = cout
    sfx = "" if m == 2 else "4"
    ws = _ws(lib().query("dana_conv3x3_winograd%s_workspace_bytes" % sfx, batch, h, w, cin, cout), x.device)
    e0 = _prof_begin()
    lib().call("dana_conv3x3_winograd%s_nhwc_masked" % sfx, _p(x), up, _p(out), _p(scale), _p(shift), _p(mask), batch,
               h, w, cin, cout, in_stride, out_stride, mask_stride, (EPI_RELU if relu else 0) | wfl, _p(ws), ws.numel(),
               _stream())
    if keep_v is not None and m == 4:
        keep_v.append(ws)
    _prof_end(e0, ("wino3x3 M=%d N=%d K=%d s1", (batch * h * w, cout, 9 * cin)), 2.0 * batch * h * w * cout * 9 * cin,
              # bytes of the batched GEMM launch itself: V[planes][tiles][cin], U[planes][cout][cin], M[planes][tiles][cout]
              4.0 * (m + 2) * (m + 2) * (batch * ((h + m - 1) // m) * ((w + m - 1) // m) * (cin + cout) + cout * cin),
              # what the matrix cores execute: (m+2)^2 plane GEMMs of [tiles x cin] . [cin x cout]
              executed=2.0 * (m + 2) * (m + 2) * batch * ((h + m - 1) // m) * ((w + m - 1) // m) * cin * cout)
    return out, h, w


def set_mfma_mode(mode):
    """0: fp32 contractions on v_mfma_f32_32x32x2_f32; 1 (default): exact bf16x3 split, six products on the bf16 matrix
    cores with fp32 accumulation (include/dana_hip.h: dana_set_mfma_mode). Returns the previous mode."""
    prev = lib().query("dana_get_mfma_mode")
    lib().call("dana_set_mfma_mode", int(mode))
    return prev


def get_mfma_mode():
    return lib().query("dana_get_mfma_mode")


def force_tile(tile):
    """debug (include/dana_hip_debug.h: dana_debug_force_tile): 0 = the dispatcher's tile choice, 2 = 128x64, 3 = 64x64,
    4 = 128x128, 5 = 64x128 for every split-kernel launch that can take it."""
    lib().call("dana_debug_force_tile", int(tile))


def cumask_stream(cus_per_xcd, n_xcd=8, cus_in_xcd=32):
    """debug: a torch stream restricted to CUs `cus_per_xcd` (an iterable of CU indices 0..31) of EVERY XCD
    (include/dana_hip_debug.h: dana_debug_stream_create_cumask; mask bit i = CU i // n_xcd of XCD i % n_xcd)."""
    import ctypes
    words = (n_xcd * cus_in_xcd + 31) // 32
    mask = (ctypes.c_uint * words)()
    for c in cus_per_xcd:
        for x in range(n_xcd):
            i = int(c) * n_xcd + x
            mask[i >> 5] |= 1 << (i & 31)
    out = ctypes.c_void_p()
    lib().call("dana_debug_stream_create_cumask", ctypes.cast(mask, ctypes.c_void_p), words, n_xcd,
               ctypes.cast(ctypes.byref(out), ctypes.c_void_p))
    return torch.cuda.ExternalStream(out.value)


def set_epilogue_mode(mode):
    """0 (default): the split kernel's epilogue runs on the accumulator registers; 1: through an LDS C tile (the round-1..4
    form, kept as the bit-identity reference: include/dana_hip_debug.h dana_set_epilogue_mode). Returns the previous mode."""
    prev = lib().query("dana_get_epilogue_mode")
    lib().call("dana_set_epilogue_mode", int(mode))
    return prev


SPLIT_K = True  # split-K for tile-starved GEMMs (few tiles, long K); tools flip it for A/Bs


def _splitk_slices(m, n, k, batch):
    """how many K slices (1 = no split): only when the 64x64-tile grid leaves most of the 256 CUs' slots empty"""
    if batch != 1 or n % 4 or n <= 8 or k < 512 or k % 16:
        return 1
    tiles = ((m + 63) // 64) * ((n + 63) // 64)
    if tiles >= 512:
        return 1
    s = max(1, min(8, round(768.0 / tiles)))
    steps = k // 16
    while s > 1 and (steps % s or k // s < 256):
        s -= 1
    return s


def gemm_nt(a, b, m, n, k, lda=0, ldb=0, out=None, ldc=0, scale=None, shift=None, residual=None, ldr=0, batch=1,
            batch_a=0, batch_b=0, batch_c=0, alpha=1.0, relu=False, k_true=0, force_slices=0):
    """c[z][m][n] = epi(alpha * a[z][m][:k] . b[z][n][:k]); both operands K-contiguous. `a` may be a W3 too (the activation
    rows as split planes, ops.split_weight(x, m, k): planes x planes kernel, no split in the K loop; needs b as a W3)."""
    bp, wfl = _wf(b)
    if isinstance(a, W3):
        if not wfl or a.n != m or a.k != k or batch != a.batch:
            raise ValueError("gemm_nt: split activation planes need split weights and matching m / k / batch")
        out = torch.empty((batch, m, n) if batch > 1 else (m, n), dtype=torch.float32, device=a.t.device) if out is None else out
        ldc = ldc or n
        e0 = _prof_begin()
        lib().call("dana_gemm_nt", _p(a.t), bp, _p(out), _p(scale), _p(shift), _p(residual), m, n, k, a.kp, b.kp, ldc,
                   ldr, batch, 3 * m * a.kp, 3 * n * b.kp, batch_c or m * n, float(alpha), (EPI_RELU if relu else 0) | wfl | A_SPLIT3,
                   _stream())
        _prof_end(e0, ("gemm M=%d N=%d K=%d b%d", (m, n, k, batch)), 2.0 * batch * m * n * (k_true or k),
                  batch * (6.0 * (m * k + n * k) + 4.0 * m * n * (2 if residual is not None else 1)))
        return out
    _chk(a, "a")
    lda = lda or k
    if wfl:
        if ldb not in (0, k) or batch != 1 or b.n != n or b.k != k:
            raise ValueError("gemm_nt: a split weight is used whole ([n][k] as split)")
        ldb = b.kp
    else:
        ldb = ldb or k
    if out is None:
        out = torch.empty((batch, m, n) if batch > 1 else (m, n), dtype=torch.float32, device=a.device)
        ldc = n
        batch_c = m * n
    e0 = _prof_begin()
    slices = (force_slices or _splitk_slices(m, n, k, batch)) if SPLIT_K else 1
    if wfl and slices > 1 and (k // slices) % 16:
        slices = 1
    if slices > 1:
        # too few output tiles to fill the chip and a long K: the K range is cut into `slices` batched launches-in-one
        # (blockIdx.z walks K) into fp32 slabs, summed in slice order with the epilogue by a second small kernel
        kc = k // slices
        part = torch.empty((slices, m, n), dtype=torch.float32, device=a.device)
        # (split planes are K-step-major: slice z's first K-step starts kc * 3 * n bf16 elements after slice z-1's)
        lib().call("dana_gemm_nt", _p(a), bp, _p(part), None, None, None, m, n, kc, lda, ldb, n, 0, slices, kc,
                   kc * 3 * n if wfl else kc, m * n, 1.0, wfl, _stream())
        lib().call("dana_splitk_reduce", _p(part), slices, m, n, _p(out), ldc, _p(scale), _p(shift), _p(residual), ldr,
                   float(alpha), EPI_RELU if relu else 0, _stream())
    else:
        lib().call("dana_gemm_nt", _p(a), bp, _p(out), _p(scale), _p(shift), _p(residual), m, n, k, lda, ldb, ldc,
                   ldr, batch, batch_a, batch_b, batch_c, float(alpha), (EPI_RELU if relu else 0) | wfl, _stream())
    _prof_end(e0, ("gemm M=%d N=%d K=%d b%d", (m, n, k, batch)), 2.0 * batch * m * n * (k_true or k),
              4.0 * batch * (m * k + n * k + m * n * (2 if residual is not None else 1)))
    return out


# ------------------------------------------------------------------------------------------------
# layout / pooling / packing
# ------------------------------------------------------------------------------------------------
def nchw_to_nhwc(x, cpad=None, out=None, out_stride=0):
    x = _chk(x.contiguous(), "x")
    B, C, H, W = x.shape
    cpad = cpad or C
    if out is None:
        out = torch.empty((B, H, W, cpad), dtype=torch.float32, device=x.device)
    lib().call("dana_nchw_to_nhwc", _p(x), _p(out), B, C, H, W, cpad, out_stride, _stream())
    return out


def nhwc_to_nchw(x, B, C, H, W, in_stride=0):
    _chk(x, "x")
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    lib().call("dana_nhwc_to_nchw", _p(x), _p(out), B, C, H, W, in_stride, _stream())
    return out


def maxpool_out_size(H, W):
    oh = (H - 3 + 1) // 2 + 1
    ow = (W - 3 + 1) // 2 + 1
    if (oh - 1) * 2 >= H:
        oh -= 1
    if (ow - 1) * 2 >= W:
        ow -= 1
    return oh, ow


def maxpool3x3s2_ceil(x, B, H, W, C, out=None):
    _chk(x, "x")
    oh, ow = maxpool_out_size(H, W)
    if out is None:
        out = torch.empty((B * oh * ow, C), dtype=torch.float32, device=x.device)
    lib().call("dana_maxpool3x3s2_ceil_nhwc", _p(x), _p(out), B, H, W, C, _stream())
    return out, oh, ow


def depthwise_corr(feat, kernels, n_maps, H, W, C, kh, kw, maps_per_kernel=1, feat_stride=0):
    """F.conv2d(feat, kernel.view(C,1,kh,kw), groups=C) in NHWC: -> ([n_maps*(H-kh+1)*(W-kw+1)][C], oh, ow)"""
    oh, ow = H - kh + 1, W - kw + 1
    out = torch.empty((n_maps * oh * ow, C), dtype=torch.float32, device=feat.device)
    lib().call("dana_depthwise_corr_nhwc", _p(_chk(feat, "feat")), _p(_chk(kernels, "kernels")), _p(out), n_maps, H, W, C,
               kh, kw, maps_per_kernel, feat_stride, _stream())
    return out, oh, ow


def depthwise_corr_backward(grad_out, feat, kernels, n_maps, H, W, C, kh, kw, maps_per_kernel=1, feat_stride=0,
                            need_feat=True, grad_kernels=None):
    """adjoints of depthwise_corr: -> (grad_feat [n_maps*H*W][C] or None, grad_kernels [kernels][kh*kw][C], accumulated
    into `grad_kernels` when that is given)"""
    gf = torch.empty((n_maps * H * W, C), dtype=torch.float32, device=grad_out.device) if need_feat else None
    acc = grad_kernels is not None
    if grad_kernels is None:
        nk = (n_maps + maps_per_kernel - 1) // maps_per_kernel
        grad_kernels = torch.empty((nk, kh * kw, C), dtype=torch.float32, device=grad_out.device)
    lib().call("dana_depthwise_corr_backward_nhwc", _p(_chk(grad_out, "grad_out")), _p(_chk(feat, "feat")),
               _p(_chk(kernels, "kernels")), _p(gf), _p(grad_kernels), n_maps, H, W, C, kh, kw, maps_per_kernel, feat_stride,
               int(acc), _stream())
    return gf, grad_kernels


def batch_stats(x, rows, channels, ld=0):
    """per-channel (mean, biased variance) over the rows: what nn.BatchNorm2d normalises with in train mode"""
    _chk(x, "x")
    mean = torch.empty((channels,), dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    ws = _ws(lib().query("dana_colsum_workspace_bytes", rows, channels), x.device)
    lib().call("dana_batch_stats", _p(x), _p(mean), _p(var), rows, channels, ld, _p(ws), ws.numel(), _stream())
    return mean, var


def bn_train_backward(grad_out, x, mean, var, gamma, eps, rows, channels, grad_gamma, grad_beta):
    """adjoint of a train-mode BatchNorm on NHWC rows; dgamma / dbeta are ACCUMULATED into grad_gamma / grad_beta -> grad_x"""
    gx = torch.empty((rows, channels), dtype=torch.float32, device=x.device)
    lib().call("dana_bn_train_backward", _p(_chk(grad_out, "grad_out")), _p(_chk(x, "x")), _p(mean), _p(var),
               _p(_chk(gamma.detach().contiguous(), "gamma")), float(eps), rows, channels, _p(gx), _p(grad_gamma),
               _p(grad_beta), 1, _stream())
    return gx


def scale_shift_relu_(x, scale, shift, rows, channels, relu=True):
    lib().call("dana_scale_shift_relu", _p(_chk(x, "x")), _p(_chk(scale, "scale")), _p(_chk(shift, "shift")), rows,
               channels, int(bool(relu)), _stream())
    return x


def maxpool2x2s2(x, B, H, W, C):
    _chk(x, "x")
    out = torch.empty((B * (H // 2) * (W // 2), C), dtype=torch.float32, device=x.device)
    lib().call("dana_maxpool2x2s2_nhwc", _p(x), _p(out), B, H, W, C, _stream())
    return out, H // 2, W // 2


def maxpool2x2s2_backward(x, grad_out, B, H, W, C):
    """adjoint of maxpool2x2s2 w.r.t. its input x [B*H*W][C]; grad_out [B*(H//2)*(W//2)][C]"""
    gin = torch.empty((B * H * W, C), dtype=torch.float32, device=x.device)
    lib().call("dana_maxpool2x2s2_backward_nhwc", _p(_chk(x, "x")), _p(_chk(grad_out, "grad_out")), _p(gin), B, H, W, C,
               _stream())
    return gin


def sigmoid_(x):
    lib().call("dana_sigmoid", _p(_chk(x, "x")), x.numel(), _stream())
    return x


def scale_rows_by_group(x, vec, rows, rows_per_group, channels):
    """out[r] = x[r] * vec[r // rows_per_group] (channel-wise)"""
    out = torch.empty((rows, channels), dtype=torch.float32, device=x.device)
    lib().call("dana_scale_rows_by_group", _p(_chk(x, "x")), _p(_chk(vec, "vec")), _p(out), rows, rows_per_group,
               channels, _stream())
    return out


def avgpool(x, B, H, W, C, k, stride):
    _chk(x, "x")
    oh, ow = (H - k) // stride + 1, (W - k) // stride + 1
    out = torch.empty((B, oh * ow, C), dtype=torch.float32, device=x.device)
    lib().call("dana_avgpool_nhwc", _p(x), _p(out), B, H, W, C, k, stride, _stream())
    return out


def spatial_mean(x, groups, positions, channels, in_stride=0):
    _chk(x, "x")
    out = torch.empty((groups, channels), dtype=torch.float32, device=x.device)
    lib().call("dana_spatial_mean_nhwc", _p(x), _p(out), groups, positions, channels, in_stride, _stream())
    return out


def add_pe(x, pe, rows, length, channels, in_stride=0, out=None, out_stride=0):
    _chk(x, "x")
    _chk(pe, "pe")
    if out is None:
        out = torch.empty((rows, channels), dtype=torch.float32, device=x.device)
    lib().call("dana_add_pe", _p(x), _p(pe), _p(out), rows, length, channels, in_stride, out_stride, _stream())
    return out


def add_pe_groups(x, pe, groups, rows_per_group, length, channels, in_group_stride, out):
    """out[g][r] = x[g * in_group_stride + r * channels ...] + pe[r % length] for every group in one launch"""
    lib().call("dana_add_pe_groups", _p(_chk(x, "x")), _p(_chk(pe, "pe")), _p(_chk(out, "out")), groups, rows_per_group, length,
               channels, in_group_stride, rows_per_group * channels, _stream())
    return out


def colmean_sub_(x, groups, length, dim, ld=0):
    _chk(x, "x")
    ws = _ws(lib().query("dana_colmean_sub_workspace_bytes", groups, length, dim), x.device)
    lib().call("dana_colmean_sub", _p(x), groups, length, dim, ld, _p(ws), ws.numel(), _stream())
    return x


def transpose_batched(x, groups, rows, cols, ldi=0, out=None, ldo=0, in_batch=0, out_batch=0):
    _chk(x, "x")
    ldi = ldi or cols
    ldo = ldo or rows
    in_batch = in_batch or rows * ldi
    out_batch = out_batch or cols * ldo
    if out is None:  # (the kernel writes the zero tail of padded rows itself)
        out = torch.empty((groups, cols, ldo), dtype=torch.float32, device=x.device)
    lib().call("dana_transpose_batched", _p(x), _p(out), groups, rows, cols, ldi, ldo, in_batch, out_batch, _stream())
    return out


def packed_view(w):
    """the [O][KH*KW*I] kernel layout of a conv weight / gradient as a VIEW, when the tensor is stored that way
    (channels-last strides: the trainer keeps trainable conv weights and their gradients in the kernels' own layout);
    None otherwise"""
    if w.dim() != 4:
        return None
    O, I, KH, KW = w.shape
    if w.stride() != (I * KH * KW, 1, KW * I, I) and not (KH == 1 and KW == 1 and w.is_contiguous()):
        return None
    return w.detach().permute(0, 2, 3, 1).reshape(O, KH * KW * I) if (KH, KW) != (1, 1) else w.detach().reshape(O, I)


def pack_conv_weight(w, stem=False):
    if not stem:
        v = packed_view(w)
        if v is not None and v.data_ptr() == w.data_ptr():
            return _chk(v, "weight")
    w = _chk(w.detach().contiguous(), "weight")
    O, I, KH, KW = w.shape
    out = torch.empty((O, 7 * 8 * 4) if stem else (O, KH * KW * I), dtype=torch.float32, device=w.device)
    lib().call("dana_pack_conv_weight", _p(w), _p(out), O, I, KH, KW, int(stem), _stream())
    return out


def bn_fold(bn_weight, bn_bias, running_mean, running_var, eps):
    n = bn_weight.numel()
    scale = torch.empty((n,), dtype=torch.float32, device=bn_weight.device)
    shift = torch.empty_like(scale)
    lib().call("dana_bn_fold", _p(_chk(bn_weight.detach().contiguous(), "gamma")),
               _p(_chk(bn_bias.detach().contiguous(), "beta")), _p(_chk(running_mean.contiguous(), "mean")),
               _p(_chk(running_var.contiguous(), "var")), float(eps), _p(scale), _p(shift), n, _stream())
    return scale, shift


# ------------------------------------------------------------------------------------------------
# attention reductions
# ------------------------------------------------------------------------------------------------
def rowdot(x, w, bias, rows, dim, ld=0):
    _chk(x, "x")
    out = torch.empty((rows,), dtype=torch.float32, device=x.device)
    lib().call("dana_rowdot", _p(x), _p(_chk(w.detach().contiguous(), "w")),
               _p(_chk(bias.detach().contiguous(), "bias")) if bias is not None else None, _p(out), rows, dim, ld,
               _stream())
    return out


def softmax_rows_(x, groups, length, ld=0):
    _chk(x, "x")
    lib().call("dana_softmax_rows", _p(x), groups, length, ld, _stream())
    return x


def softmax_rows_to(x, out, groups, length, ld_in=0, ld_out=0):
    """out[g][:length] = softmax(x[g][:length]), x untouched"""
    lib().call("dana_softmax_rows_to", _p(_chk(x, "x")), _p(_chk(out, "out")), groups, length, ld_in, ld_out, _stream())
    return out


def labels_posneg(labels_f):
    """float labels [n] -> int64 [2n]: the labels, then n zeros (rois_label of dana.py:191-194)"""
    n = labels_f.numel()
    out = torch.empty(2 * n, dtype=torch.int64, device=labels_f.device)
    lib().call("dana_labels_posneg_i64", _p(_chk(labels_f, "labels")), _p(out), n, _stream())
    return out


def ba_apply_(s, w, groups, length, dim, ld=0, gamma=0.1, slope=0.01):
    _chk(s, "s")
    _chk(w, "w")
    lib().call("dana_ba_apply", _p(s), _p(w), groups, length, dim, ld, float(gamma), float(slope), _stream())
    return s


def attn_softmax_unary_(scores, unary, rows, rows_per_batch, nseg, length, ld, kpad, unary_gamma, out_scale,
                        unary_batch_stride=0):
    _chk(scores, "scores")
    _chk(unary, "unary")
    lib().call("dana_attn_softmax_unary", _p(scores), _p(unary), rows, rows_per_batch, unary_batch_stride, nseg, length,
               ld, kpad,
               float(unary_gamma), float(out_scale), _stream())
    return scores


# ------------------------------------------------------------------------------------------------
# backward building blocks (groundwork for the training step)
# ------------------------------------------------------------------------------------------------
def conv2d_wgrad(grad_out, x, batch, in_h, in_w, cin, cout, kh, kw, stride, pad, in_stride=0, grad_stride=0,
                 out=None, row_scale=None):
    """dW in the packed layout [cout][kh*kw*cin] = row_scale * sum over output pixels of grad_out^T . im2col(x);
    accumulated into `out` when that is given."""
    _chk(grad_out, "grad_out")
    _chk(x, "x")
    accumulate = out is not None
    if out is None:
        out = torch.empty((cout, kh * kw * cin), dtype=torch.float32, device=x.device)
    ws = _ws(lib().query("dana_conv2d_wgrad_workspace_bytes", batch, in_h, in_w, cin, cout, kh, kw, stride, pad),
             x.device)
    e0 = _prof_begin()
    lib().call("dana_conv2d_wgrad_nhwc", _p(grad_out), _p(x), _p(out), batch, in_h, in_w, cin, cout, kh, kw, stride,
               pad, in_stride, grad_stride, _p(row_scale), int(accumulate), _p(ws), ws.numel(), _stream())
    m = batch * ((in_h + 2 * pad - kh) // stride + 1) * ((in_w + 2 * pad - kw) // stride + 1)
    _prof_end(e0, ("wgrad%dx%d M=%d N=%d K=%d s%d", (kh, kw, m, cout, kh * kw * cin, stride)),
              2.0 * m * cout * kh * kw * cin)
    return out


def gemm_tn_batched(y, x, planes, m, n, k, out, ldy=0, ldx=0, batch_y=0, batch_x=0, batch_out=0, n_valid=None, accumulate=True):
    """out[z][r][k] (+)= sum_m y[z][m][r] * x[z][m][k] for r < n_valid, all planes z in one launch (the adjoints of torch.bmm
    w.r.t. its right operand: d value / d key of the attention, dana.py:140-150)"""
    _chk(y, "y")
    _chk(x, "x")
    _chk(out, "out")
    n_valid = n if n_valid is None else n_valid
    ws = _ws(lib().query("dana_gemm_tn_batched_workspace_bytes", planes, m, n, k), x.device)
    e0 = _prof_begin()
    lib().call("dana_gemm_tn_batched", _p(y), _p(x), _p(out), planes, m, n, k, ldy, ldx, batch_y or m * (ldy or n),
               batch_x or m * (ldx or k), batch_out or n_valid * k, n_valid, int(accumulate), _p(ws), ws.numel(), _stream())
    _prof_end(e0, ("wgrad1x1 M=%d N=%d K=%d b%d", (m, n_valid, k, planes)), 2.0 * planes * m * n_valid * k)
    return out


def downsample_gather(x, batch, ih, iw, channels, stride, in_stride=0):
    """[batch*oh*ow][channels] rows of the pixels a strided 1x1 conv reads"""
    _chk(x, "x")
    oh, ow = (ih - 1) // stride + 1, (iw - 1) // stride + 1
    out = torch.empty((batch * oh * ow, channels), dtype=torch.float32, device=x.device)
    lib().call("dana_downsample_gather_nhwc", _p(x), _p(out), batch, ih, iw, channels, stride, in_stride, _stream())
    return out, oh, ow


def conv3x3_wgrad_winograd(grad_out, x, batch, h, w, cin, cout, in_stride=0, grad_stride=0, out=None, row_scale=None, v=None):
    """conv2d_wgrad of a stride-1 pad-1 3x3 conv through the F(4x4,3x3) domain (4x fewer multiplies). v: the forward
    launch's kept workspace (conv3x3_winograd(keep_v=...)): its V planes replace the input transform of x."""
    _chk(grad_out, "grad_out")
    _chk(x, "x")
    accumulate = out is not None
    if out is None:
        out = torch.empty((cout, 9 * cin), dtype=torch.float32, device=x.device)
    ws = _ws(lib().query("dana_conv3x3_wgrad_winograd4_workspace_bytes", batch, h, w, cin, cout), x.device)
    e0 = _prof_begin()
    if v is not None:
        lib().call("dana_conv3x3_wgrad_winograd4_v", _p(grad_out), _p(v), _p(out), batch, h, w, cin, cout, grad_stride,
                   _p(row_scale), int(accumulate), _p(ws), ws.numel(), _stream())
    else:
        lib().call("dana_conv3x3_wgrad_winograd4", _p(grad_out), _p(x), _p(out), batch, h, w, cin, cout, in_stride,
                   grad_stride, _p(row_scale), int(accumulate), _p(ws), ws.numel(), _stream())
    _prof_end(e0, ("wgradwino M=%d N=%d K=%d", (batch * h * w, cout, 9 * cin)), 2.0 * batch * h * w * cout * 9 * cin)
    return out


def conv2d_dgrad_weight(w_packed, cout, cin, kh, kw, scale=None):
    """flipped / transposed (and frozen-BN scaled) weights [cin][kh*kw*cout]: the data gradient is a forward conv on them"""
    wd = torch.empty((cin, kh * kw * cout), dtype=torch.float32, device=w_packed.device)
    lib().call("dana_conv2d_dgrad_weight", _p(_chk(w_packed, "w_packed")), _p(scale), _p(wd), cout, cin, kh, kw,
               _stream())
    return wd


def conv2d_dgrad(grad_out, w_packed, batch, in_h, in_w, cin, cout, kh, kw, stride, pad, scale=None, wd=None, ud=None,
                 residual=None, mask=None, mask_stride=0, compact_out=False, out=None):
    """grad w.r.t. the NHWC conv input [batch*in_h*in_w][cin]; grad_out [batch*oh*ow][cout].
    wd / ud: cached conv2d_dgrad_weight() / its Winograd transform. residual: added to the result; mask: activation
    whose ReLU adjoint is applied last (zero where mask <= 0). For a strided 1x1 conv residual must be COMPACT
    ([batch*oh*ow][cin], e.g. the downsample branch's compact gradient) and compact_out=True returns the compact
    result without scattering."""
    global PROF_ROLE
    if PROFILE is not None and PROF_ROLE is None:
        PROF_ROLE = "dgrad"
        try:
            return conv2d_dgrad(grad_out, w_packed, batch, in_h, in_w, cin, cout, kh, kw, stride, pad, scale, wd, ud,
                                residual, mask, mask_stride, compact_out, out)
        finally:
            PROF_ROLE = None
    _chk(grad_out, "grad_out")
    if wd is None:
        wd = conv2d_dgrad_weight(w_packed, cout, cin, kh, kw, scale)
    oh = (in_h + 2 * pad - kh) // stride + 1
    ow = (in_w + 2 * pad - kw) // stride + 1
    if stride == 1:
        if ud is not None and residual is None:
            gx, _, _ = conv3x3_winograd(grad_out, batch, oh, ow, cout, ud, cin, mask=mask, mask_stride=mask_stride, out=out,
                                        out_stride=cin if out is not None else 0)
            return gx
        gx = out if out is not None else torch.empty((batch * in_h * in_w, cin), dtype=torch.float32, device=grad_out.device)
        e0 = _prof_begin()
        lib().call("dana_conv2d_nhwc_masked", _p(grad_out), _p(wd), _p(gx), None, None, _p(residual), _p(mask), batch, oh,
                   ow, cout, cin, kh, kw, 1, kh - 1 - pad, 0, cin, 0, mask_stride, 0, _stream())
        _prof_end(e0, ("dgrad%dx%d M=%d N=%d K=%d", (kh, kw, batch * in_h * in_w, cin, kh * kw * cout)),
                  2.0 * batch * in_h * in_w * cin * kh * kw * cout)
        return gx
    if kh != 1 or kw != 1 or pad != 0:
        raise NotImplementedError("strided data gradient only for the 1x1 convs of the Caffe bottleneck")
    compact, _, _ = conv2d_nhwc(grad_out, batch, oh, ow, cout, wd, cin, 1, 1, 1, 0, residual=residual)
    if compact_out:
        return compact
    gx = torch.empty((batch * in_h * in_w, cin), dtype=torch.float32, device=grad_out.device)
    lib().call("dana_upsample_scatter_nhwc", _p(compact), _p(gx), _p(mask), batch, oh, ow, in_h, in_w, cin, stride,
               _stream())
    return gx


def relu_mask_(grad, act, rows, channels, ld_grad=0, ld_act=0):
    lib().call("dana_relu_mask", _p(_chk(grad, "grad")), _p(_chk(act, "act")), rows, channels, ld_grad, ld_act, _stream())
    return grad


def axpy_rows_(y, x, rows, channels, ld_y=0, ld_x=0, alpha=1.0, accumulate=True):
    lib().call("dana_axpy_rows", _p(_chk(y, "y")), _p(_chk(x, "x")), rows, channels, ld_y, ld_x, float(alpha),
               int(bool(accumulate)), _stream())
    return y


def mul_rows_(y, x, rows, channels, ld_y=0, ld_x=0):
    """y *= x over strided rows (attention_type 'product')"""
    lib().call("dana_mul_rows", _p(_chk(y, "y")), _p(_chk(x, "x")), rows, channels, ld_y, ld_x, _stream())
    return y


def rowscale_(dw, scale, rows, cols):
    lib().call("dana_rowscale", _p(_chk(dw, "dw")), _p(_chk(scale, "scale")), rows, cols, _stream())
    return dw


def unpack_conv_weight_grad(packed, grad_oihw, cout, cin, kh, kw, accumulate):
    lib().call("dana_unpack_conv_weight_grad", _p(_chk(packed, "packed")), _p(_chk(grad_oihw, "grad")), cout, cin, kh, kw,
               int(bool(accumulate)), _stream())
    return grad_oihw


def colsum(x, rows, channels, ld=0, out=None, alpha=1.0):
    _chk(x, "x")
    accumulate = out is not None
    if out is None:
        out = torch.empty((channels,), dtype=torch.float32, device=x.device)
    ws = _ws(lib().query("dana_colsum_workspace_bytes", rows, channels), x.device)
    lib().call("dana_colsum", _p(x), _p(out), rows, channels, ld, float(alpha), int(accumulate), _p(ws), ws.numel(),
               _stream())
    return out


def colsum_batched(x, batch, rows, channels, out, ld=0, x_batch=0, out_batch=0, alpha=1.0, accumulate=True):
    """out[z*out_batch + c] (+)= alpha * sum_r x[z*x_batch + r*ld + c], every z in one launch pair"""
    _chk(x, "x")
    ws = _ws(batch * lib().query("dana_colsum_workspace_bytes", rows, channels), x.device)
    lib().call("dana_colsum_batched", _p(x), _p(out), batch, rows, channels, ld, x_batch or rows * (ld or channels),
               out_batch or channels, float(alpha), int(accumulate), _p(ws), ws.numel(), _stream())
    return out


def avgpool_backward(grad_out, B, H, W, C, k, stride):
    _chk(grad_out, "grad_out")
    gin = torch.empty((B, H * W, C), dtype=torch.float32, device=grad_out.device)
    lib().call("dana_avgpool_backward_nhwc", _p(grad_out), _p(gin), B, H, W, C, k, stride, _stream())
    return gin


def softmax_rows_backward_(grad, prob, rows, length, ld_grad=0, ld_prob=0):
    lib().call("dana_softmax_rows_backward", _p(_chk(grad, "grad")), _p(_chk(prob, "prob")), rows, length, ld_grad,
               ld_prob, _stream())
    return grad


def gemm_small(a, a_strides, b, b_strides, m, n, k, out=None, c_strides=None, alpha=1.0):
    """c[m][n] (+)= alpha * sum_k a(m,k) b(k,n) with element strides (a: (m,k), b: (k,n), c: (m,n)); tiny problems only"""
    _chk(a, "a")
    _chk(b, "b")
    accumulate = out is not None
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        c_strides = (n, 1)
    lib().call("dana_gemm_small", _p(a), a_strides[0], a_strides[1], _p(b), b_strides[0], b_strides[1], _p(out),
               c_strides[0], c_strides[1], m, n, k, float(alpha), int(accumulate), _stream())
    return out


def rowdot_backward(x, grad_out, w, rows, dim, grad_x=None, grad_w=None, ld_x=0, ld_grad_x=0):
    """adjoint of rowdot(): returns grad_w [dim] (accumulated into grad_w when given); grad_x += grad_out (x) w"""
    _chk(x, "x")
    _chk(grad_out, "grad_out")
    accumulate = grad_w is not None
    if grad_w is None:
        grad_w = torch.empty((dim,), dtype=torch.float32, device=x.device)
    ws = _ws(lib().query("dana_colsum_workspace_bytes", rows, dim), x.device)
    lib().call("dana_rowdot_backward", _p(x), _p(grad_out), _p(_chk(w.detach().contiguous(), "w")), _p(grad_x),
               _p(grad_w), rows, dim, ld_x, ld_grad_x, int(accumulate), _p(ws), ws.numel(), _stream())
    return grad_w


def broadcast_rows(x, groups, positions, channels, alpha=1.0, out=None):
    _chk(x, "x")
    accumulate = out is not None
    if out is None:
        out = torch.empty((groups * positions, channels), dtype=torch.float32, device=x.device)
    lib().call("dana_broadcast_rows", _p(x), _p(out), groups, positions, channels, float(alpha), int(accumulate),
               _stream())
    return out


def ba_backward_prep(s, weights, grad_s, groups, length, dim):
    """-> (gvec [groups][dim] = weights^T s per group, gsum [groups][dim] = column sums of grad_s per group): one launch"""
    gvec = torch.empty((groups, dim), dtype=torch.float32, device=s.device)
    gsum = torch.empty((groups, dim), dtype=torch.float32, device=s.device)
    lib().call("dana_ba_backward_prep", _p(_chk(s, "s")), _p(_chk(weights, "weights")), _p(_chk(grad_s, "grad_s")), _p(gvec),
               _p(gsum), groups, length, dim, _stream())
    return gvec, gsum


def ba_backward_(grad_s, s, weights, gvec, gsum, groups, length, dim, gamma=0.1, slope=0.01):
    grad_w = torch.empty((groups * length,), dtype=torch.float32, device=s.device)
    lib().call("dana_ba_backward", _p(_chk(grad_s, "grad_s")), _p(_chk(s, "s")), _p(_chk(weights, "weights")),
               _p(_chk(gvec, "gvec")), _p(_chk(gsum, "gsum")), _p(grad_w), groups, length, dim, float(gamma),
               float(slope), _stream())
    return grad_w


def attn_softmax_unary_backward_(grad_a, a, unary, rows, rows_per_batch, nseg, length, ld, kpad, unary_gamma, out_scale,
                                 alpha, unary_batch_stride=0):
    lib().call("dana_attn_softmax_unary_backward", _p(_chk(grad_a, "grad_a")), _p(_chk(a, "a")), _p(_chk(unary, "unary")),
               rows, rows_per_batch, unary_batch_stride, nseg, length, ld, kpad, float(unary_gamma), float(out_scale),
               float(alpha), _stream())
    return grad_a


def scale_by_device_scalar_(x, scalar_dev):
    lib().call("dana_scale_by_device_scalar", _p(_chk(x, "x")), x.numel(), _p(_chk(scalar_dev, "scalar")), _stream())
    return x


def rpn_loss_backward(heads, head_row_stride, h, losses3, grad_cls=1.0, grad_box=1.0, sigma=3.0, inside_weight=1.0,
                      grad_dev=None):
    """d(grad_cls * rpn_loss_cls + grad_box * rpn_loss_bbox) / d heads, same [B*H*W][row stride] layout;
    grad_dev: the two upstream scalars as a device tensor instead"""
    _chk(heads, "heads")
    g = torch.empty_like(heads)
    lib().call("dana_rpn_loss_backward", _p(heads), head_row_stride, _p(h["labels"]), _p(h["argmax"]), _p(h["gt_boxes"]),
               _p(h["base_anchors"]), h["B"], h["A"], h["H"], h["W"], h["stride"], h["n_gt"], float(sigma),
               float(inside_weight), 1.0 / h["num_examples"], _p(h.get("inv_ne_dev")), _p(_chk(losses3, "losses3")),
               float(grad_cls),
               float(grad_box), _p(grad_dev), _p(g), _stream())
    return g


def linear_wgrad(g, x, m, n, k, ldx=0, ldg=0):
    """(dW [n][k], db [n]) of y = x . w^T + b from g = dL/dy: the two pieces of a Linear's backward that nothing on the
    data-gradient chain waits for (backward.py issues them on the weight-gradient stream)"""
    return conv2d_wgrad(g, x, 1, 1, m, k, n, 1, 1, 1, 0, in_stride=ldx, grad_stride=ldg), colsum(g, m, n, ld=ldg)


def linear_backward(g, x, w, m, n, k, ldx=0, ldg=0, ldw=0, need_dx=True, dx_out=None, dx_ld=0, need_dw=True):
    """y[m][n] = x[m][:k] . w[n][:k]^T + b: returns (dW [n][k] dense, db [n], dx [m][k] or None).
    n % 4 == 0 and k % 64 == 0 (the split-M TN MFMA kernel); dx goes through the forward GEMM on w^T and is
    ACCUMULATED into dx_out (row stride dx_ld) when that is given."""
    ldw = ldw or k
    dw = db = None
    if need_dw:
        dw, db = linear_wgrad(g, x, m, n, k, ldx=ldx, ldg=ldg)
    dx = None
    if need_dx:
        wt = torch.empty((k, n), dtype=torch.float32, device=g.device)  # [k][n] = w^T
        lib().call("dana_transpose_batched", _p(_chk(w, "w")), _p(wt), 1, n, k, ldw, n, n * ldw, k * n, _stream())
        if dx_out is None:
            dx = gemm_nt(g, wt, m, k, n, lda=ldg)
        else:
            dx = gemm_nt(g, wt, m, k, n, lda=ldg, out=dx_out, ldc=dx_ld or k, residual=dx_out, ldr=dx_ld or k)
    return dw, db, dx


def sgd_momentum_(params, grads, buf, lr, momentum, weight_decay, grad_scale=1.0, first_step=False):
    """fused torch.optim.SGD(momentum) update of one flat fp32 segment (numel % 4 == 0), in place"""
    n = params.numel()
    lib().call("dana_sgd_momentum", _p(_chk(params, "params")), _p(_chk(grads, "grads")), _p(_chk(buf, "buf")), n,
               float(lr), float(momentum), float(weight_decay), float(grad_scale), int(bool(first_step)), _stream())
    return params


def adam_(params, grads, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """fused torch.optim.Adam update of one flat fp32 segment (numel % 4 == 0), in place; step counts from 1"""
    lib().call("dana_adam", _p(_chk(params, "params")), _p(_chk(grads, "grads")), _p(_chk(exp_avg, "exp_avg")),
               _p(_chk(exp_avg_sq, "exp_avg_sq")), params.numel(), float(lr), float(beta1), float(beta2), float(eps),
               float(weight_decay), float(grad_scale), int(step), _stream())
    return params
