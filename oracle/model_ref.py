"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU fp32 restatement, in plain torch ops over a reference-compatible ``state_dict``, of the DAnA
forward path: ``_DAnARCNN.forward`` (lib/model/framework/dana.py:87-220) and everything it calls
(resnet.py trunk, rpn.py head, proposal_layer.py, anchor/proposal target layers, losses).
It is functional (no nn.Module state) so that every stage can be fed reference intermediates.

Pinned: tests/golden/make_golden.py runs this next to the imported reference on the same seeded
inputs/weights (bit-identical outputs required there) and stores golden vectors; the GPU parity
tests compare the HIP path with this file and with those vectors.
Also used as bench.py's ``cpu_baseline`` (kind "port").
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import native

# ------------------------------------------------------------------------------------------------
# configuration: lib/model/utils/config.py:19-303 merged with cfgs/res50.yml and utils.py:70-71
# ------------------------------------------------------------------------------------------------
CFG = {
    "ANCHOR_SCALES": [4, 8, 16, 32],
    "ANCHOR_RATIOS": [0.5, 1, 2],
    "FEAT_STRIDE": 16,
    "POOLING_SIZE": 7,
    "MAX_NUM_GT_BOXES": 50,
    "TRAIN": dict(RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_NMS_THRESH=0.7, BATCH_SIZE=128,
                  FG_FRACTION=0.25, FG_THRESH=0.5, BG_THRESH_HI=0.5, BG_THRESH_LO=0.0,
                  RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3, RPN_FG_FRACTION=0.5, RPN_BATCHSIZE=256,
                  BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0), BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2)),
    "TEST": dict(RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, RPN_NMS_THRESH=0.7),
}


# ------------------------------------------------------------------------------------------------
# anchors: lib/model/rpn/generate_anchors.py:45-105 (np.round = banker's rounding, :91-92)
# ------------------------------------------------------------------------------------------------
def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(4, 8, 16, 32)):
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    w = h = float(base_size)
    cx = cy = 0.5 * (base_size - 1)
    size = w * h
    ws = np.round(np.sqrt(size / ratios))
    hs = np.round(ws * ratios)
    out = []
    for rw, rh in zip(ws, hs):  # ratio-major, scale-minor
        for s in scales:
            sw, sh = rw * s, rh * s
            out.append([cx - 0.5 * (sw - 1), cy - 0.5 * (sh - 1), cx + 0.5 * (sw - 1), cy + 0.5 * (sh - 1)])
    return np.asarray(out, dtype=np.float64)


def anchor_grid(base_anchors, H, W, stride):
    """proposal_layer.py:80-93: [K*A, 4] float32, k = h*W + w major, anchor minor."""
    sx = torch.arange(W, dtype=torch.float32) * stride
    sy = torch.arange(H, dtype=torch.float32) * stride
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack([xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)], 1)
    a = torch.from_numpy(np.asarray(base_anchors)).float()
    return (a.view(1, -1, 4) + shifts.view(-1, 1, 4)).reshape(-1, 4)


# ------------------------------------------------------------------------------------------------
# box arithmetic: lib/model/rpn/bbox_transform.py
# ------------------------------------------------------------------------------------------------
def bbox_transform_inv(boxes, deltas):
    """:77-103, boxes/deltas [B, N, 4]"""
    widths = boxes[..., 2] - boxes[..., 0] + 1.0
    heights = boxes[..., 3] - boxes[..., 1] + 1.0
    ctr_x = boxes[..., 0] + 0.5 * widths
    ctr_y = boxes[..., 1] + 0.5 * heights
    pcx = deltas[..., 0] * widths + ctr_x
    pcy = deltas[..., 1] * heights + ctr_y
    pw = torch.exp(deltas[..., 2]) * widths
    ph = torch.exp(deltas[..., 3]) * heights
    return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], -1)


def clip_boxes(boxes, im_info):
    """:125-133"""
    out = boxes.clone()
    for i in range(boxes.shape[0]):
        out[i, :, 0::2] = out[i, :, 0::2].clamp(0, float(im_info[i, 1]) - 1)
        out[i, :, 1::2] = out[i, :, 1::2].clamp(0, float(im_info[i, 0]) - 1)
    return out


def bbox_transform_batch(ex, gt):
    """:36-75 (3-D branch and the 2-D-anchors branch give the same formula under broadcasting)"""
    ew = ex[..., 2] - ex[..., 0] + 1.0
    eh = ex[..., 3] - ex[..., 1] + 1.0
    ecx = ex[..., 0] + 0.5 * ew
    ecy = ex[..., 1] + 0.5 * eh
    gw = gt[..., 2] - gt[..., 0] + 1.0
    gh = gt[..., 3] - gt[..., 1] + 1.0
    gcx = gt[..., 0] + 0.5 * gw
    gcy = gt[..., 1] + 0.5 * gh
    return torch.stack([(gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)], -1)


def bbox_overlaps_batch(anchors, gt_boxes):
    """:168-257. anchors [N,4] or [B,N,4|5]; gt_boxes [B,K,5] -> [B,N,K]"""
    B = gt_boxes.shape[0]
    if anchors.dim() == 2:
        anchors = anchors.view(1, -1, 4).expand(B, -1, 4)
    elif anchors.shape[2] == 5:
        anchors = anchors[:, :, 1:5]
    gt = gt_boxes[:, :, :4]
    gx = gt[:, :, 2] - gt[:, :, 0] + 1
    gy = gt[:, :, 3] - gt[:, :, 1] + 1
    ax = anchors[:, :, 2] - anchors[:, :, 0] + 1
    ay = anchors[:, :, 3] - anchors[:, :, 1] + 1
    g_area = (gx * gy).unsqueeze(1)
    a_area = (ax * ay).unsqueeze(2)
    iw = torch.min(anchors[:, :, None, 2], gt[:, None, :, 2]) - torch.max(anchors[:, :, None, 0], gt[:, None, :, 0]) + 1
    iw = iw.clamp(min=0)
    ih = torch.min(anchors[:, :, None, 3], gt[:, None, :, 3]) - torch.max(anchors[:, :, None, 1], gt[:, None, :, 1]) + 1
    ih = ih.clamp(min=0)
    ua = a_area + g_area - iw * ih
    ov = iw * ih / ua
    ov = ov.masked_fill(((gx == 1) & (gy == 1)).unsqueeze(1), 0)
    ov = ov.masked_fill(((ax == 1) & (ay == 1)).unsqueeze(2), -1)
    return ov


def smooth_l1(pred, tgt, w_in, w_out, sigma=1.0, dims=(1,)):
    """lib/model/utils/net_utils.py:71-85"""
    s2 = sigma ** 2
    d = w_in * (pred - tgt)
    ad = d.abs()
    sign = (ad < 1.0 / s2).detach().float()
    loss = w_out * (d.pow(2) * (s2 / 2.0) * sign + (ad - 0.5 / s2) * (1.0 - sign))
    for i in sorted(dims, reverse=True):
        loss = loss.sum(i)
    return loss.mean()


# ------------------------------------------------------------------------------------------------
# trunk: lib/model/framework/resnet.py:66-146 (Caffe style: stride on the first 1x1, ceil maxpool)
# ------------------------------------------------------------------------------------------------
def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def bottleneck(x, sd, p, stride):
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=stride), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], padding=1), sd, p + ".bn2"))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if (p + ".downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
    return F.relu(out + x)


def layer(x, sd, p, blocks, stride):
    for i in range(blocks):
        x = bottleneck(x, sd, "%s.%d" % (p, i), stride if i == 0 else 1)
    return x


def rcnn_base(x, sd):
    """dana.py:344-345: conv1, bn1, relu, maxpool, layer1..3 (keys RCNN_base.0/1/4/5/6)"""
    x = F.relu(_bn(F.conv2d(x, sd["RCNN_base.0.weight"], stride=2, padding=3), sd, "RCNN_base.1"))
    x = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
    # blocks per stage from the state dict itself: [3, 4, 6] for resnet50() (what dana.py:337 always builds), [3, 4, 23]
    # for the resnet101 trunk of resnet.py:199 (the build's opt-in for BASELINE configs[3])
    nb = [1 + max(int(k.split(".")[2]) for k in sd if k.startswith("RCNN_base.%d." % s)) for s in (4, 5, 6)]
    x = layer(x, sd, "RCNN_base.4", nb[0], 1)
    x = layer(x, sd, "RCNN_base.5", nb[1], 2)
    return layer(x, sd, "RCNN_base.6", nb[2], 2)


def rcnn_top(x, sd):
    """dana.py:346,387-389"""
    return layer(x, sd, "RCNN_top.0", 3, 2).mean(3).mean(2)


def positional_encoding(max_len, d_model=1024):
    """dana.py:309-320"""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0., max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0., d_model, 2) * -(math.log(10000.0) / float(d_model)))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


# ------------------------------------------------------------------------------------------------
# attention, RPN level: dana.py:118-154
# ------------------------------------------------------------------------------------------------
def rpn_attention(base_feat, pos_support_feat, sd, n_shot, use_ba, inter=None):
    B = pos_support_feat.shape[0]
    h, w = base_feat.shape[2:]
    support_mat = pos_support_feat.transpose(0, 1).reshape(n_shot, B, 1024, -1).transpose(2, 3)
    query_mat = base_feat.reshape(B, 1024, -1).transpose(1, 2)
    q = _lin(query_mat, sd, "rpn_adapt_q_layer")
    q = q - q.mean(1, keepdim=True)
    pe = positional_encoding(400)
    feats = []
    for i in range(n_shot):
        s = support_mat[i] + pe
        if use_ba:  # BA block :133-137
            wgt = F.softmax(_lin(s, sd, "rpn_channel_k_layer"), 1)
            g = torch.bmm(wgt.transpose(1, 2), s)
            s = s + 0.1 * F.leaky_relu(g)
        k = _lin(s, sd, "rpn_adapt_k_layer")
        k = k - k.mean(1, keepdim=True)
        a = F.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(256), dim=2)
        u = F.softmax(_lin(s, sd, "rpn_unary_layer"), dim=1)
        a = a + 0.1 * u.transpose(1, 2)
        feats.append(torch.bmm(a, s))
    dense = torch.stack(feats, 0).mean(0).transpose(1, 2).reshape(B, 1024, h, w)
    if inter is not None:
        inter["rpn_q"] = q
        inter["dense_support_feature"] = dense
    if sd["RCNN_rpn.RPN_Conv.weight"].shape[1] == 1024:  # attention_type 'product' (dana.py:155-156): _RPN(1024)
        return base_feat * dense
    return torch.cat([base_feat, dense], 1)


# ------------------------------------------------------------------------------------------------
# RPN head + proposal layer: rpn.py:58-78, proposal_layer.py:49-190
# ------------------------------------------------------------------------------------------------
def rpn_head(corr, sd):
    x = F.relu(F.conv2d(corr, sd["RCNN_rpn.RPN_Conv.weight"], sd["RCNN_rpn.RPN_Conv.bias"], padding=1))
    cls = F.conv2d(x, sd["RCNN_rpn.RPN_cls_score.weight"], sd["RCNN_rpn.RPN_cls_score.bias"])
    B, C2, H, W = cls.shape
    prob = F.softmax(cls.view(B, 2, C2 // 2 * H, W), 1).view(B, C2, H, W)
    bbox = F.conv2d(x, sd["RCNN_rpn.RPN_bbox_pred.weight"], sd["RCNN_rpn.RPN_bbox_pred.bias"])
    return cls, prob, bbox


def proposal_layer(prob, bbox, im_info, key, nms_inclusive=True, inter=None):
    c = CFG[key]
    base = generate_anchors(scales=CFG["ANCHOR_SCALES"], ratios=CFG["ANCHOR_RATIOS"])
    A = base.shape[0]
    B, _, H, W = bbox.shape
    anchors = anchor_grid(base, H, W, CFG["FEAT_STRIDE"]).unsqueeze(0).expand(B, -1, 4)
    scores = prob[:, A:].permute(0, 2, 3, 1).reshape(B, -1)
    deltas = bbox.permute(0, 2, 3, 1).reshape(B, -1, 4)
    proposals = clip_boxes(bbox_transform_inv(anchors, deltas), im_info)
    _, order = torch.sort(scores, 1, True)
    out = torch.zeros(B, c["RPN_POST_NMS_TOP_N"], 5)
    for i in range(B):
        o = order[i]
        if 0 < c["RPN_PRE_NMS_TOP_N"] < scores.numel():  # :148 compares with the whole batch's count
            o = o[:c["RPN_PRE_NMS_TOP_N"]]
        p, s = proposals[i][o], scores[i][o]
        keep = torch.from_numpy(native.nms(p.numpy(), s.numpy(), c["RPN_NMS_THRESH"], nms_inclusive))
        keep = keep[:c["RPN_POST_NMS_TOP_N"]]
        out[i, :, 0] = i
        out[i, :keep.numel(), 1:] = p[keep]
    if inter is not None:
        inter["rpn_scores"] = scores
        inter["rpn_proposals"] = proposals
    return out


# ------------------------------------------------------------------------------------------------
# training-only target layers (host RNG = np.random, as the reference)
# ------------------------------------------------------------------------------------------------
def anchor_target_layer(cls_shape, gt_boxes, im_info):
    """lib/model/rpn/anchor_target_layer.py:48-193"""
    t = CFG["TRAIN"]
    B = gt_boxes.shape[0]
    H, W = cls_shape
    base = generate_anchors(scales=CFG["ANCHOR_SCALES"], ratios=CFG["ANCHOR_RATIOS"])
    A = base.shape[0]
    all_anchors = anchor_grid(base, H, W, CFG["FEAT_STRIDE"])
    total = all_anchors.shape[0]
    keep = ((all_anchors[:, 0] >= 0) & (all_anchors[:, 1] >= 0) & (all_anchors[:, 2] < int(im_info[0][1])) &
            (all_anchors[:, 3] < int(im_info[0][0])))
    inds = torch.nonzero(keep).view(-1)
    anchors = all_anchors[inds]
    n = inds.numel()
    labels = torch.full((B, n), -1.0)
    ov = bbox_overlaps_batch(anchors, gt_boxes)
    max_ov, argmax_ov = ov.max(2)
    gt_max, _ = ov.max(1)
    labels[max_ov < t["RPN_NEGATIVE_OVERLAP"]] = 0
    gt_max[gt_max == 0] = 1e-5
    k = ov.eq(gt_max.view(B, 1, -1).expand_as(ov)).sum(2)
    if k.sum() > 0:
        labels[k > 0] = 1
    labels[max_ov >= t["RPN_POSITIVE_OVERLAP"]] = 1
    num_fg = int(t["RPN_FG_FRACTION"] * t["RPN_BATCHSIZE"])
    sum_fg = (labels == 1).int().sum(1)
    sum_bg = (labels == 0).int().sum(1)
    for i in range(B):
        if sum_fg[i] > num_fg:
            fg = torch.nonzero(labels[i] == 1).view(-1)
            rnd = torch.from_numpy(np.random.permutation(fg.numel())).long()
            labels[i][fg[rnd[:fg.numel() - num_fg]]] = -1
        num_bg = t["RPN_BATCHSIZE"] - (labels == 1).int().sum(1)[i]
        if sum_bg[i] > num_bg:
            bg = torch.nonzero(labels[i] == 0).view(-1)
            rnd = torch.from_numpy(np.random.permutation(bg.numel())).long()
            labels[i][bg[rnd[:bg.numel() - num_bg]]] = -1
    offset = torch.arange(B) * gt_boxes.shape[1]
    am = argmax_ov + offset.view(B, 1)
    targets = bbox_transform_batch(anchors, gt_boxes.view(-1, 5)[am.view(-1)].view(B, -1, 5)[:, :, :4])
    w_in = torch.zeros(B, n)
    w_in[labels == 1] = 1.0
    num_examples = (labels[B - 1] >= 0).sum().item()  # :156 uses the LAST image of the batch
    w_out = torch.zeros(B, n)
    w_out[labels == 1] = 1.0 / num_examples
    w_out[labels == 0] = 1.0 / num_examples

    def unmap(d, fill):
        if d.dim() == 2:
            r = torch.full((B, total), float(fill), dtype=d.dtype)
            r[:, inds] = d
        else:
            r = torch.full((B, total, d.shape[2]), float(fill), dtype=d.dtype)
            r[:, inds, :] = d
        return r

    labels = unmap(labels, -1).view(B, H, W, A).permute(0, 3, 1, 2).reshape(B, 1, A * H, W)
    targets = unmap(targets, 0).view(B, H, W, A * 4).permute(0, 3, 1, 2)
    w_in = unmap(w_in, 0).view(B, total, 1).expand(B, total, 4).reshape(B, H, W, 4 * A).permute(0, 3, 1, 2)
    w_out = unmap(w_out, 0).view(B, total, 1).expand(B, total, 4).reshape(B, H, W, 4 * A).permute(0, 3, 1, 2)
    return labels, targets, w_in, w_out


def proposal_target_layer(all_rois, gt_boxes):
    """lib/model/rpn/proposal_target_layer_cascade.py:33-213"""
    t = CFG["TRAIN"]
    B = gt_boxes.shape[0]
    gt_append = torch.zeros_like(gt_boxes)
    gt_append[:, :, 1:5] = gt_boxes[:, :, :4]
    all_rois = torch.cat([all_rois, gt_append], 1)
    R = t["BATCH_SIZE"]
    fg_per = int(np.round(t["FG_FRACTION"] * R)) or 1
    ov = bbox_overlaps_batch(all_rois, gt_boxes)
    max_ov, assign = ov.max(2)
    offset = (torch.arange(B) * gt_boxes.shape[1]).view(-1, 1) + assign
    labels = gt_boxes[:, :, 4].contiguous().view(-1)[offset.view(-1)].view(B, -1)
    labels_b = torch.zeros(B, R)
    rois_b = torch.zeros(B, R, 5)
    gt_b = torch.zeros(B, R, 5)
    for i in range(B):
        fg = torch.nonzero(max_ov[i] >= t["FG_THRESH"]).view(-1)
        bg = torch.nonzero((max_ov[i] < t["BG_THRESH_HI"]) & (max_ov[i] >= t["BG_THRESH_LO"])).view(-1)
        nf, nb = fg.numel(), bg.numel()
        if nf > 0 and nb > 0:
            fg_n = min(fg_per, nf)
            fg = fg[torch.from_numpy(np.random.permutation(nf)).long()[:fg_n]]
            bg_n = R - fg_n
            bg = bg[torch.from_numpy(np.floor(np.random.rand(bg_n) * nb)).long()]
        elif nf > 0:
            fg = fg[torch.from_numpy(np.floor(np.random.rand(R) * nf)).long()]
            fg_n, bg_n = R, 0
            bg = bg[:0]
        elif nb > 0:
            bg = bg[torch.from_numpy(np.floor(np.random.rand(R) * nb)).long()]
            fg_n, bg_n = 0, R
            fg = fg[:0]
        else:
            raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
        keep = torch.cat([fg, bg], 0)
        labels_b[i] = labels[i][keep]
        if fg_n < R:
            labels_b[i][fg_n:] = 0
        rois_b[i] = all_rois[i][keep]
        rois_b[i, :, 0] = i
        gt_b[i] = gt_boxes[i][assign[i][keep]]
    tg = bbox_transform_batch(rois_b[:, :, 1:5], gt_b[:, :, :4])
    tg = (tg - torch.tensor(t["BBOX_NORMALIZE_MEANS"])) / torch.tensor(t["BBOX_NORMALIZE_STDS"])
    fgm = (labels_b > 0).unsqueeze(2).float()
    # :83-91: only fg rows are COPIED into the zero-initialised target tensor; images whose labels sum to 0 keep
    # nothing (same thing). A select, not a product: a bg row whose raw target is inf / nan (degenerate box) stays 0.
    targets = torch.where(fgm > 0, tg, torch.zeros_like(tg))
    w_in = fgm.expand(B, R, 4).clone()
    w_out = (w_in > 0).float()
    return rois_b, labels_b, targets, w_in, w_out


def sampled_targets(rois_b, labels_b, gt_boxes):
    """The deterministic tail of proposal_target_layer_cascade.py:176-213 for an ALREADY sampled batch: given the sampled
    rois [B,R,5] and their labels [B,R] (e.g. the reference's own, from a golden fixture) recompute the gt assignment
    (:113-117) and the normalised regression targets / weights (:61-110,205-211). -> the 5-tuple of proposal_target_layer."""
    t = CFG["TRAIN"]
    B, R = labels_b.shape
    ov = bbox_overlaps_batch(rois_b, gt_boxes)
    _, assign = ov.max(2)
    gt_b = torch.stack([gt_boxes[i][assign[i]] for i in range(B)], 0)
    tg = bbox_transform_batch(rois_b[:, :, 1:5], gt_b[:, :, :4])
    tg = (tg - torch.tensor(t["BBOX_NORMALIZE_MEANS"])) / torch.tensor(t["BBOX_NORMALIZE_STDS"])
    fgm = (labels_b > 0).unsqueeze(2).float()
    w_in = fgm.expand(B, R, 4).clone()
    return rois_b, labels_b, torch.where(fgm > 0, tg, torch.zeros_like(tg)), w_in, (w_in > 0).float()


# ------------------------------------------------------------------------------------------------
# RoI-level head: dana.py:244-292
# ------------------------------------------------------------------------------------------------
def rcnn_head(pooled, support_pooled, sd, n_shot, inter=None):
    fc7 = rcnn_top(pooled, sd)
    bbox_pred = _lin(fc7, sd, "RCNN_bbox_pred")
    n_roi = pooled.shape[0]
    B = support_pooled.shape[0]
    pe = positional_encoding(49)
    smat, qmat = [], []
    for qf, tf in zip(pooled.chunk(B, 0), support_pooled.chunk(B, 0)):
        tf = tf.reshape(1, n_shot, 1024, -1).transpose(2, 3).repeat(qf.shape[0], 1, 1, 1)
        qf = qf.reshape(qf.shape[0], 1024, -1).transpose(1, 2)
        smat.append((tf.reshape(-1, 49, 1024) + pe).view(-1, n_shot, 49, 1024))
        qmat.append(qf + pe)
    smat = torch.cat(smat, 0).transpose(0, 1)
    qmat = torch.cat(qmat, 0)
    q = _lin(qmat, sd, "rcnn_adapt_q_layer")
    q = q - q.mean(1, keepdim=True)
    feats = []
    for i in range(n_shot):
        s = smat[i]
        k = _lin(s, sd, "rcnn_adapt_k_layer")
        k = k - k.mean(1, keepdim=True)
        a = F.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(256), dim=2)
        u = F.softmax(_lin(s, sd, "rcnn_unary_layer"), dim=1)
        a = a + 0.1 * u.transpose(1, 2)
        feats.append(torch.bmm(a, s))
    dense = torch.stack(feats, 0).mean(0)
    if sd["rcnn_transform_layer.weight"].shape[1] == 1024:  # attention_type 'product' (dana.py:285-286)
        corr = _lin(qmat * dense, sd, "rcnn_transform_layer")
    else:
        corr = _lin(torch.cat([qmat, dense], 2), sd, "rcnn_transform_layer")
    x = F.relu(_lin(corr.reshape(n_roi, -1), sd, "output_score_layer.linear1"))
    score = _lin(x, sd, "output_score_layer.linear2")
    if inter is not None:
        inter["fc7"] = fc7
        inter["rcnn_dense"] = dense
    return bbox_pred, F.softmax(score, 1), score


# ------------------------------------------------------------------------------------------------
# the whole forward: dana.py:87-220
# ------------------------------------------------------------------------------------------------
def roi_align_torch(feat, rois, scale, P):
    """Differentiable (w.r.t. feat) restatement of ROIAlign_cpu.cpp:116-245 with sampling_ratio = 0, used only to
    obtain reference GRADIENTS through autograd (tests/test_gpu_backward.py); checked against the C oracle."""
    B, C, H, W = feat.shape
    outs = []

    def prep(v, size):
        valid = (v >= -1.0) & (v <= size)
        v = v.clamp(min=0)
        lo = v.floor().long()
        edge = lo >= size - 1
        hi = torch.where(edge, torch.full_like(lo, size - 1), lo + 1)
        lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
        v = torch.where(edge, lo.to(v.dtype), v)
        frac = v - lo.to(v.dtype)
        return valid, lo, hi, frac, 1.0 - frac

    for r in rois.detach().float():
        b = int(r[0])
        x1, y1, x2, y2 = [r[i] * scale for i in range(1, 5)]
        rw = torch.clamp(x2 - x1, min=1.0)
        rh = torch.clamp(y2 - y1, min=1.0)
        bw, bh = rw / P, rh / P
        gh, gw = int(math.ceil(float(rh) / P)), int(math.ceil(float(rw) / P))
        ip = torch.arange(P, dtype=torch.float32).view(P, 1)
        ys = (y1 + ip * bh + (torch.arange(gh, dtype=torch.float32).view(1, gh) + 0.5) * bh / gh).reshape(-1)
        xs = (x1 + ip * bw + (torch.arange(gw, dtype=torch.float32).view(1, gw) + 0.5) * bw / gw).reshape(-1)
        vy, ylo, yhi, ly, hy = prep(ys, H)
        vx, xlo, xhi, lx, hx = prep(xs, W)
        plane = feat[b]
        ly, hy, lx, hx = [t.to(feat.dtype) for t in (ly, hy, lx, hx)]

        def g(yi, xi):
            return plane[:, yi][:, :, xi]

        val = (hy[:, None] * hx[None, :] * g(ylo, xlo) + hy[:, None] * lx[None, :] * g(ylo, xhi)
               + ly[:, None] * hx[None, :] * g(yhi, xlo) + ly[:, None] * lx[None, :] * g(yhi, xhi))
        val = val * (vy[:, None] & vx[None, :]).to(feat.dtype)
        outs.append(val.view(C, P, gh, P, gw).sum((2, 4)) / (gh * gw))
    return torch.stack(outs, 0)


def forward(sd, im_data, im_info, gt_boxes, num_boxes, support_ims, training, n_way=2, n_shot=3, use_ba=False,
            nms_inclusive=True, inter=None, differentiable=False, sampled=None, pooling="align"):
    """sampled: optional (rois [B,R,5], labels [B,R], targets [B,R,4], w_in, w_out) used INSTEAD of this call's own
    proposal_target_layer draw (everything downstream of the sampling is then a deterministic function of it).
    differentiable=True: the state-dict tensors may require grad (RoIAlign through roi_align_torch; the proposal
    layer sees detached inputs, as rpn.py:73 passes .data)"""
    B = im_data.shape[0]
    base_feat = rcnn_base(im_data, sd)
    sup = rcnn_base(support_ims.reshape(-1, *support_ims.shape[2:]), sd)
    if training:
        sup = sup.view(-1, n_way * n_shot, *sup.shape[1:])
        pos = sup[:, :n_shot].contiguous()
        neg = sup[:, n_shot:n_way * n_shot].contiguous()
        pos_pooled = F.avg_pool2d(pos.view(-1, 1024, 20, 20), 14, 1).view(B, n_shot, 1024, 7, 7)
        neg_pooled = F.avg_pool2d(neg.view(-1, 1024, 20, 20), 14, 1).view(B, n_shot, 1024, 7, 7)
    else:
        sup = sup.view(-1, n_shot, *sup.shape[1:])
        pos = sup[:, :n_shot]
        pos_pooled = F.avg_pool2d(pos.reshape(-1, 1024, 20, 20), 14, 1).view(B, n_shot, 1024, 7, 7)
    if inter is not None:
        inter["base_feat"] = base_feat
        inter["pos_support_feat"] = pos
    corr = rpn_attention(base_feat, pos, sd, n_shot, use_ba, inter)
    cls, prob, bbox = rpn_head(corr, sd)
    if inter is not None:
        inter["rpn_cls_score"] = cls
        inter["rpn_bbox_pred"] = bbox
    rois = proposal_layer(prob.detach(), bbox.detach(), im_info, "TRAIN" if training else "TEST", nms_inclusive, inter)
    if inter is not None:
        inter["rpn_rois"] = rois.clone()
    rpn_loss_cls = rpn_loss_bbox = 0
    rois_label = None
    if training:
        H, W = cls.shape[2:]
        lab, tg, w_in, w_out = anchor_target_layer((H, W), gt_boxes, im_info)
        sc = cls.view(B, 2, -1, W).permute(0, 2, 3, 1).reshape(B, -1, 2)
        lab = lab.view(B, -1)
        keep = lab.view(-1).ne(-1).nonzero().view(-1)
        rpn_loss_cls = F.cross_entropy(sc.reshape(-1, 2)[keep], lab.view(-1)[keep].long())
        rpn_loss_bbox = smooth_l1(bbox, tg, w_in, w_out, sigma=3, dims=(1, 2, 3))
        if sampled is not None:  # stage-wise parity: the caller supplies the sampled batch (see sampled_targets)
            rois, rois_label, rois_target, rw_in, rw_out = sampled
        else:
            rois, rois_label, rois_target, rw_in, rw_out = proposal_target_layer(rois, gt_boxes)
        if inter is not None:
            inter["sampled"] = (rois.clone(), rois_label.clone(), rois_target.clone(), rw_in.clone(), rw_out.clone())
        rois_label = rois_label.view(-1).long()
        rois_target = rois_target.view(-1, 4)
        rw_in = rw_in.view(-1, 4)
        rw_out = rw_out.view(-1, 4)
    if inter is not None:
        inter["rois"] = rois
    if pooling == "pool":  # dana.py:183-184: cfg.POOLING_MODE == 'pool' -> RCNN_roi_pool (ROIPool.h)
        out_np, arg_np = native.roi_pool_forward(base_feat.detach().numpy(), rois.view(-1, 5).numpy(), 1.0 / 16.0, 7, 7)
        pooled = torch.from_numpy(out_np)
        if differentiable:
            # ROIPool_cuda.cu:79-108: the gradient of a bin goes to ITS argmax element (-1: empty bin, no gradient) -- a
            # gather through the forward's argmax, not torch.amax (which splits ties, and post-ReLU maps are full of them)
            r5 = rois.view(-1, 5)
            arg = torch.from_numpy(arg_np).long()
            Bf, Cf, Hf, Wf = base_feat.shape
            src = base_feat.reshape(Bf, Cf, Hf * Wf)[r5[:, 0].long()]
            pooled = (torch.gather(src, 2, arg.clamp(min=0).reshape(arg.size(0), Cf, 49)) * (arg.reshape(arg.size(0), Cf, 49) >= 0)
                      ).reshape(arg.size(0), Cf, 7, 7)
    elif differentiable:
        pooled = roi_align_torch(base_feat, rois.view(-1, 5), 1.0 / 16.0, 7)
    else:
        pooled = torch.from_numpy(native.roi_align_forward(base_feat.detach().numpy(), rois.view(-1, 5).numpy(),
                                                           1.0 / 16.0, 7, 7, 0))
    if inter is not None:
        inter["pooled_feat"] = pooled
    bbox_pred, cls_prob, cls_score = rcnn_head(pooled, pos_pooled, sd, n_shot, inter)
    loss_cls = loss_bbox = 0
    if training:
        _, neg_prob, neg_score = rcnn_head(pooled, neg_pooled, sd, n_shot)
        cls_prob = torch.cat([cls_prob, neg_prob], 0)
        cls_score = torch.cat([cls_score, neg_score], 0)
        rois_label = torch.cat([rois_label, torch.zeros_like(rois_label)], 0)
        loss_bbox = smooth_l1(bbox_pred, rois_target, rw_in, rw_out)
        # 2-way loss with 1:2:1 hard-negative mining, dana.py:203-215
        fg = (rois_label == 1).nonzero().squeeze(-1)
        bg = (rois_label == 0).nonzero().squeeze(-1)
        sm = F.softmax(cls_score, 1)[bg, :]
        n_all = rois_label.shape[0]
        bg0 = max(1, min(fg.shape[0] * 2, int(n_all * 0.25)))
        bg1 = max(1, min(fg.shape[0], bg0))
        _, sidx = torch.sort(sm[:, 1], descending=True)
        real_bg = bg[sidx]
        top0 = real_bg[real_bg < int(n_all * 0.5)][:bg0]
        top1 = real_bg[real_bg >= int(n_all * 0.5)][:bg1]
        idx = torch.cat([fg, top0, top1], 0)
        loss_cls = F.cross_entropy(cls_score[idx], rois_label[idx])
    return rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, loss_cls, loss_bbox, rois_label


# ------------------------------------------------------------------------------------------------
# sibling model `meta` (Meta R-CNN): lib/model/framework/meta.py:39-142,241-251
# ------------------------------------------------------------------------------------------------
def mined_cross_entropy(cls_score, rois_label):
    """the 2-way loss with 1:2:1 hard-negative mining shared by dana.py:203-215, meta.py:111-123, fsod.py:163-175"""
    fg = (rois_label == 1).nonzero().squeeze(-1)
    bg = (rois_label == 0).nonzero().squeeze(-1)
    sm = F.softmax(cls_score, 1)[bg, :]
    n_all = rois_label.shape[0]
    bg0 = max(1, min(fg.shape[0] * 2, int(n_all * 0.25)))
    bg1 = max(1, min(fg.shape[0], bg0))
    _, sidx = torch.sort(sm[:, 1], descending=True)
    real_bg = bg[sidx]
    idx = torch.cat([fg, real_bg[real_bg < int(n_all * 0.5)][:bg0], real_bg[real_bg >= int(n_all * 0.5)][:bg1]], 0)
    return F.cross_entropy(cls_score[idx], rois_label[idx])


def meta_forward(sd, im_data, im_info, gt_boxes, num_boxes, support_ims, all_cls_gt_boxes, training, n_way=2, n_shot=3,
                 nms_inclusive=True, differentiable=False):
    """differentiable: RoIAlign as torch ops (roi_align_torch) so that autograd reaches the trunk (gradient tests)"""
    B = im_data.shape[0]
    base_feat = rcnn_base(im_data, sd)
    # PRN (meta.py:241-251): class-attentive vectors = sigmoid(layer4(maxpool2(base(support))).mean)
    sf = torch.sigmoid(rcnn_top(F.max_pool2d(rcnn_base(support_ims.reshape(-1, *support_ims.shape[2:]), sd), 2), sd))
    if training:
        sf = sf.view(-1, n_way * n_shot, sf.shape[1])
        pos, neg = sf[:, :n_shot].mean(1), sf[:, n_shot:n_way * n_shot].mean(1)
    else:
        pos = sf.view(-1, n_shot, sf.shape[1]).mean(1)
    cls, prob, bbox = rpn_head(base_feat, sd)
    rois = proposal_layer(prob.detach(), bbox.detach(), im_info, "TRAIN" if training else "TEST", nms_inclusive)
    rpn_loss_cls = rpn_loss_bbox = 0
    rois_label = None
    if training:
        H, W = cls.shape[2:]
        lab, tg, w_in, w_out = anchor_target_layer((H, W), all_cls_gt_boxes, im_info)  # meta.py:65: ALL classes' boxes
        sc = cls.view(B, 2, -1, W).permute(0, 2, 3, 1).reshape(-1, 2)
        keep = lab.view(-1).ne(-1).nonzero().view(-1)
        rpn_loss_cls = F.cross_entropy(sc[keep], lab.view(-1)[keep].long())
        rpn_loss_bbox = smooth_l1(bbox, tg, w_in, w_out, sigma=3, dims=(1, 2, 3))
        rois, rois_label, rois_target, rw_in, rw_out = proposal_target_layer(rois, gt_boxes)
        rois_label = rois_label.view(-1).long()
        rois_target, rw_in, rw_out = rois_target.view(-1, 4), rw_in.view(-1, 4), rw_out.view(-1, 4)
    if differentiable:
        pooled = roi_align_torch(base_feat, rois.view(-1, 5), 1.0 / 16.0, 7)
    else:
        pooled = torch.from_numpy(native.roi_align_forward(base_feat.detach().numpy(), rois.view(-1, 5).numpy(),
                                                           1.0 / 16.0, 7, 7, 0))
    fc7 = rcnn_top(pooled, sd)
    R = rois.size(1)

    def head(vec):  # meta.py:129-142
        comb = (fc7.view(B, R, -1) * vec.view(B, 1, -1)).view(B * R, -1)
        score = _lin(comb, sd, "RCNN_cls_score.0")
        return F.softmax(score, 1), score

    bbox_pred = _lin(fc7, sd, "RCNN_bbox_pred")
    cls_prob, cls_score = head(pos)
    loss_cls = loss_bbox = 0
    if training:
        neg_prob, neg_score = head(neg)
        cls_prob = torch.cat([cls_prob, neg_prob], 0)
        cls_score = torch.cat([cls_score, neg_score], 0)
        rois_label = torch.cat([rois_label, torch.zeros_like(rois_label)], 0)
        loss_bbox = smooth_l1(bbox_pred, rois_target, rw_in, rw_out)
        loss_cls = mined_cross_entropy(cls_score, rois_label)
    return rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, loss_cls, loss_bbox, rois_label


# ------------------------------------------------------------------------------------------------
# sibling model `fsod` (attention-RPN + multi-relation head): lib/model/framework/fsod.py:79-249
# ------------------------------------------------------------------------------------------------
def fsod_forward(sd, im_data, im_info, gt_boxes, num_boxes, support_ims, training, n_way=2, n_shot=3, nms_inclusive=True,
                 differentiable=False):
    """differentiable: RoIAlign as torch ops (roi_align_torch) so that autograd reaches the trunk (gradient tests)"""
    B = im_data.shape[0]
    base_feat = rcnn_base(im_data, sd)
    sup = rcnn_base(support_ims.reshape(-1, *support_ims.shape[2:]), sd)
    if training:
        sup = sup.view(-1, n_way * n_shot, *sup.shape[1:])
        pos = F.avg_pool2d(sup[:, :n_shot].mean(1), 14, 1)  # [B,1024,7,7] (fsod.py:98-101)
        neg = F.avg_pool2d(sup[:, n_shot:n_way * n_shot].mean(1), 14, 1)
    else:
        pos = F.avg_pool2d(sup.view(-1, n_shot, *sup.shape[1:]).mean(1), 14, 1)
    # attention RPN (fsod.py:109-116): depth-wise cross-correlation of the query map with the pooled support
    corr = torch.stack([F.conv2d(base_feat[b:b + 1], pos[b].view(1024, 1, 7, 7), groups=1024)[0] for b in range(B)], 0)
    cls, prob, bbox = rpn_head(corr, sd)
    rois = proposal_layer(prob.detach(), bbox.detach(), im_info, "TRAIN" if training else "TEST", nms_inclusive)
    rpn_loss_cls = rpn_loss_bbox = 0
    rois_label = None
    if training:
        H, W = cls.shape[2:]
        lab, tg, w_in, w_out = anchor_target_layer((H, W), gt_boxes, im_info)
        sc = cls.view(B, 2, -1, W).permute(0, 2, 3, 1).reshape(-1, 2)
        keep = lab.view(-1).ne(-1).nonzero().view(-1)
        rpn_loss_cls = F.cross_entropy(sc[keep], lab.view(-1)[keep].long())
        rpn_loss_bbox = smooth_l1(bbox, tg, w_in, w_out, sigma=3, dims=(1, 2, 3))
        rois, rois_label, rois_target, rw_in, rw_out = proposal_target_layer(rois, gt_boxes)
        rois_label = rois_label.view(-1).long()
        rois_target, rw_in, rw_out = rois_target.view(-1, 4), rw_in.view(-1, 4), rw_out.view(-1, 4)
    if differentiable:
        pooled = roi_align_torch(base_feat, rois.view(-1, 5), 1.0 / 16.0, 7)
    else:
        pooled = torch.from_numpy(native.roi_align_forward(base_feat.detach().numpy(), rois.view(-1, 5).numpy(),
                                                           1.0 / 16.0, 7, 7, 0))
    R = rois.size(1)
    n = B * R
    bbox_pred = _lin(rcnn_top(pooled, sd), sd, "RCNN_bbox_pred")

    def head(support):  # fsod.py:181-249, support [B,1024,7,7]
        srep = support.view(B, 1, 1024, 7, 7).expand(B, R, 1024, 7, 7).reshape(n, 1024, 7, 7)
        # global relation
        gf = torch.cat([pooled, srep], 1).mean(3).mean(2)
        g = _lin(F.relu(_lin(F.relu(_lin(gf, sd, "global_fc_1")), sd, "global_fc_2")), sd, "global_cls_score")
        # local correlation
        cr = F.conv2d(pooled, sd["corr_conv.weight"])
        cs = F.conv2d(support, sd["corr_conv.weight"])
        oc = (cr.view(B, R, 1024, 49) * cs.view(B, 1, 1024, 49)).sum(3).view(n, 1024)
        c = _lin(oc, sd, "corr_cls_score")
        # patch relation
        x = F.relu(F.conv2d(torch.cat([pooled, srep], 1), sd["patch_conv_1.weight"]))
        x = F.avg_pool2d(x, 3, 1)
        x = F.relu(F.conv2d(x, sd["patch_conv_2.weight"]))
        x = F.relu(F.conv2d(x, sd["patch_conv_3.weight"]))
        x = F.avg_pool2d(x, 3, 1).view(n, 1024)
        pt = _lin(x, sd, "patch_cls_score")
        score = (g + c + pt) / 10.0
        return F.softmax(score, 1), score

    cls_prob, cls_score = head(pos)
    loss_cls = loss_bbox = 0
    if training:
        neg_prob, neg_score = head(neg)
        cls_prob = torch.cat([cls_prob, neg_prob], 0)
        cls_score = torch.cat([cls_score, neg_score], 0)
        rois_label = torch.cat([rois_label, torch.zeros_like(rois_label)], 0)
        loss_bbox = smooth_l1(bbox_pred, rois_target, rw_in, rw_out)
        loss_cls = mined_cross_entropy(cls_score, rois_label)
    return rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, loss_cls, loss_bbox, rois_label


# ------------------------------------------------------------------------------------------------
# sibling model `fgn`: lib/model/framework/fgn.py:45-165. Its head's bn1 / bn2 are ordinary BatchNorm layers: batch
# statistics (and running-stat updates) in train mode, running statistics in eval mode.
# ------------------------------------------------------------------------------------------------
def fgn_forward(sd, im_data, im_info, gt_boxes, num_boxes, support_ims, training, n_way=2, n_shot=3, nms_inclusive=True,
                bn_state=None, differentiable=False):
    """bn_state: dict that receives the updated running statistics of bn1 / bn2 (train mode);
    differentiable: RoIAlign as torch ops (roi_align_torch) so that autograd reaches the trunk (gradient tests)"""
    B = im_data.shape[0]
    base_feat = rcnn_base(im_data, sd)
    sup = rcnn_base(support_ims.reshape(-1, *support_ims.shape[2:]), sd)
    if training:
        sup = sup.view(-1, n_way * n_shot, *sup.shape[1:])
        pos_map, neg_map = sup[:, :n_shot].mean(1), sup[:, n_shot:n_way * n_shot].mean(1)
        neg_rcnn = F.avg_pool2d(neg_map, 14, 1)
    else:
        pos_map = sup.view(-1, n_shot, *sup.shape[1:]).mean(1)
    pos_rpn = F.avg_pool2d(pos_map, 20)       # [B,1024,1,1]
    pos_rcnn = F.avg_pool2d(pos_map, 14, 1)   # [B,1024,7,7]
    cls, prob, bbox = rpn_head(base_feat * pos_rpn, sd)  # attention RPN (fgn.py:75-82)
    rois = proposal_layer(prob.detach(), bbox.detach(), im_info, "TRAIN" if training else "TEST", nms_inclusive)
    rpn_loss_cls = rpn_loss_bbox = 0
    rois_label = None
    if training:
        H, W = cls.shape[2:]
        lab, tg, w_in, w_out = anchor_target_layer((H, W), gt_boxes, im_info)
        sc = cls.view(B, 2, -1, W).permute(0, 2, 3, 1).reshape(-1, 2)
        keep = lab.view(-1).ne(-1).nonzero().view(-1)
        rpn_loss_cls = F.cross_entropy(sc[keep], lab.view(-1)[keep].long())
        rpn_loss_bbox = smooth_l1(bbox, tg, w_in, w_out, sigma=3, dims=(1, 2, 3))
        rois, rois_label, rois_target, rw_in, rw_out = proposal_target_layer(rois, gt_boxes)
        rois_label = rois_label.view(-1).long()
        rois_target, rw_in, rw_out = rois_target.view(-1, 4), rw_in.view(-1, 4), rw_out.view(-1, 4)
    if differentiable:
        pooled = roi_align_torch(base_feat, rois.view(-1, 5), 1.0 / 16.0, 7)
    else:
        pooled = torch.from_numpy(native.roi_align_forward(base_feat.detach().numpy(), rois.view(-1, 5).numpy(),
                                                           1.0 / 16.0, 7, 7, 0))
    R = rois.size(1)
    n = B * R
    bbox_pred = _lin(rcnn_top(pooled, sd), sd, "RCNN_bbox_pred")
    run = {k: sd[k].detach().clone() for k in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var")}

    def bn(x, p):
        return F.batch_norm(x, run[p + ".running_mean"], run[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            training, 0.1, 1e-5)

    def head(support):  # fgn.py:145-165
        comb = torch.cat([support.view(B, 1, 1024, 7, 7).expand(B, R, 1024, 7, 7).reshape(n, 1024, 7, 7), pooled], 1)
        x = F.relu(bn(F.conv2d(comb, sd["cls_conv1.weight"]), "bn1"))
        x = F.relu(bn(F.conv2d(x, sd["cls_conv2.weight"]), "bn2"))
        score = _lin(x.reshape(n, -1), sd, "RCNN_cls_score")
        return F.softmax(score, 1), score

    cls_prob, cls_score = head(pos_rcnn)
    loss_cls = loss_bbox = 0
    if training:
        neg_prob, neg_score = head(neg_rcnn)
        cls_prob = torch.cat([cls_prob, neg_prob], 0)
        cls_score = torch.cat([cls_score, neg_score], 0)
        rois_label = torch.cat([rois_label, torch.zeros_like(rois_label)], 0)
        loss_bbox = smooth_l1(bbox_pred, rois_target, rw_in, rw_out)
        loss_cls = mined_cross_entropy(cls_score, rois_label)
    if bn_state is not None:
        bn_state.update(run)
    return rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, loss_cls, loss_bbox, rois_label


# ------------------------------------------------------------------------------------------------
# sibling model on the same ops: plain Faster R-CNN, lib/model/framework/faster_rcnn.py:35-103
# ------------------------------------------------------------------------------------------------
def frcnn_forward(sd, im_data, im_info, gt_boxes, num_boxes, training, nms_inclusive=True, pooling="align",
                  differentiable=False):
    """differentiable: RoIAlign as torch ops (roi_align_torch) so that autograd reaches the trunk (gradient tests)"""
    B = im_data.shape[0]
    base_feat = rcnn_base(im_data, sd)
    cls, prob, bbox = rpn_head(base_feat, sd)
    rois = proposal_layer(prob.detach(), bbox.detach(), im_info, "TRAIN" if training else "TEST", nms_inclusive)
    rpn_loss_cls = rpn_loss_bbox = 0
    rois_label = None
    if training:
        H, W = cls.shape[2:]
        lab, tg, w_in, w_out = anchor_target_layer((H, W), gt_boxes, im_info)
        sc = cls.view(B, 2, -1, W).permute(0, 2, 3, 1).reshape(-1, 2)
        keep = lab.view(-1).ne(-1).nonzero().view(-1)
        rpn_loss_cls = F.cross_entropy(sc[keep], lab.view(-1)[keep].long())
        rpn_loss_bbox = smooth_l1(bbox, tg, w_in, w_out, sigma=3, dims=(1, 2, 3))
        rois, rois_label, rois_target, rw_in, rw_out = proposal_target_layer(rois, gt_boxes)
        rois_label = rois_label.view(-1).long()
        rois_target, rw_in, rw_out = rois_target.view(-1, 4), rw_in.view(-1, 4), rw_out.view(-1, 4)
    r5 = rois.view(-1, 5).numpy()
    if differentiable:
        assert pooling == "align"
        pooled = roi_align_torch(base_feat, rois.view(-1, 5), 1.0 / 16.0, 7)
    elif pooling == "align":
        pooled = torch.from_numpy(native.roi_align_forward(base_feat.detach().numpy(), r5, 1.0 / 16.0, 7, 7, 0))
    else:
        pooled = torch.from_numpy(native.roi_pool_forward(base_feat.detach().numpy(), r5, 1.0 / 16.0, 7, 7)[0])
    fc7 = rcnn_top(pooled, sd)
    bbox_pred = _lin(fc7, sd, "RCNN_bbox_pred")
    cls_score = _lin(fc7, sd, "RCNN_cls_score")
    cls_prob = F.softmax(cls_score, 1)
    loss_cls = loss_bbox = 0
    if training:
        loss_cls = F.cross_entropy(cls_score, rois_label)
        loss_bbox = smooth_l1(bbox_pred, rois_target, rw_in, rw_out)
    return (rois, cls_prob.view(B, rois.size(1), -1), bbox_pred.view(B, rois.size(1), -1), rpn_loss_cls, rpn_loss_bbox,
            loss_cls, loss_bbox, rois_label)


# ------------------------------------------------------------------------------------------------
# inference post-processing: inference.py:106-140, utils.py:312-317
# ------------------------------------------------------------------------------------------------
def postprocess(rois, cls_prob, bbox_pred, im_info, thresh=0.05, nms_thresh=0.3, nms_inclusive=True):
    """one image: -> cls_dets [K,5] (x1,y1,x2,y2,score) in descending score order"""
    t = CFG["TRAIN"]
    boxes = rois[:, :, 1:5]
    deltas = bbox_pred.view(-1, 4) * torch.tensor(t["BBOX_NORMALIZE_STDS"]) + torch.tensor(t["BBOX_NORMALIZE_MEANS"])
    pred = clip_boxes(bbox_transform_inv(boxes, deltas.view(1, -1, 4)), im_info)
    pred = pred / float(im_info[0][2])
    scores = cls_prob.squeeze()
    pred = pred.squeeze(0)
    inds = torch.nonzero(scores[:, 1] > thresh).view(-1)
    if inds.numel() == 0:
        return torch.zeros(0, 5)
    cls_scores, cls_boxes = scores[:, 1][inds], pred[inds]
    # utils.py:313 uses torch.sort's default (unstable) order; exact score ties are therefore undefined in the
    # reference -- the oracle (and the HIP path) break them by index (stable), tests exclude tied rows vs golden
    _, order = torch.sort(cls_scores, dim=0, descending=True, stable=True)
    dets = torch.cat((cls_boxes, cls_scores.unsqueeze(1)), 1)[order]
    keep = torch.from_numpy(native.nms(cls_boxes[order].numpy(), cls_scores[order].numpy(), nms_thresh, nms_inclusive))
    return dets[keep]
