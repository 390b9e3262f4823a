"""Training-only target assignment and losses of the DAnA path.

Host-side logic in torch ops on whatever device the tensors live on (GPU in the product path); the
sampling keeps the reference's host RNG stream (``np.random``) call for call so that a seeded run
reproduces the reference's choices. Moving these onto the device with a counter-based RNG is
SURVEY.md's "next" row N2.

Mirrors (semantics; same names and argument meaning):
  _AnchorTargetLayer.forward    lib/model/rpn/anchor_target_layer.py:48-193
  _ProposalTargetLayer.forward  lib/model/rpn/proposal_target_layer_cascade.py:33-213
  bbox_overlaps_batch / bbox_transform_batch   lib/model/rpn/bbox_transform.py:36-75,168-257
  _smooth_l1_loss               lib/model/utils/net_utils.py:71-85
"""
import numpy as np
import torch

from .config import cfg


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """lib/model/rpn/generate_anchors.py:45-105 -> float64 [len(ratios)*len(scales), 4], ratio-major."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    ctr = 0.5 * (base_size - 1)
    area = float(base_size) * float(base_size)
    ws = np.round(np.sqrt(area / ratios))  # banker's rounding, as numpy in the reference (:91-92)
    hs = np.round(ws * ratios)
    w = (ws[:, None] * scales[None, :]).reshape(-1)
    h = (hs[:, None] * scales[None, :]).reshape(-1)
    return np.stack([ctr - 0.5 * (w - 1), ctr - 0.5 * (h - 1), ctr + 0.5 * (w - 1), ctr + 0.5 * (h - 1)], 1)


def shifted_anchors(base_anchors, feat_h, feat_w, feat_stride, device):
    """proposal_layer.py:80-93 / anchor_target_layer.py:66-79: all anchors, (h, w, a) order, fp32."""
    sx = torch.arange(feat_w, dtype=torch.float32, device=device) * feat_stride
    sy = torch.arange(feat_h, dtype=torch.float32, device=device) * feat_stride
    shifts = torch.stack([sx.repeat(feat_h), sy.repeat_interleave(feat_w), sx.repeat(feat_h),
                          sy.repeat_interleave(feat_w)], 1)
    return (base_anchors.view(1, -1, 4) + shifts.view(-1, 1, 4)).reshape(-1, 4)


def bbox_overlaps_batch(anchors, gt_boxes):
    B = gt_boxes.size(0)
    if anchors.dim() == 2:
        anchors = anchors.unsqueeze(0).expand(B, -1, -1)
    elif anchors.size(2) == 5:
        anchors = anchors[:, :, 1:5]
    gt = gt_boxes[:, :, :4]
    gw = gt[:, :, 2] - gt[:, :, 0] + 1
    gh = gt[:, :, 3] - gt[:, :, 1] + 1
    aw = anchors[:, :, 2] - anchors[:, :, 0] + 1
    ah = anchors[:, :, 3] - anchors[:, :, 1] + 1
    iw = (torch.min(anchors[:, :, None, 2], gt[:, None, :, 2]) -
          torch.max(anchors[:, :, None, 0], gt[:, None, :, 0]) + 1).clamp_(min=0)
    ih = (torch.min(anchors[:, :, None, 3], gt[:, None, :, 3]) -
          torch.max(anchors[:, :, None, 1], gt[:, None, :, 1]) + 1).clamp_(min=0)
    inter = iw * ih
    ov = inter / ((aw * ah).unsqueeze(2) + (gw * gh).unsqueeze(1) - inter)
    ov.masked_fill_(((gw == 1) & (gh == 1)).unsqueeze(1), 0)   # zero-area gt (padding rows)
    ov.masked_fill_(((aw == 1) & (ah == 1)).unsqueeze(2), -1)  # zero-area anchors/rois
    return ov


def bbox_transform_batch(ex_rois, gt_rois):
    ew = ex_rois[..., 2] - ex_rois[..., 0] + 1.0
    eh = ex_rois[..., 3] - ex_rois[..., 1] + 1.0
    ecx = ex_rois[..., 0] + 0.5 * ew
    ecy = ex_rois[..., 1] + 0.5 * eh
    gw = gt_rois[..., 2] - gt_rois[..., 0] + 1.0
    gh = gt_rois[..., 3] - gt_rois[..., 1] + 1.0
    gcx = gt_rois[..., 0] + 0.5 * gw
    gcy = gt_rois[..., 1] + 0.5 * gh
    return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), -1)


def _smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma=1.0, dim=[1]):
    sigma_2 = sigma ** 2
    in_box_diff = bbox_inside_weights * (bbox_pred - bbox_targets)
    abs_diff = torch.abs(in_box_diff)
    near = (abs_diff < 1. / sigma_2).detach().float()
    loss = bbox_outside_weights * (torch.pow(in_box_diff, 2) * (sigma_2 / 2.) * near +
                                   (abs_diff - (0.5 / sigma_2)) * (1. - near))
    for i in sorted(dim, reverse=True):
        loss = loss.sum(i)
    return loss.mean()


def _host_perm(n, device):
    return torch.from_numpy(np.random.permutation(n)).long().to(device)


def anchor_target_layer(feat_h, feat_w, gt_boxes, im_info, base_anchors):
    """-> labels [B,1,A*H,W], bbox_targets / inside / outside weights [B,4A,H,W]."""
    tr = cfg.TRAIN
    dev = gt_boxes.device
    B = gt_boxes.size(0)
    A = base_anchors.size(0)
    all_anchors = shifted_anchors(base_anchors, feat_h, feat_w, cfg.FEAT_STRIDE[0], dev)
    total = all_anchors.size(0)
    im_h, im_w = int(im_info[0][0]), int(im_info[0][1])  # image 0 decides for the whole batch (:85-86)
    inside = ((all_anchors[:, 0] >= 0) & (all_anchors[:, 1] >= 0) & (all_anchors[:, 2] < im_w) &
              (all_anchors[:, 3] < im_h))
    inds_inside = torch.nonzero(inside).view(-1)
    anchors = all_anchors[inds_inside]
    n = inds_inside.numel()

    overlaps = bbox_overlaps_batch(anchors, gt_boxes)
    max_overlaps, argmax_overlaps = overlaps.max(2)
    gt_max_overlaps, _ = overlaps.max(1)
    labels = gt_boxes.new_full((B, n), -1)
    if not tr.RPN_CLOBBER_POSITIVES:
        labels[max_overlaps < tr.RPN_NEGATIVE_OVERLAP] = 0
    gt_max_overlaps[gt_max_overlaps == 0] = 1e-5
    is_gt_best = overlaps.eq(gt_max_overlaps.view(B, 1, -1)).sum(2)
    labels[is_gt_best > 0] = 1
    labels[max_overlaps >= tr.RPN_POSITIVE_OVERLAP] = 1
    if tr.RPN_CLOBBER_POSITIVES:
        labels[max_overlaps < tr.RPN_NEGATIVE_OVERLAP] = 0

    num_fg = int(tr.RPN_FG_FRACTION * tr.RPN_BATCHSIZE)
    sum_fg = (labels == 1).sum(1).tolist()
    sum_bg = (labels == 0).sum(1).tolist()
    for i in range(B):
        if sum_fg[i] > num_fg:
            fg_inds = torch.nonzero(labels[i] == 1).view(-1)
            rnd = _host_perm(fg_inds.numel(), dev)
            labels[i][fg_inds[rnd[:fg_inds.numel() - num_fg]]] = -1
        num_bg = tr.RPN_BATCHSIZE - min(sum_fg[i], num_fg)
        if sum_bg[i] > num_bg:
            bg_inds = torch.nonzero(labels[i] == 0).view(-1)
            rnd = _host_perm(bg_inds.numel(), dev)
            labels[i][bg_inds[rnd[:bg_inds.numel() - num_bg]]] = -1

    matched = torch.gather(gt_boxes[:, :, :4], 1, argmax_overlaps.unsqueeze(2).expand(-1, -1, 4))
    bbox_targets = bbox_transform_batch(anchors.unsqueeze(0), matched)
    inside_w = (labels == 1).float() * tr.RPN_BBOX_INSIDE_WEIGHTS[0]
    assert tr.RPN_POSITIVE_WEIGHT < 0
    num_examples = int((labels[B - 1] >= 0).sum())  # the LAST image's count is used for all (:156)
    outside_w = (labels >= 0).float() * (1.0 / num_examples)

    def unmap(data, fill):
        shape = (B, total) + tuple(data.shape[2:])
        ret = data.new_full(shape, fill)
        ret[:, inds_inside] = data
        return ret

    labels = unmap(labels, -1).view(B, feat_h, feat_w, A).permute(0, 3, 1, 2).reshape(B, 1, A * feat_h, feat_w)
    bbox_targets = unmap(bbox_targets, 0).view(B, feat_h, feat_w, A * 4).permute(0, 3, 1, 2).contiguous()

    def expand4(wt):
        return unmap(wt, 0).view(B, total, 1).expand(B, total, 4).reshape(B, feat_h, feat_w, 4 * A) \
            .permute(0, 3, 1, 2).contiguous()

    return labels, bbox_targets, expand4(inside_w), expand4(outside_w)


def proposal_target_layer(all_rois, gt_boxes, num_boxes=None):
    """-> rois [B,R,5], labels [B,R], bbox_targets, inside weights, outside weights [B,R,4]."""
    tr = cfg.TRAIN
    dev = gt_boxes.device
    B = gt_boxes.size(0)
    gt_as_rois = torch.zeros_like(gt_boxes)
    gt_as_rois[:, :, 1:5] = gt_boxes[:, :, :4]
    all_rois = torch.cat([all_rois, gt_as_rois], 1)
    rois_per_image = int(tr.BATCH_SIZE)
    fg_rois_per_image = int(np.round(tr.FG_FRACTION * rois_per_image)) or 1

    overlaps = bbox_overlaps_batch(all_rois, gt_boxes)
    max_overlaps, gt_assignment = overlaps.max(2)
    labels = torch.gather(gt_boxes[:, :, 4], 1, gt_assignment)
    fg_mask = max_overlaps >= tr.FG_THRESH
    bg_mask = (max_overlaps < tr.BG_THRESH_HI) & (max_overlaps >= tr.BG_THRESH_LO)
    fg_counts = fg_mask.sum(1).tolist()
    bg_counts = bg_mask.sum(1).tolist()

    keep_all, fg_taken = [], []
    for i in range(B):
        nf, nb = fg_counts[i], bg_counts[i]
        fg_inds = torch.nonzero(fg_mask[i]).view(-1)
        bg_inds = torch.nonzero(bg_mask[i]).view(-1)
        if nf > 0 and nb > 0:
            fg_n = min(fg_rois_per_image, nf)
            fg_inds = fg_inds[_host_perm(nf, dev)[:fg_n]]
            pick = np.floor(np.random.rand(rois_per_image - fg_n) * nb)
            bg_inds = bg_inds[torch.from_numpy(pick).long().to(dev)]
        elif nf > 0:
            pick = np.floor(np.random.rand(rois_per_image) * nf)
            fg_inds = fg_inds[torch.from_numpy(pick).long().to(dev)]
            fg_n = rois_per_image
            bg_inds = bg_inds[:0]
        elif nb > 0:
            pick = np.floor(np.random.rand(rois_per_image) * nb)
            bg_inds = bg_inds[torch.from_numpy(pick).long().to(dev)]
            fg_n = 0
            fg_inds = fg_inds[:0]
        else:
            raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
        keep_all.append(torch.cat([fg_inds, bg_inds], 0))
        fg_taken.append(fg_n)
    keep = torch.stack(keep_all, 0)  # [B, R]
    is_fg_slot = torch.arange(rois_per_image, device=dev).unsqueeze(0) < torch.tensor(fg_taken, device=dev).unsqueeze(1)
    labels_batch = torch.gather(labels, 1, keep) * is_fg_slot.float()
    rois_batch = torch.gather(all_rois, 1, keep.unsqueeze(2).expand(-1, -1, 5)).clone()
    rois_batch[:, :, 0] = torch.arange(B, device=dev, dtype=rois_batch.dtype).unsqueeze(1)
    gt_sel = torch.gather(gt_boxes, 1, torch.gather(gt_assignment, 1, keep).unsqueeze(2).expand(-1, -1, 5))

    targets = bbox_transform_batch(rois_batch[:, :, 1:5], gt_sel[:, :, :4])
    if tr.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
        means = torch.tensor(tr.BBOX_NORMALIZE_MEANS, device=dev)
        stds = torch.tensor(tr.BBOX_NORMALIZE_STDS, device=dev)
        targets = (targets - means) / stds
    pos = (labels_batch > 0).unsqueeze(2)
    bbox_targets = torch.where(pos, targets, torch.zeros_like(targets))
    inside_w = pos.float() * torch.tensor(tr.BBOX_INSIDE_WEIGHTS, device=dev)
    outside_w = (inside_w > 0).float()
    return rois_batch, labels_batch, bbox_targets, inside_w, outside_w
