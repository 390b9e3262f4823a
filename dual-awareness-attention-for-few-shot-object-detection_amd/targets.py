"""Anchor table and the box-regression loss of the DAnA path. The target layers themselves
(rows a12, a13) are device kernels: csrc/targets.hip, csrc/sampling.hip behind ops.anchor_target_* / ops.proposal_target_*.

Mirrors (semantics; same names and argument meaning):
  generate_anchors   lib/model/rpn/generate_anchors.py:45-105
  _smooth_l1_loss    lib/model/utils/net_utils.py:71-85
"""
import numpy as np
import torch



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
