"""Portable seeded weights and inputs: a counter-based generator
(splitmix64 -> Box-Muller, keyed by tensor name) so that this build, the oracle and the reference
imported in the build container all see bit-identical random-init weights without shipping 150 MB.
Used by bench.py (random-init weights of the reference architecture, synthetic inputs) and tests."""
import zlib

import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def _uniform(name, n, seed, stream=0):
    key = np.uint64(zlib.crc32(name.encode()) + (seed << 32) + (stream << 56))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + _splitmix64(key)
        bits = _splitmix64(idx)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)  # (0,1)


def normal(name, shape, std=1.0, mean=0.0, seed=0):
    n = int(np.prod(shape))
    u1, u2 = _uniform(name, n, seed, 0), _uniform(name, n, seed, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((z * std + mean).astype(np.float32).reshape(shape))


def uniform(name, shape, lo=0.0, hi=1.0, seed=0):
    n = int(np.prod(shape))
    return torch.from_numpy((_uniform(name, n, seed, 2) * (hi - lo) + lo).astype(np.float32).reshape(shape))


def fill_state_dict(template, seed=0, profile="init"):
    """template: a DAnARCNN state_dict (names+shapes) -> new dict of seeded tensors.
    profile "init": the reference's init scales (resnet.py:122-128 He-normal convs, unit BN;
    dana.py:45-69,222-238 N(0, .01)/N(0, .001) heads, zero biases).
    profile "test": same conv/linear scales, but non-trivial frozen-BN statistics chosen so that
    activations stay O(1) through the residual trunk (a random-init net with unit BN blows up to ~1e4
    and saturates every softmax / overflows the unclamped exp of bbox_transform_inv), and slightly
    larger score-head scales so that scores and attention are well spread (robust ordering)."""
    out = {}
    trunk = lambda k: k.startswith("RCNN_base") or k.startswith("RCNN_top")  # noqa: E731
    for k, t in template.items():
        shape = tuple(t.shape)
        test = profile == "test"
        stem_bn = k.startswith("RCNN_base.1.")
        last_bn = ".bn3." in k  # closes the residual branch: keep it small so activations stay O(1)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=t.dtype)
        elif "running_mean" in k:
            out[k] = normal(k, shape, 0.1, 0.0, seed) if test else torch.zeros(shape)
        elif "running_var" in k:
            if not test:
                out[k] = torch.ones(shape)
            elif stem_bn:  # inputs are pixel-scaled (~N(0,64^2)): the stem BN brings them back to O(1)
                out[k] = uniform(k, shape, 300.0, 500.0, seed)
            else:
                out[k] = uniform(k, shape, 0.5, 1.5, seed)
        elif len(shape) == 4 and trunk(k):
            n = shape[2] * shape[3] * shape[0]
            out[k] = normal(k, shape, float(np.sqrt(2.0 / n)), 0.0, seed)
        elif trunk(k) and k.endswith("weight"):  # BN gamma
            if not test:
                out[k] = torch.ones(shape)
            elif last_bn:
                out[k] = uniform(k, shape, 0.2, 0.5, seed)
            else:
                out[k] = uniform(k, shape, 0.7, 1.3, seed)
        elif trunk(k) and k.endswith("bias"):  # BN beta
            out[k] = normal(k, shape, 0.1, 0.0, seed) if test else torch.zeros(shape)
        elif k.endswith("bias"):
            out[k] = normal(k, shape, 0.05, 0.0, seed) if test else torch.zeros(shape)
        elif k.startswith("RCNN_bbox_pred"):
            out[k] = normal(k, shape, 0.001, 0.0, seed)
        elif k.startswith("output_score_layer") or k.startswith("rcnn_transform_layer"):
            b = 1.0 / float(np.sqrt(shape[-1]))  # nn.Linear default init range (dana.py:76,78 leave it)
            out[k] = uniform(k, shape, -b, b, seed)
        else:
            std = 0.01
            if test and ("cls_score" in k or "unary" in k or "channel_k" in k):
                std = 0.03
            out[k] = normal(k, shape, std, 0.0, seed)
    return out


def episode_inputs(batch, way, shot, height=600, width=1000, seed=1996, support_size=320, max_gt=50):
    """SURVEY.md 8d synthetic episode: query ~ N(0,1)*64, supports ~ N(0,1)*64, 3 gt boxes/image."""
    im_data = normal("im_data", (batch, 3, height, width), 64.0, 0.0, seed)
    support = normal("support_ims", (batch, way * shot, 3, support_size, support_size), 64.0, 0.0, seed)
    im_info = torch.tensor([[float(height), float(width), 1.0]] * batch)
    gt = torch.zeros(batch, max_gt, 5)
    u = uniform("gt_boxes", (batch, 3, 4), 0.0, 1.0, seed)
    for b in range(batch):
        for j in range(3):
            side_w = 64 + u[b, j, 2].item() * (min(400, width - 2) - 64)
            side_h = 64 + u[b, j, 3].item() * (min(400, height - 2) - 64)
            x1 = u[b, j, 0].item() * (width - 1 - side_w)
            y1 = u[b, j, 1].item() * (height - 1 - side_h)
            gt[b, j] = torch.tensor([x1, y1, x1 + side_w, y1 + side_h, 1.0])
    num_boxes = torch.full((batch,), 3, dtype=torch.int64)
    return im_data, im_info, gt, num_boxes, support


def tame_fsod_weights(sd):
    """test-profile weights for the `fsod` sibling: its depth-wise correlations sum 49 products per channel, so the
    layers that consume them are scaled down to keep the RPN / head logits O(1) (saturated logits make every score
    tie and the proposal order arbitrary). Applied identically by the golden generator and the tests."""
    sd = dict(sd)
    sd["RCNN_rpn.RPN_Conv.weight"] = sd["RCNN_rpn.RPN_Conv.weight"] * 0.01
    sd["corr_cls_score.weight"] = sd["corr_cls_score.weight"] * 0.02
    return sd


def tame_product_weights(sd):
    """test-profile weights for attention_type='product' (dana.py:155-156,285-286): query * attended has ~10x the magnitude
    of the concatenated features, so the two layers that consume it are scaled down to keep the RPN objectness logits and
    the class scores away from saturation (the reference's CPU NMS re-sorts tied scores in an unstable order,
    nms_cpu.cpp:24: saturated scores make the proposal order arbitrary). Applied identically by the golden generator and
    the tests."""
    sd = dict(sd)
    sd["RCNN_rpn.RPN_Conv.weight"] = sd["RCNN_rpn.RPN_Conv.weight"] * 0.05
    sd["rcnn_transform_layer.weight"] = sd["rcnn_transform_layer.weight"] * 0.05
    return sd


def tame_res101_weights(sd):
    """test-profile weights for the opt-in resnet101 trunk (DAnARCNN.trunk_layers = (3, 4, 23, 3)): 17 more residual
    blocks in layer3 leave base_feat ~6x larger than the res50 profile was tuned for, so the RPN conv is scaled down to
    keep objectness logits and box deltas away from saturation / the unclamped exp of bbox_transform_inv. Applied
    identically by the tests and bench.py."""
    sd = dict(sd)
    sd["RCNN_rpn.RPN_Conv.weight"] = sd["RCNN_rpn.RPN_Conv.weight"] * 0.1
    return sd


def tame_fgn_weights(sd):
    """test-profile weights for the `fgn` sibling: its RPN runs on base_feat * mean(support) (magnitude ~7x base_feat),
    so RPN_Conv is scaled down to keep the objectness logits away from saturation (exactly tied scores make the proposal
    order arbitrary). Applied identically by the golden generator and the tests."""
    sd = dict(sd)
    sd["RCNN_rpn.RPN_Conv.weight"] = sd["RCNN_rpn.RPN_Conv.weight"] * 0.1
    return sd
