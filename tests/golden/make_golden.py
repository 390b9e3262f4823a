"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (imported from
/root/reference in the build container, see ref_import.py) on seeded synthetic inputs and portable
seeded weights, and assert on the way that the oracle (oracle/) reproduces the reference
bit-for-bit on the CPU. Fixtures are data only (inputs are regenerated from the seeded generator;
outputs and a few strided intermediates are stored). Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]

import ref_import as R  # noqa: E402
from oracle import model_ref as O, native as ON  # noqa: E402
import dana_amd.synthetic as S  # noqa: E402  (pure numpy/torch helper, no HIP needed)

torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


def rand_boxes(rng, n, w=1000.0, h=600.0, cluster=True):
    """boxes with heavy mutual overlap (clusters) so NMS has work to do"""
    if cluster:
        c = rng.uniform([0, 0], [w, h], size=(max(n // 8, 1), 2))
        ctr = c[rng.integers(0, len(c), n)] + rng.normal(0, 12, size=(n, 2))
    else:
        ctr = rng.uniform([0, 0], [w, h], size=(n, 2))
    wh = rng.uniform(16, 200, size=(n, 2))
    b = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
    b[:, 0::2] = b[:, 0::2].clip(0, w - 1)
    b[:, 1::2] = b[:, 1::2].clip(0, h - 1)
    return b.astype(np.float32)


def op_goldens(ref):
    C = ref["C"]
    rng = np.random.default_rng(1996)
    out = {}
    # anchors (generate_anchors.py:45) for both scale sets
    for tag, scales in (("a4", [4, 8, 16, 32]), ("a3", [8, 16, 32])):
        a_ref = ref["generate_anchors"].generate_anchors(scales=np.array(scales), ratios=np.array([0.5, 1, 2]))
        a_or = O.generate_anchors(scales=scales, ratios=[0.5, 1, 2])
        assert np.array_equal(a_ref, a_or), "anchors mismatch"
        out["anchors_" + tag] = a_ref
    # nms (reference CPU op, >=)
    for tag, n, thr in (("n256_t03", 256, 0.3), ("n256_t07", 256, 0.7), ("n2048_t07", 2048, 0.7)):
        boxes = rand_boxes(rng, n)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        keep_ref = C.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
        keep_or = ON.nms(boxes, scores, thr, inclusive=True)
        assert np.array_equal(keep_ref, keep_or), "nms mismatch " + tag
        out["nms_%s_boxes" % tag], out["nms_%s_scores" % tag], out["nms_%s_keep" % tag] = boxes, scores, keep_ref
        out["nms_%s_keep_gt" % tag] = ON.nms(boxes, scores, thr, inclusive=False)  # CUDA-variant (>), oracle only
    # exact-tie IoU case: IoU(a,b) = 0.5 exactly -> CPU op (>=) suppresses at thr 0.5, CUDA op (>) keeps
    tb = np.array([[0, 0, 9, 9], [0, 5, 9, 14], [100, 100, 120, 120]], dtype=np.float32)  # inter 50, union 150 -> 1/3
    tb2 = np.array([[0, 0, 9, 9], [0, 0, 9, 4], [50, 50, 60, 60]], dtype=np.float32)  # inter 50, union 100 -> 0.5
    ts = np.array([0.9, 0.8, 0.7], dtype=np.float32)
    k = C.nms(torch.from_numpy(tb2), torch.from_numpy(ts), 0.5).numpy()
    assert np.array_equal(k, ON.nms(tb2, ts, 0.5, True)) and list(k) == [0, 2]
    assert list(ON.nms(tb2, ts, 0.5, False)) == [0, 1, 2]
    out["nms_tie_boxes"], out["nms_tie_scores"], out["nms_tie_keep_ge"] = tb2, ts, k
    # roi_align (reference CPU op), incl. degenerate / out-of-bounds / zero-padded rois
    feat = rng.normal(0, 1, size=(2, 8, 38, 63)).astype(np.float32)
    rois = np.zeros((32, 5), dtype=np.float32)
    bx = rand_boxes(rng, 32, cluster=False)
    rois[:, 1:] = bx
    rois[:, 0] = rng.integers(0, 2, 32)
    rois[0] = [0, 0, 0, 0, 0]                     # zero-padded roi (proposal_layer.py:186-188)
    rois[1] = [1, 50, 60, 50, 60]                 # degenerate (w=h=0 -> clamped to 1)
    rois[2] = [0, -40, -40, 30, 30]               # partly outside (negative)
    rois[3] = [1, 900, 500, 1200, 800]            # beyond the right/bottom edge
    rois[4] = [0, 0, 0, 999, 599]                 # whole image (max adaptive grid 6x9)
    rois[5] = [1, 1100, 700, 1300, 900]           # entirely outside -> zeros
    r_ref = C.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1.0 / 16, 7, 7, 0).numpy()
    r_or = ON.roi_align_forward(feat, rois, 1.0 / 16, 7, 7, 0)
    assert np.array_equal(r_ref, r_or), "roi_align mismatch"
    r_ref2 = C.roi_align_forward(torch.from_numpy(feat), torch.from_numpy(rois), 1.0 / 16, 7, 7, 2).numpy()
    assert np.array_equal(r_ref2, ON.roi_align_forward(feat, rois, 1.0 / 16, 7, 7, 2))
    out["ra_feat"], out["ra_rois"], out["ra_out_sr0"], out["ra_out_sr2"] = feat, rois, r_ref, r_ref2
    # bbox_transform_inv + clip_boxes (bbox_transform.py:77-133)
    bt = ref["bbox_transform"]
    A4 = torch.from_numpy(out["anchors_a4"]).float()
    H, W = 5, 7
    anchors = O.anchor_grid(out["anchors_a4"], H, W, 16)
    deltas = torch.from_numpy(rng.normal(0, 0.5, size=(2, H * W * 12, 4)).astype(np.float32))
    im_info = torch.tensor([[80.0, 112.0, 1.0], [70.0, 100.0, 1.0]])
    p_ref = bt.clip_boxes(bt.bbox_transform_inv(anchors.unsqueeze(0).expand(2, -1, 4), deltas, 2), im_info, 2)
    p_or = O.clip_boxes(O.bbox_transform_inv(anchors.unsqueeze(0).expand(2, -1, 4), deltas), im_info)
    assert torch.equal(p_ref, p_or), "decode mismatch"
    for b in range(2):
        p_c = ON.decode_clip(A4.numpy(), deltas[b].numpy(), H, W, 16, float(im_info[b, 0]), float(im_info[b, 1]))
        assert np.abs(p_c - p_ref[b].numpy()).max() < 1e-3  # libm expf vs torch.exp: few-ulp
    out["dec_deltas"], out["dec_im_info"], out["dec_out"] = deltas.numpy(), im_info.numpy(), p_ref.numpy()
    # bbox_overlaps_batch incl. zero-area masks (bbox_transform.py:168-257)
    gt = torch.zeros(2, 6, 5)
    gt[:, :3, :4] = torch.from_numpy(rand_boxes(rng, 6, 112, 80, False)).view(2, 3, 4)
    gt[:, :3, 4] = 1
    an = anchors.clone()
    an[5] = 0
    ov_ref = bt.bbox_overlaps_batch(an, gt)
    assert torch.equal(ov_ref, O.bbox_overlaps_batch(an, gt)), "overlaps mismatch"
    out["ov_anchors"], out["ov_gt"], out["ov_out"] = an.numpy(), gt.numpy(), ov_ref.numpy()
    # positional encodings (dana.py:309-320)
    for L in (49, 400):
        pe_ref = ref["dana"].PositionalEncoding(max_len=L).pe
        assert torch.equal(pe_ref, O.positional_encoding(L))
        out["pe%d_sample" % L] = pe_ref[0, ::7, ::37].numpy()
    save("ops", **out)


def R_iou(a, b):
    """row-wise IoU (+1 convention) of two [n,4] box arrays"""
    x1, y1 = np.maximum(a[:, 0], b[:, 0]), np.maximum(a[:, 1], b[:, 1])
    x2, y2 = np.minimum(a[:, 2], b[:, 2]), np.minimum(a[:, 3], b[:, 3])
    it = np.clip(x2 - x1 + 1, 0, None) * np.clip(y2 - y1 + 1, 0, None)
    return it / ((a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1) + (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1) - it)


def e2e(tag, use_ba, training, B, way, shot, H, W, nms_seed=7, slim=False, attention_type="concat"):
    """Run reference + oracle on the seeded episode; store reference outputs (slim: the eight outputs only, no strided
    intermediates -- the full-size train-mode fixtures stay a few tens of KB)."""
    m = R.build_model(use_ba, way, shot, attention_type)
    sd = S.fill_state_dict(m.state_dict(), seed=11, profile="test")
    if attention_type == "product":
        sd = S.tame_product_weights(sd)
    m.load_state_dict(sd)
    im_data, im_info, gt, nb, sup = S.episode_inputs(B, way if training else 1, shot, H, W, seed=1996)
    m.train() if training else m.eval()
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_ref = m(im_data, im_info, gt, nb, sup)
    np.random.seed(nms_seed)
    inter = {}
    with torch.no_grad():
        out_or = O.forward(sd, im_data, im_info, gt, nb, sup, training, way, shot, use_ba, nms_inclusive=True,
                           inter=inter)
    names = ["rois", "cls_prob", "bbox_pred", "rpn_loss_cls", "rpn_loss_bbox", "RCNN_loss_cls", "RCNN_loss_bbox",
             "rois_label"]
    if training and slim and (out_ref[0] - out_or[0]).abs().max().item() > 1e-3:
        # At 600x1000 the reference and the oracle each sort ~21 500 scores and run NMS over 12 000 boxes per image: a
        # near-tie (the oracle re-associates a few bmm / linear calls: roundoff-level differences) flips ONE discrete
        # decision, the candidate list shifts by a slot and the same np.random stream then draws another subset. That is
        # not an error of either side, and nothing position-wise survives it -- so the oracle is pinned STAGE-WISE here:
        # the reference's own sampled batch goes in (sampled_targets: the regression targets are a deterministic function
        # of rois + labels), and everything downstream of the sampling must agree; the RPN losses do not depend on it.
        n_roi = out_ref[0].shape[0] * out_ref[0].shape[1]
        prop_ref = out_ref[0]
        frac = float((R_iou(prop_ref.reshape(-1, 5)[:, 1:].numpy(), out_or[0].reshape(-1, 5)[:, 1:].numpy()) >= 1 - 1e-3).mean())
        print("  sampled rois differ position-wise (%.1f%% equal): a flipped near-tie upstream; pinning stage-wise" % (100 * frac))
        inj = O.sampled_targets(out_ref[0], out_ref[7][:n_roi].float().view(out_ref[0].shape[0], -1), gt)
        np.random.seed(nms_seed)
        with torch.no_grad():
            out_or = O.forward(sd, im_data, im_info, gt, nb, sup, training, way, shot, use_ba, nms_inclusive=True,
                               sampled=inj)
    store = {}
    for n, a, b in zip(names, out_ref, out_or):
        if a is None or (not torch.is_tensor(a) and a == 0):
            assert b is None or (not torch.is_tensor(b) and b == 0), n
            continue
        a, b = a.detach(), b.detach()
        diff = (a.float() - b.float()).abs().max().item() if a.numel() else 0.0
        print("  %-16s ref-vs-oracle max|d| = %.3e" % (n, diff))
        # the oracle re-associates a few bmm/linear calls (contiguous vs strided views): roundoff-level only
        tol = {"rois": 1e-3, "rois_label": 0.0}.get(n, 2e-5)
        assert diff <= tol, (n, diff)
        store[n] = a.numpy()
    if slim:
        store["meta"] = np.array([int(use_ba), int(training), B, way, shot, H, W, 11, 1996, nms_seed,
                                  int(attention_type == "product")])
        save("e2e_" + tag, **store)
        return
    # strided intermediates from the ORACLE trace (pinned to the reference through the outputs above)
    store["base_feat_s"] = inter["base_feat"][:, ::16].numpy()
    store["dense_s"] = inter["dense_support_feature"][:, ::16].numpy()
    store["rpn_cls_score"] = inter["rpn_cls_score"].numpy()
    store["rpn_bbox_pred"] = inter["rpn_bbox_pred"].numpy()
    store["pooled_s"] = inter["pooled_feat"][:, ::32].numpy()
    if not training and B == 1:
        # inference post-processing with the reference's own functions (inference.py:106-140, utils.py:312-317)
        ref = R.load()
        bt, C = ref["bbox_transform"], ref["C"]
        rois_r, prob_r, pred_r = out_ref[0], out_ref[1], out_ref[2]
        deltas = pred_r.view(-1, 4) * torch.FloatTensor((0.1, 0.1, 0.2, 0.2)) + torch.FloatTensor((0.0, 0.0, 0.0, 0.0))
        pb = bt.clip_boxes(bt.bbox_transform_inv(rois_r[:, :, 1:5], deltas.view(1, -1, 4), 1), im_info, 1)
        pb = (pb / im_info[0][2].item()).squeeze()
        sc = prob_r.squeeze()
        for thr_tag, thr in (("t05", 0.05), ("t62", 0.62)):
            inds = torch.nonzero(sc[:, 1] > thr).view(-1)
            cs, cb = sc[:, 1][inds], pb[inds, :]
            _, order = torch.sort(cs, 0, True)
            dets = torch.cat((cb, cs.unsqueeze(1)), 1)[order]
            keep = C.nms(cb[order, :], cs[order], 0.3)
            dets = dets[keep.view(-1).long()]
            d_or = O.postprocess(out_or[0], out_or[1], out_or[2], im_info, thresh=thr)
            # exact score ties (degenerate border rois pool identical features) are ordered by torch.sort's
            # unstable inner sort in the reference's nms (nms_cpu.cpp:24): compare the untied detections only
            vals, cnts = np.unique(sc[:, 1].numpy(), return_counts=True)
            tied = set(vals[cnts > 1].tolist())
            untied = lambda d: d[[i for i in range(d.shape[0]) if float(d[i, 4]) not in tied]]  # noqa: E731
            a, b_ = untied(dets), untied(d_or)
            assert a.shape == b_.shape and (a - b_).abs().max().item() <= 1e-4, "postprocess mismatch"
            store["dets_" + thr_tag] = dets.numpy()
            store["dets_tied_scores"] = np.array(sorted(tied), dtype=np.float32)
            print("  postprocess thresh %.2f: %d detections" % (thr, dets.shape[0]))
    store["meta"] = np.array([int(use_ba), int(training), B, way, shot, H, W, 11, 1996, nms_seed,
                              int(attention_type == "product")])
    save("e2e_" + tag, **store)


def e2e_frcnn(tag, training, B, H, W, nms_seed=7):
    """the sibling `frcnn` model (utils.py:109-110) on the same trunk / RPN / RoI ops"""
    m = R.build_frcnn()
    sd = S.fill_state_dict(m.state_dict(), seed=13, profile="test")
    m.load_state_dict(sd)
    im_data, im_info, gt, nb, _ = S.episode_inputs(B, 1, 1, H, W, seed=1996)
    m.train() if training else m.eval()
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_ref = m(im_data, im_info, gt, nb)
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_or = O.frcnn_forward(sd, im_data, im_info, gt, nb, training, nms_inclusive=True)
    names = ["rois", "cls_prob", "bbox_pred", "rpn_loss_cls", "rpn_loss_bbox", "RCNN_loss_cls", "RCNN_loss_bbox",
             "rois_label"]
    store = {}
    for n, a, b in zip(names, out_ref, out_or):
        if a is None or (not torch.is_tensor(a) and a == 0):
            assert b is None or (not torch.is_tensor(b) and b == 0), n
            continue
        a, b = a.detach(), b.detach()
        diff = (a.float() - b.float()).abs().max().item() if a.numel() else 0.0
        print("  %-16s ref-vs-oracle max|d| = %.3e" % (n, diff))
        assert diff <= {"rois": 1e-3, "rois_label": 0.0}.get(n, 2e-5), (n, diff)
        store[n] = a.numpy()
    store["meta"] = np.array([int(training), B, H, W, 13, 1996, nms_seed])
    save("e2e_frcnn_" + tag, **store)


def e2e_meta(tag, training, B, way, shot, H, W, nms_seed=7):
    """the sibling `meta` model (utils.py:113-114): PRN class-attentive vectors x RoI features"""
    m = R.build_meta(way, shot)
    sd = S.fill_state_dict(m.state_dict(), seed=17, profile="test")
    m.load_state_dict(sd)
    im_data, im_info, gt, nb, sup = S.episode_inputs(B, way if training else 1, shot, H, W, seed=1996)
    all_gt = gt.clone()  # the synthetic episodes carry one class: all-class boxes = the episode's boxes
    m.train() if training else m.eval()
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_ref = m(im_data, im_info, gt, nb, sup, all_gt)
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_or = O.meta_forward(sd, im_data, im_info, gt, nb, sup, all_gt, training, way, shot, nms_inclusive=True)
    names = ["rois", "cls_prob", "bbox_pred", "rpn_loss_cls", "rpn_loss_bbox", "RCNN_loss_cls", "RCNN_loss_bbox",
             "rois_label"]
    store = {}
    for n, a, b in zip(names, out_ref, out_or):
        if a is None or (not torch.is_tensor(a) and a == 0):
            assert b is None or (not torch.is_tensor(b) and b == 0), n
            continue
        a, b = a.detach(), b.detach()
        diff = (a.float() - b.float()).abs().max().item() if a.numel() else 0.0
        print("  %-16s ref-vs-oracle max|d| = %.3e" % (n, diff))
        assert diff <= {"rois": 1e-3, "rois_label": 0.0}.get(n, 2e-5), (n, diff)
        store[n] = a.numpy()
    store["meta"] = np.array([int(training), B, way, shot, H, W, 17, 1996, nms_seed])
    save("e2e_meta_" + tag, **store)


def e2e_fsod(tag, training, B, way, shot, H, W, nms_seed=7):
    """the sibling `fsod` model (utils.py:111-112)"""
    m = R.build_fsod(way, shot)
    sd = S.tame_fsod_weights(S.fill_state_dict(m.state_dict(), seed=19, profile="test"))
    m.load_state_dict(sd)
    im_data, im_info, gt, nb, sup = S.episode_inputs(B, way if training else 1, shot, H, W, seed=1996)
    m.train() if training else m.eval()
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_ref = m(im_data, im_info, gt, nb, sup)
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_or = O.fsod_forward(sd, im_data, im_info, gt, nb, sup, training, way, shot, nms_inclusive=True)
    names = ["rois", "cls_prob", "bbox_pred", "rpn_loss_cls", "rpn_loss_bbox", "RCNN_loss_cls", "RCNN_loss_bbox",
             "rois_label"]
    store = {}
    for n, a, b in zip(names, out_ref, out_or):
        if a is None or (not torch.is_tensor(a) and a == 0):
            assert b is None or (not torch.is_tensor(b) and b == 0), n
            continue
        a, b = a.detach(), b.detach()
        diff = (a.float() - b.float()).abs().max().item() if a.numel() else 0.0
        print("  %-16s ref-vs-oracle max|d| = %.3e" % (n, diff))
        assert diff <= {"rois": 1e-3, "rois_label": 0.0}.get(n, 2e-5), (n, diff)
        store[n] = a.numpy()
    store["meta"] = np.array([int(training), B, way, shot, H, W, 19, 1996, nms_seed])
    save("e2e_fsod_" + tag, **store)


def e2e_fgn(tag, training, B, way, shot, H, W, nms_seed=7):
    """the sibling `fgn` model (utils.py:115-116); its head's BatchNorm layers run on batch statistics in train mode"""
    m = R.build_fgn(way, shot)
    sd = S.tame_fgn_weights(S.fill_state_dict(m.state_dict(), seed=23, profile="test"))
    m.load_state_dict(sd)
    im_data, im_info, gt, nb, sup = S.episode_inputs(B, way if training else 1, shot, H, W, seed=1996)
    m.train() if training else m.eval()
    np.random.seed(nms_seed)
    with torch.no_grad():
        out_ref = m(im_data, im_info, gt, nb, sup)
    np.random.seed(nms_seed)
    bn_state = {}
    with torch.no_grad():
        out_or = O.fgn_forward(sd, im_data, im_info, gt, nb, sup, training, way, shot, nms_inclusive=True,
                               bn_state=bn_state)
    names = ["rois", "cls_prob", "bbox_pred", "rpn_loss_cls", "rpn_loss_bbox", "RCNN_loss_cls", "RCNN_loss_bbox",
             "rois_label"]
    store = {}
    for n, a, b in zip(names, out_ref, out_or):
        if a is None or (not torch.is_tensor(a) and a == 0):
            assert b is None or (not torch.is_tensor(b) and b == 0), n
            continue
        a, b = a.detach(), b.detach()
        diff = (a.float() - b.float()).abs().max().item() if a.numel() else 0.0
        print("  %-16s ref-vs-oracle max|d| = %.3e" % (n, diff))
        assert diff <= {"rois": 1e-3, "rois_label": 0.0}.get(n, 2e-5), (n, diff)
        store[n] = a.numpy()
    after = m.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var"):
        d = (after[k] - bn_state[k]).abs().max().item()
        print("  %-16s ref-vs-oracle max|d| = %.3e" % (k, d))
        assert d <= 1e-5 * max(1.0, after[k].abs().max().item()), (k, d)
        store[k] = after[k].numpy()
    store["meta"] = np.array([int(training), B, way, shot, H, W, 23, 1996, nms_seed])
    save("e2e_fgn_" + tag, **store)


if __name__ == "__main__":
    assert R.available(), "reference not present: golden vectors can only be (re)generated in the build container"
    ref = R.load()
    op_goldens(ref)
    with torch.no_grad():
        print("eval 192x256 BA off"); e2e("eval_small_cisa", False, False, 1, 1, 3, 192, 256)
        print("eval 192x256 BA on"); e2e("eval_small_ba", True, False, 1, 1, 3, 192, 256)
        print("train 192x256 B=2 BA on"); e2e("train_small_ba", True, True, 2, 2, 3, 192, 256)
        # attention_type='product' (dana.py:74-77,155-156,285-286: valid reference code that utils.get_model never selects)
        print("eval 192x256 product"); e2e("eval_small_product", True, False, 1, 1, 3, 192, 256, slim=True, attention_type="product")
        print("train 192x256 B=2 product"); e2e("train_small_product", True, True, 2, 2, 3, 192, 256, slim=True, attention_type="product")
        print("fgn eval 192x256"); e2e_fgn("eval_small", False, 1, 1, 3, 192, 256)
        print("fgn train 192x256 B=2"); e2e_fgn("train_small", True, 2, 2, 3, 192, 256)
        print("fsod eval 192x256"); e2e_fsod("eval_small", False, 1, 1, 3, 192, 256)
        print("fsod train 192x256 B=2"); e2e_fsod("train_small", True, 2, 2, 3, 192, 256)
        print("meta eval 192x256"); e2e_meta("eval_small", False, 1, 1, 3, 192, 256)
        print("meta train 192x256 B=2"); e2e_meta("train_small", True, 2, 2, 3, 192, 256)
        print("frcnn eval 192x256"); e2e_frcnn("eval_small", False, 1, 192, 256)
        print("frcnn train 192x256 B=2"); e2e_frcnn("train_small", True, 2, 192, 256)
        if "--full" in sys.argv:
            print("eval 600x1000 BA on"); e2e("eval_full_ba", True, False, 1, 1, 3, 600, 1000)
        if "--full" in sys.argv or "--train-full" in sys.argv:
            # BASELINE configs[2] / configs[1] themselves: train mode, B = 4, 600x1000, way 2, shot 3 (dana.py:87-220)
            print("train 600x1000 B=4 BA on"); e2e("train_full_ba", True, True, 4, 2, 3, 600, 1000, slim=True)
            print("train 600x1000 B=4 BA off"); e2e("train_full_cisa", False, True, 4, 2, 3, 600, 1000, slim=True)
