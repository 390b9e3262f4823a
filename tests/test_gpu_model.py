"""End-to-end parity of the HIP DAnARCNN forward (module API -> C ABI -> gfx950 kernels) against the
reference's golden vectors (tests/golden/e2e_*.npz, produced by running the reference itself) and
against the oracle on fresh inputs.

Tolerances (north_star: boxes within 1e-3 IoU, scores within 1e-4 of the reference):
  continuous maps (base_feat, attended feature, RPN heads)  rel 2e-4 of the map's max |value|
  rois (matched by position; scores are well separated)      IoU >= 1 - 1e-3 for >= 99% of rois
  cls_prob on matched rois                                   |d| <= 1e-4 ; bbox_pred |d| <= 1e-4
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, "e2e_%s.npz" % tag))


def _build(meta, dev):
    import dana_amd
    from dana_amd import synthetic as S
    use_ba, training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in meta[:10]]
    if len(meta) > 10 and int(meta[10]):  # attention_type='product' (dana.py:74-77): not a get_model() choice, as in the reference
        from dana_amd.dana import DAnARCNN
        m = DAnARCNN(["fg", "bg"], "product", 256, 256, pretrained=False, semantic_enhance=bool(use_ba), num_way=way,
                     num_shot=shot)
        m.create_architecture()
        tame = S.tame_product_weights
    else:
        m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=bool(use_ba), way=way, shot=shot,
                               classes=["fg", "bg"])
        tame = lambda sd_: sd_  # noqa: E731
    sd = tame(S.fill_state_dict(m.state_dict(), seed=wseed, profile="test"))
    m.load_state_dict(sd)
    m.to(dev)
    m.nms_inclusive = True  # the golden vectors come from the reference's CPU path (nms_cpu.cpp:60: >=)
    m.train() if training else m.eval()
    inputs = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    return m, sd, [t.to(dev) for t in inputs], inputs, (use_ba, training, B, way, shot, nseed)


def _iou(a, b):
    x1, y1 = np.maximum(a[:, 0], b[:, 0]), np.maximum(a[:, 1], b[:, 1])
    x2, y2 = np.minimum(a[:, 2], b[:, 2]), np.minimum(a[:, 3], b[:, 3])
    inter = np.clip(x2 - x1 + 1, 0, None) * np.clip(y2 - y1 + 1, 0, None)
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (aa + ab - inter)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.fixture(params=[1, 0], ids=["bf16x6", "f32mfma"])
def mfma_mode(request):
    """the reference-golden end-to-end tests run with the contractions on the bf16 matrix cores (exact 3-way split, the
    default) and on the f32 MFMA: both kernels are pinned against the reference's own outputs"""
    import dana_amd
    prev = dana_amd.ops.set_mfma_mode(request.param)
    yield request.param
    dana_amd.ops.set_mfma_mode(prev)


@pytest.mark.parametrize("tag", ["eval_small_cisa", "eval_small_ba", "eval_full_ba", "eval_small_product"])
def test_eval_forward_matches_reference_golden(golden_dir, dev, tag, mfma_mode):
    import dana_amd
    ops = dana_amd.ops
    g = _load(golden_dir, tag)
    m, sd, din, _, _ = _build(g["meta"], dev)
    m._capture = {}
    with torch.no_grad():
        rois, cls_prob, bbox_pred, l1, l2, l3, l4, lab = m(*din)
    assert (l1, l2, l3, l4, lab) == (0, 0, 0, 0, None)  # dana.py:173-178,217-218
    if "base_feat_s" in g:  # (the slim fixtures hold the eight outputs only)
        corr, B, fh, fw = m._capture["corr"]
        corr_nchw = ops.nhwc_to_nchw(corr, B, 2048, fh, fw).cpu().numpy()
        assert _rel(corr_nchw[:, :1024][:, ::16], g["base_feat_s"]) < 2e-4
        assert _rel(corr_nchw[:, 1024:][:, ::16], g["dense_s"]) < 2e-4
        heads = m._capture["rpn_heads"].view(B, fh, fw, 72).permute(0, 3, 1, 2).cpu().numpy()
        assert _rel(heads[:, :24], g["rpn_cls_score"]) < 2e-4
        assert _rel(heads[:, 24:], g["rpn_bbox_pred"]) < 2e-4
    r, rg = rois.cpu().numpy().reshape(-1, 5), g["rois"].reshape(-1, 5)
    assert r.shape == rg.shape and np.array_equal(r[:, 0], rg[:, 0])
    iou = _iou(r[:, 1:], rg[:, 1:])
    matched = iou >= 1 - 1e-3
    assert matched.mean() >= 0.99, "only %.1f%% of rois match the reference by position" % (100 * matched.mean())
    assert np.abs(cls_prob.cpu().numpy() - g["cls_prob"])[matched].max() <= 1e-4
    assert np.abs(bbox_pred.cpu().numpy() - g["bbox_pred"])[matched].max() <= 1e-4


def _assert_train_outputs(out, ref, matched=None):
    """cls_prob / bbox_pred / labels / the two RCNN losses of a train-mode forward against the reference (or oracle)
    8-tuple computed on the SAME sampled rois: unconditional, north_star's 1e-4 bar"""
    rois, cls_prob, bbox_pred, l1, l2, l3, l4, lab = out
    g_rois, g_prob, g_pred, g1, g2, g3, g4, g_lab = ref
    assert np.array_equal(np.asarray(lab.cpu()), np.asarray(g_lab))
    assert np.abs(np.asarray(cls_prob.cpu()) - np.asarray(g_prob)).max() <= 1e-4
    assert np.abs(np.asarray(bbox_pred.cpu()) - np.asarray(g_pred)).max() <= 1e-4
    for name, a, b in (("rpn_loss_cls", l1, g1), ("rpn_loss_bbox", l2, g2), ("RCNN_loss_cls", l3, g3),
                       ("RCNN_loss_bbox", l4, g4)):
        assert abs(float(a) - float(b)) <= 1e-4 * max(1.0, abs(float(b))), (name, float(a), float(b))


@pytest.mark.parametrize("tag", ["train_small_ba", "train_full_ba", "train_full_cisa", "train_small_product"])
def test_train_forward_matches_reference_golden(golden_dir, dev, mfma_mode, tag):
    """train-mode forward (dana.py:87-220) against outputs the REFERENCE produced: 192x256 B=2, and BASELINE configs[2] /
    configs[1] themselves (600x1000, B=4, way 2, shot 3, BA on / off; make_golden.py --train-full)"""
    from oracle import model_ref as O
    g = _load(golden_dir, tag)
    m, sd, din, inputs, (use_ba, training, B, way, shot, nseed) = _build(g["meta"], dev)
    # (1) the whole path incl. this build's own target sampling under the reference's np.random stream
    np.random.seed(nseed)
    with torch.no_grad():
        rois, cls_prob, bbox_pred, l1, l2, l3, l4, lab = m(*din)
    r, rg = rois.cpu().numpy().reshape(-1, 5), g["rois"].reshape(-1, 5)
    assert r.shape == rg.shape
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    if "full" not in tag:
        assert matched.mean() >= 0.97, "sampled rois diverge from the reference: %.1f%% match" % (100 * matched.mean())
    else:
        # ~21 500 sorted scores and 12 000 NMS candidates per image: one near-tie flips a discrete decision, the candidate
        # list shifts and the same np.random stream draws another subset (the reference and the CPU oracle differ from
        # EACH OTHER that way at this size, make_golden.py) -- position-wise identity is asserted where it is defined:
        # (2) below on the reference's sampled batch, _train_vs_oracle (e) on a common proposal list
        print("%s: %.1f%% of the sampled rois equal the reference's position-wise" % (tag, 100 * matched.mean()))
    for name, v in (("rpn_loss_cls", l1), ("rpn_loss_bbox", l2)):  # independent of which rois were sampled
        assert abs(float(v) - float(g[name])) <= 1e-4 * max(1.0, abs(float(g[name]))), name
    # (2) stage-wise, UNCONDITIONAL: the reference's own sampled batch (golden rois + labels; the regression targets
    # are a deterministic function of them) goes into the RoI stages -> every downstream number must agree
    n = B * rois.size(1)
    m._inject_sampled = O.sampled_targets(torch.from_numpy(g["rois"]),
                                          torch.from_numpy(g["rois_label"][:n]).float().view(B, -1), inputs[2])
    np.random.seed(nseed)
    with torch.no_grad():
        out = m(*din)
    m._inject_sampled = None
    assert np.array_equal(out[0].cpu().numpy(), g["rois"])
    _assert_train_outputs(out, [g[k] for k in ("rois", "cls_prob", "bbox_pred", "rpn_loss_cls", "rpn_loss_bbox",
                                               "RCNN_loss_cls", "RCNN_loss_bbox", "rois_label")])


def test_eval_forward_vs_oracle_fresh_inputs_cuda_nms_rule(dev):
    """fresh seed, B=2, the shipped `>` NMS rule; oracle switched to the same rule"""
    import dana_amd
    from dana_amd import synthetic as S
    from oracle import model_ref as O
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=2, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=3, profile="test")
    m.load_state_dict(sd)
    m.to(dev).eval()
    inputs = S.episode_inputs(2, 1, 2, 160, 224, seed=77)
    with torch.no_grad():
        out = m(*[t.to(dev) for t in inputs])
        ref = O.forward(sd, *inputs, False, 1, 2, True, nms_inclusive=False)
    r, rg = out[0].cpu().numpy().reshape(-1, 5), ref[0].numpy().reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.99
    assert np.abs(out[1].cpu().numpy() - ref[1].numpy())[matched].max() <= 1e-4
    assert np.abs(out[2].cpu().numpy() - ref[2].numpy())[matched].max() <= 1e-4


def test_res101_trunk_opt_in_vs_oracle(dev):
    """BASELINE configs[3]'s well-defined half: the resnet101 trunk of resnet.py:199 ([3, 4, 23, 3] blocks) behind
    DAnARCNN.trunk_layers. The reference never builds it (dana.py:337: resnet50() whatever num_layers says), so there is no
    reference golden: the oracle (pinned to the reference on the res50 goldens, its trunk loop driven by the state dict's
    own block count) is the checker -- eval forward, and one training iteration's losses under the same np.random stream."""
    from dana_amd import synthetic as S
    from dana_amd.dana import DAnARCNN
    from dana_amd.trainer import Trainer
    from oracle import model_ref as O
    torch.set_num_threads(min(64, max(torch.get_num_threads(), os.cpu_count() or 1)))

    def build(way, shot):
        m = DAnARCNN(["fg", "bg"], "concat", 256, 256, pretrained=False, semantic_enhance=True, num_way=way, num_shot=shot)
        m.trunk_layers = (3, 4, 23, 3)
        m.create_architecture()
        assert len(m.RCNN_base[6]) == 23 and len(m.state_dict()) == 346 + 17 * 18  # 17 more bottlenecks x 18 tensors
        sd = S.tame_res101_weights(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
        m.load_state_dict(sd)
        return m.to(dev), sd

    m, sd = build(1, 2)
    m.eval()
    inputs = S.episode_inputs(1, 1, 2, 160, 224, seed=77)
    with torch.no_grad():
        out = m(*[t.to(dev) for t in inputs])
        ref = O.forward(sd, *inputs, False, 1, 2, True, nms_inclusive=False)
    r, rg = out[0].cpu().numpy().reshape(-1, 5), ref[0].numpy().reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.99
    assert np.abs(out[1].cpu().numpy() - ref[1].numpy())[matched].max() <= 1e-4
    assert np.abs(out[2].cpu().numpy() - ref[2].numpy())[matched].max() <= 1e-4
    # train mode: forward losses vs the oracle, then one Trainer iteration moves all 23 layer-3 blocks
    m, sd = build(2, 2)
    m.train()
    tin = S.episode_inputs(1, 2, 2, 160, 224, seed=9)
    np.random.seed(3)
    with torch.no_grad():
        ref_t = O.forward(sd, *tin, True, 2, 2, True, nms_inclusive=False)
    tr = Trainer(m, 1e-3)
    w0 = m.RCNN_base[6][22].conv2.weight.detach().clone()
    np.random.seed(3)
    out_t = tr.step(*[t.to(dev) for t in tin])
    torch.cuda.synchronize()
    for a_, b_ in zip(out_t[3:5], ref_t[3:5]):
        assert abs(float(a_.detach()) - float(b_)) <= 1e-4 * max(1.0, abs(float(b_)))
    assert not torch.equal(w0, m.RCNN_base[6][22].conv2.weight.detach())
    assert all(torch.isfinite(p_).all() for p_ in m.parameters())


def test_product_attention_type_trains_only_forward(dev):
    """attention_type='product': forward in both modes on the HIP kernels (goldens above); a forward that would save for
    the backward says so instead of producing gradients of another model"""
    from dana_amd import synthetic as S
    from dana_amd.dana import DAnARCNN
    m = DAnARCNN(["fg", "bg"], "product", 256, 256, pretrained=False, semantic_enhance=True, num_way=2, num_shot=2)
    m.create_architecture()
    m.to(dev).train()
    inputs = [t.to(dev) for t in S.episode_inputs(1, 2, 2, 128, 160, seed=3)]
    np.random.seed(0)
    with pytest.raises(NotImplementedError, match="backward is not implemented"):
        m(*inputs)  # grad mode on: the loss bridge would need the backward
    with pytest.raises(ValueError):
        DAnARCNN(["fg", "bg"], "sum")


def test_model_rejects_cpu_and_bad_supports(dev):
    import dana_amd
    from dana_amd import synthetic as S
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=1, shot=1, classes=["fg", "bg"])
    m.eval()
    inputs = S.episode_inputs(1, 1, 1, 96, 128, seed=1)
    with pytest.raises(RuntimeError):
        m(*inputs)  # CPU: no fallback
    m.to(dev)
    bad = S.episode_inputs(1, 1, 1, 96, 128, seed=1, support_size=224)
    with pytest.raises(RuntimeError, match="320x320"):
        m(*[t.to(dev) for t in bad])


def _untied(d, tied):
    keep = [i for i in range(d.shape[0]) if float(d[i, 4]) not in tied]
    return d[keep]


@pytest.mark.parametrize("tag", ["eval_small_cisa", "eval_full_ba"])
def test_inference_postprocess_matches_oracle_and_reference_golden(golden_dir, dev, tag):
    """SURVEY.md 8f row N1 (inference.py:106-140, utils.py:312-317): decode + threshold + sort + NMS(0.3) on device,
    fed with the REFERENCE's forward outputs (golden), vs the reference's detections and the oracle."""
    import dana_amd
    from dana_amd import synthetic as S
    from oracle import model_ref as O
    g = _load(golden_dir, tag)
    use_ba, training, B, way, shot, H, W = [int(v) for v in g["meta"][:7]]
    im_info = torch.tensor([[float(H), float(W), 1.0]])
    rois, prob, pred = torch.from_numpy(g["rois"]), torch.from_numpy(g["cls_prob"]), torch.from_numpy(g["bbox_pred"])
    tied = set(g["dets_tied_scores"].tolist())
    for thr, key in ((0.05, "dets_t05"), (0.62, "dets_t62")):
        got = dana_amd.postprocess.detections(rois.to(dev), prob.to(dev), pred.to(dev), im_info.to(dev), thresh=thr,
                                              nms_inclusive=True).cpu().numpy()
        ref = O.postprocess(rois, prob, pred, im_info, thresh=thr, nms_inclusive=True).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-3   # same stable tie order -> same rows
        a, b = _untied(got, tied), _untied(g[key], tied)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-3
    # threshold above every score -> empty result
    none = dana_amd.postprocess.detections(rois.to(dev), prob.to(dev), pred.to(dev), im_info.to(dev), thresh=2.0)
    assert none.shape == (0, 5)


@pytest.mark.parametrize("B,H,W,shot,ba", [(3, 157, 203, 1, True), (2, 130, 321, 2, False)])
def test_eval_forward_odd_sizes_vs_oracle(dev, B, H, W, shot, ba):
    """odd image sizes (ceil-mode maxpool edge, clipped Winograd tiles, partial igemm tiles), B > 1"""
    import dana_amd
    from dana_amd import synthetic as S
    from oracle import model_ref as O
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=ba, way=2, shot=shot, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=21, profile="test")
    m.load_state_dict(sd)
    m.to(dev).eval()
    inputs = S.episode_inputs(B, 1, shot, H, W, seed=5)
    with torch.no_grad():
        out = m(*[t.to(dev) for t in inputs])
        ref = O.forward(sd, *inputs, False, 1, shot, ba, nms_inclusive=False)
    r, rg = out[0].cpu().numpy().reshape(-1, 5), ref[0].numpy().reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.98
    assert np.abs(out[1].cpu().numpy() - ref[1].numpy())[matched].max() <= 1e-4
    assert np.abs(out[2].cpu().numpy() - ref[2].numpy())[matched].max() <= 1e-4


def test_train_forward_and_step_with_device_rng(dev):
    """opt-in sync-free sampling (DAnARCNN.device_rng): deterministic in (rng_seed, call counter), valid batches,
    and the training iteration runs on it"""
    import dana_amd
    from dana_amd import synthetic as S
    from dana_amd.trainer import Trainer
    B, way, shot = 2, 2, 2
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=way, shot=shot, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=3, profile="test"))
    m.to(dev).train()
    m.device_rng = True
    inputs = [t.to(dev) for t in S.episode_inputs(B, way, shot, 192, 256, seed=5)]
    outs = []
    for _ in range(2):
        m._rng_calls = 0
        with torch.no_grad():
            outs.append(m(*inputs))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    with torch.no_grad():
        o3 = m(*inputs)  # next call counter: a different draw
    assert not torch.equal(o3[0], outs[0][0])
    rois, cls_prob, bbox_pred, l1, l2, l3, l4, lab = outs[0]
    assert rois.shape == (B, 128, 5) and lab.shape == (2 * B * 128,)
    assert 0 < int(lab[:B * 128].sum()) <= B * 32 and int(lab[B * 128:].sum()) == 0
    assert all(bool(torch.isfinite(x)) for x in (l1, l2, l3, l4))
    tr = Trainer(m, 1e-3)
    before = m.RCNN_rpn.RPN_Conv.weight.detach().clone()
    tr.step(*inputs)
    torch.cuda.synchronize()
    assert not torch.equal(before, m.RCNN_rpn.RPN_Conv.weight.detach())


def _train_vs_oracle(dev, B, way, shot, H, W, ba, wseed=11, iseed=1996, nseed=5, min_match=0.97):
    """train-mode forward vs the oracle under the same np.random stream: (1) own sampling -> rois >= min_match within
    1e-3 IoU + the RPN losses; (2) the oracle's sampled batch injected -> labels / cls_prob / bbox_pred / all four
    losses asserted unconditionally"""
    import dana_amd
    from dana_amd import synthetic as S
    from oracle import model_ref as O
    torch.set_num_threads(min(64, max(torch.get_num_threads(), os.cpu_count() or 1)))
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=ba, way=way, shot=shot, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=wseed, profile="test")
    m.load_state_dict(sd)
    m.to(dev).train()
    inputs = S.episode_inputs(B, way, shot, H, W, seed=iseed)
    din = [t.to(dev) for t in inputs]
    m._capture = {}
    np.random.seed(nseed)
    with torch.no_grad():
        out = m(*din)
    ours_prop = m._capture["rpn_rois"].cpu().numpy()
    m._capture = None
    np.random.seed(nseed)
    inter = {}
    with torch.no_grad():
        ref = O.forward(sd, *inputs, True, way, shot, ba, nms_inclusive=False, inter=inter)
    # (a) the proposal layer's output, matched by IoU (not by index: one flipped NMS decision shifts every later slot)
    ref_prop = inter["rpn_rois"].numpy()
    flip_free, pos_match = [], []
    for i in range(B):
        a, b = ours_prop[i, :, 1:], ref_prop[i, :, 1:]
        # EVERY proposal against every oracle proposal: one vectorised IoU matrix per image
        x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
        x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
        inter_ = np.clip(x2 - x1 + 1, 0, None) * np.clip(y2 - y1 + 1, 0, None)
        aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
        ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        hit = (inter_ / (aa[:, None] + ab[None, :] - inter_) >= 1 - 1e-3).any(1)
        assert hit.mean() >= 0.99, "image %d: only %.1f%% of the proposals have a counterpart in the oracle's" % (i, 100 * hit.mean())
        pos_match.append(float((_iou(a, b) >= 1 - 1e-3).mean()))
        flip_free.append(bool(pos_match[-1] >= 0.999))
    # (b) the RPN losses depend on the anchor sampling only (exact inputs): always comparable
    assert abs(float(out[3]) - float(ref[3])) <= 1e-4 * max(1.0, abs(float(ref[3])))  # rpn_loss_cls
    assert abs(float(out[4]) - float(ref[4])) <= 1e-4 * max(1.0, abs(float(ref[4])))  # rpn_loss_bbox
    # (c) this build's own sampling at this size, asserted by what proposal_target_layer_cascade.py:120-213 guarantees for
    # ANY proposal list -- so it holds (and can fail) whether or not a near-tie flipped a discrete decision upstream:
    # every sampled roi is a row of THIS run's proposal list or a gt box; its label is 1 exactly when its best IoU (+1
    # convention) with a gt box reaches FG_THRESH = 0.5; the fg count is min(32, fg candidates) and the batch is full.
    # (Position-wise identity with the oracle under the same np.random stream needs the same candidate list: that is
    # step (e), on the oracle's list; the position-wise fractions against the oracle's own run are printed below.)
    R = out[0].size(1)
    r, rg = out[0].cpu().numpy(), ref[0].numpy()
    lab = out[7].cpu().numpy()[:B * R].reshape(B, R)
    gt_np = inputs[2].numpy()

    def _iou_matrix(a_, b_):
        x1, y1 = np.maximum(a_[:, None, 0], b_[None, :, 0]), np.maximum(a_[:, None, 1], b_[None, :, 1])
        x2, y2 = np.minimum(a_[:, None, 2], b_[None, :, 2]), np.minimum(a_[:, None, 3], b_[None, :, 3])
        it = np.clip(x2 - x1 + 1, 0, None) * np.clip(y2 - y1 + 1, 0, None)
        aa_ = (a_[:, 2] - a_[:, 0] + 1) * (a_[:, 3] - a_[:, 1] + 1)
        ab_ = (b_[:, 2] - b_[:, 0] + 1) * (b_[:, 3] - b_[:, 1] + 1)
        return it / (aa_[:, None] + ab_[None, :] - it)

    for i in range(B):
        n_gt = int(inputs[3][i])
        cand = np.concatenate([ours_prop[i, :, 1:], gt_np[i, :n_gt, :4]], 0)
        cand_set = {row.tobytes() for row in np.ascontiguousarray(cand, dtype=np.float32)}
        assert all(np.ascontiguousarray(row, dtype=np.float32).tobytes() in cand_set for row in r[i, :, 1:]), \
            "image %d: a sampled roi is neither one of this run's proposals nor a gt box" % i
        assert np.all(r[i, :, 0] == i)
        best = _iou_matrix(r[i, :, 1:].astype(np.float64), gt_np[i, :n_gt, :4].astype(np.float64)).max(1)
        clear = np.abs(best - 0.5) > 1e-5  # (an IoU within rounding of the threshold may fall either way)
        assert np.array_equal(lab[i][clear] == 1, best[clear] >= 0.5), "image %d: labels disagree with the IoU rule" % i
        cand_best = _iou_matrix(cand.astype(np.float64), gt_np[i, :n_gt, :4].astype(np.float64)).max(1)
        n_fg_c = int((cand_best >= 0.5 + 1e-5).sum())
        n_fg_max = int((cand_best >= 0.5 - 1e-5).sum())
        n_fg = int((lab[i] == 1).sum())
        assert min(32, n_fg_c) <= n_fg <= min(32, n_fg_max), (i, n_fg, n_fg_c)
    per_image = [float((_iou(r[i, :, 1:], rg[i, :, 1:]) >= 1 - 1e-3).mean()) for i in range(B)]
    good = [v >= min_match for v in per_image]
    for i in range(B):  # same proposal list as the oracle's -> the same picks, position by position
        assert good[i] or not flip_free[i], "image %d: same proposals, but only %.1f%% of the sampled rois match" % (
            i, 100 * per_image[i])
    matched = np.array(per_image)
    # make a drift visible: how many images had a flip-free proposal list, and how many met the sampled-roi bar
    # (pytest -rP / -s shows it; the caller asserts its own floor on the flip-free count)
    # (at 600x1000 one near-tie among 12 000 sorted scores / 2 000 NMS survivors per image is the rule, not the exception:
    # all four images of the seeded full-size runs have a flip somewhere -- the position-wise fractions say how early)
    print("_train_vs_oracle B=%d %dx%d shot=%d ba=%d: flip-free proposal lists %d/%d (position-wise match %s), sampled rois "
          "matched on %d/%d images %s" % (B, H, W, shot, int(ba), sum(flip_free), B, ["%.3f" % v for v in pos_match],
                                          sum(good), B, ["%.3f" % v for v in per_image]))
    # (d) stage-wise and unconditional: the oracle's sampled batch goes into the RoI stages
    m._inject_sampled = inter["sampled"]
    np.random.seed(nseed)
    with torch.no_grad():
        out2 = m(*din)
    m._inject_sampled = None
    assert np.array_equal(out2[0].cpu().numpy(), ref[0].numpy())
    _assert_train_outputs(out2, [t.numpy() if torch.is_tensor(t) else t for t in ref])
    # (e) this build's OWN proposal-target sampling (targets.hip / sampling on 2 000 + n_gt candidates per image) at this size,
    # position by position: the oracle's proposal LIST goes in (not its sampled batch), the same np.random stream draws ->
    # the same picks in the same order, the same labels, and everything downstream within the train-output bars
    m._inject_rpn_rois = inter["rpn_rois"]
    np.random.seed(nseed)
    with torch.no_grad():
        out3 = m(*din)
    m._inject_rpn_rois = None
    assert np.array_equal(out3[0].cpu().numpy(), ref[0].numpy()), "own sampling on the oracle's proposal list picks other rois"
    _assert_train_outputs(out3, [t.numpy() if torch.is_tensor(t) else t for t in ref])
    return float(matched.mean()), flip_free


def test_train_forward_full_size_vs_oracle(dev):
    """BASELINE.json configs[1] at its full batch: 600x1000, way 2, shot 3, bs 4, CISA only, train mode"""
    _train_vs_oracle(dev, 4, 2, 3, 600, 1000, False)


def test_train_forward_full_size_ba_bs4_vs_oracle(dev):
    """BASELINE.json configs[2] at its full batch: 600x1000, way 2, shot 3, bs 4, BA + CISA (the bench headline)"""
    _train_vs_oracle(dev, 4, 2, 3, 600, 1000, True, nseed=9)


def test_dana_roi_pool_mode_vs_oracle(dev):
    """cfg.POOLING_MODE = 'pool' inside DAnA (dana.py:183-184; a resumed checkpoint may set it, train.py:100-101): the
    RoIPool operator of the `_C` boundary feeds layer4 and the RoI-level attention; vs the oracle in the same mode"""
    import dana_amd
    from dana_amd import synthetic as S
    from dana_amd.config import cfg
    from oracle import model_ref as O
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=1, shot=2, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=13, profile="test")
    m.load_state_dict(sd)
    m.to(dev).eval()
    inputs = S.episode_inputs(1, 1, 2, 160, 224, seed=3)
    old = cfg.POOLING_MODE
    cfg.POOLING_MODE = "pool"
    try:
        with torch.no_grad():
            out = m(*[t.to(dev) for t in inputs])
            ref = O.forward(sd, *inputs, False, 1, 2, True, nms_inclusive=False, pooling="pool")
            cfg.POOLING_MODE = "align"
            out_align = m(*[t.to(dev) for t in inputs])
    finally:
        cfg.POOLING_MODE = old
    r, rg = out[0].cpu().numpy().reshape(-1, 5), ref[0].numpy().reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.99
    assert np.abs(out[1].cpu().numpy() - ref[1].numpy()).reshape(-1, 2)[matched].max() <= 1e-4
    assert np.abs(out[2].cpu().numpy() - ref[2].numpy()).reshape(-1, 4)[matched].max() <= 1e-4
    assert not torch.equal(out[1], out_align[1])  # (the two pooling modes really are different operators)


def test_config4_stress_train_forward_vs_oracle(dev):
    """BASELINE.json configs[4] per-GPU shape: 800x1333 queries (50x84 map, 50 400 anchors / image,
    proposal_layer.py:72-93), way 2, shot 10 (4 000 attention keys, dana.py:126-147), 2 episodes per GPU, train mode"""
    _train_vs_oracle(dev, 2, 2, 10, 800, 1333, True, nseed=13)


def test_config4_stress_eval_forward_vs_oracle(dev):
    """configs[4] geometry in eval mode (300 rois / image through layer4 and the RoI-level attention with 490 keys)"""
    import dana_amd
    from dana_amd import synthetic as S
    from oracle import model_ref as O
    torch.set_num_threads(min(64, max(torch.get_num_threads(), os.cpu_count() or 1)))
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=10, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=11, profile="test")
    m.load_state_dict(sd)
    m.to(dev).eval()
    inputs = S.episode_inputs(2, 1, 10, 800, 1333, seed=2024)
    with torch.no_grad():
        out = m(*[t.to(dev) for t in inputs])
        ref = O.forward(sd, *inputs, False, 1, 10, True, nms_inclusive=False)
    r, rg = out[0].cpu().numpy().reshape(-1, 5), ref[0].numpy().reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.99
    assert np.abs(out[1].cpu().numpy() - ref[1].numpy())[matched].max() <= 1e-4
    assert np.abs(out[2].cpu().numpy() - ref[2].numpy())[matched].max() <= 1e-4


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_frcnn_sibling_matches_reference_golden(golden_dir, dev, tag):
    """get_model('frcnn') (utils.py:109-110) on the same HIP operators vs the reference's own outputs"""
    import dana_amd
    from dana_amd import synthetic as S
    g = _load(golden_dir, "frcnn_" + tag)
    training, B, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("frcnn", pretrained=False, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=wseed, profile="test"))
    m.to(dev)
    m.nms_inclusive = True
    m.train() if training else m.eval()
    im_data, im_info, gt, nb, _ = S.episode_inputs(B, 1, 1, H, W, seed=iseed)
    np.random.seed(nseed)
    with torch.no_grad():
        out = m(im_data.to(dev), im_info.to(dev), gt.to(dev), nb.to(dev))
    r, rg = out[0].cpu().numpy().reshape(-1, 5), g["rois"].reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.97
    if matched.all():
        assert np.abs(out[1].cpu().numpy() - g["cls_prob"]).max() <= 1e-4
        assert np.abs(out[2].cpu().numpy() - g["bbox_pred"]).max() <= 1e-4
        if training:
            assert np.array_equal(out[7].cpu().numpy(), g["rois_label"])
            for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
                assert abs(float(out[i]) - float(g[name])) <= 1e-4 * max(1.0, abs(float(g[name]))), name
    else:
        assert not training  # eval rois are deterministic: a mismatch there is a real difference
        raise AssertionError("eval rois differ from the reference: %.1f%% match" % (100 * matched.mean()))


def test_frcnn_roi_pool_mode_vs_oracle(dev):
    """cfg.POOLING_MODE = 'pool' (faster_rcnn.py:72-73): the RoIPool kernel inside a model, vs the oracle"""
    import dana_amd
    from dana_amd import synthetic as S
    from dana_amd.config import cfg
    from oracle import model_ref as O
    m = dana_amd.get_model("frcnn", pretrained=False, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=13, profile="test")
    m.load_state_dict(sd)
    m.to(dev).eval()
    im_data, im_info, gt, nb, _ = S.episode_inputs(1, 1, 1, 160, 224, seed=3)
    old = cfg.POOLING_MODE
    cfg.POOLING_MODE = "pool"
    try:
        with torch.no_grad():
            out = m(im_data.to(dev), im_info.to(dev), gt.to(dev), nb.to(dev))
            ref = O.frcnn_forward(sd, im_data, im_info, gt, nb, False, nms_inclusive=False, pooling="pool")
    finally:
        cfg.POOLING_MODE = old
    r, rg = out[0].cpu().numpy().reshape(-1, 5), ref[0].numpy().reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.99
    assert np.abs(out[1].cpu().numpy() - ref[1].numpy()).reshape(-1, 2)[matched].max() <= 1e-4
    assert np.abs(out[2].cpu().numpy() - ref[2].numpy()).reshape(-1, 4)[matched].max() <= 1e-4


def test_generalised_support_size_is_opt_in_and_runs(dev):
    """224x224 supports (BASELINE.json's wording): the reference cannot run them (dana.py:105 hard-codes the 20x20 map);
    opt-in generalisation, no oracle -- only checked to be refused by default, to run, to be finite and trainable"""
    import dana_amd
    from dana_amd import synthetic as S
    from dana_amd.trainer import Trainer
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=2, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=3, profile="test"))
    m.to(dev).train()
    inputs = [t.to(dev) for t in S.episode_inputs(1, 2, 2, 160, 224, seed=5, support_size=224)]
    with pytest.raises(RuntimeError, match="320x320"):
        with torch.no_grad():
            m(*inputs)
    m.generalised_support = True
    np.random.seed(1)
    with torch.no_grad():
        out = m(*inputs)
    assert out[0].shape == (1, 128, 5) and all(bool(torch.isfinite(x)) for x in out[3:7])
    tr = Trainer(m, 1e-3)
    np.random.seed(1)
    tr.step(*inputs)
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in m.parameters())


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_meta_sibling_matches_reference_golden(golden_dir, dev, tag):
    """get_model('meta') (utils.py:113-114) on the same HIP operators vs the reference's own outputs"""
    import dana_amd
    from dana_amd import synthetic as S
    g = _load(golden_dir, "meta_" + tag)
    training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("meta", pretrained=False, way=way, shot=shot, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=wseed, profile="test"))
    m.to(dev)
    m.nms_inclusive = True
    m.train() if training else m.eval()
    inputs = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    np.random.seed(nseed)
    with torch.no_grad():
        out = m(*[t.to(dev) for t in inputs], inputs[2].clone().to(dev))
    r, rg = out[0].cpu().numpy().reshape(-1, 5), g["rois"].reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.97
    if not training:
        assert matched.all()
    if matched.all():
        assert np.abs(out[1].cpu().numpy() - g["cls_prob"]).max() <= 1e-4
        assert np.abs(out[2].cpu().numpy() - g["bbox_pred"]).max() <= 1e-4
        if training:
            assert np.array_equal(out[7].cpu().numpy(), g["rois_label"])
            for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
                assert abs(float(out[i]) - float(g[name])) <= 1e-4 * max(1.0, abs(float(g[name]))), name


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_fsod_sibling_matches_reference_golden(golden_dir, dev, tag):
    """get_model('fsod') (utils.py:111-112) on the same HIP operators vs the reference's own outputs"""
    import dana_amd
    from dana_amd import synthetic as S
    g = _load(golden_dir, "fsod_" + tag)
    training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("fsod", pretrained=False, way=way, shot=shot, classes=["fg", "bg"])
    m.load_state_dict(S.tame_fsod_weights(S.fill_state_dict(m.state_dict(), seed=wseed, profile="test")))
    m.to(dev)
    m.nms_inclusive = True
    m.train() if training else m.eval()
    inputs = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    np.random.seed(nseed)
    with torch.no_grad():
        out = m(*[t.to(dev) for t in inputs])
    r, rg = out[0].cpu().numpy().reshape(-1, 5), g["rois"].reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.97
    if not training:
        assert matched.all()
    if matched.all():
        assert np.abs(out[1].cpu().numpy() - g["cls_prob"]).max() <= 1e-4
        assert np.abs(out[2].cpu().numpy() - g["bbox_pred"]).max() <= 1e-4
        if training:
            assert np.array_equal(out[7].cpu().numpy(), g["rois_label"])
            for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
                assert abs(float(out[i]) - float(g[name])) <= 1e-4 * max(1.0, abs(float(g[name]))), name


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_fgn_sibling_matches_reference_golden(golden_dir, dev, tag):
    """get_model('fgn') (utils.py:115-116) vs the reference's own outputs; in train mode its head's BatchNorm layers
    normalise with batch statistics and the running statistics they leave behind must match the reference's too"""
    import dana_amd
    from dana_amd import synthetic as S
    g = _load(golden_dir, "fgn_" + tag)
    training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("fgn", pretrained=False, way=way, shot=shot, classes=["fg", "bg"])
    m.load_state_dict(S.tame_fgn_weights(S.fill_state_dict(m.state_dict(), seed=wseed, profile="test")))
    m.to(dev)
    m.nms_inclusive = True
    m.train() if training else m.eval()
    inputs = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    np.random.seed(nseed)
    with torch.no_grad():
        out = m(*[t.to(dev) for t in inputs])
    r, rg = out[0].cpu().numpy().reshape(-1, 5), g["rois"].reshape(-1, 5)
    matched = _iou(r[:, 1:], rg[:, 1:]) >= 1 - 1e-3
    assert matched.mean() >= 0.97
    if not training:
        assert matched.all()
    if matched.all():
        assert np.abs(out[1].cpu().numpy() - g["cls_prob"]).max() <= 1e-4
        assert np.abs(out[2].cpu().numpy() - g["bbox_pred"]).max() <= 1e-4
        sd = m.state_dict()
        for k in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var"):
            assert np.abs(sd[k].cpu().numpy() - g[k]).max() <= 1e-4 * max(1.0, np.abs(g[k]).max()), k
        if training:
            assert int(sd["bn1.num_batches_tracked"]) == 2  # positive + negative head
            assert np.array_equal(out[7].cpu().numpy(), g["rois_label"])
            for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
                assert abs(float(out[i]) - float(g[name])) <= 1e-4 * max(1.0, abs(float(g[name]))), name


@pytest.mark.parametrize("training", [False, True], ids=["eval", "train"])
def test_folded_roi_positional_encoding_equals_the_two_output_roi_align(dev, training):
    """forward-only runs fold the RoI-level positional encoding (dana.py:259) into the Q projection and the query half of
    rcnn_transform_layer ((pooled + PE) W^T = pooled W^T + PE W^T, ONE N = 128 GEMM over `pooled`) and RoIAlign writes one
    output; the unfolded form (RoIAlign emits pooled and pooled + PE, two GEMMs) is what the backward saves. Same rois,
    cls_prob / bbox_pred / losses equal to fp32 roundoff."""
    import dana_amd
    from dana_amd import synthetic as S
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=2, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=7, profile="test"))
    m.to(dev)
    m.train() if training else m.eval()
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2 if training else 1, 2, 192, 256, seed=21)]
    outs = []
    for fold in (True, False):
        m.fold_roi_pe = fold
        np.random.seed(4)
        with torch.no_grad():
            outs.append(m(*inputs))
    a, b = outs
    # (round 4, second fold of the forward-only path: (A . S) . Wt_a^T re-associated as A . (S . Wt_a^T), dana.py:279-286)
    m.fold_roi_pe, m.fold_roi_attn = True, False
    np.random.seed(4)
    with torch.no_grad():
        c = m(*inputs)
    m.fold_roi_attn = True
    assert torch.equal(a[0], c[0])
    assert (a[1] - c[1]).abs().max().item() <= 2e-6 and (a[2] - c[2]).abs().max().item() <= 2e-5
    if training:
        for i in range(3, 7):
            assert abs(float(a[i]) - float(c[i])) <= 2e-6 * max(1.0, abs(float(c[i])))
    assert torch.equal(a[0], b[0])
    assert (a[1] - b[1]).abs().max().item() <= 2e-6 and (a[2] - b[2]).abs().max().item() <= 2e-5
    if training:
        for i in range(3, 7):
            assert abs(float(a[i]) - float(b[i])) <= 2e-6 * max(1.0, abs(float(b[i])))
        assert torch.equal(a[7], b[7])
