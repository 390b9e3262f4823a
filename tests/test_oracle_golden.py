"""CPU suite, part 1: the oracle (oracle/) against the golden vectors produced by RUNNING THE REFERENCE
(tests/golden/make_golden.py). Bit-exact for index / RoIAlign / IoU arithmetic; the end-to-end
oracle forward must reproduce the reference's outputs to fp32 roundoff."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as O, native as ON


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def test_anchor_tables(G):
    assert np.array_equal(O.generate_anchors(scales=[4, 8, 16, 32]), G["anchors_a4"])
    assert np.array_equal(O.generate_anchors(scales=[8, 16, 32]), G["anchors_a3"])
    # the 9-anchor table printed in the reference (generate_anchors.py:27-35) is MATLAB 1-based: minus 1 here
    assert np.array_equal(G["anchors_a3"][0], [-84., -40., 99., 55.])
    assert np.array_equal(G["anchors_a3"][8], [-168., -344., 183., 359.])
    assert np.array_equal(G["anchors_a4"][0], [-38., -16., 53., 31.])
    assert np.array_equal(G["anchors_a4"][11], [-168., -344., 183., 359.])


@pytest.mark.parametrize("tag,thr", [("n256_t03", 0.3), ("n256_t07", 0.7), ("n2048_t07", 0.7)])
def test_nms_matches_reference_cpu_op(G, tag, thr):
    keep = ON.nms(G["nms_%s_boxes" % tag], G["nms_%s_scores" % tag], thr, inclusive=True)
    assert np.array_equal(keep, G["nms_%s_keep" % tag])


def test_nms_tie_rule_differs_between_reference_cpu_and_cuda(G):
    b, s = G["nms_tie_boxes"], G["nms_tie_scores"]
    assert list(ON.nms(b, s, 0.5, inclusive=True)) == list(G["nms_tie_keep_ge"]) == [0, 2]
    assert list(ON.nms(b, s, 0.5, inclusive=False)) == [0, 1, 2]
    assert ON.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).shape == (0,)


def test_roi_align_matches_reference_cpu_op(G):
    for sr, key in ((0, "ra_out_sr0"), (2, "ra_out_sr2")):
        out = ON.roi_align_forward(G["ra_feat"], G["ra_rois"], 1.0 / 16, 7, 7, sr)
        assert np.array_equal(out, G[key])
    assert (G["ra_out_sr0"][5] == 0).all()  # a roi entirely outside the map pools zeros
    assert ON.roi_align_forward(G["ra_feat"], np.zeros((0, 5), np.float32), 1 / 16., 7, 7, 0).shape == (0, 8, 7, 7)


def test_decode_clip_and_overlaps_match_reference(G):
    anchors = O.anchor_grid(G["anchors_a4"], 5, 7, 16).unsqueeze(0).expand(2, -1, 4)
    p = O.clip_boxes(O.bbox_transform_inv(anchors, torch.from_numpy(G["dec_deltas"])), torch.from_numpy(G["dec_im_info"]))
    assert np.array_equal(p.numpy(), G["dec_out"])
    for b in range(2):
        pc = ON.decode_clip(G["anchors_a4"], G["dec_deltas"][b], 5, 7, 16, G["dec_im_info"][b, 0], G["dec_im_info"][b, 1])
        assert np.abs(pc - G["dec_out"][b]).max() < 1e-3
    ov = O.bbox_overlaps_batch(torch.from_numpy(G["ov_anchors"]), torch.from_numpy(G["ov_gt"]))
    assert np.array_equal(ov.numpy(), G["ov_out"])
    assert (G["ov_out"][:, 5] == -1).all() and (G["ov_out"][:, :, 3:] <= 0).all()  # zero-area masks


def test_positional_encoding(G):
    for L in (49, 400):
        assert np.array_equal(O.positional_encoding(L)[0, ::7, ::37].numpy(), G["pe%d_sample" % L])


@pytest.mark.parametrize("tag", ["eval_small_cisa", "eval_small_ba", "train_small_ba", "eval_small_product",
                                 "train_small_product"])
def test_oracle_forward_reproduces_reference_outputs(golden_dir, tag):
    """(the *_product fixtures: attention_type='product', dana.py:74-77,155-156,285-286 -- the oracle takes the type from the
    shapes of RCNN_rpn.RPN_Conv / rcnn_transform_layer in the state dict)"""
    import dana_amd
    from dana_amd import synthetic as S
    from dana_amd.dana import DAnARCNN
    g = np.load(os.path.join(golden_dir, "e2e_%s.npz" % tag))
    use_ba, training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"][:10]]
    if len(g["meta"]) > 10 and int(g["meta"][10]):
        m = DAnARCNN(["fg", "bg"], "product", 256, 256, pretrained=False, semantic_enhance=bool(use_ba), num_way=way,
                     num_shot=shot)
        m.create_architecture()
        assert m.RCNN_rpn.RPN_Conv.weight.shape[1] == 1024 and m.rcnn_transform_layer.weight.shape == (64, 1024)
        product = True
    else:
        product = False
        m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=bool(use_ba), way=way, shot=shot, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=wseed, profile="test")
    if product:
        sd = S.tame_product_weights(sd)
    inputs = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    np.random.seed(nseed)
    torch.set_num_threads(8)
    with torch.no_grad():
        out = O.forward(sd, *inputs, bool(training), way, shot, bool(use_ba), nms_inclusive=True)
    assert np.abs(out[0].numpy() - g["rois"]).max() <= 1e-3
    assert np.abs(out[1].numpy() - g["cls_prob"]).max() <= 2e-5
    assert np.abs(out[2].numpy() - g["bbox_pred"]).max() <= 2e-5
    if training:
        assert np.array_equal(out[7].numpy(), g["rois_label"])
        for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
            assert abs(float(out[i]) - float(g[name])) <= 2e-5, name
    else:
        assert out[3:] == (0, 0, 0, 0, None)


@pytest.mark.parametrize("tag", ["eval_small_cisa", "eval_small_ba", "eval_full_ba"])
def test_oracle_postprocess_matches_reference_detections(golden_dir, tag):
    """inference.py:106-140 + utils.py:312-317 on the reference's own forward outputs; detections whose score is
    exactly tied are excluded (the reference orders ties with an unstable sort inside nms, nms_cpu.cpp:24)."""
    g = np.load(os.path.join(golden_dir, "e2e_%s.npz" % tag))
    H, W = int(g["meta"][5]), int(g["meta"][6])
    im_info = torch.tensor([[float(H), float(W), 1.0]])
    tied = set(g["dets_tied_scores"].tolist())
    for thr, key in ((0.05, "dets_t05"), (0.62, "dets_t62")):
        d = O.postprocess(torch.from_numpy(g["rois"]), torch.from_numpy(g["cls_prob"]), torch.from_numpy(g["bbox_pred"]),
                          im_info, thresh=thr).numpy()
        a = d[[i for i in range(len(d)) if float(d[i, 4]) not in tied]]
        b = g[key][[i for i in range(len(g[key])) if float(g[key][i, 4]) not in tied]]
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-4
    assert len(g["dets_t62"]) < len(g["dets_t05"])


def test_pipeline_resize_restatement_matches_independent_bilinear():
    """oracle/pipeline_ref.cv_resize_linear (the cv2.INTER_LINEAR float path restated; cv2 itself is absent here) against
    torch's F.interpolate(bilinear, align_corners=False), an independent implementation of the same sampling rule"""
    import torch.nn.functional as F
    from oracle import pipeline_ref as P
    rng = np.random.RandomState(0)
    for h, w, fx in [(48, 64, 0.61), (20, 20, 16.0), (33, 31, 1.0), (40, 50, 2.5)]:
        src = rng.randn(h, w, 3).astype(np.float32) * 50
        t = torch.from_numpy(src).permute(2, 0, 1)[None]
        a = P.cv_resize_linear(src, fx=fx, fy=fx)
        b = F.interpolate(t, scale_factor=fx, mode="bilinear", align_corners=False, recompute_scale_factor=False)
        assert tuple(b.shape[2:]) == a.shape[:2]
        assert np.abs(a - b[0].permute(1, 2, 0).numpy()).max() <= 2e-3  # |values| ~ 50..200: 1e-5 relative
        d = P.cv_resize_linear(src, dsize=(2 * w + 1, h + 3))
        b3 = F.interpolate(t, size=(h + 3, 2 * w + 1), mode="bilinear", align_corners=False)
        assert np.abs(d - b3[0].permute(1, 2, 0).numpy()).max() <= 2e-3
    # identity scale is exact, and the blob is BGR, mean-subtracted (blob.py:38-39, minibatch.py:76-78)
    im = rng.randint(0, 256, size=(30, 45, 3)).astype(np.uint8)
    means = np.array([[[102.9801, 115.9465, 122.7717]]], dtype=np.float32)
    out, s = P.prep_im_for_blob(im, means, 30)
    assert s == 1.0 and np.array_equal(out, im[:, :, ::-1].astype(np.float32) - means)
    sup = P.support_crop(out, (5, 3, 24, 28), 64)  # taller than wide: height fits, width padded with zeros
    assert sup.shape == (3, 64, 64) and np.all(sup[:, :, 48:] == 0) and np.any(sup[:, :, 47] != 0)


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_oracle_frcnn_reproduces_reference_outputs(golden_dir, tag):
    """sibling `frcnn` (faster_rcnn.py:35-103): the oracle restatement vs the reference's outputs, and the product
    class carries the reference's parameter tree (328 state_dict entries, 52 trainable tensors / 28 000 846 parameters)"""
    import dana_amd
    from dana_amd import synthetic as S
    g = np.load(os.path.join(golden_dir, "e2e_frcnn_%s.npz" % tag))
    training, B, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("frcnn", pretrained=False, classes=["fg", "bg"])
    assert len(m.state_dict()) == 328 and "RCNN_cls_score.weight" in m.state_dict()
    tr = [p for p in m.parameters() if p.requires_grad]
    assert len(tr) == 52 and sum(p.numel() for p in tr) == 28000846
    sd = S.fill_state_dict(m.state_dict(), seed=wseed, profile="test")
    im_data, im_info, gt, nb, _ = S.episode_inputs(B, 1, 1, H, W, seed=iseed)
    np.random.seed(nseed)
    torch.set_num_threads(8)
    with torch.no_grad():
        out = O.frcnn_forward(sd, im_data, im_info, gt, nb, bool(training), nms_inclusive=True)
    assert np.abs(out[0].numpy() - g["rois"]).max() <= 1e-3
    assert np.abs(out[1].numpy() - g["cls_prob"]).max() <= 2e-5
    assert np.abs(out[2].numpy() - g["bbox_pred"]).max() <= 2e-5
    if training:
        assert np.array_equal(out[7].numpy(), g["rois_label"])
        for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
            assert abs(float(out[i]) - float(g[name])) <= 2e-5, name


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_oracle_meta_reproduces_reference_outputs(golden_dir, tag):
    """sibling `meta` (Meta R-CNN, meta.py:39-142,241-251): oracle restatement vs the reference's outputs; the product
    class carries the reference's parameter tree (RCNN_cls_score is a Sequential: keys RCNN_cls_score.0.*)"""
    import dana_amd
    from dana_amd import synthetic as S
    g = np.load(os.path.join(golden_dir, "e2e_meta_%s.npz" % tag))
    training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("meta", pretrained=False, way=way, shot=shot, classes=["fg", "bg"])
    assert len(m.state_dict()) == 328 and "RCNN_cls_score.0.weight" in m.state_dict()
    sd = S.fill_state_dict(m.state_dict(), seed=wseed, profile="test")
    im_data, im_info, gt, nb, sup = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    np.random.seed(nseed)
    torch.set_num_threads(8)
    with torch.no_grad():
        out = O.meta_forward(sd, im_data, im_info, gt, nb, sup, gt.clone(), bool(training), way, shot, nms_inclusive=True)
    assert np.abs(out[0].numpy() - g["rois"]).max() <= 1e-3
    assert np.abs(out[1].numpy() - g["cls_prob"]).max() <= 2e-5
    assert np.abs(out[2].numpy() - g["bbox_pred"]).max() <= 2e-5
    if training:
        assert np.array_equal(out[7].numpy(), g["rois_label"])
        for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
            assert abs(float(out[i]) - float(g[name])) <= 2e-5, name


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_oracle_fsod_reproduces_reference_outputs(golden_dir, tag):
    """sibling `fsod` (attention RPN + multi-relation head, fsod.py:79-249): oracle restatement vs the reference"""
    import dana_amd
    from dana_amd import synthetic as S
    g = np.load(os.path.join(golden_dir, "e2e_fsod_%s.npz" % tag))
    training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("fsod", pretrained=False, way=way, shot=shot, classes=["fg", "bg"])
    assert len(m.state_dict()) == 340 and "patch_conv_2.weight" in m.state_dict()
    sd = S.tame_fsod_weights(S.fill_state_dict(m.state_dict(), seed=wseed, profile="test"))
    im_data, im_info, gt, nb, sup = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    np.random.seed(nseed)
    torch.set_num_threads(8)
    with torch.no_grad():
        out = O.fsod_forward(sd, im_data, im_info, gt, nb, sup, bool(training), way, shot, nms_inclusive=True)
    assert np.abs(out[0].numpy() - g["rois"]).max() <= 1e-3
    assert np.abs(out[1].numpy() - g["cls_prob"]).max() <= 2e-5
    assert np.abs(out[2].numpy() - g["bbox_pred"]).max() <= 2e-5
    if training:
        assert np.array_equal(out[7].numpy(), g["rois_label"])
        for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
            assert abs(float(out[i]) - float(g[name])) <= 2e-5, name


@pytest.mark.parametrize("tag", ["eval_small", "train_small"])
def test_oracle_fgn_reproduces_reference_outputs(golden_dir, tag):
    """sibling `fgn` (fgn.py:45-165; train-mode BatchNorm in its head): oracle restatement vs the reference, including
    the running statistics the forward leaves behind"""
    import dana_amd
    from dana_amd import synthetic as S
    g = np.load(os.path.join(golden_dir, "e2e_fgn_%s.npz" % tag))
    training, B, way, shot, H, W, wseed, iseed, nseed = [int(v) for v in g["meta"]]
    m = dana_amd.get_model("fgn", pretrained=False, way=way, shot=shot, classes=["fg", "bg"])
    assert len(m.state_dict()) == 340 and "bn2.running_var" in m.state_dict()
    sd = S.tame_fgn_weights(S.fill_state_dict(m.state_dict(), seed=wseed, profile="test"))
    im_data, im_info, gt, nb, sup = S.episode_inputs(B, way if training else 1, shot, H, W, seed=iseed)
    np.random.seed(nseed)
    torch.set_num_threads(8)
    bn_state = {}
    with torch.no_grad():
        out = O.fgn_forward(sd, im_data, im_info, gt, nb, sup, bool(training), way, shot, nms_inclusive=True,
                            bn_state=bn_state)
    assert np.abs(out[0].numpy() - g["rois"]).max() <= 1e-3
    assert np.abs(out[1].numpy() - g["cls_prob"]).max() <= 2e-5
    assert np.abs(out[2].numpy() - g["bbox_pred"]).max() <= 2e-5
    for k in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var"):
        assert np.abs(bn_state[k].numpy() - g[k]).max() <= 1e-5 * max(1.0, np.abs(g[k]).max()), k
    if training:
        assert np.array_equal(out[7].numpy(), g["rois_label"])
        for i, name in ((3, "rpn_loss_cls"), (4, "rpn_loss_bbox"), (5, "RCNN_loss_cls"), (6, "RCNN_loss_bbox")):
            assert abs(float(out[i]) - float(g[name])) <= 2e-5, name


def test_roi_align_backward_oracle_is_the_adjoint_of_the_pinned_forward():
    """oracle.native.roi_align_backward (restating the reference's CUDA scatter, ROIAlign_cuda.cu:125-254; the reference
    has no CPU backward to run, ROIAlign.h:44) == autograd through the differentiable restatement of the forward, which
    is itself pinned bit-exactly to the reference's CPU op via the C forward"""
    import torch
    from oracle import native, model_ref as O
    torch.manual_seed(0)
    feat = torch.randn(2, 3, 20, 30, dtype=torch.float64, requires_grad=True)
    rois = torch.tensor([[0, 10., 20., 200., 180.], [1, 0., 0., 479., 319.], [1, 100., 50., 130., 90.],
                         [0, -30., -20., 40., 60.], [1, 300., 200., 700., 500.], [0, 0., 0., 0., 0.]])
    y = O.roi_align_torch(feat, rois, 1 / 16., 7)
    assert np.abs(y.detach().float().numpy() -
                  native.roi_align_forward(feat.detach().float().numpy(), rois.numpy(), 1 / 16., 7, 7, 0)).max() < 1e-5
    g = torch.randn_like(y)
    y.backward(g)
    got = native.roi_align_backward(g.float().numpy(), rois.numpy(), 1 / 16., 7, 7, 2, 3, 20, 30, 0)
    assert np.abs(got - feat.grad.numpy()).max() <= 1e-6 * np.abs(got).max()


def test_pipeline_resize_known_answer_vectors_of_cv2_inter_linear():
    """cv2 is not installed in this image, so the resize restatement cannot be run against it; these are HAND-COMPUTED
    known answers of cv2.resize(float32, interpolation=cv2.INTER_LINEAR) from OpenCV's documented sampling rule
    (imgproc/resize.cpp: fx = (dx + 0.5) * scale - 0.5; sx = floor(fx); sx < 0 -> (0, weight 0); sx >= n - 1 ->
    (n - 1, weight 0); dst = src[sx] * (1 - fx) + src[sx + 1] * fx, horizontal then vertical) on inputs whose results
    are exactly representable -- they pin the pixel-centre convention, the edge clamps and the pass order, the
    three things a restatement can get wrong; rounding inside one pass is not exercised."""
    from oracle import pipeline_ref as P

    def r(a, dsize=None, **kw):
        a = np.asarray(a, np.float32)
        return P.cv_resize_linear(a.reshape(a.shape[0], a.shape[1], 1), dsize=dsize, **kw)[:, :, 0]

    # 1 x 2 -> 1 x 4 (scale 0.5): centres at -0.25, 0.25, 0.75, 1.25 -> clamp, 1/4, 3/4, clamp
    assert np.array_equal(r([[0, 1]], dsize=(4, 1)), [[0, 0.25, 0.75, 1]])
    # 1 x 4 -> 1 x 2 (scale 2): centres at 0.5, 2.5 -> midpoints of (0,1) and (2,3)
    assert np.array_equal(r([[0, 1, 2, 3]], dsize=(2, 1)), [[0.5, 2.5]])
    # 1 x 3 -> 1 x 6 (scale 0.5): -0.25, 0.25, 0.75, 1.25, 1.75, 2.25
    assert np.array_equal(r([[0, 4, 8]], dsize=(6, 1)), [[0, 1, 3, 5, 7, 8]])
    # 2 x 2 -> 4 x 4: separable, both axes as the first case
    got = r([[0, 4], [8, 12]], dsize=(4, 4))
    col = np.array([0, 0.25, 0.75, 1], np.float32)
    assert np.array_equal(got, 8 * col[:, None] + 4 * col[None, :])
    # fx / fy form (blob.py:50: cv2.resize(im, None, None, fx=s, fy=s)): dsize = round(src * f), scale = 1 / f
    assert np.array_equal(r([[0, 2, 4, 6]], fx=0.5, fy=1.0), [[1, 5]])
    assert r(np.zeros((3, 5)), fx=1.6, fy=2.0).shape == (6, 8)
    # identity when the size does not change
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(r(a, dsize=(4, 3)), a)
    # 1 x 5 -> 1 x 3 (scale 5/3): centres 1/3, 2, 11/3 -> 1/3 between (0,1), exactly 2, 2/3 between (3,4)
    got = r([[0, 3, 6, 9, 12]], dsize=(3, 1))
    assert np.allclose(got, [[1.0, 6.0, 11.0]], atol=1e-5)
