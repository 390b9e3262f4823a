"""Episode input pipeline kernels through the C ABI vs the CPU restatement of the loaders."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEANS = np.array([102.9801, 115.9465, 122.7717], dtype=np.float32)


@pytest.mark.parametrize("h,w,target,flipped", [(375, 500, 600, False), (480, 333, 600, True), (600, 800, 600, False),
                                                (97, 131, 64, True)])
def test_prep_im_for_blob_vs_restatement(dev, h, w, target, flipped):
    from dana_amd import pipeline as PL
    from oracle import pipeline_ref as P
    rng = np.random.RandomState(h + w)
    im = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    ref, s_ref = P.prep_im_for_blob(im, MEANS, target, flipped)
    out, s = PL.prep_im_for_blob(torch.from_numpy(im).to(dev), MEANS, target, flipped=flipped)
    assert s == s_ref and tuple(out.shape) == ref.shape
    assert min(out.shape[0], out.shape[1]) == target
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-4  # same formula, float32 evaluation order differs at most
    if target == min(h, w):  # identity scale: exact
        assert np.array_equal(out.cpu().numpy(), ref)


def test_support_crop_and_query_pad_vs_restatement(dev):
    from dana_amd import pipeline as PL
    from oracle import pipeline_ref as P
    rng = np.random.RandomState(3)
    im = (rng.randn(200, 260, 3) * 60).astype(np.float32)
    d = torch.from_numpy(im).to(dev)
    for box in [(10, 20, 120, 60), (30, 5, 70, 190), (0, 0, 259, 199), (100, 100, 101, 140)]:
        ref = P.support_crop(im, box, 320)
        out = PL.support_crop(d, box, 320)
        assert np.abs(out.cpu().numpy() - ref).max() <= 2e-4, box
        assert np.array_equal(out.cpu().numpy() == 0, ref == 0) or np.abs(out.cpu().numpy() - ref).max() <= 2e-4
    for (ys, xs, ch, cw, oh, ow) in [(0, 0, 200, 260, 200, 300), (10, 0, 150, 260, 160, 260), (0, 40, 200, 200, 200, 200)]:
        ref = P.crop_pad_chw(im, ys, xs, ch, cw, oh, ow)
        out = PL.crop_pad_chw(d, ys, xs, ch, cw, oh, ow)
        assert np.array_equal(out.cpu().numpy(), ref)


def test_episode_holders_feed_the_model(dev):
    """raw uint8 frames -> holders -> DAnARCNN eval forward: same detections as feeding the restated CPU pipeline"""
    import dana_amd
    from dana_amd import pipeline as PL, synthetic as S
    from oracle import pipeline_ref as P
    rng = np.random.RandomState(9)
    q = rng.randint(0, 256, size=(120, 160, 3)).astype(np.uint8)
    sups = [rng.randint(0, 256, size=(90, 120, 3)).astype(np.uint8) for _ in range(2)]
    boxes = [(10, 8, 70, 60), (20, 30, 100, 80)]
    target = 192
    hold = PL.EpisodeHolders(1, 1, 2, 192, 256, dev)
    qd, qs = PL.prep_im_for_blob(torch.from_numpy(q).to(dev), MEANS, target)
    hold.put_query(0, qd, qs)
    ref_q, _ = P.prep_im_for_blob(q, MEANS, target)
    ref_sup = np.zeros((1, 2, 3, 320, 320), dtype=np.float32)
    for i, (s_im, bx) in enumerate(zip(sups, boxes)):
        sd, ss = PL.prep_im_for_blob(torch.from_numpy(s_im).to(dev), MEANS, target)
        sb = (np.array(bx, dtype=np.float32) * ss).astype(np.int16)  # fs_loader.py:119
        hold.put_support(0, i, sd, sb)
        rs, _ = P.prep_im_for_blob(s_im, MEANS, target)
        ref_sup[0, i] = P.support_crop(rs, sb, 320)
    hold.put_boxes(0, np.array([[10, 10, 100, 100, 1]], dtype=np.float32))
    im_data, im_info, gt, nb, sup = hold.tensors()
    assert tuple(im_data.shape) == (1, 3, 192, 256) and abs(float(im_info[0, 2]) - qs) < 1e-6
    ref_data = P.crop_pad_chw(ref_q, 0, 0, ref_q.shape[0], ref_q.shape[1], 192, 256)[None]
    assert np.abs(im_data.cpu().numpy() - ref_data).max() <= 2e-4
    assert np.abs(sup.cpu().numpy() - ref_sup).max() <= 2e-4
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=1, shot=2, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=4, profile="test"))
    m.to(dev).eval()
    with torch.no_grad():
        out = m(im_data, im_info, gt, nb, sup)
    assert out[0].shape[0] == 1 and torch.isfinite(out[1]).all()
