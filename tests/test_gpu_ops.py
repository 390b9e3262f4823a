"""GPU parity of the native operators (through the C ABI) against the oracle and the reference's
golden vectors. Bit-exact for index outputs and for RoIAlign / IoU arithmetic (kernels compiled
with -ffp-contract=off); stated float tolerances elsewhere."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def _ops():
    import dana_amd
    return dana_amd.ops


def _oracle():
    from oracle import native
    return native


def test_roi_align_golden_nchw(G, dev):
    ops = _ops()
    feat, rois = torch.from_numpy(G["ra_feat"]).to(dev), torch.from_numpy(G["ra_rois"]).to(dev)
    for sr, key in ((0, "ra_out_sr0"), (2, "ra_out_sr2")):
        out = ops.roi_align_forward(feat, rois, 1.0 / 16, 7, 7, sr).cpu().numpy()
        assert np.array_equal(out, G[key]), "RoIAlign NCHW differs from the reference golden (sr=%d)" % sr


def test_roi_align_nhwc_matches_nchw_and_pe(G, dev):
    ops = _ops()
    feat = torch.from_numpy(G["ra_feat"]).to(dev)  # [2,8,38,63]
    rois = torch.from_numpy(G["ra_rois"]).to(dev)
    B, C, H, W = feat.shape
    stride = 16  # channel block inside a wider pixel row, like base_feat inside the concat buffer
    buf = torch.full((B * H * W, stride), 7.0, device=dev)
    buf[:, :C] = feat.permute(0, 2, 3, 1).reshape(-1, C)
    pe = torch.randn(49, C, device=dev)
    ref = torch.from_numpy(G["ra_out_sr0"]).to(dev).permute(0, 2, 3, 1).reshape(-1, 49, C)
    # default = per-sample kernel: the reference's summation order, bit for bit
    pooled, pooled_pe = ops.roi_align_forward_nhwc(buf, B, H, W, C, stride, rois, 1.0 / 16, 7, 0, pe=pe)
    assert torch.equal(pooled, ref)
    assert torch.equal(pooled_pe, ref + pe.unsqueeze(0))


def test_roi_align_nhwc_is_bit_exact_on_wild_rois(dev):
    """the NHWC kernels on 300 random rois -- degenerate, out of bounds, bins of 0.1 .. 30 cells, sampling_ratio 0 and 2,
    channel counts that are not a multiple of the workgroup / of a slice -- against the oracle's restatement of
    cpu/ROIAlign_cpu.cpp: array_equal"""
    ops, orc = _ops(), _oracle()
    rng = np.random.default_rng(6)
    # (channel counts: one, two, four and eight 256-wide slices pinned to XCDs, a ragged last slice (300), and 1028 = five
    # slices, which do not divide the 8 XCDs -> the workgroup-per-bin kernel)
    for (B, C, H, W), sr in (((2, 64, 50, 84), 0), ((2, 1028, 12, 20), 0), ((1, 8, 200, 320), 0), ((2, 64, 50, 84), 2),
                             ((2, 512, 12, 20), 0), ((1, 300, 14, 9), 0), ((2, 1024, 10, 16), 2), ((1, 2048, 9, 11), 0)):
        feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
        rois = np.zeros((300, 5), np.float32)
        x1 = rng.uniform(-40, 16 * W, 300); y1 = rng.uniform(-40, 16 * H, 300)
        rois[:, 0] = rng.integers(0, B, 300)
        rois[:, 1], rois[:, 2] = x1, y1
        rois[:, 3], rois[:, 4] = x1 + rng.uniform(0, 17 * W, 300), y1 + rng.uniform(0, 17 * H, 300)
        rois[:8, 3:] = rois[:8, 1:3]  # zero-size boxes
        nhwc = torch.from_numpy(feat).to(dev).permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()
        got, _ = ops.roi_align_forward_nhwc(nhwc, B, H, W, C, C, torch.from_numpy(rois).to(dev), 1 / 16., 7, sr)
        ref = orc.roi_align_forward(feat, rois, 1 / 16., 7, 7, sr)  # [R][C][7][7]
        ref = np.transpose(ref, (0, 2, 3, 1)).reshape(300, 49, C)
        assert np.array_equal(got.cpu().numpy(), ref), (B, C, H, W, sr)


def test_roi_align_large_random_vs_oracle(dev):
    ops, orc = _ops(), _oracle()
    rng = np.random.default_rng(5)
    feat = rng.normal(size=(2, 32, 38, 63)).astype(np.float32)
    rois = np.zeros((300, 5), np.float32)
    x1 = rng.uniform(-20, 990, 300); y1 = rng.uniform(-20, 590, 300)
    rois[:, 0] = rng.integers(0, 2, 300)
    rois[:, 1], rois[:, 2] = x1, y1
    rois[:, 3], rois[:, 4] = x1 + rng.uniform(0, 500, 300), y1 + rng.uniform(0, 400, 300)
    out = ops.roi_align_forward(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev), 1 / 16., 7, 7, 0)
    assert np.array_equal(out.cpu().numpy(), orc.roi_align_forward(feat, rois, 1 / 16., 7, 7, 0))


def test_roi_align_empty_and_cpu_raises(dev):
    ops = _ops()
    feat = torch.zeros(1, 4, 8, 8, device=dev)
    out = ops.roi_align_forward(feat, torch.zeros(0, 5, device=dev), 1.0, 7, 7, 0)
    assert out.shape == (0, 4, 7, 7)
    with pytest.raises(RuntimeError):
        ops.roi_align_forward(feat.cpu(), torch.zeros(1, 5), 1.0, 7, 7, 0)


def test_roi_align_backward_matches_autograd_of_dense_formulation(dev):
    """No reference CPU backward exists (ROIAlign.h:44); check against finite structure: the backward of
    a linear op is its transpose, so <gout, fwd(x)> == <bwd(gout), x> for random x, gout."""
    ops = _ops()
    torch.manual_seed(0)
    x = torch.randn(2, 6, 20, 30, device=dev)
    rois = torch.tensor([[0, 10., 20., 200., 180.], [1, 0., 0., 479., 319.], [1, 100., 50., 130., 90.]], device=dev)
    y = ops.roi_align_forward(x, rois, 1 / 16., 7, 7, 0)
    g = torch.randn_like(y)
    gx = ops.roi_align_backward(g, rois, 1 / 16., 7, 7, 2, 6, 20, 30, 0)
    lhs, rhs = (g.double() * y.double()).sum().item(), (gx.double() * x.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_roi_align_backward_vs_oracle_scatter(dev):
    """oracle.native.roi_align_backward restates the reference's CUDA scatter (ROIAlign_cuda.cu:125-254): same fp32
    per-sample contributions, float64 sum. The HIP kernels (NCHW op of `_C`, NHWC fast path) accumulate the same
    contributions in fp32 in an unordered (atomic) order: |d| <= 2e-6 of the gradient's scale."""
    ops, orc = _ops(), _oracle()
    rng = np.random.default_rng(11)
    B, C, H, W, R = 2, 16, 38, 63, 96
    rois = np.zeros((R, 5), np.float32)
    x1 = rng.uniform(-20, 990, R); y1 = rng.uniform(-20, 590, R)
    rois[:, 0] = rng.integers(0, B, R)
    rois[:, 1], rois[:, 2] = x1, y1
    rois[:, 3], rois[:, 4] = x1 + rng.uniform(0, 500, R), y1 + rng.uniform(0, 400, R)
    rois[:4] = [[0, 0, 0, 0, 0], [1, 50, 50, 40, 40], [0, 2000, 2000, 2100, 2100], [1, 0, 0, 999, 599]]  # degenerate / OOB
    g = rng.normal(size=(R, C, 7, 7)).astype(np.float32)
    for sr in (0, 2):
        ref = orc.roi_align_backward(g, rois, 1 / 16., 7, 7, B, C, H, W, sr)
        tol = 2e-6 * np.abs(ref).max()
        got = ops.roi_align_backward(torch.from_numpy(g).to(dev), torch.from_numpy(rois).to(dev), 1 / 16., 7, 7, B, C,
                                     H, W, sr).cpu().numpy()
        assert np.abs(got - ref).max() <= tol
        g_nhwc = torch.from_numpy(g).to(dev).permute(0, 2, 3, 1).contiguous()  # [R,7,7,C]
        got2_t = ops.roi_align_backward(g_nhwc, torch.from_numpy(rois).to(dev), 1 / 16., 7, 7, B, C, H, W, sr,
                                        layout=ops.NHWC)
        got2 = got2_t.permute(0, 3, 1, 2).cpu().numpy()
        assert np.abs(got2 - ref).max() <= tol
        # the NHWC path is a GATHER over the feature cells (fixed summation order, no atomics, every cell written once --
        # no zero fill): the same call again returns the same bits, also into a buffer that held garbage
        again = ops.roi_align_backward(g_nhwc, torch.from_numpy(rois).to(dev), 1 / 16., 7, 7, B, C, H, W, sr, layout=ops.NHWC)
        assert torch.equal(got2_t, again)


def test_roi_layers_autograd_wrappers(dev):
    """dana_amd.roi_layers.{ROIAlign, ROIPool, nms} -- the mirror of lib/model/roi_layers/*.py a reference caller
    imports -- through torch autograd: forward == oracle (bit-exact), backward == the oracle's scatter / argmax routing"""
    import dana_amd
    from dana_amd.roi_layers import ROIAlign, ROIPool, nms
    orc = _oracle()
    rng = np.random.default_rng(3)
    B, C, H, W, R = 2, 8, 20, 30, 12
    feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    x1 = rng.uniform(0, 400, R); y1 = rng.uniform(0, 250, R)
    rois[:, 0] = rng.integers(0, B, R)
    rois[:, 1], rois[:, 2], rois[:, 3], rois[:, 4] = x1, y1, x1 + rng.uniform(8, 200, R), y1 + rng.uniform(8, 150, R)
    x = torch.from_numpy(feat).to(dev).requires_grad_(True)
    r = torch.from_numpy(rois).to(dev)
    layer = ROIAlign((7, 7), 1.0 / 16.0, 0)
    assert "ROIAlign(" in repr(layer)
    y = layer(x, r)
    assert np.array_equal(y.detach().cpu().numpy(), orc.roi_align_forward(feat, rois, 1 / 16., 7, 7, 0))
    g = rng.normal(size=y.shape).astype(np.float32)
    y.backward(torch.from_numpy(g).to(dev))
    ref = orc.roi_align_backward(g, rois, 1 / 16., 7, 7, B, C, H, W, 0)
    assert np.abs(x.grad.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()
    # ROIPool: forward vs the oracle, backward routes each output gradient to its argmax (ROIPool_cuda.cu:80-108)
    x2 = torch.from_numpy(feat).to(dev).requires_grad_(True)
    pool = ROIPool((7, 7), 1.0 / 16.0)
    assert "ROIPool(" in repr(pool)
    y2 = pool(x2, r)
    ref_out, ref_arg = orc.roi_pool_forward(feat, rois, 1 / 16., 7, 7)
    assert np.array_equal(y2.detach().cpu().numpy(), ref_out)
    y2.backward(torch.from_numpy(g).to(dev))
    gin = np.zeros((B, C, H * W), np.float64)
    for n in range(R):
        b = int(rois[n, 0])
        for c in range(C):
            a = ref_arg[n, c].reshape(-1)
            ok = a >= 0
            np.add.at(gin[b, c], a[ok], g[n, c].reshape(-1)[ok].astype(np.float64))
    assert np.abs(x2.grad.cpu().numpy().reshape(B, C, -1) - gin).max() <= 2e-6 * max(1.0, np.abs(gin).max())
    # nms re-export: the `_C.nms` contract (kept original indices, ascending, `>` rule)
    dets = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60]], dtype=torch.float32, device=dev)
    keep = nms(dets, torch.tensor([0.9, 0.8, 0.7], device=dev), 0.5)
    assert keep.tolist() == [0, 2] and keep.dtype == torch.int64


def test_torch_ops_dana_match_ctypes_binding_and_are_differentiable(dev):
    """torch.ops.dana.* (csrc/torch_ops.cpp, TORCH_LIBRARY over the C ABI) vs the ctypes binding of the same entry points:
    identical results; torch.ops.dana.roi_align / roi_pool back-propagate through their registered autograd formulas"""
    import dana_amd
    ops, orc = _ops(), _oracle()
    assert dana_amd._C.BINDING == "torch.ops.dana"
    rng = np.random.default_rng(17)
    B, C, H, W, R = 2, 8, 20, 30, 16
    feat = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32)).to(dev)
    rois = np.zeros((R, 5), np.float32)
    x1 = rng.uniform(0, 400, R); y1 = rng.uniform(0, 250, R)
    rois[:, 0] = rng.integers(0, B, R)
    rois[:, 1], rois[:, 2], rois[:, 3], rois[:, 4] = x1, y1, x1 + rng.uniform(8, 200, R), y1 + rng.uniform(8, 150, R)
    r = torch.from_numpy(rois).to(dev)
    a = torch.ops.dana.roi_align_forward(feat, r, 1 / 16., 7, 7, 0)
    assert torch.equal(a, ops.roi_align_forward(feat, r, 1 / 16., 7, 7, 0))
    p, arg = torch.ops.dana.roi_pool_forward(feat, r, 1 / 16., 7, 7)
    p2, arg2 = ops.roi_pool_forward(feat, r, 1 / 16., 7, 7)
    assert torch.equal(p, p2) and torch.equal(arg, arg2)
    dets = torch.from_numpy(rng.uniform(0, 300, (400, 2)).astype(np.float32)).to(dev)
    dets = torch.cat([dets, dets + torch.from_numpy(rng.uniform(20, 120, (400, 2)).astype(np.float32)).to(dev)], 1)
    sc = torch.from_numpy(rng.permutation(400).astype(np.float32)).to(dev)
    k1 = torch.ops.dana.nms(dets, sc, 0.5)
    assert torch.equal(k1.cpu(), ops.nms(dets, sc, 0.5).cpu())
    assert np.array_equal(k1.cpu().numpy(), orc.nms(dets.cpu().numpy(), sc.cpu().numpy(), 0.5, inclusive=False))
    x = feat.clone().requires_grad_(True)
    y = torch.ops.dana.roi_align(x, r, 1 / 16., 7, 7, 0)
    g = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32)).to(dev)
    y.backward(g)
    ref = orc.roi_align_backward(g.cpu().numpy(), rois, 1 / 16., 7, 7, B, C, H, W, 0)
    assert np.abs(x.grad.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()
    x2 = feat.clone().requires_grad_(True)
    torch.ops.dana.roi_pool(x2, r, 1 / 16., 7, 7).backward(g)
    ref_p = ops.roi_pool_backward(g, feat, r, arg, 1 / 16., 7, 7, B, C, H, W)  # (atomic scatter: order-dependent rounding)
    assert float((x2.grad - ref_p).abs().max()) <= 2e-6 * float(ref_p.abs().max())


def test_roi_pool_vs_oracle(dev):
    ops, orc = _ops(), _oracle()
    rng = np.random.default_rng(9)
    feat = rng.normal(size=(2, 8, 38, 63)).astype(np.float32)
    rois = np.zeros((64, 5), np.float32)
    x1 = rng.uniform(0, 900, 64); y1 = rng.uniform(0, 500, 64)
    rois[:, 0] = rng.integers(0, 2, 64)
    rois[:, 1], rois[:, 2], rois[:, 3], rois[:, 4] = x1, y1, x1 + rng.uniform(0, 300, 64), y1 + rng.uniform(0, 300, 64)
    out, arg = ops.roi_pool_forward(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev), 1 / 16., 7, 7)
    o_ref, a_ref = orc.roi_pool_forward(feat, rois, 1 / 16., 7, 7)
    assert np.array_equal(out.cpu().numpy(), o_ref) and np.array_equal(arg.cpu().numpy(), a_ref)
    g = torch.randn_like(out)
    gx = ops.roi_pool_backward(g, torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev), arg, 1 / 16., 7, 7,
                               2, 8, 38, 63)
    # scatter-add of g at argmax, done densely on the host
    ref = np.zeros((2, 8, 38 * 63), np.float64)
    gn, an = g.cpu().numpy(), arg.cpu().numpy()
    for n in range(64):
        b = int(rois[n, 0])
        for c in range(8):
            for p in range(49):
                a = an[n, c].reshape(-1)[p]
                if a >= 0:
                    ref[b, c, a] += gn[n, c].reshape(-1)[p]
    assert np.allclose(gx.cpu().numpy().reshape(2, 8, -1), ref, atol=1e-4)


@pytest.mark.parametrize("tag,thr", [("n256_t03", 0.3), ("n256_t07", 0.7), ("n2048_t07", 0.7)])
def test_nms_golden(G, dev, tag, thr):
    ops = _ops()
    boxes, scores = torch.from_numpy(G["nms_%s_boxes" % tag]).to(dev), torch.from_numpy(G["nms_%s_scores" % tag]).to(dev)
    keep_ge = ops.nms(boxes, scores, thr, inclusive=True).cpu().numpy()  # the reference CPU op's rule
    assert np.array_equal(keep_ge, G["nms_%s_keep" % tag])
    keep_gt = ops.nms(boxes, scores, thr, inclusive=False).cpu().numpy()  # the reference CUDA op's rule
    assert np.array_equal(keep_gt, G["nms_%s_keep_gt" % tag])


def test_nms_tie_and_empty(G, dev):
    import dana_amd
    ops = dana_amd.ops
    b, s = torch.from_numpy(G["nms_tie_boxes"]).to(dev), torch.from_numpy(G["nms_tie_scores"]).to(dev)
    assert list(ops.nms(b, s, 0.5, inclusive=True).cpu().numpy()) == list(G["nms_tie_keep_ge"])
    assert list(dana_amd._C.nms(b, s, 0.5).cpu().numpy()) == [0, 1, 2]  # `>` keeps the exact tie (nms.cu:60)
    e = dana_amd._C.nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5)
    assert e.numel() == 0 and e.dtype == torch.int64


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 6000, 12000])
def test_nms_sizes_vs_oracle(dev, n):
    ops, orc = _ops(), _oracle()
    rng = np.random.default_rng(n)
    c = rng.uniform([0, 0], [1000, 600], size=(max(n // 10, 1), 2))
    ctr = c[rng.integers(0, len(c), n)] + rng.normal(0, 10, size=(n, 2))
    wh = rng.uniform(16, 300, size=(n, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32)  # distinct -> unambiguous order
    keep = ops.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.7).cpu().numpy()
    assert np.array_equal(keep, orc.nms(boxes, scores, 0.7, inclusive=False))


def test_nms_max_keep_truncates_in_order(dev):
    ops, orc = _ops(), _oracle()
    rng = np.random.default_rng(3)
    n = 5000
    ctr = rng.uniform([0, 0], [1000, 600], size=(n, 2))
    wh = rng.uniform(16, 120, size=(n, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)  # already "sorted": index = rank
    keep_all = orc.nms(boxes, -np.arange(n, dtype=np.float32), 0.7, inclusive=False)
    keep, num = ops.nms_sorted(torch.from_numpy(boxes).to(dev).view(1, n, 4), 0.7, False, max_keep=300)
    assert int(num[0]) == 300 and np.array_equal(keep[0].cpu().numpy(), keep_all[:300])


@pytest.mark.parametrize("clusters,max_keep", [(40, 300), (400, 300), (3000, 300), (150, 1000)])
def test_nms_two_pass_hand_off_vs_oracle(dev, clusters, max_keep):
    """dana_nms scans the first ~3*max_keep boxes first and only then the rest: problems that finish in pass 1, problems
    that need pass 2 and problems that never reach max_keep, side by side in one call (nms.cu:70-131 semantics)"""
    ops, orc = _ops(), _oracle()
    n, P = 8000, 4  # large enough for dana_nms to band the mask (problems * blocks^2 / 2 >= 20000 tiles)
    boxes = np.zeros((P, n, 4), np.float32)
    for q in range(P):
        rng = np.random.default_rng(100 * clusters + q)
        k = max(clusters // (1 + 3 * q), 4)  # fewer clusters -> fewer survivors -> the scan runs further
        c = rng.uniform([0, 0], [1000, 600], size=(k, 2))
        ctr = c[rng.integers(0, k, n)] + rng.normal(0, 4, size=(n, 2))
        wh = rng.uniform(40, 60, size=(n, 2))
        boxes[q] = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
    keep, num = ops.nms_sorted(torch.from_numpy(boxes).to(dev), 0.7, False, max_keep=max_keep)
    seen = set()
    for q in range(P):
        ref = orc.nms(boxes[q], -np.arange(n, dtype=np.float32), 0.7, inclusive=False)[:max_keep]
        assert int(num[q]) == len(ref)
        assert np.array_equal(keep[q, :len(ref)].cpu().numpy(), ref)
        seen.add("full" if len(ref) == max_keep else "short")
    if clusters == 40:
        assert seen == {"full", "short"}  # the case list really covers both outcomes in one launch


@pytest.mark.parametrize("kind", ["disjoint", "identical", "chain"])
def test_nms_dataflow_scan_extremes_and_repeatability(dev, kind):
    """the barrier-free scan (resolver / word loaders / row workers talking through LDS flags): every box kept (64 rows
    folded per block), one box suppressing all others (nothing folded), a chain in which every box overlaps only its
    neighbour (the fixed point needs its full depth) -- 20 runs each must give the oracle's list every time"""
    ops, orc = _ops(), _oracle()
    n = 6000  # 94 column blocks: one band, two column groups at its head
    if kind == "disjoint":
        gx, gy = np.meshgrid(np.arange(100), np.arange(60))
        boxes = np.stack([gx.ravel() * 20, gy.ravel() * 20, gx.ravel() * 20 + 9, gy.ravel() * 20 + 9], 1).astype(np.float32)
    elif kind == "identical":
        boxes = np.tile(np.array([[10, 10, 200, 150]], np.float32), (n, 1))
    else:
        x = np.arange(n, dtype=np.float32) * 3  # IoU of neighbours = 8/14 > 0.5, of second neighbours 5/17 < 0.5
        boxes = np.stack([x, np.zeros(n, np.float32), x + 10, np.full(n, 10, np.float32)], 1)
    thr = 0.5
    ref = orc.nms(boxes, -np.arange(n, dtype=np.float32), thr, inclusive=False)
    t = torch.from_numpy(boxes).to(dev).view(1, n, 4).repeat(3, 1, 1).contiguous()
    for _ in range(20):
        keep, num = ops.nms_sorted(t, thr, False)
        for q in range(3):
            assert int(num[q]) == len(ref)
            assert np.array_equal(keep[q, :len(ref)].cpu().numpy(), ref)


def test_sort_desc_stable(dev):
    ops = _ops()
    rng = np.random.default_rng(0)
    s = rng.uniform(size=(3, 28728)).astype(np.float32)
    s[:, 100:200] = 0.5  # ties keep index order
    order, ss = ops.sort_desc(torch.from_numpy(s).to(dev))
    ref = np.argsort(-s, axis=1, kind="stable")
    assert np.array_equal(order.cpu().numpy(), ref)
    assert np.array_equal(ss.cpu().numpy(), np.take_along_axis(s, ref, 1))


@pytest.mark.parametrize("n,topn", [(21546, 12000), (21546, 6000), (37800, 12000), (2394, 12000), (12288, 12288),
                                    (40960, 300), (1, 5), (65, 64), (28728, 12000), (50000, 12000), (4097, 4097), (100000, 2000),
                                    (6000, 10)])
def test_topk_desc_is_the_head_of_the_stable_sort(dev, n, topn):
    """the hand-written top-k sorts (proposal_layer.py:135-150) against numpy's stable sort: the single-workgroup select +
    LDS radix sort (mode 2: wherever it can run), the multi-workgroup sample sort (mode 1: every row, also the short ones) and
    the measured dispatch: random scores, heavy ties (the cut falls INSIDE a run of equal scores), negative / zero / denormal
    values"""
    ops = _ops()
    from dana_amd._lib import lib
    rng = np.random.default_rng(n + topn)
    B = 3
    s = rng.uniform(-1.0, 1.0, size=(B, n)).astype(np.float32)
    s[1] = np.round(s[1] * 8) / 8          # 17 distinct values: every cut is inside a tie run
    s[2, ::3] = 0.0
    s[2, 1::7] = -0.0
    s[2, 5::11] = 1e-42                    # denormal
    m = min(topn, n)
    # -0.0 sorts behind +0.0 (the key is the bit pattern): compare through the same total order
    bits = s.view(np.uint32).astype(np.int64)
    key = np.where(bits & 0x80000000, (~bits) & 0xFFFFFFFF, bits | 0x80000000)
    ref = np.argsort(-key, axis=1, kind="stable")[:, :m]
    for mode in (2, 1, 0):
        lib().call("dana_set_sort_mode", mode)
        try:
            order, ss = ops.topk_desc(torch.from_numpy(s).to(dev), topn)
            torch.cuda.synchronize()
        finally:
            lib().call("dana_set_sort_mode", 0)
        assert np.array_equal(order.cpu().numpy(), ref), mode
        assert np.array_equal(ss.cpu().numpy().view(np.uint32), np.take_along_axis(s, ref, 1).view(np.uint32)), mode


def test_sample_sort_on_rows_beyond_its_tuned_sizes(dev):
    """rows longer than 65 536 scores (more keys per classify workgroup) and a row whose buckets exceed the LDS bucket sort
    (ranked from global memory): correct, not fast"""
    ops = _ops()
    rng = np.random.default_rng(5)
    for n, topn in ((200000, 3000), (600000, 600000)):
        s = rng.uniform(-1.0, 1.0, size=(1, n)).astype(np.float32)
        order, ss = ops.topk_desc(torch.from_numpy(s).to(dev), topn)
        ref = np.argsort(-s, axis=1, kind="stable")[:, :min(n, topn)]
        assert np.array_equal(order.cpu().numpy(), ref)
        assert np.array_equal(ss.cpu().numpy(), np.take_along_axis(s, ref, 1))


def test_decode_clip_golden(G, dev):
    ops = _ops()
    deltas = torch.from_numpy(G["dec_deltas"]).to(dev)  # [2, K*A, 4]
    im_info = torch.from_numpy(G["dec_im_info"]).to(dev)
    anchors = torch.from_numpy(G["anchors_a4"]).float().to(dev)
    B, n, _ = deltas.shape
    A, H, W = 12, 5, 7
    cls = torch.zeros(B, n // A, 2 * A, device=dev)
    bbox = deltas.reshape(B, H * W, A * 4).contiguous()
    props, scores = ops.rpn_decode(cls, (H * W * 2 * A, 1, 2 * A), False, bbox, (H * W * 4 * A, 1, 4 * A), im_info, anchors,
                                   B, A, H, W, 16)
    assert np.allclose(props.cpu().numpy(), G["dec_out"], rtol=1e-5, atol=1e-3)  # expf vs torch.exp: few ulp
    assert torch.allclose(scores, torch.full_like(scores, 0.5))


def test_proposal_target_hip_matches_oracle_same_rng(dev):
    """HIP _ProposalTargetLayer (2 launches + 1 count read) vs oracle.model_ref.proposal_target_layer (the pinned
    restatement of proposal_target_layer_cascade.py:33-213), same np.random stream"""
    import dana_amd
    from dana_amd import ops, synthetic as S
    from dana_amd.config import cfg
    from oracle import model_ref as O
    B = 3
    _, _, gt, _, _ = S.episode_inputs(B, 1, 1, 600, 1000, seed=21)
    rng = np.random.default_rng(4)
    rois = torch.zeros(B, 2000, 5)
    xy = rng.uniform(0, 800, size=(B, 2000, 2)); wh = rng.uniform(8, 300, size=(B, 2000, 2))
    rois[:, :, 1:3] = torch.from_numpy(xy).float()
    rois[:, :, 3] = torch.from_numpy(np.minimum(xy[..., 0] + wh[..., 0], 999)).float()
    rois[:, :, 4] = torch.from_numpy(np.minimum(xy[..., 1] + wh[..., 1], 599)).float()
    rois[:, :40, 1:] = gt[:, :1, :4] + torch.from_numpy(rng.uniform(-6, 6, size=(B, 40, 4))).float()  # some fg
    rois[:, 1900:] = 0  # zero padding
    for b in range(B):
        rois[b, :, 0] = b
    tr = cfg.TRAIN
    np.random.seed(5)
    ref = O.proposal_target_layer(rois, gt)
    np.random.seed(5)
    got = ops.proposal_target_layer(rois.to(dev), gt.to(dev), 128, 32, tr.FG_THRESH, tr.BG_THRESH_HI, tr.BG_THRESH_LO,
                                    tr.BBOX_NORMALIZE_MEANS, tr.BBOX_NORMALIZE_STDS, tr.BBOX_INSIDE_WEIGHTS, True)
    assert torch.equal(got[0].cpu(), ref[0])            # sampled rois: identical picks
    assert torch.equal(got[1].cpu(), ref[1])            # labels
    assert torch.allclose(got[2].cpu(), ref[2], atol=2e-6)  # logf vs torch.log
    assert torch.equal(got[3].cpu(), ref[3]) and torch.equal(got[4].cpu(), ref[4])


def test_conv_dual_segment_matches_two_single_launches(dev):
    """query batch + support batch in one launch (dana_conv2d_nhwc_dual) == two separate launches, bitwise"""
    import dana_amd
    from dana_amd import ops
    torch.manual_seed(8)
    n0, h0, w0, n1, h1, w1, cin, cout = 2, 19, 32, 5, 20, 20, 64, 96
    x0, x1 = torch.randn(n0 * h0 * w0, cin, device=dev), torch.randn(n1 * h1 * w1, cin, device=dev)
    x = torch.cat([x0, x1], 0).contiguous()
    for k, stride in ((3, 1), (1, 2), (1, 1)):
        w = torch.randn(cout, k * k * cin, device=dev) * 0.05
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        a, oh0, ow0 = ops.conv2d_nhwc(x0, n0, h0, w0, cin, w, cout, k, k, stride, k // 2, scale=sc, shift=sh, relu=True)
        b, oh1, ow1 = ops.conv2d_nhwc(x1, n1, h1, w1, cin, w, cout, k, k, stride, k // 2, scale=sc, shift=sh, relu=True)
        m, _, g0, g1 = ops.conv2d_nhwc_dual(x, n0, h0, w0, n1, h1, w1, cin, w, cout, k, k, stride, k // 2, scale=sc,
                                            shift=sh, relu=True)
        assert g0 == (oh0, ow0) and g1 == (oh1, ow1)
        assert torch.equal(m[:a.size(0)], a) and torch.equal(m[a.size(0):], b)


def test_anchor_targets_and_rpn_losses_hip_match_oracle_same_rng(dev):
    """HIP _AnchorTargetLayer + fused RPN losses vs oracle.model_ref.anchor_target_layer (the pinned restatement of
    anchor_target_layer.py:48-193) and the oracle's loss formulation (rpn.py:97-115), same np.random stream"""
    import torch.nn.functional as F
    import dana_amd
    from dana_amd import ops, synthetic as S, targets as T
    from dana_amd.config import cfg
    from oracle import model_ref as O
    B, H, W = 3, 38, 63
    _, im_info, gt, _, _ = S.episode_inputs(B, 1, 1, 600, 1000, seed=33)
    base = torch.from_numpy(T.generate_anchors(scales=np.array(cfg.ANCHOR_SCALES), ratios=np.array(cfg.ANCHOR_RATIOS))).float()
    tr = cfg.TRAIN
    np.random.seed(9)
    ref = O.anchor_target_layer((H, W), gt, im_info)
    np.random.seed(9)
    h = ops.anchor_target_assign(gt.to(dev), im_info.to(dev), base.to(dev), H, W, 16, tr.RPN_NEGATIVE_OVERLAP,
                                 tr.RPN_POSITIVE_OVERLAP, tr.RPN_BATCHSIZE, tr.RPN_FG_FRACTION)
    got = ops.anchor_target_outputs(h)
    assert torch.equal(got[0].cpu(), ref[0])                       # labels: same anchors kept / disabled
    assert torch.allclose(got[1].cpu(), ref[1], atol=2e-6)         # targets (logf vs torch.log)
    assert torch.equal(got[2].cpu(), ref[2]) and torch.equal(got[3].cpu(), ref[3])
    assert int((ref[0] == 1).sum()) > 0 and int((ref[0] == 0).sum()) > 0
    # fused losses vs rpn.py:97-115 in torch on the same random head outputs
    A = 12
    heads = torch.randn(B * H * W, 6 * A) * 0.5
    cls = heads[:, :2 * A].view(B, H, W, 2 * A).permute(0, 3, 1, 2)
    bbox = heads[:, 2 * A:].view(B, H, W, 4 * A).permute(0, 3, 1, 2)
    sc = cls.reshape(B, 2, A * H, W).permute(0, 2, 3, 1).reshape(-1, 2)
    lab = ref[0].view(-1)
    keep = lab.ne(-1).nonzero().view(-1)
    l_cls = F.cross_entropy(sc[keep], lab[keep].long())
    l_box = O.smooth_l1(bbox, ref[1], ref[2], ref[3], sigma=3, dims=(1, 2, 3))
    out = ops.rpn_losses(heads.to(dev), 6 * A, h, sigma=3.0).cpu()
    assert abs(float(out[0]) - float(l_cls)) <= 1e-5 * max(1.0, abs(float(l_cls)))
    assert abs(float(out[1]) - float(l_box)) <= 1e-5 * max(1.0, abs(float(l_box)))


def test_rcnn_losses_fused_vs_reference_formulation(dev):
    """dana.py:199-217: smooth-L1 + hard-negative-mined 2-way cross-entropy, and their gradient seeds vs autograd"""
    import torch.nn.functional as F
    from dana_amd import ops, targets as T
    torch.manual_seed(8)
    for n, nfg in [(256, 40), (512, 0), (128, 100), (64, 1)]:
        sp = (torch.randn(n, 2) * 2).double().requires_grad_(True)
        sn = (torch.randn(n, 2) * 2).double().requires_grad_(True)
        lab = torch.zeros(n)
        lab[torch.randperm(n)[:nfg]] = 1
        pred = torch.randn(n, 4).double().requires_grad_(True)
        tgt = torch.randn(n, 4) * 0.5
        w_in = (lab.view(-1, 1) * torch.ones(1, 4)).contiguous()
        w_out = w_in.clone()
        # reference formulation (dana.py:199-217)
        score = torch.cat([sp, sn], 0)
        label = torch.cat([lab, torch.zeros(n)], 0).long()
        fg = (label == 1).nonzero().squeeze(-1)
        bg = (label == 0).nonzero().squeeze(-1)
        soft = F.softmax(score, 1)[bg, :]
        n_all = label.shape[0]
        b0 = max(1, min(fg.shape[0] * 2, int(n_all * 0.25)))
        b1 = max(1, min(fg.shape[0], b0))
        _, order = torch.sort(soft[:, 1], descending=True, stable=True)
        real = bg[order]
        top0 = real[real < int(n_all * 0.5)][:b0]
        top1 = real[real >= int(n_all * 0.5)][:b1]
        idx = torch.cat([fg, top0, top1], 0)
        l_cls = F.cross_entropy(score[idx], label[idx])
        l_box = T._smooth_l1_loss(pred, tgt.double(), w_in.double(), w_out.double())
        (l_cls + l_box).backward()
        out, seeds = ops.rcnn_losses(sp.detach().float().to(dev), sn.detach().float().to(dev), lab.to(dev),
                                     pred.detach().float().to(dev), tgt.to(dev), w_in.to(dev), w_out.to(dev),
                                     with_grad=True)
        o = out.cpu()
        assert int(o[2]) == idx.numel()
        assert abs(float(o[0]) - float(l_cls.detach())) <= 2e-6 * max(1.0, abs(float(l_cls.detach())))
        assert abs(float(o[1]) - float(l_box.detach())) <= 2e-6 * max(1.0, abs(float(l_box.detach())))
        assert (seeds[0].cpu().double() - sp.grad).abs().max() <= 1e-7
        assert (seeds[1].cpu().double() - sn.grad).abs().max() <= 1e-7
        assert (seeds[2].cpu().double() - pred.grad).abs().max() <= 1e-7


def _anchor_prepare(dev, gt, im_info, anchors, H, W):
    from dana_amd import ops
    from dana_amd._lib import lib
    from dana_amd.config import cfg
    B, n_gt, _ = gt.shape
    A = anchors.shape[0]
    total = H * W * A
    labels = torch.empty((B, total), dtype=torch.float32, device=dev)
    max_ov = torch.empty((B, total), dtype=torch.float32, device=dev)
    ibuf = torch.empty((3, B, total), dtype=torch.int32, device=dev)
    counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
    lib().call("dana_anchor_target_prepare", gt.data_ptr(), im_info.data_ptr(), anchors.data_ptr(), B, A, H, W, 16, n_gt,
               float(cfg.TRAIN.RPN_NEGATIVE_OVERLAP), float(cfg.TRAIN.RPN_POSITIVE_OVERLAP), labels.data_ptr(),
               max_ov.data_ptr(), ibuf[0].data_ptr(), ibuf[1].data_ptr(), ibuf[2].data_ptr(), counts.data_ptr(),
               torch.cuda.current_stream().cuda_stream)
    return labels, counts


def test_device_rng_target_sampling_is_valid_deterministic_and_uniform(dev):
    """opt-in Philox sampling of the two target layers (no host sync): same distributions as the reference's
    np.random draws (anchor_target_layer.py:137-156, proposal_target_layer_cascade.py:143-175), checked by their
    invariants, by determinism in (seed, offset) and by first moments"""
    from dana_amd import ops, targets as T
    from dana_amd.config import cfg
    np.random.seed(2)
    B, H, W, n_gt = 3, 20, 30, 8
    gt = torch.zeros(B, n_gt, 5)
    for b in range(B):
        for k in range(2 + 2 * b):
            x1, y1 = np.random.uniform(0, 300), np.random.uniform(0, 180)
            gt[b, k] = torch.tensor([x1, y1, x1 + np.random.uniform(40, 160), y1 + np.random.uniform(40, 130), 1.0])
    im_info = torch.tensor([[H * 16.0, W * 16.0, 1.0]] * B)
    anchors = torch.from_numpy(T.generate_anchors(scales=np.array(cfg.ANCHOR_SCALES),
                                                  ratios=np.array(cfg.ANCHOR_RATIOS))).float()
    gt_d, ii_d, an_d = gt.to(dev), im_info.to(dev), anchors.to(dev)
    pre, counts = _anchor_prepare(dev, gt_d, ii_d, an_d, H, W)
    pre, cnt = pre.cpu().numpy(), counts.cpu().numpy()
    tr = cfg.TRAIN
    num_fg = int(tr.RPN_FG_FRACTION * tr.RPN_BATCHSIZE)

    def run(seed, off):
        h = ops.anchor_target_assign(gt_d, ii_d, an_d, H, W, 16, tr.RPN_NEGATIVE_OVERLAP, tr.RPN_POSITIVE_OVERLAP,
                                     tr.RPN_BATCHSIZE, tr.RPN_FG_FRACTION, device_rng=(seed, off))
        return h["labels"].cpu().numpy(), float(h["inv_ne_dev"].cpu())

    lab, inv = run(7, 0)
    for b in range(B):
        nf, nb = int(cnt[b, 0]), int(cnt[b, 1])
        assert nf == int((pre[b] == 1).sum()) and nb == int((pre[b] == 0).sum())
        keep_f = min(nf, num_fg)
        keep_b = min(nb, tr.RPN_BATCHSIZE - keep_f)
        assert int((lab[b] == 1).sum()) == keep_f and int((lab[b] == 0).sum()) == keep_b
        changed = lab[b] != pre[b]
        assert np.all(lab[b][changed] == -1) and np.all(pre[b][changed] >= 0)
        if b == B - 1:
            assert abs(inv - 1.0 / (keep_f + keep_b)) < 1e-9
    assert int(cnt[:, 1].max()) > tr.RPN_BATCHSIZE  # the subsampling really had something to drop
    lab2, _ = run(7, 0)
    lab3, _ = run(7, 1)
    lab4, _ = run(8, 0)
    assert np.array_equal(lab, lab2) and not np.array_equal(lab, lab3) and not np.array_equal(lab, lab4)
    # uniformity: every background anchor of image 0 survives with probability keep_b / nb
    nb0 = int(cnt[0, 1])
    keep0 = min(nb0, tr.RPN_BATCHSIZE - min(int(cnt[0, 0]), num_fg))
    hits = np.zeros(pre.shape[1])
    trials = 300
    for t in range(trials):
        hits += run(11, 100 + t)[0][0] == 0
    p = keep0 / nb0
    freq = hits[pre[0] == 0] / trials
    assert abs(freq.mean() - p) < 1e-9  # exactly keep0 survivors per draw
    assert np.abs(freq - p).max() < 6 * np.sqrt(p * (1 - p) / trials) + 1e-3
    # a rpn loss computed from the device handle works and is finite
    heads = torch.randn(B * H * W, 6 * anchors.shape[0], device=dev) * 0.5
    h = ops.anchor_target_assign(gt_d, ii_d, an_d, H, W, 16, tr.RPN_NEGATIVE_OVERLAP, tr.RPN_POSITIVE_OVERLAP,
                                 tr.RPN_BATCHSIZE, tr.RPN_FG_FRACTION, device_rng=(7, 0))
    l3 = ops.rpn_losses(heads, 6 * anchors.shape[0], h, sigma=3.0).cpu()
    assert torch.isfinite(l3).all() and int(l3[2]) == int((lab >= 0).sum())

    # ---- proposal targets ----
    from dana_amd._lib import lib
    n_rois = 400
    rois = torch.zeros(B, n_rois, 5)
    for b in range(B):
        rois[b, :, 0] = b
        x1 = torch.rand(n_rois) * 350
        y1 = torch.rand(n_rois) * 200
        rois[b, :, 1], rois[b, :, 2] = x1, y1
        rois[b, :, 3], rois[b, :, 4] = x1 + 30 + torch.rand(n_rois) * 120, y1 + 30 + torch.rand(n_rois) * 100
        rois[b, :40, 1:] = gt[b, b % 2, :4] + torch.randn(40, 4) * 3  # some foreground candidates
    rois_d = rois.to(dev)
    R, fg_per = 128, 32
    args = (rois_d, gt_d, R, fg_per, tr.FG_THRESH, tr.BG_THRESH_HI, tr.BG_THRESH_LO, tr.BBOX_NORMALIZE_MEANS,
            tr.BBOX_NORMALIZE_STDS, tr.BBOX_INSIDE_WEIGHTS, True)
    o1 = ops.proposal_target_layer(*args, device_rng=(5, 0))
    o2 = ops.proposal_target_layer(*args, device_rng=(5, 0))
    o3 = ops.proposal_target_layer(*args, device_rng=(5, 2))
    assert all(torch.equal(a, b_) for a, b_ in zip(o1, o2)) and not torch.equal(o1[0], o3[0])
    rois_out, labels, tgt, w_in, w_out = [t.cpu() for t in o1]
    cand = torch.cat([rois[:, :, 1:], gt[:, :, :4]], 1)  # the layer appends the gt boxes to the proposals (:47-52)
    for b in range(B):
        nfg = int((labels[b] == 1).sum())
        assert 0 < nfg <= fg_per and torch.all(labels[b, :nfg] == 1) and torch.all(labels[b, nfg:] == 0)
        assert torch.all(rois_out[b, :, 0] == b)
        d = (rois_out[b, :, None, 1:] - cand[b][None]).abs().sum(-1).min(1).values
        assert torch.all(d == 0)  # every sampled roi is one of the candidates
        fgb = rois_out[b, :nfg, 1:]
        assert len({tuple(r.tolist()) for r in fgb}) == nfg  # foreground picks are drawn without replacement
        assert torch.all((w_in[b] > 0).any(1) == (labels[b] == 1))


def test_two_threads_two_streams_are_reentrant(dev):
    """include/dana_hip.h:7-9 promises re-entrancy (nn.DataParallel drives one replica per THREAD, train.py:104-105):
    two Python threads, each on its own stream of the one device, hammer `_C.nms`, `_C.roi_align_forward` and a
    contraction concurrently; every result must equal the single-threaded one bit for bit."""
    import threading
    from dana_amd import _C, ops
    rng = np.random.RandomState(5)

    def boxes(n, seed):
        r = np.random.RandomState(seed)
        xy = r.rand(n, 2).astype(np.float32) * 400
        wh = r.rand(n, 2).astype(np.float32) * 120 + 4
        return torch.from_numpy(np.concatenate([xy, xy + wh], 1)), torch.from_numpy(r.rand(n).astype(np.float32))

    work = []
    for t in range(2):
        d, s = boxes(3000 + 700 * t, 10 + t)
        feat = torch.from_numpy(rng.randn(2, 64, 38, 50).astype(np.float32))
        rois = torch.from_numpy(np.concatenate([rng.randint(0, 2, (200, 1)).astype(np.float32),
                                                boxes(200, 20 + t)[0]], 1))
        x = torch.from_numpy(rng.randn(2 * 30 * 40, 128).astype(np.float32))
        w = torch.from_numpy((rng.randn(256, 128, 3, 3) * 0.05).astype(np.float32))
        work.append([v.to(dev) for v in (d, s, feat, rois, x, w)])

    def run(item):
        d, s, feat, rois, x, w = item
        keep = _C.nms(d, s, 0.7)
        pooled = _C.roi_align_forward(feat, rois, 1.0 / 16.0, 7, 7, 0)
        y, _, _ = ops.conv2d_nhwc(x, 2, 30, 40, 128, ops.pack_conv_weight(w), 256, 3, 3, 1, 1, relu=True)
        return keep, pooled, y

    expect = [run(it) for it in work]
    torch.cuda.synchronize()
    errors, out = [], [[None] * 6 for _ in range(2)]

    def worker(t):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                for i in range(6):
                    out[t][i] = run(work[t])
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    assert not errors, errors
    for t in range(2):
        for i in range(6):
            for a, b in zip(out[t][i], expect[t]):
                assert torch.equal(a, b), "thread %d round %d differs from the single-threaded result" % (t, i)


def test_plain_rcnn_losses_vs_torch_and_out_of_range_label_is_nan(dev):
    """dana_plain_rcnn_loss (faster_rcnn.py:93-98) vs F.cross_entropy / the smooth-L1 formulation, with its seeds vs
    autograd; a label outside 0..C-1 never indexes the score row: loss and that row's gradient are NaN"""
    import torch.nn.functional as F
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    n, C = 200, 2
    sc = torch.randn(n, C, generator=g)
    lab = torch.randint(0, C, (n,), generator=g)
    bp, tg = torch.randn(n, 4, generator=g), torch.randn(n, 4, generator=g)
    win = (torch.rand(n, 4, generator=g) > 0.5).float()
    wout = win.clone()
    scr = sc.clone().requires_grad_(True)
    ref = F.cross_entropy(scr, lab)
    ref.backward()
    l2, (dc, db) = ops.plain_rcnn_losses(sc.to(dev), lab.to(dev), bp.to(dev), tg.to(dev), win.to(dev), wout.to(dev),
                                         with_grad=True)
    assert abs(float(l2[0]) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert (dc.cpu() - scr.grad).abs().max() <= 1e-7
    d = win * (bp - tg)
    sl1 = (wout * torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5)).sum(1).mean()
    assert abs(float(l2[1]) - float(sl1)) <= 2e-6 * max(1.0, abs(float(sl1)))
    bad = lab.clone()
    bad[17] = C  # one past the last class
    l2b, (dcb, _) = ops.plain_rcnn_losses(sc.to(dev), bad.to(dev), bp.to(dev), tg.to(dev), win.to(dev), wout.to(dev),
                                          with_grad=True)
    assert torch.isnan(l2b[0]) and torch.isnan(dcb[17]).all()
    keep = torch.ones(n, dtype=torch.bool)
    keep[17] = False
    assert torch.equal(dcb.cpu()[keep], dc.cpu()[keep]) and torch.isfinite(l2b[1])
