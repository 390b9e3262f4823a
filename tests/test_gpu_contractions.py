"""fp32 implicit-GEMM conv / GEMM and the small attention kernels vs a plain PyTorch fp32
CPU reference of the same op. Tolerance: fp32 roundoff with a different summation order,
|d| <= 2e-5 * sum|a*b| bound -> checked as rtol 1e-4 on the output scale.
Every test of this file runs twice: with the contractions on the bf16 matrix cores (exact 3-way split of the fp32
operands, six products, fp32 accumulation -- the default) and on the f32 MFMA (dana_set_mfma_mode)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    import dana_amd
    return dana_amd.ops


@pytest.fixture(autouse=True, params=[1, 0], ids=["bf16x6", "f32mfma"])
def mfma_mode(request):
    ops = _ops()
    prev = ops.set_mfma_mode(request.param)
    yield request.param
    ops.set_mfma_mode(prev)


def _close(a, b, tol=1e-4):
    a, b = a.double(), b.double()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, "max err %.3e vs scale %.3e" % (err, scale)


CONV_CASES = [
    # (N, H, W, Cin, Cout, k, stride, pad, relu, residual)
    (2, 19, 23, 64, 64, 1, 1, 0, True, False),
    (2, 19, 23, 64, 256, 1, 1, 0, True, True),
    (1, 38, 63, 256, 256, 3, 1, 1, True, False),
    (2, 37, 25, 256, 128, 1, 2, 0, True, False),     # strided 1x1 (Caffe bottleneck, resnet.py:71)
    (3, 7, 7, 1024, 512, 1, 2, 0, True, False),      # layer4 entry on pooled RoIs 7x7 -> 4x4
    (1, 12, 16, 2048, 512, 3, 1, 1, True, False),    # RPN conv (bias + relu)
    (2, 20, 20, 512, 1024, 1, 1, 0, False, False),   # downsample-like (no relu)
    (1, 9, 11, 128, 96, 3, 1, 1, False, True),       # N not a tile multiple
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_nhwc_vs_torch(dev, case):
    ops = _ops()
    N, H, W, Cin, Cout, k, stride, pad, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = torch.randn_like(ref) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    xd = ops.nchw_to_nhwc(x.to(dev))
    wp = ops.pack_conv_weight(w.to(dev))
    resd = ops.nchw_to_nhwc(res.to(dev)).view(-1, Cout) if use_res else None
    out, oh, ow = ops.conv2d_nhwc(xd, N, H, W, Cin, wp, Cout, k, k, stride, pad, scale=scale.to(dev),
                                  shift=shift.to(dev), residual=resd, relu=relu)
    got = ops.nhwc_to_nchw(out, N, Cout, oh, ow).cpu()
    assert got.shape == ref.shape
    _close(got, ref)


def test_conv_strided_io_inside_concat_buffer(dev):
    """input read with a pixel stride and output written with a row stride (concat-buffer fusion)"""
    ops = _ops()
    torch.manual_seed(1)
    N, H, W, Cin, Cout = 1, 10, 13, 64, 96
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cout, Cin, 1, 1) / 8
    ref = F.conv2d(x, w)
    buf = torch.full((N * H * W, 160), 3.0, device=dev)
    buf[:, :Cin] = ops.nchw_to_nhwc(x.to(dev)).view(-1, Cin)
    out = torch.full((N * H * W, 256), -5.0, device=dev)
    ops.conv2d_nhwc(buf, N, H, W, Cin, ops.pack_conv_weight(w.to(dev)), Cout, 1, 1, 1, 0, in_stride=160,
                    out=out.view(-1)[128:], out_stride=256)
    _close(out[:, 128:128 + Cout].cpu(), ref.permute(0, 2, 3, 1).reshape(-1, Cout))
    assert (out[:, :128] == -5).all() and (out[:, 128 + Cout:] == -5).all()


@pytest.mark.parametrize("hw", [(64, 80), (75, 75), (33, 47)])
def test_stem_conv_and_ceil_maxpool_vs_torch(dev, hw):
    """7x7/2 stem on NHWC4 + BN + ReLU, then MaxPool2d(3,2,0,ceil_mode=True) (resnet.py:109-113)"""
    ops = _ops()
    H, W = hw
    torch.manual_seed(2)
    x = torch.randn(2, 3, H, W) * 64
    w = torch.randn(64, 3, 7, 7) * 0.025
    scale, shift = torch.rand(64) * 0.05 + 0.02, torch.randn(64) * 0.1
    y = F.relu(F.conv2d(x, w, stride=2, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    ref = F.max_pool2d(y, 3, 2, 0, ceil_mode=True)
    x4 = ops.nchw_to_nhwc(x.to(dev), cpad=4)
    wp = ops.pack_conv_weight(w.to(dev), stem=True)
    o, oh, ow = ops.conv2d_nhwc(x4, 2, H, W, 4, wp, 64, 7, 7, 2, 3, scale=scale.to(dev), shift=shift.to(dev),
                                relu=True, stem=True)
    _close(ops.nhwc_to_nchw(o, 2, 64, oh, ow).cpu(), y)
    p, ph, pw = ops.maxpool3x3s2_ceil(o, 2, oh, ow, 64)
    assert (ph, pw) == tuple(ref.shape[2:])
    _close(ops.nhwc_to_nchw(p, 2, 64, ph, pw).cpu(), ref)


@pytest.mark.parametrize("m,n,k", [(300, 256, 1024), (2394, 1200, 256), (77, 2, 1024), (128, 1024, 3136),
                                   (513, 1024, 1200), (49, 147, 256), (1, 4, 2048)])
def test_gemm_nt_vs_torch(dev, m, n, k):
    ops = _ops()
    g = torch.Generator().manual_seed(m * 7 + n)
    a, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g)
    bias = torch.randn(n, generator=g)
    ref = (a.double() @ b.double().t()) * 0.125 + bias.double()
    out = ops.gemm_nt(a.to(dev), b.to(dev), m, n, k, shift=bias.to(dev), alpha=0.125)
    _close(out.cpu(), ref, 2e-5)


def test_gemm_nt_batched_strided(dev):
    ops = _ops()
    torch.manual_seed(4)
    B, m, n, k = 3, 130, 147, 256
    a, b = torch.randn(B, m, k), torch.randn(B, 2 * n, k)  # use the second half of each b batch
    ref = torch.bmm(a.double(), b[:, n:].double().transpose(1, 2)) / 16
    out = torch.zeros(B, m, 160, device=dev)
    bd = b.to(dev)
    ops.gemm_nt(a.to(dev), bd.view(-1)[n * k:], m, n, k, out=out, ldc=160, batch=B, batch_a=m * k, batch_b=2 * n * k,
                batch_c=m * 160, alpha=1 / 16)
    _close(out[:, :, :n].cpu(), ref, 2e-5)
    assert (out[:, :, n:] == 0).all()


def test_attention_small_kernels_vs_torch(dev):
    ops = _ops()
    torch.manual_seed(5)
    G_, L, D = 6, 400, 1024
    s = torch.randn(G_, L, D)
    w, b = torch.randn(1, D) * 0.03, torch.randn(1) * 0.1
    pe = torch.randn(L, D)
    sd = ops.add_pe(s.to(dev), pe.to(dev), G_ * L, L, D).view(G_, L, D)
    s_pe = s + pe
    assert torch.equal(sd.cpu(), s_pe)
    logits = ops.rowdot(sd, w.to(dev), b.to(dev), G_ * L, D)
    ref_logits = F.linear(s_pe, w, b).squeeze(-1)
    _close(logits.view(G_, L).cpu(), ref_logits, 2e-5)
    sm = ops.softmax_rows_(logits.clone(), G_, L).view(G_, L)
    ref_sm = F.softmax(ref_logits, 1)
    assert torch.allclose(sm.cpu(), ref_sm, rtol=1e-4, atol=1e-7)
    # BA block (dana.py:134-137)
    g = torch.bmm(ref_sm.unsqueeze(1), s_pe)
    ref_ba = s_pe + 0.1 * F.leaky_relu(g)
    ba = ops.ba_apply_(sd.clone(), sm.contiguous(), G_, L, D)
    _close(ba.cpu(), ref_ba, 1e-5)
    # zero-mean over positions (dana.py:125)
    q = torch.randn(2, 777, 256)
    qd = ops.colmean_sub_(q.to(dev).clone(), 2, 777, 256)
    _close(qd.cpu(), q - q.mean(1, keepdim=True), 1e-5)
    # softmax + unary + 1/shot over per-shot segments (dana.py:143-146,150)
    rows, nseg, Ls = 50, 3, 49
    sc = torch.randn(2, rows, 160)
    un = F.softmax(torch.randn(2, 2 * nseg, Ls), 2)  # way=2 layout: use the second `nseg` block of each batch
    ref = torch.cat([(F.softmax(sc[:, :, i * Ls:(i + 1) * Ls], 2) + 0.1 * un[:, nseg + i].unsqueeze(1)) / nseg
                     for i in range(nseg)], 2)
    scd = sc.to(dev).clone()
    ops.attn_softmax_unary_(scd, un.to(dev).view(-1)[nseg * Ls:], 2 * rows, rows, nseg, Ls, 160, 160, 0.1, 1.0 / nseg,
                            unary_batch_stride=2 * nseg * Ls)
    assert torch.allclose(scd[:, :, :nseg * Ls].cpu(), ref, rtol=1e-4, atol=1e-7)
    assert (scd[:, :, nseg * Ls:] == 0).all()
    # transpose with zero K padding, avgpool 14/1, spatial mean
    t = ops.transpose_batched(sd[:2].contiguous(), 2, L, D, ldo=416)
    assert torch.equal(t[:, :, :L].cpu(), s_pe[:2].transpose(1, 2)) and (t[:, :, L:] == 0).all()
    f = torch.randn(3, 1024, 20, 20)
    ap = ops.avgpool(ops.nchw_to_nhwc(f.to(dev)), 3, 20, 20, 1024, 14, 1)
    _close(ap.view(3, 7, 7, 1024).permute(0, 3, 1, 2).cpu(), F.avg_pool2d(f, 14, 1), 1e-5)
    y = torch.randn(5, 2048, 4, 4)
    mean = ops.spatial_mean(ops.nchw_to_nhwc(y.to(dev)), 5, 16, 2048)
    _close(mean.cpu(), y.mean(3).mean(2), 1e-5)
    # frozen-BN fold
    gam, bet, mu, var = torch.rand(64) + 0.5, torch.randn(64), torch.randn(64), torch.rand(64) + 0.5
    sc_, sh_ = ops.bn_fold(gam.to(dev), bet.to(dev), mu.to(dev), var.to(dev), 1e-5)
    x = torch.randn(10, 64)
    _close(x * sc_.cpu() + sh_.cpu(), F.batch_norm(x, mu, var, gam, bet, False, 0., 1e-5), 1e-5)


@pytest.mark.parametrize("tile", [2, 4])
@pytest.mark.parametrize("case", [(2, 38, 63, 256, 256), (3, 4, 4, 512, 512), (1, 12, 16, 2048, 512), (2, 7, 9, 64, 32),
                                  (1, 5, 6, 64, 64), (2, 1, 3, 32, 32)])
def test_winograd_3x3_vs_torch(dev, case, tile):
    """F(2x2,3x3) and F(4x4,3x3) paths (sizes that are not multiples of the tile exercise the clipped last tile
    row/column) vs torch fp32 conv"""
    ops = _ops()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    xd = ops.nchw_to_nhwc(x.to(dev))
    u = ops.winograd_filter_transform(ops.pack_conv_weight(w.to(dev)), Cout, Cin, tile)
    assert u.size(0) == (tile + 2) ** 2
    out = torch.full((N * H * W, Cout + 8), -7.0, device=dev)
    ops.conv3x3_winograd(xd, N, H, W, Cin, u, Cout, scale=scale.to(dev), shift=shift.to(dev), relu=True, out=out,
                         out_stride=Cout + 8)
    assert (out[:, Cout:] == -7).all()
    got = ops.nhwc_to_nchw(out, N, Cout, H, W, in_stride=Cout + 8).cpu()
    _close(got, ref)


@pytest.mark.parametrize("case", [(2, 19, 23, 64, 64, 3, 1, 1), (2, 20, 20, 256, 128, 1, 2, 0), (1, 38, 63, 256, 256, 3, 1, 1),
                                  (3, 7, 7, 1024, 512, 1, 2, 0), (2, 12, 16, 128, 512, 1, 1, 0)])
def test_conv_backward_blocks_vs_torch_autograd(dev, case):
    """weight gradient (split-M TN contraction on the MFMA) and data gradient vs torch autograd in fp64"""
    ops = _ops()
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    oh, ow = y.shape[2:]
    xd = ops.nchw_to_nhwc(x.detach().float().to(dev))
    gyd = ops.nchw_to_nhwc(gy.float().to(dev)).view(-1, Cout)
    wp = ops.pack_conv_weight(w.detach().float().to(dev))
    dw = ops.conv2d_wgrad(gyd, xd, N, H, W, Cin, Cout, k, k, stride, pad)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)  # packed layout [cout][kh][kw][cin]
    _close(dw.cpu(), ref_dw, 2e-5)
    dx = ops.conv2d_dgrad(gyd, wp, N, H, W, Cin, Cout, k, k, stride, pad)
    ref_dx = x.grad.permute(0, 2, 3, 1).reshape(-1, Cin)
    _close(dx.cpu(), ref_dx, 2e-5)
    # accumulate=True adds onto an existing gradient buffer
    acc = dw.clone()
    ops.conv2d_wgrad(gyd, xd, N, H, W, Cin, Cout, k, k, stride, pad, out=acc)
    _close(acc.cpu(), 2 * ref_dw, 2e-5)


@pytest.mark.parametrize("case", [(2, 38, 63, 256, 256), (16, 4, 4, 512, 128), (1, 12, 17, 128, 64), (2, 5, 3, 64, 72)])
def test_winograd_weight_gradient_vs_direct_and_autograd(dev, case):
    """F(4x4,3x3)-domain weight gradient (with row scale + accumulate) vs the direct split-M kernel and float64 autograd"""
    ops = _ops()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    gy = torch.randn(N, Cout, H, W, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5
    init = torch.randn(Cout, 9 * Cin, generator=g)
    wd = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wd, padding=1).backward(gy.double())
    ref = wd.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin) * scale.double().view(-1, 1) + init.double()
    xd, gd = ops.nchw_to_nhwc(x.to(dev)), ops.nchw_to_nhwc(gy.to(dev))
    out = init.clone().to(dev)
    ops.conv3x3_wgrad_winograd(gd, xd, N, H, W, Cin, Cout, out=out, row_scale=scale.to(dev))
    _close(out.cpu(), ref, 2e-4)
    direct = ops.conv2d_wgrad(gd, xd, N, H, W, Cin, Cout, 3, 3, 1, 1)
    fresh = ops.conv3x3_wgrad_winograd(gd, xd, N, H, W, Cin, Cout)
    _close(fresh.cpu(), direct.cpu(), 2e-4)


@pytest.mark.parametrize("m,n,k,relu", [(1000, 256, 1024, False), (4096, 512, 4608, False), (777, 130, 36, False),
                                        (2048, 256, 2304, True), (98, 1024, 3136, True)])
def test_contraction_error_vs_fp64_is_at_the_fp32_level(dev, mfma_mode, m, n, k, relu):
    """both kernels against an fp64 contraction of the same fp32 operands: max |err| / (|a| . |b|) must stay at the level
    of a plain fp32 GEMM (rocBLAS on the same data) -- in particular for the bf16x6 split, whose dropped cross terms are
    below the rounding of the fp32 accumulation"""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(n, k, generator=g)
    if relu:
        a, b = torch.relu(a) * 3.0, b * 0.02
    else:  # wide dynamic range
        a, b = a * torch.exp(torch.randn(m, k, generator=g)), b * torch.exp(torch.randn(n, k, generator=g))
    a, b = a.to(dev), b.to(dev)
    c = ops.gemm_nt(a, b, m, n, k)
    ref = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    err = ((c.double() - ref).abs() / mag).max().item()
    err_blas = (((a @ b.t()).double() - ref).abs() / mag).max().item()
    assert err <= 2.0 * err_blas + 2e-7, (err, err_blas)
    assert ops.get_mfma_mode() == mfma_mode


@pytest.mark.parametrize("k0,k1,cout,stride,h,w", [(64, 64, 256, 1, 19, 23), (128, 256, 512, 2, 21, 30), (256, 512, 1024, 2, 11, 14)])
def test_two_segment_contraction_is_expand_plus_downsample(dev, k0, k1, cout, stride, h, w):
    """dana_conv1x1_cat2_nhwc (a bottleneck's 1x1 expand conv + its strided 1x1 downsample conv as one contraction over
    the concatenated channels, BN scales folded into the weights) vs torch fp32 of resnet.py:84-100's tail"""
    import torch.nn.functional as F
    from dana_amd import ops
    if ops.get_mfma_mode() == 0:
        pytest.skip("the two-segment K walk lives in the split kernel; with dana_set_mfma_mode(0) the model runs the two "
                    "convs separately")
    torch.manual_seed(4)
    n = 3
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    t = torch.randn(n, k0, oh, ow, device=dev)
    x = torch.randn(n, k1, h, w, device=dev)
    w3, wd = torch.randn(cout, k0, device=dev) * 0.05, torch.randn(cout, k1, device=dev) * 0.05
    s3, b3 = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    sd, bd = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    ref = F.relu(F.conv2d(t, w3.view(cout, k0, 1, 1)) * s3.view(1, -1, 1, 1) + b3.view(1, -1, 1, 1) +
                 F.conv2d(x, wd.view(cout, k1, 1, 1), stride=stride) * sd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1))
    w_cat, shift = ops.pack_cat2_weight(w3, s3, b3, k0, wd, sd, bd, k1, cout)
    t_nhwc = t.permute(0, 2, 3, 1).reshape(-1, k0).contiguous()
    x_nhwc = x.permute(0, 2, 3, 1).reshape(-1, k1).contiguous()
    out, oh2, ow2 = ops.conv1x1_cat2(t_nhwc, k0, x_nhwc, k1, n, h, w, stride, w_cat, shift, cout, relu=True)
    assert (oh2, ow2) == (oh, ow)
    got = out.view(n, oh, ow, cout).permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("m,n,k", [(512, 1024, 3136), (1176, 256, 1024), (300, 72, 512)])
def test_split_k_gemm_matches_single_chain_and_fp64(dev, m, n, k):
    """tile-starved GEMMs run as K slices + a fixed-order reduction (ops._splitk_slices): same epilogue, same numbers up
    to the fp32 summation order, deterministic run to run"""
    from dana_amd import ops
    torch.manual_seed(2)
    a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev) * 0.05
    sh, r = torch.randn(n, device=dev), torch.randn(m, n, device=dev)
    assert ops._splitk_slices(m, n, k, 1) > 1
    got = ops.gemm_nt(a, b, m, n, k, shift=sh, residual=r, ldr=n, relu=True)
    again = ops.gemm_nt(a, b, m, n, k, shift=sh, residual=r, ldr=n, relu=True)
    assert torch.equal(got, again)
    ops.SPLIT_K = False
    try:
        one = ops.gemm_nt(a, b, m, n, k, shift=sh, residual=r, ldr=n, relu=True)
    finally:
        ops.SPLIT_K = True
    ref = torch.relu(a.double() @ b.double().t() + sh.double() + r.double())
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale * (k ** 0.5)
    assert float((got - one).abs().max()) <= 2e-6 * scale * (k ** 0.5)


def test_fused_bottleneck_tail_equals_the_two_launches(dev, mfma_mode):
    """dana_bottleneck_tail_nhwc (resnet.py:92-100: conv2 3x3 -> bn2 -> ReLU -> conv3 1x1 -> bn3 -> + residual -> ReLU in one
    launch, conv2's tile handed over through LDS) returns the SAME BITS as the two conv launches, incl. a ragged last row
    tile, strided input / residual / output rows, no residual, no final ReLU."""
    ops = _ops()
    if mfma_mode == 0:
        return  # split kernel only
    g = torch.Generator().manual_seed(91)
    for (n, h, w, ci, co, ldx, ldo, res, relu) in [(2, 19, 23, 64, 256, 64, 256, True, True), (1, 37, 41, 64, 256, 96, 320, True, True),
                                                    (3, 16, 8, 128, 128, 128, 128, False, True), (1, 9, 15, 64, 64, 64, 64, True, False),
                                                    (4, 75, 125, 64, 256, 64, 256, True, True)]:
        m = n * h * w
        x = torch.randn(m, ldx, generator=g).to(dev)
        w2 = (torch.randn(64, 9 * ci, generator=g) / np.sqrt(9 * ci)).to(dev)
        w3 = (torch.randn(co, 64, generator=g) / 8).to(dev)
        s2, b2 = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
        s3, b3 = (torch.rand(co, generator=g) + 0.5).to(dev), torch.randn(co, generator=g).to(dev)
        r = torch.randn(m, ldo, generator=g).to(dev) if res else None
        o2, _, _ = ops.conv2d_nhwc(x, n, h, w, ci, w2, 64, 3, 3, 1, 1, scale=s2, shift=b2, relu=True, in_stride=ldx)
        ref = torch.zeros(m, ldo, device=dev)
        ops.conv2d_nhwc(o2, n, h, w, 64, w3, co, 1, 1, 1, 0, scale=s3, shift=b3, residual=r, relu=relu, out=ref, out_stride=ldo,
                        res_stride=ldo if res else 0)
        got = torch.full((m + 136, ldo), -7.0, device=dev)  # guard rows behind the last (ragged) row tile
        got[:m].zero_()
        ops.bottleneck_tail(x, n, h, w, ci, ops.split_weight(w2, 64, 9 * ci), s2, b2, ops.split_weight(w3, co, 64), s3, b3, co,
                            residual=r, relu=relu, in_stride=ldx, out=got, out_stride=ldo, res_stride=ldo if res else 0)
        assert torch.equal(ref, got[:m]), (n, h, w, ci, co, float((ref - got[:m]).abs().max()))
        assert bool((got[m:] == -7.0).all()), "rows past M were written"


def test_presplit_weights_are_bit_identical_to_fp32_weights(dev, mfma_mode):
    """DANA_W_SPLIT3 (ops.split_weight): the B operand split into its three bf16 planes once per weight version feeds the
    same six products as the in-loop split -> every entry point returns the SAME BITS as with fp32 weight rows. Also the
    dual-geometry (query + support batch) forms against two single launches."""
    ops = _ops()
    if mfma_mode == 0:
        assert ops.split_weight(torch.zeros(16, 16, device=dev), 16, 16) is None  # f32-MFMA kernel: fp32 weights only
        return
    g = torch.Generator().manual_seed(77)
    for (n, h, w, ci, co, k, st, pad) in [(2, 19, 23, 64, 256, 1, 1, 0), (1, 20, 24, 128, 96, 3, 1, 1), (2, 21, 25, 256, 128, 1, 2, 0),
                                          (3, 7, 7, 1024, 512, 1, 2, 0), (2, 9, 9, 64, 64, 3, 1, 1)]:
        x = torch.randn(n * h * w, ci, generator=g).to(dev)
        wt = (torch.randn(co, k * k * ci, generator=g) / np.sqrt(ci * k * k)).to(dev)
        sc, sh = (torch.rand(co, generator=g) + 0.5).to(dev), torch.randn(co, generator=g).to(dev)
        a, oh, ow = ops.conv2d_nhwc(x, n, h, w, ci, wt, co, k, k, st, pad, scale=sc, shift=sh, relu=True)
        w3 = ops.split_weight(wt, co, k * k * ci)
        b, _, _ = ops.conv2d_nhwc(x, n, h, w, ci, w3, co, k, k, st, pad, scale=sc, shift=sh, relu=True)
        assert torch.equal(a, b), (n, h, w, ci, co, k)
    # GEMM (incl. the split-K route, a K tail of 4 and an N tail), Linear-style column halves of one weight
    for (m, n, k) in [(300, 256, 1024), (512, 1024, 3136), (100, 72, 512), (77, 40, 36), (4800, 256, 1024)]:
        a_ = torch.randn(m, k, generator=g).to(dev)
        b_ = (torch.randn(n, k, generator=g) / np.sqrt(k)).to(dev)
        ref = ops.gemm_nt(a_, b_, m, n, k)
        got = ops.gemm_nt(a_, ops.split_weight(b_, n, k), m, n, k)
        assert torch.equal(ref, got), (m, n, k)
    wt2 = torch.randn(64, 2048, generator=g).to(dev) / 32
    a_ = torch.randn(500, 1024, generator=g).to(dev)
    ref = ops.gemm_nt(a_, wt2.view(-1)[1024:], 500, 64, 1024, ldb=2048)
    got = ops.gemm_nt(a_, ops.split_weight(wt2.view(-1)[1024:], 64, 1024, ldw=2048), 500, 64, 1024)
    assert torch.equal(ref, got)
    # Winograd F(4x4,3x3) with split filter planes; two image groups with one plane GEMM == two single launches
    n0, h0, w0, n1, h1, w1, ci, co = 2, 19, 31, 3, 10, 10, 128, 160
    x = torch.randn(n0 * h0 * w0 + n1 * h1 * w1, ci, generator=g).to(dev)
    wt = (torch.randn(co, 9 * ci, generator=g) / np.sqrt(9 * ci)).to(dev)
    sc, sh = (torch.rand(co, generator=g) + 0.5).to(dev), torch.randn(co, generator=g).to(dev)
    u = ops.winograd_filter_transform(wt, co, ci, 4)
    u3 = ops.split_weight(u, co, ci, batch=36)
    a0, _, _ = ops.conv3x3_winograd(x, n0, h0, w0, ci, u, co, scale=sc, shift=sh, relu=True)
    a1, _, _ = ops.conv3x3_winograd(x[n0 * h0 * w0:], n1, h1, w1, ci, u, co, scale=sc, shift=sh, relu=True)
    b0, _, _ = ops.conv3x3_winograd(x, n0, h0, w0, ci, u3, co, scale=sc, shift=sh, relu=True)
    assert torch.equal(a0, b0)
    for filt in (u, u3):
        d = ops.conv3x3_winograd_dual(x, n0, h0, w0, n1, h1, w1, ci, filt, co, scale=sc, shift=sh, relu=True)
        assert torch.equal(d[:n0 * h0 * w0], a0) and torch.equal(d[n0 * h0 * w0:], a1)
    # two-segment expand + strided downsample contraction over two image groups == two single launches
    k0, k1, co, s_ = 64, 256, 256, 2
    oh0, ow0, oh1, ow1 = (h0 - 1) // s_ + 1, (w0 - 1) // s_ + 1, (h1 - 1) // s_ + 1, (w1 - 1) // s_ + 1
    m0, m1 = n0 * oh0 * ow0, n1 * oh1 * ow1
    t = torch.randn(m0 + m1, k0, generator=g).to(dev)
    xin = torch.randn(n0 * h0 * w0 + n1 * h1 * w1, k1, generator=g).to(dev)
    wc = (torch.randn(co, k0 + k1, generator=g) / 16).to(dev)
    shc = torch.randn(co, generator=g).to(dev)
    r0, _, _ = ops.conv1x1_cat2(t, k0, xin, k1, n0, h0, w0, s_, wc, shc, co)
    r1, _, _ = ops.conv1x1_cat2(t[m0:], k0, xin[n0 * h0 * w0:], k1, n1, h1, w1, s_, wc, shc, co)
    for filt in (wc, ops.split_weight(wc, co, k0 + k1)):
        d0, d1, _, _ = ops.conv1x1_cat2_dual(t, k0, xin, k1, n0, h0, w0, n1, h1, w1, s_, filt, shc, co)
        assert torch.equal(d0[:m0], r0) and torch.equal(d1[:m1], r1)
    # the 7x7 stem over NHWC4 with its packed [64][7][8][4] weight
    im = torch.randn(2, 3, 64, 80, generator=g).to(dev)
    x4 = ops.nchw_to_nhwc(im, cpad=4)
    wst = ops.pack_conv_weight((torch.randn(64, 3, 7, 7, generator=g) / 12).to(dev), stem=True)
    a, _, _ = ops.conv2d_nhwc(x4, 2, 64, 80, 4, wst, 64, 7, 7, 2, 3, relu=True, stem=True)
    b, _, _ = ops.conv2d_nhwc(x4, 2, 64, 80, 4, ops.split_weight(wst, 64, wst.numel() // 64), 64, 7, 7, 2, 3, relu=True, stem=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(9576, 1024, 256, True, 1, True), (1000, 200, 100, True, 1, False), (2394, 1200, 256, False, 4, False),
                                   (640, 256, 256, False, 9, True), (300, 128, 64, False, 1, True), (25088, 128, 1024, True, 1, True),
                                   (77, 30, 64, True, 1, False)],
                         ids=["l3-expand+res", "ragged-fp32B", "bmm-b4", "planes-b9", "one-round", "roi-proj+res", "tiny-ragged"])
def test_register_epilogue_gives_the_lds_epilogues_bits(dev, shape):
    """the split kernel's epilogue on the accumulator registers (dword buffer accesses, no LDS C tile: the default) against
    the round-1..4 form through an LDS C tile (dana_set_epilogue_mode(1)): the same arithmetic in the same order -> the same
    bits, on full, ragged (rows past M, channels past N, unaligned row strides), batched, fp32-B and pre-split-B launches"""
    from dana_amd import ops
    if ops.get_mfma_mode() == 0:
        pytest.skip("the f32-MFMA kernel has one epilogue form")
    m, n, k, res, batch, pre = shape
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(batch, m, k, generator=g).to(dev)
    w = (torch.randn(batch, n, k, generator=g) * 0.05).to(dev)
    sc, sh = (torch.rand(n, generator=g) + 0.5).to(dev), torch.randn(n, generator=g).to(dev)
    r = torch.randn(m, n, generator=g).to(dev) if res else None
    b3 = ops.split_weight(w.view(-1), n, k, batch=batch) if pre else None
    outs = []
    prev = ops.set_epilogue_mode(0)
    try:
        for mode in (1, 0):
            ops.set_epilogue_mode(mode)
            out = torch.full((batch, m, n), float("nan"), device=dev)
            if batch > 1 and pre:
                ops.lib().call("dana_gemm_nt", a.data_ptr(), b3.t.data_ptr(), out.data_ptr(), None, None, None, m, n, k, k, b3.kp, n,
                               0, batch, m * k, 3 * n * b3.kp, m * n, 1.0, ops.W_SPLIT3, ops._stream())
            elif batch > 1:
                ops.gemm_nt(a, w, m, n, k, out=out, ldc=n, batch=batch, batch_a=m * k, batch_b=n * k, batch_c=m * n)
            else:
                ops.gemm_nt(a, b3 if pre else w, m, n, k, out=out, ldc=n, scale=sc, shift=sh, residual=r, relu=True)
            outs.append(out)
        torch.cuda.synchronize()
    finally:
        ops.set_epilogue_mode(prev)
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])


def _with_env(name, value, fn):
    import os
    prev = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if prev is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = prev


@pytest.mark.parametrize("case", CONV_CASES + [(1, 40, 50, 64, 64, 3, 1, 1, True, False), (2, 31, 33, 128, 160, 3, 2, 1, True, True)])
def test_dma_kernel_gives_the_split_kernels_bits_on_convs(dev, case):
    """igemm_dma_kernel (round 5: weights by LDS-DMA, one fragment set, one staging register set, three workgroups per CU)
    against igemm_split_kernel on pre-split weights: same K-step order, same split, same six products -> the same bits on
    every conv geometry (taps, stride, padding, ragged M / N, residual, ReLU), plain and written into a wider buffer"""
    from dana_amd import ops
    if ops.get_mfma_mode() == 0:
        pytest.skip("split kernel only")
    N, H, W, Cin, Cout, k, stride, pad, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    xd = torch.randn(N * H * W, Cin, generator=g).to(dev)
    wp = (torch.randn(Cout, k * k * Cin, generator=g) / np.sqrt(Cin * k * k)).to(dev)
    w3 = ops.split_weight(wp, Cout, k * k * Cin)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    resd = torch.randn(N * oh * ow, Cout, generator=g).to(dev) if use_res else None

    def run():
        o1, _, _ = ops.conv2d_nhwc(xd, N, H, W, Cin, w3, Cout, k, k, stride, pad, scale=sc, shift=sh, residual=resd, relu=relu)
        wide = torch.full((N * oh * ow, Cout + 32), -7.0, device=dev)
        ops.conv2d_nhwc(xd, N, H, W, Cin, w3, Cout, k, k, stride, pad, scale=sc, shift=sh, residual=resd, relu=relu,
                        out=wide.view(-1)[8:], out_stride=Cout + 32)
        torch.cuda.synchronize()
        return o1, wide

    old = _with_env("DANA_DMA_KERNEL", "0", run)
    new = _with_env("DANA_DMA_KERNEL", "2", run)
    assert torch.isfinite(new[0]).all()
    assert torch.equal(old[0], new[0]) and torch.equal(old[1], new[1])


@pytest.mark.parametrize("shape", [(9576, 1024, 256, True, 1), (1000, 200, 100, True, 1), (640, 256, 256, False, 9), (300, 128, 64, False, 1),
                                   (25088, 128, 1024, True, 1), (77, 30, 64, True, 1), (150000, 64, 64, False, 1), (513, 72, 512, False, 1)],
                         ids=["l3-expand+res", "ragged", "planes-b9", "one-round", "roi-proj+res", "tiny-ragged", "n64", "n72"])
def test_dma_kernel_gives_the_split_kernels_bits_on_gemms(dev, shape):
    """... and on GEMM-type launches: fp32 activation rows (APRE = 0) AND the activation rows as split planes (APRE = 1, two /
    three / four / six stages), batched and ragged"""
    import os
    from dana_amd import ops
    if ops.get_mfma_mode() == 0:
        pytest.skip("split kernel only")
    m, n, k, res, batch = shape
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(batch * m, k, generator=g).to(dev)
    w = (torch.randn(batch * n, k, generator=g) * 0.05).to(dev)
    sc, sh = (torch.rand(n, generator=g) + 0.5).to(dev), torch.randn(n, generator=g).to(dev)
    r = torch.randn(m, n, generator=g).to(dev) if res else None
    w3 = ops.split_weight(w, n, k, batch=batch)
    a3 = ops.split_weight(a, m, k, batch=batch)

    def run(planes=False):
        out = torch.full((batch * m, n), float("nan"), device=dev)
        if planes:
            if batch > 1:
                ops.gemm_nt(a3, w3, m, n, k, out=out, ldc=n, batch=batch, batch_c=m * n)
            else:
                ops.gemm_nt(a3, w3, m, n, k, out=out, ldc=n, scale=sc, shift=sh, residual=r, relu=True)
        elif batch > 1:
            ops.lib().call("dana_gemm_nt", a.data_ptr(), w3.t.data_ptr(), out.data_ptr(), None, None, None, m, n, k, k, w3.kp, n,
                           0, batch, m * k, 3 * n * w3.kp, m * n, 1.0, ops.W_SPLIT3, ops._stream())
        else:
            ops.gemm_nt(a, w3, m, n, k, out=out, ldc=n, scale=sc, shift=sh, residual=r, relu=True, force_slices=1)
        torch.cuda.synchronize()
        return out

    old = _with_env("DANA_DMA_KERNEL", "0", run)
    new = _with_env("DANA_DMA_KERNEL", "2", run)
    assert torch.isfinite(new).all() and torch.equal(old, new)
    for st in ("2", "3", "4", "6", "13", "14"):  # (1x: two fragment sets)
        pp = _with_env("DANA_PP_STAGES", st, lambda: run(True))
        assert torch.equal(old, pp), "planes x planes, %s stages" % st


@pytest.mark.parametrize("case", CONV_CASES)
def test_register_epilogue_bits_on_every_conv_case(dev, case):
    """... and on every CONV_CASES shape (strided / padded convs, residual, ReLU, N not a tile multiple), plus the same conv
    written into a wider buffer (row stride > channels) and with a ReLU-adjoint mask (the data-gradient form)"""
    from dana_amd import ops
    if ops.get_mfma_mode() == 0:
        pytest.skip("the f32-MFMA kernel has one epilogue form")
    N, H, W, Cin, Cout, k, stride, pad, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    xd = torch.randn(N * H * W, Cin, generator=g).to(dev)
    wp = (torch.randn(Cout, k * k * Cin, generator=g) / np.sqrt(Cin * k * k)).to(dev)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    resd = torch.randn(N * oh * ow, Cout, generator=g).to(dev) if use_res else None
    ldo = Cout + 32
    maskd = torch.randn(N * oh * ow, Cout, generator=g).to(dev) if Cout % 4 == 0 else None
    got = []
    prev = ops.set_epilogue_mode(0)
    try:
        for mode in (1, 0):
            ops.set_epilogue_mode(mode)
            o1, _, _ = ops.conv2d_nhwc(xd, N, H, W, Cin, wp, Cout, k, k, stride, pad, scale=sc, shift=sh, residual=resd, relu=relu)
            wide = torch.full((N * oh * ow, ldo), -7.0, device=dev)
            ops.conv2d_nhwc(xd, N, H, W, Cin, wp, Cout, k, k, stride, pad, scale=sc, shift=sh, residual=resd, relu=relu,
                            out=wide.view(-1)[8:], out_stride=ldo)
            o3 = None
            if maskd is not None:
                o3 = torch.empty(N * oh * ow, Cout, device=dev)
                ops.lib().call("dana_conv2d_nhwc_masked", xd.data_ptr(), wp.data_ptr(), o3.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                               resd.data_ptr() if use_res else None, maskd.data_ptr(), N, H, W, Cin, Cout, k, k, stride, pad, 0, 0, 0, 0,
                               ops.EPI_RELU if relu else 0, ops._stream())
            got.append((o1, wide, o3))
        torch.cuda.synchronize()
    finally:
        ops.set_epilogue_mode(prev)
    assert torch.equal(got[0][0], got[1][0])
    assert torch.equal(got[0][1], got[1][1]) and (got[1][1][:, :8] == -7).all() and (got[1][1][:, 8 + Cout:] == -7).all()
    assert torch.equal(got[1][1][:, 8:8 + Cout], got[1][0])
    if maskd is not None:
        assert torch.equal(got[0][2], got[1][2])
        assert torch.equal(got[1][2], torch.where(maskd > 0, got[1][0], torch.zeros_like(got[1][0])))



def test_large_lds_kernels_launch_on_every_visible_device():
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: one process driving several devices (nn.DataParallel's
    thread per device, include/dana_hip.h) must opt in on each of them (csrc/common.h DeviceOnce). Runs a 128 x 128-tile
    contraction (49 KB, LDS epilogue form 67.6 KB), the planes x planes kernel (73.7 KB) and a weight-gradient launch on
    EVERY visible device; skips on a one-GPU box (the driver's 8-GPU node runs it)."""
    from dana_amd import ops
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one visible device")
    if ops.get_mfma_mode() == 0:
        pytest.skip("split kernels only")
    g = torch.Generator().manual_seed(3)
    a0, w0 = torch.randn(1024, 256, generator=g), torch.randn(384, 256, generator=g) * 0.05
    ref = a0.double() @ w0.double().t()
    for d in range(n_dev):
        with torch.cuda.device(d):
            dv = torch.device("cuda", d)
            a, w = a0.to(dv), w0.to(dv)
            w3, a3 = ops.split_weight(w, 384, 256), ops.split_weight(a, 1024, 256)
            outs = [ops.gemm_nt(a, w, 1024, 384, 256, force_slices=1), ops.gemm_nt(a, w3, 1024, 384, 256, force_slices=1),
                    ops.gemm_nt(a3, w3, 1024, 384, 256)]
            prev = ops.set_epilogue_mode(1)
            try:
                outs.append(ops.gemm_nt(a, w, 1024, 384, 256, force_slices=1))
            finally:
                ops.set_epilogue_mode(prev)
            dw = ops.linear_wgrad(outs[0], a, 1024, 384, 256)[0]
            torch.cuda.synchronize(dv)
            for o in outs:
                _close(o.cpu(), ref.float())
            _close(dw.cpu(), (outs[0].cpu().double().t() @ a0.double()).float(), 2e-4)
