"""Backward building blocks (training-step groundwork) vs torch autograd of the oracle's functional blocks."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-4):
    a, b = a.double(), b.double()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, "max err %.3e vs scale %.3e" % (err, scale)


@pytest.mark.parametrize("stride,inplanes,planes,hw", [(1, 256, 64, (13, 17)), (2, 256, 128, (14, 18)), (2, 512, 256, (9, 12))])
def test_bottleneck_backward_vs_autograd(dev, stride, inplanes, planes, hw):
    """dL/dx and dL/dW of one Caffe bottleneck (frozen BN, stride on the first 1x1, optional downsample)"""
    import dana_amd
    from dana_amd import ops, backward as BW
    from dana_amd.dana import Bottleneck, DAnARCNN
    import torch.nn as nn
    from oracle import model_ref as O
    torch.manual_seed(stride * 100 + planes)
    ds = None
    if stride != 1 or inplanes != planes * 4:
        ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
    blk = Bottleneck(inplanes, planes, stride, ds)
    for mod in blk.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.1)
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    blk.eval()
    N, (H, W) = 2, hw
    x = torch.randn(N, inplanes, H, W)
    # reference: autograd through the oracle's functional bottleneck (double precision)
    sd = {"b." + k: v.detach().double().requires_grad_(v.dtype.is_floating_point and "conv" in k or "downsample.0" in k)
          for k, v in blk.state_dict().items()}
    xr = x.double().requires_grad_(True)
    y = O.bottleneck(xr, sd, "b", stride)
    gy = torch.randn(y.shape, dtype=torch.double)
    y.backward(gy)
    # HIP: forward with saved activations, then the adjoint
    blk.to(dev)
    helper = DAnARCNN(["fg", "bg"], num_shot=1)
    bp = helper._block_plan(blk)
    xd = ops.nchw_to_nhwc(x.to(dev)).view(-1, inplanes)
    saved = []
    o3, h1, w1 = helper._bottleneck(xd, N, H, W, bp, save=saved)
    _close(ops.nhwc_to_nchw(o3, N, planes * 4, h1, w1).cpu(), y.detach(), 1e-4)
    g = ops.nchw_to_nhwc(gy.float().to(dev)).view(-1, planes * 4).contiguous()
    grads = BW.WeightGrads()
    dx = BW.bottleneck_backward(g, saved[0], N, H, W, bp, grads, "b")
    _close(ops.nhwc_to_nchw(dx, N, inplanes, H, W).cpu(), xr.grad)
    names = [("conv1", bp["c1"], blk.conv1), ("conv2", bp["c2"], blk.conv2), ("conv3", bp["c3"], blk.conv3)]
    if ds is not None:
        names.append(("downsample.0", bp["ds"], blk.downsample[0]))
    for nm, c, mod in names:
        grads.finish_conv("b." + nm, c, mod.weight)
        _close(mod.weight.grad.cpu(), sd["b.%s.weight" % nm].grad)
    assert not grads.packed
