"""Backward pass of the DAnA forward path on the HIP kernels (groundwork for the training step, SURVEY.md 8d
variant S). What `loss.backward()` does through autograd + cuDNN/cuBLAS in the reference (train.py:141-143) is
assembled here from the C-ABI building blocks: data gradients on the forward implicit-GEMM kernel with
transformed weights, weight gradients on the split-M TN MFMA kernel, and the element-wise adjoints of
backward.hip. Frozen BatchNorm (dana.py:362-385) only scales gradients; BN parameters, conv1/bn1/layer1 get none
(dana.py:350-360, cfg.RESNET.FIXED_BLOCKS = 1).

Status: the bottleneck / conv / linear adjoints below are complete and tested against autograd
(tests/test_gpu_backward.py); the orchestration of the full model backward is the next round's work."""
import torch

from . import ops


class WeightGrads:
    """Accumulates packed weight gradients per conv (query and support passes share the weights)."""

    def __init__(self):
        self.packed = {}

    def add_conv(self, key, g, x, n, h, w, c, in_stride=0, grad_stride=0):
        buf = self.packed.get(key)
        if buf is None:
            self.packed[key] = ops.conv2d_wgrad(g, x, n, h, w, c["cin"], c["cout"], c["k"], c["k"], c["stride"],
                                                c["pad"], in_stride=in_stride, grad_stride=grad_stride)
        else:
            ops.conv2d_wgrad(g, x, n, h, w, c["cin"], c["cout"], c["k"], c["k"], c["stride"], c["pad"],
                             in_stride=in_stride, grad_stride=grad_stride, out=buf)

    def finish_conv(self, key, c, param):
        """apply the frozen-BN scale to the rows and add into param.grad (OIHW)"""
        buf = self.packed.pop(key)
        if c.get("scale") is not None:
            ops.rowscale_(buf, c["scale"], c["cout"], c["k"] * c["k"] * c["cin"])
        fresh = param.grad is None
        if fresh:
            param.grad = torch.empty_like(param)
        ops.unpack_conv_weight_grad(buf, param.grad, c["cout"], c["cin"], c["k"], c["k"], accumulate=not fresh)


def conv_backward(g, x, n, h, w, c, grads, key, need_dx=True, in_stride=0):
    """g: gradient w.r.t. the conv+BN output [n*oh*ow][cout] (ReLU mask already applied).
    Records dW (raw, scale applied at finish) and returns dx [n*h*w][cin] (or None)."""
    grads.add_conv(key, g, x, n, h, w, c, in_stride=in_stride)
    if not need_dx:
        return None
    return ops.conv2d_dgrad(g, c["w"], n, h, w, c["cin"], c["cout"], c["k"], c["k"], c["stride"], c["pad"],
                            scale=c.get("scale"))


def bottleneck_backward(g_out, saved, n, h, w, bp, grads, key, need_dx=True):
    """Adjoint of DAnARCNN._bottleneck. saved = dict(x, o1, o2, o3, h1, w1) from the forward; g_out = dL/d(o3).
    Returns dL/dx [n*h*w][cin] (None if not needed). g_out is consumed (masked in place)."""
    h1, w1 = saved["h1"], saved["w1"]
    m_out = n * h1 * w1
    cout = bp["c3"]["cout"]
    g = ops.relu_mask_(g_out, saved["o3"], m_out, cout)  # through the final ReLU (resnet.py:100)
    # main branch: conv3 <- conv2 <- conv1
    g2 = conv_backward(g, saved["o2"], n, h1, w1, bp["c3"], grads, key + ".conv3")
    ops.relu_mask_(g2, saved["o2"], m_out, bp["c2"]["cout"])
    g1 = conv_backward(g2, saved["o1"], n, h1, w1, bp["c2"], grads, key + ".conv2")
    ops.relu_mask_(g1, saved["o1"], m_out, bp["c1"]["cout"])
    dx = conv_backward(g1, saved["x"], n, h, w, bp["c1"], grads, key + ".conv1", need_dx=need_dx)
    # residual branch (resnet.py:96-99)
    if bp["ds"] is not None:
        dxr = conv_backward(g, saved["x"], n, h, w, bp["ds"], grads, key + ".downsample.0", need_dx=need_dx)
        if need_dx:
            ops.axpy_rows_(dx, dxr, n * h * w, bp["c1"]["cin"])
    elif need_dx:
        ops.axpy_rows_(dx, g, n * h * w, cout)
    return dx
