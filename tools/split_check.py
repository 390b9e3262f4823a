"""Accuracy of the contraction kernels against an fp64 contraction of the same fp32 operands (run once per DANA_MFMA_SPLIT
setting: the switch is read once per process). Prints max |err| / (|a| . |b|) per shape: the fp32 rounding level is 6e-8."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (m, n, k) in [(300, 96, 64), (1000, 256, 1024), (4096, 512, 4608), (777, 130, 36)]:
    a = torch.randn(m, k, device=dev) * torch.exp(torch.randn(m, k, device=dev))
    b = torch.randn(n, k, device=dev) * torch.exp(torch.randn(n, k, device=dev))
    c = ops.gemm_nt(a, b, m, n, k)
    ref = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    rel = ((c.double() - ref).abs() / mag).max().item()
    t = (a @ b.t())
    rel_t = ((t.double() - ref).abs() / mag).max().item()
    print("gemm %5d x %4d x %4d   max err / (|a|.|b|) = %.3e    (rocBLAS fp32: %.3e)" % (m, n, k, rel, rel_t))
x = torch.randn(2 * 20 * 30, 64, device=dev)
w = torch.randn(96, 9 * 64, device=dev)
out, oh, ow = ops.conv2d_nhwc(x, 2, 20, 30, 64, w, 96, 3, 3, 1, 1)
xr = x.view(2, 20, 30, 64).permute(0, 3, 1, 2).double()
wr = w.view(96, 3, 3, 64).permute(0, 3, 1, 2).double()
ref = torch.nn.functional.conv2d(xr, wr, padding=1).permute(0, 2, 3, 1).reshape(-1, 96)
mag = torch.nn.functional.conv2d(xr.abs(), wr.abs(), padding=1).permute(0, 2, 3, 1).reshape(-1, 96)
print("conv3x3 64->96: max err / mag = %.3e" % ((out.double() - ref).abs() / mag).max().item())

# model-like data: post-ReLU activations (exact zeros, one-sided), small weights, long K; and the Winograd path
def relerr(c, ref, mag):
    return ((c.double() - ref).abs() / mag).max().item(), ((c.double() - ref).norm() / ref.norm()).item()
for (m, n, k) in [(2048, 256, 2304), (1024, 64, 3136), (98, 1024, 3136)]:
    a = torch.relu(torch.randn(m, k, device=dev)) * 3.0
    b = torch.randn(n, k, device=dev) * 0.02
    c = ops.gemm_nt(a, b, m, n, k)
    ref = a.double() @ b.double().t(); mag = a.double().abs() @ b.double().abs().t()
    t = a @ b.t()
    print("relu-gemm %5d x %4d x %4d  max err/mag %.3e  l2 rel %.3e   (rocBLAS: %.3e %.3e)" % ((m, n, k) + relerr(c, ref, mag) + relerr(t, ref, mag)))
x = torch.relu(torch.randn(2 * 24 * 32, 256, device=dev))
w = torch.randn(256, 9 * 256, device=dev) * 0.02
xr = x.view(2, 24, 32, 256).permute(0, 3, 1, 2).double()
wr = w.view(256, 3, 3, 256).permute(0, 3, 1, 2).double()
ref = torch.nn.functional.conv2d(xr, wr, padding=1).permute(0, 2, 3, 1).reshape(-1, 256)
mag = torch.nn.functional.conv2d(xr.abs(), wr.abs(), padding=1).permute(0, 2, 3, 1).reshape(-1, 256)
out, oh, ow = ops.conv2d_nhwc(x, 2, 24, 32, 256, w, 256, 3, 3, 1, 1)
print("conv3x3 256->256 direct  : max err/mag %.3e  l2 rel %.3e" % relerr(out, ref, mag))
for tile in (2, 4):
    u = ops.winograd_filter_transform(w, 256, 256, tile=tile) if hasattr(ops, "winograd_filter_transform") else None
    if u is None:
        break
    out = ops.conv3x3_winograd(x, 2, 24, 32, 256, u, 256)
    out = out[0] if isinstance(out, tuple) else out
    print("conv3x3 256->256 winograd F(%dx%d): max err/mag %.3e  l2 rel %.3e" % ((tile, tile) + relerr(out, ref, mag)))
