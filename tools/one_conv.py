"""Run one conv shape repeatedly (for rocprofv3 --pmc): tools/one_conv.py N H W Cin Cout k stride res iters"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import dana_amd
from dana_amd import ops
n, h, w, ci, co, k, st, res, iters = [int(v) for v in sys.argv[1:10]]
dev = torch.device('cuda:0')
x = torch.randn(n * h * w, ci, device=dev)
wt = torch.randn(co, k * k * ci, device=dev) * 0.05
sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
oh, ow = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
r = torch.randn(n * oh * ow, co, device=dev) if res else None
for _ in range(iters):
    ops.conv2d_nhwc(x, n, h, w, ci, wt, co, k, k, st, k // 2, scale=sc, shift=sh, residual=r, relu=True)
torch.cuda.synchronize()
