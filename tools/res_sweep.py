"""HBM-bound 1x1 expand convs with a residual: tile choice (run with DANA_MFMA_SPLIT=1|2|3|4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops
dev = torch.device('cuda:0')
for (m, ci, co, res) in [(150000, 64, 256, 1), (150000, 64, 256, 0), (37500, 128, 512, 1), (9576, 256, 1024, 1), (9576, 1024, 256, 0), (9576, 512, 1024, 0), (8192, 512, 2048, 1), (8192, 2048, 512, 0)]:
    x = torch.randn(m, ci, device=dev); w = torch.randn(co, ci, device=dev) * 0.05
    sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    r = torch.randn(m, co, device=dev) if res else None
    out = torch.empty(m, co, device=dev)
    f = lambda: ops.conv2d_nhwc(x, 1, m, 1, ci, w, co, 1, 1, 1, 0, scale=sc, shift=sh, residual=r, relu=True, out=out)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    gb = 4.0 * (m * ci + m * co * (2 if res else 1)) / 1e9
    print("M=%6d %4d->%4d res=%d  %7.1f us  %6.1f TF/s  %5.2f TB/s" % (m, ci, co, res, us, 2.0 * m * ci * co / us / 1e6, gb / us * 1e3))
