"""Tile sweep for the igemm kernel on the trunk's layer shapes (run with DANA_IGEMM_TILE=0..3)."""
import os, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import dana_amd
from dana_amd import ops
dev = torch.device('cuda:0')
SHAPES = [  # (name, N_img, H, W, Cin, Cout, k, stride)
    ("l1c2  M150000 64->64 3x3", 4, 150, 250, 64, 64, 3, 1),
    ("l1c3  M150000 64->256 1x1", 4, 150, 250, 64, 256, 1, 1),
    ("l2c2  M37500 128->128 3x3", 4, 75, 125, 128, 128, 3, 1),
    ("l2c3  M37500 128->512 1x1", 4, 75, 125, 128, 512, 1, 1),
    ("l2c1  M37500 512->128 1x1", 4, 75, 125, 512, 128, 1, 1),
    ("l3c1  M9576 1024->256 1x1", 4, 38, 63, 1024, 256, 1, 1),
    ("l4c3  M8192 512->2048 1x1", 512, 4, 4, 512, 2048, 1, 1),
    ("l3c2  M9576 256->256 3x3", 4, 38, 63, 256, 256, 3, 1),
    ("l3c3  M9576 256->1024 1x1", 4, 38, 63, 256, 1024, 1, 1),
    ("l3c2x2 M19152 256->256 3x3", 8, 38, 63, 256, 256, 3, 1),
    ("l3c3x2 M19152 256->1024 1x1", 8, 38, 63, 256, 1024, 1, 1),
    ("rpn   M9576 2048->512 3x3", 4, 38, 63, 2048, 512, 3, 1),
    ("l4c2  M8192 512->512 3x3", 512, 4, 4, 512, 512, 3, 1),
]
for name, n, h, w, ci, co, k, st in SHAPES:
    x = torch.randn(n * h * w, ci, device=dev)
    wt = torch.randn(co, k * k * ci, device=dev) * 0.05
    sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
    for _ in range(3):
        out, oh, ow = ops.conv2d_nhwc(x, n, h, w, ci, wt, co, k, k, st, k // 2, scale=sc, shift=sh, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_nhwc(x, n, h, w, ci, wt, co, k, k, st, k // 2, scale=sc, shift=sh, relu=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gf = 2.0 * n * oh * ow * co * k * k * ci / 1e9
    print("%-32s %8.1f us %7.1f TF/s" % (name, us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (us * 1e-6) / 1e3))
