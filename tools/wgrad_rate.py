"""Rate of dana_conv2d_wgrad_nhwc (1x1 shapes of the training iteration), both kernel modes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops
dev = torch.device('cuda:0')
SH = [(9600, 256, 1024), (9600, 1024, 256), (8192, 512, 2048), (8192, 2048, 512), (38400, 128, 512), (38400, 512, 128),
      (25088, 1024, 64), (6272, 1024, 160), (153600, 64, 256), (65536, 1024, 1024)]
for mode in (1, 0):
    ops.set_mfma_mode(mode)
    for (m, ci, co) in SH:
        x = torch.randn(m, ci, device=dev); g = torch.randn(m, co, device=dev)
        f = lambda: ops.conv2d_wgrad(g, x, 1, m, 1, ci, co, 1, 1, 1, 0)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print("mode %d  M=%6d cin=%4d cout=%4d  %7.1f us  %6.1f TF/s" % (mode, m, ci, co, us, 2.0 * m * ci * co / us / 1e6))
