"""Asymptotic rate of dana_gemm_nt on large shapes (tile-count / K sensitivity)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops
dev = torch.device('cuda:0')
for (m, n, k) in [(8192, 512, 4608), (8192, 1024, 4608), (8192, 2048, 4608), (16384, 4096, 4096), (8192, 2048, 512), (8192, 2048, 256)]:
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev)
    out = torch.empty(m, n, device=dev)
    for _ in range(2): ops.gemm_nt(a, b, m, n, k, out=out, ldc=n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm_nt(a, b, m, n, k, out=out, ldc=n)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("gemm %6d x %5d x %5d  %9.1f us  %7.1f TF/s" % (m, n, k, us, 2.0 * m * n * k / us / 1e6))
