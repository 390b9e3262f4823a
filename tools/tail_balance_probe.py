"""What would splitting the LAST partial round of a launch along K buy (verdict r3 item 1a)? Measured with the existing kernels
on the step's three launches with a bad last round: the launch as it is (T tiles on 512 slots), its full rounds alone, its
remainder tiles alone, and the remainder as 2 K-slices (batched slices, the split-K path's first half; + ~6 us reduce pass).
usage: python tools/tail_balance_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402,F401
from dana_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)
ops.SPLIT_K = False


def bench(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def planes(batch, m, n, k):
    a = torch.randn(batch, m, k, device=dev)
    w = torch.randn(batch, n, k, device=dev) * 0.05
    b = ops.split_weight(w.view(-1), n, k, batch=batch)
    out = torch.empty(batch, m, n, device=dev)
    return lambda: ops.lib().call("dana_gemm_nt", a.data_ptr(), b.t.data_ptr(), out.data_ptr(), None, None, None, m, n, k, k, b.kp, n, 0,
                                  batch, m * k, 3 * n * b.kp, m * n, 1.0, ops.W_SPLIT3, ops._stream())


def bmm(batch, m, n, k):
    a = torch.randn(batch, m, k, device=dev)
    w = torch.randn(batch, n, k, device=dev) * 0.05
    out = torch.empty(batch, m, n, device=dev)
    return lambda: ops.gemm_nt(a, w, m, n, k, out=out, ldc=n, batch=batch, batch_a=m * k, batch_b=n * k, batch_c=m * n)


print("| launch | tiles | as it is us | full rounds only us | remainder alone us | remainder as 2 K-slices us (+ ~6 reduce) | bound of the saving us |")
print("|---|---|---|---|---|---|---|")
# RPN conv planes: 36 x (5 x 4 tiles), K = 2048: 720 tiles = 512 + 208 -> 26 planes ~ 520 tiles, 10 planes = 200 tiles
t_all, t_full, t_rem, t_rem2 = bench(planes(36, 640, 512, 2048)), bench(planes(26, 640, 512, 2048)), bench(planes(10, 640, 512, 2048)), bench(planes(20, 640, 512, 1024))
print("| RPN conv plane GEMMs 36 x 640 x 512 x 2048 | 720 | %.1f | %.1f | %.1f | %.1f | %.1f |" % (t_all, t_full, t_rem, t_rem2, t_rem - t_rem2 - 6))
# A.S: 4 images x (19 x 8 tiles), K = 1200 (fp32 B): 608 = 512 + 96 -> rows for 512 tiles: 64 M-tiles over 4 images = 16 per image (2048 rows); remainder 3 M-tiles (346 rows)
t_all, t_full, t_rem, t_rem2 = bench(bmm(4, 2394, 1024, 1200)), bench(bmm(4, 2048, 1024, 1200)), bench(bmm(4, 346, 1024, 1200)), bench(bmm(8, 346, 1024, 592))
print("| A.S 4 x 2394 x 1024 x 1200 | 608 | %.1f | %.1f | %.1f | %.1f | %.1f |" % (t_all, t_full, t_rem, t_rem2, t_rem - t_rem2 - 6))
# layer3 expand conv: 75 x 8 tiles, K = 256: 600 = 512 + 88 -> 64 M-tiles (8192 rows) + 11 (1384 rows)
t_all, t_full, t_rem, t_rem2 = bench(bmm(1, 9576, 1024, 256)), bench(bmm(1, 8192, 1024, 256)), bench(bmm(1, 1384, 1024, 256)), bench(bmm(2, 1384, 1024, 128))
print("| layer3 expand conv 9576 x 1024 x 256 (no epilogue) | 600 | %.1f | %.1f | %.1f | %.1f | %.1f |" % (t_all, t_full, t_rem, t_rem2, t_rem - t_rem2 - 6))
