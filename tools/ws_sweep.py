"""The warp-specialised persistent contraction kernel (csrc/igemm_ws.h) against igemm_split_kernel on the GEMM-type shapes of
the step: bit-equality of the results and duration alone on the chip (median of HIP-event brackets).
usage: python tools/ws_sweep.py [iters] [out.md]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402,F401
from dana_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)

# (name, M, N, K, residual, batch, presplit weights)
SHAPES = [
    ("l1 conv1 64->64", 150000, 64, 64, False, 1, True),
    ("l1 reduce 256->64", 150000, 64, 256, False, 1, True),
    ("l2 reduce 512->128", 37500, 128, 512, False, 1, True),
    ("l2 expand 128->512 +res", 37500, 512, 128, True, 1, True),
    ("l3 reduce 1024->256", 9576, 256, 1024, False, 1, True),
    ("l3 expand 256->1024 +res", 9576, 1024, 256, True, 1, True),
    ("l3 both batches reduce", 19176, 256, 1024, False, 1, True),
    ("l3 both batches expand +res", 19176, 1024, 256, True, 1, True),
    ("l4 reduce 2048->512", 8192, 512, 2048, False, 1, True),
    ("l4 expand 512->2048 +res", 8192, 2048, 512, True, 1, True),
    ("roi q/t proj 1024->128 +res", 25088, 128, 1024, True, 1, True),
    ("roi tr 1024->64 +res", 25088, 64, 1024, True, 1, True),
    ("rpn q-proj", 9576, 256, 1024, False, 1, True),
    ("QK^T b4 (fp32 B)", 2394, 1200, 256, False, 4, False),
    ("A.S b4 (fp32 B)", 2394, 1024, 1200, False, 4, False),
    ("wino planes l2 (36 x 2344 x 128 x 128)", 2344, 128, 128, False, 36, True),
    ("wino planes l3 (36 x 640 x 256 x 256)", 640, 256, 256, False, 36, True),
    ("wino planes rpn (36 x 640 x 512 x 2048)", 640, 512, 2048, False, 36, True),
    ("wino planes l4 (36 x 2048 x 512 x 512)", 2048, 512, 512, False, 36, True),
    ("ragged M=1000 N=200 K=100", 1000, 200, 100, True, 1, False),
]


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


lines = ["| shape | GF | split kernel us | warp-specialised us | ratio | same bits |", "|---|---|---|---|---|---|"]
for name, m, n, k, res, batch, pre in SHAPES:
    a = torch.randn(batch, m, k, device=dev)
    w = torch.randn(batch, n, k, device=dev) * 0.05
    sc, sh = torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    b = ops.split_weight(w.view(-1), n, k, batch=batch) if pre and batch == 1 else w
    kw = dict(scale=sc, shift=sh, residual=r, relu=True)
    if batch > 1:
        if pre:
            b = ops.split_weight(w.view(-1), n, k, batch=batch)
        kw = dict(batch=batch, batch_a=m * k, batch_c=m * n)
    outs = {}
    times = {}
    for mode in (0, 2):
        ops.set_ws_mode(mode)
        out = torch.empty(batch, m, n, device=dev)

        def fn():
            if batch > 1 and pre:
                # (the split planes of a batched weight: batch_b in bf16 elements, as the Winograd path passes them)
                ops.lib().call("dana_gemm_nt", a.data_ptr(), b.t.data_ptr(), out.data_ptr(), None, None, None, m, n, k, k, b.kp, n, 0,
                               batch, m * k, 3 * n * b.kp, m * n, 1.0, ops.W_SPLIT3, ops._stream())
            elif batch > 1:
                ops.gemm_nt(a, w, m, n, k, out=out, ldc=n, batch_b=n * k, **kw)
            else:
                ops.gemm_nt(a, b, m, n, k, out=out, ldc=n, **kw)
        times[mode] = bench(fn)
        outs[mode] = out.clone()
    same = torch.equal(outs[0], outs[2])
    gf = 2.0 * batch * m * n * k / 1e9
    lines.append("| %s | %.2f | %.1f | %.1f | %.2f | %s |" % (name, gf, times[0], times[2], times[2] / times[0],
                                                            "yes" if same else "NO (max |d| %.3e)" % float((outs[0] - outs[2]).abs().max())))
    print(lines[-1], flush=True)
ops.set_ws_mode(1)
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        fh.write("\n".join(lines) + "\n")
