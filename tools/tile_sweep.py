"""Which block tile is fastest for each contraction shape of the bs-4 step? Times the split kernel with the tile forced
(dana_debug_force_tile 4: 128x128, 2: 128x64, 5: 64x128, 3: 64x64) and with the dispatcher's own choice (1 -> 0).
usage: python tools/tile_sweep.py [out.md]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: F401
from dana_amd import ops
dev = torch.device("cuda:0")
# (kind, M-geometry (n, h, w), cin, cout, k, stride, residual) -- the distinct shapes of gpurun_out/*/launches.txt
CONV = [
    ("l1 c1a", 4, 150, 250, 64, 64, 1, 1, 0), ("l1 c2", 4, 150, 250, 64, 64, 3, 1, 0), ("l1 c3", 4, 150, 250, 64, 256, 1, 1, 1),
    ("l1 c1", 4, 150, 250, 256, 64, 1, 1, 0),
    ("l2 c1s2", 4, 150, 250, 256, 128, 1, 2, 0), ("l2 ds", 4, 150, 250, 256, 512, 1, 2, 0), ("l2 c3", 4, 75, 125, 128, 512, 1, 1, 1),
    ("l2 c1", 4, 75, 125, 512, 128, 1, 1, 0),
    ("l3 c1s2", 4, 75, 125, 512, 256, 1, 2, 0), ("l3 ds", 4, 75, 125, 512, 1024, 1, 2, 0), ("l3 c3", 4, 38, 63, 256, 1024, 1, 1, 1),
    ("l3 c1", 4, 38, 63, 1024, 256, 1, 1, 0),
    ("l4 c1s2", 512, 7, 7, 1024, 512, 1, 2, 0), ("l4 ds", 512, 7, 7, 1024, 2048, 1, 2, 0), ("l4 c3", 512, 4, 4, 512, 2048, 1, 1, 1),
    ("l4 c1", 512, 4, 4, 2048, 512, 1, 1, 0),
    ("sup l3 c3", 24, 20, 20, 256, 1024, 1, 1, 1), ("sup l3 c1", 24, 20, 20, 1024, 256, 1, 1, 0),
]
GEMM = [  # (name, m, n, k, batch)
    ("q-proj", 9576, 256, 1024, 1), ("k-proj", 4800, 256, 1024, 1), ("QK^T", 2394, 1200, 256, 4), ("A.S", 2394, 1024, 1200, 4),
    ("rpn heads", 9576, 72, 512, 1), ("roi q-proj", 25088, 256, 1024, 1), ("roi transform", 25088, 64, 1024, 1),
    ("roi QK^T", 6272, 147, 256, 4), ("roi A.S", 6272, 1024, 160, 4), ("ffn1", 512, 1024, 3136, 1), ("k2-proj", 1176, 256, 1024, 1),
    # the 36 plane GEMMs of the Winograd F(4x4) convs: [4x4 tiles][cin] x [cout][cin]
    ("wino rpn", 640, 512, 2048, 36), ("wino l4", 512, 512, 512, 36), ("wino l3", 640, 256, 256, 36), ("wino l2", 2432, 128, 128, 36),
    ("wino sup l3", 600, 256, 256, 36), ("wino sup l2", 2400, 128, 128, 36),
]
if os.environ.get("SWEEP_ONLY"):
    CONV = []
    GEMM = [g for g in GEMM if g[0].startswith(os.environ["SWEEP_ONLY"])]
MODES = [(1, "auto"), (4, "128x128"), (2, "128x64"), (5, "64x128"), (3, "64x64")]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


rows = []
for name, n, h, w, ci, co, k, st, res in CONV:
    x = torch.randn(n * h * w, ci, device=dev)
    wt = torch.randn(co, k * k * ci, device=dev) * 0.05
    sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    oh, ow = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    r = torch.randn(n * oh * ow, co, device=dev) if res else None
    gf = 2.0 * n * oh * ow * co * k * k * ci / 1e9
    t = []
    for mode, _ in MODES:
        ops.force_tile(0 if mode == 1 else mode)
        t.append(timeit(lambda: ops.conv2d_nhwc(x, n, h, w, ci, wt, co, k, k, st, k // 2, scale=sc, shift=sh, residual=r, relu=True)))
    rows.append(("%s M=%d N=%d K=%d%s" % (name, n * oh * ow, co, k * k * ci, " +res" if res else ""), gf, t))
for name, m, n, k, b in GEMM:
    a = torch.randn(b * m, k, device=dev)
    bm = torch.randn(b * n, k, device=dev)
    out = torch.empty(b * m, n, device=dev)
    gf = 2.0 * b * m * n * k / 1e9
    t = []
    for mode, _ in MODES:
        ops.force_tile(0 if mode == 1 else mode)
        t.append(timeit(lambda: ops.gemm_nt(a, bm, m, n, k, out=out, ldc=n, batch=b, batch_a=m * k, batch_b=n * k, batch_c=m * n)))
    rows.append(("gemm %s M=%d N=%d K=%d b%d" % (name, m, n, k, b), gf, t))
ops.force_tile(0)
L = ["| shape | GF | " + " | ".join("%s us" % nm for _, nm in MODES) + " | best |", "|---|---|" + "---|" * (len(MODES) + 1)]
for name, gf, t in rows:
    best = min(range(1, len(t)), key=lambda i: t[i])
    L.append("| %s | %.2f | %s | %s (%.0f TF/s) |" % (name, gf, " | ".join("%.1f" % v for v in t), MODES[best][1], gf / t[best] * 1e3))
text = "\n".join(L) + "\n"
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
