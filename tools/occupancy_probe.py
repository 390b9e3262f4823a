"""Two questions about the split kernel, answered per contraction shape of the bs-4 step with interleaved in-process A/Bs
(median of event-bracketed launches, each launch alone on the chip):
 (1) epilogue on the accumulator registers (default) vs through the LDS C tile (dana_set_epilogue_mode(1));
 (2) what does one more resident workgroup per CU buy? The same tile with its dynamic LDS padded (DANA_LDS_PAD, read per
     launch) so that fewer workgroups fit: 128x64 at 3 vs 2 per CU, 64x64 at 4 vs 3 vs 2.
usage: python tools/occupancy_probe.py [out.md]"""
import os, sys
os.environ["DANA_LDS_PAD"] = "0"  # (present from the first launch on: the kernels opt in to the full 160 KB once)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: F401
from dana_amd import ops
dev = torch.device("cuda:0")
CONV = [
    ("l1 c1a", 4, 150, 250, 64, 64, 1, 1, 0), ("l1 c3", 4, 150, 250, 64, 256, 1, 1, 1), ("l1 c1", 4, 150, 250, 256, 64, 1, 1, 0),
    ("l2 c3", 4, 75, 125, 128, 512, 1, 1, 1), ("l2 c1", 4, 75, 125, 512, 128, 1, 1, 0),
    ("l3 c3", 4, 38, 63, 256, 1024, 1, 1, 1), ("l3 c1", 4, 38, 63, 1024, 256, 1, 1, 0),
    ("l4 c3", 512, 4, 4, 512, 2048, 1, 1, 1), ("l4 c1", 512, 4, 4, 2048, 512, 1, 1, 0),
    ("sup l3 c3", 24, 20, 20, 256, 1024, 1, 1, 1),
]
GEMM = [("QK^T", 2394, 1200, 256, 4), ("A.S", 2394, 1024, 1200, 4), ("roi q-proj", 25088, 256, 1024, 1),
        ("wino rpn", 640, 512, 2048, 36), ("wino l4", 512, 512, 512, 36), ("wino l3", 640, 256, 256, 36), ("wino l2", 2432, 128, 128, 36)]
REPS, ROUNDS = 8, 5


def bracket(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


def ab(fn, variants):
    """variants: list of (label, setup()) -> {label: median us}"""
    t = {lab: [] for lab, _ in variants}
    for lab, setup in variants:
        setup()
        fn()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for lab, setup in variants:
            setup()
            t[lab].append(bracket(fn))
    return {lab: sorted(v)[len(v) // 2] for lab, v in t.items()}


def setv(mode, epi, pad):
    def f():
        ops.set_mfma_mode(mode)
        ops.set_epilogue_mode(epi)
        os.environ["DANA_LDS_PAD"] = str(pad)
    return f


# 128x64 staging = 36.9 KB (3 per CU by registers), + 20 KB -> 2 per CU; 64x64 = 24.6 KB (4 per CU by registers), + 16 KB -> 3, + 40 KB -> 2
VARS = [("auto reg-epi", setv(1, 0, 0)), ("auto lds-epi", setv(1, 1, 0)),
        ("128x128 reg", setv(4, 0, 0)), ("128x128 lds", setv(4, 1, 0)),
        ("128x64 3/CU", setv(2, 0, 0)), ("128x64 2/CU", setv(2, 0, 20480)),
        ("64x64 4/CU", setv(3, 0, 0)), ("64x64 3/CU", setv(3, 0, 16384)), ("64x64 2/CU", setv(3, 0, 40960))]
rows = []
for name, n, h, w, ci, co, k, st, res in CONV:
    x = torch.randn(n * h * w, ci, device=dev)
    wt = torch.randn(co, k * k * ci, device=dev) * 0.05
    w3 = ops.split_weight(wt, co, k * k * ci)
    sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    oh, ow = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    r = torch.randn(n * oh * ow, co, device=dev) if res else None
    gf = 2.0 * n * oh * ow * co * k * k * ci / 1e9
    out = torch.empty(n * oh * ow, co, device=dev)
    res_t = ab(lambda: ops.conv2d_nhwc(x, n, h, w, ci, w3, co, k, k, st, k // 2, scale=sc, shift=sh, residual=r, relu=True, out=out, out_stride=co), VARS)
    rows.append(("%s M=%d N=%d K=%d%s" % (name, n * oh * ow, co, k * k * ci, " +res" if res else ""), gf, res_t))
for name, m, n, k, b in GEMM:
    a = torch.randn(b * m, k, device=dev)
    bm = torch.randn(b * n, k, device=dev)
    out = torch.empty(b * m, n, device=dev)
    gf = 2.0 * b * m * n * k / 1e9
    res_t = ab(lambda: ops.gemm_nt(a, bm, m, n, k, out=out, ldc=n, batch=b, batch_a=m * k, batch_b=n * k, batch_c=m * n), VARS)
    rows.append(("gemm %s M=%d N=%d K=%d b%d" % (name, m, n, k, b), gf, res_t))
setv(1, 0, 0)()
labs = [lab for lab, _ in VARS]
L = ["| shape | GF | " + " | ".join(labs) + " |", "|---|---|" + "---|" * len(labs)]
for name, gf, t in rows:
    L.append("| %s | %.2f | %s |" % (name, gf, " | ".join("%.1f" % t[lab] for lab in labs)))
text = "\n".join(L) + "\n"
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
