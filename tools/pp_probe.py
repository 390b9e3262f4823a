"""planes x planes kernel (igemm_pp_kernel: both operands pre-split, LDS-DMA staging, 2 or 3 stages) against the split kernel
(fp32 activation rows split in the K loop, pre-split weights) on the step's GEMM-type shapes: bit equality and duration
(median of event-bracketed launches, interleaved, each launch alone on the chip).
usage: python tools/pp_probe.py [out.md]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: F401
from dana_amd import ops
dev = torch.device("cuda:0")
SHAPES = [  # (name, m, n, k, batch, residual)
    ("l1 c1 256->64", 150000, 64, 256, 1, 0), ("l1 c3 64->256 +res", 150000, 256, 64, 1, 1),
    ("l2 c1 512->128", 37500, 128, 512, 1, 0), ("l2 c3 128->512 +res", 37500, 512, 128, 1, 1),
    ("l3 c1 1024->256", 9576, 256, 1024, 1, 0), ("l3 c3 256->1024 +res", 9576, 1024, 256, 1, 1),
    ("sup l3 c1", 9600, 256, 1024, 1, 0), ("sup l3 c3 +res", 9600, 1024, 256, 1, 1),
    ("l4 c1 2048->512", 8192, 512, 2048, 1, 0), ("l4 c3 512->2048 +res", 8192, 2048, 512, 1, 1),
    ("roi q-proj", 25088, 256, 1024, 1, 0), ("rpn q-proj", 9576, 256, 1024, 1, 0),
    ("wino l2 planes", 2432, 128, 128, 36, 0), ("wino l3 planes", 640, 256, 256, 36, 0), ("wino l4 planes", 512, 512, 512, 36, 0),
    ("wino rpn planes", 640, 512, 2048, 36, 0), ("ragged", 1000, 200, 100, 1, 1),
]
REPS, ROUNDS = 8, 5


def bracket(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


rows = []
for name, m, n, k, b, res in SHAPES:
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(b * m, k, generator=g).to(dev)
    w = (torch.randn(b * n, k, generator=g) * 0.05).to(dev)
    r = torch.randn(m, n, generator=g).to(dev) if res else None
    sc, sh = (torch.rand(n, generator=g) + 0.5).to(dev), torch.randn(n, generator=g).to(dev)
    w3 = ops.split_weight(w, n, k, batch=b)
    a3 = ops.split_weight(a, m, k, batch=b)
    o_ref = torch.empty(b * m, n, device=dev)
    o_pp = torch.empty(b * m, n, device=dev)

    def run_split():
        if b > 1:
            ops.lib().call("dana_gemm_nt", a.data_ptr(), w3.t.data_ptr(), o_ref.data_ptr(), None, None, None, m, n, k, k, w3.kp, n,
                           0, b, m * k, 3 * n * w3.kp, m * n, 1.0, ops.W_SPLIT3, ops._stream())
        else:
            ops.gemm_nt(a, w3, m, n, k, out=o_ref, ldc=n, scale=sc, shift=sh, residual=r, relu=True)

    def run_pp():
        if b > 1:
            ops.gemm_nt(a3, w3, m, n, k, out=o_pp, ldc=n, batch=b, batch_c=m * n)
        else:
            ops.gemm_nt(a3, w3, m, n, k, out=o_pp, ldc=n, scale=sc, shift=sh, residual=r, relu=True)

    STAGES = ("2", "3", "4", "6", "13", "14")
    t = {"split": [], "dma": []}
    t.update({"pp" + st: [] for st in STAGES})
    same = {}
    os.environ["DANA_DMA_KERNEL"] = "0"
    run_split()
    torch.cuda.synchronize()
    o_old = o_ref.clone()
    os.environ["DANA_DMA_KERNEL"] = "2"
    o_ref.fill_(float("nan"))
    run_split()
    torch.cuda.synchronize()
    same["dma"] = bool(torch.equal(o_ref, o_old))
    for st in STAGES:
        os.environ["DANA_PP_STAGES"] = st
        o_pp.fill_(float("nan"))
        run_pp()
        torch.cuda.synchronize()
        same[st] = bool(torch.equal(o_old, o_pp))
    for _ in range(ROUNDS):
        os.environ["DANA_DMA_KERNEL"] = "0"
        t["split"].append(bracket(run_split))
        os.environ["DANA_DMA_KERNEL"] = "2"
        t["dma"].append(bracket(run_split))
        for st in STAGES:
            os.environ["DANA_PP_STAGES"] = st
            t["pp" + st].append(bracket(run_pp))
    med = {k_: sorted(v)[len(v) // 2] for k_, v in t.items()}
    rows.append((name, m, n, k, b, 2.0 * b * m * n * k / 1e9, med, same))
L = ["| shape | GF | round-4 split kernel (2/CU) us | dma kernel, fp32 rows (3/CU) us | planes x planes 2 stages (3/CU) | 3 stages (2/CU) | 4 stages (1/CU) | 6 stages (1/CU) | 3 stages + 2 fragment sets | 4 stages + 2 fragment sets | dma / split | best planes / split | same bits |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for name, m, n, k, b, gf, med, same in rows:
    best = min(med["pp" + st_] for st_ in STAGES)
    L.append("| %s M=%d N=%d K=%d b%d | %.2f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f | %.2f | %s |" % (
        name, m, n, k, b, gf, med["split"], med["dma"], med["pp2"], med["pp3"], med["pp4"], med["pp6"], med["pp13"], med["pp14"], med["dma"] / med["split"],
        best / med["split"], "yes" if all(same.values()) else "NO %s" % same))
text = "\n".join(L) + "\n"
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
