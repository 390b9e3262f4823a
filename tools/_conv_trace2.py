import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops
from dana_amd._lib import lib
dev = torch.device("cuda:0")
for (n, h, w, ci, co, res) in [(4, 38, 63, 256, 1024, 1), (4, 38, 63, 1024, 256, 0), (4, 150, 250, 64, 256, 1), (512, 4, 4, 2048, 512, 0)]:
    x = torch.randn(n * h * w, ci, device=dev); wt = torch.randn(co, ci, device=dev) * 0.05
    sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    r = torch.randn(n * h * w, co, device=dev) if res else None
    run = lambda: ops.conv2d_nhwc(x, n, h, w, ci, wt, co, 1, 1, 1, 0, scale=sc, shift=sh, residual=r, relu=True)
    for _ in range(5): run()
    torch.cuda.synchronize()
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    lib().call("dana_set_igemm_trace", buf.data_ptr()); run(); torch.cuda.synchronize(); lib().call("dana_set_igemm_trace", None)
    raw = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8); raw = raw[raw[:, 3] > 0]
    t = raw.astype(np.float64)
    pro, lp, epi = t[:,1]-t[:,0], t[:,2]-t[:,1], t[:,3]-t[:,2]
    setup = t[:, 4]; first = (raw[:, 7] >> np.uint64(32)).astype(np.float64); lds = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
    print("M=%d N=%d K=%d res=%d blocks=%d | prologue %.0f = setup %.0f + first-tiles %.0f + second-stage %.0f | loop %.0f | epilogue %.0f = to-LDS %.0f + out %.0f" % (
        n*h*w, co, ci, res, len(t), np.median(pro), np.median(setup), np.median(first-setup), np.median(pro-first), np.median(lp), np.median(epi), np.median(lds), np.median(epi-lds)))
