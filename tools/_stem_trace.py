import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops
from dana_amd._lib import lib
dev = torch.device("cuda:0")
n, h, w = 4, 600, 1000
x = torch.randn(n * h * w, 4, device=dev); x[:, 3] = 0
wt = torch.randn(64, 7 * 8 * 4, device=dev) * 0.05
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
run = lambda: ops.conv2d_nhwc(x, n, h, w, 4, wt, 64, 7, 7, 2, 3, scale=sc, shift=sh, relu=True, stem=True)
for _ in range(5): run()
torch.cuda.synchronize()
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib().call("dana_set_igemm_trace", buf.data_ptr())
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib().call("dana_set_igemm_trace", None)
t = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8); t = t[t[:, 3] > 0].astype(np.float64)
pro, lp, epi, tot = t[:,1]-t[:,0], t[:,2]-t[:,1], t[:,3]-t[:,2], t[:,3]-t[:,0]
print("stem: %d blocks, %.1f us; cycles p50: prologue %.0f loop %.0f (14 steps: %.0f/step) epilogue %.0f total %.0f" % (len(t), e0.elapsed_time(e1)*1e3, np.median(pro), np.median(lp), np.median(lp)/14, np.median(epi), np.median(tot)))
ghz = np.median(tot / np.maximum((t[:,5]-t[:,6]) * 10.0, 1.0)); print("clock %.2f GHz" % ghz)
raw = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8); raw = raw[raw[:, 3] > 0]
setup = raw[:, 4].astype(np.float64); first = (raw[:, 7] >> np.uint64(32)).astype(np.float64); lds = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
print("prologue split: setup (args, addresses, masks) %.0f | first tiles landed + split + barrier %.0f | second stage to loop %.0f" % (np.median(setup), np.median(first - setup), np.median(pro - first)))
print("epilogue split: residual issue + acc->LDS + barrier %.0f | scale/shift loads, LDS reads, stores %.0f" % (np.median(lds), np.median(epi - lds)))
