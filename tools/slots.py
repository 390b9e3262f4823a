"""Which blocks of a launch share a CU? (dispatch order vs. CU slots; dana_set_igemm_trace) usage: slots.py h"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: F401
from dana_amd import ops
from dana_amd._lib import lib
h = int(sys.argv[1])
dev = torch.device("cuda:0")
x = torch.randn(h * 128, 256, device=dev)
wt = torch.randn(128, 256, device=dev) * 0.05
run = lambda: ops.conv2d_nhwc(x, 1, h, 128, 256, wt, 128, 1, 1, 1, 0)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
lib().call("dana_set_igemm_trace", buf.data_ptr())
run()
torch.cuda.synchronize()
lib().call("dana_set_igemm_trace", None)
t = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8)
nb = int((t[:, 3] > 0).sum())
hw = t[:nb, 4]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
cuid = xcc * 1000 + se * 16 + cu
start = t[:nb, 6].astype(np.float64)  # wall clock at start (100 MHz)
start = (start - start.min()) / 100.0
print("blocks", nb, "distinct CUs", len(np.unique(cuid)))
first = {}
order = []
for b in range(nb):
    c = cuid[b]
    first.setdefault(c, []).append(b)
# for the first 24 blocks: which other block ids share the CU, and their start times
for b in list(range(0, 12)) + list(range(256, 262)) + list(range(512, 518)):
    if b < nb:
        mates = first[cuid[b]]
        print("block %4d xcc %d se %d cu %2d start %6.2f us  CU mates (id:start) %s" % (
            b, xcc[b], se[b], cu[b], start[b], " ".join("%d:%.1f" % (m, start[m]) for m in mates)))
# how many of blocks 0..255 share a CU with another block of 0..255?
for lo, hi in ((0, 256), (256, 512), (0, 512)):
    ids = cuid[lo:min(hi, nb)]
    u, c = np.unique(ids, return_counts=True)
    print("blocks [%d,%d): %d distinct CUs, max per CU %d" % (lo, hi, len(u), c.max() if len(c) else 0))
