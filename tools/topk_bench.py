"""The proposal layer's sort, alone (dana_topk_desc): the single-workgroup select + LDS sort (dana_set_sort_mode(2)), the
multi-workgroup sample sort (mode 1) and the dispatch (mode 0) on the bench's shapes. usage: topk_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dana_amd import ops  # noqa: E402
from dana_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for B, n, topn in ((4, 21546, 12000), (1, 21546, 6000), (2, 37800, 12000), (2, 50400, 12000), (1, 300, 300)):
    s = torch.from_numpy(rng.uniform(size=(B, n)).astype(np.float32)).to(dev)
    row = []
    for mode in (2, 1, 0):
        lib().call("dana_set_sort_mode", mode)
        for _ in range(5):
            ops.topk_desc(s, topn)
        torch.cuda.synchronize()
        ts = []
        for _ in range(50):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.topk_desc(s, topn)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        row.append(sorted(ts)[len(ts) // 2])
    lib().call("dana_set_sort_mode", 0)
    print("B=%d n=%d topn=%d: single-workgroup %.1f us, sample sort %.1f us, dispatch %.1f us" % (B, n, topn, row[0], row[1], row[2]))
