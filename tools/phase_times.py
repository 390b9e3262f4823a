"""GPU-side duration of each phase of the default (two-stream, eager) train-mode forward: events recorded on the caller's
stream at the phase marks of DAnARCNN._forward_gen (`model._gpu_events`), averaged over a few steps, unprofiled.
usage: python tools/phase_times.py [steps] [--graph]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402
from dana_amd import synthetic as S  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
np.random.seed(1996)
with torch.no_grad():
    for _ in range(5):
        m(*inputs)
    torch.cuda.synchronize()
    acc, total = {}, 0.0
    order = []
    for _ in range(steps):
        m._gpu_events = []
        m(*inputs)
        torch.cuda.synchronize()
        ev = m._gpu_events
        for (n0, e0), (n1, e1) in zip(ev, ev[1:]):
            if n1 not in acc:
                order.append(n1)
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
        total += ev[0][1].elapsed_time(ev[-1][1])
    m._gpu_events = None
for n in order:
    print("%-55s %7.3f ms" % (n, acc[n] / steps))
print("%-55s %7.3f ms" % ("begin -> last mark", total / steps))
