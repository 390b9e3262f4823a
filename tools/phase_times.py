"""GPU time of the forward's phases on the caller's stream (the model's own `mark()` events), multi-stream eager step"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
np.random.seed(0)
with torch.no_grad():
    for _ in range(5):
        m(*inputs)
    acc = {}
    K = 20
    for _ in range(K):
        m._gpu_events = []
        m(*inputs)
        torch.cuda.synchronize()
        ev = m._gpu_events
        for (n0, e0), (n1, e1) in zip(ev, ev[1:]):
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1) / K
    m._gpu_events = None
tot = sum(acc.values())
for k, v in acc.items():
    print("%-55s %7.3f ms" % (k, v))
print("%-55s %7.3f ms" % ("total (begin .. rcnn losses)", tot))
