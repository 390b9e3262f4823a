"""which host call blocks when a training iteration is slow? times every C-ABI call and every torch.empty"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import dana_amd
from dana_amd import synthetic as S, _lib, ops
from dana_amd.trainer import Trainer

dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, 1e-5)
np.random.seed(0)
for _ in range(4):
    tr.step(*inputs)
torch.cuda.synchronize()
L = _lib.lib()
orig_call = L.call
log = []
step = [0]


def timed_call(name, *a):
    t0 = time.perf_counter()
    orig_call(name, *a)
    dt = time.perf_counter() - t0
    if dt > 5e-4:
        log.append((step[0], name, dt * 1e3))


L.call = timed_call
orig_empty = torch.empty


def timed_empty(*a, **k):
    t0 = time.perf_counter()
    r = orig_empty(*a, **k)
    dt = time.perf_counter() - t0
    if dt > 5e-4:
        log.append((step[0], "torch.empty%s" % (tuple(a[0]) if a and isinstance(a[0], (tuple, list)) else ""), dt * 1e3))
    return r


torch.empty = timed_empty
ts = []
for i in range(40):
    step[0] = i
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(*inputs)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms:", " ".join("%.0f" % t for t in ts))
med = sorted(ts)[len(ts) // 2]
slow = [i for i, t in enumerate(ts) if t > med + 5]
print("slow steps:", slow)
for s_, name, dt in log:
    print("step %2d%s  %-40s %.2f ms" % (s_, "*" if s_ in slow else " ", name, dt))
