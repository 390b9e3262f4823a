"""40 training iterations, per-iteration wall time (sync after each): prints the pattern of slow iterations"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.trainer import Trainer
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
if "single" in sys.argv:
    m._single_stream = True
if "devrng" in sys.argv:
    m.device_rng = True
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, 1e-5)
np.random.seed(0)
for _ in range(5):
    tr.step(*inputs)
import gc
if "nogc" in sys.argv:
    gc.disable()
if "freeze" in sys.argv:
    gc.collect()
    gc.freeze()
ts = []
for i in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(*inputs)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
med = sorted(ts)[20]
print("median %.1f mean %.1f | %s" % (med, sum(ts) / len(ts), " ".join("%d" % round(t) if t > med + 3 else "." for t in ts)))
