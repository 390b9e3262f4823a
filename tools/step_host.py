"""host-side phase clock of Trainer.step: when does each phase RETURN on the host (no syncs added), pipelined steps"""
import sys, time, gc
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import dana_amd
from dana_amd import synthetic as S
from dana_amd.trainer import Trainer

dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, 1e-5)
np.random.seed(0)
if "nogc" in sys.argv:
    gc.disable()
for _ in range(4):
    tr.step(*inputs)
torch.cuda.synchronize()
rows = []
T0 = time.perf_counter()
for i in range(12):
    t = [time.perf_counter()]
    tr.zero_grad()
    t.append(time.perf_counter())
    with torch.enable_grad():
        out = m(*inputs)
        loss = out[3].mean() + out[4].mean() + out[5].mean() + out[6].mean()
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    tr.optimizer_step()
    t.append(time.perf_counter())
    rows.append([(b - a) * 1e3 for a, b in zip(t, t[1:])])
torch.cuda.synchronize()
print("total %.2f ms/step" % ((time.perf_counter() - T0) / 12 * 1e3))
print("zero  fwd   bwd   opt (host ms)")
for r in rows:
    print(" ".join("%5.1f" % x for x in r))
