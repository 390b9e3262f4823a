"""why are some training iterations slow? per-iteration time + number of device allocations made during it"""
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
np.random.seed(1996)
with torch.no_grad():
    for _ in range(25):
        m(*inputs)
tr = Trainer(m, 1e-5)
out = []
for i in range(30):
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    f0 = torch.cuda.memory_stats()["num_device_free"]
    g0 = sum(s["collections"] for s in gc.get_stats())
    t0 = time.perf_counter()
    tr.step(*inputs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats()
    out.append("%.0f(a%d,f%d,gc%d)" % (dt, st["num_device_alloc"] - a0, st["num_device_free"] - f0,
                                      sum(s["collections"] for s in gc.get_stats()) - g0))
print(" ".join(out))
print("reserved %.2f GB" % (torch.cuda.memory_stats()["reserved_bytes.all.current"] / 2**30))
