"""per-iteration wall time of Trainer.step (host-synchronised after each), plus allocator stats"""
import sys, time
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
if len(sys.argv) > 1 and sys.argv[1] == "single":
    m._single_stream = True
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, 1e-5)
np.random.seed(0)
ts = []
for i in range(16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(*inputs)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
st = torch.cuda.memory_stats()
print("ms:", " ".join("%.1f" % t for t in ts))
print("reserved %.2f GB, allocated peak %.2f GB, num_alloc_retries %d, cudaMalloc calls %d" % (
    st["reserved_bytes.all.current"] / 2**30, st["allocated_bytes.all.peak"] / 2**30, st["num_alloc_retries"],
    st["num_device_alloc"]))
t0 = time.perf_counter()
for i in range(10):
    tr.step(*inputs)
torch.cuda.synchronize()
print("pipelined: %.2f ms/step" % ((time.perf_counter() - t0) * 100))
