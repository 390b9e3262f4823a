"""per-launch table of the MFMA contractions of one training step (HIP events on the launch stream, single stream)"""
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import dana_amd
from dana_amd import ops, synthetic as S
from dana_amd.trainer import Trainer

dev = torch.device("cuda:0")
B, way, shot = 4, 2, 3
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=way, shot=shot, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(B, way, shot, 600, 1000, seed=1996)]
tr = Trainer(m, 1e-5)
np.random.seed(0)
for _ in range(3):
    tr.step(*inputs)
torch.cuda.synchronize()
ops.PROFILE = []
K = 5
import time
t0 = time.perf_counter()
for _ in range(K):
    tr.step(*inputs)
torch.cuda.synchronize()
print("step (with per-launch events): %.2f ms" % ((time.perf_counter() - t0) / K * 1e3))
prof, ops.PROFILE = ops.PROFILE, None
per = {}
for tag, f, e0, e1, _nb, _ex in prof:
    a = per.setdefault(tag, [0.0, 0.0, 0])
    a[0] += f / K
    a[1] += e0.elapsed_time(e1) / K
    a[2] += 1
tot = sum(v[1] for v in per.values())
print("contractions: %.2f ms/step, %.1f GF/step, %.1f TF/s" % (tot, sum(v[0] for v in per.values()) / 1e9,
                                                               sum(v[0] for v in per.values()) / tot / 1e9))
kinds = {}
for tag, (f, t, c) in per.items():
    k = kinds.setdefault(tag.split(" ")[0], [0.0, 0.0])
    k[0] += f
    k[1] += t
for k, (f, t) in sorted(kinds.items(), key=lambda x: -x[1][1]):
    print("  %-10s %8.2f ms %9.1f GF %7.1f TF/s" % (k, t, f / 1e9, f / t / 1e9))
for tag, (f, t, c) in sorted(per.items(), key=lambda x: -x[1][1])[:45]:
    print("%-46s x%-3d %8.1f us %8.2f GF %7.1f TF/s" % (tag, c // K, t * 1e3, f / 1e9, f / t / 1e9))
t0 = time.perf_counter()
for _ in range(K):
    tr.step(*inputs)
torch.cuda.synchronize()
print("step: %.2f ms" % ((time.perf_counter() - t0) / K * 1e3))
