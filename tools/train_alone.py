"""eager training iteration and forward, timed alone in a fresh process (3 x 15 iterations, 2 x 30 forwards): the tool
behind profiles/r5_role_streams.md. usage: stream_alias_ab.py TAG"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.trainer import Trainer
cfgname = sys.argv[1]
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, lr=1e-5)
np.random.seed(0)


def timeit(fn, k=15):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


def fwd():
    with torch.no_grad():
        m(*inputs)


print("%-8s train %s ms | forward %s ms" % (cfgname, " ".join("%.2f" % timeit(lambda: tr.step(*inputs)) for _ in range(3)),
                                            " ".join("%.3f" % timeit(fwd, 30) for _ in range(2))), flush=True)
