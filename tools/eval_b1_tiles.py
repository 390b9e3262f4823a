"""eval_b1 (BASELINE configs[0] shape on the HIP path) with the split kernel's tile shape forced, hipGraph replay:
is the dispatcher's choice the best one for the bs-1 critical path? usage: python tools/eval_b1_tiles.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops, synthetic as S
from dana_amd.graphs import GraphedDAnA
dev = torch.device("cuda:0")
inputs = [t.to(dev) for t in S.episode_inputs(1, 1, 3, 600, 1000, seed=1996)]
def med_step(run, k=30):
    for _ in range(5): run(*run.inputs)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True)]; marks[0].record()
    for _ in range(k):
        run(*run.inputs); marks.append(torch.cuda.Event(enable_timing=True)); marks[-1].record()
    torch.cuda.synchronize()
    iv = sorted(a.elapsed_time(b) for a, b in zip(marks, marks[1:]))
    return iv[len(iv) // 2]
res = {}
for rnd in range(2):
    for name, tile in (("dispatcher", 0), ("128x64", 2), ("64x64", 3), ("128x128", 4), ("64x128", 5)):
        ops.force_tile(tile)
        m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
        m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
        m.to(dev).eval()
        run = GraphedDAnA(m, *inputs)
        res.setdefault(name, []).append(med_step(run))
        del run, m
ops.force_tile(0)
for k, v in res.items():
    print("%-12s eval_b1 ms: %s" % (k, " ".join("%.3f" % x for x in v)))
