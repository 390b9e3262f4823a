"""same-process A/B of a DAnARCNN attribute on the replayed bs-4 train-mode forward: one model + ProgramDAnA per value,
interleaved rounds, median GPU-side step interval. usage: python tools/ab_forward_attr.py ATTR v1 v2 ... [--rounds N]
(values: None / True / False / strings)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.program import ProgramDAnA
args = sys.argv[1:]
rounds = 4
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
attr, vals = args[0], [{"None": None, "True": True, "False": False}.get(v, v) for v in args[1:]]
dev = torch.device("cuda:0")
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
runs = {}
for v in vals:
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
    m.to(dev).train()
    setattr(m, attr, v)
    np.random.seed(0)
    runs[str(v)] = ProgramDAnA(m, *inputs)
res = {k: [] for k in runs}
for r in range(rounds):
    for name, run in runs.items():
        for _ in range(5):
            run(*run.inputs)
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True)]
        marks[0].record()
        for _ in range(30):
            run(*run.inputs)
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        torch.cuda.synchronize()
        iv = sorted(a.elapsed_time(b) for a, b in zip(marks, marks[1:]))
        res[name].append(iv[len(iv) // 2])
for name, v in res.items():
    print("%s = %-10s median forward ms per round: %s" % (attr, name, " ".join("%.3f" % x for x in v)))
