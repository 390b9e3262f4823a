"""same-process A/B of the saving forward's data-gradient-weight prefetch (DAnARCNN.prefetch_dgrad): two models, two
ProgramTrainers, interleaved rounds of the replayed bs-4 training iteration. usage: python tools/ab_prefetch.py [rounds]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.program import ProgramTrainer
from dana_amd.trainer import Trainer
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
runs = {}
for name, pf in (("prefetch on wgrad", "wgrad"), ("prefetch on layer4", "layer4"), ("prefetch on neg_head", "neg_head"),
                 ("prefetch on targets", "targets"), ("derived at the backward's start", None)):
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
    m.to(dev).train()
    m.prefetch_dgrad = pf
    np.random.seed(0)
    runs[name] = ProgramTrainer(Trainer(m, 1e-5), *inputs)
res = {k: [] for k in runs}
for r in range(rounds):
    for name, pt in runs.items():
        for _ in range(3):
            pt.step(*pt.inputs)
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True)]
        marks[0].record()
        for _ in range(15):
            pt.step(*pt.inputs)
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        torch.cuda.synchronize()
        iv = sorted(a.elapsed_time(b) for a, b in zip(marks, marks[1:]))
        res[name].append(iv[len(iv) // 2])
for name, v in res.items():
    print("%-36s median iteration ms per round: %s" % (name, " ".join("%.3f" % x for x in v)))
