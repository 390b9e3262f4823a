"""One fresh process: the eager bs-4 train-mode forward with the two TRUNK roles (the caller's stream = query trunk, `support`)
on every ordered pair of the process's streams (the default stream + the role streams; the other roles take the remaining
streams in order), interleaved rounds, median GPU-side step interval. Run K times on one box (tools/population_roles.sh):
is the "slow population" of profiles/r5_step_time_populations.txt a property of the PAIR of hardware queues the two trunks
sit on?   usage: python tools/population_roles.py [rounds]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.dana import DAnARCNN
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
np.random.seed(0)
with torch.no_grad():
    for _ in range(5):
        m(*inputs)
torch.cuda.synchronize()
roles = list(DAnARCNN._ROLE_STREAMS)
pool = {"default": torch.cuda.current_stream(dev)}
pool.update({r: DAnARCNN._role_streams[(r, str(dev))] for r in roles})
names = ["default", "support", "targets", "layer4", "neg_head"]  # (wgrad's stream stays the rpn side role's)
res = {}
for _ in range(rounds):
    for a in names:
        for b in names:
            if a == b:
                continue
            rest = [n for n in names if n not in (a, b)]
            assign = {"support": b, "targets": rest[0], "layer4": rest[1], "neg_head": rest[2], "wgrad": "wgrad"}
            for role, src in assign.items():
                DAnARCNN._role_streams[(role, str(dev))] = pool[src]
            with torch.no_grad(), torch.cuda.stream(pool[a]):
                for _ in range(3):
                    m(*inputs)
                torch.cuda.synchronize()
                marks = [torch.cuda.Event(enable_timing=True)]
                marks[0].record()
                for _ in range(24):
                    m(*inputs)
                    marks.append(torch.cuda.Event(enable_timing=True))
                    marks[-1].record()
                torch.cuda.synchronize()
            iv = sorted(x.elapsed_time(y) for x, y in zip(marks, marks[1:]))
            res.setdefault((a, b), []).append(iv[len(iv) // 2])
print("pid %d  median forward ms, rows = the caller's stream, columns = the support trunk's stream (%s)" % (os.getpid(), " ".join(names)))
for a in names:
    print("   %-9s " % a + "  ".join("%-11s" % ("/".join("%.2f" % x for x in res[(a, b)]) if a != b else "-") for b in names), flush=True)
