"""phase_times.py for the launch-program replay: the phase marks' events are recorded INTO the programs and re-recorded by
every replay, so the GPU-side phase durations of the replayed (not host-bound) forward can be read. usage: [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.program import ProgramDAnA
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
np.random.seed(1996)
with torch.no_grad():
    for _ in range(3):
        m(*inputs)
m._gpu_events = []
run = ProgramDAnA(m, *inputs, warmup=0)
ev = m._gpu_events
m._gpu_events = None
acc = {}
for _ in range(5):
    run(*run.inputs)
for _ in range(steps):
    run(*run.inputs)
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(ev, ev[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
for n, _ in ev[1:]:
    print("%-78s %7.3f ms" % (n, acc[n] / steps))
print("%-78s %7.3f ms" % ("begin -> last mark", sum(acc.values()) / steps))
