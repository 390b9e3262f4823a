"""GPU idle time at the training forward's ONE host round trip, launch-program replay and eager, WITHOUT a profiler:
events on the caller's stream behind program 1 and in front of program 2, host time of the parts of draw_and_upload.
usage: python tools/roundtrip_gap.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops, synthetic as S
from dana_amd.program import ProgramDAnA
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
np.random.seed(0)
run = ProgramDAnA(m, *inputs)
med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
T = {}


def timed(name):
    fn = getattr(ops, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T.setdefault(name, []).append(1e3 * (time.perf_counter() - t0))
        return r
    setattr(ops, name, w)


for nm in ("anchor_target_draw", "proposal_target_draw", "_pinned_upload"):
    timed(nm)
p1s, gaps, p2s, hosts, blocked, steps = [], [], [], [], [], []
for it in range(40):
    e0, e1, e2, e3 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e0.record()
    run.p1.run()
    e1.record()
    t0 = time.perf_counter()
    hw0 = ops.HOST_WAIT[0]
    ops.draw_and_upload(run.req, dev, static=run.drawn)
    t1 = time.perf_counter()
    e2.record()
    run.p2.run()
    e3.record()
    torch.cuda.synchronize()
    if it >= 8:
        p1s.append(e0.elapsed_time(e1)); gaps.append(e1.elapsed_time(e2)); p2s.append(e2.elapsed_time(e3))
        hosts.append(1e3 * (t1 - t0)); blocked.append(1e3 * (ops.HOST_WAIT[0] - hw0)); steps.append(e0.elapsed_time(e3))
print("program replay: program 1 %.3f ms | GPU idle at the round trip %.3f ms | program 2 %.3f ms | step %.3f ms | host in draw_and_upload "
      "%.3f ms of which blocked %.3f" % (med(p1s), med(gaps), med(p2s), med(steps), med(hosts), med(blocked)))
for k, v in T.items():
    print("  host %-22s median %.3f ms" % (k, med(v[8:])))
ops.GAP_EVENTS = []
with torch.no_grad():
    for _ in range(30):
        m(*inputs)
torch.cuda.synchronize()
v = [a.elapsed_time(b) for a, b in ops.GAP_EVENTS[8:]]
print("eager forward: GPU idle at the round trip %.3f ms (median of %d)" % (med(v), len(v)))
