"""How long does the GPU idle at the training forward's host round trip? hipGraph replay, host RNG: events behind graph 1
and in front of graph 2. usage: sync_gap.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops, synthetic as S
from dana_amd.graphs import GraphedDAnA
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
g = GraphedDAnA(m, *inputs)
np.random.seed(0)
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
g1s, gaps, g2s, hosts = [], [], [], []
for it in range(30):
    cur = torch.cuda.current_stream()
    e0, e1, e2, e3 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    g.side.wait_stream(cur)
    with torch.cuda.stream(g.side):
        g.g0.replay()
    e0.record()
    g.g1.replay()
    e1.record()
    t0 = time.perf_counter()
    hw0 = ops.HOST_WAIT[0]
    ops.draw_and_upload(g.req, g.drawn.device, static=g.drawn)
    t1 = time.perf_counter()
    T.setdefault("blocked", []).append(1e3 * (ops.HOST_WAIT[0] - hw0))
    cur.wait_stream(g.side)
    e2.record()
    g.g2.replay()
    e3.record()
    torch.cuda.synchronize()
    if it >= 5:
        g1s.append(e0.elapsed_time(e1)); gaps.append(e1.elapsed_time(e2)); g2s.append(e2.elapsed_time(e3)); hosts.append(1e3 * (t1 - t0))
med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
print("graph 1 %.3f ms | GPU idle at the round trip %.3f ms | graph 2 %.3f ms | host in draw_and_upload %.3f ms (incl. waiting for graph 1)" % (
    med(g1s), med(gaps), med(g2s), med(hosts)))
for k, v in T.items():
    v = v[10:] if k == "_pinned_upload" else v[5:]
    print("  host %-22s median %.3f ms (calls per step %d)" % (k, med(v), 2 if k == "_pinned_upload" else 1))

# the same round trip in the EAGER forward
ops.GAP_EVENTS = []
with torch.no_grad():
    for _ in range(25):
        m(*inputs)
torch.cuda.synchronize()
v = [a.elapsed_time(b) for a, b in ops.GAP_EVENTS[5:]]
print("eager forward: GPU idle at the round trip %.3f ms (median of %d)" % (med(v), len(v)))
