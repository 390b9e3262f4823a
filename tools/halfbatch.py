"""Two half-batches in flight: one hipGraph of the bs-4 train-mode forward (device RNG: the forward is ONE graph) against
two bs-2 graphs replayed concurrently on two streams. usage: halfbatch.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.graphs import GraphedDAnA
dev = torch.device("cuda:0")


def model():
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
    m.to(dev).train()
    m.device_rng = True
    return m


full = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
m4 = model()
g4 = GraphedDAnA(m4, *full)
halves = []
for h in range(2):
    mh = model()
    inp = [t[2 * h:2 * h + 2].contiguous() for t in full]
    halves.append((GraphedDAnA(mh, *inp), torch.cuda.Stream(device=dev)))


def one():
    g4(*g4.inputs)


def two():
    cur = torch.cuda.current_stream()
    for g, st in halves:
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            g(*g.inputs)
    for _, st in halves:
        cur.wait_stream(st)


for name, fn in (("one bs-4 graph", one), ("two bs-2 graphs, two streams", two), ("one bs-4 graph", one), ("two bs-2 graphs, two streams", two)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print("%-32s %.3f ms per 4 episodes = %.1f query-images/s" % (name, 1e3 * dt, 4 / dt), flush=True)
