"""where is the host when a training iteration stalls? a watchdog thread samples the Python stacks of all threads every
millisecond; for slow iterations it prints the frames that were on top during the excess time"""
import sys, time, threading, traceback, collections
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.trainer import Trainer
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
m._single_stream = "single" in sys.argv
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, 1e-5)
np.random.seed(0)
for _ in range(5):
    tr.step(*inputs)
torch.cuda.synchronize()
samples = []  # (time, {thread: top frames})
stop = [False]
main_id = threading.get_ident()


def watch():
    while not stop[0]:
        now = time.perf_counter()
        fr = sys._current_frames()
        snap = {}
        for tid, f in fr.items():
            if tid == threading.get_ident():
                continue
            st = traceback.extract_stack(f)[-3:]
            snap[tid] = " <- ".join("%s:%d %s" % (x.filename.split("/")[-1], x.lineno, x.name) for x in reversed(st))
        samples.append((now, snap))
        time.sleep(0.001)


th = threading.Thread(target=watch, daemon=True)
th.start()
spans = []
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(*inputs)
    torch.cuda.synchronize()
    spans.append((t0, time.perf_counter()))
stop[0] = True
th.join()
ds = [(b - a) * 1e3 for a, b in spans]
med = sorted(ds)[len(ds) // 2]
print("ms:", " ".join("%.0f" % d for d in ds))
for i, (a, b) in enumerate(spans):
    if (b - a) * 1e3 > med + 6:
        cnt = collections.Counter()
        for t, snap in samples:
            if a <= t <= b:
                for tid, s in snap.items():
                    cnt[("main " if tid == main_id else "other") + " " + s] += 1
        print("--- slow iteration %d (%.0f ms): most sampled stacks" % (i, (b - a) * 1e3))
        for s, c in cnt.most_common(6):
            print("   %4d  %s" % (c, s))
# reference: a normal iteration
i = ds.index(med)
a, b = spans[i]
cnt = collections.Counter()
for t, snap in samples:
    if a <= t <= b:
        for tid, s in snap.items():
            cnt[("main " if tid == main_id else "other") + " " + s] += 1
print("--- median iteration %d (%.0f ms)" % (i, med))
for s, c in cnt.most_common(6):
    print("   %4d  %s" % (c, s))
