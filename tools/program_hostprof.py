"""Host time of one replayed training iteration (program.ProgramTrainer), by part: program 1, the host round trip's own
work (draws + upload, the blocked wait excluded), program 2 split into its C runs / torch ops / host callbacks.
usage: python tools/program_hostprof.py [iters]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402
from dana_amd import ops, program, synthetic as S  # noqa: E402
from dana_amd.trainer import Trainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, 1e-5)
np.random.seed(0)
pt = program.ProgramTrainer(tr, *inputs)
print("program 1:", pt.p1.stats)
print("program 2:", pt.p2.stats)
for _ in range(3):
    pt.step(*pt.inputs)
torch.cuda.synchronize()
acc = dict(p1=0.0, draws=0.0, p2=0.0)
for _ in range(iters):
    torch.cuda.synchronize()
    for fb, _, _ in tr.groups:
        fb.zero_grad_bookkeeping()
    t0 = time.perf_counter()
    pt.p1.run()
    t1 = time.perf_counter()
    ops.HOST_WAIT[0] = 0.0
    ops.draw_and_upload(pt.req, dev, static=pt.drawn)
    t2 = time.perf_counter()
    pt.p2.run()
    t3 = time.perf_counter()
    pt._after_step()
    acc["p1"] += t1 - t0
    acc["draws"] += t2 - t1 - ops.HOST_WAIT[0]
    acc["p2"] += t3 - t2
torch.cuda.synchronize()
print("host ms per iteration: program 1 %.2f, draws + upload (wait excluded) %.2f, program 2 %.2f" % tuple(
    1e3 * acc[k] / iters for k in ("p1", "draws", "p2")))
# program 2 by entry kind
kinds = {program._CRUN: "C runs", program._ATEN: "torch ops", program._HOST: "host callbacks"}
tk = {k: 0.0 for k in kinds}
p2 = pt.p2
for _ in range(iters):
    torch.cuda.synchronize()
    for fb, _, _ in tr.groups:
        fb.zero_grad_bookkeeping()
    pt.p1.run()
    ops.draw_and_upload(pt.req, dev, static=pt.drawn)
    cur = ops._get_cur(ops._raw_device())
    last = cur
    for kind, a, b in p2._plan:
        t0 = time.perf_counter()
        if kind == program._CRUN:
            assert p2._crun(p2._c, a, b) == 0
        elif kind == program._ATEN:
            st = b[2]
            if st != last:
                ops._set_cur(stream_id=st[0], device_index=st[1], device_type=st[2])
                last = st
            a(*b[0], **b[1])
        else:
            if b != last:
                ops._set_cur(stream_id=b[0], device_index=b[1], device_type=b[2])
                last = b
            a()
        tk[kind] += time.perf_counter() - t0
    ops._set_cur(stream_id=cur[0], device_index=cur[1], device_type=cur[2])
    pt._after_step()
torch.cuda.synchronize()
print("program 2 by kind (ms per iteration): " + ", ".join("%s %.2f" % (kinds[k], 1e3 * tk[k] / iters) for k in kinds))
