"""what-if timing of the eager training iteration (NOT numerically valid variants): where would the wall clock go if a
piece were free? usage: train_exp.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops, synthetic as S
from dana_amd.trainer import Trainer
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
tr = Trainer(m, lr=1e-5)
np.random.seed(0)


def timeit(tag, k=12):
    for _ in range(4):
        tr.step(*inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        tr.step(*inputs)
    torch.cuda.synchronize()
    print("%-44s %.3f ms/iteration" % (tag, 1e3 * (time.perf_counter() - t0) / k), flush=True)


timeit("baseline")
real_bwd = ops.roi_align_backward
ops.roi_align_backward = lambda grad, rois, sc, ph, pw, b, c, h, w, sr, layout=0: torch.zeros(
    (b, h, w, c) if layout else (b, c, h, w), device=grad.device)
timeit("RoIAlign backward free")
ops.roi_align_backward = real_bwd
real_step = tr.optimizer_step


def no_bump():
    e = m._epoch
    real_step()
    m._epoch = e


tr.optimizer_step = no_bump
timeit("no weight re-derivation (stale copies)")
tr.optimizer_step = real_step
real_wgrad, real_wino = ops.conv2d_wgrad, ops.conv3x3_wgrad_winograd
ops.conv2d_wgrad = lambda grad_out, x, batch, in_h, in_w, cin, cout, kh, kw, stride, pad, **kw_: (
    kw_.get("out") if kw_.get("out") is not None else torch.zeros(cout, kh * kw * cin, device=x.device))
ops.conv3x3_wgrad_winograd = lambda grad_out, x, batch, h, w, cin, cout, **kw_: (
    kw_.get("out") if kw_.get("out") is not None else torch.zeros(cout, 9 * cin, device=x.device))
try:
    timeit("conv weight gradients free")
except Exception as e:  # noqa: BLE001
    print("wgrad stub failed:", e)
ops.conv2d_wgrad, ops.conv3x3_wgrad_winograd = real_wgrad, real_wino
m._single_stream = True
timeit("single stream")
