"""RoIAlign forward at the bench's shape (512 rois from the model itself, 38x63x1024 map inside the 2048-wide buffer)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops, synthetic as S
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
m._capture = {}
np.random.seed(1)
with torch.no_grad():
    out = m(*inputs)
corr, B, fh, fw = m._capture["corr"]
rois = out[0].reshape(-1, 5).contiguous()
plan = m._get_plan()
f = lambda: ops.roi_align_forward_nhwc(corr, B, fh, fw, 1024, 2048, rois, 1.0 / 16.0, 7, 0, pe=plan["pe49"])  # noqa: E731
for _ in range(5):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
alg = 4.0 * (rois.size(0) * 1024 * 49 * 2 + B * 1024 * fh * fw + 5 * rois.size(0))
print("roi_align_fwd_nhwc: %d rois, %.1f us, algorithmic %.0f MB -> %.2f TB/s" % (rois.size(0), us, alg / 1e6, alg / us / 1e6))
