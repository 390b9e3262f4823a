"""the proposal layer's NMS pair on the bench's own proposals (4 images, 12000 boxes each, thr 0.7, 2000 keeps)"""
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
    m(*inputs)
heads = m._capture["rpn_heads"]
corr, B, fh, fw = m._capture["corr"]
hw, nh = fh * fw, 72
plan = m._get_plan()
props, scores = ops.rpn_decode(heads, (hw * nh, 1, nh), False, heads.view(-1)[24:], (hw * nh, 1, nh), inputs[1].float().contiguous(),
                               plan["anchors"], B, 12, fh, fw, 16)
order, _ = ops.sort_desc(scores)
top = torch.stack([props[i][order[i, :12000].long()] for i in range(B)], 0).contiguous()
f = lambda: ops.nms_sorted(top, 0.7, False, 2000)  # noqa: E731
for _ in range(5):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    f()
e1.record(); torch.cuda.synchronize()
print("nms pair (mask + scan), 4 x 12000 boxes -> 2000 keeps: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 30))
