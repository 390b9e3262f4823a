"""Where does the proposal layer's greedy NMS stop? (bench inputs, train mode: pre 12000 / post 2000, thr 0.7)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops, synthetic as S
from dana_amd.config import cfg
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (600, 1000)
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, H, W, seed=1996)]
m._capture = {}
np.random.seed(1)
with torch.no_grad():
    m(*inputs)
heads = m._capture["rpn_heads"]
corr, B, fh, fw = m._capture["corr"]
A, nh = 12, 72
hw = fh * fw
plan = m._get_plan()
props, scores = ops.rpn_decode(heads, (hw * nh, 1, nh), False, heads.view(-1)[24:], (hw * nh, 1, nh), inputs[1].float().contiguous(),
                               plan["anchors"], B, A, fh, fw, 16)
order, _ = ops.sort_desc(scores)
for pre, post in ((12000, 2000), (6000, 300)):
    top = torch.stack([props[i][order[i, :pre].long()] for i in range(B)], 0).contiguous()
    keep, num = ops.nms_sorted(top, 0.7, False, post)
    torch.cuda.synchronize()
    for i in range(B):
        k = int(num[i])
        print("pre %d post %d image %d: kept %d, last kept position %d" % (pre, post, i, k, int(keep[i, k - 1])))
