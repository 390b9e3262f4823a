"""debug: HIP model_backward vs oracle autograd in fp32 and fp64 (max-norm and L2 relative errors per parameter)"""
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import dana_amd
from dana_amd import synthetic as S, backward as BW
from oracle import model_ref as O

use_ba = len(sys.argv) > 1 and sys.argv[1] == "ba"
dev = torch.device("cuda:0")
B, way, shot, H, W = 2, 2, 3, 192, 256
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=use_ba, way=way, shot=shot, classes=["fg", "bg"])
sd = S.fill_state_dict(m.state_dict(), seed=21, profile="test")
m.load_state_dict(sd)
m.to(dev).train()
m.nms_inclusive = True
inputs = S.episode_inputs(B, way, shot, H, W, seed=22)
weights = (1.0, 0.5, 2.0, 1.5)


def trainable(k):
    if "bn" in k or "downsample.1" in k or "running_" in k or "num_batches" in k:
        return False
    return not (k.startswith("RCNN_base.0") or k.startswith("RCNN_base.1") or k.startswith("RCNN_base.4"))


def oracle(dtype):
    osd = {k: (v.clone().to(dtype).requires_grad_(True) if v.dtype.is_floating_point and trainable(k)
               else (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
    np.random.seed(33)
    ins = [t.to(dtype) if t.dtype.is_floating_point else t for t in inputs]
    out = O.forward(osd, *ins, training=True, n_way=way, n_shot=shot, use_ba=use_ba, nms_inclusive=True,
                    differentiable=True)
    sum(wt * l for wt, l in zip(weights, out[3:7])).backward()
    return osd, out


o32, out32 = oracle(torch.float32)
try:
    o64, out64 = oracle(torch.float64)
    same = np.array_equal(out64[7].numpy(), out32[7].numpy()) and np.allclose(out64[0].numpy(), out32[0].numpy(), atol=1e-2)
except Exception as e:  # noqa
    print("fp64 oracle failed:", repr(e))
    o64, same = None, False
print("fp64 oracle sampled the same rois:", same)
m.save_for_backward = True
np.random.seed(33)
with torch.no_grad():
    res = m(*[t.to(dev) for t in inputs])
print("labels equal:", np.array_equal(res[7].cpu().numpy(), out32[7].numpy()))
BW.model_backward(m, weights)
torch.cuda.synchronize()
params = dict(m.named_parameters())
gmax = max(v.grad.abs().max().item() for v in o32.values() if v.dtype.is_floating_point and v.requires_grad)
rows = []
for k, v in o32.items():
    if not (v.dtype.is_floating_point and v.requires_grad):
        continue
    g = params[k].grad.cpu().double()
    r32 = v.grad.double()
    row = [k, r32.abs().max().item(), (g - r32).abs().max().item() / (r32.abs().max().item() + 1e-3 * gmax),
           (g - r32).norm().item() / (r32.norm().item() + 1e-12)]
    if same:
        r64 = o64[k].grad
        row += [(g - r64).abs().max().item() / (r64.abs().max().item() + 1e-3 * gmax),
                (r32 - r64).abs().max().item() / (r64.abs().max().item() + 1e-3 * gmax)]
    rows.append(row)
rows.sort(key=lambda r: -r[2])
print("%-44s %10s %10s %10s %10s %10s" % ("param", "max|ref|", "hip-o32", "l2 rel", "hip-o64", "o32-o64"))
for r in rows[:25]:
    print("%-44s " % r[0] + " ".join("%10.3e" % x for x in r[1:]))
