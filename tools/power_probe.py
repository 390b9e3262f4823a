"""Is the step limited by what the chip may draw? The SAME train-mode forward (same launches, same shapes, same tiles) on the
bench's N(0,1)-like data and with every floating-point input and parameter zeroed: the instruction streams are identical, only
the toggle rate of the multiplier arrays and data paths differs (profiles/r2_gemm_power.md showed one large GEMM running at 2.3
GHz on zeros and 1.5 GHz on N(0,1) operands). usage: python tools/power_probe.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402
from dana_amd import synthetic as S  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")


def run(kind):
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=11, profile="test")
    inputs = S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)
    if kind == "zeros":
        sd = {k: (torch.zeros_like(v) if v.dtype.is_floating_point and "running_var" not in k else v) for k, v in sd.items()}
        inputs = [inputs[0] * 0, inputs[1], inputs[2], inputs[3], inputs[4] * 0]
    m.load_state_dict(sd)
    m.to(dev).train()
    din = [t.to(dev) for t in inputs]
    np.random.seed(1996)
    with torch.no_grad():
        for _ in range(8):
            m(*din)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(*din)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("| data | ms per train-mode forward (bs 4, eager, two streams) |")
print("|---|---|")
for rep in range(2):
    for kind in ("bench", "zeros"):
        print("| %s | %.3f |" % (kind, run(kind)), flush=True)
