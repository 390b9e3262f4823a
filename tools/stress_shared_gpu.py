"""Two of these at once on ONE GPU (python tools/stress_shared_gpu.py graph & python tools/stress_shared_gpu.py graph) must
both print the same loss: contention lets a side stream run far ahead of / behind the caller's stream, which is how two
allocation-order hazards of round 3 showed up (the Trainer's shared two-stream trunk buffers, the early upload of the
anchor draws). MT=0: the default trunk; MF=0..3: merge_from. usage: stress_shared_gpu.py eager|graph|train|gtrain
(train / gtrain: five iterations of Trainer.step / GraphedTrainer.step from fixed inputs and draws -> a parameter checksum)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.graphs import GraphedDAnA
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=5, profile="test"))
m.to(dev).train()
if os.environ.get("MT", "1") == "1":
    m.merge_trunk, m.merge_from = True, int(os.environ.get("MF", 3))
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=9)]
mode = sys.argv[1]
if mode in ("train", "gtrain"):
    from dana_amd.trainer import Trainer
    from dana_amd.graphs import GraphedTrainer
    tr = Trainer(m, lr=1e-3)
    step = GraphedTrainer(tr, *inputs).step if mode == "gtrain" else tr.step
    np.random.seed(11)
    for _ in range(5):
        out = step(*inputs)
    torch.cuda.synchronize()
    chk = sum(float(p.detach().double().abs().sum()) for n_, p in m.named_parameters() if p.requires_grad)
    print("ok", mode, "%.10f" % chk, " ".join("%.6f" % float(x.detach()) for x in out[3:7]))
    sys.exit(0)
if mode == "eager":
    for _ in range(5):
        np.random.seed(3)
        with torch.no_grad():
            out = m(*inputs)
    torch.cuda.synchronize()
else:
    run = GraphedDAnA(m, *inputs)
    for _ in range(5):
        np.random.seed(3)
        out = run(*inputs)
    torch.cuda.synchronize()
print("ok", mode, float(out[3]))
