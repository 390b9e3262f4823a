"""debug: which ingredient of the training-iteration capture breaks hipStreamEndCapture (one mode per process)"""
import os, sys, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.graphs import GraphedDAnA, GraphedTrainer
from dana_amd.trainer import Trainer
mode = int(sys.argv[1])
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=2, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=5, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(1, 2, 2, 160, 224, seed=6)]
if mode >= 1:
    tr = Trainer(m, 0.01, bucket_bytes=8 << 20)
    np.random.seed(1)
    tr.step(*inputs)
if mode in (0, 1):
    run = GraphedDAnA(m, *inputs)
elif mode == 2:
    m.save_for_backward = True
    run = GraphedDAnA(m, *inputs)
elif mode == 3:
    with torch.no_grad():
        m(*inputs)
    m._epoch += 1
    m._plan = None
    run = GraphedDAnA(m, *inputs, warmup=0)
elif mode == 4:
    m.device_rng = True
    gt = GraphedTrainer(tr, *inputs, warmup=1)
    gt.step(*inputs)
elif mode == 6:
    os.environ["DANA_DBG_NOZERO"] = "1"
    gt = GraphedTrainer(tr, *inputs, warmup=1)
    np.random.seed(2)
    gt.step(*inputs)
elif mode == 7:
    os.environ["DANA_DBG_NOZERO"] = "1"
    gt = GraphedTrainer(tr, *inputs, warmup=0)
    np.random.seed(2)
    gt.step(*inputs)
elif mode in (8, 9, 10, 11, 12, 13):
    from dana_amd import backward as BW, ops
    m.save_for_backward = True
    if mode == 10:
        m._single_stream = True
    if mode in (12, 13):
        orig = m._stream
        collapse = "layer4" if mode == 12 else "wgrad"
        m._stream = lambda name, d: torch.cuda.current_stream() if name == collapse else orig(name, d)
    st = torch.cuda.Stream()
    ones = torch.ones(4, device=dev)
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(2):
            np.random.seed(3)
            tr.zero_grad()
            m(*inputs)
            BW.model_backward(m, ones)
        np.random.seed(3)
        tr.zero_grad()
        m(*inputs)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad():
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            if mode == 9:
                gen = BW.model_backward_gen(m, ones)
                next(gen)
            elif mode == 11:
                for (fb, lr_mult, wd), buf in zip(tr.groups, tr.bufs):
                    ops.sgd_momentum_(fb.params, fb.grads, buf, 0.01 * lr_mult, 0.9, wd, grad_scale=1.0, first_step=False)
            else:
                BW.model_backward(m, ones)
    print("captured", flush=True)
    g.replay()
elif mode == 5:
    gt = GraphedTrainer(tr, *inputs, warmup=1)
    np.random.seed(2)
    gt.step(*inputs)
if mode <= 3:
    np.random.seed(2)
    run(*inputs)
torch.cuda.synchronize()
print("mode %d OK" % mode, flush=True)
