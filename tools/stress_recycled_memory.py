"""Do results depend on what the caching allocator recycles? Four training iterations of every model (DAnA also replayed from
hipGraphs) with garbage / NaN written into freed blocks before every step, against a clean run: losses and a parameter sample
must agree to 1e-3 (RoIAlign-backward atomics are unordered). usage: python tools/stress_recycled_memory.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import synthetic as S
from dana_amd.trainer import Trainer
dev = torch.device("cuda:0")
def garbage(seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    ts = [torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev, generator=g) for n in (1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 3 << 16, 5 << 14, 1 << 12)]
    del ts
def nanfill(seed):
    ts = [torch.full((n,), float("nan"), device=dev) for n in (1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 3 << 16, 5 << 14, 1 << 12)]
    del ts
def run(name, fill, graphed=False):
    torch.cuda.empty_cache()
    way, shot = 2, 2
    m = dana_amd.get_model(name, pretrained=False, use_BA_block=True, way=way, shot=shot, classes=["fg", "bg"])
    sd = S.fill_state_dict(m.state_dict(), seed=21, profile="test")
    if name == "fsod": sd = S.tame_fsod_weights(sd)
    if name == "fgn": sd = S.tame_fgn_weights(sd)
    m.load_state_dict(sd); m.to(dev).train(); m.nms_inclusive = True
    e = [t.to(dev) for t in S.episode_inputs(2, way, shot, 192, 256, seed=23)]
    inputs = e[:4] if name == "frcnn" else (e + [e[2].clone()] if name == "meta" else e)
    tr = Trainer(m, 1e-5 if name == "fsod" else 0.01)
    losses = []
    stepper = tr.step
    if graphed:
        from dana_amd.graphs import GraphedTrainer
        np.random.seed(1); tr.step(*inputs); np.random.seed(2); tr.step(*inputs)
        gt = GraphedTrainer(tr, *inputs, warmup=0)
        stepper = gt.step
    for it in range(4):
        if fill: fill(it)
        np.random.seed(40 + it)
        out = stepper(*inputs)
        losses.append([float(x) for x in out[3:7]])
    torch.cuda.synchronize()
    chk = torch.cat([p.detach().reshape(-1)[::97].double() for p in m.parameters() if p.requires_grad]).cpu().numpy()
    return np.array(losses), chk
for name in ("DAnA", "frcnn", "meta", "fgn", "fsod"):
    for graphed in ((False, True) if name == "DAnA" else (False,)):
        l0, c0 = run(name, None, graphed)
        for fname, f in (("garbage", garbage), ("nan", nanfill)):
            l1, c1 = run(name, f, graphed)
            dl = np.abs(l0 - l1).max() / max(1.0, np.abs(l0).max()); dc = np.abs(c0 - c1).max() / max(1e-9, np.abs(c0).max())
            ok = np.isfinite(l1).all() and dl < 1e-3 and dc < 1e-3
            print("%-5s graphed=%d fill=%-7s  losses rel diff %.2e  params rel diff %.2e  %s" % (name, graphed, fname, dl, dc, "ok" if ok else "MISMATCH"), flush=True)
