"""N processes (default 8) training on ONE GPU at once, as eight ranks of one host would issue their iterations: does the
per-process HOST time of an iteration stretch when N interpreters share the host? Every process runs
the bs-4 training iteration K times behind a gloo barrier -- eagerly (Trainer.step), then as launch programs
(program.ProgramTrainer) -- and reports its host enqueue time per iteration (time in the step call minus the time blocked
in the one D2H read) and its wall time per iteration; the GPU is shared, so the wall time is ~N x the single-process one.
usage: python tools/stress_host_8proc.py [N] [K]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, k, port):
    sys.path.insert(0, ROOT)
    import dana_amd
    from dana_amd import ops, program, synthetic as S
    from dana_amd.trainer import Trainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
    m.to(dev).train()
    inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996 + rank)]
    tr = Trainer(m, 1e-5, process_group=None)
    tr.weights.collective = tr.biases.collective = False  # (the exchange is not what is measured: N replicas of one rank)
    np.random.seed(rank)
    pt = program.ProgramTrainer(tr, *inputs)
    rows = {}
    for name, step, args_ in (("eager", tr.step, inputs), ("program", pt.step, pt.inputs)):
        for _ in range(3):
            step(*args_)
        torch.cuda.synchronize()
        dist.barrier()
        host, t0 = 0.0, time.perf_counter()
        for _ in range(k):
            ops.HOST_WAIT[0] = 0.0
            h0 = time.perf_counter()
            step(*args_)
            host += time.perf_counter() - h0 - ops.HOST_WAIT[0]
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        dist.barrier()
        rows[name] = (1e3 * host / k, 1e3 * wall / k)
    out = [None] * world
    dist.all_gather_object(out, rows)
    if rank == 0:
        print("| processes on one GPU | launch | host enqueue ms / iteration (min .. max over processes) | wall ms / iteration |")
        print("|---|---|---|---|")
        for name in ("eager", "program"):
            hs = [r[name][0] for r in out]
            ws = [r[name][1] for r in out]
            print("| %d | %s | %.2f .. %.2f (mean %.2f) | %.1f .. %.1f |" % (world, name, min(hs), max(hs), sum(hs) / len(hs),
                                                                          min(ws), max(ws)))
    dist.destroy_process_group()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    print("host: %d logical cores" % os.cpu_count())
    for world in sorted({1, n}):
        mp.spawn(worker, args=(world, k, 29700 + world), nprocs=world, join=True)
