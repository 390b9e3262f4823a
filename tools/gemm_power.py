"""Is the split GEMM power / clock limited? Same launches on random, constant and zero operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops
dev = torch.device('cuda:0')
m, n, k = 16384, 4096, 4096
for name, mk in [("randn", lambda *s: torch.randn(*s, device=dev)), ("ones", lambda *s: torch.ones(*s, device=dev)),
                 ("zeros", lambda *s: torch.zeros(*s, device=dev)), ("randn", lambda *s: torch.randn(*s, device=dev)),
                 ("small-int", lambda *s: torch.randint(-3, 4, s, device=dev).float())]:
    for mode in (1, 0):
        ops.set_mfma_mode(mode)
        a, b = mk(m, k), mk(n, k)
        out = torch.empty(m, n, device=dev)
        for _ in range(3): ops.gemm_nt(a, b, m, n, k, out=out, ldc=n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm_nt(a, b, m, n, k, out=out, ldc=n)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print("%-9s mode %d  %8.1f us  %6.1f TF/s" % (name, mode, us, 2.0 * m * n * k / us / 1e6), flush=True)
