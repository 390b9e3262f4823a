"""Where does a short contraction launch spend its time? Per-block timestamps of ONE split-kernel launch
(dana_set_igemm_trace): prologue / K loop / epilogue durations in shader cycles, start and end spread across the grid.
usage: python tools/igemm_trace.py n h w cin cout k stride residual [mode]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: F401
from dana_amd import ops
from dana_amd._lib import lib
n, h, w, ci, co, k, st, res = [int(v) for v in sys.argv[1:9]]
mode = int(sys.argv[9]) if len(sys.argv) > 9 else 1
dev = torch.device("cuda:0")
ops.set_mfma_mode(1 if mode else 0)
if mode > 1:
    ops.force_tile(mode)
x = torch.randn(n * h * w, ci, device=dev)
wt = torch.randn(co, k * k * ci, device=dev) * 0.05
sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
oh, ow = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
r = torch.randn(n * oh * ow, co, device=dev) if res else None
run = lambda: ops.conv2d_nhwc(x, n, h, w, ci, wt, co, k, k, st, k // 2, scale=sc, shift=sh, residual=r, relu=True)  # noqa: E731
for _ in range(5):
    run()
torch.cuda.synchronize()
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib().call("dana_set_igemm_trace", buf.data_ptr())
e0.record()
run()
e1.record()
torch.cuda.synchronize()
lib().call("dana_set_igemm_trace", None)
t = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8)
t = t[t[:, 3] > 0]
nb = len(t)
start, loop, loop_end, end, hw, wall = [t[:, i].astype(np.float64) for i in range(6)]
hwid = t[:, 4]
xcc = (hwid >> np.uint64(32)).astype(np.int64) & 0xf
cu = ((hwid >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
se = ((hwid >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
M = n * oh * ow
print("shape M=%d N=%d K=%d res=%d mode=%d: %d blocks, launch %.1f us (events), %.2f GF -> %.1f TF/s" % (
    M, co, k * k * ci, res, mode, nb, e0.elapsed_time(e1) * 1e3, 2.0 * M * co * k * k * ci / 1e9,
    2.0 * M * co * k * k * ci / (e0.elapsed_time(e1) * 1e-3) / 1e12))
pro, lp, epi, tot = loop - start, loop_end - loop, end - loop_end, end - start
q = lambda v: "min %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (v.min(), np.median(v), np.percentile(v, 90), v.max())  # noqa: E731
print("cycles per block  prologue: %s" % q(pro))
print("                  K loop  : %s" % q(lp))
print("                  epilogue: %s" % q(epi))
print("                  total   : %s" % q(tot))
wall_us = (wall - wall.min()) / 100.0
clk = np.median(tot) and (tot / 1.0)
# shader clock estimate: total cycles of the longest-lived block vs its wall time cannot be separated without a start wall
# stamp; report end-time spread and the blocks-per-CU picture instead
print("end wall-clock spread: last block ends %.1f us after the first; p10 %.1f p50 %.1f p90 %.1f us" % (
    wall_us.max(), np.percentile(wall_us, 10), np.median(wall_us), np.percentile(wall_us, 90)))
# start times in the wall domain, assuming ~2.1 GHz for the conversion of the block's own duration
wstart = t[:, 6].astype(np.float64)
ghz_meas = np.median(tot / np.maximum((wall - wstart) * 10.0, 1.0))  # cycles per ns
print("effective shader clock during the launch: %.2f GHz (median over blocks: cycles / 100 MHz wall ticks)" % ghz_meas)
for ghz in (ghz_meas,):
    start_us = wall_us - tot / (ghz * 1e3)
    s0 = start_us.min()
    print("block starts (wall, %.1f GHz assumed): p50 %.1f us p90 %.1f us max %.1f us after the first; kernel span %.1f us" % (
        ghz, np.median(start_us - s0), np.percentile(start_us - s0, 90), (start_us - s0).max(), wall_us.max() - s0))
    late = start_us - s0 > 0.3 * (wall_us.max() - s0)
    print("blocks starting in a later round: %d of %d" % (int(late.sum()), nb))
    first = start_us - s0 < 2.0  # dispatched with the launch: cold instruction cache, cold TLB, every block in the same phase
    for nm, sel in (("first round", first), ("later rounds", ~first)):
        if sel.sum():
            print("  %-12s (%4d blocks)  prologue p50 %6.0f  K loop p50 %6.0f  epilogue p50 %6.0f  total p50 %6.0f cycles" % (
                nm, int(sel.sum()), np.median(pro[sel]), np.median(lp[sel]), np.median(epi[sel]), np.median(tot[sel])))
ids = xcc * 1000 + se * 16 + cu
u, c = np.unique(ids, return_counts=True)
print("distinct (xcc, se, cu) ids seen: %d; blocks per id min %d max %d" % (len(u), c.min(), c.max()))
