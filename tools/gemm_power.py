"""Is the split GEMM power / clock limited? The same 16384 x 4096 x 4096 launch on zero / constant / small-integer /
N(0,1) operands, on both contraction kernels, with the shader clock and the socket power sampled WHILE it runs
(amd-smi / rocm-smi / hwmon, whichever the box offers) -> markdown table (profiles/r2_gemm_power.md).
usage: python tools/gemm_power.py [out.md]"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402,F401
from dana_amd import ops  # noqa: E402


def _hwmon():
    out = []
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        f, p = os.path.join(d, "freq1_input"), None
        for cand in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(d, cand)):
                p = os.path.join(d, cand)
        if os.path.exists(f) or p:
            out.append((f if os.path.exists(f) else None, p))
    return out


def _read(path):
    try:
        with open(path) as fh:
            return float(fh.read().strip())
    except (OSError, ValueError):
        return None


def _smi_sample():
    """-> (sclk MHz, power W) via amd-smi metric JSON, else rocm-smi JSON, else (None, None)"""
    try:
        j = json.loads(subprocess.run(["amd-smi", "metric", "-g", "0", "--clock", "--power", "--json"],
                                      capture_output=True, text=True, timeout=5).stdout)
        g = j[0] if isinstance(j, list) else (j.get("gpu_data") or [j])[0]
        clk = g.get("clock", {})
        vals = [v.get("clk", {}).get("value") for k, v in clk.items() if k.startswith("gfx") and isinstance(v, dict)]
        vals = [float(v) for v in vals if isinstance(v, (int, float))]
        pw = g.get("power", {}).get("socket_power", {})
        pw = pw.get("value") if isinstance(pw, dict) else pw
        return (max(vals) if vals else None), (float(pw) if isinstance(pw, (int, float)) else None)
    except Exception:
        pass
    try:
        j = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                                      timeout=5).stdout)
        c = j.get("card0", {})
        sclk = next((v for k, v in c.items() if "sclk" in k.lower() and "level" not in k.lower()), None)
        pw = next((v for k, v in c.items() if "power" in k.lower() and "(w)" in k.lower()), None)
        f = lambda x: float("".join(ch for ch in str(x) if ch.isdigit() or ch == ".")) if x is not None else None  # noqa: E731
        return f(sclk), f(pw)
    except Exception:
        return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.hw = _hwmon()
        self.stop = False
        self.clk, self.pw = [], []

    def run(self):
        while not self.stop:
            got = False
            for f, p in self.hw:
                c = _read(f) if f else None
                w = _read(p) if p else None
                if c:
                    self.clk.append(c / 1e6)
                    got = True
                if w:
                    self.pw.append(w / 1e6)
                    got = True
            if not got:
                c, w = _smi_sample()
                if c:
                    self.clk.append(c)
                if w:
                    self.pw.append(w)
            time.sleep(0.02)


def main():
    dev = torch.device("cuda:0")
    m, n, k = 16384, 4096, 4096
    rows = []
    makers = [("zeros", lambda *s: torch.zeros(*s, device=dev)), ("ones", lambda *s: torch.ones(*s, device=dev)),
              ("small-int", lambda *s: torch.randint(-3, 4, s, device=dev).float()),
              ("N(0,1)", lambda *s: torch.randn(*s, device=dev))]
    for name, mk in makers:
        for mode in (1, 0):
            ops.set_mfma_mode(mode)
            a, b = mk(m, k), mk(n, k)
            out = torch.empty(m, n, device=dev)
            for _ in range(3):
                ops.gemm_nt(a, b, m, n, k, out=out, ldc=n)
            torch.cuda.synchronize()
            reps = 150 if mode else 60  # ~1 s of back-to-back launches: long enough for the firmware to settle the clock
            smp = Sampler()
            smp.start()
            time.sleep(0.1)
            smp.clk, smp.pw = [], []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm_nt(a, b, m, n, k, out=out, ldc=n)
            e1.record()
            torch.cuda.synchronize()
            smp.stop = True
            smp.join()
            us = e0.elapsed_time(e1) * 1e3 / reps
            ghz = float("nan")
            if mode:  # effective shader clock of one more launch from the kernel's own per-block stamps
                from dana_amd._lib import lib
                tb = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
                lib().call("dana_set_igemm_trace", tb.data_ptr())
                ops.gemm_nt(a, b, m, n, k, out=out, ldc=n)
                torch.cuda.synchronize()
                lib().call("dana_set_igemm_trace", None)
                import numpy as np
                tt = tb.cpu().numpy().astype(np.uint64).reshape(-1, 8)
                tt = tt[tt[:, 3] > 0].astype(np.float64)
                ghz = float(np.median((tt[:, 3] - tt[:, 0]) / np.maximum((tt[:, 5] - tt[:, 6]) * 10.0, 1.0)))
            tail = lambda v: sorted(v[len(v) // 3:])[len(v[len(v) // 3:]) // 2] if len(v) >= 3 else (v[-1] if v else float("nan"))  # noqa: E731
            rows.append((name, "bf16x6 split" if mode else "f32 MFMA", us, 2.0 * m * n * k / us / 1e6, ghz, tail(smp.clk),
                         tail(smp.pw), len(smp.clk), len(smp.pw)))
            print("%-9s %-12s %8.1f us %6.1f TF/s  in-kernel clock %.2f GHz  smi sclk %.0f MHz  power %.0f W  (%d / %d samples)" % rows[-1], flush=True)
    ops.set_mfma_mode(1)
    L = ["# Operand data vs clock / power / throughput of one large contraction (round 2)", "",
         "`python tools/gemm_power.py` on the GPU box: `dana_gemm_nt` M=16384 N=4096 K=4096, ~1 s of back-to-back launches per row,",
         "SMI shader clock and socket power sampled every 20 ms while they run (median of the last two thirds of the samples;",
         "the SMI clock is an instantaneous register read and noisy). `in-kernel clock` = shader cycles (s_memtime) each block of one",
         "more launch counted between its first and last instruction / the 100 MHz wall clock over the same span, median over the",
         "blocks (dana_set_igemm_trace): the clock the CUs really ran at while the kernel executed (split kernel only).", "",
         "| operands | kernel | us / launch | algorithmic TFLOP/s | in-kernel clock (GHz) | SMI sclk (MHz) | power (W) | samples |",
         "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        L.append("| %s | %s | %.1f | %.1f | %.2f | %.0f | %.0f | %d / %d |" % r)
    text = "\n".join(L) + "\n"
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write(text)
    print(text)


if __name__ == "__main__":
    main()
