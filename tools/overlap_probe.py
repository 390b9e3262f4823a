"""Why do two independent, under-filled contraction launches not overlap?

Two layer-3 launches of the two trunks (query rows M = 4*38*63 = 9 576, support rows M = 24*400 = 9 600; resnet.py:84-100 as
dana.py:98-100 calls it twice) are run (a) each alone, (b) back to back on one stream, (c) on two plain streams with no
dependency, (d) on two CU-masked streams (every XCD split lo / hi: 16+16, 20+12, 24+8 CUs), (e) one plain + one masked.
GPU-side spans come from events; WHAT happens inside comes from the kernels' own stamps (dana_set_igemm_trace: per tile
shader-cycle counter at start / first K-step / loop end / end, the 100 MHz wall clock at start and end, HW_ID): which CU each
tile ran on, how many tiles of which launch were resident on a CU at the same time, the tile's duration and the shader
clock it saw (cycles / wall) alone and concurrent. rocprofv3 PMC counters cannot see this: counter collection serialises
dispatches.  usage: python tools/overlap_probe.py [iters] [pair ...]   pairs: reduce expand mixed"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402,F401
from dana_amd import ops  # noqa: E402
from dana_amd._lib import lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
which = sys.argv[2:] or ["reduce", "expand", "mixed", "quant"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
MQ, MS = 4 * 38 * 63, 24 * 400


def conv(m, cin, cout, res):
    x = torch.randn(m, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.05
    ws = ops.split_weight(w, cout, cin)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    r = torch.randn(m, cout, device=dev) if res else None
    out = torch.empty(m, cout, device=dev)
    fn = lambda: ops.conv2d_nhwc(x, 1, 1, m, cin, ws, cout, 1, 1, 1, 0, scale=sc, shift=sh, residual=r, relu=True,  # noqa: E731
                                 out=out, out_stride=cout)
    fn.tiles = ((m + 127) // 128) * ((cout + 127) // 128)
    fn.gf = 2.0 * m * cin * cout / 1e9
    fn.ksteps = cin // 16
    return fn


TR_WORDS = 8192 * 8
trace_a = torch.zeros(TR_WORDS, dtype=torch.int64, device=dev)
trace_b = torch.zeros(TR_WORDS, dtype=torch.int64, device=dev)


def set_trace(buf):
    lib().call("dana_set_igemm_trace", buf.data_ptr() if buf is not None else None)


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def alone(fn, stream=None):
    st = stream or torch.cuda.current_stream()
    with torch.cuda.stream(st):
        for _ in range(10):
            fn()
        sp = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            sp.append(e0.elapsed_time(e1) * 1e3)
    return med(sp)


SLEEP_US = 150.0
_cyc_per_us = None


def gate(main):
    """hold `main` busy for ~SLEEP_US so that the host can enqueue BOTH launches behind the fork event before either may
    start (a Python-issued launch costs 10-18 us of host time: without the gate the second launch is simply late)"""
    global _cyc_per_us
    if _cyc_per_us is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000000)
        torch.cuda.synchronize()
        e0.record()
        torch.cuda._sleep(4000000)
        e1.record()
        torch.cuda.synchronize()
        _cyc_per_us = 4000000 / (e0.elapsed_time(e1) * 1e3)
    torch.cuda._sleep(int(SLEEP_US * _cyc_per_us))


def overlapping_streams(n_try=8):
    """two torch streams that sit on DIFFERENT hardware queues (HIP multiplexes streams over four; two streams on one
    queue run one behind the other whatever the dependencies say): checked with two 100 us spin kernels"""
    main = torch.cuda.current_stream()
    pool = [torch.cuda.Stream(device=dev) for _ in range(n_try)]
    for s_ in pool:
        with torch.cuda.stream(s_):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    gate(main)
    torch.cuda.synchronize()
    for i in range(n_try):
        for j in range(i + 1, n_try):
            sp = []
            for _ in range(3):
                gate(main)
                fork = torch.cuda.Event()
                fork.record(main)
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                pool[i].wait_event(fork)
                pool[j].wait_event(fork)
                with torch.cuda.stream(pool[i]):
                    e0.record()
                    torch.cuda._sleep(int(100 * _cyc_per_us))
                    e1.record()
                with torch.cuda.stream(pool[j]):
                    torch.cuda._sleep(int(100 * _cyc_per_us))
                    e2.record()
                torch.cuda.synchronize()
                sp.append(max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3)
            if med(sp) < 140.0:
                print("streams %d and %d overlap (two 100 us spins take %.0f us together)" % (i, j, med(sp)), flush=True)
                return pool[i], pool[j]
            print("streams %d and %d share a hardware queue (two 100 us spins take %.0f us)" % (i, j, med(sp)), flush=True)
    raise RuntimeError("no pair of overlapping streams found")


def pair(a, b, sa, sb, traced=False):
    """one overlapped (or, sa is sb: back-to-back) pair per iteration; span = first event -> the later end. Returns the
    median span and, with traced=True, the stamps of the iteration whose span was the median."""
    main = torch.cuda.current_stream()
    spans, stamps = [], []
    for it in range(iters + 8):
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        if traced:
            trace_a.zero_()
            trace_b.zero_()
            torch.cuda.synchronize()
        gate(main)
        e0.record(main)  # (on the gate's stream: an event record IN FRONT of the kernel on a CU-masked stream delays that
        #                   stream's first kernel by ~75 us in this ROCm build -- measured with the first version of this tool)
        fork = torch.cuda.Event()
        fork.record(main)
        sa.wait_event(fork)
        sb.wait_event(fork)
        with torch.cuda.stream(sa):
            if traced:
                set_trace(trace_a)
            a()
            ea.record()
        with torch.cuda.stream(sb):
            if traced:
                set_trace(trace_b)
            b()
            eb.record()
        set_trace(None)
        main.wait_stream(sa)
        main.wait_stream(sb)
        torch.cuda.synchronize()
        if it >= 8:
            spans.append(max(e0.elapsed_time(ea), e0.elapsed_time(eb)) * 1e3)
            if traced:
                stamps.append((trace_a.cpu().numpy().copy(), trace_b.cpu().numpy().copy()))
    order = np.argsort(spans)
    mid = order[len(order) // 2]
    return spans[mid], (stamps[mid] if traced else None)


def decode(buf):
    t = buf.astype(np.uint64).reshape(-1, 8)
    t = t[t[:, 3] > 0]
    hw = t[:, 4]
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
    sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
    return dict(cyc=(t[:, 3] - t[:, 0]).astype(np.float64), pro=(t[:, 1] - t[:, 0]).astype(np.float64),
                loop=(t[:, 2] - t[:, 1]).astype(np.float64), epi=(t[:, 3] - t[:, 2]).astype(np.float64),
                w0=t[:, 6].astype(np.float64) / 100.0, w1=t[:, 5].astype(np.float64) / 100.0,  # us
                cuid=xcc * 4096 + se * 64 + sh * 32 + cu, xcc=xcc)


def describe(tag, sa_, sb_, ksteps):
    A, B = decode(sa_), decode(sb_)
    t0 = min(A["w0"].min(), B["w0"].min())
    out = []
    for nm, T, ks in (("A", A, ksteps[0]), ("B", B, ksteps[1])):
        dur = T["w1"] - T["w0"]
        ghz = np.median(T["cyc"] / np.maximum(dur, 1e-3) / 1e3)
        out.append("%s: %d tiles on %d CUs; first tile starts %+.1f us, last ends %.1f us; tile wall p50 %.1f / p90 %.1f / max %.1f us; "
                   "loop %.3f us per K-step (p50); shader clock %.2f GHz" % (
                       nm, len(dur), len(np.unique(T["cuid"])), T["w0"].min() - t0, T["w1"].max() - t0, np.median(dur),
                       np.percentile(dur, 90), dur.max(), np.median(T["loop"]) / ks / (ghz * 1e3), ghz))
    # residency: per CU, how many tiles are live at once (both launches together), and for how long
    cu_all = np.concatenate([A["cuid"], B["cuid"]])
    w0 = np.concatenate([A["w0"], B["w0"]])
    w1 = np.concatenate([A["w1"], B["w1"]])
    kind = np.concatenate([np.zeros(len(A["w0"])), np.ones(len(B["w0"]))])
    span = w1.max() - w0.min()
    n_cu = len(np.unique(cu_all))
    both, peak, busy = 0, [], 0.0
    shared_time = 0.0
    for c in np.unique(cu_all):
        sel = cu_all == c
        ev = sorted([(s, 1) for s in w0[sel]] + [(e, -1) for e in w1[sel]])
        live, pk, last, t_busy, t_multi = 0, 0, None, 0.0, 0.0
        for tt, d in ev:
            if last is not None and live > 0:
                t_busy += tt - last
                if live > 1:
                    t_multi += tt - last
            live += d
            pk = max(pk, live)
            last = tt
        peak.append(pk)
        busy += t_busy
        shared_time += t_multi
        if len(np.unique(kind[sel])) == 2:
            both += 1
    peak = np.array(peak)
    # tiles that shared their CU with another tile for > 20 % of their life vs the rest
    out.append("CUs used %d (of 256); CUs that hosted tiles of BOTH launches %d; peak co-resident tiles per CU: 1 on %d CUs, 2 on %d, >=3 on %d; "
               "CU-time busy %.0f %% of (CUs used x span %.1f us), of which %.0f %% with >= 2 tiles resident" % (
                   n_cu, both, int((peak == 1).sum()), int((peak == 2).sum()), int((peak >= 3).sum()), 100 * busy / (n_cu * span), span,
                   100 * shared_time / max(busy, 1e-9)))
    per_xcc = [int(((np.concatenate([A["xcc"], B["xcc"]])) == x).sum()) for x in range(8)]
    out.append("tiles per XCD: %s" % per_xcc)
    print("  [%s]" % tag)
    for line in out:
        print("    " + line)


def run_pair(name, a, b):
    print("## %s: A = %d tiles x %d K-steps (%.2f GF), B = %d tiles x %d K-steps (%.2f GF)" % (
        name, a.tiles, a.ksteps, a.gf, b.tiles, b.ksteps, b.gf), flush=True)
    ta, tb = alone(a), alone(b)
    main = torch.cuda.current_stream()
    s1, s2 = S12
    serial, st_serial = pair(a, b, s1, s1, traced=True)
    conc, st_conc = pair(a, b, s1, s2, traced=True)
    serial_u, _ = pair(a, b, s1, s1)
    conc_u, _ = pair(a, b, s1, s2)
    print("| mode | span us | vs back-to-back |")
    print("|---|---|---|")
    print("| A alone / B alone | %.1f / %.1f | |" % (ta, tb))
    print("| back to back, one stream | %.1f (traced %.1f) | 1.00 |" % (serial_u, serial))
    print("| two plain streams | %.1f (traced %.1f) | %.2f |" % (conc_u, conc, conc_u / serial_u))
    rows = {}
    for lo in (16, 20, 12):
        ma = ops.cumask_stream(range(0, lo))
        mb = ops.cumask_stream(range(lo, 32))
        ta_m, tb_m = alone(a, ma), alone(b, mb)
        sp, st = pair(a, b, ma, mb, traced=True)
        sp_u, _ = pair(a, b, ma, mb)
        rows[lo] = st
        print("| CU masks %d + %d of every XCD's 32 (A alone on its mask %.1f, B alone %.1f) | %.1f (traced %.1f) | %.2f |" % (
            lo, 32 - lo, ta_m, tb_m, sp_u, sp, sp_u / serial_u), flush=True)
        torch.cuda.synchronize()
        for m_ in (ma, mb):  # a masked stream owns a hardware queue of its own: give it back
            lib().call("dana_debug_stream_destroy", m_.cuda_stream)
    describe("back to back (traced)", st_serial[0], st_serial[1], (a.ksteps, b.ksteps))
    describe("two plain streams (traced)", st_conc[0], st_conc[1], (a.ksteps, b.ksteps))
    for lo in (16, 20, 12):
        describe("CU masks %d + %d (traced)" % (lo, 32 - lo), rows[lo][0], rows[lo][1], (a.ksteps, b.ksteps))
    del main


def quantisation():
    """one launch alone: rows chosen so that the 128x128 tiles are 512 (= the chip's 2 x 256 slots), 600 (the layer-3
    shape), 768 and 1 024"""
    print("## tail quantisation: expand conv 256->1024 + residual, ONE launch, M varied")
    print("| M | tiles | us | us per 512 tiles | TF/s |")
    print("|---|---|---|---|---|")
    for m in (4096, 8192, 9576, 12288, 16384, 19176, 24576):
        f = conv(m, 256, 1024, True)
        t = alone(f)
        print("| %d | %d | %.1f | %.1f | %.1f |" % (m, f.tiles, t, t * 512 / f.tiles, f.gf / t * 1e3), flush=True)
    print("## the same for the reduce conv 1024->256 (tile-starved: one tile per CU up to 256 tiles)")
    print("| M | tiles | us | TF/s |")
    print("|---|---|---|---|")
    for m in (9576, 16384, 19176, 32768):
        f = conv(m, 1024, 256, False)
        t = alone(f)
        print("| %d | %d | %.1f | %.1f |" % (m, f.tiles, t, f.gf / t * 1e3), flush=True)


S12 = overlapping_streams()
if "quant" in which:
    quantisation()


if "reduce" in which:   # the next block's reduce conv of both trunks (K = 1 024, 150 tiles each: tile-starved, long K)
    run_pair("layer3 reduce conv 1x1 1024->256, query rows + support rows", conv(MQ, 1024, 256, False), conv(MS, 1024, 256, False))
if "expand" in which:   # expand conv + residual + ReLU (K = 256, 600 tiles each: 1.2 rounds at two workgroups per CU)
    run_pair("layer3 expand conv 1x1 256->1024 + residual, query rows + support rows", conv(MQ, 256, 1024, True), conv(MS, 256, 1024, True))
if "mixed" in which:    # what the two trunk streams typically hold at the same time: one of each
    run_pair("layer3 expand (query rows) + reduce (support rows)", conv(MQ, 256, 1024, True), conv(MS, 1024, 256, False))
