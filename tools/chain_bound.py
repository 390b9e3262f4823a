"""Upper bound of what tile-level chaining of DEPENDENT contractions could buy (verdict r3 item 1b): time a bottleneck's
consecutive launches (a) as the product issues them -- one stream, each waiting for its predecessor -- and (b) with the
dependency IGNORED, on two streams, i.e. the perfect overlap a flag-chained launch can at most reach (it would add the
producer's release, the consumer's acquire and the polling on top). Spans are per pair, GPU-side, medians. Layer2 / layer3 identity-block shapes of one batch.
usage: python tools/chain_bound.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402,F401
from dana_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)


def conv(m, cin, cout, res):
    x = torch.randn(m, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.05
    ws = ops.split_weight(w, cout, cin)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    r = torch.randn(m, cout, device=dev) if res else None
    out = torch.empty(m, cout, device=dev)
    return lambda: ops.conv2d_nhwc(x, 1, 1, m, cin, ws, cout, 1, 1, 1, 0, scale=sc, shift=sh, residual=r, relu=True, out=out,
                                   out_stride=cout)


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


side = torch.cuda.Stream(device=dev)


def span_two_streams(a, b):
    """median GPU-side span of ONE overlapped pair: from an event in front of A (caller's stream) to the later of the events
    behind A and behind B (B on the side stream, forked in front of the first event). Per-iteration spans, not a loop
    average: the cross-stream fork / join between iterations is host and scheduler time a chained launch would not pay."""
    main = torch.cuda.current_stream()
    spans = []
    for it in range(iters + 20):
        fork = torch.cuda.Event()
        fork.record()
        side.wait_event(fork)
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        a()
        ea.record()
        with torch.cuda.stream(side):
            b()
            eb.record()
        main.wait_stream(side)
        torch.cuda.synchronize()
        if it >= 20:
            spans.append(max(e0.elapsed_time(ea), e0.elapsed_time(eb)) * 1e3)
    spans.sort()
    return spans[len(spans) // 2]


def span_one_stream(a, b):
    spans = []
    for it in range(iters + 20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        a()
        b()
        e1.record()
        torch.cuda.synchronize()
        if it >= 20:
            spans.append(e0.elapsed_time(e1) * 1e3)
    spans.sort()
    return spans[len(spans) // 2]


print("| pair (one batch) | A alone us | B alone us | A then B, one stream us | A and B, two streams (no dependency) us | bound of chaining |")
print("|---|---|---|---|---|---|")
for name, m, c_mid, c_out in (("layer3 conv3(+res) -> next conv1", 9576, 256, 1024), ("layer2 conv3(+res) -> next conv1", 37500, 128, 512),
                              ("layer3, query+support rows", 19176, 256, 1024)):
    a = conv(m, c_mid, c_out, True)     # expand conv + residual + ReLU (resnet.py:95-100)
    b = conv(m, c_out, c_mid, False)    # the next block's reduce conv (resnet.py:84-86)
    ta, tb = timed(a), timed(b)
    ts, tp = span_one_stream(a, b), span_two_streams(a, b)
    print("| %s | %.1f | %.1f | %.1f | %.1f | %.1f us = %.0f %% of the pair |" % (name, ta, tb, ts, tp, ts - tp, 100 * (ts - tp) / ts))
