"""Which torch (aten) operators does one training iteration issue besides this package's own C-ABI launches, and from
where? (The launch-program recorder of dana_amd/program.py replays C-ABI calls + the few torch ops it is told about:
everything this tool lists must either be allocation-only or go through ops.t_*.) usage: python tools/aten_ops_in_step.py [fwd]"""
import collections
import os
import sys
import traceback

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd  # noqa: E402
from dana_amd import synthetic as S  # noqa: E402
from dana_amd.trainer import Trainer  # noqa: E402

ALLOC_ONLY = {"empty", "empty_strided", "empty_like", "view", "as_strided", "_unsafe_view", "reshape", "permute", "transpose",
              "t", "slice", "select", "narrow", "expand", "unsqueeze", "squeeze", "detach", "alias", "_reshape_alias", "split",
              "unbind", "contiguous", "record_stream", "is_pinned", "new_empty", "new_empty_strided", "flatten", "chunk",
              "split_with_sizes", "unflatten", "lift_fresh", "_local_scalar_dense", "item", "is_same_size", "sym_size",
              "is_nonzero", "numel", "size", "stride", "storage_offset", "dim"}


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()
        self.where = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in ALLOC_ONLY:
            on_gpu = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            dev = (kwargs or {}).get("device")
            if on_gpu or (dev is not None and "cuda" in str(dev)):
                self.seen[func.__name__] += 1
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "dual-awareness" in fr.filename or "dana_amd" in fr.filename:
                        self.where[func.__name__]["%s:%d" % (os.path.basename(fr.filename), fr.lineno)] += 1
                        break
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
fwd_only = len(sys.argv) > 1 and sys.argv[1] == "fwd"
tr = None if fwd_only else Trainer(m, 1e-3)
np.random.seed(0)
for _ in range(3):
    if fwd_only:
        with torch.no_grad():
            m(*inputs)
    else:
        tr.step(*inputs)
torch.cuda.synchronize()
spy = Spy()
with spy:
    if fwd_only:
        with torch.no_grad():
            m(*inputs)
    else:
        tr.step(*inputs)
torch.cuda.synchronize()
print("aten ops touching the GPU in one %s: %d calls of %d kinds" % ("forward" if fwd_only else "training iteration",
                                                                     sum(spy.seen.values()), len(spy.seen)))
for name, n in spy.seen.most_common():
    print("%5d  %-28s %s" % (n, name, ", ".join("%s x%d" % kv for kv in spy.where[name].most_common(12))))
