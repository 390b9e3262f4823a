"""Launch programs: the eager step's own launch sequence, recorded once and replayed without the Python in between.

Why not a hipGraph (graphs.py)? An eager training iteration is ~350 C-ABI launches + ~75 torch ops + ~250 event records /
waits issued from ~2 300 lines of Python: 11-13 ms of host time per 16.4 ms iteration (BENCH_r05 `host_enqueue_ms_per_step`).
Its hipGraph capture replays in 3.5 ms of host time but 2 ms SLOWER on the GPU (18.3 vs 16.3 ms): ROCm 7.2's
hipStreamEndCapture crashes on a side stream forked from an already forked stream, so the captured iteration keeps its
weight gradients and the box branch inline, and hipGraphLaunch places the nodes on hardware queues of its own choosing --
while WHICH of the model's five role streams share a hardware queue is worth 1-2 ms (profiles/r5_role_streams.md).

A `LaunchProgram` records at the two boundaries every GPU operation of this package crosses -- `_lib._Lib.call` (the C ABI:
function pointer + arguments, the stream is one of them) and torch (`Event.record` / `Event.wait`, the handful of aten ops
the backward uses: zeros, add_, copy_, clone) -- while the step runs EAGERLY, on the model's own role streams, in the order
the host issues it. Replay walks that list: the same launches with the same arguments on the same streams behind the same
event edges, i.e. the eager iteration's GPU schedule at a fraction of its host time (one ctypes call per launch, no shape
logic, no allocator, no tensor objects). Host-side steps that must stay live (the one host round trip of the training
forward: counts -> np.random draws -> upload; the bucket all-reduces of a multi-rank iteration) are either run between two
programs or recorded as CALLBACK entries and re-executed in place.

What makes the replay valid:
* addresses: every tensor the recorded step allocates comes from a private `torch.cuda.MemPool` that lives as long as the
  program; a tensor that crossed streams (`Tensor.record_stream`) is kept alive, so its block is never recycled inside the
  recording -- the allocator defers such a reuse by QUERYING an event, which a replay cannot reproduce. Same-stream recycling
  stays (stream order makes it safe, and it keeps the working set small);
* data-dependent host logic: none inside a program (the draws are the cut between two programs; shapes are static);
* weight-derived tensors (Winograd filters, data-gradient weights, merged heads) are re-derived by launches INSIDE the
  program from the live flat parameter buffer, as in graphs.GraphedTrainer.
A torch op the recorder does not know how to replay raises at record time (loudly, never a silent divergence).
"""
import contextlib
import ctypes
import gc
import os
import struct

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib, ops
from . import backward as BW

_EV_RECORD = torch._C._CudaEventBase.record
_EV_WAIT = torch._C._CudaEventBase.wait
_CALL, _REC, _WAIT, _ATEN, _HOST, _CRUN = 0, 1, 2, 3, 4, 5

# aten ops that only make views / allocate / inspect: nothing to replay (the dispatch mode sees them all)
_PASSIVE = {"empty", "empty_strided", "empty_like", "new_empty", "new_empty_strided", "view", "as_strided", "_unsafe_view",
            "reshape", "_reshape_alias", "permute", "transpose", "t", "slice", "select", "narrow", "expand", "unsqueeze",
            "squeeze", "detach", "alias", "split", "split_with_sizes", "unbind", "chunk", "flatten", "unflatten",
            "contiguous", "record_stream", "is_pinned", "lift_fresh", "is_same_size", "view_as", "expand_as"}


# ops whose OUTPUT SHAPE (or host-side result) depends on the data: a replay cannot reproduce them
_DATA_DEPENDENT = {"nonzero", "masked_select", "unique", "unique_consecutive", "_unique", "_unique2", "unique_dim", "index",
                   "where", "bincount", "repeat_interleave", "_local_scalar_dense", "item", "equal", "is_nonzero"}


class _AtenRecorder(TorchDispatchMode):
    def __init__(self, prog):
        super().__init__()
        self.prog = prog

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        prog = self.prog
        if prog._suspended:
            return out
        name = func.__name__.split(".")[0]
        if name in _PASSIVE:
            return out
        tens = [a for a in list(args) + list(kwargs.values()) if torch.is_tensor(a)]
        on_gpu = any(t.is_cuda for t in tens) or (torch.is_tensor(out) and out.is_cuda)
        if not on_gpu:
            return out
        if torch.is_tensor(out) and any(out is t for t in tens) and not name.endswith("_"):
            return out  # returned its input unchanged
        if torch.is_tensor(out) and not out.is_cuda:
            # a device-to-host read: whatever the host does with it would be baked into the program as of the recording
            raise RuntimeError("LaunchProgram: torch op %s reads device data to the host inside a recorded step: it has no "
                               "replay rule (host logic belongs between two programs, or in a host_callback)" % func.__name__)
        st = ops._get_cur(ops._raw_device())
        if prog._as_call(name, args, kwargs, out, tens):
            prog.keep.append(out)
            return out  # (re-issued by the C loop: a fill, a device-to-device copy or an axpy of this library)
        if name.endswith("_") or func.__name__.endswith(".out"):
            prog._add((_ATEN, func, (args, kwargs, st)))            # in place / into a given output: the same call again
        elif name in ("zeros", "zeros_like"):
            prog._add((_ATEN, torch.ops.aten.zero_.default, ((out,), {}, st)))
        elif name in ("clone", "_to_copy") and len(tens) == 1 and tens[0].is_cuda and out.is_cuda:
            prog._add((_ATEN, torch.ops.aten.copy_.default, ((out, tens[0]), {}, st)))
        elif name in ("ones", "full", "ones_like", "full_like"):
            val = 1.0 if name.startswith("ones") else (args[1] if len(args) > 1 else kwargs["fill_value"])
            prog._add((_ATEN, torch.ops.aten.fill_.Scalar, ((out, val), {}, st)))
        elif (torch.is_tensor(out) and name not in _DATA_DEPENDENT and hasattr(getattr(torch.ops.aten, name, None), "out")
              and not kwargs.get("dtype")):
            # a functional op with an out= overload (cat of the merged RPN head, a stray add): into the recorded output
            prog._add((_ATEN, getattr(torch.ops.aten, name).out, (args, dict(kwargs, out=out), st)))
        else:
            raise RuntimeError("LaunchProgram: torch op %s inside a recorded step has no replay rule (route it through a "
                               "C-ABI op, an in-place / out= form, or add the rule in program.py)" % func.__name__)
        prog.keep.append(out)
        return out


class LaunchProgram:
    """One recorded segment of a step. `with prog.recording(): <eager code>`, then `prog.run()` any number of times."""

    def __init__(self, device, pool=None):
        self.device = torch.device(device)
        self.entries = []
        self.keep = []          # tensors / events / ctypes buffers the entries point at
        self.pool = pool if pool is not None else torch.cuda.MemPool()
        self.main = None        # (stream_id, device_index, device_type) the step was recorded on
        self.main_raw = None
        self._suspended = 0
        self._entry_ev = torch.cuda.Event()
        self._exit_ev = torch.cuda.Event()
        self.stats = {}

    # ---- recording -----------------------------------------------------------------------------------------------
    def _add(self, e):
        self.entries.append(e)

    def add_call(self, fn, name, args):
        """_lib._Lib.call hook: the launch has just been issued eagerly"""
        if self._suspended:
            return
        conv = []
        for ty, a in zip(fn.argtypes, args):
            if isinstance(a, ctypes._SimpleCData):
                conv.append(a)      # already a ctypes value (a ctypes.cast() result also keeps its host array alive)
            elif a is None:
                conv.append(ty())   # null pointer
            else:
                conv.append(ty(a))  # converted ONCE: the replay hands ctypes its own types
        self.entries.append((_CALL, fn, tuple(conv)))

    def _as_call(self, name, args, kwargs, out, tens):
        """the backward's three torch ops -- zeros / zero_, copy_ / clone between device tensors, grad.add_ -- as C-ABI
        entries of the program (same bytes / same fp32 additions), so that they sit INSIDE the C loop's runs instead of
        cutting them; anything else stays a torch entry"""
        if not torch.is_tensor(out) or not out.is_cuda or not out.is_contiguous() or kwargs.get("alpha", 1) != 1:
            return False
        L, s = _lib.lib(), ops._stream()
        nbytes = out.numel() * out.element_size()
        if name in ("zeros", "zeros_like", "zero_"):
            self.add_call(L.fn["dana_fill_zero"], "dana_fill_zero", (out.data_ptr(), nbytes, s))
            return True
        src = None
        if name == "copy_" and len(args) >= 2 and torch.is_tensor(args[1]):
            src = args[1]
        elif name in ("clone", "_to_copy") and len(tens) == 1:
            src = tens[0]
        if src is not None:
            if (src.is_cuda and src.is_contiguous() and src.dtype == out.dtype and src.numel() == out.numel()
                    and src.device == out.device):
                self.add_call(L.fn["dana_copy_d2d"], "dana_copy_d2d", (out.data_ptr(), src.data_ptr(), nbytes, s))
                return True
            return False
        if name == "add_" and len(args) == 2 and torch.is_tensor(args[1]):
            x = args[1]
            if (x.is_cuda and x.is_contiguous() and x.dtype == out.dtype == torch.float32 and x.numel() == out.numel()
                    and out.numel() % 4 == 0 and out.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and out.numel() > 0):
                self.add_call(L.fn["dana_axpy_rows"], "dana_axpy_rows", (out.data_ptr(), x.data_ptr(), 1, out.numel(), 0, 0, 1.0,
                                                                          1, s))
                return True
        return False

    def host_callback(self, fn):
        """run `fn()` now (unrecorded) and again at this position of every replay: a host-side step that must stay live
        inside the program (the bucket all-reduces of a multi-rank backward)"""
        self._suspended += 1
        try:
            out = fn()
        finally:
            self._suspended -= 1
        self.entries.append((_HOST, fn, ops._get_cur(ops._raw_device())))  # (re-run with the same current stream)
        return out

    @contextlib.contextmanager
    def suspended(self):
        self._suspended += 1
        try:
            yield
        finally:
            self._suspended -= 1

    @contextlib.contextmanager
    def recording(self):
        if _lib.RECORDER is not None:
            raise RuntimeError("LaunchProgram: a recording is already active")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("LaunchProgram: not inside a hipGraph capture")
        prog = self
        ev_cls, t_cls = torch.cuda.Event, torch.Tensor
        orig_rec, orig_wait, orig_rs = ev_cls.record, ev_cls.wait, t_cls.record_stream
        main = ops._get_cur(ops._raw_device())
        if self.main is None:
            self.main, self.main_raw = tuple(main), ops._stream()
        elif tuple(main) != self.main:
            raise RuntimeError("LaunchProgram: recording continues on another stream than it started on")

        def rec(ev, stream=None):
            if stream is None:
                stream = ops.cur_stream()
            orig_rec(ev, stream)
            if not prog._suspended:
                prog.entries.append((_REC, ev, stream))

        def wait(ev, stream=None):
            if stream is None:
                stream = ops.cur_stream()
            orig_wait(ev, stream)
            if not prog._suspended:
                prog.entries.append((_WAIT, ev, stream))

        def record_stream(t, s):
            prog.keep.append(t)  # crossed streams: never recycled inside the recording (module docstring)
            return orig_rs(t, s)

        # A torch.cuda.MemPool that dies while allocations are routed to ANOTHER pool aborts the process (the allocator frees
        # a dead pool's blocks through synchronize_and_free_events, which asserts that no capture / pool routing is underway):
        # stale programs are collected BEFORE the routing starts, and the cyclic collector stays off while it lasts.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        ev_cls.record, ev_cls.wait, t_cls.record_stream = rec, wait, record_stream
        _lib.RECORDER = self
        try:
            with torch.cuda.use_mem_pool(self.pool, device=self.device), _AtenRecorder(self), torch.no_grad():
                yield self
        finally:
            _lib.RECORDER = None
            ev_cls.record, ev_cls.wait, t_cls.record_stream = orig_rec, orig_wait, orig_rs
            if gc_was_on:
                gc.enable()
            self._finish()

    def _finish(self):
        n = {k: 0 for k in (_CALL, _REC, _WAIT, _ATEN, _HOST)}
        streams = set()
        for e in self.entries:
            n[e[0]] += 1
            if e[0] == _CALL:
                streams.add(e[2][-1].value if hasattr(e[2][-1], "value") else None)
        self.stats = dict(launches=n[_CALL], event_records=n[_REC], event_waits=n[_WAIT], torch_ops=n[_ATEN],
                          host_callbacks=n[_HOST], streams=len(streams))
        self._compile()

    _SIG = {ctypes.c_int: "i", ctypes.c_long: "l", ctypes.c_size_t: "l", ctypes.c_ulonglong: "l", ctypes.c_void_p: "p",
            ctypes.c_float: "f", ctypes.c_double: "d", ctypes.c_uint: "i"}

    def _compile(self):
        """hand every run of C-ABI calls / event operations to the C-side executor (csrc/program.hip: one dana_program_run
        call re-issues the whole run); torch ops and host callbacks stay Python entries between the runs. Without libffi
        (dana_program_create fails) the replay walks the Python list: same launches, more host time."""
        L = _lib.lib()
        self._plan, self._c = list(self.entries), None
        if os.environ.get("DANA_PROGRAM_C", "1") == "0":
            return
        h = ctypes.c_void_p()
        try:
            L.call("dana_program_create", ctypes.cast(ctypes.byref(h), ctypes.c_void_p))
        except _lib.DanaError:
            return
        plan, begin, count = [], None, 0
        for e in self.entries:
            if e[0] in (_CALL, _REC, _WAIT):
                if e[0] == _CALL:
                    fn, args = e[1], e[2]
                    sig = "".join(self._SIG[t] for t in fn.argtypes)
                    words = (ctypes.c_ulonglong * max(len(args), 1))()
                    for k, (c, a) in enumerate(zip(sig, args)):
                        v = a.value
                        if c == "f":
                            words[k] = struct.unpack("<I", struct.pack("<f", v))[0]
                        elif c == "d":
                            words[k] = struct.unpack("<Q", struct.pack("<d", v))[0]
                        else:
                            words[k] = (int(v) if v is not None else 0) & 0xFFFFFFFFFFFFFFFF
                    L.call("dana_program_add_call", h, ctypes.cast(fn, ctypes.c_void_p), sig.encode(),
                           ctypes.cast(words, ctypes.c_void_p), len(args))
                else:
                    name = "dana_program_add_event_record" if e[0] == _REC else "dana_program_add_event_wait"
                    L.call(name, h, e[1].cuda_event, e[2].cuda_stream)
                if begin is None:
                    begin = count
                count += 1
            else:
                if begin is not None:
                    plan.append((_CRUN, begin, count))
                    begin = None
                plan.append(e)
        if begin is not None:
            plan.append((_CRUN, begin, count))
        self._plan, self._c = plan, h
        self._crun = L.fn["dana_program_run"]
        self.stats["c_segments"] = sum(1 for e in plan if e[0] == _CRUN)

    def __del__(self):
        h = getattr(self, "_c", None)
        if h is not None:
            try:
                _lib.lib().fn["dana_program_destroy"](h)
            except Exception:
                pass

    # ---- replay --------------------------------------------------------------------------------------------------
    def run(self):
        """replay on the streams the step was recorded on; the caller's current stream is joined in front and behind if
        it is not the recording's own"""
        cur = ops._get_cur(ops._raw_device())
        bridge = tuple(cur) != self.main
        if bridge:
            cur_s = torch.cuda.Stream(stream_id=cur[0], device_index=cur[1], device_type=cur[2])
            main_s = torch.cuda.Stream(stream_id=self.main[0], device_index=self.main[1], device_type=self.main[2])
            _EV_RECORD(self._entry_ev, cur_s)
            _EV_WAIT(self._entry_ev, main_s)
        set_cur, last = ops._set_cur, cur
        try:
            crun, h = getattr(self, "_crun", None), self._c
            for kind, a, b in self._plan:
                if kind == _CRUN:
                    if crun(h, a, b):
                        raise _lib.DanaError("replayed launch failed: %s" % _lib.lib().cdll.dana_last_error().decode())
                elif kind == _CALL:
                    if a(*b):
                        raise _lib.DanaError("replayed launch failed: %s" % _lib.lib().cdll.dana_last_error().decode())
                elif kind == _REC:
                    _EV_RECORD(a, b)
                elif kind == _WAIT:
                    _EV_WAIT(a, b)
                elif kind == _ATEN:
                    st = b[2]
                    if st != last:
                        set_cur(stream_id=st[0], device_index=st[1], device_type=st[2])
                        last = st
                    a(*b[0], **b[1])
                else:
                    if b != last:
                        set_cur(stream_id=b[0], device_index=b[1], device_type=b[2])
                        last = b
                    a()
        finally:
            if last != cur:
                set_cur(stream_id=cur[0], device_index=cur[1], device_type=cur[2])
        if bridge:
            _EV_RECORD(self._exit_ev, main_s)
            _EV_WAIT(self._exit_ev, cur_s)


def _static_like(t):
    return t.detach().clone() if torch.is_tensor(t) else t


class ProgramDAnA:
    """model(*inputs) as launch-program replays (train or eval mode, host or device RNG): the eager forward's launches on
    the eager forward's streams, minus its Python. Inputs are copied into static buffers (skipped when the caller passes
    `runner.inputs` themselves); outputs are static tensors, valid until the next call. Bakes in what GraphedDAnA bakes in
    (mode, shapes, cfg, the weight-derived tensors of this moment): re-record after changing any of them."""

    def __init__(self, model, *example_inputs, warmup=2):
        dev = example_inputs[0].device
        if dev.type != "cuda":
            raise RuntimeError("ProgramDAnA needs HIP tensors")
        if not hasattr(model, "_forward_gen"):
            raise RuntimeError("ProgramDAnA drives DAnARCNN (the siblings run eagerly)")
        self.model = model
        self.inputs = [_static_like(t) for t in example_inputs]
        with torch.no_grad():
            for _ in range(warmup):  # eager: fills the plan / constant caches, creates the role streams
                model(*self.inputs)
        torch.cuda.synchronize(dev)
        self._device_rng = bool(getattr(model, "device_rng", False)) and model.training
        if self._device_rng:
            # the Philox call counter must be DATA in a replay (a launch argument would redraw the recording's numbers):
            # the uint64 in device memory the hipGraph path uses, continuing the eager call sequence
            model._rng_counter(dev).fill_(2 * model._rng_calls)
            model._rng_counter_as_data = True
            torch.cuda.synchronize(dev)
        self.p1 = LaunchProgram(dev)
        self.p2 = self.req = self.drawn = None
        calls0 = model._rng_calls
        gen = model._forward_gen(*self.inputs)
        out = None
        try:
            with self.p1.recording():
                try:
                    req = next(gen)
                    if req["stage"] == "anchor":
                        req = next(gen)
                except StopIteration as done:
                    req, out = None, done.value
        finally:
            model._rng_counter_as_data = False
        if req is not None:
            assert req["stage"] == "draw"
            self.req = req
            self.drawn = torch.zeros((req["layout"]["words"],), dtype=torch.int32, device=dev)
            ops.draw_and_upload(req, dev, static=self.drawn)
            self.p2 = LaunchProgram(dev, pool=self.p1.pool)
            with self.p2.recording():
                try:
                    gen.send(self.drawn)
                    raise RuntimeError("the forward paused more often than expected")
                except StopIteration as done:
                    out = done.value
        self.outputs = out
        # host RNG: the recording drew nothing that counts. Device RNG: the recording WAS forward number calls0 (its
        # kernels ran and advanced the device counter), every replay is one more
        model._rng_calls = calls0 + (1 if self._device_rng else 0)
        torch.cuda.synchronize(dev)

    def __call__(self, *inputs):
        for s, t in zip(self.inputs, inputs):
            if torch.is_tensor(t) and t is not s:
                s.copy_(t, non_blocking=True)
        self.p1.run()
        if self.p2 is not None:
            ops.draw_and_upload(self.req, self.drawn.device, static=self.drawn)
            self.p2.run()
        if self._device_rng:
            self.model._rng_calls += 1  # (the replay advanced the device Philox counter by one forward)
        return self.outputs


class ProgramTrainer:
    """Trainer.step (train.py:125-143: zero_grad, forward, summed loss, backward, SGD) as two launch programs around the one
    host round trip. Multi-rank: the bucket all-reduces are host callbacks INSIDE the second program, issued where the eager
    backward issues them (no cut of the backward is needed, unlike the hipGraph form). SGD only; the learning rate is baked
    in: `rerecord()` after `trainer.adjust_learning_rate`.

        pt = ProgramTrainer(trainer, *example_inputs)
        out = pt.step(*inputs)          # the model's 8-tuple (static tensors)"""

    def __init__(self, trainer, *example_inputs, warmup=2):
        if trainer.optimizer != "sgd":
            raise RuntimeError("ProgramTrainer: SGD only (Adam's step count is a launch parameter)")
        self.trainer, self.model = trainer, trainer.model
        if not hasattr(self.model, "_forward_gen"):
            raise RuntimeError("ProgramTrainer drives DAnARCNN (the siblings train eagerly)")
        dev = example_inputs[0].device
        self.inputs = [_static_like(t) for t in example_inputs]
        snap = None
        if warmup > 0:  # eager warm-up iterations must not move the training trajectory (as in GraphedTrainer)
            import numpy as _np
            snap = ([fb.params.clone() for fb, _, _ in trainer.groups], [b.clone() for b in trainer.bufs], trainer.steps,
                    _np.random.get_state())
            for _ in range(warmup):
                trainer.step(*self.inputs)
            torch.cuda.synchronize(dev)
            for (fb, _, _), p0 in zip(trainer.groups, snap[0]):
                fb.params.copy_(p0)
            for b, b0 in zip(trainer.bufs, snap[1]):
                b.copy_(b0)
            trainer.steps = snap[2]
            _np.random.set_state(snap[3])
            self.model._epoch += 1
            torch.cuda.synchronize(dev)
        self._record()

    def rerecord(self):
        self._record()

    def _record(self):
        tr, model = self.trainer, self.model
        dev = self.inputs[0].device
        import numpy as _np
        # the recording IS an eager iteration (its launches really run): parameters, momentum, the step count and the host
        # RNG are put back behind it, so that constructing the runner does not move the training trajectory
        torch.cuda.synchronize(dev)
        snap = ([fb.params.clone() for fb, _, _ in tr.groups], [b.clone() for b in tr.bufs], _np.random.get_state())
        model._get_plan()      # frozen-weight packs / BN folds are made ONCE, outside the program ...
        model._epoch += 1      # ... every trainable conv's derived tensors are re-derived INSIDE it, from the live weights
        prev_save = getattr(model, "save_for_backward", False)
        model.save_for_backward = True
        self.p1 = self.p2 = self.outputs = None  # (a re-record: the old programs' pool dies here, outside any routing)
        self.p1 = LaunchProgram(dev)
        self.p2 = self.req = self.drawn = None
        for fb, _, _ in tr.groups:
            fb.zero_grad_bookkeeping()
        self._device_rng = bool(getattr(model, "device_rng", False))
        calls0 = model._rng_calls
        if self._device_rng:  # the Philox call counter as data in device memory (see ProgramDAnA)
            model._rng_counter(dev).fill_(2 * calls0)
            model._rng_counter_as_data = True
            torch.cuda.synchronize(dev)

        def backward_and_sgd(prog):
            # multi-rank: a bucket that becomes complete leaves on its all-reduce from a host callback (FlatBuckets._launch),
            # re-issued at the same position of every replay
            for fb, _, _ in tr.groups:
                fb.recorder = prog
            BW.model_backward(model, (1.0, 1.0, 1.0, 1.0))
            prog.host_callback(self._wait_buckets)
            for (fb, lr_mult, wd), buf in zip(tr.groups, tr.bufs):
                # first_step=False: with a zero momentum buffer  buf = m * 0 + g  IS torch.optim.SGD's first step
                ops.sgd_momentum_(fb.params, fb.grads, buf, tr.lr * lr_mult, tr.momentum, wd, grad_scale=1.0 / fb.world,
                                  first_step=False)

        try:
            gen = model._forward_gen(*self.inputs)
            out = None
            with self.p1.recording() as p1:
                for fb, _, _ in tr.groups:
                    fb.grads.zero_()
                try:
                    req = next(gen)
                    if req["stage"] == "anchor":
                        req = next(gen)
                except StopIteration as done:  # device RNG: no host round trip, the whole iteration is ONE program
                    req, out = None, done.value
                    backward_and_sgd(p1)
            if req is not None:
                assert req["stage"] == "draw"
                self.req = req
                self.drawn = torch.zeros((req["layout"]["words"],), dtype=torch.int32, device=dev)
                ops.draw_and_upload(req, dev, static=self.drawn)
                self.p2 = LaunchProgram(dev, pool=self.p1.pool)
                with self.p2.recording() as p2:
                    try:
                        gen.send(self.drawn)
                        raise RuntimeError("the forward paused more often than expected")
                    except StopIteration as done:
                        out = done.value
                    backward_and_sgd(p2)
        finally:
            model.save_for_backward = prev_save
            model._rng_counter_as_data = False
            for fb, _, _ in tr.groups:
                fb.recorder = None
        self.outputs = tuple(t.detach() if torch.is_tensor(t) else t for t in out)
        torch.cuda.synchronize(dev)
        for (fb, _, _), p0 in zip(tr.groups, snap[0]):
            fb.params.copy_(p0)
        for b_, b0 in zip(tr.bufs, snap[1]):
            b_.copy_(b0)
        _np.random.set_state(snap[2])
        model._rng_calls = calls0
        if self._device_rng:
            model._rng_counter(dev).fill_(2 * calls0)
        model._epoch += 1
        torch.cuda.synchronize(dev)

    def _wait_buckets(self):
        for fb, _, _ in self.trainer.groups:
            fb.wait_issued()

    def _after_step(self):
        self.trainer.steps += 1
        # the fused SGD wrote the weights through raw pointers and the program re-derived its OWN packed / Winograd copies
        # before this step's update: an eager forward that follows must re-derive them from the live weights
        self.model._epoch += 1

    def step(self, *inputs):
        for s, t in zip(self.inputs, inputs):
            if torch.is_tensor(t) and t is not s:
                s.copy_(t, non_blocking=True)
        for fb, _, _ in self.trainer.groups:
            fb.zero_grad_bookkeeping()
        self.p1.run()
        if self.p2 is not None:
            ops.draw_and_upload(self.req, self.drawn.device, static=self.drawn)
            self.p2.run()
        if self._device_rng:
            self.model._rng_calls += 1
        self._after_step()
        return self.outputs
