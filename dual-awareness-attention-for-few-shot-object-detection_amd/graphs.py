"""hipGraph replay of the DAnA forward and of the whole training iteration.

Eagerly, one bs-4 train-mode forward is ~430 C-ABI calls issued from Python (6.5 ms of host time for 7.5 ms of GPU
work in round 1), a training iteration ~1 500: with one process per GPU and eight of them on one host that is the wall
every further kernel gain runs into. The launch sequence is static -- shapes, pointers and parameters do not depend on
the data -- except for ONE host round trip: the reference's `np.random` draws in the two target layers need the fg / bg
counts (anchor_target_layer.py:137-156, proposal_target_layer_cascade.py:143-175). So the step is captured as

    G1 = trunk .. RPN .. proposals .. target layers up to their counts
    [host: read 4*B counts, np.random draws (the reference's stream), ONE pinned upload into a static buffer]
    G2 = RPN losses, sampled-batch gather, RoIAlign, layer4, attention heads, RCNN losses
         (+ backward + fused SGD for the training iteration)

and replayed with two graph launches per step. With `model.device_rng = True` (Philox on the device, call counter in
device memory) there is no host round trip and the whole step is one graph. Multi-rank training cuts the backward
graph once more, where the gradients of everything except the trunk are final, so that the RCCL all-reduce of those
buckets overlaps the trunk's backward (the eager path launches bucket by bucket; a collective cannot sit inside a
captured graph portably).

Capture uses torch's hipGraph wrapper (`torch.cuda.CUDAGraph`) for stream capture, the private memory pool and replay;
every node in the graphs is one of this package's HIP kernels (plus a handful of memset / copy nodes).
"""
import torch

from . import ops
from . import backward as BW


def _static_like(t):
    return t.detach().clone() if torch.is_tensor(t) else t


class GraphedDAnA:
    """model(*inputs) as hipGraph replays. Inputs are copied into static buffers (skipped when the caller passes the
    static buffers themselves: `runner.inputs`); outputs are static tensors, valid until the next call.

    The graphs bake in: the model's mode (train / eval), input shapes, cfg, and the weight-derived tensors of this
    moment (packed / Winograd-transformed weights, folded BN). Re-capture (`GraphedDAnA(model, ...)` again) after
    changing any of them; for training use `GraphedTrainer`, whose graphs re-derive them from the live weights."""

    def __init__(self, model, *example_inputs, warmup=2):
        dev = example_inputs[0].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedDAnA needs HIP tensors")
        if not hasattr(model, "_forward_gen"):
            raise RuntimeError("GraphedDAnA drives DAnARCNN (the siblings run eagerly)")
        self.model = model
        self.inputs = [_static_like(t) for t in example_inputs]
        self.stream = torch.cuda.Stream(device=dev)
        self.req = self.drawn = self.g2 = None
        cur = ops.cur_stream()
        self.stream.wait_stream(cur)
        with ops.on_stream(self.stream), torch.no_grad():
            for _ in range(warmup):  # eager: fills the plan / constant caches, sets kernel attributes
                model(*self.inputs)
        torch.cuda.synchronize(dev)
        if getattr(model, "device_rng", False):
            model._rng_counter(dev).fill_(2 * model._rng_calls)  # continues the eager call sequence
        # the forward that runs DURING capture bumps the host-side call count but draws nothing (its kernels only run in the
        # replays, which advance the device counter themselves): the count is put back behind the capture
        calls0 = model._rng_calls
        torch.cuda.synchronize(dev)
        self.side = torch.cuda.Stream(device=dev)  # the anchor-target graph replays here, beside the trunk
        self.g0 = None
        opts = dict(stream=self.stream, capture_error_mode="thread_local")
        with torch.no_grad():
            gen = model._forward_gen(*self.inputs)
            out = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **opts):
                try:
                    req = next(gen)
                except StopIteration as done:  # eval mode / device RNG: the whole forward is one graph
                    req, out = None, done.value
            if req is not None and req["stage"] == "anchor":
                self.g0, g = g, torch.cuda.CUDAGraph()  # graph 0: the anchor targets' first half (inputs only)
                with torch.cuda.graph(g, pool=self.g0.pool(), **opts):
                    req = next(gen)
            self.g1 = g
            if req is not None:
                assert req["stage"] == "draw"
                self.req = dict(req, anchor_stream=self.side)
                self.drawn = torch.zeros((req["layout"]["words"],), dtype=torch.int32, device=dev)
                self.g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g2, pool=self.g1.pool(), **opts):
                    try:
                        gen.send(self.drawn)
                        raise RuntimeError("the forward paused more often than expected")
                    except StopIteration as done:
                        out = done.value
        self.outputs = out
        model._rng_calls = calls0
        cur.wait_stream(self.stream)

    def __call__(self, *inputs):
        for s, t in zip(self.inputs, inputs):
            if torch.is_tensor(t) and t is not s:
                s.copy_(t, non_blocking=True)
        cur = ops.cur_stream()
        if self.g0 is not None:
            self.side.wait_stream(cur)
            with ops.on_stream(self.side):
                self.g0.replay()
        self.g1.replay()
        if self.g0 is None and self.g2 is None and getattr(self.model, "device_rng", False) and self.model.training:
            # the replay advanced the DEVICE Philox counter by one forward; keep the host count in step, so that an eager
            # device-RNG forward (or a later capture, which refills the device counter from it) continues the stream
            # instead of repeating earlier draws
            self.model._rng_calls += 1
        if self.g2 is not None:
            # anchor counts: behind the side stream only -> the anchor draws overlap the trunk (graph 1) on the GPU
            ops.draw_and_upload(self.req, self.drawn.device, static=self.drawn)
            cur.wait_stream(self.side)
            self.g2.replay()
        return self.outputs


class GraphedTrainer:
    """Trainer.step (train.py:125-143: zero_grad, forward, summed loss, backward, SGD) as hipGraph replays.

        gt = GraphedTrainer(trainer, *example_inputs)
        out = gt.step(*inputs)          # the model's 8-tuple (static tensors)

    SGD only (train.py's default; Adam's bias correction is a per-step launch parameter). The learning rate is baked
    in: call `recapture()` after `trainer.adjust_learning_rate`. The graphs re-derive every weight-dependent tensor
    (Winograd-domain filters, data-gradient weights, the merged RPN head) from the live flat parameter buffer on every
    replay, so the optimizer's in-place updates are picked up without re-capturing."""

    def __init__(self, trainer, *example_inputs, warmup=2):
        if trainer.optimizer != "sgd":
            raise RuntimeError("GraphedTrainer: SGD only (Adam's step count is a launch parameter)")
        self.trainer = trainer
        self.model = trainer.model
        if not hasattr(self.model, "_forward_gen"):
            raise RuntimeError("GraphedTrainer drives DAnARCNN (the siblings train eagerly)")
        dev = example_inputs[0].device
        self.inputs = [_static_like(t) for t in example_inputs]
        self.stream = torch.cuda.Stream(device=dev)
        self.ones = torch.ones(4, dtype=torch.float32, device=dev)  # d(sum of the four losses) / d(each)
        cur = ops.cur_stream()
        self.stream.wait_stream(cur)
        # eager warm-up iterations (caches, kernel attributes, allocator pools) must not move the training trajectory:
        # parameters, optimizer state and the step count are put back afterwards (warmup=0: the caller's own first
        # iterations are the warm-up)
        snap = None
        if warmup > 0:
            import numpy as _np
            snap = ([fb.params.clone() for fb, _, _ in trainer.groups], [b.clone() for b in trainer.bufs], trainer.steps,
                    self.model._rng_calls, _np.random.get_state())  # (host-RNG mode draws np.random in the warm-up too)
        with ops.on_stream(self.stream):
            for _ in range(warmup):
                trainer.step(*self.inputs)
        torch.cuda.synchronize(dev)
        if snap is not None:
            for (fb, _, _), p0 in zip(trainer.groups, snap[0]):
                fb.params.copy_(p0)
            for b, b0 in zip(trainer.bufs, snap[1]):
                b.copy_(b0)
            trainer.steps, self.model._rng_calls = snap[2], snap[3]
            _np.random.set_state(snap[4])
            self.model._epoch += 1
            torch.cuda.synchronize(dev)
        cur.wait_stream(self.stream)
        self._capture()

    def recapture(self):
        self._capture()

    def _capture(self):
        tr, model = self.trainer, self.model
        dev = self.inputs[0].device
        self.graphs = []     # [[graph, [(FlatBuckets, bucket index) to all-reduce after it]]]
        self.req = self.drawn = self.g_anchor = None
        self.side = torch.cuda.Stream(device=dev)
        model._epoch += 1    # every trainable conv's derived tensors are re-derived INSIDE the capture
        model._plan = None
        prev_save = getattr(model, "save_for_backward", False)
        model.save_for_backward = True
        collective = tr.weights.collective
        for fb, _, _ in tr.groups:
            fb.capture_only = True
            fb.zero_grad_bookkeeping()
        pool = [None]

        def graph():
            g = torch.cuda.CUDAGraph()
            self.graphs.append([g, []])
            kw = {} if pool[0] is None else {"pool": pool[0]}
            if pool[0] is None and self.g_anchor is not None:
                kw = {"pool": self.g_anchor.pool()}
            return torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local", **kw)

        def ready_buckets(seen):
            now = [(fb, i) for fb, _, _ in tr.groups for i in fb.launch_order]
            fresh = [x for x in now if x not in seen]
            seen.extend(fresh)
            return fresh

        # the device counter continues the host-side call sequence (eager forwards and replays both advance the host count;
        # the capture's own forward does not draw: its count is put back below), on the first capture and on a re-capture
        if getattr(model, "device_rng", False):
            model._rng_counter(dev).fill_(2 * model._rng_calls)
        calls0 = model._rng_calls
        torch.cuda.synchronize(dev)
        try:
            with torch.no_grad():
                gen = model._forward_gen(*self.inputs)
                out, bgen, seen, req = None, None, [], None
                first = True
                # ---- graph 0 (host RNG only): the anchor targets' first half; replayed on the side stream ----
                # ---- graph 1: zero the gradients, forward up to the host round trip (or to its end) ----
                while out is None and (req is None or req["stage"] != "draw"):
                    with graph():
                        if not first or self.model.device_rng:
                            for fb, _, _ in tr.groups:
                                fb.grads.zero_()
                        try:
                            req = next(gen)
                        except StopIteration as done:
                            out = done.value
                        if out is not None:
                            bgen = self._backward_until_cut(model._ctx, collective)
                    if first and req is not None and req["stage"] == "anchor":
                        self.g_anchor = self.graphs.pop()[0]
                        pool[0] = self.g_anchor.pool()
                    else:
                        pool[0] = self.graphs[0][0].pool()
                    first = False
                if out is None:
                    # ---- graph 2: behind the draws: the rest of the forward, then the backward ----
                    self.req = dict(req, anchor_stream=self.side)
                    self.drawn = torch.zeros((req["layout"]["words"],), dtype=torch.int32, device=dev)
                    with graph():
                        try:
                            gen.send(self.drawn)
                            raise RuntimeError("the forward paused more often than expected")
                        except StopIteration as done:
                            out = done.value
                        bgen = self._backward_until_cut(model._ctx, collective)
                if collective:
                    # multi-rank: the first buckets leave here; the trunk's backward is its own graph, replayed while
                    # they travel; the SGD launch sits behind every bucket's sum in a last graph
                    self.graphs[-1][1] = ready_buckets(seen)
                    with graph():
                        for _ in bgen:
                            raise RuntimeError("model_backward_gen paused twice")
                    self.graphs[-1][1] = ready_buckets(seen)
                    with graph():
                        self._sgd()
        finally:
            model.save_for_backward = prev_save
            model._rng_calls = calls0  # (the capture's forward drew nothing)
            for fb, _, _ in tr.groups:
                fb.capture_only = False
                fb.zero_grad_bookkeeping()
        self.outputs = tuple(t.detach() if torch.is_tensor(t) else t for t in out)
        self.collective = collective
        ops.cur_stream().wait_stream(self.stream)

    def _sgd(self):
        tr = self.trainer
        for (fb, lr_mult, wd), buf in zip(tr.groups, tr.bufs):
            # first_step=False: with a zero momentum buffer  buf = m * 0 + g  IS torch.optim.SGD's first step
            ops.sgd_momentum_(fb.params, fb.grads, buf, tr.lr * lr_mult, tr.momentum, wd, grad_scale=1.0 / fb.world,
                              first_step=False)

    def _backward_until_cut(self, ctx, collective):
        """single rank: the whole backward + SGD into the current capture (returns None). Multi-rank: the backward up to
        the point where everything but the trunk is final; returns the paused generator"""
        if not collective:
            BW.model_backward(self.model, self.ones, ctx=ctx)
            self._sgd()
            return None
        gen = BW.model_backward_gen(self.model, self.ones, ctx=ctx)
        next(gen)
        return gen

    def step(self, *inputs):
        tr = self.trainer
        for s, t in zip(self.inputs, inputs):
            if torch.is_tensor(t) and t is not s:
                s.copy_(t, non_blocking=True)
        works = []
        last = len(self.graphs) - 1
        cur = ops.cur_stream()
        if self.g_anchor is not None:
            self.side.wait_stream(cur)
            with ops.on_stream(self.side):
                self.g_anchor.replay()
        for k, (g, buckets) in enumerate(self.graphs):
            if k == 1 and self.req is not None:
                ops.draw_and_upload(self.req, self.drawn.device, static=self.drawn)
                cur.wait_stream(self.side)
            if k == last and self.collective:
                for w in works:  # the SGD graph runs behind every bucket's sum
                    w.wait()
                for fb, _, _ in tr.groups:
                    fb.join_comm()
            g.replay()
            for fb, i in buckets:
                works.append(fb.reduce_bucket(i))
        tr.steps += 1
        if getattr(self.model, "device_rng", False):
            self.model._rng_calls += 1  # (the replay advanced the device Philox counter by one forward: see GraphedDAnA)
        # the fused SGD wrote the weights through raw pointers (no tensor._version bump) and the graph re-derived its OWN
        # packed / Winograd / two-segment copies at the start of the replay, i.e. before this step's update: an eager
        # forward that follows (per-epoch eval, Trainer.step) must re-derive them from the live weights
        self.model._epoch += 1
        self.model._plan = None
        return self.outputs
