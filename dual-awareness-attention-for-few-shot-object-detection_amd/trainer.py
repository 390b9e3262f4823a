"""The reference's training iteration (train.py:125-143) on the HIP path, one process per GPU:

    zero_grad -> model(...) -> loss = sum of the four loss means -> loss.backward() -> optimizer.step()

* `FlatBuckets`: parameters, gradients and momentum live in flat fp32 buffers laid out in the order
  backward.model_backward FINISHES the gradients, cut into buckets at parameter boundaries. As soon as a bucket's
  gradients are final it leaves on an asynchronous all-reduce (RCCL over xGMI when the process group is `nccl`;
  `gloo` on CPU in the tests) while the rest of the backward still runs: the one exchange step of the data-parallel
  path (replaces nn.DataParallel's reduce-add onto GPU 0, train.py:104-105).
* `Trainer`: torch.optim.SGD(momentum) semantics of train.py:76-87 (2x learning rate and no weight decay for biases)
  as one fused HIP launch per flat segment; the 1/world_size of the gradient mean is folded into that launch.
"""
import torch
import torch.distributed as dist

from . import backward as BW
from . import ops
from .config import cfg


class FlatBuckets:
    """Flat storage for an ordered list of (name, tensor) + bucketed asynchronous all-reduce of the gradients."""

    def __init__(self, named_params, bucket_bytes=32 << 20, process_group=None, world_size=None, always_reduce=False):
        named_params = list(named_params)
        if not named_params:
            raise ValueError("FlatBuckets: no parameters")
        dev = named_params[0][1].device
        self.names = [n for n, _ in named_params]
        self.offsets, off = {}, 0
        for n, p in named_params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatBuckets: %s must be float32 on %s" % (n, dev))
            self.offsets[n] = (off, p.numel())
            off += (p.numel() + 3) // 4 * 4  # every parameter starts 16-byte aligned
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        # buckets: consecutive parameters up to bucket_bytes (a larger parameter is its own bucket)
        self.buckets, start, cur = [], 0, []
        for n in self.names:
            o, k = self.offsets[n]
            if cur and (o + k - start) * 4 > bucket_bytes:
                self.buckets.append((start, o, cur))
                start, cur = o, []
            cur.append(n)
        self.buckets.append((start, off, cur))
        self.bucket_of = {n: i for i, (_, _, ns) in enumerate(self.buckets) for n in ns}
        self.group = process_group
        self.world = world_size if world_size is not None else (
            dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1)
        self._pending = [len(ns) for _, _, ns in self.buckets]
        self._works = []
        self.launch_order = []
        # always_reduce: issue the collectives even in a 1-rank group (a single-GPU box still drives RCCL end to end)
        self.collective = (self.world > 1 or always_reduce) and dist.is_available() and dist.is_initialized()
        self._comm = None  # the stream every bucket's all_reduce is issued from (see _launch)
        for n, p in named_params:  # re-point parameter and gradient storage into the flat buffers
            o, k = self.offsets[n]
            if p.dim() == 4:
                # conv weights live in the kernels' own layout [O][KH][KW][I] (= channels-last strides of the OIHW
                # parameter): the forward reads them in place, the weight-gradient kernel accumulates into .grad in
                # place -- no pack / unpack launches per step; state_dict / load_state_dict see the logical OIHW tensor
                O, I, KH, KW = p.shape
                pv = self.params[o:o + k].view(O, KH, KW, I).permute(0, 3, 1, 2)
                gv = self.grads[o:o + k].view(O, KH, KW, I).permute(0, 3, 1, 2)
            else:
                pv, gv = self.params[o:o + k].view(p.shape), self.grads[o:o + k].view(p.shape)
            pv.copy_(p.detach())
            p.data = pv
            p.grad = gv

    def zero_grad(self):
        self.grads.zero_()
        self.zero_grad_bookkeeping()

    def zero_grad_bookkeeping(self):
        self._pending = [len(ns) for _, _, ns in self.buckets]
        self._works = []
        self.launch_order = []

    def mark_ready(self, names):
        """the gradients of `names` are final: launch every bucket that became complete"""
        for n in names:
            i = self.bucket_of.get(n)
            if i is None:
                continue  # frozen parameter
            self._pending[i] -= 1
            if self._pending[i] == 0:
                self._launch(i)
            elif self._pending[i] < 0:
                raise RuntimeError("FlatBuckets: %s marked ready twice" % n)

    def _launch(self, i):
        self.launch_order.append(i)
        if not self.collective or getattr(self, "capture_only", False):
            return  # (capture_only: graphs.GraphedTrainer records the order and issues the collectives between replays)
        rec = getattr(self, "recorder", None)
        if rec is not None:
            # program.ProgramTrainer: the collective is a HOST step inside the recorded backward -- issued now, and again at
            # this position (on this stream) of every replay
            rec.host_callback(lambda: self._works.append(self.reduce_bucket(i)))
            return
        self._works.append(self.reduce_bucket(i))

    def reduce_bucket(self, i):
        """issue bucket i's asynchronous all_reduce behind the current stream's work -> the work handle"""
        s, e, _ = self.buckets[i]
        if not self.grads.is_cuda:
            return dist.all_reduce(self.grads[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        # RCCL orders a collective behind whatever stream is CURRENT when it is issued. The gradients of a bucket are
        # final on the caller's stream at this point (backward joins its weight-gradient / box-branch side streams
        # into it before marking a stage ready), but which stream that is depends on where in the backward the stage
        # ends. So: pin that moment with an event and issue every bucket from ONE dedicated stream behind it -- the
        # exchange no longer depends on the caller's stream context, and all buckets (weights and biases) leave in one
        # program order on every rank.
        owner = getattr(self, "_comm_owner", None)
        if self._comm is None:
            if owner is not None and owner._comm is None:
                owner._comm = torch.cuda.Stream(device=self.grads.device)
            self._comm = owner._comm if owner is not None else torch.cuda.Stream(device=self.grads.device)
        final = ops.record_event()
        self._comm.wait_event(final)
        with ops.on_stream(self._comm):
            return dist.all_reduce(self.grads[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def join_comm(self):
        if self._comm is not None:
            ops.cur_stream().wait_stream(self._comm)

    def wait_all(self):
        """all buckets must have left; blocks the current stream (not the host, for nccl) until the sums arrived"""
        missing = [i for i, k in enumerate(self._pending) if k > 0]
        if missing:
            raise RuntimeError("FlatBuckets: gradients never marked ready in buckets %s (e.g. %s)"
                               % (missing, self.buckets[missing[0]][2][:3]))
        for w in self._works:
            w.wait()  # nccl: the CURRENT stream waits for the collective; gloo: the host does
        if self._works and self._comm is not None:
            ops.cur_stream().wait_stream(self._comm)
        self._works = []

    def wait_issued(self):
        """wait_all() for a replayed iteration (program.ProgramTrainer): the buckets were issued by the program's host
        callbacks, not through mark_ready, so there is no pending count to check"""
        for w in self._works:
            w.wait()
        if self._works and self._comm is not None:
            ops.cur_stream().wait_stream(self._comm)
        self._works = []

    def broadcast_params(self, src=0):
        if self.world > 1:
            dist.broadcast(self.params, src, group=self.group)


class Trainer:
    def __init__(self, model, lr, momentum=None, weight_decay=None, double_bias=None, bias_decay=None,
                 process_group=None, bucket_bytes=32 << 20, optimizer="sgd", always_reduce=False):
        if optimizer not in ("sgd", "adam"):  # train.py:84-87
            raise ValueError("optimizer must be 'sgd' or 'adam'")
        self.optimizer = optimizer
        self.model = model
        self.lr = float(lr)
        self.momentum = float(cfg.TRAIN.MOMENTUM if momentum is None else momentum)
        wd = float(cfg.TRAIN.WEIGHT_DECAY if weight_decay is None else weight_decay)
        double_bias = cfg.TRAIN.DOUBLE_BIAS if double_bias is None else double_bias
        bias_decay = cfg.TRAIN.BIAS_DECAY if bias_decay is None else bias_decay
        params = dict(model.named_parameters())
        order = [n for _, names in BW.grad_stages(model) for n in names if params[n].requires_grad]
        left = [n for n, p in params.items() if p.requires_grad and n not in set(order)]
        if left:
            raise RuntimeError("trainable parameters without a HIP gradient: %s" % left[:5])
        # train.py:79-84: 'bias' in key -> lr * (DOUBLE_BIAS + 1), weight decay only if BIAS_DECAY
        w_names = [n for n in order if "bias" not in n]
        b_names = [n for n in order if "bias" in n]
        self.weights = FlatBuckets([(n, params[n]) for n in w_names], bucket_bytes, process_group,
                                   always_reduce=always_reduce)
        self.biases = FlatBuckets([(n, params[n]) for n in b_names], bucket_bytes, process_group,
                                  always_reduce=always_reduce)
        self.biases._comm_owner = self.weights  # one issue stream for both groups
        self.groups = [(self.weights, 1.0, wd), (self.biases, float(double_bias) + 1.0, wd if bias_decay else 0.0)]
        self.bufs = [torch.zeros_like(fb.params) for fb, _, _ in self.groups]  # SGD momentum / Adam exp_avg
        self.bufs2 = [torch.zeros_like(fb.params) for fb, _, _ in self.groups] if optimizer == "adam" else None
        self.steps = 0
        if type(model).__name__ == "DAnARCNN" and not model.merge_trunk and getattr(model, "train_merged", True):
            # the training iteration keeps the query and the support batch in ONE set of activation buffers (the trunk's
            # launches stay two per conv on two streams: merge_from 3), so that the backward's 1x1 weight / data gradients
            # run once over both batches (backward.bottleneck_backward_merged). A preference for the forwards that save
            # for THIS trainer's backward only: the model's configured path (merge_trunk / merge_from) stays what every
            # eval / inference / bench forward takes; model.train_merged = False keeps the two-buffer form here as well.
            model._train_merge = (True, 3)
        model._plan = None  # parameter storage moved: re-pack on the next forward
        model._grad_ready_cb = self._on_ready
        for fb, _, _ in self.groups:
            fb.broadcast_params(0)  # one-time parameter broadcast (replicas start identical)
        if self.weights.world > 1:
            # ... and the frozen parameters / BatchNorm buffers, which nn.DataParallel re-replicates from GPU 0 on every
            # step (train.py:104-105): one coalesced broadcast of everything outside the flat buffers
            flat = set(w_names) | set(b_names)
            rest = [t for k, t in model.state_dict(keep_vars=True).items() if k not in flat and t.dtype.is_floating_point]
            if rest:
                buf = torch.cat([t.detach().reshape(-1) for t in rest])
                dist.broadcast(buf, 0, group=process_group)
                off = 0
                for t in rest:
                    t.detach().copy_(buf[off:off + t.numel()].view_as(t))
                    off += t.numel()

    def _on_ready(self, names):
        for fb, _, _ in self.groups:
            fb.mark_ready(names)

    def zero_grad(self):
        for fb, _, _ in self.groups:
            fb.zero_grad()

    def optimizer_step(self):
        from . import ops
        for (fb, lr_mult, wd), buf in zip(self.groups, self.bufs):
            fb.wait_all()
            if self.optimizer == "adam":
                ops.adam_(fb.params, fb.grads, buf, self.bufs2[self.groups.index((fb, lr_mult, wd))], self.lr * lr_mult,
                          self.steps + 1, weight_decay=wd, grad_scale=1.0 / fb.world)
            else:
                ops.sgd_momentum_(fb.params, fb.grads, buf, self.lr * lr_mult, self.momentum, wd,
                                  grad_scale=1.0 / fb.world, first_step=self.steps == 0)
        self.steps += 1
        self.model._epoch += 1  # weights changed under the packed / Winograd-transformed copies: re-derive those

    def step(self, *inputs):
        """one training iteration on the model's own forward arguments (DAnA: im_data, im_info, gt_boxes, num_boxes,
        support_ims; frcnn: without the supports); returns the model's 8-tuple (losses detached)"""
        self.zero_grad()
        model = self.model
        if type(model).__name__ == "DAnARCNN":
            # train.py:138-143 differentiates the plain sum of the four (scalar) losses: upstream gradients (1, 1, 1, 1).
            # Calling the HIP backward directly instead of through the autograd bridge (four .mean() launches, three adds,
            # the engine's thread hop) keeps the host ahead of the GPU at the forward -> backward hand-over, where the eager
            # iteration is issue-bound (the ORDER in which the backward's three chains are issued alone is worth 1.2 ms).
            prev = getattr(model, "save_for_backward", False)
            model.save_for_backward = True
            try:
                with torch.no_grad():
                    out = model(*inputs)
            finally:
                model.save_for_backward = prev
            BW.model_backward(model, (1.0, 1.0, 1.0, 1.0))
        else:
            with torch.enable_grad():
                out = model(*inputs)
                loss = out[3].mean() + out[4].mean() + out[5].mean() + out[6].mean()  # train.py:138-139
            loss.backward()
        self.optimizer_step()
        return out

    # ---- checkpoint / resume (train.py:92-101,181-189 save and restore model + optimizer state) -----------------
    def _views(self, flats):
        """name -> view of the flat optimizer buffers in the parameter's logical shape (conv weights live [O][KH][KW][I] in
        the flat buffers, the checkpoint speaks OIHW like torch.optim's momentum_buffer entries)"""
        out, params = {}, dict(self.model.named_parameters())
        for (fb, _, _), flat in zip(self.groups, flats):
            for n in fb.names:
                o, k = fb.offsets[n]
                p = params[n]
                if p.dim() == 4:
                    O, I, KH, KW = p.shape
                    out[n] = flat[o:o + k].view(O, KH, KW, I).permute(0, 3, 1, 2)
                else:
                    out[n] = flat[o:o + k].view(p.shape)
        return out

    def state_dict(self):
        """what a resumed run needs besides model.state_dict(): decayed lr, momentum, step count (Adam's bias correction,
        SGD's first-step rule) and the momentum / moment buffers per parameter name"""
        out = {"lr": self.lr, "momentum": self.momentum, "steps": self.steps, "optimizer": self.optimizer,
               "momentum_buffer": {n: v.detach().clone().contiguous() for n, v in self._views(self.bufs).items()}}
        if self.optimizer == "adam":
            out["exp_avg_sq"] = {n: v.detach().clone().contiguous() for n, v in self._views(self.bufs2).items()}
        return out

    def load_state_dict(self, state):
        if state.get("optimizer", self.optimizer) != self.optimizer:
            raise ValueError("checkpoint holds %s state, this trainer runs %s" % (state.get("optimizer"), self.optimizer))
        mom = self._views(self.bufs)
        missing = sorted(set(mom) - set(state["momentum_buffer"]))
        if missing:
            raise KeyError("optimizer checkpoint lacks %d of %d parameters, e.g. %s" % (len(missing), len(mom), missing[:3]))
        self.lr, self.momentum, self.steps = float(state["lr"]), float(state["momentum"]), int(state["steps"])
        for n, v in mom.items():
            v.copy_(state["momentum_buffer"][n])
        if self.optimizer == "adam":
            for n, v in self._views(self.bufs2).items():
                v.copy_(state["exp_avg_sq"][n])

    def adjust_learning_rate(self, decay=0.1):
        """net_utils.adjust_learning_rate (train.py:118-120)"""
        self.lr *= decay
