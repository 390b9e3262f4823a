"""Backward pass of the DAnA forward path on the HIP kernels (groundwork for the training step, SURVEY.md 8d
variant S). What `loss.backward()` does through autograd + cuDNN/cuBLAS in the reference (train.py:141-143) is
assembled here from the C-ABI building blocks: data gradients on the forward implicit-GEMM kernel with
transformed weights, weight gradients on the split-M TN MFMA kernel, and the element-wise adjoints of
backward.hip. Frozen BatchNorm (dana.py:362-385) only scales gradients; BN parameters, conv1/bn1/layer1 get none
(dana.py:350-360, cfg.RESNET.FIXED_BLOCKS = 1).

`model_backward` is the whole-model adjoint: it consumes the context a `save_for_backward` forward left in
`model._ctx` and accumulates `.grad` on every trainable parameter, checked against autograd of the oracle
(tests/test_gpu_backward.py)."""
import math

import torch

from . import ops
from .config import cfg


class WeightGrads:
    """Accumulates packed weight gradients per conv (query and support passes share the weights).
    With `stream`, the weight-gradient launches go to that side stream: they only consume (g, x) and nothing on the
    data-gradient chain waits for them, so their tiles fill the CUs the chain's launches leave idle in their tails."""

    def __init__(self, stream=None, model=None):
        self.packed = {}
        self.convs = {}
        self.stream = stream
        self.model = model
        self.direct = {}  # key -> packed VIEW of param.grad (trainer layout): weight gradients land there directly
        # ONE weight-gradient stream for every caller stream (the box branch issues its weight gradients from the layer4
        # stream, the RPN chain from its own, the rest of the backward from the caller's). Round 4 gave every caller stream
        # a side stream of its own; measured in round 5 (profiles/r5_role_streams.md): more streams than hardware queues
        # make unrelated chains share a queue, and a second weight-gradient stream costs the iteration 1.3-1.7 ms.
        self.side = {}    # caller stream handle -> [side stream, operands kept alive while its launches are in flight]
        self.compact = {}  # gathered input rows of strided 1x1 convs (shared by a block's conv1 and downsample conv)

    def _side_for_current(self):
        cur = ops.cur_stream()
        ent = self.side.get(cur.cuda_stream)
        if ent is None:
            ent = self.side[cur.cuda_stream] = [self.stream, []]
        return ent

    def _direct_view(self, key):
        """the parameter's gradient as a packed [cout][kh*kw*cin] view, if the trainer stores it that way"""
        if key in self.direct:
            return self.direct[key]
        v = None
        if self.model is not None:
            g = self.model.get_parameter(key + ".weight").grad
            if g is not None:
                v = ops.packed_view(g)
                if v is not None and v.data_ptr() != g.data_ptr():
                    v = None
        self.direct[key] = v
        return v

    def add_conv(self, key, g, x, n, h, w, c, in_stride=0, grad_stride=0, v=None):
        """v: the forward launch's kept Winograd workspace (V planes of x), if that conv ran in the F(4x4,3x3) domain"""
        self.convs[key] = c
        if self.stream is None:
            self._launch(key, g, x, n, h, w, c, in_stride, grad_stride, v)
            return
        st, keep = self._side_for_current()
        ready = ops.record_event()
        st.wait_event(ready)
        keep.append((g, x, v))
        with ops.on_stream(st):
            self._launch(key, g, x, n, h, w, c, in_stride, grad_stride, v)

    def side_run(self, fn, *keep):
        """fn() on the weight-gradient stream of the caller's stream, behind everything queued so far; `keep` stays
        alive until join(). For work that consumes the chain's gradients and feeds nothing back into it (a Linear's
        dW / db and their accumulation): off the data-gradient chain, joined with the conv weight gradients."""
        if self.stream is None:
            fn()
            return
        st, kept = self._side_for_current()
        ready = ops.record_event()
        st.wait_event(ready)
        kept.append(keep)
        with ops.on_stream(st):
            fn()

    def linear(self, g, x, m, n, k, then, ldx=0, ldg=0):
        """dW / db of a Linear (ops.linear_wgrad) on the side stream; then(dw, db) accumulates them there"""
        # eager issue: on the caller's chain (with the RPN chain, the box branch and the heads running from the backward's
        # start the side stream costs 0.3 ms, round 4); under stream capture the side stream (the box branch shares the
        # caller's stream there)
        if not torch.cuda.is_current_stream_capturing():
            then(*ops.linear_wgrad(g, x, m, n, k, ldx=ldx, ldg=ldg))
            return
        self.side_run(lambda: then(*ops.linear_wgrad(g, x, m, n, k, ldx=ldx, ldg=ldg)), g, x)

    def _launch(self, key, g, x, n, h, w, c, in_stride, grad_stride, v=None):
        view = self._direct_view(key)
        if c["k"] == 1 and c["stride"] > 1 and c["pad"] == 0:
            # a strided 1x1 conv (first block of layer2-4: conv1 and the downsample conv read the same pixels): gather those
            # pixels once into plain rows -- both weight gradients then run on the software-pipelined plain-row kernel
            ck = (x.data_ptr(), n, h, w, c["cin"], c["stride"], in_stride, ops.cur_stream().cuda_stream)
            ent = self.compact.get(ck)
            if ent is None:
                ent = self.compact[ck] = (ops.downsample_gather(x, n, h, w, c["cin"], c["stride"], in_stride), x)
            (x, h, w), in_stride = ent[0], 0
            c = dict(c, stride=1)
        u = c.get("u")
        if u is not None and u.size(0) == 36 and c["cin"] % 64 == 0:
            # the conv ran in the F(4x4,3x3) domain forward: so does its weight gradient (4x fewer multiplies)
            out = view if view is not None else self.packed.get(key)
            res = ops.conv3x3_wgrad_winograd(g, x, n, h, w, c["cin"], c["cout"], in_stride=in_stride,
                                             grad_stride=grad_stride, out=out,
                                             row_scale=c.get("scale") if view is not None else None,
                                             v=v)
            if out is None:
                self.packed[key] = res
            return
        if view is not None:  # scale by the frozen BN and accumulate straight into param.grad: nothing to finish
            ops.conv2d_wgrad(g, x, n, h, w, c["cin"], c["cout"], c["k"], c["k"], c["stride"], c["pad"],
                             in_stride=in_stride, grad_stride=grad_stride, out=view, row_scale=c.get("scale"))
            return
        buf = self.packed.get(key)
        if buf is None:
            self.packed[key] = ops.conv2d_wgrad(g, x, n, h, w, c["cin"], c["cout"], c["k"], c["k"], c["stride"],
                                                c["pad"], in_stride=in_stride, grad_stride=grad_stride)
        else:
            ops.conv2d_wgrad(g, x, n, h, w, c["cin"], c["cout"], c["k"], c["k"], c["stride"], c["pad"],
                             in_stride=in_stride, grad_stride=grad_stride, out=buf)

    def join(self):
        """the caller's stream waits for every weight-gradient launch issued so far"""
        if self.stream is None:
            self.compact.clear()
            del _FRESH.grads[:]
            return
        cur = ops.cur_stream()
        for t in _FRESH.grads:
            t.record_stream(cur)
        del _FRESH.grads[:]
        # the packed gradients were allocated in the weight-gradient stream's pool and are finished on the CALLER's stream
        # (finish_conv): without this a buffer popped there goes back to that pool while the caller's kernel still reads it --
        # harmless as long as every caller had a weight-gradient stream to itself, a wrong RPN_Conv gradient with the one
        # shared stream of round 5 (the trunk's next weight gradient took the block)
        for t in self.packed.values():
            t.record_stream(cur)
        for st, keep in self.side.values():
            if keep:
                done = torch.cuda.Event()
                done.record(st)
                cur.wait_event(done)
                del keep[:]
        self.compact.clear()

    def finish_conv(self, key, c, param):
        """apply the frozen-BN scale to the rows and add into param.grad (OIHW)"""
        buf = self.packed.pop(key)
        if c.get("scale") is not None:
            ops.rowscale_(buf, c["scale"], c["cout"], c["k"] * c["k"] * c["cin"])
        fresh = param.grad is None
        if fresh:
            param.grad = torch.empty_like(param)
        ops.unpack_conv_weight_grad(buf, param.grad, c["cout"], c["cin"], c["k"], c["k"], accumulate=not fresh)

    def finish_all(self, model, prefix=""):
        self.join()
        for key in list(self.packed):
            if key.startswith(prefix):
                self.finish_conv(key, self.convs[key], model.get_parameter(key + ".weight"))


def _dgrad_weights(c):
    """flipped / transposed / BN-scaled weights of the data-gradient conv (+ their Winograd transform when the
    forward conv took the Winograd path), cached in the plan entry: computed once per weight update"""
    if c.get("wd") is None:
        c["wd"] = ops.conv2d_dgrad_weight(c["w"], c["cout"], c["cin"], c["k"], c["k"], c.get("scale"))
        c["ud"] = None
        if c.get("u") is not None and c["cout"] % 64 == 0:
            c["ud"] = ops.winograd_filter_transform(c["wd"], c["cin"], c["cout"], 2 if c["u"].size(0) == 16 else 4)
    return c["wd"], c["ud"]


def conv_dgrad(g, n, h, w, c, residual=None, mask=None, compact_out=False, out=None):
    """dL/d(conv input) [n*h*w][cin] from g = dL/d(conv+BN output); + residual, then the ReLU adjoint of `mask`"""
    wd, ud = _dgrad_weights(c)
    return ops.conv2d_dgrad(g, c["w"], n, h, w, c["cin"], c["cout"], c["k"], c["k"], c["stride"], c["pad"], wd=wd, ud=ud,
                            residual=residual, mask=mask, compact_out=compact_out, out=out)


def conv_backward(g, x, n, h, w, c, grads, key, need_dx=True, in_stride=0):
    """g: gradient w.r.t. the conv+BN output [n*oh*ow][cout] (ReLU mask already applied).
    Records dW (raw, scale applied at finish) and returns dx [n*h*w][cin] (or None)."""
    grads.add_conv(key, g, x, n, h, w, c, in_stride=in_stride)
    return conv_dgrad(g, n, h, w, c) if need_dx else None


def bottleneck_backward(g, saved, n, h, w, bp, grads, key, need_dx=True, mask_dx=True, g_masked=False):
    """Adjoint of DAnARCNN._bottleneck. saved = dict(x, o1, o2, o3, h1, w1) from the forward; g = dL/d(o3), with the
    final ReLU's adjoint (resnet.py:100) already applied when g_masked. Returns dL/dx [n*h*w][cin] (None if not
    needed); with mask_dx the ReLU adjoint of the layer that produced x is already applied to it (x is then the
    previous bottleneck's output, so the caller passes g_masked=True there). Every ReLU adjoint and the residual
    sum ride in the epilogue of a data-gradient conv."""
    h1, w1 = saved["h1"], saved["w1"]
    m_out = n * h1 * w1
    cout = bp["c3"]["cout"]
    if not g_masked:
        ops.relu_mask_(g, saved["o3"], m_out, cout, ld_act=saved.get("o3_ld", 0))
    x = saved["x"]
    grads.add_conv(key + ".conv3", g, saved["o2"], n, h1, w1, bp["c3"])
    g2 = conv_dgrad(g, n, h1, w1, bp["c3"], mask=saved["o2"])
    grads.add_conv(key + ".conv2", g2, saved["o1"], n, h1, w1, bp["c2"], v=saved.get("v2"))
    g1 = conv_dgrad(g2, n, h1, w1, bp["c2"], mask=saved["o1"])
    grads.add_conv(key + ".conv1", g1, x, n, h, w, bp["c1"])
    if bp["ds"] is not None:
        grads.add_conv(key + ".downsample.0", g, x, n, h, w, bp["ds"])
    if not need_dx:
        return None
    mk = x if mask_dx else None
    if bp["ds"] is None:  # identity shortcut (resnet.py:96-99): dx = dgrad(conv1) + g
        return conv_dgrad(g1, n, h, w, bp["c1"], residual=g, mask=mk)
    if bp["c1"]["stride"] == 1:
        dxr = conv_dgrad(g, n, h, w, bp["ds"])
        return conv_dgrad(g1, n, h, w, bp["c1"], residual=dxr, mask=mk)
    # both 1x1 convs are strided (resnet.py:71, downsample): sum the compact gradients, scatter once
    cr = conv_dgrad(g, n, h, w, bp["ds"], compact_out=True)
    return conv_dgrad(g1, n, h, w, bp["c1"], residual=cr, mask=mk)


def bottleneck_backward_merged(g, sq, ss, sm, bp, grads, key):
    """bottleneck_backward for an identity-shortcut block (no downsample, stride 1) over the [query | support] buffers of
    DAnARCNN._rcnn_base_dual: g = dL/d(o3) of BOTH batches in one [Mq + Ms][cout] tensor, ReLU adjoint applied. The three
    1x1 convs are row-wise contractions -- their weight gradients and data gradients run ONCE over all rows (half the
    launches, twice the reduction length per weight-gradient slice); the 3x3 conv in the middle keeps one call per batch
    (its Winograd tiles follow the image geometry), writing into the two row ranges of one buffer. -> dL/dx, merged, with
    the ReLU adjoint of the layer that produced x applied."""
    mq, mt = sm["mq_out"], sm["m_out"]
    c1, c2, c3 = bp["c1"], bp["c2"], bp["c3"]
    grads.add_conv(key + ".conv3", g, sm["o2"], 1, mt, 1, c3)
    g2 = conv_dgrad(g, 1, mt, 1, c3, mask=sm["o2"])
    g1 = torch.empty((mt, c2["cin"]), dtype=torch.float32, device=g.device)
    ud = _dgrad_weights(c2)[1]
    dual = ud is not None and ud.size(0) == 36
    for part, s_ in ((slice(0, mq), sq), (slice(mq, mt), ss)):
        grads.add_conv(key + ".conv2", g2[part], sm["o1"][part], s_["n"], s_["h1"], s_["w1"], c2, v=s_.get("v2"))
        if not dual:
            conv_dgrad(g2[part], s_["n"], s_["h1"], s_["w1"], c2, mask=sm["o1"][part], out=g1[part])
    if dual:
        # both batches' 3x3 data gradients as ONE batched plane GEMM (two input / output transforms around it): the
        # 600-tile launches of one batch leave a third of the chip's slots empty
        ops.conv3x3_winograd_dual_dgrad(g2, sq["n"], sq["h1"], sq["w1"], ss["n"], ss["h1"], ss["w1"], c2["cout"], ud,
                                        c2["cin"], mask=sm["o1"], out=g1)
    grads.add_conv(key + ".conv1", g1, sm["x"], 1, mt, 1, c1)
    return conv_dgrad(g1, 1, mt, 1, c1, residual=g, mask=sm["x"])


def _block_convs(prefix, bp):
    names = [prefix + ".conv3.weight", prefix + ".conv2.weight", prefix + ".conv1.weight"]
    if bp["ds"] is not None:
        names.append(prefix + ".downsample.0.weight")
    return names


def grad_stages(model, plan=None):
    """[(stage, [parameter names])] in the order model_backward FINISHES the gradients: the trainer lays its flat
    gradient buffer out in this order so that all-reduce buckets can leave while the rest of the backward runs.
    plan: the forward's plan (the backward passes the one its context saved: asking the model for a plan INSIDE the
    backward -- autograd runs it with gradients disabled, i.e. not `_live()` -- re-derived and pre-split every trainable
    weight once per iteration for nothing: ~100 small launches at the head of the backward)."""
    if type(model).__name__ in ("FasterRCNN", "MetaRCNN", "FGN", "FSOD"):
        return frcnn_grad_stages(model, plan)
    plan = plan if plan is not None else model._get_plan()
    lin = lambda n: [n + ".weight", n + ".bias"]  # noqa: E731
    st = [("box branch", lin("RCNN_bbox_pred") + [n for bi in (2, 1, 0)
                                                 for n in _block_convs("RCNN_top.0.%d" % bi, plan["layer4"][bi])])]
    st.append(("roi heads", lin("output_score_layer.linear2") + lin("output_score_layer.linear1")
               + lin("rcnn_adapt_q_layer") + lin("rcnn_transform_layer") + lin("rcnn_adapt_k_layer")
               + lin("rcnn_unary_layer")))
    rpn_att = lin("rpn_adapt_q_layer") + lin("rpn_adapt_k_layer") + lin("rpn_unary_layer")
    if model.semantic_enhance:
        rpn_att += lin("rpn_channel_k_layer")
    st.append(("rpn", lin("RCNN_rpn.RPN_cls_score") + lin("RCNN_rpn.RPN_bbox_pred") + lin("RCNN_rpn.RPN_Conv") + rpn_att))
    for li in (2, 1):  # RCNN_base.6 (layer3) then RCNN_base.5 (layer2); layer1 is frozen
        layer = plan["layers"][li]
        for bi in reversed(range(len(layer))):
            key = "RCNN_base.%d.%d" % (4 + li, bi)
            st.append((key, _block_convs(key, layer[bi])))
    return st


def _ready(model, names):
    cb = getattr(model, "_grad_ready_cb", None)
    if cb is not None:
        cb(names)


class _Fresh(__import__("threading").local):
    """gradients _acc() allocated since the last WeightGrads.join() (possibly in a side stream's pool); per THREAD: under
    nn.DataParallel every replica's backward runs in its own thread (train.py:104-105)"""

    def __init__(self):
        self.grads = []


_FRESH = _Fresh()


def _acc(param, g):
    g = g.view_as(param)
    if param.grad is None:
        # (the reference's optimizer.zero_grad() sets grads to None, so the bridge path comes here every iteration.)
        # When this runs on the weight-gradient side stream, the clone's block belongs to THAT stream's pool, while the
        # optimizer / clipping / all-reduce read it on the caller's stream: join() marks it as used there
        param.grad = g.clone()
        _FRESH.grads.append(param.grad)
    else:
        param.grad.add_(g)


def _attention_backward(d_dense, ld_dd, a, unary, q, k_, s_mat, Bn, rows_b, nseg, L, Kp, dq, ugamma, k_batch, s_batch,
                        u_batch, d_k_out, d_s_out, d_u_out, vw=1024):
    """Adjoint of one dual-awareness attention  dense = ((softmax_seg(q k^T / sqrt(dq)) + ugamma u) / nseg) s
    (dana.py:118-154 / 258-283) for Bn images of rows_b query rows each.
      d_dense [Bn*rows_b][vw] (row stride ld_dd; vw = width of the value rows s_mat: 1024, or 64 where the values are the
      re-associated table S . Wt_a^T of the RoI heads); a = the saved attention [Bn][rows_b][Kp]; q [Bn*rows_b][dq];
      k_ / s_mat / unary: key, value and unary rows of image b start at b * k_batch / s_batch / u_batch (floats).
    Accumulates into d_k_out (rows of image b at b*k_batch), d_s_out (b*s_batch), d_u_out (b*u_batch); returns d_q."""
    dev = a.device
    K = nseg * L
    dA = torch.zeros((Bn, rows_b, Kp), dtype=torch.float32, device=dev)
    ops.gemm_nt(d_dense, s_mat, rows_b, K, vw, lda=ld_dd, out=dA, ldc=Kp, batch=Bn, batch_a=rows_b * ld_dd,
                batch_b=s_batch, batch_c=rows_b * Kp)
    # d s[b] += a[b]^T . d_dense[b], every image in one launch (a's zero-padded columns K..Kp-1 are computed, not stored)
    ops.gemm_tn_batched(a, d_dense, Bn, rows_b, Kp, vw, d_s_out, ldy=Kp, ldx=ld_dd, batch_y=rows_b * Kp,
                        batch_x=rows_b * ld_dd, batch_out=s_batch, n_valid=K)
    ops.colsum_batched(dA, Bn, rows_b, K, d_u_out, ld=Kp, x_batch=rows_b * Kp, out_batch=u_batch, alpha=ugamma / nseg)
    ops.attn_softmax_unary_backward_(dA, a, unary, Bn * rows_b, rows_b, nseg, L, Kp, Kp, ugamma, 1.0 / nseg,
                                     1.0 / math.sqrt(dq), unary_batch_stride=u_batch)
    kt = ops.transpose_batched(k_, Bn, K, dq, ldi=dq, ldo=Kp, in_batch=k_batch)  # [Bn][dq][Kp], zero padded
    d_q = ops.gemm_nt(dA, kt, rows_b, dq, Kp, lda=Kp, ldb=Kp, batch=Bn, batch_a=rows_b * Kp, batch_b=dq * Kp)
    # d k[b] += dS0[b]^T . q[b]
    ops.gemm_tn_batched(dA, q, Bn, rows_b, Kp, dq, d_k_out, ldy=Kp, ldx=dq, batch_y=rows_b * Kp, batch_x=rows_b * dq,
                        batch_out=k_batch, n_valid=K)
    return d_q.view(Bn * rows_b, dq)


def _rpn_conv_plan(model, ctx):
    """the RPN 3x3 conv as a plan entry, one per saved forward (its data-gradient weights are derived into it once)"""
    c = ctx.get("_c_rpn")
    if c is None:
        plan = ctx["plan"]
        c = ctx["_c_rpn"] = dict(cin=model.RCNN_rpn.din, cout=512, k=3, stride=1, pad=1, w=plan["rpn_conv_w"], scale=None,
                                 u=plan["rpn_conv_u"])
    return c


def prefetch_dgrad_weights(model, ctx, dev):
    """The backward's weight-only launches (flipped / transposed / BN-scaled data-gradient weights and their Winograd
    transforms: ~70 per iteration, 0.3-0.4 ms of kernels) issued from the SAVING FORWARD behind the RPN head, on the role
    stream `model.prefetch_dgrad` names (layer4: idle until RoIAlign): they run under the proposal layer and the host round
    trip instead of in front of the backward's first contractions (a kernel trace of the replayed iteration showed 0.65 ms
    without a contraction there). The weights do not change between a forward and its backward. Records
    ctx['dgw_prefetched']. Measured -0.2 ms per iteration on `layer4`, +0.5 ms on `wgrad` / `targets` (dana.py)."""
    role = getattr(model, "prefetch_dgrad", None)
    if getattr(model, "_single_stream", False) or not role or torch.cuda.is_current_stream_capturing():
        return
    plan = ctx["plan"]
    prep = model._stream(role, dev)
    ev0 = ops.record_event()
    prep.wait_event(ev0)  # (behind the optimizer's update of the weights on the caller's stream)
    with ops.on_stream(prep):
        _dgrad_weights(_rpn_conv_plan(model, ctx))
        for bp in reversed(plan["layer4"]):
            for name in ("c3", "c2", "c1", "ds"):
                if bp.get(name) is not None:
                    _dgrad_weights(bp[name])
        for layer in reversed(plan["layers"][1:]):  # (layer1 is frozen and in front of every trainable layer: no data gradient)
            for bp in reversed(layer):
                for name in ("c3", "c2", "c1", "ds"):
                    if bp.get(name) is not None:
                        _dgrad_weights(bp[name])
        ctx["dgw_prefetched"] = ops.record_event()


def _rpn_chain(model, ctx, g1, g2, g_dev, grads_r, rpnw_ready=None):
    """Adjoint of the RPN branch: RPN losses -> heads -> 3x3 conv -> RPN-level attention (rpn.py:58-115, dana.py:118-154),
    on the CURRENT stream. It reads the forward's saved tensors only; -> (d_corr [B*hw][2048]: the gradient into
    [base_feat | attended], d_s_pe [B][shot*L][1024]: into the positive supports' PE-added maps). Weight gradients of the
    branch are accumulated into .grad (through grads_r) before it returns."""
    plan = ctx["plan"]
    B, shot = ctx["B"], ctx["shot"]
    fh, fw = ctx["fh"], ctx["fw"]
    hw = fh * fw
    L = ctx["s_pe"].size(1) // shot
    d = model.rpn_reduce_dim
    corr = ctx["corr"]
    dev = corr.device
    ug = model.unary_gamma
    c_rpn = _rpn_conv_plan(model, ctx)
    # -- RPN: losses -> heads -> 3x3 conv (rpn.py:58-115) --
    rpn = model.RCNN_rpn
    nh = ctx["nh"]
    d_heads = ops.rpn_loss_backward(ctx["rpn_heads"], nh, ctx["at"], ctx["rpn_l"], g1, g2, sigma=3.0,
                                    inside_weight=cfg.TRAIN.RPN_BBOX_INSIDE_WEIGHTS[0], grad_dev=g_dev)
    ns = rpn.nc_score_out
    grads_r.linear(d_heads, ctx["rpn_x"], B * hw, nh, 512,
                 lambda dw, db: (_acc(rpn.RPN_cls_score.weight, dw[:ns]), _acc(rpn.RPN_cls_score.bias, db[:ns]),
                                 _acc(rpn.RPN_bbox_pred.weight, dw[ns:]), _acc(rpn.RPN_bbox_pred.bias, db[ns:])))
    _, _, d_x = ops.linear_backward(d_heads, ctx["rpn_x"], plan["rpn_head_w"], B * hw, nh, 512, need_dw=False)
    ops.relu_mask_(d_x, ctx["rpn_x"], B * hw, 512)
    if rpnw_ready is not None:
        ops.cur_stream().wait_event(rpnw_ready)
    grads_r.add_conv("RCNN_rpn.RPN_Conv", d_x, corr, B, fh, fw, c_rpn, v=ctx.get("rpn_v"))
    _acc(rpn.RPN_Conv.bias, ops.colsum(d_x, B * hw, 512))
    d_corr = conv_dgrad(d_x, B, fh, fw, c_rpn)  # [B*hw][2048]

    # -- RPN-level attention (dana.py:118-154): corr = [base_feat | dense] --
    K1 = shot * L
    s_pe, kp, qp, unary = ctx["s_pe"], ctx["kp"], ctx["qp"], ctx["unary"]
    d_s_pe = torch.zeros((B, K1, 1024), dtype=torch.float32, device=dev)
    d_kp = torch.zeros((B * K1, d), dtype=torch.float32, device=dev)
    d_un = torch.zeros((B * shot, L), dtype=torch.float32, device=dev)
    d_qp = _attention_backward(d_corr.view(-1)[1024:], 2048, ctx["scores"], unary, qp, kp, s_pe, B, hw, shot, L, K1, d,
                               ug, K1 * d, K1 * 1024, K1, d_kp, d_s_pe, d_un)
    ops.colmean_sub_(d_qp, B, hw, d)
    ops.colmean_sub_(d_kp, B * shot, L, d)
    wq = model.rpn_adapt_q_layer.weight.detach()
    grads_r.linear(d_qp, corr, B * hw, d, 1024,
                 lambda dw, db: (_acc(model.rpn_adapt_q_layer.weight, dw), _acc(model.rpn_adapt_q_layer.bias, db)), ldx=2048)
    ops.linear_backward(d_qp, corr, wq, B * hw, d, 1024, ldx=2048, dx_out=d_corr, dx_ld=2048, need_dw=False)
    wk = model.rpn_adapt_k_layer.weight.detach()
    grads_r.linear(d_kp, s_pe, B * K1, d, 1024,
                 lambda dw, db: (_acc(model.rpn_adapt_k_layer.weight, dw), _acc(model.rpn_adapt_k_layer.bias, db)))
    ops.linear_backward(d_kp, s_pe, wk, B * K1, d, 1024, dx_out=d_s_pe, dx_ld=1024, need_dw=False)
    ops.softmax_rows_backward_(d_un, unary, B * shot, L)
    wu = model.rpn_unary_layer.weight.detach()
    _acc(model.rpn_unary_layer.weight, ops.rowdot_backward(s_pe, d_un, wu, B * K1, 1024, grad_x=d_s_pe))
    _acc(model.rpn_unary_layer.bias, ops.colsum(d_un, B * K1, 1))
    if model.semantic_enhance:  # BA block (dana.py:133-137)
        s_pre, ba_w = ctx["s_pre"], ctx["ba_w"]
        G = B * shot
        gvec, gsum = ops.ba_backward_prep(s_pre, ba_w, d_s_pe, G, L, 1024)  # (one launch; a loop of 4 per group until round 6)
        d_w = ops.ba_backward_(d_s_pe, s_pre, ba_w, gvec, gsum, G, L, 1024, gamma=model.channel_gamma, slope=0.01)
        ops.softmax_rows_backward_(d_w, ba_w, G, L)
        wc = model.rpn_channel_k_layer.weight.detach()
        _acc(model.rpn_channel_k_layer.weight, ops.rowdot_backward(s_pre, d_w, wc, G * L, 1024, grad_x=d_s_pe))
        _acc(model.rpn_channel_k_layer.bias, ops.colsum(d_w, G * L, 1))
    grads_r.finish_all(model, "RCNN_rpn")
    return d_corr, d_s_pe



def _take_ctx(model, ctx):
    """the saved-for-backward context to differentiate: the one handed in (the loss bridge captured it at forward time)
    or the model's latest. A context is consumed exactly once; its tensors are released here."""
    if ctx is None:
        ctx = model._ctx
    if ctx is None or ctx.get("consumed"):
        raise RuntimeError("no saved training forward to differentiate (run a train-mode forward with grad enabled "
                           "or model.save_for_backward = True first; each forward can be differentiated once)")
    return ctx


def _release_ctx(model, ctx):
    keep = {"consumed": True}
    ctx.clear()
    ctx.update(keep)
    if model._ctx is ctx:
        model._ctx = None


def model_backward(model, grad_losses=(1.0, 1.0, 1.0, 1.0), ctx=None):
    """model_backward_gen run to completion (the eager path and single-graph captures)"""
    for _ in model_backward_gen(model, grad_losses, ctx=ctx):
        pass


def model_backward_gen(model, grad_losses=(1.0, 1.0, 1.0, 1.0), ctx=None):
    """(generator; pauses ONCE, where the gradients of everything except the trunk are final and every side stream is
    joined into the caller's stream: graphs.GraphedTrainer ends one hipGraph there and starts the next, so that the
    RCCL all-reduce of the finished buckets overlaps the trunk's backward)

    d(sum_i grad_losses[i] * loss_i)/d(parameters) for the four training losses (rpn_loss_cls, rpn_loss_bbox,
    RCNN_loss_cls, RCNN_loss_bbox) of the last `save_for_backward` forward: what train.py:141-143's
    `loss.backward()` computes, accumulated into `.grad` of the trainable parameters (BN, conv1 and layer1 are
    frozen: dana.py:350-385)."""
    if type(model).__name__ in ("FasterRCNN", "MetaRCNN", "FGN", "FSOD"):
        frcnn_backward(model, grad_losses, ctx=ctx)
        return
    ctx = _take_ctx(model, ctx)
    plan = ctx["plan"]
    B, shot, way, R, Ns = ctx["B"], ctx["shot"], ctx["way"], ctx["R"], ctx["Ns"]
    fh, fw = ctx["fh"], ctx["fw"]
    hw = fh * fw
    P2 = 49
    L = ctx["s_pe"].size(1) // shot  # positions of a support map (400 for the reference's 320x320 supports)
    n_roi = B * R
    d, dq = model.rpn_reduce_dim, model.rcnn_reduce_dim
    g_dev = None
    if isinstance(grad_losses, torch.Tensor):  # upstream gradients stay on the device: no host sync in the backward
        g_dev = grad_losses.detach().to(torch.float32).contiguous()
        g1 = g2 = g3 = g4 = 1.0
    else:
        g1, g2, g3, g4 = [float(x) for x in grad_losses]
    corr = ctx["corr"]
    dev = corr.device
    grads = WeightGrads(None if getattr(model, "_single_stream", False) else model._stream("wgrad", dev), model)
    ug = model.unary_gamma

    # -- the trunk's data-gradient weights (flipped / transposed / BN-scaled copies, Winograd-domain filters: ~45 small
    #    launches that depend on the weights only) are derived at the head of the weight-gradient stream instead of one by
    #    one in front of the trunk's data-gradient launches that need them (the chain every other launch of the trunk's
    #    backward waits for). (Round 4 gave them a stream of their own; which hardware queue that stream landed on decided
    #    1-2 ms of the iteration: profiles/r5_role_streams.md.) --
    rpn = model.RCNN_rpn
    c_rpn = _rpn_conv_plan(model, ctx)
    dgw_ready = l4w_ready = rpnw_ready = None
    prefetch = not getattr(model, "_single_stream", False)
    seen = set()

    def derive(saved):
        for sv in reversed(saved):
            for name in ("c3", "c2", "c1", "ds"):
                c = sv["bp"].get(name)
                if c is not None and id(c) not in seen:
                    seen.add(id(c))
                    _dgrad_weights(c)

    early = ctx.get("dgw_prefetched")
    if early is not None:
        # round 6: the saving forward already issued them on the weight-gradient stream, under its own trunk
        # (prefetch_dgrad_weights): the backward's three chains start at once instead of behind ~70 weight-only launches
        rpnw_ready = l4w_ready = dgw_ready = early
    elif prefetch:
        prep = model._stream("wgrad", dev)  # (at the head of the weight-gradient stream: nothing is queued there yet)
        ev0 = ops.record_event()
        prep.wait_event(ev0)
        with ops.on_stream(prep):
            # (in the order the backward needs them: the RPN chain and the box branch start at once, then the trunk)
            _dgrad_weights(c_rpn)
            rpnw_ready = ops.record_event()
            derive(ctx["l4_saved"])
            l4w_ready = ops.record_event()
            derive(ctx["q_saved"])
            dgw_ready = ops.record_event()

    # -- RPN chain (_rpn_chain). It depends on the forward's saved tensors only and meets the RoI stage's gradients in
    #    base_feat / the support maps, so it runs on a stream of its own FROM THE START of the backward, beside the box
    #    branch and the RoI heads (round 4: it used to follow them on the caller's stream, 1.4 ms of launches with nothing
    #    beside them). Under stream capture its weight gradients stay inline on the chain's stream (a side stream forked
    #    from an already forked stream crashes hipStreamEndCapture on ROCm 7.2). Issuing the chain even earlier -- from the
    #    eager forward, right behind the RPN head, under the proposal layer and the host round trip -- was built and
    #    measured: +-0 (18.10 / 18.03 vs 18.07 ms): the eager iteration is host-bound there, the chain's ~100 launches
    #    delay the host's count read by what they save on the GPU. Letting every side stream enter the capture through
    #    an event of the capturing stream itself (a flat fork structure) does not avoid that crash either (measured). --
    main = ops.cur_stream()
    single = getattr(model, "_single_stream", False)
    capturing = torch.cuda.is_current_stream_capturing()
    rpn_early = not single
    rpn_out = rpn_done = None
    rpn_start = ops.record_event() if rpn_early else None

    def launch_rpn_chain():
        rpn_stream = model._stream("support", dev)  # (the forward's support stream: idle in the backward)
        rpn_stream.wait_event(rpn_start)
        with ops.on_stream(rpn_stream):
            grads_r = WeightGrads(None if capturing else model._stream("wgrad", dev), model)
            out = _rpn_chain(model, ctx, g1, g2, g_dev, grads_r, rpnw_ready)
            for t_ in out:
                t_.record_stream(main)
            return out, ops.record_event()

    if rpn_early:  # (issued FIRST: its two 300 us launches buy the host the time to issue the other chains; round 4: 1.2 ms)
        rpn_out, rpn_done = launch_rpn_chain()

    # -- seeds: d RCNN losses / d (scores, bbox_pred) were written by the fused loss kernel (dana_rcnn_loss);
    #    the upstream scalars g3 / g4 ride as alpha on the first launches that consume them --
    d_score_pos, d_score_neg, d_bbox = ctx["loss_seeds"]
    if g_dev is not None:
        ops.scale_by_device_scalar_(d_score_pos, g_dev[2:])
        ops.scale_by_device_scalar_(d_score_neg, g_dev[2:])
        ops.scale_by_device_scalar_(d_bbox, g_dev[3:])

    # -- box branch: RCNN_bbox_pred <- mean <- layer4 (dana.py:246,387-389). Independent of the attention heads until
    #    the two gradients of the pooled features meet, so it runs on the forward's layer4 stream: the heads' backward
    #    (many small launches) fills the CUs its big launches leave idle in their tails. --
    main = ops.cur_stream()
    # (under stream capture the box branch stays on the caller's stream: a weight-gradient side stream forked from an
    # already forked stream crashes hipStreamEndCapture on ROCm 7.2 -- tools/graph_debug.py modes 8 / 12 / 13)
    l4_stream = main if (getattr(model, "_single_stream", False) or torch.cuda.is_current_stream_capturing()) \
        else model._stream("layer4", dev)
    seeds_ready = ops.record_event()
    stages = grad_stages(model, plan)
    with ops.on_stream(l4_stream):
        l4_stream.wait_event(seeds_ready)
        if l4w_ready is not None:
            l4_stream.wait_event(l4w_ready)
        wb = model.RCNN_bbox_pred.weight.detach()
        _acc(model.RCNN_bbox_pred.weight,
             ops.gemm_small(d_bbox, (1, 4), ctx["fc7"], (2048, 1), 4, 2048, n_roi, alpha=g4))
        _acc(model.RCNN_bbox_pred.bias, ops.colsum(d_bbox, n_roi, 4, alpha=g4))
        d_fc7 = ops.gemm_small(d_bbox, (4, 1), wb, (2048, 1), n_roi, 2048, 4, alpha=g4)
        l4 = ctx["l4_saved"]
        npos = l4[-1]["h1"] * l4[-1]["w1"]
        g = ops.broadcast_rows(d_fc7, n_roi, npos, 2048, alpha=1.0 / npos)
        for i, sv in enumerate(reversed(l4)):  # the first block's input is the RoIAlign output: no ReLU in front of it
            g = bottleneck_backward(g, sv, sv["n"], sv["h"], sv["w"], sv["bp"], grads, sv["key"],
                                    mask_dx=i < len(l4) - 1, g_masked=i > 0)
        d_pooled = g  # [n_roi*49][1024]
        d_pooled.record_stream(main)
        grads.finish_all(model, "RCNN_top")
        _ready(model, stages[0][1])
        box_done = ops.record_event()

    # -- RoI-level attention heads (dana.py:248-292), positive then negative supports --
    q_pe, q2, sp_pe, k2, un2 = ctx["q_pe"], ctx["q2"], ctx["sp_pe"], ctx["k2"], ctx["un2"]
    K2, K2p = ctx["K2"], ctx["K2p"]
    wt = model.rcnn_transform_layer.weight.detach()
    w1 = model.output_score_layer.linear1.weight.detach()
    w2 = model.output_score_layer.linear2.weight.detach()
    rd = model.rcnn_dim
    nhid = w1.size(0)
    d_q2 = torch.zeros((n_roi * P2, dq), dtype=torch.float32, device=dev)
    d_trq = torch.zeros((n_roi * P2, rd), dtype=torch.float32, device=dev)
    d_sp_pe = torch.zeros((Ns * P2, 1024), dtype=torch.float32, device=dev)
    d_k2 = torch.zeros((Ns * P2, dq), dtype=torch.float32, device=dev)
    d_un2 = torch.zeros((Ns, P2), dtype=torch.float32, device=dev)
    d_wt = torch.zeros_like(wt)
    d_sw = None
    for hc in ctx["heads"]:
        off = hc["offset"]
        hi = 0 if off == 0 else 1  # rows of cls_score_all: positive-support scores first (dana.py:194)
        ds = d_score_pos if hi == 0 else d_score_neg
        _acc(model.output_score_layer.linear2.weight,
             ops.gemm_small(ds, (1, 2), hc["hid"], (nhid, 1), 2, nhid, n_roi, alpha=g3))
        _acc(model.output_score_layer.linear2.bias, ops.colsum(ds, n_roi, 2, alpha=g3))
        d_hid = ops.gemm_small(ds, (2, 1), w2, (nhid, 1), n_roi, nhid, 2, alpha=g3)
        ops.relu_mask_(d_hid, hc["hid"], n_roi, nhid)
        lin1 = model.output_score_layer.linear1
        grads.linear(d_hid, hc["tr"], n_roi, nhid, P2 * rd, lambda dw, db: (_acc(lin1.weight, dw), _acc(lin1.bias, db)))
        _, _, d_tr = ops.linear_backward(d_hid, hc["tr"], w1, n_roi, nhid, P2 * rd, need_dw=False)
        ops.axpy_rows_(d_trq, d_tr, n_roi * P2, rd)
        if hc["dense"] is None:
            # the forward ran  tr = A . (S . Wt_a^T) + q half  (DAnARCNN.fold_roi_attn): the attention's VALUE rows are the
            # [147][64] table sw of each image, so its adjoint works on 64-wide rows -- the [n*49][1024] gradient of the
            # attended tensor, its two GEMMs against Wt_a and the two K = 1024 attention adjoints per head do not exist
            if d_sw is None:
                d_sw = torch.zeros((Ns * P2, rd), dtype=torch.float32, device=dev)
            d_qh = _attention_backward(d_tr, rd, hc["sc2"], un2.view(-1)[off * P2:], q2, k2.view(-1)[off * P2 * dq:],
                                       ctx["sw"].view(-1)[off * P2 * rd:], B, R * P2, shot, P2, K2p, dq, ug,
                                       way * shot * P2 * dq, way * shot * P2 * rd, way * shot * P2,
                                       d_k2.view(-1)[off * P2 * dq:], d_sw.view(-1)[off * P2 * rd:],
                                       d_un2.view(-1)[off * P2:], vw=rd)
        else:
            grads.linear(d_tr, hc["dense"], n_roi * P2, rd, 1024,
                         lambda dw, db: ops.axpy_rows_(d_wt.view(-1)[1024:], dw, rd, 1024, ld_y=2048))
            _, _, d_dense = ops.linear_backward(d_tr, hc["dense"], wt.view(-1)[1024:], n_roi * P2, rd, 1024, ldw=2048,
                                                need_dw=False)
            d_qh = _attention_backward(d_dense, 1024, hc["sc2"], un2.view(-1)[off * P2:], q2, k2.view(-1)[off * P2 * dq:],
                                       sp_pe.view(-1)[off * P2 * 1024:], B, R * P2, shot, P2, K2p, dq, ug,
                                       way * shot * P2 * dq, way * shot * P2 * 1024, way * shot * P2,
                                       d_k2.view(-1)[off * P2 * dq:], d_sp_pe.view(-1)[off * P2 * 1024:],
                                       d_un2.view(-1)[off * P2:])
        ops.axpy_rows_(d_q2, d_qh, n_roi * P2, dq)
    if d_sw is not None:
        # sw = sp_pe . Wt_a^T (once per support, both heads): d Wt_a = d_sw^T . sp_pe, d sp_pe += d_sw . Wt_a
        grads.linear(d_sw, sp_pe, Ns * P2, rd, 1024,
                     lambda dw, db: ops.axpy_rows_(d_wt.view(-1)[1024:], dw, rd, 1024, ld_y=2048))
        ops.linear_backward(d_sw, sp_pe, wt.view(-1)[1024:], Ns * P2, rd, 1024, ldw=2048, dx_out=d_sp_pe, dx_ld=1024,
                            need_dw=False)

    # -- RoI-level query side: Q projection + the q half of rcnn_transform_layer; PE is additive --
    ops.colmean_sub_(d_q2, n_roi, P2, dq)
    wq2 = model.rcnn_adapt_q_layer.weight.detach()
    grads.linear(d_q2, q_pe, n_roi * P2, dq, 1024,
                 lambda dw, db: (_acc(model.rcnn_adapt_q_layer.weight, dw), _acc(model.rcnn_adapt_q_layer.bias, db)))
    _, _, d_q_pe = ops.linear_backward(d_q2, q_pe, wq2, n_roi * P2, dq, 1024, need_dw=False)

    def _transform_grads(dw, db):  # (both halves of rcnn_transform_layer's weight gradient are in d_wt now)
        ops.axpy_rows_(d_wt, dw, rd, 1024, ld_y=2048)
        _acc(model.rcnn_transform_layer.weight, d_wt)
        _acc(model.rcnn_transform_layer.bias, db)

    grads.linear(d_trq, q_pe, n_roi * P2, rd, 1024, _transform_grads)
    ops.linear_backward(d_trq, q_pe, wt, n_roi * P2, rd, 1024, ldw=2048, dx_out=d_q_pe, dx_ld=1024, need_dw=False)
    main.wait_event(box_done)
    ops.axpy_rows_(d_pooled, d_q_pe, n_roi * P2, 1024)
    if ctx.get("roi_argmax") is not None:
        # cfg.POOLING_MODE == 'pool' (dana.py:183-184): every bin's gradient goes to its argmax element (ROIPool_cuda.cu:79-108)
        g_nchw = ops.roi_pool_backward(ops.nhwc_to_nchw(d_pooled, n_roi, 1024, 7, 7), None, ctx["rois"].view(-1, 5),
                                       ctx["roi_argmax"], 1.0 / 16.0, 7, 7, B, 1024, fh, fw)
        d_bf = ops.nchw_to_nhwc(g_nchw).view(B * fh * fw, 1024)
    else:
        d_bf = ops.roi_align_backward(d_pooled.view(n_roi, 7, 7, 1024), ctx["rois"].view(-1, 5), 1.0 / 16.0, 7, 7, B, 1024,
                                      fh, fw, 0, layout=ops.NHWC)  # [B][fh][fw][1024]

    # -- RoI-level support side: K projection, unary term, PE, 14x14 average pool (dana.py:105-108,271-277) --
    ops.colmean_sub_(d_k2, Ns, P2, dq)
    wk2 = model.rcnn_adapt_k_layer.weight.detach()
    grads.linear(d_k2, sp_pe, Ns * P2, dq, 1024,
                 lambda dw, db: (_acc(model.rcnn_adapt_k_layer.weight, dw), _acc(model.rcnn_adapt_k_layer.bias, db)))
    ops.linear_backward(d_k2, sp_pe, wk2, Ns * P2, dq, 1024, dx_out=d_sp_pe, dx_ld=1024, need_dw=False)
    ops.softmax_rows_backward_(d_un2, un2, Ns, P2)
    wu2 = model.rcnn_unary_layer.weight.detach()
    _acc(model.rcnn_unary_layer.weight, ops.rowdot_backward(sp_pe, d_un2, wu2, Ns * P2, 1024, grad_x=d_sp_pe))
    _acc(model.rcnn_unary_layer.bias, ops.colsum(d_un2, Ns * P2, 1))
    (sh_, sw_), pool = ctx["sup_map"], ctx["sup_pool"]
    d_sup = ops.avgpool_backward(d_sp_pe, Ns, sh_, sw_, 1024, pool[0], pool[1])  # [Ns][L][1024]
    grads.join()  # (the heads' Linear weight / bias gradients were accumulated on the weight-gradient stream)
    _ready(model, stages[1][1])

    if rpn_early:
        main.wait_event(rpn_done)
        for n_ in stages[2][1]:  # (gradients first allocated on the chain's streams are read on the caller's from here on)
            g_ = model.get_parameter(n_).grad
            if g_ is not None:
                g_.record_stream(main)
    else:
        rpn_out = _rpn_chain(model, ctx, g1, g2, g_dev, grads, rpnw_ready)
    d_corr, d_s_pe = rpn_out
    K1 = shot * L
    for b in range(B):  # the positive supports' PE-added maps (dana.py:103,130)
        ops.axpy_rows_(d_sup.view(-1)[b * way * shot * L * 1024:], d_s_pe[b], K1, 1024)
    grads.finish_all(model, "RCNN_rpn")
    _ready(model, stages[2][1])
    if dgw_ready is not None:
        ops.cur_stream().wait_event(dgw_ready)
    yield "heads, RPN and attention done; trunk next"

    # -- trunk: query (RoIAlign + RPN paths meet in base_feat) and supports; layer3, layer2 (layer1 is frozen) --
    ops.axpy_rows_(d_corr, d_bf, B * hw, 1024, ld_y=2048)
    g = torch.empty((B * hw, 1024), dtype=torch.float32, device=dev)
    ops.axpy_rows_(g, d_corr, B * hw, 1024, ld_x=2048, accumulate=False)
    #    block by block for both batches, so that each block's weight gradient is final (and may be all-reduced)
    #    while the earlier blocks are still being differentiated
    gq, gs = g, d_sup.view(Ns * L, 1024)
    qs, ss = ctx["q_saved"], ctx["s_saved"]
    ms = ctx.get("m_saved") or []
    merged_ok = len(ms) == len(qs) and getattr(model, "merge_backward", True)
    nblk = len(qs)
    gm = None  # dL/d(block output) of both batches in one buffer (while the blocks run merged)
    for i in range(nblk - 1, -1, -1):
        sq, s_ = qs[i], ss[i]
        bp = sq["bp"]
        if merged_ok and bp["ds"] is None and bp["c1"]["stride"] == 1 and i > 0:
            sm = ms[i]
            if gm is None:  # enter the merged form: the two gradients into the two row ranges of one buffer
                cout = bp["c3"]["cout"]
                if i == nblk - 1:  # (the last block's outputs live in corr / sup: mask per batch, then join)
                    ops.relu_mask_(gq, sq["o3"], sm["mq_out"], cout, ld_act=sq.get("o3_ld", 0))
                    ops.relu_mask_(gs, s_["o3"], sm["m_out"] - sm["mq_out"], cout, ld_act=s_.get("o3_ld", 0))
                gm = torch.empty((sm["m_out"], cout), dtype=torch.float32, device=dev)
                ops.axpy_rows_(gm, gq, sm["mq_out"], cout, accumulate=False)
                ops.axpy_rows_(gm[sm["mq_out"]:], gs, sm["m_out"] - sm["mq_out"], cout, accumulate=False)
            gm = bottleneck_backward_merged(gm, sq, s_, sm, bp, grads, sq["key"])
            gq, gs = gm[:sm["mq_in"]], gm[sm["mq_in"]:]
        else:
            gm = None
            gq = bottleneck_backward(gq, sq, sq["n"], sq["h"], sq["w"], sq["bp"], grads, sq["key"], need_dx=i > 0,
                                     g_masked=i < nblk - 1)
            gs = bottleneck_backward(gs, s_, s_["n"], s_["h"], s_["w"], s_["bp"], grads, s_["key"], need_dx=i > 0,
                                     g_masked=i < nblk - 1)
        grads.finish_all(model, sq["key"] + ".")
        _ready(model, _block_convs(sq["key"], sq["bp"]))
    assert not grads.packed
    _release_ctx(model, ctx)


# ---- sibling model `frcnn` (lib/model/framework/faster_rcnn.py): the same adjoints without the attention ----------------
def frcnn_grad_stages(model, plan=None):
    plan = plan if plan is not None else model._get_plan()
    lin = lambda n: [n + ".weight", n + ".bias"]  # noqa: E731
    cls = "RCNN_cls_score.0" if type(model).__name__ == "MetaRCNN" else "RCNN_cls_score"  # meta.py:199-201: a Sequential
    extra = []
    if type(model).__name__ == "FSOD":  # fsod.py:29-75: the three relation heads replace RCNN_cls_score
        st = [("roi head", lin("RCNN_bbox_pred") + lin("global_fc_1") + lin("global_fc_2") + lin("global_cls_score")
               + ["corr_conv.weight"] + lin("corr_cls_score") + ["patch_conv_1.weight", "patch_conv_2.weight",
                                                                 "patch_conv_3.weight"] + lin("patch_cls_score")
               + [n for bi in (2, 1, 0) for n in _block_convs("RCNN_top.0.%d" % bi, plan["layer4"][bi])])]
        st.append(("rpn", lin("RCNN_rpn.RPN_cls_score") + lin("RCNN_rpn.RPN_bbox_pred") + lin("RCNN_rpn.RPN_Conv")))
        for li in (2, 1):
            layer = plan["layers"][li]
            for bi in reversed(range(len(layer))):
                key = "RCNN_base.%d.%d" % (4 + li, bi)
                st.append((key, _block_convs(key, layer[bi])))
        return st
    if type(model).__name__ == "FGN":  # fgn.py:29-41: the relation head's two convs and their (trainable) BatchNorms
        extra = ["cls_conv2.weight", "cls_conv1.weight"] + lin("bn2") + lin("bn1")
    st = [("roi head", lin("RCNN_bbox_pred") + lin(cls) + extra
           + [n for bi in (2, 1, 0) for n in _block_convs("RCNN_top.0.%d" % bi, plan["layer4"][bi])])]
    st.append(("rpn", lin("RCNN_rpn.RPN_cls_score") + lin("RCNN_rpn.RPN_bbox_pred") + lin("RCNN_rpn.RPN_Conv")))
    for li in (2, 1):
        layer = plan["layers"][li]
        for bi in reversed(range(len(layer))):
            key = "RCNN_base.%d.%d" % (4 + li, bi)
            st.append((key, _block_convs(key, layer[bi])))
    return st


def frcnn_backward(model, grad_losses=(1.0, 1.0, 1.0, 1.0), ctx=None):
    """d(sum_i grad_losses[i] * loss_i)/d(parameters) of the last training forward of FasterRCNN (faster_rcnn.py:31-105)
    or MetaRCNN (meta.py:39-142): RoI head <- mean <- layer4 <- RoIAlign, RPN losses <- heads <- 3x3 conv, both into
    base_feat, then layer3 / layer2 of the trunk (conv1, layer1 and every BN are frozen: faster_rcnn.py:129-160).
    meta adds the class-attentive vectors: score = Linear(fc7 * mean_shots(sigmoid(mean(layer4(maxpool2(trunk(support)))))))
    for the positive and the negative supports, so its support batch is differentiated through layer4 and the trunk too."""
    ctx = _take_ctx(model, ctx)
    meta = type(model).__name__ == "MetaRCNN"
    fgn = type(model).__name__ == "FGN"
    fsod = type(model).__name__ == "FSOD"
    plan, B, R, fh, fw = ctx["plan"], ctx["B"], ctx["R"], ctx["fh"], ctx["fw"]
    n_roi, hw = B * R, fh * fw
    g_dev = None
    if isinstance(grad_losses, torch.Tensor):  # upstream gradients stay on the device: no host sync in the backward
        g_dev = grad_losses.detach().to(torch.float32).contiguous()
        g1 = g2 = g3 = g4 = 1.0
        for seed, k in zip(ctx["loss_seeds"], (2, 2, 3) if (meta or fgn or fsod) else (2, 3)):  # (cls seeds..., bbox seed) x (g3, g4)
            ops.scale_by_device_scalar_(seed, g_dev[k:])
    else:
        g1, g2, g3, g4 = [float(x) for x in grad_losses]
    fc7 = ctx["fc7"]
    dev = fc7.device
    grads = WeightGrads(None if getattr(model, "_single_stream", False) else model._stream("wgrad", dev), model)
    stages = frcnn_grad_stages(model, plan)
    gs = None
    if meta:
        d_pos, d_neg, d_bbox = ctx["loss_seeds"]  # written by the fused mined-loss kernel (dana_rcnn_loss)
        lin_c = model.RCNN_cls_score[0]
        wc = lin_c.weight.detach()
        d_fc7 = ops.gemm_small(d_bbox, (4, 1), model.RCNN_bbox_pred.weight.detach(), (2048, 1), n_roi, 2048, 4, alpha=g4)
        Ns, shot, way = ctx["Ns"], ctx["shot"], ctx["way"]
        att = ctx["att"]
        d_att = torch.zeros((B, way * shot, 2048), dtype=torch.float32, device=dev)
        for hc in ctx["heads"]:
            ds = d_pos if hc["offset"] == 0 else d_neg
            _acc(lin_c.weight, ops.gemm_small(ds, (1, 2), hc["comb"], (2048, 1), 2, 2048, n_roi, alpha=g3))
            _acc(lin_c.bias, ops.colsum(ds, n_roi, 2, alpha=g3))
            d_comb = ops.gemm_small(ds, (2, 1), wc, (2048, 1), n_roi, 2048, 2, alpha=g3)
            d_fc7.add_(ops.scale_rows_by_group(d_comb, hc["vec"], n_roi, R, 2048))
            # the shots' mean of the attentive vectors: d vec[b] = sum over the image's rois of d_comb * fc7
            d_vec = (d_comb * fc7).view(B, R, 2048).sum(1) / shot
            d_att[:, hc["offset"]:hc["offset"] + shot] += d_vec.unsqueeze(1)
        d_pre = (d_att.view(Ns, 2048) * att * (1.0 - att)).contiguous()  # sigmoid adjoint (meta.py:250)
        sl4 = ctx["sl4_saved"]
        npos_s = sl4[-1]["h1"] * sl4[-1]["w1"]
        g = ops.broadcast_rows(d_pre, Ns, npos_s, 2048, alpha=1.0 / npos_s)
        for i, sv in enumerate(reversed(sl4)):  # the first block's input is the max-pooled map: its ReLU adjoint is the trunk's
            g = bottleneck_backward(g, sv, sv["n"], sv["h"], sv["w"], sv["bp"], grads, sv["key"], mask_dx=i < len(sl4) - 1,
                                    g_masked=i > 0)
        # 2x2 / 2 max pool (meta.py:247) back onto the support maps: the window's (first) maximum takes the gradient
        (sh_, sw_), (mh, mw) = ctx["sup_hw"], ctx["mp_hw"]
        gs = ops.maxpool2x2s2_backward(ctx["sup"].view(Ns * sh_ * sw_, 1024), g.contiguous().view(Ns * mh * mw, 1024), Ns, sh_,
                                       sw_, 1024)
    elif fsod:
        # -- multi-relation head (fsod.py:181-249): score = (global + local-correlation + patch) / 10 for the positive
        #    and the negative support; every [roi | support] concatenation is a split layer (roi half + support half) --
        d_pos, d_neg, d_bbox = ctx["loss_seeds"]
        Ns, shot, way, L = ctx["Ns"], ctx["shot"], ctx["way"], ctx["L"]
        P2, d, dq_ = 49, 1024, 256
        pooled, g_roi, corr_roi = ctx["pooled"], ctx["g_roi"], ctx["corr_roi"]
        d_fc7 = ops.gemm_small(d_bbox, (4, 1), model.RCNN_bbox_pred.weight.detach(), (2048, 1), n_roi, 2048, 4, alpha=g4)
        w1 = model.global_fc_1.weight.detach()
        w2 = model.global_fc_2.weight.detach()
        wg = model.global_cls_score.weight.detach()
        wcc = model.corr_conv.weight.detach().view(d, d).contiguous()
        wcs = model.corr_cls_score.weight.detach()
        wp1 = model.patch_conv_1.weight.detach().view(dq_, 2 * d).contiguous()
        wp3 = model.patch_conv_3.weight.detach().view(d, dq_).contiguous()
        wps = model.patch_cls_score.weight.detach()
        c_p2 = dict(cin=dq_, cout=dq_, k=3, stride=1, pad=0, w=ctx["wp2"], scale=None, u=None)
        d_pooled_head = torch.zeros((n_roi * P2, d), dtype=torch.float32, device=dev)
        d_g_roi = torch.zeros((n_roi, d), dtype=torch.float32, device=dev)
        d_corr_roi = torch.zeros((n_roi * P2, d), dtype=torch.float32, device=dev)
        gs = torch.zeros((Ns * L, 1024), dtype=torch.float32, device=dev)  # d(support trunk output)
        d_w1 = torch.zeros((d, 2 * d), dtype=torch.float32, device=dev)
        d_wp1 = torch.zeros((dq_, 2 * d), dtype=torch.float32, device=dev)
        d_wcc = torch.zeros((d, d), dtype=torch.float32, device=dev)
        d_pos_kernel = None  # d(pooled positive support) from the attention RPN, added below

        def to_supports(d_map, offset):  # mean over the shots (fsod.py:98-101): every shot gets d_map / shot
            for b_ in range(B):
                for s_ in range(shot):
                    ops.axpy_rows_(gs.view(-1)[(b_ * way * shot + offset + s_) * L * 1024:], d_map[b_], L, 1024,
                                   alpha=1.0 / shot)

        d_supports = {}
        for hc in ctx["heads"]:
            ds = (d_pos if hc["offset"] == 0 else d_neg)
            a3 = g3 / 10.0  # fsod.py:237: the three scores are summed and divided by 10
            support = hc["support"]
            d_support = torch.zeros((B * P2, d), dtype=torch.float32, device=dev)
            # .. global relation: Linear(2) <- relu fc2 <- relu fc1([mean(roi) | mean(support)])
            _acc(model.global_cls_score.weight, ops.gemm_small(ds, (1, 2), hc["h2"], (d, 1), 2, d, n_roi, alpha=a3))
            _acc(model.global_cls_score.bias, ops.colsum(ds, n_roi, 2, alpha=a3))
            d_h2 = ops.gemm_small(ds, (2, 1), wg, (d, 1), n_roi, d, 2, alpha=a3)
            ops.relu_mask_(d_h2, hc["h2"], n_roi, d)
            dw2, db2, d_h1 = ops.linear_backward(d_h2, hc["h1"], w2, n_roi, d, d)
            _acc(model.global_fc_2.weight, dw2)
            _acc(model.global_fc_2.bias, db2)
            ops.relu_mask_(d_h1, hc["h1"], n_roi, d)
            dw1r, db1, _ = ops.linear_backward(d_h1, g_roi, w1, n_roi, d, d, ldw=2 * d, dx_out=d_g_roi, dx_ld=d)
            ops.axpy_rows_(d_w1, dw1r, d, d, ld_y=2 * d)
            _acc(model.global_fc_1.bias, db1)
            d_gs = ops.spatial_mean(d_h1, B, R, d)  # the support half was broadcast over the image's R rois
            d_gs.mul_(float(R))
            dw1s = ops.gemm_small(d_gs, (1, d), hc["m_sup"], (d, 1), d, d, B)            # [d][d] = d_gs^T . mean(support)
            ops.axpy_rows_(d_w1.view(-1)[d:], dw1s, d, d, ld_y=2 * d)
            d_m_sup = ops.gemm_small(d_gs, (d, 1), w1.view(-1)[d:], (2 * d, 1), B, d, d)  # [B][d] = d_gs . w1[:, d:]
            ops.broadcast_rows(d_m_sup, B, P2, d, alpha=1.0 / P2, out=d_support)
            # .. local correlation: Linear(2) <- sum over the 49 positions of corr_conv(roi) * corr_conv(support)
            _acc(model.corr_cls_score.weight, ops.gemm_small(ds, (1, 2), hc["oc"], (d, 1), 2, d, n_roi, alpha=a3))
            _acc(model.corr_cls_score.bias, ops.colsum(ds, n_roi, 2, alpha=a3))
            d_oc = ops.gemm_small(ds, (2, 1), wcs, (d, 1), n_roi, d, 2, alpha=a3)        # [n][1024] = a 1x1 output map
            g_feat, g_kern = ops.depthwise_corr_backward(d_oc, corr_roi, hc["corr_sup"], n_roi, 7, 7, d, 7, 7,
                                                         maps_per_kernel=R)
            ops.axpy_rows_(d_corr_roi, g_feat, n_roi * P2, d)
            dwc_s, _, _ = ops.linear_backward(g_kern.view(B * P2, d), support.view(B * P2, d), wcc, B * P2, d, d,
                                              dx_out=d_support, dx_ld=d)
            ops.axpy_rows_(d_wcc, dwc_s, d, d)
            # .. patch relation: Linear(2) <- avgpool3 <- relu 1x1 <- relu 3x3 <- avgpool 3/1 <- relu 1x1([roi | support])
            _acc(model.patch_cls_score.weight, ops.gemm_small(ds, (1, 2), hc["x4"], (d, 1), 2, d, n_roi, alpha=a3))
            _acc(model.patch_cls_score.bias, ops.colsum(ds, n_roi, 2, alpha=a3))
            d_x4 = ops.gemm_small(ds, (2, 1), wps, (d, 1), n_roi, d, 2, alpha=a3)
            d_x3 = ops.avgpool_backward(d_x4, n_roi, 3, 3, d, 3, 1).view(n_roi * 9, d)
            ops.relu_mask_(d_x3, hc["x3"], n_roi * 9, d)
            dwp3, _, d_x2 = ops.linear_backward(d_x3, hc["x2"], wp3, n_roi * 9, d, dq_)
            _acc(model.patch_conv_3.weight, dwp3.view(d, dq_, 1, 1))
            ops.relu_mask_(d_x2, hc["x2"], n_roi * 9, dq_)
            grads.add_conv("patch_conv_2", d_x2, hc["x1"].view(n_roi * 25, dq_), n_roi, 5, 5, c_p2)
            d_x1 = conv_dgrad(d_x2, n_roi, 5, 5, c_p2)
            d_x0 = ops.avgpool_backward(d_x1, n_roi, 7, 7, dq_, 3, 1).view(n_roi * P2, dq_)
            ops.relu_mask_(d_x0, hc["x0"], n_roi * P2, dq_)
            dwp1r, _, _ = ops.linear_backward(d_x0, pooled.view(n_roi * P2, d), wp1, n_roi * P2, dq_, d, ldw=2 * d,
                                              dx_out=d_pooled_head, dx_ld=d)
            ops.axpy_rows_(d_wp1, dwp1r, dq_, d, ld_y=2 * d)
            d_p_sup = ops.spatial_mean(d_x0, B, R, P2 * dq_)  # the support half was broadcast over the image's rois
            d_p_sup.mul_(float(R))
            dwp1s, _, _ = ops.linear_backward(d_p_sup.view(B * P2, dq_), support.view(B * P2, d), wp1.view(-1)[d:],
                                              B * P2, dq_, d, ldw=2 * d, dx_out=d_support, dx_ld=d)
            ops.axpy_rows_(d_wp1.view(-1)[d:], dwp1s, dq_, d, ld_y=2 * d)
            d_supports[hc["offset"]] = d_support
        # the roi halves shared by both heads: mean over the 49 positions, corr_conv(rois)
        ops.broadcast_rows(d_g_roi, n_roi, P2, d, alpha=1.0 / P2, out=d_pooled_head)
        dwc_r, _, _ = ops.linear_backward(d_corr_roi, pooled.view(n_roi * P2, d), wcc, n_roi * P2, d, d,
                                          dx_out=d_pooled_head, dx_ld=d)
        ops.axpy_rows_(d_wcc, dwc_r, d, d)
        _acc(model.global_fc_1.weight, d_w1)
        _acc(model.patch_conv_1.weight, d_wp1.view(dq_, 2 * d, 1, 1))
        _acc(model.corr_conv.weight, d_wcc.view(d, d, 1, 1))
    elif fgn:
        # -- relation head (fgn.py:145-165): Linear <- ReLU/BN2 <- conv2 <- ReLU/BN1 <- (support half + roi half) of conv1;
        #    bn1 / bn2 are ORDINARY BatchNorms in train mode: their adjoint goes through the batch statistics --
        d_pos, d_neg, d_bbox = ctx["loss_seeds"]
        Ns, shot, way, L = ctx["Ns"], ctx["shot"], ctx["way"], ctx["L"]
        lin_c, wl = model.RCNN_cls_score, ctx["wl"]
        d_fc7 = ops.gemm_small(d_bbox, (4, 1), model.RCNN_bbox_pred.weight.detach(), (2048, 1), n_roi, 2048, 4, alpha=g4)
        for bn_ in (model.bn1, model.bn2):
            for p_ in (bn_.weight, bn_.bias):
                if p_.grad is None:
                    p_.grad = torch.zeros_like(p_)
        c2 = dict(cin=512, cout=128, k=3, stride=1, pad=0, w=ctx["w2"], scale=None, u=None)
        d_roi_half = torch.zeros((n_roi * 25, 512), dtype=torch.float32, device=dev)
        gs = torch.zeros((Ns * L, 1024), dtype=torch.float32, device=dev)  # d(support trunk output)
        w1g = model.cls_conv1.weight

        def acc_w1_half(packed, lo):  # packed [512][3*3*1024] -> cls_conv1.weight.grad[:, lo:lo+1024] (OIHW)
            tmp = torch.empty((512, 1024, 3, 3), dtype=torch.float32, device=dev)
            ops.unpack_conv_weight_grad(packed, tmp, 512, 1024, 3, 3, accumulate=False)
            if w1g.grad is None:
                w1g.grad = torch.zeros_like(w1g)
            w1g.grad[:, lo:lo + 1024].add_(tmp)

        def to_supports(d_map, offset):  # mean over the shots (fgn.py:57-60): every shot gets d_map / shot
            for b_ in range(B):
                for s_ in range(shot):
                    ops.axpy_rows_(gs.view(-1)[(b_ * way * shot + offset + s_) * L * 1024:], d_map[b_], L, 1024,
                                   alpha=1.0 / shot)

        for hc in ctx["heads"]:
            ds = d_pos if hc["offset"] == 0 else d_neg
            dwl = ops.gemm_small(ds, (1, 2), hc["x2"], (1152, 1), 2, 1152, n_roi, alpha=g3)   # [2][(h,w,c)]
            _acc(lin_c.weight, dwl.view(2, 9, 128).permute(0, 2, 1).reshape(2, 1152))            # -> the NCHW flatten (c,h,w)
            _acc(lin_c.bias, ops.colsum(ds, n_roi, 2, alpha=g3))
            d_x2 = ops.gemm_small(ds, (2, 1), wl, (1152, 1), n_roi, 1152, 2, alpha=g3).view(n_roi * 9, 128)
            ops.relu_mask_(d_x2, hc["x2"], n_roi * 9, 128)
            x2_pre, m2, v2 = hc["bn2"]
            d_x2pre = ops.bn_train_backward(d_x2, x2_pre, m2, v2, model.bn2.weight, model.bn2.eps, n_roi * 9, 128,
                                            model.bn2.weight.grad, model.bn2.bias.grad)
            grads.add_conv("cls_conv2", d_x2pre, hc["x1"], n_roi, 5, 5, c2)
            d_x1 = conv_dgrad(d_x2pre, n_roi, 5, 5, c2, mask=hc["x1"])  # (+ the ReLU adjoint of bn1's output)
            x1_pre, m1, v1 = hc["bn1"]
            d_x1pre = ops.bn_train_backward(d_x1, x1_pre, m1, v1, model.bn1.weight, model.bn1.eps, n_roi * 25, 512,
                                            model.bn1.weight.grad, model.bn1.bias.grad)
            ops.axpy_rows_(d_roi_half, d_x1pre, n_roi * 25, 512)
            d_s_half = ops.spatial_mean(d_x1pre, B, R, 25 * 512)  # broadcast over the image's R rois: sum = R * mean
            d_s_half.mul_(float(R))
            acc_w1_half(ops.conv2d_wgrad(d_s_half, hc["support"], B, 7, 7, 1024, 512, 3, 3, 1, 0), 0)
            d_support = ops.conv2d_dgrad(d_s_half, ctx["w1_sup"], B, 7, 7, 1024, 512, 3, 3, 1, 0)   # [B*49][1024]
            to_supports(ops.avgpool_backward(d_support, B, 20, 20, 1024, 14, 1), hc["offset"])          # AvgPool2d(14, 1)
        acc_w1_half(ops.conv2d_wgrad(d_roi_half, ctx["pooled"], n_roi, 7, 7, 1024, 512, 3, 3, 1, 0), 1024)
        d_pooled_head = ops.conv2d_dgrad(d_roi_half, ctx["w1_roi"], n_roi, 7, 7, 1024, 512, 3, 3, 1, 0)  # [n*49][1024]
    else:
        d_cls, d_bbox = ctx["loss_seeds"]  # d(loss_cls + loss_bbox) / d(cls_score, bbox_pred)
        C = d_cls.size(1)
        _acc(model.RCNN_cls_score.weight, ops.gemm_small(d_cls, (1, C), fc7, (2048, 1), C, 2048, n_roi, alpha=g3))
        _acc(model.RCNN_cls_score.bias, ops.colsum(d_cls, n_roi, C, alpha=g3))
        d_fc7 = ops.gemm_small(d_bbox, (4, 1), model.RCNN_bbox_pred.weight.detach(), (2048, 1), n_roi, 2048, 4, alpha=g4)
        d_fc7.add_(ops.gemm_small(d_cls, (C, 1), model.RCNN_cls_score.weight.detach(), (2048, 1), n_roi, 2048, C, alpha=g3))
    _acc(model.RCNN_bbox_pred.weight, ops.gemm_small(d_bbox, (1, 4), fc7, (2048, 1), 4, 2048, n_roi, alpha=g4))
    _acc(model.RCNN_bbox_pred.bias, ops.colsum(d_bbox, n_roi, 4, alpha=g4))
    l4 = ctx["l4_saved"]
    npos = l4[-1]["h1"] * l4[-1]["w1"]
    g = ops.broadcast_rows(d_fc7, n_roi, npos, 2048, alpha=1.0 / npos)
    for i, sv in enumerate(reversed(l4)):  # the first block's input is the RoIAlign output: no ReLU in front of it
        g = bottleneck_backward(g, sv, sv["n"], sv["h"], sv["w"], sv["bp"], grads, sv["key"], mask_dx=i < len(l4) - 1,
                                g_masked=i > 0)
    grads.finish_all(model, "RCNN_top")
    if fsod:
        grads.finish_all(model, "patch_conv_2")
        ops.axpy_rows_(g, d_pooled_head, n_roi * 49, 1024)  # the pooled features also feed the relation heads' roi halves
    if fgn:
        grads.finish_all(model, "cls_conv2")
        ops.axpy_rows_(g, d_pooled_head, n_roi * 49, 1024)  # the pooled features also feed the relation head's roi half
    _ready(model, stages[0][1])
    d_bf = ops.roi_align_backward(g.view(n_roi, 7, 7, 1024), ctx["rois"].view(-1, 5), 1.0 / 16.0, 7, 7, B, 1024, fh, fw,
                                  0, layout=ops.NHWC).view(B * hw, 1024)
    # -- RPN (rpn.py:58-115) on base_feat --
    rpn = model.RCNN_rpn
    nh = ctx["nh"]
    d_heads = ops.rpn_loss_backward(ctx["rpn_heads"], nh, ctx["at"], ctx["rpn_l"], g1, g2, sigma=3.0,
                                    inside_weight=cfg.TRAIN.RPN_BBOX_INSIDE_WEIGHTS[0], grad_dev=g_dev)
    rfh, rfw = ctx.get("rfh", fh), ctx.get("rfw", fw)  # the RPN input's own geometry (fsod: the correlation map is smaller)
    rhw = rfh * rfw
    dwh, dbh, d_x = ops.linear_backward(d_heads, ctx["rpn_x"], plan["rpn_head_w"], B * rhw, nh, 512)
    ns = rpn.nc_score_out
    _acc(rpn.RPN_cls_score.weight, dwh[:ns])
    _acc(rpn.RPN_cls_score.bias, dbh[:ns])
    _acc(rpn.RPN_bbox_pred.weight, dwh[ns:])
    _acc(rpn.RPN_bbox_pred.bias, dbh[ns:])
    ops.relu_mask_(d_x, ctx["rpn_x"], B * rhw, 512)
    c_rpn = dict(cin=rpn.din, cout=512, k=3, stride=1, pad=1, w=plan["rpn_conv_w"], scale=None, u=plan["rpn_conv_u"])
    grads.add_conv("RCNN_rpn.RPN_Conv", d_x, ctx["rpn_feat"], B, rfh, rfw, c_rpn)
    _acc(rpn.RPN_Conv.bias, ops.colsum(d_x, B * rhw, 512))
    if fsod:
        # attention RPN (fsod.py:109-116): the RPN ran on the depth-wise correlation of base_feat with the pooled positive
        # support -> d base = full correlation of d rfeat with that kernel + RoIAlign path; d kernel -> positive supports
        d_rfeat = conv_dgrad(d_x, B, rfh, rfw, c_rpn)
        gq, d_pos_kernel = ops.depthwise_corr_backward(d_rfeat, ctx["base"], ctx["pos"], B, fh, fw, 1024, 7, 7)
        ops.axpy_rows_(gq, d_bf, B * hw, 1024)
        ops.axpy_rows_(d_supports[0], d_pos_kernel.view(B * 49, 1024), B * 49, 1024)
        for off_, d_sup_ in d_supports.items():  # AvgPool2d(14, 1) of the shots' mean map (fsod.py:98-101)
            to_supports(ops.avgpool_backward(d_sup_, B, 20, 20, 1024, 14, 1), off_)
    elif fgn:
        # the RPN ran on base_feat * pos_rpn[image] (fgn.py:75-82): d base = d rfeat * pos_rpn + RoIAlign path, and
        # d pos_rpn[image] = sum over the pixels of d rfeat * base -> AvgPool2d(20) -> the positive supports' mean map
        d_rfeat = conv_dgrad(d_x, B, fh, fw, c_rpn)
        gq = ops.scale_rows_by_group(d_rfeat, ctx["pos_rpn"], B * hw, hw, 1024)
        ops.axpy_rows_(gq, d_bf, B * hw, 1024)
        d_pos_rpn = (d_rfeat * ctx["base"]).view(B, hw, 1024).sum(1).contiguous()
        to_supports(ops.broadcast_rows(d_pos_rpn, B, ctx["L"], 1024, alpha=1.0 / ctx["L"]).view(B, ctx["L"], 1024), 0)
    else:
        gq = conv_dgrad(d_x, B, fh, fw, c_rpn, residual=d_bf)  # d base_feat = RPN path + RoIAlign path
    grads.finish_all(model, "RCNN_rpn")
    _ready(model, stages[1][1])
    qs, ss = ctx["q_saved"], ctx.get("s_saved") or []
    nblk = len(qs)
    for i in range(nblk - 1, -1, -1):  # query batch and (meta) support batch, block by block: shared weights
        sq = qs[i]
        gq = bottleneck_backward(gq, sq, sq["n"], sq["h"], sq["w"], sq["bp"], grads, sq["key"], need_dx=i > 0,
                                 g_masked=i < nblk - 1)
        if gs is not None:
            s_ = ss[i]
            gs = bottleneck_backward(gs, s_, s_["n"], s_["h"], s_["w"], s_["bp"], grads, s_["key"], need_dx=i > 0,
                                     g_masked=i < nblk - 1)
        grads.finish_all(model, sq["key"] + ".")
        _ready(model, _block_convs(sq["key"], sq["bp"]))
    assert not grads.packed
    _release_ctx(model, ctx)
