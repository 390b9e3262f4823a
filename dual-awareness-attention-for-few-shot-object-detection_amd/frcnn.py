"""Sibling model `frcnn` of the reference's factory (utils.py:109-110): the plain class-agnostic Faster R-CNN of
lib/model/framework/faster_rcnn.py:17-203 on the SAME HIP operators as the DAnA path --
Caffe ResNet-50 trunk -> RPN -> proposal layer -> (train) anchor / proposal targets -> RoIAlign or RoIPool ->
layer4 -> RCNN_cls_score / RCNN_bbox_pred -> losses. Same parameter tree and state_dict keys as the reference class.
`frcnn` and `meta` are trainable on the HIP kernels too: a training forward saves its context and hands the four losses
to autograd (`_LossBridge`), `loss.backward()` runs backward.frcnn_backward / meta_backward (POOLING_MODE 'align')."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import cfg
from .dana import DAnARCNN, _LossBridge, _RPNParams


class FasterRCNN(DAnARCNN):
    def __init__(self, classes, num_layers=50, pretrained=False):
        nn.Module.__init__(self)
        self.model_path = "data/pretrained_model/resnet50_caffe.pth"
        self.dout_base_model = 1024
        self.pretrained = pretrained
        self.classes = classes
        self.n_classes = len(classes)
        self.class_agnostic = True
        self.semantic_enhance = False
        self.use_winograd = True
        self.winograd_tile = 4
        self.winograd_min_cin = 128
        self.merge_trunk = False
        self.presplit_weights = True
        self.nms_inclusive = False
        self.device_rng, self.rng_seed, self._rng_calls = False, 1996, 0
        self.RCNN_rpn = _RPNParams(self.dout_base_model)
        self._plan, self._consts, self._conv_cache, self._epoch = None, {}, {}, 0
        self._ctx = self._grad_anchor = None
        self._init_modules()
        self._init_weights()

    def _init_modules(self):
        DAnARCNN._init_modules(self)  # trunk, RCNN_top, RCNN_bbox_pred, the freezing rules (faster_rcnn.py:129-160)
        self.RCNN_cls_score = nn.Linear(2048, self.n_classes)

    def _init_weights(self):
        DAnARCNN._init_weights(self)
        self.RCNN_cls_score.weight.data.normal_(0, 0.01)
        self.RCNN_cls_score.bias.data.zero_()

    # ---- shared stages of the sibling detectors (frcnn, meta): trunk -> RPN -> targets -> RoI features -> layer4 ----
    def _stages(self, im_data, im_info, gt_boxes, anchor_gt_boxes=None, rpn_input=None, ctx=None):
        """-> dict(B, R, n_roi, rois, rpn losses, rois_label / targets (train), pooled, fc7 [n_roi][2048]).
        anchor_gt_boxes: boxes the anchor-target layer sees (meta.py:65 passes ALL classes' boxes); default gt_boxes.
        rpn_input(base, B, fh, fw, plan) -> (feature [B*h*w][1024], h, w): what the RPN runs on instead of base_feat
        (fsod.py:109-119: the attention RPN's correlation map, which is smaller than base_feat)"""
        plan = self._get_plan()
        dev = im_data.device
        training = self.training
        B = im_data.size(0)
        im_info = im_info.data.float().contiguous()
        gt_boxes = gt_boxes.data
        anchor_gt = gt_boxes if anchor_gt_boxes is None else anchor_gt_boxes.data
        inputs_ready = ops.record_event()
        main = ops.cur_stream()
        # ctx (frcnn only): everything backward.frcnn_backward needs is saved into it
        base, fh, fw = self._rcnn_base(im_data, plan, save=ctx["q_saved"] if ctx is not None else None)  # faster_rcnn.py:43
        # -- RPN (rpn.py:58-115) on base_feat (or on the model's own RPN input) --
        rfeat, rh, rw = (base, fh, fw) if rpn_input is None else rpn_input(base, B, fh, fw, plan)
        base_hw = (fh, fw)
        fh, fw = rh, rw  # the RPN / anchor / proposal geometry below is the RPN input's
        hw = fh * fw
        rpn = self.RCNN_rpn
        if plan["rpn_conv_u"] is not None:
            x, _, _ = ops.conv3x3_winograd(rfeat, B, fh, fw, rpn.din, plan["rpn_conv_b3"] or plan["rpn_conv_u"], 512, shift=plan["rpn_conv_b"],
                                           relu=True)
        else:
            x, _, _ = ops.conv2d_nhwc(rfeat, B, fh, fw, rpn.din, plan["rpn_conv_b3"] or plan["rpn_conv_w"], 512, 3, 3, 1, 1,
                                      shift=plan["rpn_conv_b"], relu=True)
        nh = rpn.nc_score_out + rpn.nc_bbox_out
        heads = ops.gemm_nt(x, plan["rpn_head_w3"] or plan["rpn_head_w"], B * hw, nh, 512, shift=plan["rpn_head_b"])
        A = plan["anchors"].size(0)
        key = "TRAIN" if training else "TEST"
        rois = ops.proposal_layer(heads, (hw * nh, 1, nh), False, heads.view(-1)[rpn.nc_score_out:], (hw * nh, 1, nh),
                                  im_info, plan["anchors"], B, A, fh, fw, rpn.feat_stride, cfg[key].RPN_PRE_NMS_TOP_N,
                                  cfg[key].RPN_POST_NMS_TOP_N, cfg[key].RPN_NMS_THRESH, self.nms_inclusive)
        st = dict(B=B, rpn_loss_cls=0, rpn_loss_bbox=0, rois_label=None, labels_f=None)
        if training:
            tr_ = cfg.TRAIN
            side = self._stream("targets", dev)
            side.wait_event(inputs_ready)
            with ops.on_stream(side):
                at = ops.anchor_target_assign(anchor_gt.float(), im_info, plan["anchors"], fh, fw, rpn.feat_stride,
                                              tr_.RPN_NEGATIVE_OVERLAP, tr_.RPN_POSITIVE_OVERLAP, tr_.RPN_BATCHSIZE,
                                              tr_.RPN_FG_FRACTION)
            at["ibuf"].record_stream(main)
            at["labels"].record_stream(main)
            main.wait_stream(side)
            rpn_l = ops.rpn_losses(heads, nh, at, sigma=3.0, inside_weight=tr_.RPN_BBOX_INSIDE_WEIGHTS[0])
            st["rpn_loss_cls"], st["rpn_loss_bbox"] = rpn_l[0], rpn_l[1]
            if ctx is not None:
                ctx.update(rpn_x=x, rpn_heads=heads, at=at, rpn_l=rpn_l, nh=nh, rpn_feat=rfeat, rfh=fh, rfw=fw)
            fg_per = int(np.round(tr_.FG_FRACTION * tr_.BATCH_SIZE)) or 1
            rois, rois_label, rois_target, rois_inside_ws, rois_outside_ws = ops.proposal_target_layer(
                rois, gt_boxes.float(), int(tr_.BATCH_SIZE), fg_per, tr_.FG_THRESH, tr_.BG_THRESH_HI, tr_.BG_THRESH_LO,
                tr_.BBOX_NORMALIZE_MEANS, tr_.BBOX_NORMALIZE_STDS, tr_.BBOX_INSIDE_WEIGHTS,
                tr_.BBOX_NORMALIZE_TARGETS_PRECOMPUTED)
            st["labels_f"] = rois_label.reshape(-1).contiguous()
            st["rois_label"] = st["labels_f"].long()
            st["rois_target"] = rois_target.view(-1, 4)
            st["rois_inside_ws"] = rois_inside_ws.view(-1, 4)
            st["rois_outside_ws"] = rois_outside_ws.view(-1, 4)
        R = rois.size(1)
        n_roi = B * R
        P = cfg.POOLING_SIZE
        fh, fw = base_hw
        # -- RoI pooling (faster_rcnn.py:70-73) on base_feat: both modes of the reference --
        if cfg.POOLING_MODE == "align":
            pooled, _ = ops.roi_align_forward_nhwc(base, B, fh, fw, 1024, 1024, rois.view(-1, 5), 1.0 / 16.0, P, 0)
        elif cfg.POOLING_MODE == "pool":
            nchw = ops.nhwc_to_nchw(base, B, 1024, fh, fw)
            pooled_nchw, _ = ops.roi_pool_forward(nchw, rois.view(-1, 5).contiguous(), 1.0 / 16.0, P, P)
            pooled = ops.nchw_to_nhwc(pooled_nchw)
        else:
            raise NotImplementedError("POOLING_MODE '%s'" % cfg.POOLING_MODE)
        st.update(rois=rois, R=R, n_roi=n_roi, pooled=pooled, plan=plan,
                  fc7=self._head_to_tail(pooled, n_roi, P, P, plan, save=ctx["l4_saved"] if ctx is not None else None))
        if ctx is not None:
            ctx.update(plan=plan, B=B, R=R, rois=rois, fh=fh, fw=fw, fc7=st["fc7"])
        return st

    def _head_to_tail(self, x, n, h, w, plan, save=None):
        """layer4 + spatial mean (faster_rcnn.py:183-185) on an NHWC batch of n maps -> [n][2048]"""
        for bi, bp in enumerate(plan["layer4"]):
            x, h, w = self._bottleneck(x, n, h, w, bp, save=save, key="RCNN_top.0.%d" % bi)
        return ops.spatial_mean(x, n, h * w, 2048)

    def forward(self, im_data, im_info, gt_boxes, num_boxes):
        ctx = None
        bridge = self.training and torch.is_grad_enabled() and type(self) is FasterRCNN
        if self.training and type(self) is FasterRCNN and (bridge or getattr(self, "save_for_backward", False)):
            if cfg.POOLING_MODE != "align":
                raise NotImplementedError("the HIP backward of frcnn covers POOLING_MODE 'align'")
            ctx = dict(q_saved=[], l4_saved=[])
        self._ctx = None
        st = self._stages(im_data, im_info, gt_boxes, ctx=ctx)
        B, R, n_roi, fc7 = st["B"], st["R"], st["n_roi"], st["fc7"]
        wb, bb = self._w(self.RCNN_bbox_pred)
        wc, bc = self._w(self.RCNN_cls_score)
        bbox_pred = ops.gemm_nt(fc7, wb, n_roi, 4, 2048, shift=bb)
        cls_score = ops.gemm_nt(fc7, wc, n_roi, self.n_classes, 2048, shift=bc)
        cls_prob = ops.softmax_rows_(cls_score.clone(), n_roi, self.n_classes)
        RCNN_loss_cls = RCNN_loss_bbox = 0
        rpn_loss_cls, rpn_loss_bbox = st["rpn_loss_cls"], st["rpn_loss_bbox"]
        if self.training:  # faster_rcnn.py:93-98
            if ctx is None:  # the same fused launch, without the gradient seeds
                l2, _ = ops.plain_rcnn_losses(cls_score, st["rois_label"], bbox_pred, st["rois_target"],
                                              st["rois_inside_ws"], st["rois_outside_ws"], with_grad=False)
                RCNN_loss_cls, RCNN_loss_bbox = l2[0], l2[1]
            else:
                # the two loss tails ([n_roi][2], [n_roi][4]) and their gradient seeds: one HIP launch (dana_plain_rcnn_loss)
                l2, (d_cls, d_bbox) = ops.plain_rcnn_losses(cls_score, st["rois_label"], bbox_pred, st["rois_target"],
                                                           st["rois_inside_ws"], st["rois_outside_ws"], with_grad=True)
                RCNN_loss_cls, RCNN_loss_bbox = l2[0], l2[1]
                ctx.update(loss_seeds=(d_cls, d_bbox))
                self._ctx = ctx
                if bridge:  # loss.backward() (train.py:141-143) runs backward.frcnn_backward on the HIP kernels
                    dev = im_data.device
                    if self._grad_anchor is None or self._grad_anchor.device != dev:
                        self._grad_anchor = torch.zeros(1, device=dev, requires_grad=True)
                    rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox = _LossBridge.apply(
                        self._grad_anchor, self, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox)
        return (st["rois"], cls_prob.view(B, R, -1), bbox_pred.view(B, R, -1), rpn_loss_cls, rpn_loss_bbox,
                RCNN_loss_cls, RCNN_loss_bbox, st["rois_label"])


class MetaRCNN(FasterRCNN):
    """Sibling model `meta` (utils.py:113-114): Meta R-CNN, lib/model/framework/meta.py:18-251. The Predictor-head
    Remodeling Network turns every support image into a class-attentive vector, sigmoid(mean(layer4(maxpool2(trunk)))),
    the shots' mean multiplies the RoI features channel-wise in front of a 2-way Linear; positive + negative supports
    and the 1:2:1 hard-negative-mined loss as in DAnA. Trainable: backward.meta_backward."""

    def __init__(self, classes, num_layers=50, pretrained=False, num_way=2, num_shot=5):
        self.n_way, self.n_shot = num_way, num_shot
        FasterRCNN.__init__(self, classes, num_layers, pretrained)

    def _init_modules(self):
        FasterRCNN._init_modules(self)
        self.RCNN_cls_score = nn.Sequential(nn.Linear(2048, 2))  # meta.py:199-201 (state_dict key RCNN_cls_score.0.*)

    def _init_weights(self):  # meta.py:144-159: RCNN_cls_score keeps its default init
        from .dana import DAnARCNN
        DAnARCNN._init_weights(self)

    def forward(self, im_data, im_info, gt_boxes, num_boxes, support_ims, all_cls_gt_boxes=None):
        if all_cls_gt_boxes is None:
            raise RuntimeError("meta: all_cls_gt_boxes is required (meta.py:48,65)")
        training = self.training
        shot = self.n_shot
        way = self.n_way if training else 1
        ctx = None
        bridge = training and torch.is_grad_enabled()
        if training and (bridge or getattr(self, "save_for_backward", False)):
            if cfg.POOLING_MODE != "align":
                raise NotImplementedError("the HIP backward of meta covers POOLING_MODE 'align'")
            ctx = dict(q_saved=[], l4_saved=[], s_saved=[], sl4_saved=[], heads=[])
        self._ctx = None
        st = self._stages(im_data, im_info, gt_boxes, anchor_gt_boxes=all_cls_gt_boxes, ctx=ctx)
        B, R, n_roi, fc7, plan = st["B"], st["R"], st["n_roi"], st["fc7"], st["plan"]
        # PRN (meta.py:241-251) on every support image
        sup_ims = support_ims.reshape(-1, support_ims.size(2), support_ims.size(3), support_ims.size(4))
        Ns = sup_ims.size(0)
        if Ns != B * way * shot:
            raise RuntimeError("support_ims must hold batch*way*shot = %d images, got %d" % (B * way * shot, Ns))
        sup, sh_, sw_ = self._rcnn_base(sup_ims, plan, save=ctx["s_saved"] if ctx is not None else None)
        mp, mh, mw = ops.maxpool2x2s2(sup, Ns, sh_, sw_, 1024)
        att = ops.sigmoid_(self._head_to_tail(mp, Ns, mh, mw, plan, save=ctx["sl4_saved"] if ctx is not None else None))
        wb, bb = self._w(self.RCNN_bbox_pred)
        wc, bc = self._w(self.RCNN_cls_score[0])
        bbox_pred = ops.gemm_nt(fc7, wb, n_roi, 4, 2048, shift=bb)

        def head(offset):  # supports [offset, offset + shot) of every episode: their mean vector x the RoI features
            vec = torch.empty((B, 2048), dtype=torch.float32, device=fc7.device)
            for b in range(B):
                vec[b:b + 1] = ops.spatial_mean(att.view(-1)[(b * way * shot + offset) * 2048:], 1, shot, 2048)
            comb = ops.scale_rows_by_group(fc7, vec, n_roi, R, 2048)
            score = ops.gemm_nt(comb, wc, n_roi, 2, 2048, shift=bc)
            if ctx is not None:
                ctx["heads"].append(dict(offset=offset, vec=vec, comb=comb))
            return ops.softmax_rows_(score.clone(), n_roi, 2), score

        cls_prob, cls_score = head(0)
        RCNN_loss_cls = RCNN_loss_bbox = 0
        rois_label = st["rois_label"]
        rpn_loss_cls, rpn_loss_bbox = st["rpn_loss_cls"], st["rpn_loss_bbox"]
        if training:
            neg_prob, neg_score = head(shot)
            cls_prob = torch.cat([cls_prob, neg_prob], 0)
            rois_label = torch.cat([rois_label, torch.zeros_like(rois_label)], 0)
            rl, seeds = ops.rcnn_losses(cls_score, neg_score, st["labels_f"], bbox_pred, st["rois_target"].contiguous(),
                                        st["rois_inside_ws"].contiguous(), st["rois_outside_ws"].contiguous(),
                                        with_grad=ctx is not None)
            RCNN_loss_cls, RCNN_loss_bbox = rl[0], rl[1]
            if ctx is not None:
                ctx.update(loss_seeds=seeds, att=att, sup=sup, sup_hw=(sh_, sw_), mp_hw=(mh, mw), Ns=Ns, shot=shot, way=way)
                self._ctx = ctx
                if bridge:  # loss.backward() (train.py:141-143) runs backward.meta_backward on the HIP kernels
                    dev = im_data.device
                    if self._grad_anchor is None or self._grad_anchor.device != dev:
                        self._grad_anchor = torch.zeros(1, device=dev, requires_grad=True)
                    rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox = _LossBridge.apply(
                        self._grad_anchor, self, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox)
        return (st["rois"], cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox, rois_label)
