"""Sibling model `fgn` of the reference's factory (utils.py:115-116): lib/model/framework/fgn.py:18-259 on the same HIP
operators.

* attention RPN (fgn.py:63-82): base_feat is re-weighted channel-wise by the global mean of the positive supports'
  mean map (`dana_scale_rows_by_group`), the RPN runs on that;
* head (fgn.py:145-165): [support 7x7 | roi 7x7] -> 3x3 conv (2048->512, no padding) -> BatchNorm -> ReLU -> 3x3 conv
  (512->128) -> BatchNorm -> ReLU -> Linear(1152, 2). The concatenation never exists (the support half of the first conv
  is computed once per image and added as a residual). bn1 / bn2 are ORDINARY BatchNorm layers, unlike the trunk's:
  batch statistics and running-statistics updates in train mode (`dana_batch_stats`), running statistics in eval mode.
Same parameter tree as the reference class. Trainable: backward.frcnn_backward's `fgn` branch (train-mode BatchNorm
adjoints, the split first conv, the channel re-weighting of the RPN input, the support trunk)."""
import torch
import torch.nn as nn

from . import ops
from .config import cfg
from .dana import _LossBridge
from .frcnn import FasterRCNN


class FGN(FasterRCNN):
    def __init__(self, classes, num_layers=50, pretrained=False, num_way=2, num_shot=5):
        self.n_way, self.n_shot = num_way, num_shot
        FasterRCNN.__init__(self, classes, num_layers, pretrained)

    def _init_modules(self):
        FasterRCNN._init_modules(self)
        self.cls_conv1 = nn.Conv2d(2048, 512, 3, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(512)
        self.cls_conv2 = nn.Conv2d(512, 128, 3, padding=0, bias=False)
        self.bn2 = nn.BatchNorm2d(128)
        self.RCNN_cls_score = nn.Linear(1152, 2)
        mods = self._modules  # the reference's registration order (fgn.py:29-41 then :207-219)
        for k in ("RCNN_rpn", "cls_conv1", "bn1", "cls_conv2", "bn2", "RCNN_base", "RCNN_top", "RCNN_cls_score",
                  "RCNN_bbox_pred"):
            mods[k] = mods.pop(k)

    def _init_weights(self):  # fgn.py:167-183: RCNN_cls_score keeps its default init
        from .dana import DAnARCNN
        DAnARCNN._init_weights(self)

    def _bn(self, x, rows, bn, save=None):
        """nn.BatchNorm2d on NHWC rows, in place, followed by ReLU: batch statistics + running update when bn.training.
        save: list that receives (pre-BN copy of x, batch mean, batch var) for the backward"""
        C = bn.num_features
        if bn.training:
            mean, var = ops.batch_stats(x, rows, C)
            if save is not None:
                save.append((x.clone(), mean, var))
            with torch.no_grad():  # F.batch_norm's running update: momentum 0.1, UNBIASED variance
                bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
                bn.running_var.mul_(1 - bn.momentum).add_(var, alpha=bn.momentum * rows / max(rows - 1, 1))
                bn.num_batches_tracked += 1
        else:
            mean, var = bn.running_mean, bn.running_var
        scale, shift = ops.bn_fold(bn.weight, bn.bias, mean, var, bn.eps)
        return ops.scale_shift_relu_(x, scale, shift, rows, C, relu=True)

    def forward(self, im_data, im_info, gt_boxes, num_boxes, support_ims, all_cls_gt_boxes=None):
        training = self.training
        shot = self.n_shot
        way = self.n_way if training else 1
        B = im_data.size(0)
        dev = im_data.device
        plan = self._get_plan()
        sup_ims = support_ims.reshape(-1, support_ims.size(2), support_ims.size(3), support_ims.size(4))
        Ns = sup_ims.size(0)
        if Ns != B * way * shot:
            raise RuntimeError("support_ims must hold batch*way*shot = %d images, got %d" % (B * way * shot, Ns))
        ctx = None
        bridge = training and torch.is_grad_enabled()
        if training and (bridge or getattr(self, "save_for_backward", False)):
            if cfg.POOLING_MODE != "align":
                raise NotImplementedError("the HIP backward of fgn covers POOLING_MODE 'align'")
            ctx = dict(q_saved=[], l4_saved=[], s_saved=[], heads=[])
        self._ctx = None
        sup, sh_, sw_ = self._rcnn_base(sup_ims, plan, save=ctx["s_saved"] if ctx is not None else None)  # [Ns*400][1024]
        if (sh_, sw_) != (20, 20):
            raise RuntimeError("support images must be 320x320 (fgn.py:34-35: AvgPool2d(20) / AvgPool2d(14, 1) of a 20x20 map)")
        L = sh_ * sw_

        def mean_map(offset):  # mean over the shots [offset, offset + shot): [B][400*1024]
            m = torch.empty((B, L * 1024), dtype=torch.float32, device=dev)
            for b in range(B):
                m[b:b + 1] = ops.spatial_mean(sup.view(-1)[(b * way * shot + offset) * L * 1024:], 1, shot, L * 1024)
            return m

        pos_map = mean_map(0)
        pos_rpn = ops.spatial_mean(pos_map, B, L, 1024)            # AvgPool2d(20): [B][1024]
        pos_rcnn = ops.avgpool(pos_map, B, sh_, sw_, 1024, 14, 1)  # AvgPool2d(14, 1): [B][49][1024]

        def attention_rpn_input(base, B_, fh, fw, plan_):
            if ctx is not None:
                ctx["base"] = base
            return ops.scale_rows_by_group(base, pos_rpn, B_ * fh * fw, fh * fw, 1024), fh, fw

        st = self._stages(im_data, im_info, gt_boxes, rpn_input=attention_rpn_input, ctx=ctx)
        R, n_roi, pooled, fc7 = st["R"], st["n_roi"], st["pooled"], st["fc7"]
        wb, bb = self._w(self.RCNN_bbox_pred)
        bbox_pred = ops.gemm_nt(fc7, wb, n_roi, 4, 2048, shift=bb)
        w1 = self.cls_conv1.weight.detach()
        w1_sup = ops.pack_conv_weight(w1[:, :1024].contiguous())   # torch.cat([support, roi], 1): support channels first
        w1_roi = ops.pack_conv_weight(w1[:, 1024:].contiguous())
        w2 = ops.pack_conv_weight(self.cls_conv2.weight)
        # Linear(1152, 2) reads the NCHW flatten (c, h, w); the activations here are (h, w, c)
        wl = self.RCNN_cls_score.weight.detach().view(2, 128, 9).permute(0, 2, 1).reshape(2, 1152).contiguous()
        bl = self.RCNN_cls_score.bias.detach().contiguous()
        roi_half, _, _ = ops.conv2d_nhwc(pooled, n_roi, 7, 7, 1024, w1_roi, 512, 3, 3, 1, 0)  # [n*25][512], shared

        def head(support, offset):  # support [B][49][1024]
            saved = [] if ctx is not None else None
            s_half, _, _ = ops.conv2d_nhwc(support, B, 7, 7, 1024, w1_sup, 512, 3, 3, 1, 0)  # [B*25][512]
            x = ops.broadcast_rows(s_half, B, R, 25 * 512)                                      # [n*25][512]
            ops.axpy_rows_(x, roi_half, n_roi * 25, 512)
            x1 = self._bn(x, n_roi * 25, self.bn1, save=saved)
            x, _, _ = ops.conv2d_nhwc(x1, n_roi, 5, 5, 512, w2, 128, 3, 3, 1, 0)                # [n*9][128]
            x2 = self._bn(x, n_roi * 9, self.bn2, save=saved)
            score = ops.gemm_nt(x2, wl, n_roi, 2, 1152, shift=bl)
            if ctx is not None:
                ctx["heads"].append(dict(offset=offset, support=support, x1=x1, x2=x2, bn1=saved[0], bn2=saved[1]))
            return ops.softmax_rows_(score.clone(), n_roi, 2), score

        cls_prob, cls_score = head(pos_rcnn, 0)
        RCNN_loss_cls = RCNN_loss_bbox = 0
        rois_label = st["rois_label"]
        if training:
            neg_prob, neg_score = head(ops.avgpool(mean_map(shot), B, sh_, sw_, 1024, 14, 1), shot)
            cls_prob = torch.cat([cls_prob, neg_prob], 0)
            rois_label = torch.cat([rois_label, torch.zeros_like(rois_label)], 0)
            rl, seeds = ops.rcnn_losses(cls_score, neg_score, st["labels_f"], bbox_pred, st["rois_target"].contiguous(),
                                        st["rois_inside_ws"].contiguous(), st["rois_outside_ws"].contiguous(),
                                        with_grad=ctx is not None)
            RCNN_loss_cls, RCNN_loss_bbox = rl[0], rl[1]
        rpn_loss_cls, rpn_loss_bbox = st["rpn_loss_cls"], st["rpn_loss_bbox"]
        if ctx is not None:
            ctx.update(loss_seeds=seeds, sup=sup, Ns=Ns, shot=shot, way=way, L=L, pos_rpn=pos_rpn, pooled=pooled,
                       w1_sup=w1_sup, w1_roi=w1_roi, w2=w2, wl=wl)
            self._ctx = ctx
            if bridge:  # loss.backward() (train.py:141-143) runs backward.frcnn_backward (fgn branch) on the HIP kernels
                if self._grad_anchor is None or self._grad_anchor.device != dev:
                    self._grad_anchor = torch.zeros(1, device=dev, requires_grad=True)
                rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox = _LossBridge.apply(
                    self._grad_anchor, self, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox)
        return (st["rois"], cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox, rois_label)
