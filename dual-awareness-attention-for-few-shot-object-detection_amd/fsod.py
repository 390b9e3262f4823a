"""Sibling model `fsod` of the reference's factory (utils.py:111-112): the attention-RPN + multi-relation detector of
lib/model/framework/fsod.py:19-327 on the same HIP operators.

* attention RPN (fsod.py:96-119): the query feature map is cross-correlated depth-wise with the 7x7 pooled mean of the
  positive supports (`dana_depthwise_corr_nhwc`), and the RPN runs on that (h-6) x (w-6) map;
* multi-relation head (fsod.py:181-249): global relation (mean-pooled [roi | support] -> 2 FC -> 2-way), local
  correlation (1x1 conv on both, depth-wise 7x7 correlation -> 2-way), patch relation (1x1 conv on [roi | support],
  3x3/1 average pool, 3x3 conv, 1x1 conv, average pool -> 2-way); the three scores are summed and divided by 10.
  The [roi | support] concatenations never exist: the 1x1 layers are split into their roi and support halves, the
  support half is computed once per image and added as a residual.
Same parameter tree as the reference class. Trainable: backward.frcnn_backward's `fsod` branch (the three relation heads,
the depth-wise correlation adjoints of the attention RPN and of the local-correlation head, the support trunk)."""
import torch
import torch.nn as nn

from . import ops
from .config import cfg
from .dana import _LossBridge
from .frcnn import FasterRCNN


class FSOD(FasterRCNN):
    def __init__(self, classes, num_layers=50, pretrained=False, num_way=2, num_shot=5):
        self.n_way, self.n_shot = num_way, num_shot
        FasterRCNN.__init__(self, classes, num_layers, pretrained)

    def _init_modules(self):
        FasterRCNN._init_modules(self)
        del self.RCNN_cls_score  # fsod.py:267: the relation heads replace it
        d = 1024
        self.global_fc_1, self.global_fc_2, self.global_cls_score = nn.Linear(2 * d, d), nn.Linear(d, d), nn.Linear(d, 2)
        self.corr_conv = nn.Conv2d(d, d, 1, padding=0, bias=False)
        self.corr_cls_score = nn.Linear(d, 2)
        self.patch_conv_1 = nn.Conv2d(2 * d, d // 4, 1, padding=0, bias=False)
        self.patch_conv_2 = nn.Conv2d(d // 4, d // 4, 3, padding=0, bias=False)
        self.patch_conv_3 = nn.Conv2d(d // 4, d, 1, padding=0, bias=False)
        self.patch_cls_score = nn.Linear(d, 2)
        for m in (self.global_fc_1, self.global_fc_2, self.global_cls_score, self.corr_conv, self.corr_cls_score,
                  self.patch_conv_1, self.patch_conv_2, self.patch_conv_3, self.patch_cls_score):
            nn.init.normal_(m.weight, std=0.01)  # fsod.py:49-75
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        # keep the reference's registration order of the state_dict (fsod.py:29-75 then :262-268)
        order = ["RCNN_rpn", "global_fc_1", "global_fc_2", "global_cls_score", "corr_conv", "corr_cls_score",
                 "patch_conv_1", "patch_conv_2", "patch_conv_3", "patch_cls_score", "RCNN_base", "RCNN_top",
                 "RCNN_bbox_pred"]
        mods = self._modules
        for k in order:
            mods[k] = mods.pop(k)

    def _init_weights(self):
        from .dana import DAnARCNN
        DAnARCNN._init_weights(self)

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
                raise NotImplementedError("the HIP backward of fsod covers POOLING_MODE 'align'")
            ctx = dict(q_saved=[], l4_saved=[], s_saved=[], heads=[])
        self._ctx = None
        sup, sh_, sw_ = self._rcnn_base(sup_ims, plan, save=ctx["s_saved"] if ctx is not None else None)  # [Ns*400][1024]
        if (sh_, sw_) != (20, 20):
            raise RuntimeError("support images must be 320x320 (fsod.py:44: AvgPool2d(14) of a 20x20 map -> 7x7)")
        L = sh_ * sw_

        def pooled_support(offset):  # mean over the shots [offset, offset+shot), then AvgPool2d(14, 1): [B][49][1024]
            m = torch.empty((B, L * 1024), dtype=torch.float32, device=dev)
            for b in range(B):
                m[b:b + 1] = ops.spatial_mean(sup.view(-1)[(b * way * shot + offset) * L * 1024:], 1, shot, L * 1024)
            return ops.avgpool(m, B, sh_, sw_, 1024, 14, 1)

        pos = pooled_support(0)

        def attention_rpn_input(base, B_, fh, fw, plan_):
            if ctx is not None:
                ctx["base"] = base
            return ops.depthwise_corr(base, pos, B_, fh, fw, 1024, 7, 7)

        st = self._stages(im_data, im_info, gt_boxes, rpn_input=attention_rpn_input, ctx=ctx)
        R, n_roi, pooled, fc7 = st["R"], st["n_roi"], st["pooled"], st["fc7"]
        wb, bb = self._w(self.RCNN_bbox_pred)
        bbox_pred = ops.gemm_nt(fc7, wb, n_roi, 4, 2048, shift=bb)
        P2, d = 49, 1024
        # roi halves of the relation heads: computed once, shared by the positive and the negative support
        w1, b1 = self._w(self.global_fc_1)
        w2, b2 = self._w(self.global_fc_2)
        wg, bg = self._w(self.global_cls_score)
        wcc = self.corr_conv.weight.detach().view(d, d).contiguous()
        wcs, bcs = self._w(self.corr_cls_score)
        wp1 = self.patch_conv_1.weight.detach().view(d // 4, 2 * d).contiguous()
        wp2 = ops.pack_conv_weight(self.patch_conv_2.weight)
        wp3 = self.patch_conv_3.weight.detach().view(d, d // 4).contiguous()
        wps, bps = self._w(self.patch_cls_score)
        g_roi = ops.spatial_mean(pooled, n_roi, P2, d)                       # avgpool_fc of the roi half [n][1024]
        corr_roi = ops.gemm_nt(pooled, wcc, n_roi * P2, d, d)                # corr_conv(rois) [n*49][1024]

        def head(support, offset=0):  # support [B][49][1024]
            # global relation (fsod.py:185-199): fc1([mean(roi) | mean(support)]) = roi half + support half
            m_sup = ops.spatial_mean(support, B, P2, d)
            g_sup = ops.gemm_nt(m_sup, w1.view(-1)[d:], B, d, d, ldb=2 * d)
            h1 = ops.gemm_nt(g_roi, w1, n_roi, d, d, ldb=2 * d, shift=b1, residual=ops.broadcast_rows(g_sup, B, R, d),
                             ldr=d, relu=True)
            h2 = ops.gemm_nt(h1, w2, n_roi, d, d, shift=b2, relu=True)
            s_g = ops.gemm_nt(h2, wg, n_roi, 2, d, shift=bg)
            # local correlation (fsod.py:201-216)
            corr_sup = ops.gemm_nt(support, wcc, B * P2, d, d)
            oc, _, _ = ops.depthwise_corr(corr_roi, corr_sup, n_roi, 7, 7, d, 7, 7, maps_per_kernel=R)
            s_c = ops.gemm_nt(oc, wcs, n_roi, 2, d, shift=bcs)
            # patch relation (fsod.py:218-234)
            p_sup = ops.gemm_nt(support, wp1.view(-1)[d:], B * P2, d // 4, d, ldb=2 * d)  # [B][49*256]
            x0 = ops.gemm_nt(pooled, wp1, n_roi * P2, d // 4, d, ldb=2 * d,
                             residual=ops.broadcast_rows(p_sup, B, R, P2 * (d // 4)), ldr=d // 4, relu=True)
            x1 = ops.avgpool(x0, n_roi, 7, 7, d // 4, 3, 1)                                    # 7x7 -> 5x5
            x2, _, _ = ops.conv2d_nhwc(x1, n_roi, 5, 5, d // 4, wp2, d // 4, 3, 3, 1, 0, relu=True)  # -> 3x3
            x3 = ops.gemm_nt(x2, wp3, n_roi * 9, d, d // 4, relu=True)
            x4 = ops.avgpool(x3, n_roi, 3, 3, d, 3, 1)                                         # -> 1x1
            s_p = ops.gemm_nt(x4, wps, n_roi, 2, d, shift=bps)
            score = (s_g + s_c + s_p) / 10.0  # fsod.py:237 (soft_gamma)
            if ctx is not None:
                ctx["heads"].append(dict(offset=offset, support=support, m_sup=m_sup, h1=h1, h2=h2, corr_sup=corr_sup, oc=oc,
                                         x0=x0, x1=x1, x2=x2, x3=x3, x4=x4))
            return ops.softmax_rows_(score.clone(), n_roi, 2), score.contiguous()

        cls_prob, cls_score = head(pos)
        RCNN_loss_cls = RCNN_loss_bbox = 0
        rois_label = st["rois_label"]
        if training:
            neg_prob, neg_score = head(pooled_support(shot), shot)
            cls_prob = torch.cat([cls_prob, neg_prob], 0)
            rois_label = torch.cat([rois_label, torch.zeros_like(rois_label)], 0)
            rl, seeds = ops.rcnn_losses(cls_score, neg_score, st["labels_f"], bbox_pred, st["rois_target"].contiguous(),
                                        st["rois_inside_ws"].contiguous(), st["rois_outside_ws"].contiguous(),
                                        with_grad=ctx is not None)
            RCNN_loss_cls, RCNN_loss_bbox = rl[0], rl[1]
        rpn_loss_cls, rpn_loss_bbox = st["rpn_loss_cls"], st["rpn_loss_bbox"]
        if ctx is not None:
            ctx.update(loss_seeds=seeds, Ns=Ns, shot=shot, way=way, L=L, pos=pos, pooled=pooled, g_roi=g_roi,
                       corr_roi=corr_roi, wp2=wp2)
            self._ctx = ctx
            if bridge:  # loss.backward() (train.py:141-143) runs backward.frcnn_backward (fsod branch) on the HIP kernels
                if self._grad_anchor is None or self._grad_anchor.device != dev:
                    self._grad_anchor = torch.zeros(1, device=dev, requires_grad=True)
                rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox = _LossBridge.apply(
                    self._grad_anchor, self, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox)
        return (st["rois"], cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox, rois_label)
