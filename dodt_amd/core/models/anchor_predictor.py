"""RPN anchor predictor on the device: the dense head between the 3x3 RPN crops and NMS #1
(avod/core/models/dt_rpn_model.py:430-537).  slim.conv2d 3x3 VALID on a 3x3 crop and the 1x1
convs are fully connected layers over the anchors."""
import numpy as np

from dodt_amd import ops


class AnchorPredictor(object):
    def __init__(self, ctx, params, dtype='f32'):
        """params[name] = dict(w, b), names cls_fc6/7/8 and reg_fc6/7/8, TF conv shapes."""
        def mat(name):
            w = np.asarray(params[name]['w'], np.float32)
            return w.reshape(-1, w.shape[-1]), np.asarray(params[name]['b'], np.float32)
        wc, bc = mat('cls_fc6')
        wr, br = mat('reg_fc6')
        self.width = wc.shape[1]
        if wr.shape != wc.shape:
            raise ValueError('cls_fc6 and reg_fc6 must have the same size')
        # both branches read the same fused crop: one launch, columns [cls | reg]
        self.fc6 = ops.FullyConnected(ctx, np.concatenate([wc, wr], 1),
                                      np.concatenate([bc, br]), True, dtype=dtype)
        self.cls7 = ops.FullyConnected(ctx, *mat('cls_fc7'), relu=True, dtype=dtype)
        self.reg7 = ops.FullyConnected(ctx, *mat('reg_fc7'), relu=True, dtype=dtype)
        self.cls8 = ops.FullyConnected(ctx, *mat('cls_fc8'), relu=False, dtype=dtype)
        self.reg8 = ops.FullyConnected(ctx, *mat('reg_fc8'), relu=False, dtype=dtype)
        self.layers = [self.fc6, self.cls7, self.reg7, self.cls8, self.reg8]
        # the two output layers as one launch over the [cls | reg] rows of fc7: block-diagonal weights
        # (a branch's columns see zeros for the other branch's activations); fp32 heads
        (w8c, b8c), (w8r, b8r) = mat('cls_fc8'), mat('reg_fc8')
        w8 = np.zeros((w8c.shape[0] + w8r.shape[0], w8c.shape[1] + w8r.shape[1]), np.float32)
        w8[:w8c.shape[0], :w8c.shape[1]] = w8c
        w8[w8c.shape[0]:, w8c.shape[1]:] = w8r
        self.out8 = ops.FullyConnected(ctx, w8, np.concatenate([b8c, b8r]), False, dtype=dtype)
        if not (self.out8.can_split() and w8c.shape[0] == self.width == w8r.shape[0]):
            self.out8.close()
            self.out8 = None
        self.ctx = ctx

    def make_scratch(self, n_max):
        """Hidden activations [cls | reg] of fc6 and fc7; one set per concurrent stream."""
        return [self.ctx.empty((n_max, 2 * self.width), np.float32) for _ in range(2)]

    def forward(self, ctx, d_bev_roi, d_img_roi, n, d_objectness, d_offsets, scratch):
        """rois (n,3,3,1) -> objectness (n,2), offsets (n,6); all on ctx's stream."""
        w = self.width
        h6, h7 = scratch
        self.fc6.forward(d_bev_roi, n, h6, d_x2=d_img_roi, ctx=ctx)
        self.cls7.forward(h6, n, h7, ldx=2 * w, ldy=2 * w, ctx=ctx)
        self.reg7.forward(h6.offset(4 * w, (1,)), n, h7.offset(4 * w, (1,)),
                          ldx=2 * w, ldy=2 * w, ctx=ctx)
        if self.out8 is not None:
            self.out8.forward_split(h7, n, [d_objectness, d_offsets], [self.cls8.N, self.reg8.N], ctx=ctx)
            return
        self.cls8.forward(h7, n, d_objectness, ldx=2 * w, ctx=ctx)
        self.reg8.forward(h7.offset(4 * w, (1,)), n, d_offsets, ldx=2 * w, ctx=ctx)

    def flops(self, n):
        return sum(l.flops(n) for l in self.layers)

    def close(self):
        for l in self.layers + ([self.out8] if self.out8 is not None else []):
            l.close()
