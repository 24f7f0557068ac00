"""Stage-2 dense heads on the device: early-fusion fully connected layers
(avod/core/avod_fc_layers/fusion_fc_layers.py:94-180 with fusion_method 'mean') and the
correlation-offsets head built from the same stack
(avod/builders/avod_corr_layers_builder.py:126-169)."""
import numpy as np

from dodt_amd import ops


class EarlyFusionFcLayers(object):
    """mean of the input crops -> flatten -> fc6.. (ReLU) -> linear output layers
    (build_output_layers, fusion_fc_layers.py:94-133).  `outputs` names the output layers in
    params: ('cls_out', 'off_out', 'ang_out') for box_4ca / box_3d, whose ANG_VECS_OUTPUT_SIZE
    is 2 (avod_fc_layer_utils.py:11-17; the DODT config's avod_box_representation,
    pyramid_cars_with_aug_dt_5_tracking.config:29), ('cls_out', 'off_out') for box_4c / box_8c,
    ('off_out',) for the corr head."""

    def __init__(self, ctx, params, outputs=('cls_out', 'off_out', 'ang_out'), dtype='f32'):
        names = sorted(k for k in params if k.startswith('fc'))
        self.hidden = [ops.FullyConnected(ctx, params[k]['w'], params[k]['b'], True, dtype=dtype)
                       for k in names]
        self.outputs = [ops.FullyConnected(ctx, params[k]['w'], params[k]['b'], False, dtype=dtype)
                        for k in outputs]
        self.width = max(l.N for l in self.hidden)
        self.ctx = ctx

    def make_scratch(self, n_max):
        """Ping-pong hidden activations; one set per concurrent stream."""
        return [self.ctx.empty((n_max, self.width), np.float32) for _ in range(2)]

    def forward(self, ctx, d_rois, d_rois2, n, d_n, d_outs, scratch):
        """d_rois (n,h,w,c) [and d_rois2, fused by mean]; *d_n rows are valid;
        d_outs: one (n, size) array per output layer."""
        x, x2, ldx = d_rois, d_rois2, None
        for i, l in enumerate(self.hidden):
            y = scratch[i & 1]
            l.forward(x, n, y, ldx=ldx, ldy=self.width, d_x2=x2, d_m=d_n, ctx=ctx)
            x, x2, ldx = y, None, self.width
        for l, d_y in zip(self.outputs, d_outs):
            l.forward(x, n, d_y, ldx=ldx, d_m=d_n, ctx=ctx)

    def flops(self, n):
        return sum(l.flops(n) for l in self.hidden + self.outputs)

    def close(self):
        for l in self.hidden + self.outputs:
            l.close()
