"""Stage-2 dense heads on the device: early-fusion fully connected layers
(avod/core/avod_fc_layers/fusion_fc_layers.py:94-180 with fusion_method 'mean') and the
correlation-offsets head built from the same stack
(avod/builders/avod_corr_layers_builder.py:126-169)."""
import os

import numpy as np

from dodt_amd import ops


class EarlyFusionFcLayers(object):
    """mean of the input crops -> flatten -> fc6.. (ReLU) -> linear output layers
    (build_output_layers, fusion_fc_layers.py:94-133).  `outputs` names the output layers in
    params: ('cls_out', 'off_out', 'ang_out') for box_4ca / box_3d, whose ANG_VECS_OUTPUT_SIZE
    is 2 (avod_fc_layer_utils.py:11-17; the DODT config's avod_box_representation,
    pyramid_cars_with_aug_dt_5_tracking.config:29), ('cls_out', 'off_out') for box_4c / box_8c,
    ('off_out',) for the corr head."""

    K_ALIGN = 32     # the LDS-DMA GEMM walks K in stages of 32 (csrc/gemm.hip: fc_dma_kernel)

    def __init__(self, ctx, params, outputs=('cls_out', 'off_out', 'ang_out'), dtype='f32'):
        names = sorted(k for k in params if k.startswith('fc'))
        # The first layer's K is the flattened crop (7*7*32 = 1568 for the fusion head, 7*7*25 = 1225
        # for the correlation head).  A K that is no multiple of 32 would keep that GEMM on the
        # register-staged kernel: its weights get zero rows up to the next multiple and the crops are
        # written as rows of `in_ld` floats whose tail stays zero (FramePairPipeline allocates them
        # zeroed; the crop kernel never touches the pad) -- the same sums, + 0 * 0 terms.
        w0 = np.asarray(params[names[0]]['w'], np.float32)
        self.in_k = w0.shape[0]
        self.in_ld = -(-self.in_k // self.K_ALIGN) * self.K_ALIGN
        if self.in_ld != self.in_k:
            w0 = np.concatenate([w0, np.zeros((self.in_ld - self.in_k, w0.shape[1]), np.float32)], 0)
        self.hidden = [ops.FullyConnected(ctx, w0 if k == names[0] else params[k]['w'], params[k]['b'],
                                          True, dtype=dtype) for k in names]
        # for callers that hand over packed (n,h,w,c) crops of a K that is no multiple of 32: the first
        # layer with its own K (made on first use; it takes the register-staged kernel)
        self._packed_first = None
        self._first_params = (params[names[0]]['w'], params[names[0]]['b'], dtype)
        self.outputs = [ops.FullyConnected(ctx, params[k]['w'], params[k]['b'], False, dtype=dtype)
                        for k in outputs]
        # build_output_layers' layers all read fc_drop (fusion_fc_layers.py:94-133): one layer with the
        # weights side by side and one launch whose columns go to the separate outputs (fp32 heads)
        self.fused_out = None
        if len(outputs) > 1:
            w = np.concatenate([np.asarray(params[k]['w'], np.float32) for k in outputs], 1)
            b = np.concatenate([np.asarray(params[k]['b'], np.float32) for k in outputs])
            fused = ops.FullyConnected(ctx, w, b, False, dtype=dtype)
            if fused.can_split():
                self.fused_out = fused
            else:
                fused.close()
        self.width = max(l.N for l in self.hidden)
        self.ctx = ctx
        # bf16 heads keep their hidden activations as bf16 rows in HBM when every layer has that path
        # (csrc/gemm.hip: fc_bf16_dma_kernel; DODT_FC_BF16_ROWS=0: float32 activations, rounded on every load)
        self.bf16_rows = dtype == 'bf16' and os.environ.get('DODT_FC_BF16_ROWS', '1') != '0' and \
            all(l.bf16_row_elems() > 0 for l in self.hidden) and \
            (self.fused_out.bf16_row_elems() > 0 if self.fused_out is not None
             else all(l.bf16_row_elems() > 0 for l in self.outputs))
        self.in_ld16 = self.hidden[0].bf16_row_elems() if self.bf16_rows else 0

    def make_scratch(self, n_max):
        """Ping-pong hidden activations + the fused input rows; one set per concurrent stream."""
        s = [self.ctx.empty((n_max, self.width), np.float32) for _ in range(2)] + \
            [self.ctx.zeros((n_max, self.in_ld), np.float32)]
        if self.bf16_rows:      # [3], [4]: ping-pong bf16 activations, [5]: the first layer's bf16 rows
            s += [self.ctx.zeros((n_max, self.width), np.uint16) for _ in range(2)] + \
                 [self.ctx.zeros((n_max, self.in_ld16), np.uint16)]
        return s

    def forward(self, ctx, d_rois, d_rois2, n, d_n, d_outs, scratch):
        """d_rois: n rows of `in_ld` floats, the flattened (h,w,c) crop in front and zeros behind it
        (= (n,h,w,c) when in_ld == h*w*c) [and d_rois2, fused by mean]; *d_n rows are valid;
        d_outs: one (n, size) array per output layer."""
        x, x2 = d_rois, d_rois2
        ldx = self._row_floats(d_rois, n, 'd_rois')
        if x2 is not None and self._row_floats(x2, n, 'd_rois2') != ldx:
            raise ValueError('d_rois and d_rois2 must have the same row layout')
        if self.bf16_rows and len(scratch) > 5:
            # bf16((a + b) / 2) [or bf16(a)] as rows of in_ld16 elements with a zero tail, then bf16 rows all the way
            ops.rows_to_bf16(ctx, x, x2, n, d_n, self.in_k, ldx, scratch[5], self.in_ld16)
            x16, ld16 = scratch[5], self.in_ld16
            for i, l in enumerate(self.hidden):
                y16 = scratch[3 + (i & 1)]
                l.forward_bf16(x16, n, y16, ldx=ld16, ldy=self.width, d_m=d_n, y_bf16=True, ctx=ctx)
                x16, ld16 = y16, self.width
            if self.fused_out is not None:
                self.fused_out.forward_split_bf16(x16, n, d_outs, [l.N for l in self.outputs], ldx=ld16, d_m=d_n, ctx=ctx)
            else:
                for l, d_y in zip(self.outputs, d_outs):
                    l.forward_bf16(x16, n, d_y, ldx=ld16, d_m=d_n, y_bf16=False, ctx=ctx)
            return
        first = self.hidden[0]
        if ldx != self.in_ld:          # packed rows of in_k floats, in_k % 32 != 0
            if self._packed_first is None:
                w, b, dtype = self._first_params
                self._packed_first = ops.FullyConnected(self.ctx, w, b, True, dtype=dtype)
            first = self._packed_first
        if x2 is not None and len(scratch) > 2 and ldx == self.in_ld == self.in_k and self.in_k % 4 == 0:
            # the mean of the two crops as its own pass (avod_fc_layer_utils.py:38-41: (a + b) / 2,
            # the same float32 arithmetic): fc6 then runs as the plain GEMM, twice as fast as the
            # form that averages inside its K loop
            ops.mean_fusion(ctx, x, x2, n, d_n, self.in_ld, scratch[2])
            x, x2 = scratch[2], None
        for i, l in enumerate([first] + self.hidden[1:]):
            y = scratch[i & 1]
            l.forward(x, n, y, ldx=ldx, ldy=self.width, d_x2=x2, d_m=d_n, ctx=ctx)
            x, x2, ldx = y, None, self.width
        if self.fused_out is not None and ldx % 4 == 0:
            self.fused_out.forward_split(x, n, d_outs, [l.N for l in self.outputs], ldx=ldx, d_m=d_n, ctx=ctx)
            return
        for l, d_y in zip(self.outputs, d_outs):
            l.forward(x, n, d_y, ldx=ldx, d_m=d_n, ctx=ctx)

    def _row_floats(self, d, n, name):
        """Floats per row of a crop block: `in_ld` (zero-tailed rows, what make_scratch / the pipeline
        allocate) or `in_k` (packed (n,h,w,c) crops); anything else would be read at the wrong offsets
        and past its end, so it is an error.  The zero tail is the caller's: the crop kernel never writes it."""
        if d.dtype != np.float32:
            raise ValueError('%s must be float32' % name)
        size = int(np.prod(d.shape, dtype=np.int64))
        rows = d.shape[0] if len(d.shape) > 1 else 0
        if rows < n or rows == 0 or size % rows:
            raise ValueError('%s holds %s, fewer than %d rows' % (name, d.shape, n))
        per_row = size // rows
        if per_row not in (self.in_ld, self.in_k):
            raise ValueError('%s rows hold %d floats; the head takes %d (crop + zero tail) or %d (packed)'
                             % (name, per_row, self.in_ld, self.in_k))
        return per_row

    def flops(self, n):
        """Of the layers as the reference defines them (the zero rows of a padded K do not count)."""
        return sum(l.flops(n) for l in self.hidden + self.outputs) \
            - 2.0 * n * (self.in_ld - self.in_k) * self.hidden[0].N

    def close(self):
        for l in self.hidden + self.outputs + [self.fused_out, self._packed_first]:
            if l is not None:
                l.close()
