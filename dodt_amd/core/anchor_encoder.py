"""avod/core/anchor_encoder.py:99-150 offset_to_anchor on the GPU (float32)."""
import numpy as np

from dodt_amd import device, ops


def offset_to_anchor(anchors, offsets, ctx=None):
    a = np.asarray(anchors, dtype=np.float32)
    t = np.asarray(offsets, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 6 or t.shape != a.shape:
        raise TypeError('Invalid anchor format')
    ctx = ctx or device.default_context()
    n = len(a)
    d_out = ctx.empty((n, 6), np.float32)
    ops.offset_to_anchor(ctx, ctx.array(a), ctx.array(t), n, None, d_out)
    return d_out.download()
