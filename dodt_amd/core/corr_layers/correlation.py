"""correlation(input_a, input_b, ...) on the GPU -- same signature as
avod/core/corr_layers/correlation.py:7-27 (the reference's tf.load_op_library wrapper)."""
import numpy as np

from dodt_amd import device, ops


def correlation(input_a, input_b, kernel_size=1, max_displacement=20, stride_1=1, stride_2=2,
                padding=20, ctx=None):
    """(batch, H, W, C) x2 -> (batch, out_h, out_w, (2*(max_displacement//stride_2)+1)**2)."""
    a = np.ascontiguousarray(input_a, dtype=np.float32)
    b = np.ascontiguousarray(input_b, dtype=np.float32)
    if a.ndim != 4 or a.shape != b.shape:
        raise ValueError('input_a and input_b must be (batch, H, W, C) of equal shape')
    if kernel_size != 1 or stride_1 != 1:
        raise NotImplementedError('DODT uses kernel_size=1, stride_1=1')
    ctx = ctx or device.default_context()
    n, h, w, c = a.shape
    r = max_displacement // stride_2
    oh = h + 2 * padding - 2 * max_displacement
    ow = w + 2 * padding - 2 * max_displacement
    out = np.empty((n, oh, ow, (2 * r + 1) ** 2), np.float32)
    d_out = ctx.empty(out.shape[1:], np.float32)
    for i in range(n):
        ops.correlation(ctx, ctx.array(a[i]), ctx.array(b[i]), (h, w, c), max_displacement,
                        stride_2, padding, d_out)
        out[i] = d_out.download()
    return out
