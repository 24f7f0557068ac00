"""The two tf.image ops the hot path calls, on the GPU, with TF's signatures:
tf.image.crop_and_resize (call sites avod/core/models/dt_rpn_model.py:418-428,
dt_avod_model.py:253-273) and tf.image.non_max_suppression
(dt_rpn_model.py:587-591, dt_avod_model.py:606-613)."""
import numpy as np

from dodt_amd import device, ops


def crop_and_resize(image, boxes, box_ind, crop_size, ctx=None):
    """image (1,H,W,C) float32; boxes (n,4) [y1,x1,y2,x2] normalised; box_ind
    (n,) all zero (one image per call, as in the reference); -> (n,ch,cw,C)."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    if img.ndim != 4 or img.shape[0] != 1:
        raise ValueError('image must be (1, H, W, C)')
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    if b.ndim != 2 or b.shape[1] != 4:
        raise ValueError('boxes must be (n, 4)')
    if box_ind is not None and np.any(np.asarray(box_ind) != 0):
        raise ValueError('box_ind must be all zeros (single image)')
    ctx = ctx or device.default_context()
    n = len(b)
    ch, cw = int(crop_size[0]), int(crop_size[1])
    _, h, w, c = img.shape
    if n == 0:
        return np.zeros((0, ch, cw, c), np.float32)
    d_out = ctx.empty((n, ch, cw, c), np.float32)
    ops.crop_and_resize(ctx, ctx.array(img), (h, w, c), ctx.array(b), n, None,
                        (ch, cw), d_out)
    return d_out.download()


def non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5, ctx=None):
    """-> int32 indices of the selected boxes, in descending score order."""
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    if b.ndim != 2 or b.shape[1] != 4:
        raise ValueError('boxes must be 2-D with 4 columns')
    if s.shape != (len(b),):
        raise ValueError('scores has incompatible shape')
    if not 0 <= iou_threshold <= 1:
        raise ValueError('iou_threshold must be in [0, 1]')
    ctx = ctx or device.default_context()
    n = len(b)
    k = int(min(max_output_size, n))
    if k <= 0:
        return np.zeros((0,), np.int32)
    d_sel = ctx.empty((k,), np.int32)
    d_cnt = ctx.zeros((1,), np.int32)
    ops.nms(ctx, ctx.array(b), ctx.array(s), n, None, k, iou_threshold, d_sel, d_cnt)
    cnt = int(d_cnt.download()[0])
    return d_sel.download()[:cnt]
