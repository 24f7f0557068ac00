"""avod/core/orientation_encoder.py:20-34 tf_angle_vector_to_orientation on the GPU, and the
evaluator's use of it for box_4ca (avod/core/dt_evaluator.py:1166-1212)."""
import numpy as np

from dodt_amd import device, ops


def tf_angle_vector_to_orientation(angle_vectors, ctx=None):
    """(N,2) angle vectors [x, y] -> (N,) orientation angles atan2(y, x), float32."""
    v = np.asarray(angle_vectors, dtype=np.float32)
    if v.ndim != 2 or v.shape[1] != 2:
        raise ValueError('angle vectors must be (N, 2)')
    ctx = ctx or device.default_context()
    n = len(v)
    d_out = ctx.empty((n,), np.float32)
    if n:
        ops.angle_vector_to_orientation(ctx, ctx.array(v), n, None, d_out)
    return d_out.download()


def predicted_boxes_3d_and_scores(boxes_3d, scores, orientations=None, corr_offsets=None,
                                  frame_mark=0, ctx=None):
    """One frame's share of DtEvaluator.get_avod_predicted_boxes_3d_and_scores
    (dt_evaluator.py:1134-1259): (n,7) box_3d rows after NMS #2, their scores, the regressed
    orientations (box_4ca; None for box_4c) and, for frame 0 of a pair, the correlation offsets
    -> (n,17) float32 records."""
    b = np.asarray(boxes_3d, dtype=np.float32)
    n = len(b)
    if b.ndim != 2 or b.shape[1] != 7:
        raise ValueError('boxes_3d must be (N, 7)')
    ctx = ctx or device.default_context()
    if n == 0:
        return np.zeros((0, 17), np.float32)
    d_rec, d_cnt = ctx.empty((n, 17), np.float32), ctx.empty((1,), np.int32)
    sel = ctx.array(np.arange(n, dtype=np.int32))
    cnt = ctx.array(np.array([n], np.int32))
    ops.pack_detections(
        ctx, ctx.array(b), ctx.array(np.asarray(scores, np.float32)), sel, cnt, n,
        float(frame_mark), d_rec, d_cnt,
        d_corr_offsets=None if corr_offsets is None
        else ctx.array(np.asarray(corr_offsets, np.float32)),
        d_orientations=None if orientations is None
        else ctx.array(np.asarray(orientations, np.float32)))
    return d_rec.download()
