"""Stage-2 box decode on the GPU: the composition the graph applies between the
FC head and NMS #2 (avod/core/models/dt_avod_model.py:464-469,575-603), built
from box_3d_encoder.anchors_to_box_3d(fix_lw=True) (:230-322),
box_4c_encoder.tf_box_3d_to_box_4c (:85-165), tf_offsets_to_box_4c (:474-484),
tf_box_4c_to_box_3d (:369-458) and box_3d_encoder.tf_box_3d_to_anchor (:188-227)."""
import numpy as np

from dodt_amd import device, ops


def decode_box_4c_predictions(top_anchors, offsets_4c, ground_plane, bev_extents,
                              ctx=None):
    """-> (prediction_boxes_3d (N,7), prediction_anchors (N,6),
           avod_bev_boxes_tf_order (N,4) metres [z1,x1,z2,x2])."""
    a = np.asarray(top_anchors, dtype=np.float32)
    t = np.asarray(offsets_4c, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 6 or t.shape != (len(a), 10):
        raise TypeError('Invalid box_4c format')
    ctx = ctx or device.default_context()
    n = len(a)
    e = np.asarray(bev_extents, dtype=np.float64)
    d3, d6, d4 = ctx.empty((n, 7)), ctx.empty((n, 6)), ctx.empty((n, 4))
    ops.box_4c_decode(ctx, ctx.array(a), ctx.array(t), n, None, ground_plane,
                      [e[0][0], e[0][1], e[1][0], e[1][1]], d3, d6, d4)
    return d3.download(), d6.download(), d4.download()
