"""Anchor projection on the GPU -- same function names and return values as
avod/core/anchor_projector.py (numpy branch: float64 in, :13-156; the float32
`tf_` twin :159-251)."""
import numpy as np

from dodt_amd import device, ops


def _flat_extents(bev_extents):
    e = np.asarray(bev_extents, dtype=np.float64)
    return [e[0][0], e[0][1], e[1][0], e[1][1]]


def project_to_bev(anchors, bev_extents, ctx=None):
    """-> (bev_box_corners, bev_box_corners_norm), each N x [x1, z1, x2, z2]."""
    a = np.asarray(anchors)
    ctx = ctx or device.default_context()
    n = len(a)
    if n == 0:
        return np.zeros((0, 4), a.dtype), np.zeros((0, 4), a.dtype)
    ext = _flat_extents(bev_extents)
    d_a = ctx.array(a.astype(np.float32))
    d_bev = ctx.empty((n, 4), np.float32)
    d_norm = ctx.empty((n, 4), np.float32)
    ops.project_anchors_f32(ctx, d_a, n, None, ext, np.zeros(12), (1, 1),
                            d_bev=d_bev, d_bev_norm_tf=d_norm)
    return d_bev.download(), d_norm.download()[:, [1, 0, 3, 2]]


def project_to_image_space(anchors, stereo_calib_p2, image_shape, ctx=None):
    """numpy-branch twin: float64 anchors -> float32 (box_corners,
    box_corners_norm) in [x1, y1, x2, y2]."""
    a = np.asarray(anchors, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 6:
        raise ValueError("Invalid shape for anchors {}, should be "
                         "(N, 6)".format(a.shape[-1]))
    ctx = ctx or device.default_context()
    n = len(a)
    h, w = float(image_shape[0]), float(image_shape[1])
    d_a = ctx.array(a)
    d_img = ctx.empty((n, 4), np.float32)
    ops.project_anchors_f64(ctx, d_a, None, n, None, [0, 1, 0, 1],
                            stereo_calib_p2, (w, h), d_img_norm=d_img)
    norm = d_img.download()[:, [1, 0, 3, 2]]
    # pixel corners: the same projection normalised by a unit image
    ops.project_anchors_f64(ctx, d_a, None, n, None, [0, 1, 0, 1],
                            stereo_calib_p2, (1.0, 1.0), d_img_norm=d_img)
    return d_img.download()[:, [1, 0, 3, 2]], norm


def tf_project_to_image_space(anchors, stereo_calib_p2, image_shape, ctx=None):
    """float32 twin (anchor_projector.py:159-251)."""
    a = np.asarray(anchors, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 6:
        raise ValueError("Invalid shape for anchors {}, should be "
                         "(N, 6)".format(a.shape[-1]))
    ctx = ctx or device.default_context()
    n = len(a)
    h, w = float(image_shape[0]), float(image_shape[1])
    d_a = ctx.array(a)
    d_img = ctx.empty((n, 4), np.float32)
    ops.project_anchors_f32(ctx, d_a, n, None, [0, 1, 0, 1], stereo_calib_p2,
                            (w, h), d_img_norm_tf=d_img)
    norm = d_img.download()[:, [1, 0, 3, 2]]
    ops.project_anchors_f32(ctx, d_a, n, None, [0, 1, 0, 1], stereo_calib_p2,
                            (1.0, 1.0), d_img_norm_tf=d_img)
    return d_img.download()[:, [1, 0, 3, 2]], norm


def reorder_projected_boxes(box_corners):
    """[x1, y1, x2, y2] -> [y1, x1, y2, x2]."""
    b = np.asarray(box_corners)
    return b[:, [1, 0, 3, 2]]
