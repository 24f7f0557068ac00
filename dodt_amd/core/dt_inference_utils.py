"""Detection records -> KITTI label rows on the host, vectorised.

Mirrors avod/core/dt_inference_utils.py:135-215 (convert_pred_to_kitti_format) and the
projection it calls, avod/core/box_3d_projector.py:86-159.  The reference reads the
calibration and the image size from its dataset object; here they are arguments, because the
dataset layer is out of scope (SURVEY 2 "OOS").  Input: the 17-column records the device
writes (dodt_pack_detections / FramePairPipeline.d_records)."""
import numpy as np


def project_boxes_to_image_space(boxes_3d, calib_p2, truncate=False, image_size=None,
                                 discard_before_truncation=True):
    """boxes_3d (n,7) [x,y,z,l,w,h,ry] -> ((n,4) [x1,y1,x2,y2], valid (n,) bool).
    Row-wise the result of box_3d_projector.project_to_image_space; rows it returns None for
    have valid False."""
    b = np.asarray(boxes_3d, dtype=np.float64).reshape(-1, 7)
    p = np.asarray(calib_p2, dtype=np.float64)
    l, w, h, ry = b[:, 3], b[:, 4], b[:, 5], b[:, 6]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) * 0.5
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1]) * 0.5
    xc, zc = l[:, None] * sx, w[:, None] * sz                       # (n,8)
    yc = np.concatenate([np.zeros((len(b), 4)), -np.repeat(h[:, None], 4, 1)], 1)
    c, s = np.cos(ry)[:, None], np.sin(ry)[:, None]
    X = c * xc + s * zc + b[:, 0:1]
    Y = yc + b[:, 1:2]
    Z = -s * xc + c * zc + b[:, 2:3]
    u = p[0, 0] * X + p[0, 1] * Y + p[0, 2] * Z + p[0, 3]
    v = p[1, 0] * X + p[1, 1] * Y + p[1, 2] * Z + p[1, 3]
    q = p[2, 0] * X + p[2, 1] * Y + p[2, 2] * Z + p[2, 3]
    u, v = u / q, v / q
    box = np.stack([u.min(1), v.min(1), u.max(1), v.max(1)], 1)
    valid = np.ones(len(b), dtype=bool)
    if not truncate:
        return box, valid
    if not image_size:
        raise ValueError('Image size must be provided')
    iw, ih = float(image_size[0]), float(image_size[1])
    valid &= ~((box[:, 0] > iw) | (box[:, 1] > ih) | (box[:, 2] < 0) | (box[:, 3] < 0))
    bw, bh = box[:, 2] - box[:, 0], box[:, 3] - box[:, 1]
    if discard_before_truncation:
        valid &= ~((bw > iw * 0.8) | (bh > ih * 0.8))
    box = np.stack([np.maximum(box[:, 0], 0), np.maximum(box[:, 1], 0),
                    np.minimum(box[:, 2], iw), np.minimum(box[:, 3], ih)], 1)
    if not discard_before_truncation:
        bw, bh = box[:, 2] - box[:, 0], box[:, 3] - box[:, 1]
        valid &= ~((bw > iw * 0.8) & (bh > ih * 0.8))
    return box, valid


def kitti_label_table(all_predictions, stereo_calib_p2, image_size, score_threshold):
    """The numbers of convert_pred_to_kitti_format: (class indices (k,), table (k,16) float64 rounded to 3
    decimals, columns [0, 0, 0, -10, x1, y1, x2, y2, h, w, l, x, y, z, ry, score]) of the records that
    pass the score threshold and project into the image; (None, None) when nothing survives."""
    p = np.asarray(all_predictions, dtype=np.float64)
    p = p[p[:, 7] >= score_threshold]
    if len(p) == 0:
        return None, None
    boxes, valid = project_boxes_to_image_space(p[:, 0:7], stereo_calib_p2, truncate=True,
                                                image_size=image_size)
    p, boxes = p[valid], boxes[valid]
    if len(boxes) == 0:
        return None, None
    k = np.zeros([len(boxes), 16])
    k[:, 3] = -10
    k[:, 4:8] = boxes
    k[:, 8], k[:, 9], k[:, 10] = p[:, 5], p[:, 4], p[:, 3]
    k[:, 11:14] = p[:, 0:3]
    k[:, 14:16] = p[:, 6:8]
    return p[:, 8].astype(np.int32), np.round(k, 3)


def convert_pred_to_kitti_format(all_predictions, stereo_calib_p2, image_size, classes,
                                 score_threshold):
    """(n,17) records -> the reference's text table: rows [type, -1, -1, -10, x1, y1, x2, y2,
    h, w, l, x, y, z, ry, score] as strings (np.column_stack of a str column and floats, as
    the reference builds it), or [] when nothing survives."""
    types, k = kitti_label_table(all_predictions, stereo_calib_p2, image_size, score_threshold)
    if k is None:
        return []
    obj_types = [classes[i] for i in types]
    empty = -1 * np.ones((len(k), 2), dtype=np.int32)
    return np.column_stack([obj_types, empty, k[:, 3:16]])
