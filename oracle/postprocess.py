"""Oracle for the detection post-processing: image-space projection of
a box_3d with truncation and the KITTI label rows.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).  Pinned by tests/golden/kitti_format.npz, which the reference's own
box_3d_projector.project_to_image_space produced (tests/golden/make_goldens_kitti.py).

Restated from:
  wavedata/wavedata/tools/obj_detection/obj_utils.py:315-345   compute_box_corners_3d
  wavedata/wavedata/tools/core/calib_utils.py:394-410          project_to_image
  avod/core/box_3d_projector.py:86-159                         project_to_image_space
  avod/core/dt_inference_utils.py:135-215                      convert_pred_to_kitti_format
  avod/core/dt_evaluator.py:1134-1259                          get_avod_predicted_boxes_3d_and_scores
                                                               (pinned by tests/golden/box4ca.npz,
                                                               tests/golden/make_goldens_box4ca.py)
"""
import numpy as np

from oracle import points as opoints


def box_corners_3d(box_3d):
    """(3,8) corners of [x,y,z,l,w,h,ry] (bottom centre), rotation about y."""
    x, y, z, l, w, h, ry = [float(v) for v in box_3d]
    rot = np.array([[+np.cos(ry), 0, +np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, +np.cos(ry)]])
    xc = np.array([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2])
    yc = np.array([0, 0, 0, 0, -h, -h, -h, -h])
    zc = np.array([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2])
    c = np.dot(rot, np.array([xc, yc, zc]))
    c[0] += x
    c[1] += y
    c[2] += z
    return c


def project_box_to_image_space(box_3d, p2, truncate=False, image_size=None,
                               discard_before_truncation=True):
    """[x1,y1,x2,y2] of the projected corners, or None (box_3d_projector.py:86-159)."""
    uv = opoints.project_to_image(box_corners_3d(box_3d), np.asarray(p2, np.float64))
    box = np.array([uv[0].min(), uv[1].min(), uv[0].max(), uv[1].max()])
    if not truncate:
        return box
    if not image_size:
        raise ValueError('Image size must be provided')
    iw, ih = image_size[0], image_size[1]
    if box[0] > iw or box[1] > ih or box[2] < 0 or box[3] < 0:
        return None
    if discard_before_truncation:
        if box[2] - box[0] > iw * 0.8 or box[3] - box[1] > ih * 0.8:
            return None
    box[0] = max(box[0], 0)
    box[1] = max(box[1], 0)
    box[2] = min(box[2], iw)
    box[3] = min(box[3], ih)
    if not discard_before_truncation:
        if box[2] - box[0] > iw * 0.8 and box[3] - box[1] > ih * 0.8:
            return None
    return box


def convert_pred_to_kitti_format(all_predictions, p2, image_size, classes, score_threshold):
    """dt_inference_utils.py:135-215 with the calibration and the image size passed in
    (the reference reads them from the dataset).  Returns (types, (n,15) float array):
    [trunc -1, occl -1, alpha -10, x1,y1,x2,y2, h,w,l, x,y,z, ry, score], 3 decimals."""
    p = np.asarray(all_predictions, dtype=np.float64)
    p = p[p[:, 7] >= score_threshold]
    rows, types = [], []
    for r in p:
        box = project_box_to_image_space(r[0:7], p2, truncate=True, image_size=image_size)
        if box is None:
            continue
        k = np.zeros(16)
        k[3] = -10
        k[4:8] = box
        k[8], k[9], k[10] = r[5], r[4], r[3]
        k[11:14] = r[0:3]
        k[14:16] = r[6:8]
        k = np.round(k, 3)
        rows.append(np.concatenate([[-1, -1], k[3:16]]))
        types.append(classes[int(r[8])])
    if not rows:
        return [], np.zeros((0, 15))
    return types, np.asarray(rows)


def orientation_corrected_boxes_3d(boxes_3d, orientations):
    """dt_evaluator.py:1166-1212, box_rep 'box_4ca': the box decoded from its four corners
    knows its heading only modulo 90 degrees (its long side may have become the short one)
    and modulo 180; the angle the network regressed (orientations) decides.  float32
    arithmetic like the arrays the evaluator receives from sess.run; python-float constants
    take part as float32 (numpy scalar-with-array casting).  ry - pi/2 below -pi is left
    unwrapped, as in the reference."""
    f32 = np.float32
    b = np.array(boxes_3d, dtype=f32, copy=True)
    ori = np.asarray(orientations, dtype=f32)
    two_pi, pi = f32(2 * np.pi), f32(np.pi)
    diff = b[:, 6] - ori
    diff = np.where(diff < -pi, diff + two_pi, diff).astype(f32)
    diff = np.where(diff > pi, diff - two_pi, diff).astype(f32)
    q1, q2, q3 = f32(0.25 * np.pi), f32(0.50 * np.pi), f32(0.75 * np.pi)
    pos = (q1 < diff) & (diff < q3)
    neg = (-q1 > diff) & (diff > -q3)
    swap = pos | neg
    b[swap, 3], b[swap, 4] = b[swap, 4].copy(), b[swap, 3].copy()
    b[pos, 6] = b[pos, 6] + q2
    b[neg, 6] = b[neg, 6] - q2
    flip = np.abs(diff) >= q3
    b[flip, 6] = b[flip, 6] + pi
    above = b[:, 6] > pi
    b[above, 6] = b[above, 6] - two_pi
    return b


def avod_predicted_boxes_3d_and_scores(boxes_3d, orientations, softmax, corr_offsets):
    """dt_evaluator.py:1134-1259 for one frame pair, box_rep 'box_4ca' (orientations not None)
    or 'box_4c' (None).  Lists of two per-frame arrays; corr_offsets (n0,3) of frame 0.
    -> (n0 + n1, 17): box_3d(7), score, type, frame-0 box shifted by the correlation offsets
    (zeros for frame 1), frame mark."""
    boxes = [np.asarray(b, np.float32) if orientations is None
             else orientation_corrected_boxes_3d(b, orientations[i])
             for i, b in enumerate(boxes_3d)]
    shifted = boxes[0].copy()
    off = np.asarray(corr_offsets, np.float32)
    shifted[:, 0] += off[:, 0]
    shifted[:, 2] += off[:, 1]
    shifted[:, 6] += off[:, 2]
    corr = [shifted, np.zeros((len(boxes[1]), 7))]
    rows = []
    for i in range(2):
        fg = np.asarray(softmax[i])[:, 1:]
        types = np.argmax(fg, axis=1)
        scores = fg[np.arange(len(fg)), types]
        rows.append(np.column_stack([boxes[i], scores, types, corr[i],
                                     np.ones((len(boxes[i]), 1)) * i]))
    return np.concatenate(rows, axis=0)
