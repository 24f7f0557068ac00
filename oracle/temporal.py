"""Oracle for the temporal module (the "M" of S+T+M): association of a
keyframe pair's detections and interpolation of the frames between them.  TEST
INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/temporal.npz, produced
by the reference's own code (tests/golden/make_goldens_temporal.py).

Restated from:
  avod/core/dt_evaluator_utils.py:212-294   interpolate_non_keyframe_predicitons
  avod/core/dt_evaluator_utils.py:296-362   interpolate_trajectory
  wavedata/wavedata/tools/obj_detection/evaluation.py:43-262  three_d_iou and helpers
    (the base intersection is rasterised at 1 cm with PIL, as the reference does)
`recover(frame_index, rows)` stands for recovery_coordinate (:192-210), whose OXTS inputs
belong to the dataset layer (out of scope); None leaves the rows in keyframe-0 coordinates.
"""
import copy

import numpy as np


def _corners(b):
    """[ry,l,h,w,tx,ty,tz] -> x(4), z(4) of the base rectangle (evaluation.py:144-178)."""
    rot = np.array([[np.cos(b[0]), np.sin(b[0])], [-np.sin(b[0]), np.cos(b[0])]])
    xc = np.multiply(b[1] / 2, np.array([1, 1, -1, -1]))
    zc = np.multiply(b[3] / 2, np.array([1, -1, -1, 1]))
    t = np.dot(rot, np.array([xc, zc]))
    return t[0] + b[4], t[1] + b[6]


def _base_intersection(box, other, res=0.01):
    from PIL import Image, ImageDraw
    xb, zb = _corners(box)
    xi, zi = _corners(other)
    if xb.max() < xi.min() or xi.max() < xb.min() or zb.max() < zi.min() or zi.max() < zb.min():
        return 0.0
    xa, za = np.append(xb, xi), np.append(zb, zi)
    maxs = np.array([xa.max(), za.max()])
    mins = np.array([xa.min(), za.min()])
    dims = np.int32(np.ceil((maxs - mins) / res))
    masks = []
    for x, z in ((xb, zb), (xi, zi)):
        img = Image.new('L', (int(dims[0]), int(dims[1])), 0)
        draw = ImageDraw.Draw(img, 'L')
        co = np.reshape(np.transpose(np.array([(x - mins[0]) / res, (z - mins[1]) / res])), 8)
        co = np.append(co, co[0:2])
        draw.polygon(co.ravel().tolist(), outline=255, fill=255)
        del draw
        masks.append(np.asarray(img))
    inter = np.logical_and(masks[0], masks[1])
    return min(100, np.size(np.flatnonzero(inter)) * np.square(res))


def three_d_iou(box, boxes):
    """box (7,), boxes (n,7) in [ry,l,h,w,tx,ty,tz] -> (n,) IoU (evaluation.py:43-92)."""
    boxes = np.atleast_2d(np.asarray(boxes, dtype=np.float64))
    box = np.asarray(box, dtype=np.float64)
    diag = np.sqrt(box[1] ** 2 + box[2] ** 2 + box[3] ** 2) / 2
    diags = np.sqrt(boxes[:, 1] ** 2 + boxes[:, 2] ** 2 + boxes[:, 3] ** 2) / 2
    dist = np.sqrt(((boxes[:, 4:7] - box[4:7]) ** 2).sum(1))
    iou = np.zeros(len(boxes))
    for i in np.nonzero(diag + diags >= dist)[0]:
        o = boxes[i]
        h_int = max(0.0, min(box[5], o[5]) - max(box[5] - box[2], o[5] - o[2]))
        inter = h_int * _base_intersection(box, o)
        iou[i] = inter / (np.prod(box[1:4]) + np.prod(o[1:4]) - inter)
    return iou


def _interpolate_trajectory(trajectories, num):
    dense = []
    for track in trajectories:
        new = []
        a, b = track
        if a is not None and b is not None:
            a, b = a[:-4], b[:-4]
            new.append(a)
            offsets = b[[0, 2, 6]] - a[[0, 2, 6]]
            score = max(a[7], b[7])
            for i in range(num - 2):
                o = copy.deepcopy(a)
                o[[0, 2, 6]] += offsets * (i + 1.0) / (num - 1)
                o[7] = score
                new.append(o)
            b[7] = score
            new.append(b)
        elif a is None:
            offsets = b[-4:-1]
            b = b[:-4]
            d = np.sqrt(offsets[0] ** 2 + offsets[1] ** 2)
            if d <= b[4] / 2:
                dx, dz = d * np.cos(b[6]), d * np.sin(b[6])
                for i in range(num - 1):
                    o = copy.deepcopy(b)
                    o[0] -= dx * (num - i - 2) / (num - 1)
                    o[2] -= dz * (num - i - 2) / (num - 1)
                    new.append(o)
                new.append(b)
            else:
                for i in range(num - 1):
                    new.append(None if i <= num / 2 else copy.deepcopy(b))
                new.append(b)
        else:
            offsets = a[-4:-1]
            a = a[:-4]
            d = np.sqrt(offsets[0] ** 2 + offsets[1] ** 2)
            if d <= a[4] / 2:
                dx, dz = d * np.cos(a[6]), d * np.sin(a[6])
                new.append(a)
                for i in range(num - 1):
                    o = copy.deepcopy(a)
                    o[0] += dx * (i + 1.0) / (num - 1)
                    o[2] += dz * (i + 1.0) / (num - 1)
                    new.append(o)
            else:
                new.append(a)
                for i in range(num - 1):
                    new.append(None if i >= num / 2 else copy.deepcopy(a))
        dense.append(new)
    return dense


def interpolate_non_keyframe_predictions(predictions, n_frames, threshold, recover=None):
    """predictions (n,17) of one keyframe pair (frame mark in the last column), n_frames =
    tau + 1 frames from keyframe 0 to keyframe 1 (1: a lone frame).  Returns a list of
    n_frames arrays (k,13): box_3d(7), score, type, first four columns of the shifted box."""
    p = np.asarray(predictions, dtype=np.float64)
    rec = recover or (lambda i, rows: rows)
    lists = [p[p[:, -1] == i] for i in range(min(n_frames, 2))]
    if n_frames == 1:
        return [lists[0][lists[0][:, 7] > threshold][:, :-4]]
    kept = [q[q[:, 7] > threshold] for q in lists]
    if n_frames == 2:
        out = [q[:, :-4] for q in kept]
        out[1] = rec(1, out[1])
        return out
    tracks = []
    if len(kept[0]) == 0:
        if len(kept[1]) == 0:
            return [np.zeros((0, 13)) for _ in range(n_frames)]
        for o in kept[1]:
            tracks.append([None, o])
    else:
        free = list(range(len(kept[1])))
        for cur in kept[0]:
            t = [cur]
            if not free:
                t.append(None)
            else:
                # (the reference scores against ALL frame-1 boxes, matched ones included)
                ious = three_d_iou(cur[[6, 3, 5, 4, 0, 1, 2]], kept[1][:, [6, 3, 5, 4, 0, 1, 2]])
                best = int(np.argmax(ious))
                if ious[best] > 0:
                    t.append(kept[1][best])
                    free.remove(best)
                else:
                    t.append(None)
            tracks.append(t)
        for j in free:
            tracks.append([None, kept[1][j]])
    dense = _interpolate_trajectory(tracks, n_frames)
    out = [[] for _ in range(n_frames)]
    for t in dense:
        for i in range(n_frames):
            if t[i] is not None:
                out[i].append(t[i])
    out = [np.asarray(o, dtype=np.float64).reshape(-1, 13) for o in out]
    for i in range(1, n_frames):
        out[i] = rec(i, out[i])
    return out
