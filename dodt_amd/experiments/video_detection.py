"""Video-level association of per-frame detections on the host (the MoI
variant): stands where avod/experiments/video_detection.py stands for

  iou_2d                  :70-90    BEV IoU of two boxes scaled by 3.8, axis-aligned hulls
  cal_transformed_ious    :109-126  the later box registered into the earlier frame first
  track_iou               :235-277  greedy IoU tracker over the frames of a video
  label_interpolation     :371-413  fill the frames between keyframes of a stride
  cal_label               :415-440

numpy, float64.  The reference looks ego-motion and calibration up in its dataset object;
here the caller passes `ego(frame_a, frame_b) -> (trans, matrix, delta)` (e.g. built on
dodt_amd.datasets.kitti.kitti_tracking_utils.coordinate_transform) and the calibration.
"""
from copy import deepcopy

import numpy as np

from dodt_amd.datasets.kitti import kitti_tracking_utils as ktu

PLANE = np.asarray([0, -1, 0, 1.65])


def _bev_hull(box3d_kitti, scale=3.8):
    """[min x, max z, max x, min z] of the ortho-aligned BEV corners of a KITTI-ordered box
    [l, w, h, x, y, z, ry] with its dimensions scaled (np_box_3d_to_box_4c,
    avod/core/box_4c_encoder.py:18-82, of the box_3d [x,y,z,l,w,h,ry])."""
    l, w, h, x, y, z, ry = [float(v) for v in box3d_kitti]
    l, w = scale * l, scale * w
    half_pi = np.pi / 2
    ortho_ry = np.round(ry / half_pi) * half_pi
    # box_3d_to_anchor(ortho_rotate=True): dimensions swap with |cos|, |sin| of the snapped angle
    dim_x = l * np.abs(np.cos(ortho_ry)) + w * np.abs(np.sin(ortho_ry))
    dim_z = w * np.abs(np.cos(ortho_ry)) + l * np.abs(np.sin(ortho_ry))
    hx, hz = dim_x / 2, dim_z / 2
    xc = np.asarray([hx, hx, -hx, -hx])
    zc = np.array([hz, -hz, -hz, hz])
    d = ry - ortho_ry
    tr = np.array([[np.cos(d), np.sin(d), x], [-np.sin(d), np.cos(d), z], [0, 0, 1]])
    c = np.matmul(tr, np.vstack([xc, zc, np.ones(4)]))[0:2]
    return np.asarray([np.min(c[0]), np.max(c[1]), np.max(c[0]), np.min(c[1])])


def two_d_iou(box, boxes):
    """wavedata/.../obj_detection/evaluation.py:6-41 (rounded to 3 decimals)."""
    boxes = np.atleast_2d(np.asarray(boxes, np.float64))
    iou = np.zeros(len(boxes), np.float64)
    w_int = np.minimum(box[2], boxes[:, 2]) - np.maximum(box[0], boxes[:, 0])
    h_int = np.minimum(box[3], boxes[:, 3]) - np.maximum(box[1], boxes[:, 1])
    ne = np.logical_and(w_int > 0, h_int > 0)
    if ne.any():
        inter = w_int[ne] * h_int[ne]
        union = (box[2] - box[0]) * (box[3] - box[1]) + \
            (boxes[ne, 2] - boxes[ne, 0]) * (boxes[ne, 3] - boxes[ne, 1]) - inter
        iou[ne] = inter / union
    return iou.round(3)


def iou_2d(box3d_1, box3d_2):
    """video_detection.py:70-90: boxes [l, w, h, x, y, z, ry].  The hulls are
    [min x, max z, max x, min z]: with z1 > z2 two_d_iou's height is negative and the IoU 0,
    exactly as the reference computes it."""
    return float(two_d_iou(_bev_hull(box3d_1), _bev_hull(box3d_2)[None])[0])


def label_transform_box(box3d_kitti, r0_rect, tr_velo_to_cam, trans, matrix, delta):
    """KittiTrackingDataset.label_transform for one box (kitti_tracking_dataset.py:338-372):
    the later frame's box [l, w, h, x, y, z, ry] in the earlier frame's coordinates."""
    l, w, h, x, y, z, ry = [float(v) for v in box3d_kitti]
    corners = ktu.box_corners_3d([x, y, z, l, w, h, ry])
    velo = ktu._rect_to_velo(corners, r0_rect, tr_velo_to_cam)
    velo = (velo + trans) @ matrix
    rect = ktu._velo_to_rect(velo, r0_rect, tr_velo_to_cam)
    t = np.mean(rect, axis=0)
    t[1] += h / 2.0
    return np.asarray([l, w, h, t[0], t[1], t[2], ry + delta])


def cal_transformed_ious(ego, calib, item1, item2):
    """video_detection.py:109-126.  ego(frame_id_1, frame_id_2) -> (trans, matrix, delta);
    calib = (r0_rect, tr_velo_to_cam)."""
    trans, matrix, delta = ego(item1['frame_id'], item2['frame_id'])
    moved = label_transform_box(item2['boxes3d'], calib[0], calib[1], trans, matrix, delta)
    return iou_2d(np.asarray(item1['boxes3d'], np.float64), moved)


def track_iou(ego, calib, detections, sigma_l, sigma_h, sigma_iou, t_min):
    """video_detection.py:235-277: detections[k] = list of dicts ('frame_id', 'boxes3d'
    [l,w,h,x,y,z,ry], 'scores', ...) of frame k ([] = nothing)."""
    tracks_active, tracks_finished = [], []
    for detections_frame in detections:
        if len(detections_frame) == 0:
            continue
        dets = [det for det in detections_frame if det['scores'] >= sigma_l]
        updated = []
        for track in tracks_active:
            if len(dets) > 0:
                ious = [cal_transformed_ious(ego, calib, track['trajectory'][-1], x) for x in dets]
                best = int(np.argmax(ious))
                if ious[best] > sigma_iou:
                    track['trajectory'].append(dets[best])
                    track['max_score'] = max(track['max_score'], dets[best]['scores'])
                    updated.append(track)
                    del dets[best]
            if len(updated) == 0 or track is not updated[-1]:
                if track['max_score'] >= sigma_h and len(track['trajectory']) >= t_min:
                    tracks_finished.append(track)
        tracks_active = updated + [{'trajectory': [det], 'max_score': det['scores'],
                                    'start_frame': det['frame_id']} for det in dets]
    tracks_finished += [t for t in tracks_active
                        if t['max_score'] >= sigma_h and len(t['trajectory']) >= t_min]
    return tracks_finished


def cal_label(pre_label, next_label, inc, stride):
    """video_detection.py:415-440: objects present in both keyframes, 2-D box and (x,y,z)
    interpolated linearly, rounded to 3 decimals."""
    new_label = []
    for pre_obj in pre_label:
        for next_obj in next_label:
            if pre_obj['obj_id'] != next_obj['obj_id']:
                continue
            obj = {'obj_id': pre_obj['obj_id'], 'info': pre_obj['info'],
                   'score': max(pre_obj['score'], next_obj['score'])}
            b2 = deepcopy(pre_obj['boxes_2d'])
            b2 += inc / stride * (next_obj['boxes_2d'] - pre_obj['boxes_2d'])
            obj['boxes_2d'] = np.asarray([round(i, 3) for i in b2])
            b3 = deepcopy(pre_obj['boxes_3d'])
            b3[[3, 4, 5]] += inc / stride * (next_obj['boxes_3d'][[3, 4, 5]]
                                             - pre_obj['boxes_3d'][[3, 4, 5]])
            obj['boxes_3d'] = np.asarray([round(i, 3) for i in b3])
            new_label.append(obj)
    return new_label


def label_interpolation(labels, stride):
    """video_detection.py:371-413: labels[k] = list of objects of frame k, or []; keyframes
    every `stride` frames; the frames between two keyframes are filled by cal_label (both
    present), by a copy (one present) or stay empty."""
    def add_stride(labels_out, idx):
        pre_label, next_label = labels[idx[0]], labels[idx[-1]]
        n_mid = len(idx) - 2
        if len(pre_label) == 0:
            pre_label = next_label
            labels_out.append(pre_label)
            labels_out.extend([[] if len(next_label) == 0 else next_label] * n_mid)
        else:
            labels_out.append(pre_label)
            if len(next_label) == 0:
                labels_out.extend([pre_label] * n_mid)
                next_label = pre_label
            else:
                labels_out.extend(cal_label(pre_label, next_label, j, stride)
                                  for j in range(1, n_mid + 1))
        labels_out.append(next_label)

    labels_out, temp = [], []
    for i in range(len(labels)):
        if len(temp) == stride + 1:
            add_stride(labels_out, temp)
            temp = []
        temp.append(i)
    if len(temp) != 0:
        if len(temp) == stride + 1:
            add_stride(labels_out, temp)
        else:
            labels_out.extend(labels[idx] for idx in temp)
    return labels_out
