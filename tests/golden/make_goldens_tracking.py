#!/usr/bin/env python3
"""Golden vectors for the video-level trackers of the temporal module:
the reference's own track_through_ious (avod/core/dt_evaluator_utils.py:436-511) and, from
avod/experiments/video_detection.py, iou_2d (:70-90), cal_transformed_ious (:109-126),
track_iou (:235-277), label_interpolation / cal_label (:371-440), run in the build container
on seeded synthetic detections and the OXTS / calibration files the reference's tests bundle.

Run:  python tests/golden/make_goldens_tracking.py     (needs /root/reference; writes tracking.npz)

video_detection.py imports the config system (avod.protos.*_pb2: generated files the
repository does not hold) at module scope although none of the functions above touches it:
inert stand-in modules are registered for those names, like the tensorflow ones of
make_goldens.py; nothing of protobuf or TensorFlow runs.  The dataset object the trackers
query is a plain object carrying the reference's own KittiTrackingDataset methods (unbound).
Stored as flat arrays: every detection gets a serial number in column 0 of its row, tracks
are lists of serial numbers.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402
import make_goldens_box4ca as mb  # noqa: E402


class _InertPb2(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.startswith('avod.protos.') and name.endswith('_pb2'):
            return importlib.machinery.ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        m = MagicMock()
        m.__name__, m.__spec__ = spec.name, spec
        return m

    def exec_module(self, module):
        pass


def moving_objects(rng, n_obj, n_frames, order='hwl'):
    """n_obj cars driving straight; per frame a KITTI-ordered box and a score."""
    x0, z0 = rng.uniform(-20, 20, n_obj), rng.uniform(8, 50, n_obj)
    vx, vz = rng.uniform(-0.3, 0.3, n_obj), rng.uniform(0.2, 1.2, n_obj)
    l, w, h = rng.uniform(3.4, 4.6, n_obj), rng.uniform(1.5, 1.8, n_obj), rng.uniform(1.4, 1.7, n_obj)
    ry = rng.uniform(-0.3, 0.3, n_obj) + np.pi / 2 * rng.integers(0, 2, n_obj)
    frames = []
    for k in range(n_frames + 1):
        rows = []
        for o in range(n_obj):
            dims = [h[o], w[o], l[o]] if order == 'hwl' else [l[o], w[o], h[o]]
            rows.append(dims + [x0[o] + vx[o] * k + rng.normal(0, 0.03),
                                1.6, z0[o] + vz[o] * k + rng.normal(0, 0.03), ry[o]])
        frames.append(np.asarray(rows))
    return frames


def main():
    mb.import_evaluator()
    sys.meta_path.insert(0, _InertPb2())
    import avod.core.dt_evaluator_utils as deu
    from avod.experiments import video_detection as vd
    from avod.datasets.kitti.kitti_tracking_dataset import KittiTrackingDataset as DS
    from wavedata.tools.core import calib_utils
    rng = np.random.default_rng(20261006)
    out = {}

    # ---- A: track_through_ious ---------------------------------------------------------------
    def det(serial, box, offs, score, frame_id):
        return {'frame_id': str(frame_id), 'info': ['Car', -1, -1, -10], 'serial': serial,
                'boxes2d': np.zeros(4, np.float32), 'boxes3d': np.array(box, np.float32),
                'offsets': np.array(offs, np.float32), 'scores': np.array(score, np.float32)}
    for case, (n_obj, n_frames, drop) in enumerate([(5, 6, 0.0), (8, 7, 0.25), (3, 4, 0.5)]):
        boxes = moving_objects(rng, n_obj, n_frames)
        serial = 0
        table = []
        dets_for_track, dets_for_ious = [], [{}]
        for k in range(n_frames):
            track_item, iou_item = [], []
            for o in range(n_obj):
                if rng.uniform() >= drop:
                    sc = float(rng.uniform(0.3, 1.0))
                    track_item.append(det(serial, boxes[k][o], boxes[k + 1][o], sc, k))
                    table.append([serial, k, 0, sc] + list(boxes[k][o]) + list(boxes[k + 1][o]))
                    serial += 1
                if rng.uniform() >= drop:      # what pair k reports for its second frame
                    sc = float(rng.uniform(0.3, 1.0))
                    d = det(serial, boxes[k + 1][o], boxes[k + 1][o], sc, k + 1)
                    del d['offsets']
                    iou_item.append(d)
                    table.append([serial, k, 1, sc] + list(boxes[k + 1][o]) + [0] * 7)
                    serial += 1
            dets_for_track.append(track_item)
            dets_for_ious.append(iou_item)
        tracks = deu.track_through_ious([list(t) for t in dets_for_track],
                                        [list(t) if isinstance(t, list) else t
                                         for t in dets_for_ious], 0.6, 0.1, 2)
        out['ttI%d_table' % case] = np.asarray(table, np.float64)
        out['ttI%d_tracks' % case] = np.asarray(
            [[t['start_frame'], float(t['max_score']), len(t['trajectory'])]
             + [d['serial'] for d in t['trajectory']] + [-1] * (16 - len(t['trajectory']))
             for t in tracks], np.float64).reshape(-1, 19)
        print('track_through_ious case', case, ':', len(table), 'detections ->', len(tracks), 'tracks')

    # ---- B: video_detection --------------------------------------------------------------------
    root = os.path.join(mg.REF, 'avod/tests/datasets/Kitti/tracking/training')
    ds = types.SimpleNamespace(oxts_dir=root + '/oxts', calib_dir=root + '/calib',
                               bev_source='lidar')
    ds.get_oxts = lambda n: DS.get_oxts(ds, n)
    ds.coordinate_transform = lambda n: DS.coordinate_transform(ds, n)
    ds.label_transform = lambda labels, names: DS.label_transform(ds, labels, names)
    ds.kitti_utils = types.SimpleNamespace(
        get_calib=lambda src, name: calib_utils.read_tracking_calibration(ds.calib_dir,
                                                                          int(name[:2])))
    calib = calib_utils.read_tracking_calibration(ds.calib_dir, 0)
    out['r0'], out['tr'] = calib.r0_rect, calib.tr_velodyne_to_cam
    out['oxts_lines'] = np.array(open(root + '/oxts/0000.txt').read().splitlines()[:12])
    # iou_2d on seeded pairs [l, w, h, x, y, z, ry]
    a = moving_objects(rng, 40, 1, order='lwh')
    pa, pb = a[0], a[1].copy()
    pb[:, 3] += rng.uniform(-4, 4, 40)
    pb[:, 6] += rng.uniform(-0.5, 0.5, 40)
    pb[:5] = pa[:5]                                 # identical boxes
    out['iou2d_a'], out['iou2d_b'] = pa, pb
    out['iou2d'] = np.asarray([vd.iou_2d(pa[i].copy(), pb[i].copy()) for i in range(40)])
    # cal_transformed_ious: frames 2 -> 4 of video 0
    i1 = [{'frame_id': 2, 'boxes3d': pa[i].copy()} for i in range(40)]
    i2 = [{'frame_id': 4, 'boxes3d': pb[i].copy()} for i in range(40)]
    out['trans_iou'] = np.asarray([vd.cal_transformed_ious(ds, 0, i1[i], i2[i]) for i in range(40)])
    # track_iou over 9 frames (the bundled OXTS file covers them)
    boxes = moving_objects(rng, 6, 8, order='lwh')
    serial, table, detections = 0, [], []
    for k in range(9):
        frame = []
        for o in range(6):
            if k in (3,) and o < 3:
                continue                       # missed detections
            sc = float(rng.uniform(0.05, 1.0))
            frame.append({'frame_id': k, 'info': ['Car', -1, -1, -10], 'serial': serial,
                          'boxes2d': np.zeros(4, np.float32),
                          'boxes3d': np.array(boxes[k][o], np.float32),
                          'scores': np.array(sc, np.float32)})
            table.append([serial, k, sc] + list(boxes[k][o]))
            serial += 1
        detections.append(frame if k != 6 else [])
    # iou_2d orders its hulls [min x, max z, max x, min z], so two_d_iou sees a negative height
    # and every IoU is 0 (checked above): no association ever happens and with t_min = 2 the
    # reference returns no track at all; t_min = 1 keeps the bookkeeping observable
    out['ti_tracks_tmin2'] = np.asarray(len(vd.track_iou(ds, 0, [list(f) for f in detections], 0.1, 0.5, 0.1, 2)))
    tracks = vd.track_iou(ds, 0, detections, 0.1, 0.5, 0.1, 1)
    out['ti_table'] = np.asarray(table, np.float64)
    out['ti_empty_frame'] = np.asarray(6)
    out['ti_tracks'] = np.asarray(
        [[t['start_frame'], float(t['max_score']), len(t['trajectory'])]
         + [d['serial'] for d in t['trajectory']] + [-1] * (16 - len(t['trajectory']))
         for t in tracks], np.float64).reshape(-1, 19)
    print('track_iou:', len(table), 'detections ->', len(tracks), 'tracks; nonzero transformed ious',
          int((out['trans_iou'] > 0).sum()), 'iou_2d', int((out['iou2d'] > 0).sum()))
    # label_interpolation: 11 frames, stride 3, objects appearing / disappearing
    def obj(oid, k):
        return {'obj_id': oid, 'info': np.array(['Car', '0', '0', '-10']),
                'boxes_2d': np.asarray([100.0 + 7 * k + oid, 50.0, 180.0 + 9 * k, 120.0 + oid]),
                'boxes_3d': np.asarray([1.5, 1.6, 4.0, -3.0 + 0.37 * k + oid, 1.6, 20.0 + 1.13 * k, 0.1 * oid]),
                'score': 0.5 + 0.04 * ((k + oid) % 7)}
    labels = []
    for k in range(11):
        ids = {0: [0, 1, 2], 3: [0, 2, 5], 4: [], 7: [3], 8: [3, 4], 10: [4]}.get(k, [0])
        labels.append([obj(i, k) for i in ids])
    res = vd.label_interpolation([list(f) for f in labels], 3)
    rows = []
    for k, frame in enumerate(res):
        for o in frame:
            rows.append([k, o['obj_id'], o['score']] + list(o['boxes_2d']) + list(o['boxes_3d']))
    out['li_in'] = np.asarray([[k, o['obj_id'], o['score']] + list(o['boxes_2d']) + list(o['boxes_3d'])
                               for k, f in enumerate(labels) for o in f], np.float64)
    out['li_out'] = np.asarray(rows, np.float64)
    out['li_frames'] = np.asarray([len(labels), len(res)])
    print('label_interpolation:', len(labels), 'frames ->', len(res), 'frames,', len(rows), 'objects')
    np.savez_compressed(os.path.join(mg.HERE, 'tracking.npz'), **out)


if __name__ == '__main__':
    main()
