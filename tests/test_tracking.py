"""Video-level trackers of the temporal module on the host against fixtures
the reference's own code produced (tests/golden/make_goldens_tracking.py):
track_through_ious (avod/core/dt_evaluator_utils.py:436-511); iou_2d, cal_transformed_ious,
track_iou, label_interpolation (avod/experiments/video_detection.py)."""
import os

import numpy as np
import pytest

from dodt_amd.core import dt_evaluator_utils as host
from dodt_amd.datasets.kitti import kitti_tracking_utils as ktu
from dodt_amd.experiments import video_detection as vd

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tracking.npz'))


def _tracks(tracks):
    rows = [[t['start_frame'], float(t['max_score']), len(t['trajectory'])]
            + [d['serial'] for d in t['trajectory']] + [-1] * (16 - len(t['trajectory']))
            for t in tracks]
    return np.asarray(rows, np.float64).reshape(-1, 19)


@pytest.mark.parametrize('case', [0, 1, 2])
def test_track_through_ious_matches_reference(case):
    table = G['ttI%d_table' % case]
    n_pairs = int(table[:, 1].max()) + 1
    dets_for_track = [[] for _ in range(n_pairs)]
    dets_for_ious = [{}] + [[] for _ in range(n_pairs)]
    for r in table:
        d = {'serial': int(r[0]), 'frame_id': str(int(r[1] + r[2])),
             'boxes3d': r[4:11].astype(np.float32), 'scores': np.float32(r[3])}
        if r[2] == 0:
            d['offsets'] = r[11:18].astype(np.float32)
            dets_for_track[int(r[1])].append(d)
        else:
            dets_for_ious[int(r[1]) + 1].append(d)
    before = [len(f) for f in dets_for_track]
    got = _tracks(host.track_through_ious(dets_for_track, dets_for_ious, 0.6, 0.1, 2))
    want = G['ttI%d_tracks' % case]
    assert got.shape == want.shape and np.array_equal(got[:, [0, 2]], want[:, [0, 2]])
    assert np.array_equal(got[:, 3:], want[:, 3:])                  # the same detections
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=0, atol=0)
    assert [len(f) for f in dets_for_track] == before              # inputs untouched


def _ego():
    lines = [str(l) for l in G['oxts_lines']]

    def ego(fa, fb):
        return ktu.coordinate_transform(ktu.Oxts(lines[int(fa)]), ktu.Oxts(lines[int(fb)]))
    return ego


def test_iou_2d_and_transformed_iou_match_reference():
    a, b = G['iou2d_a'], G['iou2d_b']
    got = np.asarray([vd.iou_2d(a[i], b[i]) for i in range(len(a))])
    assert np.array_equal(got, G['iou2d'])
    # as written in the reference the hull is [min x, max z, max x, min z]: the IoU of a box
    # with itself is 0 (video_detection.py:82-90) -- reproduced, not fixed
    assert got[0] == 0.0 and np.array_equal(a[0], b[0])
    calib = (G['r0'], G['tr'])
    t = np.asarray([vd.cal_transformed_ious(_ego(), calib, {'frame_id': 2, 'boxes3d': a[i]},
                                            {'frame_id': 4, 'boxes3d': b[i]})
                    for i in range(len(a))])
    assert np.array_equal(t, G['trans_iou'])
    # the registration itself moves the box (label_transform, kitti_tracking_dataset.py:338-372)
    trans, matrix, delta = _ego()(2, 4)
    moved = vd.label_transform_box(b[7], calib[0], calib[1], trans, matrix, delta)
    assert 0.2 < np.linalg.norm(moved[3:6] - b[7][3:6]) < 3.0 and moved[6] == b[7][6] + delta
    # a hull in image order does overlap: two_d_iou itself is the usual IoU
    assert vd.two_d_iou(np.array([0., 0, 2, 2]), np.array([[1., 1, 3, 3]]))[0] == 0.143


def test_track_iou_matches_reference():
    table = G['ti_table']
    detections = [[] for _ in range(9)]
    for r in table:
        detections[int(r[1])].append({'serial': int(r[0]), 'frame_id': int(r[1]),
                                      'boxes3d': r[3:10].astype(np.float32),
                                      'scores': np.float32(r[2])})
    detections[int(G['ti_empty_frame'])] = []
    calib = (G['r0'], G['tr'])
    got = _tracks(vd.track_iou(_ego(), calib, detections, 0.1, 0.5, 0.1, 1))
    assert np.array_equal(got, G['ti_tracks'])
    assert len(vd.track_iou(_ego(), calib, detections, 0.1, 0.5, 0.1, 2)) == int(G['ti_tracks_tmin2'])


def test_label_interpolation_matches_reference():
    n_in, n_out = [int(v) for v in G['li_frames']]
    labels = [[] for _ in range(n_in)]
    for r in G['li_in']:
        labels[int(r[0])].append({'obj_id': int(r[1]), 'info': None, 'score': float(r[2]),
                                  'boxes_2d': r[3:7].copy(), 'boxes_3d': r[7:14].copy()})
    res = vd.label_interpolation(labels, 3)
    assert len(res) == n_out
    rows = np.asarray([[k, o['obj_id'], o['score']] + list(o['boxes_2d']) + list(o['boxes_3d'])
                       for k, f in enumerate(res) for o in f], np.float64)
    assert rows.shape == G['li_out'].shape and np.array_equal(rows, G['li_out'])
    # fewer frames than a stride: passed through
    assert vd.label_interpolation(labels[:2], 3) == labels[:2]
