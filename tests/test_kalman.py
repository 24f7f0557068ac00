"""The Kalman track filter against vectors from the reference's own Tracker
(tests/golden/make_goldens_kalman.py) and the KF association pipeline by what it must do (its
assignment step is "parity unpinned", see dodt_amd/experiments/video_detection_kf.py).  CPU only."""
import os

import numpy as np
import pytest

from dodt_amd.experiments import video_detection_kf as vkf
from dodt_amd.utils.kalman_tracker import Tracker

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'kalman.npz'))


def test_tracker_matrices_match_reference():
    t = Tracker()
    for name in ('F', 'H', 'P', 'Q', 'R'):
        assert np.array_equal(np.asarray(getattr(t, name), np.float64), G['init_' + name]), name
    assert (t.id, t.hits, t.no_losses, t.dets, t.box) == (0, 0, 0, [], [])


@pytest.mark.parametrize('case', range(6))
def test_tracker_sequences_match_reference(case):
    ops, zs, x0 = G['c%d_ops' % case], G['c%d_z' % case], G['c%d_x0' % case]
    trk = Tracker()
    trk.L, trk.R_scaler = [float(v) for v in G['c%d_LR' % case]]
    trk.update_R()
    trk.x_state = np.array([[x0[0], 0, x0[1], 0, x0[2], 0, x0[3], 0]], np.float64).T
    for k in range(len(ops)):
        if ops[k]:
            trk.kalman_filter(zs[k][:, None])
        else:
            trk.predict_only()
        # float64, the same operation order: agreement to the last bits (np.linalg.inv and the
        # reference's scipy.linalg.inv both go through LAPACK's getrf / getri)
        assert np.allclose(trk.x_state[:, 0], G['c%d_x' % case][k], rtol=1e-12, atol=1e-12)
        assert np.allclose(trk.P, G['c%d_P' % case][k], rtol=1e-12, atol=1e-12)
    assert trk.x_state.shape == (8, 1) and trk.x_state.dtype == np.float64


def _det(frame, x, z, score=0.9, ry=0.1):
    return {'frame_id': frame, 'boxes3d': np.array([1.5, 1.6, 4.0, x, 1.6, z, ry]), 'boxes2d': np.array([0., 0., 10., 10.]) + x,
            'scores': score}


EGO0 = lambda a, b: (np.zeros(3), np.eye(3), 0.0)           # a parked ego vehicle  # noqa: E731
CALIB = (np.eye(3), np.hstack([np.eye(3), np.zeros((3, 1))]))


def test_iou_3d_of_shifted_boxes():
    a, b = _det(0, 0.0, 20.0, ry=0.0)['boxes3d'], _det(0, 1.0, 20.0, ry=0.0)['boxes3d']    # 4 m long along x, shifted 1 m
    assert abs(vkf.iou_3d(a, a) - 1.0) < 1e-12
    assert abs(vkf.iou_3d(a, b) - 3.0 / 5.0) < 1e-12
    up = b.copy()
    up[4] -= 0.75                                                # half the height higher
    assert abs(vkf.iou_3d(a, up) - (3 * 0.75) / (2 * 4 * 1.5 - 3 * 0.75)) < 1e-12
    assert vkf.cal_transformed_ious(EGO0, CALIB, _det(0, 0.0, 20.0, ry=0.0), _det(2, 1.0, 20.0, ry=0.0)) == \
        pytest.approx(0.6, abs=1e-9)


def test_inside_and_correct_direction():
    assert vkf.inside(_det(0, 0.0, 20.0)) and not vkf.inside(_det(0, 30.0, 20.0))
    assert not vkf.inside(_det(0, 0.0, 75.0)) and not vkf.inside(_det(0, 0.0, -1.0))
    trk = Tracker()
    trk.dets = [_det(0, 0, 20, ry=0.2), _det(1, 0, 21, ry=-0.3), _det(2, 0, 22, ry=0.1)]
    new = _det(3, 0, 23, ry=-0.4)
    vkf.correct_direction(trk, new)
    assert all(d['boxes3d'][-1] > 0 for d in trk.dets) and new['boxes3d'][-1] == 0.4
    short = Tracker()
    short.dets = [_det(0, 0, 20, ry=-0.2)]
    vkf.correct_direction(short, new)
    assert new['boxes3d'][-1] == 0.4                       # fewer than three detections: untouched


def test_assignment_edge_cases_and_gate():
    m, ud, ut = vkf.assign_detections_to_trackers(EGO0, CALIB, [], [], 0.1)
    assert m.size == 0 and ud == [] and ut == []
    m, ud, ut = vkf.assign_detections_to_trackers(EGO0, CALIB, [], [_det(0, 0, 20), _det(0, 5, 30)], 0.1)
    assert m.size == 0 and ud == [0, 1] and ut == []
    m, ud, ut = vkf.assign_detections_to_trackers(EGO0, CALIB, [_det(0, 0, 20)], [], 0.1)
    assert m.size == 0 and ud == [] and ut == [0]
    trks = [_det(0, -5.0, 20.0), _det(0, 6.0, 35.0)]
    dets = [_det(1, 6.2, 35.5), _det(1, -5.1, 20.6), _det(1, 20.0, 60.0)]
    m, ud, ut = vkf.assign_detections_to_trackers(EGO0, CALIB, trks, dets, 0.1)
    assert sorted(map(tuple, m.tolist())) == [(0, 1), (1, 0)] and ud == [2] and ut == []
    # an assigned pair below the gate is unmatched on both sides
    m, ud, ut = vkf.assign_detections_to_trackers(EGO0, CALIB, [_det(0, 0.0, 20.0)], [_det(1, 25.0, 60.0)], 0.1)
    assert m.shape == (0, 2) and ud == [0] and ut == [0]


def test_interpolation_fills_the_stride():
    trk = Tracker()
    trk.dets = [_det(4, 0.0, 20.0, score=0.6)]
    nxt = _det(8, 4.0, 28.0, score=0.8)
    vkf.interpolation_detections(trk, nxt, 4)
    assert [d['frame_id'] for d in trk.dets] == [4, 5, 6, 7, 8]
    # the reference adds ONE increment to a copy of the last real detection for every frame in between
    for d in trk.dets[1:4]:
        assert np.allclose(d['boxes3d'][[3, 5]], [1.0, 22.0]) and d['is_virtual'] and d['scores'] == 0.8
    assert trk.dets[-1] is nxt


def test_kf_pipeline_follows_objects_and_drops_noise():
    rng = np.random.default_rng(3)
    stride, n_key = 2, 9
    starts = [(-6.0, 15.0, 0.15, 1.9), (5.0, 40.0, -0.1, -1.2), (0.5, 25.0, 0.0, 0.8)]
    frames = []
    for k in range(n_key):
        dets = [_det(k * stride, x0 + vx * k + rng.normal(0, 0.02), z0 + vz * k + rng.normal(0, 0.02), ry=1.5)
                for (x0, z0, vx, vz) in starts]          # (driving along z: consecutive keyframes overlap)
        if k == 4:
            dets.pop(1)                                    # a missed detection: the track coasts
        if k in (2, 6):
            dets.append(_det(k * stride, rng.uniform(-3, 3), rng.uniform(50, 60), score=0.95))   # clutter, once each
        dets.append(_det(k * stride, 1.0, 30.0, score=0.05))                                     # below sigma_l
        frames.append(dets)
    tracks = vkf.kf_pipeline(EGO0, CALIB, frames, stride, n_key * stride, sigma_l=0.3, iou_threshold=0.1)
    assert len(tracks) == 3 and len({t.id for t in tracks}) == 3
    for t in tracks:
        ids = [d['frame_id'] for d in t.dets]
        assert ids == sorted(ids) and ids[0] == 0 and len(set(ids)) == len(ids)
        assert ids == list(range(ids[0], ids[-1] + 1))     # every frame of the stride is filled
        assert t.hits >= 7
        x0, z0, vx, vz = min(starts, key=lambda s: abs(s[0] - t.dets[0]['boxes3d'][3]) + abs(s[1] - t.dets[0]['boxes3d'][5]))
        real = [d for d in t.dets if not d.get('is_virtual')]
        for d in real:
            k = d['frame_id'] / stride
            assert abs(d['boxes3d'][3] - (x0 + vx * k)) < 0.2 and abs(d['boxes3d'][5] - (z0 + vz * k)) < 0.2
        # the filter's positions follow the measurements [x, y, z, z]
        assert abs(t.box[0] - t.dets[-1]['boxes3d'][3]) < 1.0 and abs(t.box[2] - t.box[3]) < 1e-9
