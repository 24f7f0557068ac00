"""Temporal module: oracle against the reference's golden vectors, the
host module against the oracle and the goldens."""
import os

import numpy as np
import pytest

from oracle import temporal as otemp

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'temporal.npz'))
CASES = sorted({int(k[1:k.index('_')]) for k in G.files if k.startswith('c')})


def _want(cid):
    n = int(G['c%d_nframes' % cid])
    return [G['c%d_out%d' % (cid, i)] for i in range(n)], n


@pytest.mark.parametrize('cid', CASES)
def test_oracle_matches_reference_goldens(cid):
    want, n = _want(cid)
    seen = []

    def recover(i, rows):
        seen.append(i)
        return rows
    got = otemp.interpolate_non_keyframe_predictions(G['c%d_pred' % cid], n, 0.1, recover)
    assert len(got) == n
    for g, w in zip(got, want):
        assert g.shape == w.shape and np.array_equal(g, w)
    assert seen == list(G['c%d_recovered' % cid])


def test_oracle_iou_matches_reference():
    b = G['iou_boxes']
    for i in range(12):
        assert np.array_equal(otemp.three_d_iou(b[i], b), G['iou_matrix'][i])


from dodt_amd.core import dt_evaluator_utils as host  # noqa: E402


@pytest.mark.parametrize('cid', CASES)
def test_host_module_matches_reference_goldens(cid):
    want, n = _want(cid)
    seen = []

    def recover(i, rows):
        seen.append(i)
        return rows
    got = host.interpolate_non_keyframe_predictions(G['c%d_pred' % cid], n, 0.1, recover)
    assert len(got) == n
    for g, w in zip(got, want):
        assert g.shape == w.shape
        np.testing.assert_allclose(g, w, rtol=1e-12, atol=1e-12)
    assert seen == list(G['c%d_recovered' % cid])


def test_exact_iou_agrees_with_the_rasterised_one():
    b = G['iou_boxes']                       # [ry,l,h,w,tx,ty,tz]
    std = b[:, [4, 5, 6, 1, 3, 2, 0]]        # -> [x,y,z,l,w,h,ry]
    for i in range(12):
        exact = host.three_d_iou(std[i], std)
        raster = G['iou_matrix'][i]
        assert np.array_equal(exact > 1e-3, raster > 1e-3) or np.abs(exact - raster).max() < 2e-2
        np.testing.assert_allclose(exact, raster, atol=2e-2)
    assert abs(host.three_d_iou(std[0], std[0:1])[0] - 1.0) < 1e-9


def test_rectangle_intersection_known_answers():
    a = np.array([0, 0, 0, 4.0, 2.0, 1.5, 0.0])
    assert abs(host.base_intersection(a, a) - 8.0) < 1e-12
    b = a.copy()
    b[0] += 1.0                                      # shifted by 1 along its length
    assert abs(host.base_intersection(a, b) - 6.0) < 1e-12
    c = a.copy()
    c[6] = np.pi / 2                                 # crossed: 2 x 2 overlap
    assert abs(host.base_intersection(a, c) - 4.0) < 1e-9
    d = a.copy()
    d[2] += 10
    assert host.base_intersection(a, d) == 0.0


def _random_boxes(rng, n, spread):
    return np.stack([rng.uniform(-spread, spread, n), rng.normal(1.65, 0.2, n), rng.uniform(0, 2 * spread, n),
                     rng.normal(3.9, 0.4, n), rng.normal(1.6, 0.15, n), rng.normal(1.5, 0.1, n),
                     rng.uniform(-np.pi, np.pi, n)], 1)


def test_batched_iou_equals_the_pair_by_pair_form():
    """three_d_iou / three_d_iou_matrix clip all candidate pairs per numpy call; three_d_iou_one_by_one is the
    Python loop they replaced.  Same polygons bit for bit, the shoelace sums may differ in their last bits."""
    rng = np.random.default_rng(11)
    a, b = _random_boxes(rng, 60, 8.0), _random_boxes(rng, 90, 8.0)       # crowded: most pairs overlap
    b[:5] = a[:5]                                                         # identical boxes
    b[5:10, [3, 4]] *= 0.3
    b[5:10, [0, 2, 6]] = a[5:10][:, [0, 2, 6]]                            # contained, same heading
    b[10:15, 6] = a[10:15, 6] = 0.0                                       # axis-aligned both
    b[15, :] = a[15, :]
    b[15, 0] += a[15, 3]                                                  # touching along an edge
    m = host.three_d_iou_matrix(a, b)
    assert m.shape == (60, 90) and (m > 0).sum() > 500
    for i in range(len(a)):
        want = host.three_d_iou_one_by_one(a[i], b)
        np.testing.assert_allclose(m[i], want, rtol=1e-12, atol=1e-13)      # shoelace sums of slivers cancel
        assert np.array_equal(host.three_d_iou(a[i], b), m[i])
    np.testing.assert_allclose(np.diag(m)[:5], 1.0, rtol=1e-12)
    assert m[15, 15] < 1e-12
    assert host.three_d_iou_matrix(np.zeros((0, 7)), b).shape == (0, 90)
    assert host.three_d_iou(a[0], np.zeros((0, 7))).shape == (0,)
    far = b.copy()
    far[:, 0] += 1000.0
    assert not host.three_d_iou_matrix(a, far).any()


def test_two_claims_on_one_detection():
    """Two keyframe-0 detections whose best match is the same keyframe-1 detection: the reference's
    next_idx.remove raises (dt_evaluator_utils.py:266-268); 'next_best' hands the later one its best free match."""
    k0 = np.zeros((2, 17))
    k0[:, :7] = [[0, 1.65, 10, 4, 1.6, 1.5, 0], [1.0, 1.65, 10.2, 4, 1.6, 1.5, 0]]
    k1 = np.zeros((2, 17))
    k1[:, :7] = [[0.5, 1.65, 10.1, 4, 1.6, 1.5, 0], [3.0, 1.65, 11.5, 4, 1.6, 1.5, 0]]
    k1[:, 16] = 1
    p = np.concatenate([k0, k1])
    p[:, 7] = [0.9, 0.8, 0.7, 0.6]
    with pytest.raises(ValueError):
        host.interpolate_non_keyframe_predictions(p, 3, 0.1)
    out = host.interpolate_non_keyframe_predictions(p, 3, 0.1, on_conflict='next_best')
    assert [len(o) for o in out] == [2, 2, 2]
    np.testing.assert_allclose(sorted(out[2][:, 0]), [0.5, 3.0])          # both keyframe-1 boxes were matched
    np.testing.assert_allclose(sorted(out[1][:, 0]), [0.25, 2.0])         # midpoints of (0 -> 0.5) and (1 -> 3)
    with pytest.raises(ValueError):
        host.interpolate_non_keyframe_predictions(p, 3, 0.1, on_conflict='ignore')


def test_tracking_encoder_feeds_the_tracker():
    """encode_tracking_dets: records of a sequence's pairs -> the two lists track_through_ious takes (the
    reference goes through text files, dt_evaluator_utils.py:368-434); three cars driving on, one pair
    without detections in the middle."""
    from dodt_amd import synth
    n_pairs, tau = 5, 2
    pairs = []
    for k in range(n_pairs):
        rec = np.zeros((6, 17), np.float32)
        for f in range(2):
            for c in range(3):
                z = 12.0 + 8 * c + 0.8 * (k * tau + f * tau)
                rec[3 * f + c] = [-4.0 + 4 * c, 1.65, z, 3.9, 1.6, 1.5, 0.1, 0.9 - 0.1 * c, 0,
                                  *((-4.0 + 4 * c, 1.65, z + 0.8 * tau, 3.9, 1.6, 1.5, 0.1) if f == 0 else (0,) * 7), f]
        pairs.append((k * tau, k * tau + tau, rec if k != 2 else np.zeros((0, 17), np.float32)))
    dt, di = host.encode_tracking_dets(pairs, synth.P2, synth.IMAGE_WH, ['Car'], 0.1)
    assert len(dt) == n_pairs - 1 and len(di) == n_pairs and di[0] == {}
    assert all(len(f) == 3 for f in dt) and all(len(f) == 3 for f in di[1:])
    d = dt[1][2]
    assert d['frame_id'] == '2' and d['boxes3d'].shape == (7,) and d['offsets'].shape == (7,)
    np.testing.assert_allclose(d['boxes3d'], [1.5, 1.6, 3.9, 4.0, 1.65, 29.6, 0.1], atol=1e-3)   # h w l x y z ry
    np.testing.assert_allclose(d['offsets'][5], 31.2, atol=1e-3)
    tracks = host.track_through_ious(dt, di, 0.5, 0.1, 2)
    assert len(tracks) >= 3 and max(len(t['trajectory']) for t in tracks) >= 2
