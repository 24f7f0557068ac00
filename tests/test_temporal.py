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
