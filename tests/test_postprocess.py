"""Detection post-processing: oracle against the reference's golden
vectors, the vectorised host module against the oracle."""
import os

import numpy as np
import pytest

from dodt_amd.core import dt_inference_utils as host
from oracle import postprocess as opost

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'kitti_format.npz'))


def test_oracle_projection_matches_reference_goldens():
    boxes, p2, size = G['boxes_3d'], G['p2'], tuple(G['image_size'])
    for key, before in (('img_boxes', True), ('img_boxes_discard_after', False)):
        want = G[key]
        for i, b in enumerate(boxes):
            got = opost.project_box_to_image_space(b, p2, True, size, before)
            if np.isnan(want[i, 0]):
                assert got is None, (key, i)
            else:
                assert got is not None and np.array_equal(got, want[i]), (key, i)
    raw = np.stack([opost.project_box_to_image_space(b, p2) for b in boxes])
    assert np.array_equal(raw, G['img_boxes_raw'])
    with pytest.raises(ValueError):
        opost.project_box_to_image_space(boxes[0], p2, truncate=True)


def test_host_projection_matches_goldens_and_oracle():
    boxes, p2, size = G['boxes_3d'], G['p2'], tuple(G['image_size'])
    for key, before in (('img_boxes', True), ('img_boxes_discard_after', False)):
        got, valid = host.project_boxes_to_image_space(boxes, p2, True, size, before)
        want = G[key]
        assert np.array_equal(valid, np.isfinite(want[:, 0]))
        np.testing.assert_allclose(got[valid], want[valid], rtol=1e-12, atol=1e-9)
    raw, valid = host.project_boxes_to_image_space(boxes, p2)
    assert valid.all()
    np.testing.assert_allclose(raw, G['img_boxes_raw'], rtol=1e-12, atol=1e-9)


def test_kitti_rows_from_records():
    rng = np.random.default_rng(5)
    boxes, p2, size = G['boxes_3d'], G['p2'], tuple(G['image_size'])
    rec = np.zeros((len(boxes), 17))
    rec[:, :7] = boxes
    rec[:, 7] = rng.uniform(0, 1, len(boxes))
    types, rows = opost.convert_pred_to_kitti_format(rec, p2, size, ['Car'], 0.1)
    table = host.convert_pred_to_kitti_format(rec, p2, size, ['Car'], 0.1)
    assert table.shape == (len(rows), 16) and list(table[:, 0]) == types
    np.testing.assert_allclose(table[:, 1:].astype(np.float64), rows, atol=1.001e-3)
    # h, w, l order and the constant columns (dt_inference_utils.py:186-207)
    r0 = table[0, 1:].astype(np.float64)
    assert r0[0] == -1 and r0[1] == -1 and r0[2] == -10
    kept = rec[rec[:, 7] >= 0.1]
    valid = np.isfinite(G['img_boxes'][rec[:, 7] >= 0.1][:, 0])
    first = kept[valid][0]
    np.testing.assert_allclose(r0[7:10], np.round([first[5], first[4], first[3]], 3))
    assert host.convert_pred_to_kitti_format(rec, p2, size, ['Car'], 2.0) == []
    assert opost.convert_pred_to_kitti_format(rec, p2, size, ['Car'], 2.0)[0] == []


B4 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'box4ca.npz'))


def test_oracle_box_4ca_records_match_reference_goldens():
    """avod/core/dt_evaluator.py:1134-1259 run by tests/golden/make_goldens_box4ca.py:
    every column of every record, bit for bit (float32 arithmetic, thresholds included)."""
    for c in range(int(B4['n_cases'])):
        boxes = [B4['c%d_boxes_3d_%d' % (c, f)] for f in range(2)]
        ori = [B4['c%d_orientations_%d' % (c, f)] for f in range(2)]
        sm = [B4['c%d_softmax_%d' % (c, f)] for f in range(2)]
        got = opost.avod_predicted_boxes_3d_and_scores(boxes, ori, sm, B4['c%d_corr_offsets' % c])
        want = B4['c%d_records' % c]
        assert got.shape == want.shape
        assert np.array_equal(got, want), c
    # the thresholds are exercised: some rows swapped, some flipped, some untouched
    want, b0 = B4['c1_records'], B4['c1_boxes_3d_0']
    n0 = len(b0)
    assert 0 < (want[:n0, 3] != b0[:, 3]).sum() < n0
    assert (np.abs(want[:n0, 6] - b0[:, 6]) > 3.0).any()
