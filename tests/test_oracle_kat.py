"""Known answers asserted by the reference's own unit tests, replayed on the
oracle.  Each test names the reference test it
takes its numbers from.  CPU only."""
import numpy as np

from oracle import anchors as oanchors
from oracle import boxes as oboxes
from oracle import points as opoints
from oracle import tfops


def test_voxelize_2d_layout():
    """wavedata/.../voxel_grid_2d_test.py:38-59"""
    pts = np.array([[-39.99, 4.99, 0], [39.99, 4.99, 0], [-39.99, -4.99, 0],
                    [39.99, -4.99, 0], [-39.99, 4.99, 69.99],
                    [39.99, 4.99, 69.99], [-39.99, -4.99, 69.99],
                    [39.99, -4.99, 69.99], [-39.99, 4.99, 69.99],
                    [39.99, 4.99, 69.99], [-39.99, -4.99, 69.99],
                    [39.99, -4.99, 69.99]])
    filled = np.floor((pts * 10) + [400, 0, 0]).astype(np.int32)
    filled[:, 1] = 0
    expect = -np.ones((800, 1, 700))
    for idx in filled:
        expect[tuple(idx)] = 0
    v = opoints.voxelize_2d(pts, 0.1)
    assert (v.min_voxel_coord == [-400, 0, 0]).all()
    assert (v.max_voxel_coord == [399, 0, 699]).all()
    assert (v.num_divisions == [800, 1, 700]).all()
    assert (v.leaf_layout_2d == expect).all()


def test_voxelize_2d_extents():
    """voxel_grid_2d_test.py:61-79"""
    rng = np.random.default_rng(0)
    pts = (rng.random((70000, 3)) * [80, 8, 60]) - [40, 4, 0]
    bad = np.array([[-30, 30], [-3, 3], [10, 60]])
    try:
        opoints.voxelize_2d(pts, 0.1, bad)
        raise AssertionError('expected ValueError')
    except ValueError:
        pass
    v = opoints.voxelize_2d(pts, 0.1, np.array([[-50, 50], [-5, 5], [0, 70]]))
    assert (v.num_divisions == [1000, 1, 700]).all()
    assert v.leaf_layout_2d.shape == (1000, 1, 700)


def test_map_to_index():
    """voxel_grid_2d_test.py:81-118"""
    rng = np.random.default_rng(1)
    pts = (rng.random((70000, 3)) * [80, 8, 60]) - [40, 4, 0]
    v = opoints.voxelize_2d(pts, 0.1, np.array([[-50, 50], [-5, 5], [0, 70]]))
    assert (opoints.map_to_index(v, np.array([[0, 0]])) == [500, 0]).all()
    assert (opoints.map_to_index(v, np.array([[0, 0]]) + 0.1) == [501, 1]).all()
    assert (opoints.map_to_index(v, np.array([[-50, 0]])) == [0, 0]).all()
    assert (opoints.map_to_index(v, np.array([[50, 70]])) == [1000, 700]).all()
    assert (opoints.map_to_index(v, np.array([[60, 80]])) == [1000, 700]).all()


def test_get_point_filter():
    """wavedata/.../obj_utils_test.py:65-94"""
    pts = np.array([[0, 1, 0], [0, -1, 0], [5, 1, 5], [-5, 1, 5]])
    ext = [[-2, 2], [-2, 2], [-2, 2]]
    f1 = opoints.point_filter(pts.T, ext, [0, -1, 0, 0], 0.5)
    f2 = opoints.point_filter(pts.T, ext, [0, -1, 0, 0], 2.0)
    assert f1.sum() == 1 and f2.sum() == 2
    np.testing.assert_allclose(pts[f1], [[0, 1, 0]])
    np.testing.assert_allclose(pts[f2], [[0, 1, 0], [0, -1, 0]])


def test_create_slice_filter():
    """avod/datasets/kitti/kitti_utils_test.py:19-40"""
    pc = np.array([[1.0, 1.0, 1.0], [0.0, 1.0, 3.0], [1.0, 1.0, 1.0]])
    f = opoints.slice_filter(pc, [[-2, 2], [-5, 5], [-2, 2]], [0, 1, 0, 0],
                             0.2, 2.0)
    np.testing.assert_equal(f, [False, True, False])


def test_dist_to_plane():
    """wavedata/.../geometry_utils_test.py:9"""
    p = [[1, 1, 1]]
    assert abs(opoints.dist_to_plane([0, 0, 1, 0], p)[0] - 1) < 1e-7
    assert abs(opoints.dist_to_plane([1, 1, 1, 0], p)[0] - np.sqrt(3)) < 1e-7
    assert abs(opoints.dist_to_plane([-1, -1, -1, 0], p)[0] + np.sqrt(3)) < 1e-7


def test_integral_image_2d():
    """wavedata/.../integral_image_2d_test.py:9"""
    sat = oanchors.summed_area_table(np.ones((3, 3), dtype=np.float32))

    def q(rows):
        return oanchors.sat_query(sat, np.array(rows).T.astype(np.uint32))
    assert list(q([[0, 0, 1, 1], [0, 0, 2, 2], [0, 0, 3, 3]])) == [1, 4, 9]
    assert list(q([[1, 1, 2, 2], [1, 1, 3, 3]])) == [1, 4]
    assert list(q([[0, 0, 3, 1]])) == [3]
    assert list(q([[0, 0, 2312, 162]])) == [9]


def test_tile_anchors_3d():
    """avod/core/anchor_generators/grid_anchor_3d_generator_test.py:27-70"""
    plane = np.array([0., -1., 0., 0.])
    clusters = np.array([[1., 1., 1.], [2., 1., 1.]])
    hp = np.pi / 2
    expect = np.array([[-0.5, 0., 0.5, 1., 1., 1., 0.],
                       [-0.5, 0., 0.5, 1., 1., 1., hp],
                       [-0.5, 0., 0.5, 2., 1., 1., 0.],
                       [-0.5, 0., 0.5, 2., 1., 1., hp],
                       [0.5, 0., 0.5, 1., 1., 1., 0.],
                       [0.5, 0., 0.5, 1., 1., 1., hp],
                       [0.5, 0., 0.5, 2., 1., 1., 0.],
                       [0.5, 0., 0.5, 2., 1., 1., hp]])
    got = oanchors.tile_anchors_3d([(-1., 1.), (-1., 0.), (0., 1.)], clusters,
                                   [1, 1], plane)
    np.testing.assert_almost_equal(got, expect, decimal=3)
    assert oanchors.tile_anchors_3d([(0., 0.), (-1., 0.), (0., 2.)], clusters,
                                    [1, 1], plane).shape == (0, 7)
    assert oanchors.tile_anchors_3d([(-1., 1.), (-1., 0.), (0., 0.)], clusters,
                                    [1, 1], plane).shape == (0, 7)


def test_project_to_bev():
    """avod/core/anchor_projector_test.py:16-90 (three cases) and :92-126
    (the float32 tensor twin gives the same numbers)"""
    cases = [
        ([[1, 0, 3, 2, 0, 6], [3, 0, 3, 2, 0, 2]], [[0, 5], [0, 10]],
         [[0, 4, 2, 10], [2, 6, 4, 8]]),
        ([[0, 0, 3, 2, 0, 6], [3, 0, 3, 2, 0, 2]], [[-5, 5], [0, 10]],
         [[4, 4, 6, 10], [7, 6, 9, 8]]),
        ([[0, 0, 0, 10, 0, 2]], [[-3, 3], [0, 10]], [[-2, 9, 8, 11]]),
    ]
    for anchors, ext, expect in cases:
        rng = np.tile(np.diff(ext, axis=1).flatten(), 2)
        for dt in (np.float64, np.float32):
            b, n = oboxes.project_to_bev(np.asarray(anchors, dtype=np.float64),
                                         ext, dtype=dt)
            np.testing.assert_allclose(b, expect, rtol=1e-5)
            np.testing.assert_allclose(n, np.asarray(expect) / rng, rtol=1e-5)


def test_reorder_projected_boxes():
    """anchor_projector_test.py:209-224"""
    got = oboxes.reorder_projected_boxes(np.array([[1, 2, 3, 4], [5, 6, 7, 8]]))
    np.testing.assert_array_equal(got, [[2, 1, 4, 3], [6, 5, 8, 7]])


def test_offset_to_anchor():
    """avod/core/anchor_encoder_test.py:67-122"""
    anchors = np.asarray([[1, 2, 3, 4, 6, 5], [0, 0, 0, 2, 3, 1]], np.float32)
    off = np.array([[0.5, 0.02, 0.01, 0.1, 0.4, 0.03],
                    [0.04, 0.1, 0.03, 0.001, 0.3, 0.03]], dtype=np.float32)
    expect = np.array([[3.0, 2.12, 3.05, 4.420, 8.9509, 5.152],
                       [0.08, 0.3, 0.03, 2.002, 4.05, 1.03]], dtype=np.float32)
    for dt in (np.float64, np.float32):
        np.testing.assert_almost_equal(
            oboxes.offset_to_anchor(anchors, off, dtype=dt), expect, decimal=3)


def test_box_3d_to_anchor():
    """avod/core/box_3d_encoder_test.py:9-77"""
    b = np.asarray([[1, 2, 3, 4, 5, 6, 0], [0, 0, 0, 1, 2, 3, 0],
                    [0, 0, 0, 1, 2, 3, np.pi / 2]], dtype=np.float64)
    np.testing.assert_allclose(oanchors.box_3d_to_anchor(b),
                               [[1, 2, 3, 4, 6, 5], [0, 0, 0, 1, 3, 2],
                                [0, 0, 0, 2, 3, 1]])
    b = np.asarray([[1, 2, 3, 4, 5, 6, np.pi],
                    [1, 2, 3, 4, 5, 6, 3 * np.pi / 2]])
    np.testing.assert_allclose(oanchors.box_3d_to_anchor(b),
                               [[1, 2, 3, 4, 6, 5], [1, 2, 3, 5, 6, 4]])
    b = np.asarray([[1, 2, 3, 4, 5, 6, np.pi * 4 / 5],
                    [1, 2, 3, 4, 5, 6, 8 * np.pi / 5]])
    np.testing.assert_allclose(oanchors.box_3d_to_anchor(b, True),
                               [[1, 2, 3, 4, 6, 5], [1, 2, 3, 5, 6, 4]])
    np.testing.assert_allclose(oboxes.box_3d_to_anchor_ortho(b, np.float64),
                               [[1, 2, 3, 4, 6, 5], [1, 2, 3, 5, 6, 4]])


def test_anchors_to_box_3d_and_back():
    """box_3d_encoder_test.py:79-140"""
    a = np.asarray([[-0.59, 1.90, 25.01, 3.2, 1.66, 1.61],
                    [-0.59, 1.90, 25.01, 1.61, 1.66, 3.2]], dtype=np.float32)
    expect = np.asarray([[-0.59, 1.90, 25.01, 3.2, 1.61, 1.66, 0],
                         [-0.59, 1.90, 25.01, 3.2, 1.61, 1.66, -1.57]])
    for dt in (np.float64, np.float32):
        np.testing.assert_almost_equal(
            oboxes.anchors_to_box_3d(a, fix_lw=True, dtype=dt), expect,
            decimal=3)
    b = np.asarray([[-0.59, 1.90, 25.01, 3.2, 1.61, 1.66, 0],
                    [-0.59, 1.90, 25.01, 3.2, 1.6, 1.66, -np.pi / 2]],
                   dtype=np.float32)
    np.testing.assert_almost_equal(
        oboxes.box_3d_to_anchor_ortho(b),
        [[-0.59, 1.90, 25.01, 3.2, 1.66, 1.61],
         [-0.59, 1.90, 25.01, 1.6, 1.66, 3.20]], decimal=2)


def test_box_3d_to_box_4c():
    """avod/core/box_4c_encoder_test.py:10-133"""
    gp = [0, -1, 0, 2]
    for dt, dec in ((np.float64, 3), (np.float32, 3)):
        f = lambda b: oboxes.box_3d_to_box_4c(np.asarray(b, float), gp, dt)[0]
        np.testing.assert_almost_equal(
            f([0, 0, 0, 2, 1, 5, 0]),
            [1, 1, -1, -1, 0.5, -0.5, -0.5, 0.5, 2, 7], decimal=dec)
        np.testing.assert_almost_equal(
            f([0, 0, 0, 2, 1, 5, -np.pi / 2]),
            [0.5, 0.5, -0.5, -0.5, 1, -1, -1, 1, 2, 7], decimal=dec)
        e1 = [0.733, 1.115, -0.733, -1.115, 0.845, -0.079, -0.845, 0.079, 2, 7]
        e2 = [0.845, 0.079, -0.845, -0.079, 0.733, -1.115, -0.733, 1.115, 2, 7]
        e3 = [0.079, 0.845, -0.079, -0.845, 1.115, -0.733, -1.115, 0.733, 2, 7]
        e4 = [1.115, 0.733, -1.115, -0.733, 0.079, -0.845, -0.079, 0.845, 2, 7]
        for k, e in zip((-1, -3, -5, -7, 1, 3, 5, 7),
                        (e1, e2, e3, e4, e4, e3, e2, e1)):
            np.testing.assert_almost_equal(
                f([0, 0, 0, 2, 1, 5, k * np.pi / 8]), e, decimal=dec)
        np.testing.assert_almost_equal(
            f([10, 0, 10, 2, 1, 5, -np.pi / 8]),
            [10.733, 11.115, 9.267, 8.885, 10.845, 9.921, 9.155, 10.079, 2, 7],
            decimal=dec)
        for y, h in ((3.0, (-1, 4)), (2.0, (0, 5)), (1.0, (1, 6))):
            np.testing.assert_almost_equal(
                f([0, y, 0, 2, 1, 5, 0]),
                [1, 1, -1, -1, 0.5, -0.5, -0.5, 0.5, h[0], h[1]])


def test_box_4c_to_box_3d():
    """box_4c_encoder_test.py:185-204"""
    gp = np.asarray([0, -1, 0, 2])
    b1 = [1.0, 0.0, -1.0, 0.5, 0.5, -1.0, 0.0, 1.0, 1.0, 3.0]
    b2 = [1.0, 0.0, -1.0, -0.5, 0.0, -1.0, 0.5, 1.0, 1.0, 3.0]
    for dt in (np.float64, np.float32):
        got = oboxes.box_4c_to_box_3d(np.asarray([b1, b2]), gp, dtype=dt)
        np.testing.assert_almost_equal(
            got[0], [0.125, 1.000, 0.125, 1.768, 1.414, 2.000, -0.785], 3)
        np.testing.assert_almost_equal(
            got[1], [-0.125, 1.000, 0.125, 1.768, 1.414, 2.000, 0.785], 3)


def test_angle_vector_to_orientation():
    """avod/core/orientation_encoder_test.py:26-87"""
    ang = np.arange(-np.pi + 0.1, np.pi, 0.3)
    vec = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    np.testing.assert_allclose(
        oboxes.angle_vector_to_orientation(vec, np.float64), ang, atol=1e-12)


def test_rot90_equivalence():
    """kitti_utils_test.py:42-56 (the BEV maps are transposed then flipped)"""
    rng = np.random.default_rng(123)
    m = rng.random((800, 700))
    np.testing.assert_allclose(np.flip(m.transpose(), axis=0), np.rot90(m))


def test_nms_twins_agree():
    rng = np.random.default_rng(5)
    n = 400
    cy, cx = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    h, w = rng.uniform(0.02, 0.2, n), rng.uniform(0.02, 0.2, n)
    boxes = np.stack([cy - h, cx - w, cy + h, cx + w], 1).astype(np.float32)
    boxes[::37, 2] = boxes[::37, 0]            # zero-area boxes
    scores = rng.uniform(0, 1, n).astype(np.float32)
    scores[10:20] = scores[10]                 # ties
    for thr, k in ((0.8, 50), (0.01, 100), (0.5, 1000)):
        a = tfops.non_max_suppression(boxes, scores, k, thr)
        b = tfops.non_max_suppression_fast(boxes, scores, k, thr)
        assert np.array_equal(a, b)
