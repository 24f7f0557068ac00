"""Ego-motion registration of a pair's second frame: the host module
and the oracle against fixtures the reference's own code produced on its bundled tracking
sequence (tests/golden/make_goldens_egomotion.py)."""
import os

import numpy as np
import pytest

from dodt_amd import config
from dodt_amd.datasets.kitti import kitti_tracking_utils as ktu
from oracle import points as opoints

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'egomotion.npz'))
C = config.PYRAMID_DODT


def test_oxts_and_coordinate_transform_match_reference():
    cur, nxt = ktu.Oxts(str(G['oxts_lines'][0])), ktu.Oxts(str(G['oxts_lines'][1]))
    for o, want in ((cur, G['oxts_0']), (nxt, G['oxts_1'])):
        assert np.array_equal([o.latitude, o.longitude, o.altitude, o.roll, o.pitch, o.yaw], want)
    trans, matrix, delta = ktu.coordinate_transform(cur, nxt)
    assert np.array_equal(trans, G['trans'])
    assert np.array_equal(matrix, G['matrix'])
    assert delta == float(G['delta'])
    with pytest.raises(ValueError):
        ktu.Oxts('1 2 3')
    with pytest.raises(ValueError):
        cur.get_rotate_matrix(nxt, 'w')


def test_point_cloud_transform_matches_reference():
    for fn in (ktu.point_cloud_transform, opoints.point_cloud_transform):
        got = fn(G['xyzi'], G['trans'], G['matrix'])
        assert got.dtype == np.float32 and np.array_equal(got, G['warped_xyzi'])
    assert np.array_equal(G['xyzi'][:, 3], G['warped_xyzi'][:, 3])       # intensity untouched


def test_oracle_bev_of_registered_frame_and_unregistered_filter_grid():
    """BEV maps from the registered cloud, occupancy grid from the raw one -- bit for bit."""
    from oracle import anchors as oanchors
    imwh = tuple(int(v) for v in G['imwh'])
    warped = opoints.point_cloud_transform(G['xyzi'], G['trans'], G['matrix'])
    cloud = opoints.lidar_in_camera_view(warped, G['r0'], G['tr'], G['p2'], imwh)
    assert cloud.shape[1] == int(G['n_fov'])
    bev = opoints.bev_input(cloud, C['ground_plane'], C['area_extents'], C['voxel_size'],
                            C['height_lo'], C['height_hi'], C['num_slices'])
    want = np.zeros((700, 800, 6))
    want[G['bev_r'], G['bev_c'], G['bev_ch']] = G['bev_val']
    assert np.array_equal(bev, want)
    raw = opoints.lidar_in_camera_view(G['xyzi'], G['r0'], G['tr'], G['p2'], imwh)
    assert raw.shape[1] == int(G['n_fov_unwarped']) != cloud.shape[1]
    vox = oanchors.sliced_voxel_grid_2d(raw, C['ground_plane'], C['area_extents'],
                                        C['voxel_size'])
    occ = np.unpackbits(G['occ_bits'])[:800 * 700].reshape(800, 700)
    assert np.array_equal((np.squeeze(vox.leaf_layout_2d) + 1).astype(bool), occ.astype(bool))


def test_recovery_coordinate_matches_reference():
    got = ktu.recovery_coordinate(G['boxes'].copy(), G['r0'], G['tr'], G['trans'], G['matrix'],
                                  float(G['delta']))
    np.testing.assert_allclose(got, G['recovered'], rtol=0, atol=1e-12)
    assert np.abs(got[:, :3] - G['boxes'][:, :3]).max() > 0.1       # it did move them
    assert len(ktu.recovery_coordinate(np.zeros((0, 9)), G['r0'], G['tr'], G['trans'],
                                       G['matrix'], 0.0)) == 0
