"""The numpy oracle against golden vectors produced by the reference's own code
(tests/golden/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import anchors as oanchors
from oracle import boxes as oboxes
from oracle import points as opoints
from dodt_amd import config as cfg

FRAMES = ['obj000001', 'obj000217d4', 'trk0000_000003d4', 'trk0001_000005d4']
EDGES = ['one_per_slice', 'ties_one_slice', 'extent_edges', 'dense_random']
C = cfg.PYRAMID_DODT


@pytest.fixture(scope='module')
def frames(golden_dir):
    return np.load(os.path.join(golden_dir, 'frames.npz'))


@pytest.fixture(scope='module')
def edges(golden_dir):
    return np.load(os.path.join(golden_dir, 'edge_clouds.npz'))


@pytest.fixture(scope='module')
def enc(golden_dir):
    return np.load(os.path.join(golden_dir, 'encoders.npz'))


def _sparse(stack):
    r, c, ch = np.nonzero(stack)
    return r, c, ch, stack[r, c, ch]


def _cloud(frames, tag):
    return opoints.lidar_in_camera_view(
        frames[tag + '_xyzi'], frames[tag + '_r0'], frames[tag + '_tr'],
        frames[tag + '_p2'], frames[tag + '_imwh'])


@pytest.mark.parametrize('tag', FRAMES)
def test_fov_filter_matches_reference(frames, tag):
    xyzi = frames[tag + '_xyzi']
    keep, _ = opoints.fov_mask(xyzi, frames[tag + '_r0'], frames[tag + '_tr'],
                               frames[tag + '_p2'], frames[tag + '_imwh'])
    want = np.unpackbits(frames[tag + '_fov_bits'])[:len(xyzi)].astype(bool)
    assert np.array_equal(keep, want)
    cloud = _cloud(frames, tag)
    assert cloud.shape[1] == int(frames[tag + '_n_fov'])
    np.testing.assert_array_equal(cloud.sum(axis=1), frames[tag + '_fov_sum'])


@pytest.mark.parametrize('tag', FRAMES)
def test_bev_maps_match_reference_bit_exact(frames, tag):
    stack = opoints.bev_input(_cloud(frames, tag), C['ground_plane'],
                              C['area_extents'], C['voxel_size'],
                              C['height_lo'], C['height_hi'], C['num_slices'])
    assert stack.shape == (700, 800, 6) and stack.dtype == np.float64
    r, c, ch, v = _sparse(stack)
    assert np.array_equal(r, frames[tag + '_bev_r'])
    assert np.array_equal(c, frames[tag + '_bev_c'])
    assert np.array_equal(ch, frames[tag + '_bev_ch'])
    assert np.array_equal(v, frames[tag + '_bev_val'])      # float64, exact


@pytest.mark.parametrize('tag', EDGES)
def test_bev_edge_clouds_match_reference(edges, tag):
    stack = opoints.bev_input(edges[tag + '_cloud'], C['ground_plane'],
                              C['area_extents'], C['voxel_size'],
                              C['height_lo'], C['height_hi'], C['num_slices'])
    r, c, ch, v = _sparse(stack)
    assert np.array_equal(r, edges[tag + '_r'])
    assert np.array_equal(c, edges[tag + '_c'])
    assert np.array_equal(ch, edges[tag + '_ch'])
    assert np.array_equal(v, edges[tag + '_val'])


def test_one_point_slices_fall_back_to_origin(edges):
    """bev_slices.py:76-99 quirk: pixel (699, 400) gets (1.65 - lo_s) / w."""
    stack = opoints.bev_input(edges['one_per_slice_cloud'], C['ground_plane'],
                              C['area_extents'], C['voxel_size'],
                              C['height_lo'], C['height_hi'], C['num_slices'])
    got = stack[699, 400, :5]
    np.testing.assert_allclose(got, [3.7, 2.7, 1.7, 0.7, -0.3], atol=1e-6)


@pytest.mark.parametrize('tag', FRAMES)
def test_anchor_filter_matches_reference(frames, tag):
    cloud = _cloud(frames, tag)
    boxes = oanchors.tile_anchors_3d(C['area_extents'], C['anchor_sizes'],
                                     C['anchor_stride'], C['ground_plane'])
    assert boxes.shape == (89600, 7)
    vox = oanchors.sliced_voxel_grid_2d(cloud, C['ground_plane'],
                                        C['area_extents'], C['voxel_size'])
    occ = (np.squeeze(vox.leaf_layout_2d) + 1).astype(bool)
    want_occ = np.unpackbits(frames[tag + '_occ_bits'])[:occ.size]
    assert np.array_equal(occ.reshape(-1), want_occ.astype(bool))
    anchors = oanchors.box_3d_to_anchor(boxes)
    mask = oanchors.empty_anchor_filter_2d(anchors, vox)
    want = np.unpackbits(frames[tag + '_anchor_bits'])[:len(mask)]
    assert np.array_equal(mask, want.astype(bool))
    assert mask.sum() == int(frames[tag + '_n_anchors'])
    kept = anchors[mask][:256]
    assert np.array_equal(kept, frames[tag + '_kept256'])


@pytest.mark.parametrize('tag', FRAMES)
def test_projections_match_reference(frames, tag):
    kept = frames[tag + '_kept256']
    c, n = oboxes.project_to_bev(kept, C['bev_extents'])
    assert np.array_equal(c, frames[tag + '_bev_corners'])
    assert np.array_equal(n, frames[tag + '_bev_norm'])
    imwh = frames[tag + '_imwh']
    ic, inorm = oboxes.project_to_image_space(kept, frames[tag + '_p2'],
                                              [imwh[1], imwh[0]])
    assert ic.dtype == np.float32
    assert np.array_equal(ic, frames[tag + '_img_corners'])
    assert np.array_equal(inorm, frames[tag + '_img_norm'])
    # float32 (TF-branch) twin agrees with the numpy branch to 1e-4 relative
    ic32, in32 = oboxes.project_to_image_space(
        kept, frames[tag + '_p2'], [imwh[1], imwh[0]], dtype=np.float32)
    np.testing.assert_allclose(in32, inorm, rtol=2e-4, atol=1e-4)


def test_encoders_match_reference(enc):
    b3 = enc['boxes_3d']
    plane = enc['plane']
    np.testing.assert_array_equal(oanchors.box_3d_to_anchor(b3),
                                  enc['anchor_plain'])
    np.testing.assert_array_equal(oanchors.box_3d_to_anchor(b3, True),
                                  enc['anchor_ortho'])
    np.testing.assert_allclose(
        oboxes.box_3d_to_box_4c(b3, plane, dtype=np.float64), enc['box_4c'],
        rtol=0, atol=1e-12)
    back = oboxes.box_4c_to_box_3d(enc['box_4c'] + enc['offsets_4c'], plane,
                                   dtype=np.float64)
    np.testing.assert_allclose(back, enc['box_3d_from_4c'], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(
        oboxes.offset_to_anchor(enc['anchor_ortho'], enc['anchor_offsets']),
        enc['regressed_anchors'])
    np.testing.assert_array_equal(
        oboxes.anchors_to_box_3d(enc['anchor_ortho'], fix_lw=True),
        enc['box_3d_from_anchor'])
    # float32 twins (TF branch) within 1e-4 of the numpy branch
    np.testing.assert_allclose(
        oboxes.box_3d_to_box_4c(b3, plane, dtype=np.float32), enc['box_4c'],
        atol=1e-4)
    np.testing.assert_allclose(
        oboxes.box_4c_to_box_3d(enc['box_4c'] + enc['offsets_4c'], plane,
                                dtype=np.float32),
        enc['box_3d_from_4c'], atol=2e-4)
