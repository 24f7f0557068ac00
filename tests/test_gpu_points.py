"""HIP voxeliser / anchor filter / projections against the oracle and the golden
vectors the reference produced.  Needs an MI355X."""
import os

import numpy as np
import pytest

from dodt_amd import config as cfg
from dodt_amd import _lib, device, ops, synth
from dodt_amd.core import anchor_filter as gpu_anchor_filter
from dodt_amd.core import anchor_projector as gpu_projector
from dodt_amd.core.anchor_generators import grid_anchor_3d_generator as gen
from dodt_amd.core.bev_generators.bev_slices import BevSlices
from oracle import anchors as oanchors
from oracle import boxes as oboxes
from oracle import points as opoints

pytestmark = pytest.mark.gpu
C = cfg.PYRAMID_DODT
FRAMES = ['obj000001', 'obj000217d4', 'trk0000_000003d4', 'trk0001_000005d4']
EDGES = ['one_per_slice', 'ties_one_slice', 'extent_edges', 'dense_random']


@pytest.fixture(scope='module')
def ctx():
    return device.default_context()


@pytest.fixture(scope='module')
def frames(golden_dir):
    return np.load(os.path.join(golden_dir, 'frames.npz'))


@pytest.fixture(scope='module')
def edges(golden_dir):
    return np.load(os.path.join(golden_dir, 'edge_clouds.npz'))


def _golden_stack(z, tag, keys=('_bev_r', '_bev_c', '_bev_ch', '_bev_val')):
    st = np.zeros((700, 800, 6))
    st[z[tag + keys[0]], z[tag + keys[1]], z[tag + keys[2]]] = z[tag + keys[3]]
    return st


def _run_raw(ctx, xyzi, r0, tr, p2, imwh):
    bp = ops.make_bev_params(C, synth.velo_to_cam(r0, tr), p2, imwh)
    d_pts = ctx.array(np.ascontiguousarray(xyzi, dtype=np.float32))
    d_out = ctx.empty((700, 800, 6), np.float32)
    d_occ = ctx.empty((700, 25), np.uint32)
    ops.bev_slices(ctx, d_pts, len(xyzi), bp, d_out, d_occ)
    assert ops.bev_status(ctx) == 0
    return d_out.download(), d_occ.download()


def _compare_bev(got, want64):
    """Cells must agree exactly; values are float32(reference float64)."""
    want = want64.astype(np.float32)
    assert np.array_equal(got != 0, want != 0), 'occupied cells differ'
    nbad = int(np.count_nonzero(got != want))
    assert nbad == 0, '%d of %d values differ (max abs %g)' % (
        nbad, np.count_nonzero(want), np.abs(got - want).max())


@pytest.mark.parametrize('tag', FRAMES)
def test_bev_from_raw_points_matches_reference(ctx, frames, tag):
    got, occ = _run_raw(ctx, frames[tag + '_xyzi'], frames[tag + '_r0'],
                        frames[tag + '_tr'], frames[tag + '_p2'], frames[tag + '_imwh'])
    _compare_bev(got, _golden_stack(frames, tag))
    # occupancy of the [0.2, 2.0) slice = the reference's leaf layout + 1
    want = np.unpackbits(frames[tag + '_occ_bits'])[:800 * 700].reshape(800, 700)
    assert np.array_equal(occ, gpu_anchor_filter.pack_occupancy(want.astype(bool)))


@pytest.mark.parametrize('tag', EDGES)
def test_bev_generate_bev_dropin_on_edge_clouds(edges, tag):
    """The avod.core-shaped entry point, camera-frame (3,N) float64 input."""
    gen_ = BevSlices(dict(height_lo=C['height_lo'], height_hi=C['height_hi'],
                          num_slices=C['num_slices']))
    maps = gen_.generate_bev('lidar', edges[tag + '_cloud'], C['ground_plane'],
                             C['area_extents'], C['voxel_size'])
    assert len(maps['height_maps']) == 5 and maps['density_map'].shape == (700, 800)
    got = np.dstack(maps['height_maps'] + [maps['density_map']])
    _compare_bev(got, _golden_stack(edges, tag, ('_r', '_c', '_ch', '_val')))


def test_bev_rejects_bad_input():
    gen_ = BevSlices(dict(height_lo=C['height_lo'], height_hi=C['height_hi'],
                          num_slices=C['num_slices']))
    with pytest.raises(ValueError):
        gen_.generate_bev('lidar', np.zeros((4, 10)), C['ground_plane'],
                          C['area_extents'], C['voxel_size'])
    with pytest.raises(ValueError):      # extents of the wrong shape
        gen_.generate_bev('lidar', np.zeros((3, 10)), C['ground_plane'],
                          [[-40, 40], [0, 70]], C['voxel_size'])


def test_bev_synthetic_full_size_matches_oracle(ctx):
    """120k-point synthetic KITTI-shaped frame (the bench workload)."""
    xyzi = synth.lidar_frame(0, 0)
    got, occ = _run_raw(ctx, xyzi, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2,
                        synth.IMAGE_WH)
    cloud = opoints.lidar_in_camera_view(xyzi, synth.R0_RECT, synth.TR_VELO_TO_CAM,
                                         synth.P2, synth.IMAGE_WH)
    assert 10000 < cloud.shape[1] < 40000
    want = opoints.bev_input(cloud, C['ground_plane'], C['area_extents'],
                             C['voxel_size'], C['height_lo'], C['height_hi'],
                             C['num_slices'])
    _compare_bev(got, want)
    vox = oanchors.sliced_voxel_grid_2d(cloud, C['ground_plane'], C['area_extents'],
                                        C['voxel_size'])
    occ_want = (np.squeeze(vox.leaf_layout_2d) + 1).astype(bool)
    assert np.array_equal(occ, gpu_anchor_filter.pack_occupancy(occ_want))


def test_bev_dense_300k_matches_oracle(ctx):
    """config 5 stress shape: 300k points, 40 boxes."""
    xyzi = synth.lidar_frame(3, 1, n_points=300000, n_boxes=40)
    got, _ = _run_raw(ctx, xyzi, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2,
                      synth.IMAGE_WH)
    cloud = opoints.lidar_in_camera_view(xyzi, synth.R0_RECT, synth.TR_VELO_TO_CAM,
                                         synth.P2, synth.IMAGE_WH)
    want = opoints.bev_input(cloud, C['ground_plane'], C['area_extents'],
                             C['voxel_size'], C['height_lo'], C['height_hi'],
                             C['num_slices'])
    _compare_bev(got, want)


def test_bev_is_deterministic(ctx):
    xyzi = synth.lidar_frame(1, 2)
    a, _ = _run_raw(ctx, xyzi, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2, synth.IMAGE_WH)
    perm = np.random.default_rng(0).permutation(len(xyzi))
    b, _ = _run_raw(ctx, xyzi, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2, synth.IMAGE_WH)
    assert np.array_equal(a, b)
    # density and occupancy do not depend on point order; heights may
    c, _ = _run_raw(ctx, xyzi[perm], synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2,
                    synth.IMAGE_WH)
    assert np.array_equal(a[:, :, 5], c[:, :, 5])
    assert np.array_equal(a != 0, c != 0)


def test_bev_empty_cloud(ctx):
    got, occ = _run_raw(ctx, np.zeros((0, 4), np.float32), synth.R0_RECT,
                        synth.TR_VELO_TO_CAM, synth.P2, synth.IMAGE_WH)
    # every slice has <= 1 member: the origin-point substitute shows up
    np.testing.assert_allclose(got[699, 400, :5], [3.7, 2.7, 1.7, 0.7, -0.3], atol=1e-5)
    assert np.count_nonzero(got) == 5 and not occ.any()


@pytest.mark.parametrize('tag', FRAMES)
def test_anchor_filter_matches_reference(ctx, frames, tag):
    boxes = gen.tile_anchors_3d(C['area_extents'], C['anchor_sizes'],
                                C['anchor_stride'], C['ground_plane'])
    assert np.array_equal(boxes, oanchors.tile_anchors_3d(
        C['area_extents'], C['anchor_sizes'], C['anchor_stride'], C['ground_plane']))
    anchors = gen.box_3d_to_anchor(boxes)
    occ = np.unpackbits(frames[tag + '_occ_bits'])[:800 * 700].reshape(800, 700).astype(bool)
    mask = gpu_anchor_filter.get_empty_anchor_filter_2d(
        anchors, occ, C['area_extents'], C['voxel_size'], ctx=ctx)
    want = np.unpackbits(frames[tag + '_anchor_bits'])[:len(mask)].astype(bool)
    assert np.array_equal(mask, want)
    assert mask.sum() == int(frames[tag + '_n_anchors'])


def test_anchor_filter_device_compaction_order(ctx, frames):
    """Kept indices come back ascending = generator order (bit-exact anchor
    indices downstream)."""
    tag = 'obj000001'
    boxes = gen.tile_anchors_3d(C['area_extents'], C['anchor_sizes'],
                                C['anchor_stride'], C['ground_plane'])
    anchors = gen.box_3d_to_anchor(boxes)
    cells, nx, nz = gen.anchor_grid_cells(anchors, C['area_extents'], C['voxel_size'])
    occ = np.unpackbits(frames[tag + '_occ_bits'])[:800 * 700].reshape(800, 700).astype(bool)
    d_keep = ctx.empty((len(anchors),), np.int32)
    d_cnt = ctx.zeros((1,), np.int32)
    ops.anchor_filter(ctx, ctx.array(gpu_anchor_filter.pack_occupancy(occ)), nx, nz,
                      ctx.array(cells), len(anchors), d_keep, d_cnt)
    cnt = int(d_cnt.download()[0])
    keep = d_keep.download()[:cnt]
    want = np.nonzero(np.unpackbits(frames[tag + '_anchor_bits'])[:len(anchors)])[0]
    assert np.array_equal(keep, want)
    # device-side projection of the kept rows
    d_anchors = ctx.array(anchors)
    d_bev = ctx.empty((cnt, 4), np.float32)
    d_img = ctx.empty((cnt, 4), np.float32)
    d_a32 = ctx.empty((cnt, 6), np.float32)
    imwh = frames[tag + '_imwh']
    ops.project_anchors_f64(ctx, d_anchors, d_keep, cnt, d_cnt,
                            C['bev_extents'].reshape(-1), frames[tag + '_p2'], imwh,
                            d_bev, d_img, d_a32)
    got_bev, got_img = d_bev.download(), d_img.download()
    want_bev = frames[tag + '_bev_norm'].astype(np.float32)[:, [1, 0, 3, 2]]
    want_img = frames[tag + '_img_norm'][:, [1, 0, 3, 2]]
    assert np.array_equal(got_bev[:256], want_bev)
    assert np.array_equal(got_img[:256], want_img)
    assert np.array_equal(d_a32.download()[:256],
                          frames[tag + '_kept256'].astype(np.float32))


@pytest.mark.parametrize('tag', FRAMES)
def test_projector_dropins(frames, tag):
    kept = frames[tag + '_kept256']
    imwh = frames[tag + '_imwh']
    c, n = gpu_projector.project_to_bev(kept, C['bev_extents'])
    np.testing.assert_allclose(c, frames[tag + '_bev_corners'], atol=1e-4)
    np.testing.assert_allclose(n, frames[tag + '_bev_norm'], atol=1e-6)
    ic, inorm = gpu_projector.project_to_image_space(kept, frames[tag + '_p2'],
                                                     [imwh[1], imwh[0]])
    assert np.array_equal(inorm, frames[tag + '_img_norm'])
    np.testing.assert_allclose(ic, frames[tag + '_img_corners'], rtol=1e-6)
    ic32, in32 = gpu_projector.tf_project_to_image_space(
        kept, frames[tag + '_p2'], [imwh[1], imwh[0]])
    _, want32 = oboxes.project_to_image_space(kept, frames[tag + '_p2'],
                                              [imwh[1], imwh[0]], dtype=np.float32)
    np.testing.assert_allclose(in32, want32, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        gpu_projector.project_to_image_space(np.zeros((3, 5)), frames[tag + '_p2'],
                                             [375, 1242])


def test_bev_of_ego_motion_registered_frame_matches_reference(ctx, golden_dir):
    """The second frame of a tracking pair (video 0000, frames 3 -> 5): dodt_bev_slices with
    the ego-motion pre-transform against what the reference's point_cloud_transform ->
    get_lidar_in_camera_view -> generate_bev produced (tests/golden/make_goldens_egomotion.py):
    BEV maps of the registered cloud exact, occupancy bits of the UN-registered cloud exact
    (kitti_tracking_utils.py:98-126: the reference re-reads the raw file for the filter)."""
    g = np.load(os.path.join(golden_dir, 'egomotion.npz'))
    bp = ops.with_ego_motion(
        ops.make_bev_params(C, synth.velo_to_cam(g['r0'], g['tr']), g['p2'], g['imwh']),
        g['trans'], g['matrix'])
    d_out = ctx.empty((700, 800, 6), np.float32)
    d_occ = ctx.empty((700, 25), np.uint32)
    ops.bev_slices(ctx, ctx.array(g['xyzi']), len(g['xyzi']), bp, d_out, d_occ)
    assert ops.bev_status(ctx) == 0
    want = np.zeros((700, 800, 6))
    want[g['bev_r'], g['bev_c'], g['bev_ch']] = g['bev_val']
    _compare_bev(d_out.download(), want)
    occ = np.unpackbits(g['occ_bits'])[:800 * 700].reshape(800, 700)
    assert np.array_equal(d_occ.download(), gpu_anchor_filter.pack_occupancy(occ.astype(bool)))
    # without the pre-transform the maps differ (the registration moves ~0.64 m)
    ops.bev_slices(ctx, ctx.array(g['xyzi']), len(g['xyzi']),
                   ops.make_bev_params(C, synth.velo_to_cam(g['r0'], g['tr']), g['p2'], g['imwh']),
                   d_out, d_occ)
    assert np.count_nonzero(d_out.download() != want.astype(np.float32)) > 1000
    assert np.array_equal(d_occ.download(), gpu_anchor_filter.pack_occupancy(occ.astype(bool)))
    # camera-frame input cannot be registered in the velodyne frame
    bad = ops.with_ego_motion(ops.make_bev_params(C, point_format=1), g['trans'], g['matrix'])
    with pytest.raises(ValueError):
        ops.bev_slices(ctx, ctx.array(np.zeros((3, 8))), 8, bad, d_out)
