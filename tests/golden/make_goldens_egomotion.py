#!/usr/bin/env python3
"""Golden vectors for the ego-motion registration of a pair's second frame (SURVEY A.2,
8f item 4): the reference's own Oxts / coordinate_transform / point_cloud_transform /
recovery_coordinate (avod/datasets/kitti/kitti_tracking_utils.py:129-215,
kitti_tracking_dataset.py:303-389, avod/core/dt_evaluator_utils.py:189-210) on the tracking
sequence its tests bundle (video 0000, frames 3 -> 5, tau = 2), run in the build container.

Run:  python tests/golden/make_goldens_egomotion.py     (needs /root/reference; writes egomotion.npz)

KittiTrackingDataset.__init__ needs the protobuf config system (not buildable here): its
methods are called unbound on a plain object that holds the two directories they read.
Stored: the two OXTS records, (trans, matrix, delta); frame 5's raw cloud (every 4th return),
its registered float32 cloud, the BEV maps of the registered cloud (sparse), the anchor-filter
occupancy of the UN-registered cloud -- the reference re-reads the raw file for that grid
(kitti_tracking_utils.py:98-126) --; seeded boxes through recovery_coordinate.
"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402
import make_goldens_box4ca as mb  # noqa: E402


def main():
    mb.import_evaluator()            # tensorflow / tensorflow.contrib stand-ins
    from avod.datasets.kitti.kitti_tracking_dataset import KittiTrackingDataset as DS
    from avod.datasets.kitti import kitti_tracking_utils as ktu
    from avod.core import dt_evaluator_utils
    from avod.core.bev_generators.bev_slices import BevSlices
    from wavedata.tools.core import calib_utils
    from wavedata.tools.core.voxel_grid_2d import VoxelGrid2D
    from wavedata.tools.obj_detection import obj_utils, tracking_utils

    root = os.path.join(mg.REF, 'avod/tests/datasets/Kitti/tracking/training')
    names = ['000003', '000005']
    ds = types.SimpleNamespace(oxts_dir=root + '/oxts', calib_dir=root + '/calib',
                               bev_source='lidar')
    ds.get_oxts = lambda n: DS.get_oxts(ds, n)
    ds.coordinate_transform = lambda n: DS.coordinate_transform(ds, n)
    ds.recovery_t = lambda *a: DS.recovery_t(ds, *a)
    ds.kitti_utils = types.SimpleNamespace(
        get_calib=lambda src, name: calib_utils.read_tracking_calibration(ds.calib_dir,
                                                                          int(name[:2])))
    trans, matrix, delta = ds.coordinate_transform(names)
    out = {'trans': trans, 'matrix': matrix, 'delta': np.float64(delta)}
    for i, n in enumerate(names):
        o = ds.get_oxts(n)
        out['oxts_%d' % i] = np.array([o.latitude, o.longitude, o.altitude, o.roll, o.pitch, o.yaw])
    out['oxts_lines'] = np.array([open(root + '/oxts/0000.txt').read().splitlines()[k]
                                  for k in (3, 5)])

    # ---- frame 5: raw cloud -> registered cloud -> camera view -> BEV maps ---------------
    raw = [tracking_utils.get_raw_lidar_point_cloud(n, root + '/velodyne')[:, ::4].copy()
           for n in names]                                      # (4, N) float32 each
    xyzi1 = np.ascontiguousarray(raw[1].T).copy()               # before the transform
    warped = DS.point_cloud_transform(ds, [raw[0], raw[1]], names)[1]      # (4, N) float32
    assert warped.dtype == np.float32
    im_wh = mg.png_size(root + '/image_2/0000/000005.png')
    cloud = tracking_utils.get_lidar_in_camera_view(warped, names[1], ds.calib_dir,
                                                    im_size=list(im_wh))
    ku = types.SimpleNamespace(area_extents=mg.AREA_EXTENTS, voxel_size=mg.VOXEL_SIZE)
    ku.create_slice_filter = lambda pc, ext, plane, lo, hi: np.logical_xor(
        obj_utils.get_point_filter(pc, ext, plane, hi), obj_utils.get_point_filter(pc, ext, plane, lo))
    plane = np.asarray([0, -1, 0, 1.65]) / np.linalg.norm([0, -1, 0])
    bev = BevSlices(mg._Cfg(), ku).generate_bev('lidar', cloud, plane, mg.AREA_EXTENTS,
                                                mg.VOXEL_SIZE)
    stack = np.dstack(bev['height_maps'] + [bev['density_map']])
    r, c, ch = np.nonzero(stack)
    # ---- the anchor filter's grid: un-registered cloud (get_lidar_point_cloud re-reads it) --
    calib = calib_utils.read_tracking_calibration(ds.calib_dir, 0)
    un = tracking_utils.get_lidar_in_camera_view(np.ascontiguousarray(xyzi1.T), names[1],
                                                 ds.calib_dir, im_size=list(im_wh))
    sf = ku.create_slice_filter(un, mg.AREA_EXTENTS, plane, 0.2, 2.0)
    vg = VoxelGrid2D()
    vg.voxelize_2d(un.T[sf], mg.VOXEL_SIZE, extents=mg.AREA_EXTENTS, ground_plane=plane,
                   create_leaf_layout=True)
    out.update(xyzi=xyzi1, warped_xyzi=np.ascontiguousarray(warped.T), imwh=np.asarray(im_wh),
               p2=calib.p2, r0=calib.r0_rect, tr=calib.tr_velodyne_to_cam,
               n_fov=np.int64(cloud.shape[1]), n_fov_unwarped=np.int64(un.shape[1]),
               bev_r=r.astype(np.int16), bev_c=c.astype(np.int16), bev_ch=ch.astype(np.int8),
               bev_val=stack[r, c, ch],
               occ_bits=np.packbits((np.squeeze(vg.leaf_layout_2d) + 1).astype(bool)))

    # ---- recovery_coordinate on seeded boxes ---------------------------------------------------
    rng = np.random.default_rng(20261005)
    n = 24
    boxes = np.stack([rng.uniform(-30, 30, n), rng.uniform(1.2, 1.9, n), rng.uniform(4, 66, n),
                      rng.uniform(3.0, 4.8, n), rng.uniform(1.4, 1.9, n), rng.uniform(1.3, 1.8, n),
                      rng.uniform(-np.pi, np.pi, n), rng.uniform(0, 1, n), np.zeros(n)], 1)
    ds_eval = types.SimpleNamespace(coordinate_transform=ds.coordinate_transform,
                                    kitti_utils=ds.kitti_utils, bev_source='lidar',
                                    recovery_t=ds.recovery_t)
    rec = dt_evaluator_utils.recovery_coordinate(ds_eval, names, boxes.copy())
    out.update(boxes=boxes, recovered=rec)
    np.savez_compressed(os.path.join(mg.HERE, 'egomotion.npz'), **out)
    print('trans', trans, 'delta', delta, 'fov warped', cloud.shape[1], 'unwarped', un.shape[1],
          'bev nnz', len(r), 'occupied', int((np.squeeze(vg.leaf_layout_2d) + 1).sum()))


if __name__ == '__main__':
    main()
