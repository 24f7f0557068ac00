#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's
own numpy code (imported read-only from /root/reference) in the build container.

Run:  python tests/golden/make_goldens.py
Needs /root/reference; it is NOT needed (and not available) on the GPU box --
only the .npz files this script writes travel there.

The reference modules import tensorflow / cv2 at module scope although the
numpy functions used here never touch them, so two inert stand-in modules are
registered first.  Nothing from TensorFlow is executed.

What is stored is data only: inputs (point clouds that the reference's tests
ship, calibration numbers, seeded random boxes) and the outputs the reference
computed for them.
"""
import os
import struct
import sys
from unittest.mock import MagicMock

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    tf = MagicMock()

    class _Tensor(object):
        pass
    tf.Tensor = _Tensor
    sys.modules['tensorflow'] = tf
    sys.modules['cv2'] = MagicMock()
    sys.path[:0] = [REF, os.path.join(REF, 'wavedata')]


def png_size(path):
    with open(path, 'rb') as f:
        head = f.read(24)
    w, h = struct.unpack('>II', head[16:24])
    return w, h


# float32-rounded config scalars, exactly what protobuf hands to python
# (avod/configs/pyramid_cars_with_aug_dt_5_tracking.config:167-176)
VOXEL_SIZE = float(np.float32(0.1))
HEIGHT_LO = float(np.float32(-0.2))
HEIGHT_HI = float(np.float32(2.3))
NUM_SLICES = 5
AREA_EXTENTS = np.array([[-40., 40.], [-5., 3.], [0., 70.]])
BEV_EXTENTS = AREA_EXTENTS[[0, 2]]
ANCHOR_STRIDE = [float(np.float32(0.5)), float(np.float32(0.5))]
# Two car size clusters (l, w, h) -- the published AVOD car clusters; the
# reference derives them from the training labels, which are not needed here.
CAR_CLUSTERS = [[3.514, 1.581, 1.511], [4.236, 1.653, 1.547]]


class _Cfg(object):
    height_lo = HEIGHT_LO
    height_hi = HEIGHT_HI
    num_slices = NUM_SLICES


def main():
    _import_reference()
    from wavedata.tools.core import calib_utils
    from wavedata.tools.core.voxel_grid_2d import VoxelGrid2D
    from wavedata.tools.obj_detection import obj_utils
    from avod.core.bev_generators.bev_slices import BevSlices
    from avod.core import anchor_projector, box_3d_encoder, anchor_filter
    from avod.core import anchor_encoder, box_4c_encoder
    from avod.core.anchor_generators import grid_anchor_3d_generator

    class _KU(object):
        """The two KittiUtils methods the path uses (kitti_utils.py:81-109),
        bound to a plain object because KittiUtils.__init__ needs the protobuf
        config system (not buildable here)."""
        area_extents = AREA_EXTENTS
        voxel_size = VOXEL_SIZE

        def create_slice_filter(self, pc, ext, plane, lo, hi):
            a = obj_utils.get_point_filter(pc, ext, plane, hi)
            b = obj_utils.get_point_filter(pc, ext, plane, lo)
            return np.logical_xor(a, b)

    ku = _KU()
    bev_gen = BevSlices(_Cfg(), ku)
    plane = np.asarray([0, -1, 0, 1.65]) / np.linalg.norm([0, -1, 0])

    def run_frame(xyzi, calib, im_wh, tag, out):
        """reference path: raw points -> fov cloud -> bev maps -> anchor mask
        -> projections."""
        pts = calib_utils.lidar_to_cam_frame(xyzi[:, :3], calib)
        front = pts[:, 2] > 0
        ptsf = pts[front]
        uv = calib_utils.project_to_image(ptsf.T, p=calib.p2).T
        imf = (uv[:, 0] > 0) & (uv[:, 0] < im_wh[0]) & \
              (uv[:, 1] > 0) & (uv[:, 1] < im_wh[1])
        cloud = ptsf[imf].T                                    # (3, N_fov)
        keep = np.zeros(len(xyzi), dtype=bool)
        keep[np.nonzero(front)[0][imf]] = True

        bev = bev_gen.generate_bev('lidar', cloud, plane, AREA_EXTENTS,
                                   VOXEL_SIZE)
        stack = np.dstack(bev['height_maps'] + [bev['density_map']])
        r, c, ch = np.nonzero(stack)

        # anchor grid + empty filter (dt_rpn_model.py:905-952)
        boxes = grid_anchor_3d_generator.tile_anchors_3d(
            AREA_EXTENTS, CAR_CLUSTERS, ANCHOR_STRIDE, plane)
        sf = ku.create_slice_filter(cloud, AREA_EXTENTS, plane, 0.2, 2.0)
        vg = VoxelGrid2D()
        vg.voxelize_2d(cloud.T[sf], VOXEL_SIZE, extents=AREA_EXTENTS,
                       ground_plane=plane, create_leaf_layout=True)
        anchors = box_3d_encoder.box_3d_to_anchor(boxes)
        amask = anchor_filter.get_empty_anchor_filter_2d(anchors, vg, 1)
        kept = anchors[amask]
        first = kept[:256]
        bev_c, bev_n = anchor_projector.project_to_bev(first, BEV_EXTENTS)
        img_c, img_n = anchor_projector.project_to_image_space(
            first, calib.p2, [im_wh[1], im_wh[0]])

        out[tag + '_xyzi'] = xyzi
        out[tag + '_p2'] = calib.p2
        out[tag + '_r0'] = calib.r0_rect
        out[tag + '_tr'] = calib.tr_velodyne_to_cam
        out[tag + '_imwh'] = np.asarray(im_wh, dtype=np.int32)
        out[tag + '_fov_bits'] = np.packbits(keep)
        out[tag + '_n_fov'] = np.int64(cloud.shape[1])
        out[tag + '_fov_sum'] = cloud.sum(axis=1)
        out[tag + '_bev_r'] = r.astype(np.int16)
        out[tag + '_bev_c'] = c.astype(np.int16)
        out[tag + '_bev_ch'] = ch.astype(np.int8)
        out[tag + '_bev_val'] = stack[r, c, ch]
        out[tag + '_occ_bits'] = np.packbits(
            (np.squeeze(vg.leaf_layout_2d) + 1).astype(bool))
        out[tag + '_anchor_bits'] = np.packbits(amask)
        out[tag + '_n_anchors'] = np.int64(amask.sum())
        out[tag + '_kept256'] = first
        out[tag + '_bev_corners'] = bev_c
        out[tag + '_bev_norm'] = bev_n
        out[tag + '_img_corners'] = img_c
        out[tag + '_img_norm'] = img_n
        print(tag, 'raw', len(xyzi), 'fov', cloud.shape[1], 'bev nnz', len(r),
              'anchors kept', int(amask.sum()))

    kitti = os.path.join(REF, 'avod/tests/datasets/Kitti')
    out = {}

    def load_object(idx, step):
        d = os.path.join(kitti, 'object/training')
        calib = calib_utils.read_calibration(d + '/calib', idx)
        xyzi = np.fromfile(d + '/velodyne/%06d.bin' % idx,
                           dtype=np.float32).reshape(-1, 4)[::step]
        wh = png_size(d + '/image_2/%06d.png' % idx)
        return np.ascontiguousarray(xyzi), calib, wh

    def load_tracking(video, idx, step):
        d = os.path.join(kitti, 'tracking/training')
        calib = calib_utils.read_tracking_calibration(d + '/calib', video)
        xyzi = np.fromfile(d + '/velodyne/%04d/%06d.bin' % (video, idx),
                           dtype=np.float32).reshape(-1, 4)[::step]
        wh = png_size(d + '/image_2/%04d/%06d.png' % (video, idx))
        return np.ascontiguousarray(xyzi), calib, wh

    # one full-size real frame + three decimated ones (every 4th return)
    run_frame(*load_object(1, 1), tag='obj000001', out=out)
    run_frame(*load_object(217, 4), tag='obj000217d4', out=out)
    run_frame(*load_tracking(0, 3, 4), tag='trk0000_000003d4', out=out)
    run_frame(*load_tracking(1, 5, 4), tag='trk0001_000005d4', out=out)
    np.savez_compressed(os.path.join(HERE, 'frames.npz'), **out)

    # ---- synthetic edge-case clouds, already in the camera frame ----------
    edge = {}
    rng = np.random.default_rng(7)

    def run_cloud(cloud, tag):
        bev = bev_gen.generate_bev('lidar', cloud, plane, AREA_EXTENTS,
                                   VOXEL_SIZE)
        stack = np.dstack(bev['height_maps'] + [bev['density_map']])
        r, c, ch = np.nonzero(stack)
        edge[tag + '_cloud'] = cloud
        edge[tag + '_r'] = r.astype(np.int16)
        edge[tag + '_c'] = c.astype(np.int16)
        edge[tag + '_ch'] = ch.astype(np.int8)
        edge[tag + '_val'] = stack[r, c, ch]
        print(tag, cloud.shape, 'nnz', len(r))

    # (a) five points, one per slice: every slice has <= 1 member, so each
    #     falls back to the origin point (bev_slices.py:76-99)
    h = np.array([0.0, 0.4, 0.9, 1.4, 1.9])
    run_cloud(np.vstack([np.linspace(-3, 3, 5), 1.65 - h,
                         np.linspace(5, 25, 5)]), 'one_per_slice')
    # (b) all points in one slice, several per cell, including exact ties in
    #     the y-bin (first-in-order must win) and points on bin edges
    xs = np.repeat(np.array([0.05, 0.05, 0.05, 1.0, 1.0, -2.25]), 1)
    ys = 1.65 - np.array([0.52, 0.58, 0.55, 0.61, 0.69, 0.50])
    zs = np.array([10.05, 10.05, 10.05, 20.0, 20.0, 30.0])
    run_cloud(np.vstack([xs, ys, zs]), 'ties_one_slice')
    # (c) points exactly on / just outside the strict extents and on slice
    #     boundaries
    xs = np.array([-40.0, -39.999, 39.999, 40.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0])
    ys = np.array([1.0, 1.0, 1.0, 1.0, 1.65 - HEIGHT_LO, 1.65 - HEIGHT_HI,
                   1.65 - 0.3, 1.65 - 0.3, -5.0, 3.0])
    zs = np.array([10.0, 10.0, 10.0, 10.0, 5.0, 6.0, 0.0, 70.0, 7.0, 8.0])
    xs = np.concatenate([xs, rng.uniform(-39, 39, 40)])
    ys = np.concatenate([ys, 1.65 - rng.uniform(-0.19, 2.29, 40)])
    zs = np.concatenate([zs, rng.uniform(1, 69, 40)])
    run_cloud(np.vstack([xs, ys, zs]), 'extent_edges')
    # (d) dense random cloud, many points per cell
    n = 4000
    run_cloud(np.vstack([rng.uniform(-6, 6, n), 1.65 - rng.uniform(-0.5, 2.6, n),
                         rng.uniform(2, 14, n)]), 'dense_random')
    np.savez_compressed(os.path.join(HERE, 'edge_clouds.npz'), **edge)

    # ---- box encoders (numpy twins) ------------------------------------------
    enc = {}
    rng = np.random.default_rng(11)
    n = 64
    boxes_3d = np.stack([rng.uniform(-30, 30, n), rng.uniform(1.0, 2.0, n),
                         rng.uniform(5, 60, n), rng.uniform(3.0, 5.0, n),
                         rng.uniform(1.4, 2.0, n), rng.uniform(1.3, 1.9, n),
                         rng.uniform(-np.pi, np.pi, n)], axis=1)
    enc['boxes_3d'] = boxes_3d
    enc['plane'] = plane
    enc['anchor_plain'] = box_3d_encoder.box_3d_to_anchor(boxes_3d)
    enc['anchor_ortho'] = box_3d_encoder.box_3d_to_anchor(boxes_3d, True)
    b4c = np.stack([box_4c_encoder.np_box_3d_to_box_4c(b, plane)
                    for b in boxes_3d])
    enc['box_4c'] = b4c
    off = rng.normal(0, 0.15, size=b4c.shape)
    enc['offsets_4c'] = off
    enc['box_3d_from_4c'] = np.stack(
        [box_4c_encoder.np_box_4c_to_box_3d(b, plane) for b in b4c + off])
    anchors = enc['anchor_ortho']
    aoff = rng.normal(0, 0.2, size=anchors.shape)
    enc['anchor_offsets'] = aoff
    enc['regressed_anchors'] = anchor_encoder.offset_to_anchor(anchors, aoff)
    enc['box_3d_from_anchor'] = box_3d_encoder.anchors_to_box_3d(anchors, True)
    np.savez_compressed(os.path.join(HERE, 'encoders.npz'), **enc)

    for f in ('frames.npz', 'edge_clouds.npz', 'encoders.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    main()
