"""BevSlices on the GPU -- drop-in for avod/core/bev_generators/bev_slices.py.

Same constructor and generate_bev signature as the reference class
(bev_slices.py:14-55); the work is done by dodt_bev_slices (HIP).  Differences a
caller can observe: the maps come back as float32 values (the reference returns
float64 and casts to float32 when feeding the network), and an empty density
slice yields an all-zero map where the reference raises IndexError
(voxel_grid_2d.py:100-102).
"""
import numpy as np

from dodt_amd import _lib, device, ops


class BevSlices(object):

    NORM_VALUES = {'lidar': np.log(16)}

    def __init__(self, config, kitti_utils=None, ctx=None):
        """config needs height_lo, height_hi, num_slices (attributes, like the
        protobuf message, or dict keys)."""
        get = (lambda k: config[k]) if isinstance(config, dict) else \
            (lambda k: getattr(config, k))
        self.height_lo = get('height_lo')
        self.height_hi = get('height_hi')
        self.num_slices = get('num_slices')
        self.kitti_utils = kitti_utils
        self.height_per_division = \
            (self.height_hi - self.height_lo) / self.num_slices
        self._ctx = ctx

    def _context(self):
        if self._ctx is None:
            self._ctx = device.default_context()
        return self._ctx

    def generate_bev(self, source, point_cloud, ground_plane, area_extents,
                     voxel_size):
        """point_cloud (3, N) in the rectified camera frame, as in the reference.

        Returns dict(height_maps=[num_slices x (Z, X)], density_map=(Z, X))."""
        if source not in self.NORM_VALUES:
            raise KeyError(source)
        pc = np.ascontiguousarray(point_cloud, dtype=np.float64)
        if pc.ndim != 2 or pc.shape[0] != 3:
            raise ValueError("Points have the wrong shape: {}".format(
                np.transpose(pc).shape))
        ext = np.asarray(area_extents, dtype=np.float64)
        if ext.shape != (3, 2):
            raise ValueError("Extents are the wrong shape {}".format(ext.shape))
        ctx = self._context()
        cfg = dict(num_slices=self.num_slices, height_lo=self.height_lo,
                   height_hi=self.height_hi, ground_plane=ground_plane,
                   area_extents=ext, voxel_size=voxel_size,
                   anchor_filter_lo=0.2, anchor_filter_hi=2.0)
        bp = ops.make_bev_params(cfg, point_format=_lib.PTS_CAM_3XN)
        nx = int(np.ceil(ext[0, 1] / voxel_size - 1) - np.floor(ext[0, 0] / voxel_size) + 1)
        nz = int(np.ceil(ext[2, 1] / voxel_size - 1) - np.floor(ext[2, 0] / voxel_size) + 1)
        n = pc.shape[1]
        d_pts = ctx.array(pc) if n else None
        d_out = ctx.empty((nz, nx, self.num_slices + 1), np.float32)
        ops.bev_slices(ctx, d_pts, n, bp, d_out)
        if ops.bev_status(ctx) & 1:
            raise ValueError("Extents are smaller than the point cloud's voxels")
        stack = d_out.download()
        return dict(
            height_maps=[np.ascontiguousarray(stack[:, :, s])
                         for s in range(self.num_slices)],
            density_map=np.ascontiguousarray(stack[:, :, self.num_slices]))
