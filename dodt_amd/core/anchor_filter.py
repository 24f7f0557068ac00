"""Empty-anchor filter on the GPU -- stands where
avod/core/anchor_filter.py:64-119 get_empty_anchor_filter_2d stands."""
import numpy as np

from dodt_amd import device, ops
from dodt_amd.core.anchor_generators import grid_anchor_3d_generator as gen


def pack_occupancy(occ_xz):
    """(X, Z) boolean occupancy -> (Z, ceil(X/32)) uint32 bit rows (the layout
    dodt_bev_slices produces)."""
    occ = np.asarray(occ_xz, dtype=bool)
    nx, nz = occ.shape
    words = (nx + 31) // 32
    padded = np.zeros((nz, words * 32), dtype=np.uint8)
    padded[:, :nx] = occ.T
    return np.packbits(padded.reshape(nz, words, 32), axis=2,
                       bitorder='little').view(np.uint32).reshape(nz, words)


def get_empty_anchor_filter_2d(anchors, occupancy_xz, area_extents, voxel_size,
                               density_threshold=1, ctx=None):
    """anchors (N,6) float64; occupancy_xz (X,Z) bool (the reference passes a
    VoxelGrid2D whose leaf layout + 1 is this array).  -> (N,) bool mask."""
    anchors = np.asarray(anchors)
    if anchors.ndim != 2 or anchors.shape[1] != 6:
        raise TypeError('Invalid anchor format')
    ctx = ctx or device.default_context()
    cells, nx, nz = gen.anchor_grid_cells(anchors, area_extents, voxel_size)
    n = len(anchors)
    d_occ = ctx.array(pack_occupancy(occupancy_xz))
    d_cells = ctx.array(cells)
    d_keep = ctx.empty((max(n, 1),), np.int32)
    d_cnt = ctx.zeros((1,), np.int32)
    ops.anchor_filter(ctx, d_occ, nx, nz, d_cells, n, d_keep, d_cnt,
                      density_threshold)
    cnt = int(d_cnt.download()[0])
    mask = np.zeros(n, dtype=bool)
    mask[d_keep.download()[:cnt]] = True
    return mask
