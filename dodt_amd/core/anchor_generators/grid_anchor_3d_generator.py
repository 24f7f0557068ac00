"""Per-configuration anchor grid (host side, computed once and kept on the GPU).

Mirrors avod/core/anchor_generators/grid_anchor_3d_generator.py:39-108 and the
index quantisation of avod/core/anchor_filter.py:84-106 +
wavedata/.../voxel_grid_2d.py:162-186.  With the ground plane fixed
the 89 600 anchors and their grid cells are constants of the configuration, so
they are produced once here and never recomputed per frame.
"""
import numpy as np


def tile_anchors_3d(area_extents, anchor_3d_sizes, anchor_stride, ground_plane):
    """-> (N,7) float64 box_3d rows, z slowest / x / size / rotation fastest."""
    sizes = np.asarray(anchor_3d_sizes, dtype=np.float64).reshape(-1, 3)
    rots = np.array([0.0, np.pi / 2.0])
    xs = np.array(np.arange(area_extents[0][0] + anchor_stride[0] / 2.0,
                            area_extents[0][1], step=anchor_stride[0]),
                  dtype=np.float32).astype(np.float64)
    zs = np.array(np.arange(area_extents[2][1] - anchor_stride[1] / 2.0,
                            area_extents[2][0], step=-anchor_stride[1]),
                  dtype=np.float32).astype(np.float64)
    grid = np.stack(np.meshgrid(zs, xs, np.arange(len(sizes)),
                                np.arange(len(rots)), indexing='ij'),
                    axis=-1).reshape(-1, 4)
    a, b, c, d = ground_plane
    out = np.zeros((len(grid), 7))
    out[:, 0] = grid[:, 1]
    out[:, 2] = grid[:, 0]
    out[:, 1] = -(a * out[:, 0] + c * out[:, 2] + d) / b
    out[:, 3:6] = sizes[grid[:, 2].astype(np.int64)]
    out[:, 6] = rots[grid[:, 3].astype(np.int64)]
    return out


def box_3d_to_anchor(boxes_3d, ortho_rotate=False):
    """avod/core/box_3d_encoder.py:85-132 (numpy branch, float64)."""
    b = np.asarray(boxes_3d, dtype=np.float64).reshape(-1, 7)
    ry = b[:, 6]
    if ortho_rotate:
        ry = np.round(ry / (np.pi / 2)) * (np.pi / 2)
    co, si = np.abs(np.cos(ry)), np.abs(np.sin(ry))
    return np.stack([b[:, 0], b[:, 1], b[:, 2], b[:, 3] * co + b[:, 4] * si,
                     b[:, 5], b[:, 4] * co + b[:, 3] * si], axis=1)


def anchor_grid_cells(anchors, area_extents, voxel_size):
    """(N,4) int32 [x1,z1,x2,z2] voxel indices of each anchor's BEV footprint:
    float32 corners / voxel_size in float32, truncated, shifted by the grid
    minimum and clipped to [0, num_divisions] (inclusive)."""
    a = np.asarray(anchors, dtype=np.float64)
    ext = np.asarray(area_extents, dtype=np.float64)
    mn = np.array([np.floor(ext[0, 0] / voxel_size), np.floor(ext[2, 0] / voxel_size)])
    mx = np.array([np.ceil(ext[0, 1] / voxel_size - 1), np.ceil(ext[2, 1] / voxel_size - 1)])
    nd = (mx - mn + 1).astype(np.int32)
    lo = np.stack([a[:, 0] - a[:, 3] / 2., a[:, 2] - a[:, 5] / 2.], 1).astype(np.float32)
    hi = np.stack([a[:, 0] + a[:, 3] / 2., a[:, 2] + a[:, 5] / 2.], 1).astype(np.float32)
    cells = np.empty((len(a), 4), dtype=np.int32)
    for col, pts in ((0, lo), (2, hi)):
        idx = np.int32(pts / np.float32(voxel_size)) - mn
        cells[:, col] = np.clip(idx[:, 0], 0, nd[0])
        cells[:, col + 1] = np.clip(idx[:, 1], 0, nd[1])
    return cells, int(nd[0]), int(nd[1])


class GridAnchor3dGenerator(object):
    def name_scope(self):
        return 'GridAnchor3dGenerator'

    def generate(self, **params):
        return tile_anchors_3d(params.get('area_3d'), params.get('anchor_3d_sizes'),
                               params.get('anchor_stride'), params.get('ground_plane'))
