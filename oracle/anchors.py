"""Oracle for the per-frame anchor grid and empty-anchor filter, SURVEY.md 8(a) row a4.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference behaviour restated (paths relative to /root/reference):
  avod/core/anchor_generators/grid_anchor_3d_generator.py:39-108  tile_anchors_3d
  avod/core/box_3d_encoder.py:85-132                              box_3d_to_anchor
  avod/datasets/kitti/kitti_utils.py:210-260  (slice 0.2..2.0, leaf layout on)
  avod/core/anchor_filter.py:64-119           get_empty_anchor_filter_2d
  wavedata/wavedata/tools/core/integral_image_2d.py:15-87
"""
import numpy as np

from oracle import points


def tile_anchors_3d(area_extents, anchor_3d_sizes, anchor_stride, ground_plane):
    """All grid anchors as box_3d rows [x, y, z, l, w, h, ry] (float64).

    Order (grid_anchor_3d_generator.py:73-84, meshgrid 'xy' + reshape): z
    slowest (far -> near), then x (left -> right), then size, then rotation
    (0, pi/2).  Centres are rounded to float32 first (:65-71)."""
    sizes = np.asarray(anchor_3d_sizes, dtype=np.float64)
    rots = np.asarray([0.0, np.pi / 2.0])
    xs = np.arange(area_extents[0][0] + anchor_stride[0] / 2.0,
                   area_extents[0][1], anchor_stride[0]).astype(np.float32)
    zs = np.arange(area_extents[2][1] - anchor_stride[1] / 2.0,
                   area_extents[2][0], -anchor_stride[1]).astype(np.float32)
    nz, nx, ns, nr = len(zs), len(xs), len(sizes), len(rots)
    n = nz * nx * ns * nr
    boxes = np.zeros((n, 7))
    # float32 centres are promoted to float64 by the meshgrid with int ranges
    gx = np.broadcast_to(xs.astype(np.float64)[None, :, None, None],
                         (nz, nx, ns, nr)).reshape(-1)
    gz = np.broadcast_to(zs.astype(np.float64)[:, None, None, None],
                         (nz, nx, ns, nr)).reshape(-1)
    si = np.broadcast_to(np.arange(ns)[None, None, :, None],
                         (nz, nx, ns, nr)).reshape(-1)
    ri = np.broadcast_to(np.arange(nr)[None, None, None, :],
                         (nz, nx, ns, nr)).reshape(-1)
    a, b, c, d = ground_plane
    boxes[:, 0] = gx
    boxes[:, 1] = -(a * gx + c * gz + d) / b
    boxes[:, 2] = gz
    boxes[:, 3:6] = sizes[si]
    boxes[:, 6] = rots[ri]
    return boxes


def box_3d_to_anchor(boxes_3d, ortho_rotate=False):
    """box_3d_encoder.py:85-132: [x,y,z,l,w,h,ry] -> [x,y,z,dim_x,dim_y,dim_z]."""
    b = np.asarray(boxes_3d, dtype=np.float64).reshape(-1, 7)
    ry = b[:, 6]
    if ortho_rotate:
        half_pi = np.pi / 2
        ry = np.round(ry / half_pi) * half_pi
    c = np.abs(np.cos(ry))
    s = np.abs(np.sin(ry))
    out = np.zeros((len(b), 6))
    out[:, 0:3] = b[:, 0:3]
    out[:, 3] = b[:, 3] * c + b[:, 4] * s
    out[:, 4] = b[:, 5]
    out[:, 5] = b[:, 4] * c + b[:, 3] * s
    return out


def sliced_voxel_grid_2d(point_cloud_3xn, ground_plane, area_extents,
                         voxel_size, height_lo=0.2, height_hi=2.0):
    """kitti_utils.py:210-260: occupancy grid of the [0.2, 2.0) slice with the
    leaf layout (-1 empty / 0 filled)."""
    m = points.slice_filter(point_cloud_3xn, area_extents, ground_plane,
                            height_lo, height_hi)
    pts = np.asarray(point_cloud_3xn).T[m]
    return points.voxelize_2d(pts, voxel_size, extents=area_extents,
                              ground_plane=ground_plane,
                              create_leaf_layout=True)


def summed_area_table(img):
    """integral_image_2d.py:15-37: inclusive 2-D prefix sum, zero-padded on
    the low side so that S[i, j] = sum(img[:i, :j])."""
    if img.ndim != 2:
        raise ValueError('Not a 2D image for integral image: input dim {}'
                         .format(img.ndim))
    s = np.zeros((img.shape[0] + 1, img.shape[1] + 1))
    s[1:, 1:] = img.cumsum(0).cumsum(1)
    return s


def sat_query(sat, boxes_4xn):
    """integral_image_2d.py:39-87: boxes are uint32 [x1, z1, x2, z2] columns,
    clipped to the table size; value = number of filled cells in
    [x1, x2) x [z1, z2)."""
    boxes = np.asarray(boxes_4xn)
    if boxes.shape[0] != 4:
        raise ValueError('Incorrect number of dimensions for query: '
                         'input dim {}'.format(boxes.shape[0]))
    if boxes.dtype != np.uint32:
        raise TypeError('boxes must be type of np.uint32')
    lim = np.array([sat.shape[0], sat.shape[1],
                    sat.shape[0], sat.shape[1]]).reshape(4, 1) - 1
    b = np.minimum(boxes, lim).astype(np.uint32)
    x1, z1, x2, z2 = b
    return sat[x2, z2] + sat[x1, z1] - sat[x2, z1] - sat[x1, z2]


def empty_anchor_filter_2d(anchors, vox, density_threshold=1):
    """anchor_filter.py:64-119.  Corners are computed in float64, stored as
    float32, divided by the voxel size in float32, truncated toward zero to
    int32, shifted by the grid minimum, clipped to [0, num_div] and cast to
    uint32."""
    a = np.asarray(anchors)
    a2 = a[:, [0, 2, 3, 5]]
    occ = np.squeeze(vox.leaf_layout_2d + 1)
    sat = summed_area_table(occ)
    lo = np.zeros((len(a2), 2), dtype=np.float32)
    hi = np.zeros((len(a2), 2), dtype=np.float32)
    lo[:, 0] = a2[:, 0] - (a2[:, 2] / 2.)
    lo[:, 1] = a2[:, 1] - (a2[:, 3] / 2.)
    hi[:, 0] = a2[:, 0] + (a2[:, 2] / 2.)
    hi[:, 1] = a2[:, 1] + (a2[:, 3] / 2.)
    box = np.zeros((len(a2), 4), dtype=np.uint32)
    box[:, :2] = points.map_to_index(vox, lo)
    box[:, 2:] = points.map_to_index(vox, hi)
    return sat_query(sat, box.T) >= density_threshold
