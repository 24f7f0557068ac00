"""Oracle for the point pipeline, SURVEY.md section 8(a) rows a0-a3.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Everything is float64, as
in the reference.  Written from the behaviour of the reference, not from its
code: the voxeliser here uses a scatter-min over (y-bin, original index) keys
instead of the reference's lexsort + unique, and is checked to give identical
results on the golden vectors.

Reference behaviour restated (paths relative to /root/reference):
  a0  wavedata/wavedata/tools/core/calib_utils.py:484-523  lidar_to_cam_frame
      wavedata/wavedata/tools/core/calib_utils.py:394-410  project_to_image
      wavedata/wavedata/tools/obj_detection/tracking_utils.py:117-150
      wavedata/wavedata/tools/obj_detection/obj_utils.py:220-268
  a1  wavedata/wavedata/tools/obj_detection/obj_utils.py:453-500 get_point_filter
      avod/datasets/kitti/kitti_utils.py:81-109             create_slice_filter
  a2  wavedata/wavedata/tools/core/voxel_grid_2d.py:43-160  voxelize_2d
      wavedata/wavedata/tools/core/geometry_utils.py:25-40  dist_to_plane
  a3  avod/core/bev_generators/bev_slices.py:33-150         generate_bev
      avod/core/bev_generators/bev_generator.py:23-41       _create_density_map
"""
import numpy as np


def velo_to_cam_matrix(r0_rect, tr_velo_to_cam):
    """(3,3) R0_rect and (3,4) Tr_velo_to_cam -> (4,4) velo->rectified-cam.

    calib_utils.py:502-519: both are padded to 4x4 homogeneous matrices and
    multiplied (float64)."""
    r0 = np.zeros((4, 4))
    r0[:3, :3] = np.asarray(r0_rect, dtype=np.float64)
    r0[3, 3] = 1.0
    tr = np.zeros((4, 4))
    tr[:3, :4] = np.asarray(tr_velo_to_cam, dtype=np.float64)
    tr[3, 3] = 1.0
    return np.dot(r0, tr)


def lidar_to_cam(xyz_velo, r0_rect, tr_velo_to_cam):
    """(N,3) velodyne-frame points (any float dtype) -> (N,3) float64 in the
    rectified camera frame.  calib_utils.py:512-523."""
    xyz = np.asarray(xyz_velo)
    hom = np.empty((xyz.shape[0], 4), dtype=np.float64)
    hom[:, :3] = xyz
    hom[:, 3] = 1.0
    m = velo_to_cam_matrix(r0_rect, tr_velo_to_cam)
    return np.dot(m, hom.T)[0:3].T


def project_to_image(points_3xn, p):
    """(3,N) camera-frame points, (3,4) camera matrix -> (2,N) pixel coords.
    calib_utils.py:394-410."""
    n = points_3xn.shape[1]
    hom = np.vstack([points_3xn, np.ones((1, n))])
    uvw = np.dot(np.asarray(p, dtype=np.float64), hom)
    return np.vstack([uvw[0] / uvw[2], uvw[1] / uvw[2]])


def lidar_in_camera_view(xyzi_velo, r0_rect, tr_velo_to_cam, p2, im_size_wh):
    """Raw (N,4) float32 xyzi -> (3,N_fov) float64 cloud the BEV generator sees.

    tracking_utils.py:117-150 / obj_utils.py:220-268: transform, keep z > 0,
    project with P2, keep 0 < u < W and 0 < v < H (all strict)."""
    pts = lidar_to_cam(np.asarray(xyzi_velo)[:, :3], r0_rect, tr_velo_to_cam)
    pts = pts[pts[:, 2] > 0]
    uv = project_to_image(pts.T, p2)
    keep = (uv[0] > 0) & (uv[0] < im_size_wh[0]) & \
           (uv[1] > 0) & (uv[1] < im_size_wh[1])
    return pts[keep].T


def fov_mask(xyzi_velo, r0_rect, tr_velo_to_cam, p2, im_size_wh):
    """Same test as lidar_in_camera_view but returned as a mask over the raw
    points together with all transformed points (used to check the fused GPU
    kernel, which never materialises the filtered cloud)."""
    pts = lidar_to_cam(np.asarray(xyzi_velo)[:, :3], r0_rect, tr_velo_to_cam)
    front = pts[:, 2] > 0
    with np.errstate(divide='ignore', invalid='ignore'):
        uv = project_to_image(pts.T, p2)
    keep = front & (uv[0] > 0) & (uv[0] < im_size_wh[0]) & \
        (uv[1] > 0) & (uv[1] < im_size_wh[1])
    return keep, pts


def point_filter(point_cloud_3xn, extents, ground_plane=None, offset_dist=2.0):
    """obj_utils.py:453-500.  extents = [[x0,x1],[y0,y1],[z0,z1]], all strict.
    With a plane: additionally (plane + [0,0,0,-offset]) . [x,y,z,1] < 0."""
    pc = np.asarray(point_cloud_3xn)
    e = np.asarray(extents, dtype=np.float64)
    keep = (pc[0] > e[0, 0]) & (pc[0] < e[0, 1]) & \
           (pc[1] > e[1, 0]) & (pc[1] < e[1, 1]) & \
           (pc[2] > e[2, 0]) & (pc[2] < e[2, 1])
    if ground_plane is not None:
        plane = np.array(ground_plane, dtype=np.float64) + \
            np.array([0.0, 0.0, 0.0, -offset_dist])
        hom = np.vstack([pc, np.ones(pc.shape[1])])
        keep = keep & (np.dot(plane, hom) < 0)
    return keep


def slice_filter(point_cloud_3xn, extents, ground_plane, height_lo, height_hi):
    """kitti_utils.py:81-109: filter(height_hi) XOR filter(height_lo)."""
    return np.logical_xor(
        point_filter(point_cloud_3xn, extents, ground_plane, height_hi),
        point_filter(point_cloud_3xn, extents, ground_plane, height_lo))


def dist_to_plane(plane, pts_nx3):
    """geometry_utils.py:25-40."""
    a, b, c, d = plane
    p = np.asarray(pts_nx3)
    return (a * p[:, 0] + b * p[:, 1] + c * p[:, 2] + d) / \
        np.sqrt(a ** 2 + b ** 2 + c ** 2)


class Voxels2D(object):
    """Result of voxelize_2d: same fields the reference's VoxelGrid2D exposes
    (voxel_grid_2d.py:16-41) that the hot path reads."""
    pass


def voxelize_2d(pts_nx3, voxel_size, extents=None, ground_plane=None,
                create_leaf_layout=True):
    """Restatement of VoxelGrid2D.voxelize_2d (voxel_grid_2d.py:43-160).

    Per (x,z) cell the reference keeps the first point, in original order, of
    the lowest floor(y/voxel_size) bin (lexsort by x, z, y is stable; unique
    returns first occurrences).  Here: scatter-min of key = ybin * n + index.
    Cells are returned in (x, z) ascending order like the reference.
    """
    pts = np.asarray(pts_nx3, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("Points have the wrong shape: {}".format(pts.shape))
    n = pts.shape[0]
    out = Voxels2D()
    out.voxel_size = voxel_size

    cells = np.floor(pts / voxel_size).astype(np.int32)
    cx = cells[:, 0].astype(np.int64)
    cy = cells[:, 1].astype(np.int64)
    cz = cells[:, 2].astype(np.int64)

    # dense id of the (x, z) column; ordering of ids == (x, z) lexicographic
    x0, z0 = cx.min(), cz.min()
    zspan = cz.max() - z0 + 1
    col = (cx - x0) * zspan + (cz - z0)
    key = (cy - cy.min()) * n + np.arange(n, dtype=np.int64)

    ncol = int(col.max()) + 1
    best = np.full(ncol, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(best, col, key)
    count = np.bincount(col, minlength=ncol)
    occupied = np.nonzero(count)[0]
    rep = best[occupied] % n

    voxel_coords = np.zeros((occupied.size, 3), dtype=np.int32)
    voxel_coords[:, 0] = occupied // zspan + x0
    voxel_coords[:, 2] = occupied % zspan + z0

    if ground_plane is None:
        out.heights = pts[rep, 1]
    else:
        out.heights = dist_to_plane(ground_plane, pts[rep])
    out.num_pts_in_voxel = count[occupied]
    out.rep_index = rep

    if extents is not None:
        ext_t = np.array(extents).transpose()
        if ext_t.shape != (2, 3):
            raise ValueError("Extents are the wrong shape {}".format(
                ext_t.shape))
        out.min_voxel_coord = np.floor(ext_t[0] / voxel_size)
        out.max_voxel_coord = np.ceil((ext_t[1] / voxel_size) - 1)
        out.min_voxel_coord[1] = 0
        out.max_voxel_coord[1] = 0
        if not (out.min_voxel_coord <= voxel_coords.min(axis=0)).all():
            raise ValueError("Extents are smaller than min_voxel_coord")
        if not (out.max_voxel_coord >= voxel_coords.max(axis=0)).all():
            raise ValueError("Extents are smaller than max_voxel_coord")
    else:
        out.min_voxel_coord = voxel_coords.min(axis=0)
        out.max_voxel_coord = voxel_coords.max(axis=0)

    out.num_divisions = ((out.max_voxel_coord - out.min_voxel_coord)
                         + 1).astype(np.int32)
    out.voxel_indices = (voxel_coords - out.min_voxel_coord).astype(int)

    if create_leaf_layout:
        out.leaf_layout_2d = -1 * np.ones(out.num_divisions.astype(int))
        out.leaf_layout_2d[out.voxel_indices[:, 0], 0,
                           out.voxel_indices[:, 2]] = 0
    else:
        out.leaf_layout_2d = []
    return out


def map_to_index(vox, map_xy):
    """VoxelGrid2D.map_to_index (voxel_grid_2d.py:162-186).  `map_xy` keeps its
    dtype: the anchor filter passes float32 corners, so the division by the
    (python float) voxel size is carried out in float32."""
    if vox.voxel_size == 0 or len(vox.min_voxel_coord) == 0 or \
            len(map_xy) == 0:
        return []
    nd = vox.num_divisions[[0, 2]]
    mn = vox.min_voxel_coord[[0, 2]]
    idx = np.int32(map_xy / vox.voxel_size) - mn
    idx[:, 0] = np.clip(idx[:, 0], 0, nd[0])
    idx[:, 1] = np.clip(idx[:, 1], 0, nd[1])
    return idx


LIDAR_DENSITY_NORM = np.log(16)     # bev_slices.py:10-12


def generate_bev(point_cloud_3xn, ground_plane, area_extents, voxel_size,
                 height_lo, height_hi, num_slices):
    """BevSlices.generate_bev (bev_slices.py:33-150).

    Returns dict(height_maps=[num_slices x (Z,X) float64], density_map=(Z,X)),
    already rotated (out[r, c] = map[c, Z-1-r]).  A slice with <= 1 member
    point is replaced by the single point (0,0,0) (bev_slices.py:76-99).
    """
    all_pts = np.transpose(point_cloud_3xn)
    per_div = (height_hi - height_lo) / num_slices
    maps = []
    for s in range(num_slices):
        lo = height_lo + s * per_div
        hi = lo + per_div
        m = slice_filter(point_cloud_3xn, area_extents, ground_plane, lo, hi)
        sp = all_pts[m]
        if len(sp) <= 1:
            sp = np.zeros((1, 3))
        vg = voxelize_2d(sp, voxel_size, extents=area_extents,
                         ground_plane=ground_plane, create_leaf_layout=False)
        hm = np.zeros((vg.num_divisions[0], vg.num_divisions[2]))
        hm[vg.voxel_indices[:, 0], vg.voxel_indices[:, 2]] = \
            (vg.heights - lo) / per_div
        maps.append(np.flip(hm.transpose(), axis=0))

    m = slice_filter(point_cloud_3xn, area_extents, ground_plane,
                     height_lo, height_hi)
    vg = voxelize_2d(all_pts[m], voxel_size, extents=area_extents,
                     ground_plane=ground_plane, create_leaf_layout=False)
    dm = np.zeros((vg.num_divisions[0], vg.num_divisions[2]))
    dm[vg.voxel_indices[:, 0], vg.voxel_indices[:, 2]] = np.minimum(
        1.0, np.log(vg.num_pts_in_voxel + 1) / LIDAR_DENSITY_NORM)
    return dict(height_maps=maps,
                density_map=np.flip(dm.transpose(), axis=0))


def bev_input(point_cloud_3xn, ground_plane, area_extents, voxel_size,
              height_lo, height_hi, num_slices):
    """The (Z, X, num_slices+1) array the dataset feeds to the network
    (kitti_tracking_dataset.py:562-568: dstack of height maps + density)."""
    b = generate_bev(point_cloud_3xn, ground_plane, area_extents, voxel_size,
                     height_lo, height_hi, num_slices)
    return np.dstack(b['height_maps'] + [b['density_map']])


def point_cloud_transform(xyzi_velo, trans, matrix):
    """KittiTrackingDataset.point_cloud_transform (avod/datasets/kitti/
    kitti_tracking_dataset.py:324-335): the pair's second frame registered into the first
    frame's velodyne coordinates, (p + trans) @ matrix in float64, stored back into the float32
    cloud (one rounding) -- pinned by tests/golden/egomotion.npz."""
    out = np.array(xyzi_velo, dtype=np.float32, copy=True)
    out[:, :3] = (out[:, :3] + np.asarray(trans, np.float64)) @ np.asarray(matrix, np.float64)
    return out
