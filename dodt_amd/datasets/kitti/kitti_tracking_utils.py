"""Ego-motion between the two frames of a DODT sample from KITTI's OXTS (GPS/IMU) records,
host side (numpy, float64).  Stands where these stand in the reference:

  Oxts                              avod/datasets/kitti/kitti_tracking_utils.py:129-215
  coordinate_transform              avod/datasets/kitti/kitti_tracking_dataset.py:303-322
  point_cloud_transform             kitti_tracking_dataset.py:324-335  (the device applies it
                                    inside dodt_bev_slices: ops.with_ego_motion)
  recovery_t / recovery_coordinate  kitti_tracking_dataset.py:374-389,
                                    avod/core/dt_evaluator_utils.py:189-210

The rotation helpers keep the reference's (non-standard) axis naming: `rotz` is built from
the pitch difference and rotates about the lidar y axis, `roty` from the yaw difference about
the lidar z axis.
"""
import numpy as np

EARTH_RADIUS = 6378137.0   # m


class Oxts(object):
    """The first six of the 30 values of one OXTS line: latitude, longitude (deg), altitude
    (m), roll, pitch, yaw (rad)."""

    def __init__(self, oxts_line):
        data = oxts_line.split() if isinstance(oxts_line, str) else list(oxts_line)
        if len(data) < 6:
            raise ValueError('an OXTS record holds at least 6 values')
        (self.latitude, self.longitude, self.altitude,
         self.roll, self.pitch, self.yaw) = [float(v) for v in data[:6]]

    @staticmethod
    def rotx(t):
        c, s = np.cos(t), np.sin(t)
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])

    @staticmethod
    def rotz(t):
        c, s = np.cos(t), np.sin(t)
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])

    @staticmethod
    def roty(t):
        c, s = np.cos(t), np.sin(t)
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])

    def distance(self, other):
        """Haversine distance over (latitude, longitude)."""
        lat1, lon1 = self.latitude * np.pi / 180.00, self.longitude * np.pi / 180.00
        lat2, lon2 = other.latitude * np.pi / 180.00, other.longitude * np.pi / 180.00
        a, b = lat2 - lat1, lon2 - lon1
        dis = 2 * EARTH_RADIUS * np.arcsin(np.sqrt(
            np.power(np.sin(a / 2), 2) + np.cos(lat1) * np.cos(lat2) * np.power(np.sin(b / 2), 2)))
        return abs(dis)

    def displacement(self, other):
        d = self.distance(other)
        delta_yaw = self.yaw - other.yaw
        delta_pitch = self.pitch - other.pitch
        return np.array([d * np.cos(delta_yaw), d * np.sin(delta_yaw), d * np.sin(delta_pitch)])

    def get_rotate_matrix(self, other, axis='y'):
        if axis == 'z':
            return self.rotz(self.pitch - other.pitch)
        if axis == 'x':
            return self.rotx(self.roll - other.roll)
        if axis == 'y':
            return self.roty(self.yaw - other.yaw)
        raise ValueError('axis must be x, y or z')

    def get_delta(self, other, theta='yaw'):
        return getattr(self, theta) - getattr(other, theta)


def read_oxts(path, frame_id):
    """The record of frame `frame_id` of a per-video OXTS file (one line per frame;
    KittiTrackingDataset.get_oxts, kitti_tracking_dataset.py:215-223)."""
    with open(path) as f:
        lines = [line.rstrip() for line in f.readlines()]
    return Oxts(lines[frame_id])


def coordinate_transform(oxts_cur, oxts_next):
    """-> (translation (3,), rotation (3,3) = Rz @ Rx @ Ry, yaw delta): represent the next
    frame in the current frame's velodyne coordinates."""
    distance = oxts_cur.displacement(oxts_next)
    delta = oxts_cur.get_delta(oxts_next, theta='yaw')
    matrix = (oxts_cur.get_rotate_matrix(oxts_next, 'z') @ oxts_cur.get_rotate_matrix(oxts_next, 'x')
              @ oxts_cur.get_rotate_matrix(oxts_next, 'y'))
    return distance, matrix, delta


def point_cloud_transform(xyzi, trans, matrix):
    """Host twin of the device's pre-transform: (N,4) float32 velodyne points -> the float32
    cloud the reference continues with (kitti_tracking_dataset.py:324-335)."""
    out = np.array(xyzi, dtype=np.float32, copy=True)
    out[:, :3] = (out[:, :3] + trans) @ matrix
    return out


# ---- rectified camera <-> velodyne (wavedata/.../core/calib_utils.py:31-68,218-225) --------
def _rect_to_velo(pts_rect, r0_rect, tr_velo_to_cam):
    ref = np.transpose(np.dot(np.linalg.inv(r0_rect), np.transpose(pts_rect)))
    inv = np.zeros_like(tr_velo_to_cam)
    inv[0:3, 0:3] = np.transpose(tr_velo_to_cam[0:3, 0:3])
    inv[0:3, 3] = np.dot(-np.transpose(tr_velo_to_cam[0:3, 0:3]), tr_velo_to_cam[0:3, 3])
    hom = np.hstack((ref, np.ones((len(ref), 1))))
    return np.dot(hom, np.transpose(inv))


def _velo_to_rect(pts_velo, r0_rect, tr_velo_to_cam):
    hom = np.hstack((pts_velo, np.ones((len(pts_velo), 1))))
    ref = np.dot(hom, np.transpose(tr_velo_to_cam))
    return np.transpose(np.dot(r0_rect, np.transpose(ref)))


def box_corners_3d(box_3d):
    """(8,3) corners of [x,y,z,l,w,h,ry] (wavedata/.../obj_utils.py:315-345)."""
    x, y, z, l, w, h, ry = [float(v) for v in box_3d[:7]]
    rot = np.array([[+np.cos(ry), 0, +np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, +np.cos(ry)]])
    xc = np.array([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2])
    yc = np.array([0, 0, 0, 0, -h, -h, -h, -h])
    zc = np.array([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2])
    c = np.dot(rot, np.array([xc, yc, zc]))
    c[0] += x
    c[1] += y
    c[2] += z
    return c.T


def recovery_t(box_3d, r0_rect, tr_velo_to_cam, trans, matrix):
    """Bottom centre of a box of the registered second frame back in that frame's own
    coordinates (kitti_tracking_dataset.py:374-389)."""
    velo = _rect_to_velo(box_corners_3d(box_3d), r0_rect, tr_velo_to_cam)
    velo = velo @ np.linalg.inv(matrix) - trans
    rect = _velo_to_rect(velo, r0_rect, tr_velo_to_cam)
    origin_t = np.mean(rect, axis=0)
    origin_t[1] += float(box_3d[5]) / 2.0
    return origin_t


def recovery_coordinate(predictions, r0_rect, tr_velo_to_cam, trans, matrix, delta):
    """dt_evaluator_utils.recovery_coordinate (:189-210) with the dataset look-ups resolved by
    the caller: rows [x,y,z,l,w,h,ry,...] of the pair's second frame, in place and returned."""
    for j in range(len(predictions)):
        predictions[j][0:3] = recovery_t(predictions[j], r0_rect, tr_velo_to_cam, trans, matrix)
        predictions[j][6] = predictions[j][6] - delta
    return predictions
