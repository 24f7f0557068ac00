"""Constant-velocity Kalman filter of one track on the host: stands where
avod/utils/kalman_tracker.py:9-89 (`Tracker`) stands for the KF variant of the video-level
association (avod/experiments/video_detection_kf.py:365-475, avod/core/tracking/kf_tracking.py).

State (8,1): [x, vx, y, vy, z, vz, ry, v_ry] of a box centre and heading, one (position, rate)
pair per measured quantity; the measurement (4,1) is the four positions.  Same attribute names
and the same float64 operation order as the reference (predict: x = F x, P = F P F' + Q; update:
S = H P H' + R, K = P H' S^-1, x += K (z - H x), P = P - K H P), so that a caller written for
the reference's class runs unchanged.
"""
import numpy as np


class Tracker(object):
    def __init__(self):
        # bookkeeping of the association loop
        self.id = 0            # track id
        self.dets = []         # detections (dicts) appended by the pipeline
        self.box = []          # the four filtered positions
        self.hits = 0          # matched detections so far
        self.no_losses = 0     # consecutive frames without a match
        self.x_state = []      # (8,1) once a caller has set it
        self.dt = 1.0
        pair = np.array([[1.0, self.dt], [0.0, 1.0]])
        self.F = np.kron(np.eye(4), pair)                 # each (position, rate) pair advances by dt
        self.H = np.eye(8)[0::2]                          # the positions are what is measured
        self.L = 10.0
        self.P = self.L * np.eye(8)
        # white-noise acceleration per pair: [[dt^4/4, dt^3/2], [dt^3/2, dt^2]]
        g = np.array([[self.dt ** 2 / 2.0], [self.dt]])
        self.Q_comp_mat = g @ g.T
        self.Q = np.kron(np.eye(4), self.Q_comp_mat)
        self.R_scaler = 1.0 / 16
        self.update_R()

    def update_R(self):
        """Measurement covariance from the current L and R_scaler."""
        self.R_diag_array = self.R_scaler * np.full(4, self.L)
        self.R = np.diag(self.R_diag_array)

    def _predict(self):
        x = self.F @ np.asarray(self.x_state, dtype=np.float64)
        self.P = self.F @ self.P @ self.F.T + self.Q
        return x

    def kalman_filter(self, z):
        """One predict + update with the measurement z (4,1)."""
        x = self._predict()
        S = self.H @ self.P @ self.H.T + self.R
        K = self.P @ self.H.T @ np.linalg.inv(S)
        x = x + K @ (np.asarray(z, dtype=np.float64) - self.H @ x)
        self.P = self.P - K @ self.H @ self.P
        self.x_state = x.astype(float)

    def predict_only(self):
        """The predict stage alone: unmatched detections (a fresh track) and unmatched tracks."""
        self.x_state = self._predict().astype(float)
