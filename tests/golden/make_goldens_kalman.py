#!/usr/bin/env python3
"""Golden vectors for the Kalman track filter ("next #4": KF tracker): the
reference's own Tracker (avod/utils/kalman_tracker.py:9-89, numpy + scipy only, loaded from its
file -- no package import) driven through seeded sequences of kalman_filter / predict_only
calls; state and covariance after every call.

Run:  python tests/golden/make_goldens_kalman.py     (needs /root/reference; writes kalman.npz)
"""
import importlib.util
import os

import numpy as np

REF = '/root/reference/avod/utils/kalman_tracker.py'


def main():
    spec = importlib.util.spec_from_file_location('ref_kalman_tracker', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(20261007)
    out = {}
    t0 = mod.Tracker()
    for name in ('F', 'H', 'P', 'Q', 'R'):
        out['init_' + name] = np.asarray(getattr(t0, name), np.float64)
    for case in range(6):
        n = int(rng.integers(5, 40))
        ops = (rng.uniform(size=n) < 0.7).astype(np.int32)        # 1 = kalman_filter, 0 = predict_only
        pos0 = rng.uniform(-30, 30, 4)
        vel = rng.uniform(-1.5, 1.5, 4)
        zs = np.stack([pos0 + vel * (k + 1) + rng.normal(0, 0.2, 4) for k in range(n)])
        trk = mod.Tracker()
        if case >= 4:                 # other noise settings, through the reference's own update_R
            trk.L = 3.0 + case
            trk.R_scaler = 0.25
            trk.update_R()
        trk.x_state = np.array([[pos0[0], 0, pos0[1], 0, pos0[2], 0, pos0[3], 0]], np.float64).T
        xs, ps = [], []
        for k in range(n):
            if ops[k]:
                trk.kalman_filter(zs[k][:, None])
            else:
                trk.predict_only()
            xs.append(np.asarray(trk.x_state, np.float64)[:, 0].copy())
            ps.append(np.asarray(trk.P, np.float64).copy())
        out['c%d_ops' % case] = ops
        out['c%d_z' % case] = zs
        out['c%d_x0' % case] = pos0
        out['c%d_x' % case] = np.stack(xs)
        out['c%d_P' % case] = np.stack(ps)
        out['c%d_LR' % case] = np.array([trk.L, trk.R_scaler])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'kalman.npz'), **out)
    print('wrote kalman.npz:', sorted(out)[:6], '...')


if __name__ == '__main__':
    main()
