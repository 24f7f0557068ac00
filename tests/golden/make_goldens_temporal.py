#!/usr/bin/env python3
"""Golden vectors for the temporal module: the reference's own
interpolate_non_keyframe_predicitons / interpolate_trajectory
(avod/core/dt_evaluator_utils.py:212-367) and three_d_iou
(wavedata/.../obj_detection/evaluation.py:43-262), run in the build container.

Run:  python tests/golden/make_goldens_temporal.py      (needs /root/reference and PIL)
The reference module imports tensorflow at module scope: the same inert stand-in as in
make_goldens.py is registered; nothing of TensorFlow runs.  `recovery_coordinate` needs the
dataset's OXTS files (out of scope here) and is replaced by a recorder that returns its input:
the fixtures hold, per case, which frames it was asked to recover.
Stored: the 17-column detection records of seeded synthetic keyframe pairs, tau, the
threshold, and the per-frame outputs.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402


class _Rows(np.ndarray):
    """The reference tests `track[i] != []` on detection rows (dt_evaluator_utils.py:302-330).
    With the numpy of its day that comparison of an array with an empty list evaluated to
    True (and `==` to False); numpy 2 raises instead.  The fixtures are generated with input
    arrays of this subclass, which answers those two comparisons the old way and is an
    ordinary ndarray otherwise, so that the reference's own code runs unmodified."""

    def __ne__(self, other):
        if isinstance(other, list) and len(other) == 0:
            return True
        return np.ndarray.__ne__(self, other)

    def __eq__(self, other):
        if isinstance(other, list) and len(other) == 0:
            return False
        return np.ndarray.__eq__(self, other)

    __hash__ = None


class _Dataset(object):
    def create_all_sample_names(self, sample_names):
        if len(sample_names) == 1:
            return list(sample_names)
        a, b = int(sample_names[0]), int(sample_names[1])
        return ['%06d' % i for i in range(a, b + 1)]


def records(rng, n0, n1, n_match, far=False):
    """Frame 0: n0 boxes; frame 1: n_match of them moved a little + (n1 - n_match) new ones."""
    def boxes(n):
        return np.stack([rng.uniform(-30, 30, n), rng.uniform(1.2, 1.9, n), rng.uniform(5, 65, n),
                         rng.uniform(3.2, 4.6, n), rng.uniform(1.5, 1.8, n), rng.uniform(1.4, 1.7, n),
                         rng.uniform(-np.pi, np.pi, n)], 1)
    b0 = boxes(n0)
    b1 = boxes(n1)
    b1[:n_match] = b0[:n_match]
    b1[:n_match, [0, 2]] += rng.uniform(-0.6, 0.6, (n_match, 2)) * (4.0 if far else 1.0)
    b1[:n_match, 6] += rng.uniform(-0.05, 0.05, n_match)
    rows = []
    for mark, b in ((0, b0), (1, b1)):
        r = np.zeros((len(b), 17))
        r[:, :7] = b
        r[:, 7] = rng.uniform(0.02, 1.0, len(b))
        r[:, 8] = 0
        if mark == 0:
            r[:, 9:16] = b
            r[:, [9, 11, 15]] += rng.uniform(-0.8, 0.8, (len(b), 3))
        r[:, 16] = mark
        rows.append(r)
    return np.concatenate(rows, 0)


def main():
    mg._import_reference()
    from unittest.mock import MagicMock
    # sub-packages of the same inert tensorflow stand-in (the import chain of the module
    # reaches `from tensorflow.contrib import slim` and friends)
    for name in ('tensorflow.contrib', 'tensorflow.contrib.slim', 'tensorflow.python',
                 'tensorflow.python.framework', 'tensorflow.python.ops',
                 'tensorflow.contrib.layers', 'tensorflow.python.platform',
                 'tensorflow.contrib.framework', 'tensorflow.python.training',
                 'tensorflow.contrib.slim.python', 'tensorflow.contrib.slim.python.slim'):
        sys.modules.setdefault(name, MagicMock())
    import avod.core.dt_evaluator_utils as ref
    from wavedata.tools.obj_detection.evaluation import three_d_iou
    calls = []

    def recorder(dataset, sample_names, predictions):
        calls.append(list(sample_names))
        return predictions
    ref.recovery_coordinate = recorder
    rng = np.random.default_rng(20260405)
    out = {}
    cases = [(0, 1, 6, 5, 4), (1, 2, 8, 7, 5), (2, 3, 5, 9, 3), (3, 3, 0, 4, 0), (4, 2, 4, 0, 0),
             (5, 2, 0, 0, 0), (6, 3, 10, 10, 10), (7, 2, 6, 6, 4), (8, 0, 5, 0, 0)]
    for cid, tau, n0, n1, nm in cases:
        pred = records(rng, n0, n1, nm, far=(cid == 7))
        names = ['000010'] if tau == 0 else ['000010', '%06d' % (10 + tau)]
        if tau == 0:
            pred = pred[pred[:, -1] == 0]
        del calls[:]
        finals, all_names = ref.interpolate_non_keyframe_predicitons(
            _Dataset(), names, pred.copy().view(_Rows), 0.1)
        out['c%d_pred' % cid] = pred
        out['c%d_tau' % cid] = np.asarray(tau)
        out['c%d_nframes' % cid] = np.asarray(len(all_names))
        out['c%d_recovered' % cid] = np.asarray([int(c[1]) - 10 for c in calls], dtype=np.int64)
        for i, f in enumerate(finals):
            f = np.asarray(f, dtype=np.float64)
            out['c%d_out%d' % (cid, i)] = f.reshape(-1, 13) if f.size else np.zeros((0, 13))
    # three_d_iou on its own: one box against many, [ry, l, h, w, tx, ty, tz]
    b = records(rng, 40, 40, 25)[:, :7]
    fmt = b[:, [6, 3, 5, 4, 0, 1, 2]]
    out['iou_boxes'] = fmt
    out['iou_matrix'] = np.stack([np.atleast_1d(three_d_iou(fmt[i], fmt)) for i in range(12)])
    np.savez_compressed(os.path.join(mg.HERE, 'temporal.npz'), **out)
    print('cases', len(cases), 'iou nonzero', int((out['iou_matrix'] > 0).sum()))


if __name__ == '__main__':
    main()
