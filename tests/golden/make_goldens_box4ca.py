#!/usr/bin/env python3
"""Golden vectors for the `box_4ca` detection records (SURVEY 8 a14 + 8f item 3): the
reference's own DtEvaluator.get_avod_predicted_boxes_3d_and_scores
(avod/core/dt_evaluator.py:1134-1259, branch box_rep == 'box_4ca' :1166-1212), run in the
build container on seeded synthetic network outputs.

Run:  python tests/golden/make_goldens_box4ca.py      (needs /root/reference; writes box4ca.npz)

The method is called unbound (it never touches `self`; DtEvaluator.__init__ needs a TF
session and the protobuf config system, neither of which exists here).  Its module imports
tensorflow and tensorflow.contrib at module scope: the same inert stand-ins as in
make_goldens.py are registered, plus one for the `tensorflow.contrib` sub-package; nothing
of TensorFlow runs.  Inputs are float32 like the arrays sess.run hands to the evaluator.
Stored per case: the prediction dict's arrays (boxes_3d, orientations, softmax per frame,
corr offsets of frame 0) and the (n0 + n1, 17) array the reference returned.
"""
import importlib.abc
import importlib.machinery
import os
import sys
from unittest.mock import MagicMock

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402


class _InertTfSubmodules(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`from tensorflow.contrib import slim` -> an inert MagicMock module."""

    def find_spec(self, name, path, target=None):
        if name.startswith('tensorflow.'):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock()
        m.__path__, m.__name__, m.__spec__ = [], spec.name, spec
        return m

    def exec_module(self, module):
        pass


def import_evaluator():
    mg._import_reference()
    sys.modules['tensorflow'].__path__ = []
    sys.meta_path.insert(0, _InertTfSubmodules())
    from avod.core import dt_evaluator
    from avod.core.models.dt_avod_model import DtAvodModel
    return dt_evaluator.DtEvaluator, DtAvodModel


def case(rng, n0, n1, special=False):
    f32 = np.float32

    def boxes(n):
        return np.stack([rng.uniform(-35, 35, n), rng.uniform(1.2, 1.9, n), rng.uniform(2, 68, n),
                         rng.uniform(2.8, 4.8, n), rng.uniform(1.4, 1.9, n),
                         rng.uniform(1.3, 1.8, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(f32)
    b = [boxes(n0), boxes(n1)]
    ori = [rng.uniform(-np.pi, np.pi, n).astype(f32) for n in (n0, n1)]
    if special:
        # differences on and next to every threshold of dt_evaluator.py:1185-1208, computed
        # in float32 exactly as the evaluator will recompute them
        targets = np.array([0.25, 0.5, 0.75, 1.0, -0.25, -0.5, -0.75, -1.0, 0.0, 1.25, -1.25,
                            1.75, -1.75, 0.2499999, 0.7500001, -0.2499999, -0.7500001],
                           np.float64) * np.pi
        for i in range(2):
            k = min(len(targets), len(b[i]))
            ori[i][:k] = (b[i][:k, 6].astype(np.float64) - targets[:k]).astype(f32)
        # ry + pi/2 above pi (wraps) and ry - pi/2 below -pi (the reference does not wrap it)
        if n0 > 20:
            b[0][17, 6], ori[0][17] = f32(3.0), f32(3.0 - 1.5)
            b[0][18, 6], ori[0][18] = f32(-3.0), f32(-3.0 + 1.5)
            b[0][19, 6], ori[0][19] = f32(2.0), f32(2.0 - 3.0)
    sm = []
    for n in (n0, n1):
        p = rng.uniform(0.0, 1.0, n).astype(f32)
        sm.append(np.stack([f32(1.0) - p, p], 1).astype(f32))
    corr = rng.normal(0, 0.4, size=(n0, 3)).astype(f32)
    return b, ori, sm, corr


def main():
    DtEvaluator, M = import_evaluator()
    rng = np.random.default_rng(20261004)
    out = {}
    shapes = [(100, 100, False), (37, 64, True), (1, 5, False), (64, 1, True), (100, 23, True)]
    for ci, (n0, n1, special) in enumerate(shapes):
        b, ori, sm, corr = case(rng, n0, n1, special)
        pred = {M.PRED_TOP_PREDICTION_BOXES_3D: [x.copy() for x in b],
                M.PRED_TOP_ORIENTATIONS: [x.copy() for x in ori],
                M.PRED_TOP_CLASSIFICATION_SOFTMAX: [x.copy() for x in sm],
                M.PRED_TOP_CORR_OFFSETS: corr.copy()}
        res = DtEvaluator.get_avod_predicted_boxes_3d_and_scores(None, pred, 'box_4ca')
        assert res.shape == (n0 + n1, 17)
        for f in range(2):
            out['c%d_boxes_3d_%d' % (ci, f)] = b[f]
            out['c%d_orientations_%d' % (ci, f)] = ori[f]
            out['c%d_softmax_%d' % (ci, f)] = sm[f]
        out['c%d_corr_offsets' % ci] = corr
        out['c%d_records' % ci] = res
        swapped = int((res[:n0, 3] != b[0][:, 3]).sum())
        print('case %d: %d + %d rows, %d of frame 0 with l/w swapped, dtype %s'
              % (ci, n0, n1, swapped, res.dtype))
    out['n_cases'] = np.asarray(len(shapes))
    np.savez_compressed(os.path.join(mg.HERE, 'box4ca.npz'), **out)


if __name__ == '__main__':
    main()
