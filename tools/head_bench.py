#!/usr/bin/env python3
"""Stand-alone times of the dense heads at the pipeline's sizes (HIP events): the RPN anchor predictor on
5 500 anchors, the stage-2 head and the correlation head on 1 024 proposals.  `python tools/head_bench.py [dtype]`"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dodt_amd import device, synth  # noqa: E402
from dodt_amd.core.avod_fc_layers.fusion_fc_layers import EarlyFusionFcLayers  # noqa: E402
from dodt_amd.core.models.anchor_predictor import AnchorPredictor  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else 'f32'
ctx = device.default_context()
hp = synth.head_params()
rng = np.random.default_rng(0)
A, P = 5500, 1024


def timed(name, fn, flops, reps=20):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    us = ctx.timer_stop() / reps * 1e3
    print('%-34s %8.1f us  %6.1f TFLOP/s' % (name, us, flops / us * 1e-6))


rpn = AnchorPredictor(ctx, hp['rpn'], dtype=dt)
x1, x2 = ctx.array(rng.normal(size=(A, 9)).astype(np.float32)), ctx.array(rng.normal(size=(A, 9)).astype(np.float32))
lo, of = ctx.empty((A, 2), np.float32), ctx.empty((A, 6), np.float32)
sc = rpn.make_scratch(A)
timed('rpn head, %d anchors' % A, lambda: rpn.forward(ctx, x1, x2, A, lo, of, sc), rpn.flops(A))
for i, l in enumerate(rpn.layers):
    print('   layer %d: K %d N %d' % (i, l.K, l.N))

avod = EarlyFusionFcLayers(ctx, hp['avod'], dtype=dt)
r1 = ctx.array(rng.normal(size=(P, avod.in_ld)).astype(np.float32))
r2 = ctx.array(rng.normal(size=(P, avod.in_ld)).astype(np.float32))
outs = [ctx.empty((P, n), np.float32) for n in (2, 10, 2)]
n_dev = ctx.array(np.array([P], np.int32))
s2 = avod.make_scratch(P)
timed('stage-2 head, %d proposals' % P, lambda: avod.forward(ctx, r1, r2, P, n_dev, outs, s2), avod.flops(P))

corr = EarlyFusionFcLayers(ctx, hp['corr'], outputs=('off_out',), dtype=dt)
rc = ctx.array(rng.normal(size=(P, corr.in_ld)).astype(np.float32))
oc = [ctx.empty((P, 3), np.float32)]
s3 = corr.make_scratch(P)
timed('correlation head, %d proposals' % P, lambda: corr.forward(ctx, rc, None, P, n_dev, oc, s3), corr.flops(P))
