#!/usr/bin/env python3
"""Time dodt_nms at the RPN's size (n ~ 5.5k candidates -> 1024, thr 0.8) and at 20k, HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dodt_amd import device, ops  # noqa: E402

ctx = device.default_context()
rng = np.random.default_rng(0)
for n, k, thr in ((5500, 1024, 0.8), (1024, 100, 0.01), (20000, 1024, 0.8), (89600, 1024, 0.8)):
    c = rng.uniform(0, 1, size=(n, 2))
    hw = rng.uniform(0.01, 0.05, size=(n, 2))
    boxes = np.concatenate([c - hw, c + hw], 1).astype(np.float32)
    scores = rng.uniform(size=n).astype(np.float32)
    d_b, d_s = ctx.array(boxes), ctx.array(scores)
    d_sel, d_cnt = ctx.empty((k,), np.int32), ctx.zeros((1,), np.int32)
    for _ in range(3):
        ops.nms(ctx, d_b, d_s, n, None, k, thr, d_sel, d_cnt)
    ctx.sync()
    ctx.timer_start()
    for _ in range(20):
        ops.nms(ctx, d_b, d_s, n, None, k, thr, d_sel, d_cnt)
    us = ctx.timer_stop() / 20 * 1e3
    print('nms n=%6d k=%5d thr=%.2f: %7.1f us, %d kept' % (n, k, thr, us, int(d_cnt.download()[0])))
