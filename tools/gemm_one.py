#!/usr/bin/env python3
"""One FC layer in a loop, for counter passes: python tools/gemm_one.py [M K N reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dodt_amd import device, ops  # noqa: E402

M, K, N, reps = (int(v) for v in (sys.argv[1:5] + ['1024', '2048', '2048', '10'][len(sys.argv) - 1:]))
ctx = device.default_context()
rng = np.random.default_rng(0)
x = ctx.array(rng.normal(size=(M, K)).astype(np.float32))
fc = ops.FullyConnected(ctx, rng.normal(size=(K, N)).astype(np.float32), np.zeros(N, np.float32), True,
                        dtype=os.environ.get('DODT_GEMM_ONE_DTYPE', 'f32'))
y = ctx.empty((M, N), np.float32)
for _ in range(reps):
    fc.forward(x, M, y)
ctx.sync()
