#!/usr/bin/env python3
"""Time the correlation kernel at the DODT size (700,800,32), HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dodt_amd import device, ops  # noqa: E402

ctx = device.default_context()
rng = np.random.default_rng(0)
a = ctx.array(rng.normal(size=(700, 800, 32)).astype(np.float32))
b = ctx.array(rng.normal(size=(700, 800, 32)).astype(np.float32))
out = ctx.empty((700, 800, 25), np.float32)
for _ in range(3):
    ops.correlation(ctx, a, b, (700, 800, 32), 5, 2, 5, out)
ctx.sync()
ctx.timer_start()
for _ in range(20):
    ops.correlation(ctx, a, b, (700, 800, 32), 5, 2, 5, out)
ms = ctx.timer_stop() / 20
mb = (2 * 700 * 800 * 32 + 700 * 800 * 25) * 4 / 1e6
print('correlation %.1f us, %.0f MB algorithmic -> %.2f TB/s' % (ms * 1e3, mb, mb / ms / 1e6 * 1e3))
