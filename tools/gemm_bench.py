#!/usr/bin/env python3
"""Time the fully connected kernel at the heads' shapes (HIP events)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dodt_amd import device, ops  # noqa: E402

ctx = device.default_context()
dt = sys.argv[1] if len(sys.argv) > 1 else 'f32'
rng = np.random.default_rng(0)
for M, K, N in [(1024, 1568, 2048), (1024, 2048, 2048), (1024, 1225, 2048), (1024, 1248, 2048), (1024, 2048, 10),
                (5500, 9, 512), (5500, 256, 256), (5500, 256, 6), (2048, 1568, 2048), (2048, 2048, 2048), (4096, 2048, 2048)]:
    x = ctx.array(rng.normal(size=(M, K)).astype(np.float32))
    rows = dt == 'bf16rows'       # the bf16 heads' round-4 path: bf16 rows in, bf16 rows out
    fc = ops.FullyConnected(ctx, rng.normal(size=(K, N)).astype(np.float32),
                            np.zeros(N, np.float32), True, dtype='bf16' if rows else dt)
    y = ctx.empty((M, N), np.float32)
    if rows:
        ld = fc.bf16_row_elems()
        if not ld or N < 128:
            fc.close()
            continue
        x16 = ctx.zeros((M, ld), np.uint16)
        ops.rows_to_bf16(ctx, x, None, M, None, K, K, x16, ld)
        y16 = ctx.empty((M, N), np.uint16)
        run = lambda: fc.forward_bf16(x16, M, y16, ldx=ld, y_bf16=True)
    else:
        run = lambda: fc.forward(x, M, y)
    for _ in range(3):
        run()
    ctx.sync()
    reps = 20
    ctx.timer_start()
    for _ in range(reps):
        run()
    ms = ctx.timer_stop() / reps
    print('M=%5d K=%5d N=%5d  %8.1f us  %7.2f TFLOP/s' % (M, K, N, ms * 1e3,
                                                          fc.flops(M) / ms / 1e9))
    fc.close()
