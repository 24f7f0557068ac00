#!/usr/bin/env python3
"""Runs both feature extractors a few times (for rocprofv3 --kernel-trace) and
prints per-layer time / TFLOP/s from HIP-event timing of whole forwards."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from dodt_amd import config, device, synth  # noqa: E402
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dt = sys.argv[2] if len(sys.argv) > 2 else 'f32'
ctx = device.default_context()
cfg = config.PYRAMID_DODT
bev = BevVggPyr(ctx=ctx, conv_dtype=dt); bev.load_params(synth.pyramid_params(6, 42)); bev._ensure(2, 700, 800, 6)
img = ImgVggPyr(ctx=ctx, conv_dtype=dt); img.load_params(synth.pyramid_params(3, 142)); img._ensure(2, 360, 1200, 4)
rng = np.random.default_rng(0)
p, s = bev.input_view()
for f in range(2):
    x = np.zeros((700, 800, 6), np.float32)
    m = rng.uniform(size=x.shape) < 0.02
    x[m] = rng.uniform(size=int(m.sum()))
    ctx.wrap(p + 4 * s * f, x.shape).upload(x)
p, s = img.input_view()
for f in range(2):
    ctx.wrap(p + 4 * s * f, (360, 1200, 4)).upload(rng.normal(0, 60, size=(360, 1200, 4)).astype(np.float32))
fb, bb = ctx.empty((2, 700, 800, 32)), ctx.empty((2, 700, 800, 1))
fi, bi = ctx.empty((2, 360, 1200, 32)), ctx.empty((2, 360, 1200, 1))
for net, f, b, name in ((bev, fb, bb, 'bev'), (img, fi, bi, 'img')):
    net.forward_device(None, f, b)
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        net.forward_device(None, f, b)
    ms = ctx.timer_stop() / reps
    print('%s forward %.3f ms  %.1f TFLOP/s' % (name, ms, net.flops() / ms / 1e9))
