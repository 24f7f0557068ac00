#!/usr/bin/env python3
"""Per-layer stand-alone times of both extractors from dodt_extractor_forward_timed (HIP event pair per
layer): `python tools/layer_table.py [reps] [dtype]`.  Honours DODT_CONV_WINO etc."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from dodt_amd import device, synth  # noqa: E402
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dt = sys.argv[2] if len(sys.argv) > 2 else 'f32'
shared = os.environ.get('LT_SHARED', '1') == '1'
ctx = device.default_context()
bev = BevVggPyr(ctx=ctx, conv_dtype=dt, shared_gpu=shared); bev.load_params(synth.pyramid_params(6, 42)); bev._ensure(2, 700, 800, 6)
img = ImgVggPyr(ctx=ctx, conv_dtype=dt, shared_gpu=shared); img.load_params(synth.pyramid_params(3, 142)); img._ensure(2, 360, 1200, 4)
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
tot = 0.0
for net, f, b, name in ((bev, fb, bb, 'bev'), (img, fi, bi, 'img')):
    net.forward_device(None, f, b)
    acc = None
    for _ in range(reps):
        cur = net.forward_timed(None, f, b)
        if acc is None:
            acc = cur
        else:
            for a_, c_ in zip(acc, cur):
                a_['ms'] += c_['ms']
    print('%s: layer, kernel, items, us, executed TFLOP/s (frac of 157.3), direct-equivalent TFLOP/s' % name)
    for a_ in acc:
        ms = a_['ms'] / reps
        tot += ms
        print('  %-16s %-26s %5d %7.1f  %6.1f (%.2f)  %6.1f' % (a_['name'], a_['kernel'], a_['items'], ms * 1e3,
              a_['flops_executed'] / ms / 1e9, a_['flops_executed'] / ms / 1e9 / 157.3, a_['flops_direct'] / ms / 1e9))
print('sum of layers %.3f ms' % tot)
