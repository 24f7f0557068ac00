#!/usr/bin/env python3
"""Diagnostic: conv1_2's map of the bf16 image / BEV pyramid, printed as a coarse error map against the oracle (run once
with DODT_CONV_BF16_FIRST2=1 and once with 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dodt_amd import synth
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr
from oracle import extractors as oext

which = sys.argv[1] if len(sys.argv) > 1 else 'img'
rng = np.random.default_rng(78)
if which == 'img':
    x = rng.normal(0, 60, size=(2, 48, 160, 3)).astype(np.float32); cls, params, pad = ImgVggPyr, synth.pyramid_params(3, seed=142), 0
else:
    x = rng.uniform(0, 1, size=(2, 60, 96, 6)).astype(np.float32); x[x < 0.7] = 0; cls, params, pad = BevVggPyr, synth.pyramid_params(6, seed=42), 4
ex = cls(conv_dtype='bf16'); ex.load_params(params)
ex.build(x, with_bottleneck=True)
col = [dict() for _ in range(2)]
for f in range(2):
    oext.vgg_pyramid(x[f], params, pad_top=pad, collect=col[f], conv_dtype='bf16', first_layer='split' if ex.first_layers_folded else 'fp32')
got = ex.activation('conv1_2'); want = np.stack([c['conv1_2'] for c in col])
d = np.abs(got - want).max(axis=3)
print('folded', ex.first_layers_folded, 'max', d.max(), 'scale', np.abs(want).max())
bad = np.argwhere(~(d < 0.02 * np.abs(want).max()))
print('bad pixels', len(bad), 'of', d.size)
for f in range(2):
    print('frame', f)
    for y in range(d.shape[1]):
        print(''.join('#' if not (d[f, y, xx] < 0.02 * np.abs(want).max()) else '.' for xx in range(d.shape[2])))
