#!/usr/bin/env python3
"""Image-net forward on a custom size (for per-kernel occupancy experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dodt_amd import device, synth
from dodt_amd.core.feature_extractors.vgg_pyramid import ImgVggPyr
h, w, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = device.default_context()
net = ImgVggPyr(ctx=ctx); net.load_params(synth.pyramid_params(3, 142)); net._ensure(b, h, w, 4)
f = ctx.empty((b, h, w, 32))
for _ in range(3):
    net.forward_device(None, f, None)
ctx.sync()
