#!/usr/bin/env python3
"""Stage timeline of the frame-pair pipeline from HIP-event marks (no profiler attached).
Prints, for three consecutive steady-state steps, when each stage starts and ends relative
to the start of the middle step's BEV convs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dodt_amd import config, device, synth  # noqa: E402
from dodt_amd.pipeline import FramePairPipeline  # noqa: E402

computed = '--injected' not in sys.argv
conv_dtype = 'bf16' if '--bf16' in sys.argv else 'f32'
head_dtype = 'bf16' if '--bf16-heads' in sys.argv else 'f32'
ctx = device.default_context()
pipe = FramePairPipeline(ctx, config.PYRAMID_DODT, **synth.pipeline_weights(config.PYRAMID_DODT),
                         head_params=synth.head_params() if computed else None,
                         conv_dtype=conv_dtype, head_dtype=head_dtype)
frames = (0, 2)
pts = [ctx.array(synth.lidar_frame(0, f)) for f in frames]
imgs = [ctx.array(synth.image_frame(0, f)) for f in frames]
heads = None if computed else [{k: ctx.array(v) for k, v in
                                synth.head_outputs(0, f, pipe.n_all, pipe.P).items()}
                               for f in frames]
n = [120000, 120000]
T = 8
pipe.mark_steps = (T - 1, T, T + 1) if '--two' in sys.argv else (T,)
ahead = '--lookahead' in sys.argv
for i in range(T + 4):
    pipe.run(pts, n, imgs, heads, lookahead=(pts, n, imgs) if ahead else None)
pipe.finish()
ctx.sync()
ref = pipe.marks['%d:bev_start' % T]
rows = []
for name, (c, slot) in pipe.marks.items():
    rows.append((ref[0].elapsed_ms(ref[1], c, slot), name))
for t, name in sorted(rows):
    print('%9.3f ms  %s' % (t, name))
