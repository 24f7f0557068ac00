import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from dodt_amd import config, device, synth, ops
from dodt_amd.pipeline import FramePairPipeline
import dodt_amd.pipeline as P
ctx = device.default_context()
pipe = FramePairPipeline(ctx, config.PYRAMID_DODT, **synth.pipeline_weights(config.PYRAMID_DODT), head_params=synth.head_params())
pts = [ctx.array(synth.lidar_frame(0, f)) for f in (0, 2)]
imgs = [ctx.array(synth.image_frame(0, f)) for f in (0, 2)]
n = [120000, 120000]
# wrap every ops function with a timer
acc = {}
for name in dir(ops):
    fn = getattr(ops, name)
    if callable(fn) and not name.startswith('_') and name[0].islower():
        def mk(fn, name):
            def w(*a, **k):
                t = time.perf_counter(); r = fn(*a, **k); acc.setdefault(name, [0, 0]); acc[name][0] += time.perf_counter() - t; acc[name][1] += 1; return r
            return w
        setattr(ops, name, mk(fn, name))
for cls, meths in ((ops.FullyConnected, ['forward']), (device.Context, ['wait_for']), (type(pipe.bev_net).__mro__[1], ['forward_device'])):
    for m in meths:
        fn = getattr(cls, m)
        def mk(fn, name):
            def w(*a, **k):
                t = time.perf_counter(); r = fn(*a, **k); acc.setdefault(name, [0, 0]); acc[name][0] += time.perf_counter() - t; acc[name][1] += 1; return r
            return w
        setattr(cls, m, mk(fn, cls.__name__ + '.' + m))
for i in range(3):
    pipe.run(pts, n, imgs)
pipe.finish(); ctx.sync()
acc.clear()
t0 = time.perf_counter()
N = 10
for i in range(N):
    pipe.run(pts, n, imgs)
t1 = time.perf_counter()
pipe.finish(); ctx.sync()
t2 = time.perf_counter()
print('host %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print('%-28s %7.1f us/step %5.1f calls/step %6.1f us/call' % (k, t / N * 1e6, c / N, t / c * 1e6))
