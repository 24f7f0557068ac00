"""HIP feature extractors (fp32 MFMA convs) against the oracle.  Needs an MI355X.

Tolerance: the north_star asks for 1e-4 on float32 box regressions; the conv
stacks are checked to a relative 1e-4 of the activation scale per layer (the
GPU sums each output as one k-ordered fmaf chain, the oracle as nine partial
GEMMs -- same terms, different association)."""
import numpy as np
import pytest

from dodt_amd import device, synth
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr
from oracle import extractors as oext
from oracle import tfops

pytestmark = pytest.mark.gpu


def _close(got, want, name, rel=1e-4):
    scale = float(np.abs(want).max()) + 1e-12
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape, name
    assert err <= rel * scale, '%s: max abs err %g vs scale %g' % (name, err, scale)


def _check_all_layers(ex, params, x, pad_top):
    collect = {}
    for f in range(x.shape[0]):
        c = {}
        oext.vgg_pyramid(x[f], params, pad_top=pad_top, collect=c)
        for k, v in c.items():
            collect.setdefault(k, []).append(v)
    for name in synth.PYRAMID_LAYERS[:-1]:     # the last layer is the returned map
        _close(ex.activation(name), np.stack(collect[name]), name)


@pytest.mark.parametrize('h,w', [(60, 96), (28, 40), (44, 72), (12, 136)])
def test_bev_pyramid_small_all_layers(h, w):
    """(h + 4) x w must be divisible by 8; every layer is compared."""
    rng = np.random.default_rng(h * w)
    x = rng.uniform(0, 1, size=(2, h, w, 6)).astype(np.float32)
    x[x < 0.7] = 0                                   # BEV maps are sparse
    params = synth.pyramid_params(6, seed=42)
    ex = BevVggPyr()
    ex.load_params(params)
    feat, ends = ex.build(x, with_bottleneck=True)
    _check_all_layers(ex, params, x, pad_top=4)
    want = np.stack([oext.vgg_pyramid(x[f], params, pad_top=4) for f in range(2)])
    _close(feat, want, 'feature_maps')
    wb = np.stack([oext.bottleneck_1x1(want[f], params['bottleneck']) for f in range(2)])
    _close(ends['bottleneck'], wb, 'bottleneck')
    ex.close()


def test_img_pyramid_small_all_layers():
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, size=(53, 170, 3), dtype=np.uint8)
    ex = ImgVggPyr()
    pre = ex.preprocess_input(img[None], (48, 160))
    want_pre = tfops.img_preprocess(img, 48, 160)
    assert np.array_equal(pre[0], want_pre)           # unfused fp32: bit exact
    params = synth.pyramid_params(3, seed=142)
    ex.load_params(params)
    x = np.stack([pre[0], pre[0][::-1].copy()])
    feat, _ = ex.build(x)
    _check_all_layers(ex, params, x, pad_top=0)
    ex.close()


def test_img_preprocess_full_size():
    img = synth.image_frame(0, 0)
    ex = ImgVggPyr()
    pre = ex.preprocess_input(img[None], (360, 1200))
    assert pre.shape == (1, 360, 1200, 3)
    assert np.array_equal(pre[0], tfops.img_preprocess(img, 360, 1200))


def test_bev_pyramid_full_size():
    """(2, 700, 800, 6) -> (2, 700, 800, 32): the bench shape, one frame checked
    against the oracle end to end and at the deepest layers."""
    rng = np.random.default_rng(11)
    x = np.zeros((2, 700, 800, 6), np.float32)
    m = rng.uniform(size=x.shape) < 0.02
    x[m] = rng.uniform(0, 1, size=int(m.sum())).astype(np.float32)
    params = synth.pyramid_params(6, seed=42)
    ex = BevVggPyr()
    ex.load_params(params)
    feat, ends = ex.build(x, with_bottleneck=True)
    assert feat.shape == (2, 700, 800, 32)
    c = {}
    want = oext.vgg_pyramid(x[1], params, pad_top=4, collect=c)
    for name in ('conv1_2', 'conv4_3', 'upconv3', 'pyramid_fusion2'):
        _close(ex.activation(name)[1], c[name], name)
    _close(feat[1], want, 'feature_maps')
    _close(ends['bottleneck'][1], oext.bottleneck_1x1(want, params['bottleneck']), 'bottleneck')
    assert abs(ex.flops() / 2 - 131.71e9) < 0.05e9
    ex.close()


def test_img_pyramid_full_size():
    params = synth.pyramid_params(3, seed=142)
    ex = ImgVggPyr()
    ex.load_params(params)
    pre = ex.preprocess_input(synth.image_frame(0, 1)[None], (360, 1200))
    x = np.concatenate([pre, pre[:, ::-1]], axis=0)
    feat, _ = ex.build(x)
    assert feat.shape == (2, 360, 1200, 32)
    c = {}
    want = oext.vgg_pyramid(x[0], params, pad_top=0, collect=c)
    for name in ('conv1_1', 'conv3_3', 'conv4_3', 'upconv1'):
        _close(ex.activation(name)[0], c[name], name)
    _close(feat[0], want, 'feature_maps')
    assert abs(ex.flops() / 2 - 100.28e9) < 0.05e9
    ex.close()


def test_extractor_errors():
    ex = BevVggPyr()
    with pytest.raises(ValueError):
        ex.build(np.zeros((1, 30, 40, 6), np.float32))       # 34 x 40 not divisible by 8
    ex2 = BevVggPyr()
    with pytest.raises(ValueError):
        ex2.build(np.zeros((1, 28, 40, 6), np.float32))      # weights not set
    with pytest.raises(NotImplementedError):
        ex2.build(np.zeros((1, 28, 40, 6), np.float32), is_training=True)


@pytest.mark.parametrize('mode', ['0', '1', '2', '4'])
def test_other_fp32_conv_paths_match_oracle(mode):
    """Every form of the fp32 3x3 stride-1 layers against the same oracle at the same 1e-4 bar, each
    in a child process (the library reads DODT_CONV_WINO once per process): the direct implicit-GEMM
    kernels (0), Winograd F(2x2,3x3) with 128 accumulators (2: the default, what every other test
    in this file runs) and its 256-accumulator variants (1), Winograd F(4x4,3x3) (4: the fastest;
    1.5e-6 of a layer's scale from the oracle where the others are at 2e-7, which is why it is not
    the default -- tests/test_gpu_heads.py::test_pair_free_running_by_conv_mode)."""
    import os
    import subprocess
    import sys
    code = '''
import numpy as np, sys
sys.path.insert(0, %r)
from dodt_amd import synth
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr
from oracle import extractors as oext
rng = np.random.default_rng(60 * 96)
x = rng.uniform(0, 1, size=(2, 60, 96, 6)).astype(np.float32)
x[x < 0.7] = 0
params = synth.pyramid_params(6, seed=42)
ex = BevVggPyr(); ex.load_params(params)
feat, ends = ex.build(x, with_bottleneck=True)
worst = 0.0
for f in range(2):
    c = {}
    want = oext.vgg_pyramid(x[f], params, pad_top=4, collect=c)
    for name in synth.PYRAMID_LAYERS[:-1]:
        got = ex.activation(name)[f]
        worst = max(worst, np.abs(got - c[name]).max() / (np.abs(c[name]).max() + 1e-12))
    worst = max(worst, np.abs(feat[f] - want).max() / (np.abs(want).max() + 1e-12))
print('WORST %%.3e' %% worst)
assert worst < 1e-4
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DODT_CONV_WINO=mode)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'WORST' in r.stdout
