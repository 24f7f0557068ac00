"""Plain-VGG extractors of the AVOD cars_example configuration (SURVEY 8a rows a8'/a9',
BASELINE.json configs[0]) on the GPU against the oracle: avod/core/feature_extractors/
bev_vgg.py:34-118, img_vgg.py:33-120, bottleneck avod/core/models/rpn_model.py:251-267.
Needs an MI355X.  Tolerances as in test_gpu_conv.py (1e-4 of each layer's scale); the oracle's
conv / pool / resize restate TF-1.3 semantics ("parity unpinned", oracle/extractors.py)."""
import numpy as np
import pytest

from dodt_amd import synth
from dodt_amd.core.feature_extractors.vgg import BevVgg, ImgVgg
from oracle import extractors as oext
from oracle import tfops

pytestmark = pytest.mark.gpu
ENCODER_LAYERS = [n for n in synth.PYRAMID_LAYERS if n.startswith('conv')]


def _close(got, want, name, rel=1e-4):
    scale = float(np.abs(want).max()) + 1e-12
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert err <= rel * scale, '%s: max abs err %g vs scale %g' % (name, err, scale)


@pytest.mark.parametrize('cls,h,w,c', [(BevVgg, 44, 52, 6),      # 44 -> 22 -> 11 -> 5 (odd pools)
                                       (BevVgg, 56, 64, 6),
                                       (ImgVgg, 60, 198, 3)])     # 198 -> 99 -> 49 -> 24
def test_plain_vgg_small_all_layers(cls, h, w, c):
    rng = np.random.default_rng(h * w + c)
    x = rng.uniform(-1, 1, size=(2, h, w, c)).astype(np.float32)
    params = synth.pyramid_params(c, seed=42, plain=True)
    ex = cls()
    ex.load_params(params)
    feat, ends = ex.build(x, with_bottleneck=True)
    oh, ow = int(h / 8 * 4), int(w / 8 * 4)
    assert feat.shape == (2, oh, ow, 256) and ends['bottleneck'].shape == (2, oh, ow, 1)
    for f in range(2):
        col = {}
        want = oext.vgg_plain(x[f], params, collect=col)
        for name in ENCODER_LAYERS:
            _close(ex.activation(name)[f], col[name], name)
        # the upsampling itself is exact arithmetic on conv4_3; checked on the device's own map
        own = tfops.resize_bilinear(ex.activation('conv4_3')[f], oh, ow)
        assert np.array_equal(feat[f], own), 'resize_bilinear of the device map'
        _close(feat[f], want, 'feature_maps')
        _close(ends['bottleneck'][f], oext.bottleneck_1x1(feat[f], params['bottleneck']),
               'bottleneck')
    ex.close()


def test_bev_vgg_full_size():
    """(1, 700, 800, 6) -> (1, 350, 400, 256): conv4 is 87 x 100 (175 floors to 87), the
    upsampled map 350 x 400 (bev_vgg.py:102-112).  79.20 GFLOP."""
    rng = np.random.default_rng(21)
    x = np.zeros((1, 700, 800, 6), np.float32)
    m = rng.uniform(size=x.shape) < 0.02
    x[m] = rng.uniform(0, 1, size=int(m.sum())).astype(np.float32)
    params = synth.pyramid_params(6, seed=42, plain=True)
    ex = BevVgg()
    ex.load_params(params)
    feat, ends = ex.build(x, with_bottleneck=True)
    assert feat.shape == (1, 350, 400, 256)
    col = {}
    want = oext.vgg_plain(x[0], params, collect=col)
    assert col['conv4_3'].shape == (87, 100, 256)
    for name in ('conv1_2', 'conv3_3', 'conv4_3'):
        _close(ex.activation(name)[0], col[name], name)
    _close(feat[0], want, 'feature_maps')
    _close(ends['bottleneck'][0], oext.bottleneck_1x1(want, params['bottleneck']), 'bottleneck')
    assert abs(ex.flops() - 79.20e9) < 0.05e9
    ex.close()


def test_img_vgg_full_size():
    """(1, 480, 1590, 3) -> (1, 240, 795, 256): 1590 -> 795 -> 397 -> 198.  106.65 GFLOP."""
    params = synth.pyramid_params(3, seed=142, plain=True)
    ex = ImgVgg()
    ex.load_params(params)
    pre = ex.preprocess_input(synth.image_frame(0, 1)[None], (480, 1590))
    assert np.array_equal(pre[0], tfops.img_preprocess(synth.image_frame(0, 1), 480, 1590))
    feat, ends = ex.build(pre, with_bottleneck=True)
    assert feat.shape == (1, 240, 795, 256)
    col = {}
    want = oext.vgg_plain(pre[0], params, collect=col)
    assert col['conv4_3'].shape == (60, 198, 256)
    for name in ('conv1_1', 'conv2_2', 'conv4_3'):
        _close(ex.activation(name)[0], col[name], name)
    _close(feat[0], want, 'feature_maps')
    _close(ends['bottleneck'][0], oext.bottleneck_1x1(want, params['bottleneck']), 'bottleneck')
    assert abs(ex.flops() - 106.65e9) < 0.05e9
    ex.close()


def test_plain_vgg_errors():
    with pytest.raises(ValueError):
        BevVgg(conv_dtype='bf16')
    ex = BevVgg()
    with pytest.raises(ValueError):
        ex.build(np.zeros((1, 8, 8, 6), np.float32))         # smaller than 16 x 16
    ex.load_params(synth.pyramid_params(6, plain=False))     # pyramid weights: unknown layers
    with pytest.raises(ValueError):
        ex.build(np.zeros((1, 32, 32, 6), np.float32))
