"""Split mode of the conv path (DODT_EXTRACTOR_SPLIT, conv_dtype='f32s'): hi + lo bf16 pairs
on the bf16 MFMA.  It claims fp32-grade results, so it runs the fp32 tests' protocol against
the fp32 oracle at the fp32 tolerance: every layer within 1e-4 of its activation scale."""
import numpy as np
import pytest

from dodt_amd import synth
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr
from oracle import extractors as oext

pytestmark = pytest.mark.gpu


def _close(got, want, name, rel=1e-4):
    scale = float(np.abs(want).max()) + 1e-12
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape, name
    assert err <= rel * scale, '%s: max abs err %g vs scale %g' % (name, err, scale)
    return err / scale


def _run(cls, x, params, pad_top):
    ex = cls(conv_dtype='f32s')
    ex.load_params(params)
    feat, ends = ex.build(x, with_bottleneck=True)
    collect = [dict() for _ in range(x.shape[0])]
    want = np.stack([oext.vgg_pyramid(x[f], params, pad_top=pad_top, collect=collect[f])
                     for f in range(x.shape[0])])
    worst = 0.0
    for name in synth.PYRAMID_LAYERS[:-1]:
        worst = max(worst, _close(ex.activation(name), np.stack([c[name] for c in collect]), name))
    worst = max(worst, _close(feat, want, 'feature_maps'))
    wb = np.stack([oext.bottleneck_1x1(want[f], params['bottleneck']) for f in range(x.shape[0])])
    _close(ends['bottleneck'], wb, 'bottleneck')
    ex.close()
    return worst


@pytest.mark.parametrize('h,w', [(60, 96), (28, 40)])
def test_bev_pyramid_split_all_layers_at_fp32_tolerance(h, w):
    rng = np.random.default_rng(h * w + 2)
    x = rng.uniform(0, 1, size=(2, h, w, 6)).astype(np.float32)
    x[x < 0.7] = 0
    worst = _run(BevVggPyr, x, synth.pyramid_params(6, seed=42), 4)
    print('worst layer error / scale: %.2e' % worst)


def test_img_pyramid_split_all_layers_at_fp32_tolerance():
    rng = np.random.default_rng(79)
    x = rng.normal(0, 60, size=(2, 48, 160, 3)).astype(np.float32)
    worst = _run(ImgVggPyr, x, synth.pyramid_params(3, seed=142), 0)
    print('worst layer error / scale: %.2e' % worst)
