"""bf16 conv path (DODT_EXTRACTOR_BF16, BASELINE.json configs[2]) against its oracle.

The oracle restates the device's scheme (oracle/extractors.py `_cbr`): weights and stored
activations rounded to bf16 (nearest even), exact products, fp32 sums, fp32 BN + ReLU.  The
two differ only where a pre-rounding fp32 sum lands within its summation-order noise (~1e-6
relative) of a bf16 rounding boundary: a rare 1-ulp flip (2^-8 relative) that later layers
average out.  Bars, stated per layer as fractions of the layer's activation scale:
max error <= 1.6e-2 (two bf16 ulps at the top of the range), mean error <= 5e-4 (flips
propagate: none in the first layers, a few per cent of the elements after ten).  Against the fp32 oracle (the reference's arithmetic) the bf16
feature maps agree to 3e-2 of their scale -- the price of the bf16 path, not a parity claim.
"""
import numpy as np
import pytest

from dodt_amd import device, synth
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr
from oracle import extractors as oext
from oracle import tfops

pytestmark = pytest.mark.gpu


def _bars(got, want, name, max_rel=1.6e-2, mean_rel=5e-4):
    assert got.shape == want.shape, name
    scale = float(np.abs(want).max()) + 1e-12
    d = np.abs(got - want)
    assert d.max() <= max_rel * scale, '%s: max err %g vs scale %g' % (name, d.max(), scale)
    assert d.mean() <= mean_rel * scale, '%s: mean err %g vs scale %g' % (name, d.mean(), scale)


def _run(cls, x, params, pad_top):
    ex = cls(conv_dtype='bf16')
    ex.load_params(params)
    feat, ends = ex.build(x, with_bottleneck=True)
    # conv1_1 runs folded into conv1_2's launch by default (conv3x3_bf16_first2_kernel): its map is not stored -- conv1_2's
    # bars check it --, and its products are hi + lo bf16 pairs, which the oracle restates (first_layer='split');
    # test_two_launch_first_layers_match_the_same_bars stores conv1_1 and checks the fp32 first layer
    folded = ex.first_layers_folded
    collect = [dict() for _ in range(x.shape[0])]
    want = np.stack([oext.vgg_pyramid(x[f], params, pad_top=pad_top, collect=collect[f], conv_dtype='bf16',
                                      first_layer='split' if folded else 'fp32') for f in range(x.shape[0])])
    for name in synth.PYRAMID_LAYERS[:-1]:
        w = np.stack([c[name] for c in collect])
        if folded and name == 'conv1_1':
            with pytest.raises(ValueError, match='folded'):
                ex.activation(name)
            continue
        got = ex.activation(name)
        assert np.array_equal(got, tfops.round_bf16(got)), name   # stored maps ARE bf16
        _bars(got, w, name)
    _bars(feat, want, 'feature_maps')
    wb = np.stack([oext.bottleneck_1x1(want[f], params['bottleneck'])
                   for f in range(x.shape[0])])
    _bars(ends['bottleneck'], wb, 'bottleneck', max_rel=3e-2, mean_rel=2e-3)
    # distance to the reference's fp32 arithmetic
    f32 = np.stack([oext.vgg_pyramid(x[f], params, pad_top=pad_top) for f in range(x.shape[0])])
    rel = np.abs(feat - f32).max() / (np.abs(f32).max() + 1e-12)
    assert rel <= 3e-2, 'bf16 vs fp32 feature maps: %g' % rel
    ex.close()
    import os
    assert folded == all(os.environ.get(k, '1') != '0' for k in ('DODT_CONV_BF16_FIRST2', 'DODT_CONV_BF16_STREAM', 'DODT_CONV_BF16_DMA'))
    return rel


@pytest.mark.parametrize('h,w', [(60, 96), (28, 40)])
def test_bev_pyramid_bf16_all_layers(h, w):
    rng = np.random.default_rng(h * w + 1)
    x = rng.uniform(0, 1, size=(2, h, w, 6)).astype(np.float32)
    x[x < 0.7] = 0
    _run(BevVggPyr, x, synth.pyramid_params(6, seed=42), 4)


def test_img_pyramid_bf16_all_layers():
    rng = np.random.default_rng(78)
    x = rng.normal(0, 60, size=(2, 48, 160, 3)).astype(np.float32)
    _run(ImgVggPyr, x, synth.pyramid_params(3, seed=142), 0)


def test_bf16_is_opt_in_and_checked():
    with pytest.raises(ValueError):
        BevVggPyr(conv_dtype='fp8')
    ex = BevVggPyr()
    assert ex._bf16 is False


def _child(env_extra, deselect):
    import os
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-m', 'gpu', '-q',
                        '-x', '-k', deselect], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_two_launch_first_layers_match_the_same_bars():
    """By default conv1_1 is computed inside conv1_2's launch (hi + lo bf16 MFMAs on the raw input, conv1_1's map kept
    in LDS).  DODT_CONV_BF16_FIRST2=0 runs the two layers as two launches (conv1_1 on the fp32 MFMA, its bf16 map stored):
    every layer, conv1_1 included, against the same bars; DODT_CONV_BF16_STREAM=0 on top puts conv1_2 and pyramid_fusion1
    back on the chunk-ring kernel."""
    _child({'DODT_CONV_BF16_FIRST2': '0'}, 'all_layers')
    _child({'DODT_CONV_BF16_FIRST2': '0', 'DODT_CONV_BF16_STREAM': '0'}, 'all_layers')


def test_template_bf16_kernel_matches_the_same_bars():
    """The 3x3 stride-1 bf16 layers run on conv_bf16_dma.h by default (LDS-DMA staging,
    row-sliding fragment reuse; every other test of this file).  DODT_CONV_BF16_DMA=0 selects the
    fp32 kernel template's bf16 instantiation instead; the library reads the switch once per
    process, so the bf16 all-layers check runs in a child process with it."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DODT_CONV_BF16_DMA='0')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-m', 'gpu', '-q',
                        '-x', '-k', 'not template_bf16'], env=env, cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_bev_pyramid_bf16_full_size():
    """One 700 x 800 BEV frame through the bf16 path at its real size -- 2 950 tiles of the folded first layers and of the
    streaming kernel, four items per workgroup, the grouped queues with all eight groups, partial tiles at the bottom
    (704 = 117 x 6 + 2 rows) -- against the same bars."""
    rng = np.random.default_rng(7008)
    x = rng.uniform(0, 1, size=(1, 700, 800, 6)).astype(np.float32)
    x[x < 0.97] = 0
    _run(BevVggPyr, x, synth.pyramid_params(6, seed=42), 4)


def test_img_pyramid_bf16_full_size():
    """... and one 360 x 1200 image frame (the padded-image form of the folded first layers: K = 36, two taps per lane half)."""
    rng = np.random.default_rng(36012)
    x = rng.normal(0, 60, size=(1, 360, 1200, 3)).astype(np.float32)
    _run(ImgVggPyr, x, synth.pyramid_params(3, seed=142), 0)
