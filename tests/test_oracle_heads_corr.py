"""CPU checks of the oracle pieces that no reference fixture pins (correlation, FC layers,
bf16 rounding): each is compared with an independent restatement written here from the
reference's source, on inputs small enough for plain loops."""
import numpy as np
import pytest

from oracle import heads as oheads
from oracle import tfops


def _correlation_loops(a, b, max_displacement, stride_2, pad):
    """Straight from the CUDA kernel's index arithmetic (correlation_kernel.cu.cc:21-119 with
    kernel_size 1, stride_1 1): one output pixel and displacement at a time."""
    h, w, c = a.shape
    d = max_displacement
    r = d // stride_2
    gw = 2 * r + 1
    oh, ow = h + 2 * pad - 2 * d, w + 2 * pad - 2 * d

    def at(img, y, x):            # the padded image (pad.cu.cc:14-73): zeros outside
        y, x = y - pad, x - pad
        if 0 <= y < h and 0 <= x < w:
            return img[y, x]
        return np.zeros(c, np.float32)
    out = np.zeros((oh, ow, gw * gw), np.float32)
    for y in range(oh):
        for x in range(ow):
            y1, x1 = y + d, x + d                       # position in the padded image
            for tc in range(gw * gw):
                s2o = (tc % gw - r) * stride_2          # x displacement
                s2p = (tc // gw - r) * stride_2         # y displacement
                pa, pb = at(a, y1, x1), at(b, y1 + s2p, x1 + s2o)
                acc = np.float32(0)
                for ch in range(c):
                    acc = np.float32(acc + np.float32(pa[ch] * pb[ch]))
                out[y, x, tc] = acc / np.float32(c)
    return out


@pytest.mark.parametrize('hw,c,md,s2,pad', [((6, 7), 4, 2, 2, 2), ((5, 5), 3, 2, 1, 2),
                                            ((7, 6), 8, 4, 2, 3), ((4, 9), 2, 1, 1, 1)])
def test_correlation_oracle_matches_kernel_index_arithmetic(hw, c, md, s2, pad):
    rng = np.random.default_rng(hw[0] * 31 + c)
    a = rng.normal(size=hw + (c,)).astype(np.float32)
    b = rng.normal(size=hw + (c,)).astype(np.float32)
    got = tfops.correlation(a, b, md, s2, pad)
    want = _correlation_loops(a, b, md, s2, pad)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_correlation_of_shifted_map_peaks_at_the_shift():
    rng = np.random.default_rng(2)
    a = rng.normal(size=(24, 24, 8)).astype(np.float32)
    b = np.roll(a, (2, -4), axis=(0, 1))                # b[y+2, x-4] = a[y, x]
    out = tfops.correlation(a, b, 4, 2, 4)
    centre = out[8:16, 8:16].mean(axis=(0, 1))
    assert int(np.argmax(centre)) == (1 + 2) * 5 + (-2 + 2)    # (dy, dx) = (+2, -4)


def test_round_bf16_is_round_to_nearest_even():
    torch = pytest.importorskip('torch')
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 3, 5000), rng.normal(0, 1e-3, 500),
                        [1.00390625, 1.01171875, -1.00390625, 0.0, 65280.0]]).astype(np.float32)
    want = torch.tensor(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(tfops.round_bf16(x), want)
    # exact halfway cases go to the even neighbour
    assert tfops.round_bf16(np.float32(1.00390625)) == np.float32(1.0)
    assert tfops.round_bf16(np.float32(1.01171875)) == np.float32(1.015625)


def test_fc_oracle_is_xw_plus_b():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(5, 7)).astype(np.float32)
    w = rng.normal(size=(7, 3)).astype(np.float32)
    b = rng.normal(size=3).astype(np.float32)
    want = np.array([[sum(float(x[m, k]) * float(w[k, n]) for k in range(7)) + float(b[n])
                      for n in range(3)] for m in range(5)])
    np.testing.assert_allclose(oheads.fc(x, w, b, relu=False), want, rtol=1e-6)
    assert (oheads.fc(x, w, b, relu=True) >= 0).all()
    # bf16 variant: same thing on rounded operands
    xb, wb = tfops.round_bf16(x), tfops.round_bf16(w)
    np.testing.assert_allclose(oheads.fc(x, w, b, False, 'bf16'), oheads.fc(xb, wb, b, False),
                               rtol=1e-6)


def test_heads_shapes_and_fusion():
    from dodt_amd import synth
    hp = synth.head_params(fc_sizes=(64, 64, 64))
    rng = np.random.default_rng(3)
    bev = rng.uniform(size=(6, 3, 3, 1)).astype(np.float32)
    img = rng.uniform(size=(6, 3, 3, 1)).astype(np.float32)
    obj, off = oheads.rpn_anchor_predictor(bev, img, hp['rpn'])
    assert obj.shape == (6, 2) and off.shape == (6, 6)
    # mean fusion is symmetric in its inputs (avod_fc_layer_utils.py:38-41)
    obj2, off2 = oheads.rpn_anchor_predictor(img, bev, hp['rpn'])
    assert np.array_equal(obj, obj2) and np.array_equal(off, off2)
    r1 = rng.uniform(size=(4, 7, 7, 32)).astype(np.float32)
    r2 = rng.uniform(size=(4, 7, 7, 32)).astype(np.float32)
    cls, o4c, ang = oheads.fusion_fc_early(r1, r2, hp['avod'])
    assert cls.shape == (4, 2) and o4c.shape == (4, 10)
    assert ang.shape == (4, 2)       # box_4ca: ANG_VECS_OUTPUT_SIZE 2 (avod_fc_layer_utils.py:11-17)
    hp4c = synth.head_params(fc_sizes=(64, 64, 64), ang_size=0)     # box_4c: no angle layer
    assert oheads.fusion_fc_early(r1, r2, hp4c['avod'])[2] is None
    assert oheads.corr_fc_early(rng.uniform(size=(4, 7, 7, 25)).astype(np.float32),
                                hp['corr']).shape == (4, 3)


def test_bf16_extractor_oracle_stays_close_to_fp32():
    from dodt_amd import synth
    from oracle import extractors as oext
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 1, size=(20, 24, 6)).astype(np.float32)
    p = synth.pyramid_params(6, 42)
    f32 = oext.vgg_pyramid(x, p, pad_top=4)
    b16 = oext.vgg_pyramid(x, p, pad_top=4, conv_dtype='bf16')
    rel = np.abs(f32 - b16).max() / np.abs(f32).max()
    assert 0 < rel < 3e-2


def test_split_first_layer_restatement_is_fp32_grade():
    """first_layer='split' (the device's folded conv1_1: x and w as hi + lo bf16 pairs, three products per term) against the
    fp32 first layer of the same bf16 restatement: the stored bf16 conv1_1 maps agree except where a 2^-17 difference flips
    a rounding -- a few elements per thousand, by one bf16 ulp --, and the network outputs stay as close to fp32."""
    from dodt_amd import synth
    from oracle import extractors as oext
    from oracle import tfops
    rng = np.random.default_rng(5)
    x = rng.normal(0, 60, size=(24, 40, 3)).astype(np.float32)
    p = synth.pyramid_params(3, 142)
    ca, cb = {}, {}
    fa = oext.vgg_pyramid(x, p, collect=ca, conv_dtype='bf16')
    fb = oext.vgg_pyramid(x, p, collect=cb, conv_dtype='bf16', first_layer='split')
    a, b = ca['conv1_1'], cb['conv1_1']
    assert np.array_equal(b, tfops.round_bf16(b))
    differ = a != b
    assert differ.mean() < 0.02
    # one ulp of an 8-bit mantissa (or, next to the ReLU's zero, the 2^-17 difference itself)
    assert np.all(np.abs(a - b)[differ] <= np.maximum(np.abs(a), np.abs(b))[differ] * 2.0 ** -7 + 1e-4 * np.abs(a).max())
    f32 = oext.vgg_pyramid(x, p)
    scale = np.abs(f32).max()
    assert np.abs(fb - f32).max() / scale < 3e-2 and np.abs(fa - fb).max() / scale < 3e-2
    # fp32 path: the option changes nothing
    assert np.array_equal(oext.vgg_pyramid(x, p, first_layer='split'), f32)
