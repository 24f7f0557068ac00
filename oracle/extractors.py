"""Oracle for the VGG-style feature extractors, SURVEY.md 8(a) rows a8, a9, a8', a9', a10.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (TF ops).

Topology restated from /root/reference/avod/core/feature_extractors:
  bev_vgg_pyramid.py:57-169  pad 4 rows on top, 2-2-3-3 encoder with three 2x2
                             pools, three (transposed conv, concat(skip, up),
                             3x3 fusion conv) decoder stages, slice 4 rows off
  img_vgg_pyramid.py:58-171  same without pad/slice
  bev_vgg.py:34-118, img_vgg.py:33-120  encoder + 4x bilinear upsample
  models/dt_rpn_model.py:298-322        1x1 bottleneck conv + BN + ReLU

``params`` maps layer name -> dict(w=..., beta=..., mean=..., var=...), names as
in the TF variable scopes (conv1_1 ... conv4_3, upconv3, pyramid_fusion3, ...).
"""
import numpy as np

from oracle import tfops

ENCODER = [('conv1', 2), ('conv2', 2), ('conv3', 3), ('conv4', 3)]


def pyramid_layer_names():
    names = []
    for block, reps in ENCODER:
        names += ['%s_%d' % (block, i + 1) for i in range(reps)]
    names += ['upconv3', 'pyramid_fusion3', 'upconv2', 'pyramid_fusion2',
              'upconv1', 'pyramid_fusion1']
    return names


def _cbr(x, p, bf16=False, first=False, last=False, first_layer='fp32'):
    """conv + BN + ReLU.  bf16: the device's bf16 conv path (include/dodt_hip.h
    DODT_EXTRACTOR_BF16): weights rounded to bf16 (the input already is, except for the
    first layer, whose arithmetic stays fp32), fp32 accumulation and BN/ReLU, output rounded
    to bf16 unless it is the network output.
    first_layer 'split' (with bf16): the first layer as the device computes it when it folds
    conv1_1 into conv1_2's launch (dodt_extractor_first_layers_folded): x and w as hi + lo bf16
    pairs (hi = bf16(v), lo = bf16(v - hi)), w_hi x_hi + w_lo x_hi + w_hi x_lo summed in fp32."""
    if bf16 and first and first_layer == 'split':
        xh = tfops.round_bf16(x)
        xl = tfops.round_bf16(x - xh)
        wh = tfops.round_bf16(p['w'])
        wl = tfops.round_bf16(p['w'] - wh)
        conv = tfops.conv2d_same(xh, wh) + tfops.conv2d_same(xh, wl) + tfops.conv2d_same(xl, wh)
        return tfops.round_bf16(tfops.bn_relu(conv, p['beta'], p['mean'], p['var']))
    w = p['w'] if (not bf16 or first) else tfops.round_bf16(p['w'])
    y = tfops.bn_relu(tfops.conv2d_same(x, w), p['beta'], p['mean'], p['var'])
    return tfops.round_bf16(y) if (bf16 and not last) else y


def _ubr(x, p, bf16=False):
    w = tfops.round_bf16(p['w']) if bf16 else p['w']
    y = tfops.bn_relu(tfops.conv2d_transpose_s2_same(x, w), p['beta'], p['mean'], p['var'])
    return tfops.round_bf16(y) if bf16 else y


def encoder(x, params, collect=None, bf16=False, first_layer='fp32'):
    """Returns [conv1, conv2, conv3, conv4] block outputs (pre-pool)."""
    outs = []
    for bi, (block, reps) in enumerate(ENCODER):
        if bi > 0:
            x = tfops.max_pool_2x2(x)
        for r in range(reps):
            name = '%s_%d' % (block, r + 1)
            x = _cbr(x, params[name], bf16, first=(name == 'conv1_1'), first_layer=first_layer)
            if collect is not None:
                collect[name] = x
        outs.append(x)
    return outs


def vgg_pyramid(x, params, pad_top=0, collect=None, conv_dtype='f32', first_layer='fp32'):
    """x (H,W,C) float32 -> (H,W,32) full-resolution pyramid feature map.
    conv_dtype 'bf16' restates the device's bf16 conv path, first_layer its first layer's
    arithmetic (see _cbr)."""
    bf16 = conv_dtype == 'bf16'
    x = np.asarray(x, dtype=np.float32)
    if pad_top:
        x = np.concatenate(
            [np.zeros((pad_top,) + x.shape[1:], dtype=np.float32), x], axis=0)
    c1, c2, c3, c4 = encoder(x, params, collect, bf16, first_layer)
    up3 = _ubr(c4, params['upconv3'], bf16)
    f3 = _cbr(np.concatenate([c3, up3], axis=2), params['pyramid_fusion3'], bf16)
    up2 = _ubr(f3, params['upconv2'], bf16)
    f2 = _cbr(np.concatenate([c2, up2], axis=2), params['pyramid_fusion2'], bf16)
    up1 = _ubr(f2, params['upconv1'], bf16)
    f1 = _cbr(np.concatenate([c1, up1], axis=2), params['pyramid_fusion1'], bf16, last=True)
    if collect is not None:
        collect.update(upconv3=up3, pyramid_fusion3=f3, upconv2=up2,
                       pyramid_fusion2=f2, upconv1=up1, pyramid_fusion1=f1)
    return f1[pad_top:]


def vgg_plain(x, params, upsample=4, collect=None):
    """Config-1 extractors (bev_vgg.py:34-118 / img_vgg.py:33-120): encoder, then
    tf.image.resize_bilinear of conv4_3 to input_pixel_size / 8 * upsampling_multiplier -- a
    float pair (bev_vgg.py:102-109) that TF casts to int32: (350, 400) for the 700 x 800 BEV
    input although conv4 is 87 x 100, (240, 795) for the 480 x 1590 image."""
    x = np.asarray(x, dtype=np.float32)
    c4 = encoder(x, params, collect)[3]
    out_h = int(x.shape[0] / 8 * upsample)
    out_w = int(x.shape[1] / 8 * upsample)
    return tfops.resize_bilinear(c4, out_h, out_w)


def bottleneck_1x1(x, p):
    """slim.conv2d(x, 1, [1,1]) + BN + ReLU -> (H,W,1)."""
    w = np.asarray(p['w'], dtype=np.float32).reshape(-1, 1)
    y = (np.asarray(x, dtype=np.float32).reshape(-1, w.shape[0]) @ w)
    y = y.reshape(x.shape[0], x.shape[1], 1)
    return tfops.bn_relu(y, p['beta'], p['mean'], p['var'])
