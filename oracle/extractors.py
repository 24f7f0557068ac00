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


def _cbr(x, p):
    return tfops.bn_relu(tfops.conv2d_same(x, p['w']),
                         p['beta'], p['mean'], p['var'])


def _ubr(x, p):
    return tfops.bn_relu(tfops.conv2d_transpose_s2_same(x, p['w']),
                         p['beta'], p['mean'], p['var'])


def encoder(x, params, collect=None):
    """Returns [conv1, conv2, conv3, conv4] block outputs (pre-pool)."""
    outs = []
    for bi, (block, reps) in enumerate(ENCODER):
        if bi > 0:
            x = tfops.max_pool_2x2(x)
        for r in range(reps):
            name = '%s_%d' % (block, r + 1)
            x = _cbr(x, params[name])
            if collect is not None:
                collect[name] = x
        outs.append(x)
    return outs


def vgg_pyramid(x, params, pad_top=0, collect=None):
    """x (H,W,C) float32 -> (H,W,32) full-resolution pyramid feature map."""
    x = np.asarray(x, dtype=np.float32)
    if pad_top:
        x = np.concatenate(
            [np.zeros((pad_top,) + x.shape[1:], dtype=np.float32), x], axis=0)
    c1, c2, c3, c4 = encoder(x, params, collect)
    up3 = _ubr(c4, params['upconv3'])
    f3 = _cbr(np.concatenate([c3, up3], axis=2), params['pyramid_fusion3'])
    up2 = _ubr(f3, params['upconv2'])
    f2 = _cbr(np.concatenate([c2, up2], axis=2), params['pyramid_fusion2'])
    up1 = _ubr(f2, params['upconv1'])
    f1 = _cbr(np.concatenate([c1, up1], axis=2), params['pyramid_fusion1'])
    if collect is not None:
        collect.update(upconv3=up3, pyramid_fusion3=f3, upconv2=up2,
                       pyramid_fusion2=f2, upconv1=up1, pyramid_fusion1=f1)
    return f1[pad_top:]


def vgg_plain(x, params, upsample=4):
    """Config-1 extractors (bev_vgg.py / img_vgg.py): encoder, then bilinear
    resize of conv4 to (H/8*4, W/8*4)."""
    x = np.asarray(x, dtype=np.float32)
    c4 = encoder(x, params)[3]
    return tfops.resize_bilinear(c4, c4.shape[0] * upsample,
                                 c4.shape[1] * upsample)


def bottleneck_1x1(x, p):
    """slim.conv2d(x, 1, [1,1]) + BN + ReLU -> (H,W,1)."""
    w = np.asarray(p['w'], dtype=np.float32).reshape(-1, 1)
    y = (np.asarray(x, dtype=np.float32).reshape(-1, w.shape[0]) @ w)
    y = y.reshape(x.shape[0], x.shape[1], 1)
    return tfops.bn_relu(y, p['beta'], p['mean'], p['var'])
