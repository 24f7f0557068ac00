"""Oracle for the dense heads.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).

TensorFlow itself is not importable here and the reference holds no golden vectors for these
layers, so the restatement below is pinned only by the definitions of the layers it replaces
(slim.conv2d / slim.fully_connected = x.w + b, ReLU by default): parity of the heads is
"unpinned" against the reference binary, exact against this restatement's arithmetic up to the
fp32 summation order (float64 accumulation here, tolerance 1e-4 * scale in the tests).
"""
import numpy as np


def fc(x, w, b, relu=True, dtype='f32'):
    """slim.fully_connected / 1x1 slim.conv2d: x (M,K) float32, w (K,N), b (N,).
    dtype 'bf16' restates the device's DODT_FC_BF16 scheme: x and w rounded to bf16 (nearest
    even), exact products, wide accumulation, fp32 bias / activation / output."""
    if dtype == 'bf16':
        from oracle import tfops
        x, w = tfops.round_bf16(x), tfops.round_bf16(w)
    y = (x.astype(np.float64) @ w.astype(np.float64) + b.astype(np.float64)).astype(np.float32)
    return np.maximum(y, np.float32(0)) if relu else y


def mean_fusion(a, b):
    """avod_fc_layer_utils.py:38-41 / dt_rpn_model.py:434-439 with both path-drop masks 1:
    (a + b) / 2 in float32."""
    return ((a.astype(np.float32) + b.astype(np.float32)) / np.float32(2.0)).astype(np.float32)


def rpn_anchor_predictor(bev_roi, img_roi, p, dtype='f32'):
    """dt_rpn_model.py:445-537.  bev_roi, img_roi (A,3,3,1); p[name] = dict(w, b) with the TF
    shapes: *_fc6 (3,3,1,256) VALID conv = FC over the flattened 3x3 crop, *_fc7
    (1,1,256,256), cls_fc8 (1,1,256,2), reg_fc8 (1,1,256,6).  -> objectness (A,2),
    offsets (A,6)."""
    x = mean_fusion(bev_roi, img_roi).reshape(len(bev_roi), -1)
    out = []
    for br in ('cls', 'reg'):
        h = x
        for li, relu in (('fc6', True), ('fc7', True), ('fc8', False)):
            q = p['%s_%s' % (br, li)]
            h = fc(h, q['w'].reshape(-1, q['w'].shape[-1]), q['b'], relu, dtype)
        out.append(h)
    return out[0], out[1]


def fusion_fc_early(bev_rois, img_rois, p, dtype='f32'):
    """fusion_fc_layers.py:136-180 (early fusion, 'mean') + build_output_layers (:94-133).
    rois (P,7,7,32) -> cls logits (P,2), offsets (P,10) and, when the parameters hold an
    `ang_out` layer (box_4ca / box_3d: avod_fc_layer_utils.py:11-17), angle vectors (P,2)
    (else None)."""
    h = mean_fusion(bev_rois, img_rois).reshape(len(bev_rois), -1)
    for name in ('fc6', 'fc7', 'fc8'):
        h = fc(h, p[name]['w'], p[name]['b'], True, dtype)
    ang = fc(h, p['ang_out']['w'], p['ang_out']['b'], False, dtype) if 'ang_out' in p else None
    return (fc(h, p['cls_out']['w'], p['cls_out']['b'], False, dtype),
            fc(h, p['off_out']['w'], p['off_out']['b'], False, dtype), ang)


def corr_fc_early(corr_rois, p, dtype='f32'):
    """avod_corr_layers_builder.py:126-169: flatten the (P,7,7,25) correlation crops, the
    same fc6..fc8 stack, off_out -> (P,3) [dx, dz, dry]."""
    h = corr_rois.reshape(len(corr_rois), -1).astype(np.float32)
    for name in ('fc6', 'fc7', 'fc8'):
        h = fc(h, p[name]['w'], p[name]['b'], True, dtype)
    return fc(h, p['off_out']['w'], p['off_out']['b'], False, dtype)
