"""Oracle for the TensorFlow-1.3 ops the hot path calls, SURVEY.md 8(a) rows a7-a11, a13.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: TF 1.3 is
not available and the reference's tests assert no numeric output of these ops.
The semantics restated here are TF-1.3's:

* slim.conv2d(padding='SAME', normalizer_fn=slim.batch_norm) = SAME conv, no
  bias, batch_norm(center=True, scale=False, epsilon=1e-3) in inference form
  x * rsqrt(var + eps) + (beta - mean * rsqrt(var + eps)), then ReLU.
* slim.max_pool2d([2,2]) = stride 2, VALID.
* slim.conv2d_transpose(3x3, stride 2, SAME): out = 2 * in,
  out[2i + k] += in[i] * w[k]  (rows >= 2*in dropped), weights [kh,kw,Cout,Cin].
* tf.image.resize_images / resize_bilinear, align_corners=False (legacy).
* tf.image.crop_and_resize (bilinear, extrapolation 0).
* tf.image.non_max_suppression (greedy, iou > thr suppresses).

Call sites in the reference (relative to /root/reference/avod/core):
  feature_extractors/bev_vgg_pyramid.py:30-178, img_vgg_pyramid.py:30-177,
  feature_extractors/img_feature_extractor.py:16-35,
  models/dt_rpn_model.py:298-322 (1x1 bottleneck), :418-428 (RPN crop),
  :587-591 (NMS #1); models/dt_avod_model.py:253-273 (crop), :606-613 (NMS #2).
All arrays are float32, NHWC with the batch dimension dropped (H, W, C).
"""
import numpy as np

F32 = np.float32
BN_EPS = F32(0.001)


def round_bf16(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32: what the device's
    v_cvt_pk_bf16_f32 stores between the layers of the bf16 conv path."""
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32)
    r = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return r.view(F32)


# Arithmetic of the conv / transposed-conv sums.  F32: the reference's (float32 throughout, one
# sgemm per tap).  np.float64 ("exact" mode, set through `exact_sums`): every sum in float64 and ONE
# rounding per layer -- not the reference's arithmetic but the value every legal float32 evaluation
# order scatters around; tests use it to measure how far such orders drift apart end to end.
ACC = F32


class exact_sums(object):
    """with tfops.exact_sums(): conv sums in float64, rounded once per layer."""

    def __enter__(self):
        global ACC
        self._old, ACC = ACC, np.float64

    def __exit__(self, *exc):
        global ACC
        ACC = self._old


def conv2d_same(x, w):
    """x (H,W,Cin), w (kh,kw,Cin,Cout) -> (H,W,Cout); stride 1, SAME.
    Accumulates tap by tap in float32 (one GEMM per tap)."""
    x = np.asarray(x, dtype=F32).astype(ACC)
    w = np.asarray(w, dtype=F32).astype(ACC)
    kh, kw, cin, cout = w.shape
    h, wd, _ = x.shape
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = np.zeros((h + kh - 1, wd + kw - 1, cin), dtype=ACC)
    xp[pt:pt + h, pl:pl + wd] = x
    out = np.zeros((h * wd, cout), dtype=ACC)
    for ky in range(kh):
        for kx in range(kw):
            patch = np.ascontiguousarray(xp[ky:ky + h, kx:kx + wd]).reshape(-1, cin)
            out += patch @ w[ky, kx]
    return out.reshape(h, wd, cout).astype(F32)


def conv2d_transpose_s2_same(x, w):
    """x (H,W,Cin), w (3,3,Cout,Cin) -> (2H,2W,Cout); stride 2, SAME.
    Gradient-of-conv form: out[2i+ky, 2j+kx] += x[i,j] @ w[ky,kx].T."""
    x = np.asarray(x, dtype=F32).astype(ACC)
    w = np.asarray(w, dtype=F32).astype(ACC)
    kh, kw, cout, cin = w.shape
    h, wd, _ = x.shape
    big = np.zeros((2 * h + 2, 2 * wd + 2, cout), dtype=ACC)
    flat = x.reshape(-1, cin)
    for ky in range(kh):
        for kx in range(kw):
            contrib = (flat @ w[ky, kx].T).reshape(h, wd, cout)
            big[ky:ky + 2 * h:2, kx:kx + 2 * wd:2] += contrib
    return big[:2 * h, :2 * wd].astype(F32)


def bn_scale_shift(beta, mean, var):
    """Inference batch-norm folded to (scale, shift): y = x*scale + shift."""
    inv = F32(1.0) / np.sqrt(np.asarray(var, dtype=F32) + BN_EPS)
    inv = inv.astype(F32)
    shift = (np.asarray(beta, dtype=F32) - np.asarray(mean, dtype=F32) * inv)
    return inv, shift.astype(F32)


def bn_relu(x, beta, mean, var):
    s, b = bn_scale_shift(beta, mean, var)
    return np.maximum(x * s + b, F32(0)).astype(F32)


def max_pool_2x2(x):
    """VALID 2x2 stride 2 (odd trailing row/col dropped)."""
    h, w, c = x.shape
    x = x[:h // 2 * 2, :w // 2 * 2]
    return x.reshape(h // 2, 2, w // 2, 2, c).max(axis=(1, 3))


def resize_bilinear(x, out_h, out_w):
    """Legacy (align_corners=False, no half-pixel) bilinear resize."""
    x = np.asarray(x, dtype=F32)
    h, w, _ = x.shape
    sy = F32(h) / F32(out_h)
    sx = F32(w) / F32(out_w)
    iy = (np.arange(out_h, dtype=F32) * sy).astype(F32)
    ix = (np.arange(out_w, dtype=F32) * sx).astype(F32)
    y0 = np.floor(iy).astype(np.int64)
    x0 = np.floor(ix).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    ly = (iy - y0.astype(F32))[:, None, None]
    lx = (ix - x0.astype(F32))[None, :, None]
    tl = x[y0][:, x0]
    tr = x[y0][:, x1]
    bl = x[y1][:, x0]
    br = x[y1][:, x1]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    return (top + (bot - top) * ly).astype(F32)


KITTI_RGB_MEAN = np.array([92.8403, 97.7996, 93.5843], dtype=F32)


def img_preprocess(img_u8, out_h, out_w):
    """img_feature_extractor.py:16-35: resize then per-channel mean subtract."""
    x = resize_bilinear(np.asarray(img_u8).astype(F32), out_h, out_w)
    return (x - KITTI_RGB_MEAN).astype(F32)


def crop_and_resize(image, boxes, crop_h, crop_w):
    """image (H,W,C), boxes (n,4) [y1,x1,y2,x2] normalised -> (n,ch,cw,C)."""
    img = np.asarray(image, dtype=F32)
    b = np.asarray(boxes, dtype=F32)
    h, w, c = img.shape
    n = b.shape[0]
    out = np.zeros((n, crop_h, crop_w, c), dtype=F32)
    hm1, wm1 = F32(h - 1), F32(w - 1)
    y1, x1, y2, x2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    if crop_h > 1:
        hs = (y2 - y1) * hm1 / F32(crop_h - 1)
    else:
        hs = np.zeros(n, dtype=F32)
    if crop_w > 1:
        ws = (x2 - x1) * wm1 / F32(crop_w - 1)
    else:
        ws = np.zeros(n, dtype=F32)
    for iy in range(crop_h):
        if crop_h > 1:
            in_y = y1 * hm1 + F32(iy) * hs
        else:
            in_y = F32(0.5) * (y1 + y2) * hm1
        in_y = in_y.astype(F32)
        oky = (in_y >= 0) & (in_y <= hm1)
        ty = np.floor(in_y)
        by = np.ceil(in_y)
        ly = (in_y - ty).astype(F32)
        tyi = np.clip(ty, 0, h - 1).astype(np.int64)
        byi = np.clip(by, 0, h - 1).astype(np.int64)
        for ix in range(crop_w):
            if crop_w > 1:
                in_x = x1 * wm1 + F32(ix) * ws
            else:
                in_x = F32(0.5) * (x1 + x2) * wm1
            in_x = in_x.astype(F32)
            ok = oky & (in_x >= 0) & (in_x <= wm1)
            lx_f = np.floor(in_x)
            rx_f = np.ceil(in_x)
            lx = (in_x - lx_f).astype(F32)[:, None]
            lxi = np.clip(lx_f, 0, w - 1).astype(np.int64)
            rxi = np.clip(rx_f, 0, w - 1).astype(np.int64)
            tl = img[tyi, lxi]
            tr = img[tyi, rxi]
            bl = img[byi, lxi]
            br = img[byi, rxi]
            top = tl + (tr - tl) * lx
            bot = bl + (br - bl) * lx
            val = top + (bot - top) * ly[:, None]
            out[:, iy, ix] = np.where(ok[:, None], val, F32(0))
    return out


def softmax2(logits):
    """tf.nn.softmax over the last axis (float32, max-subtracted)."""
    x = np.asarray(logits, dtype=F32)
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return (e / e.sum(axis=-1, keepdims=True)).astype(F32)


def nms_order(scores):
    """Candidate order: score descending, ties by ascending index.  TF-1.3 uses
    an unstable std::sort (tie order unspecified); the build fixes this one."""
    s = np.asarray(scores, dtype=F32)
    return np.lexsort((np.arange(len(s)), -s.astype(np.float64)))


def _iou_gt(bi, bj, thr):
    """TF-1.3 non_max_suppression_op.cc IOUGreaterThanThreshold, float32."""
    ymin_i, xmin_i = min(bi[0], bi[2]), min(bi[1], bi[3])
    ymax_i, xmax_i = max(bi[0], bi[2]), max(bi[1], bi[3])
    ymin_j, xmin_j = min(bj[0], bj[2]), min(bj[1], bj[3])
    ymax_j, xmax_j = max(bj[0], bj[2]), max(bj[1], bj[3])
    area_i = F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i)
    area_j = F32(ymax_j - ymin_j) * F32(xmax_j - xmin_j)
    if area_i <= 0 or area_j <= 0:
        return False
    iy = max(F32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), F32(0))
    ix = max(F32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), F32(0))
    inter = F32(iy * ix)
    iou = F32(inter / F32(F32(area_i + area_j) - inter))
    return iou > thr


def non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """Greedy NMS; returns selected original indices (int32) in score order."""
    b = np.asarray(boxes, dtype=F32)
    thr = F32(iou_threshold)
    order = nms_order(scores)
    keep = []
    for i in order:
        if len(keep) >= max_output_size:
            break
        ok = True
        for j in reversed(keep):
            if _iou_gt(b[i], b[j], thr):
                ok = False
                break
        if ok:
            keep.append(int(i))
    return np.asarray(keep, dtype=np.int32)


def non_max_suppression_fast(boxes, scores, max_output_size, iou_threshold):
    """Vectorised twin of non_max_suppression (same float32 arithmetic), for
    test sizes where the pure-python loop is too slow."""
    b = np.asarray(boxes, dtype=F32)
    thr = F32(iou_threshold)
    order = nms_order(scores)
    ymin = np.minimum(b[:, 0], b[:, 2])
    ymax = np.maximum(b[:, 0], b[:, 2])
    xmin = np.minimum(b[:, 1], b[:, 3])
    xmax = np.maximum(b[:, 1], b[:, 3])
    area = ((ymax - ymin).astype(F32) * (xmax - xmin).astype(F32)).astype(F32)
    alive = np.ones(len(b), dtype=bool)
    keep = []
    for pos, i in enumerate(order):
        if len(keep) >= max_output_size:
            break
        if not alive[i]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        rest = rest[alive[rest]]
        if rest.size == 0 or area[i] <= 0:
            continue
        iy = np.maximum((np.minimum(ymax[i], ymax[rest]) -
                         np.maximum(ymin[i], ymin[rest])).astype(F32), F32(0))
        ix = np.maximum((np.minimum(xmax[i], xmax[rest]) -
                         np.maximum(xmin[i], xmin[rest])).astype(F32), F32(0))
        inter = (iy * ix).astype(F32)
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = (inter / ((area[i] + area[rest]).astype(F32) - inter)
                   ).astype(F32)
        sup = (iou > thr) & (area[rest] > 0)
        alive[rest[sup]] = False
    return np.asarray(keep, dtype=np.int32)


def correlation(a, b, max_displacement=5, stride_2=2, pad=5):
    """The reference's Correlation op with kernel_size 1, stride_1 1
    (avod/core/ops/correlation/correlation_kernel.cu.cc:21-119, pad.cu.cc:14-73):
    out[y,x,k] = 1/C sum_c Apad[y+d, x+d, c] * Bpad[y+d+s2p, x+d+s2o, c], k over a
    (2r+1)^2 grid with r = d // stride_2, channel sum sequential in float32 (the CUDA
    kernel's lane-0 loop over 32 per-lane products).  PARITY UNPINNED: the op is
    GPU-only and its tests assert nothing."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    h, w, c = a.shape
    d = max_displacement
    r = d // stride_2
    gw = 2 * r + 1
    oh, ow = h + 2 * pad - 2 * d, w + 2 * pad - 2 * d
    ap = np.zeros((h + 2 * pad, w + 2 * pad, c), dtype=F32)
    bp = np.zeros_like(ap)
    ap[pad:pad + h, pad:pad + w] = a
    bp[pad:pad + h, pad:pad + w] = b
    out = np.zeros((oh, ow, gw * gw), dtype=F32)
    ac = ap[d:d + oh, d:d + ow]
    for k in range(gw * gw):
        s2p, s2o = (k // gw - r) * stride_2, (k % gw - r) * stride_2
        bc = bp[d + s2p:d + s2p + oh, d + s2o:d + s2o + ow]
        acc = np.zeros((oh, ow), dtype=F32)
        for ch in range(c):
            acc = (acc + (ac[:, :, ch] * bc[:, :, ch]).astype(F32)).astype(F32)
        out[:, :, k] = acc / F32(c)
    return out
