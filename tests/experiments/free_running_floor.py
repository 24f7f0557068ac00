"""CPU experiment behind DESIGN.md section 2's free-running table: how far do two *legal* fp32
evaluations of the same frame pair drift apart end to end?

The whole pair (extractors, heads, both NMS, records) is run by the oracle with the 3x3 stride-1
convolutions evaluated in several arithmetics:
  f64    exact reference (float64 sums, one rounding per layer)            -- "truth"
  f32    the oracle's own order (one sgemm per tap, taps summed in fp32)
  f32b   another direct fp32 order (one im2col sgemm, K = 9 Cin): what a different BLAS / Eigen does
  w23    emulated Winograd F(2x2,3x3), points {0, +-1, inf}, fp32
  w43    emulated Winograd F(4x4,3x3), points {0, +-1, +-2, inf}, fp32 (round 2's kernel)
  w43p   the same with the points given on the command line (default {0, +-2/3, +-3/2, inf})
and the detections of every variant are matched against f64 and f32 with the metric of
tests/test_gpu_heads.py.  Test infrastructure (imports oracle/); not collected by pytest.

    python tests/experiments/free_running_floor.py [a b] > profiles/r3_free_running_floor.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dodt_amd import config, synth            # noqa: E402
from oracle import extractors as oext         # noqa: E402
from oracle import pipeline as opipe          # noqa: E402
from oracle import tfops                      # noqa: E402
import wino_points as wp                      # noqa: E402

C = config.PYRAMID_DODT
ORIG_CONV = tfops.conv2d_same


def conv_f64(x, w):
    return wp.direct64(x, w).astype(np.float32)


def conv_f32b(x, w):
    H, W, Cin = x.shape
    xp = np.zeros((H + 2, W + 2, Cin), np.float32)
    xp[1:-1, 1:-1] = x
    cols = np.concatenate([xp[ky:ky + H, kx:kx + W].reshape(-1, Cin) for ky in range(3) for kx in range(3)], 1)
    return (cols @ w.reshape(9 * Cin, -1)).reshape(H, W, -1)


def make_wino(points, m):
    AT, G, BT = wp.matrices(points, m=m)

    def conv(x, w):
        H, W, Cin = x.shape
        if Cin < 8:                               # first layers stay on the direct kernel
            return ORIG_CONV(x, w)
        Hp, Wp = -(-H // m) * m, -(-W // m) * m
        xx = np.zeros((Hp, Wp, Cin), np.float32)
        xx[:H, :W] = x
        # in row bands: the transformed tiles of a whole 704 x 800 x 64 map do not fit comfortably
        out = np.empty((Hp, Wp, w.shape[3]), np.float32)
        band = 32 * m
        for y0 in range(0, Hp, band):
            y1 = min(Hp, y0 + band)
            seg = np.zeros((y1 - y0 + 2, Wp, Cin), np.float32)
            lo, hi = max(y0 - 1, 0), min(y1 + 1, Hp)
            seg[lo - (y0 - 1):hi - (y0 - 1)] = xx[lo:hi]
            out[y0:y1] = wino_rows(seg, w, AT, G, BT, m)
        return out[:H, :W]
    return conv


def wino_rows(seg, w, AT, G, BT, m):
    """seg: rows y0-1 .. y1 (already with its vertical halo), width unpadded."""
    Hs, W, Cin = seg.shape
    H = Hs - 2
    n = m + 2
    U = np.einsum('ik,klcd,jl->ijcd', G, w.astype(np.float64), G).astype(np.float32)
    xp = np.zeros((Hs, W + 2, Cin), np.float32)
    xp[:, 1:-1] = seg
    th, tw = H // m, W // m
    iy = (np.arange(th) * m)[:, None] + np.arange(n)[None]
    ix = (np.arange(tw) * m)[:, None] + np.arange(n)[None]
    d = xp[iy[:, None, :, None], ix[None, :, None, :]]
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    V = np.einsum('ik,abklc->abilc', BT32, d).astype(np.float32)
    V = np.einsum('abilc,jl->abijc', V, BT32).astype(np.float32)
    M = np.empty((th, tw, n, n, w.shape[3]), np.float32)
    for i in range(n):
        for j in range(n):
            M[:, :, i, j] = (np.ascontiguousarray(V[:, :, i, j]).reshape(-1, Cin) @ U[i, j]).reshape(th, tw, -1)
    Y = np.einsum('ik,abklc->abilc', AT32, M).astype(np.float32)
    Y = np.einsum('abilc,jl->abijc', Y, AT32).astype(np.float32)
    return Y.transpose(0, 2, 1, 3, 4).reshape(H, W, -1)


def extract_with(conv, bev, img_u8, bev_params, img_params):
    """oracle.pipeline.extract with the 3x3 stride-1 convolutions replaced."""
    tfops.conv2d_same = conv
    try:
        return opipe.extract(bev, img_u8, bev_params, img_params, C['img_dims'])
    finally:
        tfops.conv2d_same = ORIG_CONV


def matched(got, ref, tol):
    used = np.zeros(len(got), bool)
    hits = 0
    for r in ref:
        d = (np.abs(got - r) / (1.0 + 0.1 * np.abs(r))).max(axis=1)
        d[used] = np.inf
        j = int(np.argmin(d)) if len(d) else -1
        if j >= 0 and d[j] <= tol:
            used[j] = True
            hits += 1
    return hits / max(len(ref), 1)


def main():
    pts = [float(v) for v in sys.argv[1:3]] if len(sys.argv) >= 3 else [2.0 / 3.0, 1.5]
    variants = [('f64', conv_f64), ('f32', ORIG_CONV), ('f32b', conv_f32b),
                ('w23', make_wino([0, 1, -1], 2)), ('w43', make_wino([0, 1, -1, 2, -2], 4)),
                ('w43p', make_wino([0, pts[0], -pts[0], pts[1], -pts[1]], 4))]
    hp = synth.head_params()
    w = synth.pipeline_weights(C)
    frames = (0, 2)
    ptsc = [synth.lidar_frame(4, f) for f in frames]
    imgs = [synth.image_frame(4, f) for f in frames]
    inps = [opipe.frame_inputs(p, C, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2, synth.IMAGE_WH) for p in ptsc]
    res, feats_all = {}, {}
    for name, conv in variants:
        t0 = time.time()
        feats = [extract_with(conv, inps[k]['bev'], imgs[k], w['bev_params'], w['img_params']) for k in range(2)]
        res[name] = opipe.pair_detections_computed(inps, feats, hp, C, synth.P2, synth.IMAGE_WH, 1024)
        feats_all[name] = [f[0] for f in feats]
        print('# %s: %.0f s' % (name, time.time() - t0), file=sys.stderr)
    print('points of w43p: {0, +-%.6g, +-%.6g, inf}' % tuple(pts))
    print('feature-map error against f64 (max / rms, of the map\'s scale), BEV maps of both frames:')
    for name, _ in variants[1:]:
        e = [np.abs(feats_all[name][k] - feats_all['f64'][k]) / np.abs(feats_all['f64'][k]).max() for k in range(2)]
        print('  %-5s max %.2e  rms %.2e' % (name, max(v.max() for v in e), np.sqrt(np.mean([np.mean(v ** 2) for v in e]))))
    for base in ('f64', 'f32'):
        print('agreement with %s: fraction of its proposals within 1e-3, of its detections within 1e-4 / 1e-3 / 1e-2 / 5e-2 '
              '(frame 0 | frame 1)' % base)
        for name, _ in variants:
            if name == base:
                continue
            cells = []
            for f in range(2):
                a, b = res[name][f], res[base][f]
                n_a, n_b = len(a['det_idx']), len(b['det_idx'])
                top = matched(a['top_anchors'], b['top_anchors'], 1e-3)
                fr = [matched(a['records'][:n_a, :8], b['records'][:n_b, :8], t) for t in (1e-4, 1e-3, 1e-2, 5e-2)]
                cells.append('prop %.3f det %d/%d %.2f %.2f %.2f %.2f' % ((top, n_a, n_b) + tuple(fr)))
            print('  %-5s %s | %s' % (name, cells[0], cells[1]))


if __name__ == '__main__':
    main()
