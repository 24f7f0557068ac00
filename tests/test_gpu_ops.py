"""HIP ROI crop, NMS and box encoders against the oracle.  Needs an MI355X."""
import os

import numpy as np
import pytest

from dodt_amd import config as cfg
from dodt_amd import device, ops, tf_image
from dodt_amd.core import anchor_encoder as gpu_anchor_encoder
from dodt_amd.core import box_4c_encoder as gpu_box_4c
from oracle import boxes as oboxes
from oracle import tfops

pytestmark = pytest.mark.gpu
C = cfg.PYRAMID_DODT


@pytest.fixture(scope='module')
def ctx():
    return device.default_context()


def _boxes(rng, n, spread=1.0):
    """normalised [y1,x1,y2,x2]; a third partly or fully outside the image"""
    cy, cx = rng.uniform(-0.3, 1.3, n), rng.uniform(-0.3, 1.3, n)
    h, w = rng.uniform(0.0, 0.3, n) * spread, rng.uniform(0.0, 0.3, n) * spread
    return np.stack([cy - h, cx - w, cy + h, cx + w], 1).astype(np.float32)


@pytest.mark.parametrize('hwc,crop', [((44, 50, 32), (7, 7)), ((88, 100, 1), (3, 3)),
                                      ((30, 40, 25), (7, 7)), ((16, 16, 4), (1, 1)),
                                      ((9, 11, 3), (2, 5))])
def test_crop_and_resize_matches_oracle(hwc, crop):
    rng = np.random.default_rng(hash(hwc) & 0xffff)
    img = rng.normal(size=hwc).astype(np.float32)
    b = _boxes(rng, 300)
    b[0] = [0, 0, 1, 1]
    b[1] = [0.5, 0.5, 0.5, 0.5]            # degenerate
    b[2] = [1.0, 1.0, 0.0, 0.0]            # flipped
    b[3] = [-21.0, -3.0, 24.0, 5.0]        # far outside, like image-space anchors
    got = tf_image.crop_and_resize(img[None], b, np.zeros(len(b), np.int32), crop)
    want = tfops.crop_and_resize(img, b, crop[0], crop[1])
    assert got.shape == want.shape
    assert np.array_equal(got, want), np.abs(got - want).max()   # unfused fp32: bit exact


def test_crop_full_size_properties(ctx):
    """P = 1024 boxes on a (700,800,32) map: identity box reproduces the map's
    corners; a box outside the image gives zeros; linear in the image."""
    rng = np.random.default_rng(3)
    img = rng.normal(size=(700, 800, 32)).astype(np.float32)
    b = _boxes(rng, 1024, spread=0.2)
    b[0] = [0, 0, 1, 1]
    b[1] = [2, 2, 3, 3]
    out = tf_image.crop_and_resize(img[None], b, None, (7, 7))
    assert np.array_equal(out[0, 0, 0], img[0, 0]) and np.array_equal(out[0, 6, 6], img[699, 799])
    assert not out[1].any()
    out2 = tf_image.crop_and_resize((2 * img)[None], b, None, (7, 7))
    assert np.array_equal(out2, 2 * out)
    sub = rng.choice(1024, 64, replace=False)
    want = tfops.crop_and_resize(img, b[sub], 7, 7)
    assert np.array_equal(out[sub], want)


def test_crop_rejects_bad_input():
    with pytest.raises(ValueError):
        tf_image.crop_and_resize(np.zeros((2, 4, 4, 1), np.float32), np.zeros((1, 4)), None, (3, 3))
    with pytest.raises(ValueError):
        tf_image.crop_and_resize(np.zeros((1, 4, 4, 1), np.float32), np.zeros((1, 5)), None, (3, 3))


def _nms_case(rng, n, scale):
    cy, cx = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    h, w = rng.uniform(0.01, scale, n), rng.uniform(0.01, scale, n)
    boxes = np.stack([cy - h, cx - w, cy + h, cx + w], 1).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    return boxes, scores


@pytest.mark.parametrize('n,k,thr,scale', [
    (1, 10, 0.5, 0.1), (2, 1, 0.5, 0.1), (63, 100, 0.5, 0.2), (64, 64, 0.3, 0.2),
    (65, 1000, 0.8, 0.2), (1000, 100, 0.01, 0.05), (2048, 300, 0.8, 0.1),
    (2049, 1024, 0.8, 0.1), (4100, 1024, 0.5, 0.05), (8192, 1024, 0.8, 0.03),
    (9000, 300, 0.8, 0.03), (14000, 1024, 0.8, 0.03), (20000, 2000, 0.6, 0.02),
    # many chunks of rows on the counted-rank path; the bitonic path above 32768 candidates
    (16000, 1024, 0.3, 0.1), (40000, 700, 0.7, 0.015)])
def test_nms_matches_oracle(n, k, thr, scale):
    rng = np.random.default_rng(n * 31 + k)
    boxes, scores = _nms_case(rng, n, scale)
    got = tf_image.non_max_suppression(boxes, scores, k, thr)
    want = tfops.non_max_suppression_fast(boxes, scores, k, thr)
    assert got.dtype == np.int32
    assert np.array_equal(got, want)


def test_nms_ties_zero_area_and_flipped_boxes():
    rng = np.random.default_rng(17)
    boxes, scores = _nms_case(rng, 3000, 0.08)
    scores[100:400] = scores[100]                 # many exact ties -> index order
    scores[1000:1100] = np.float32(-0.5)          # negative scores sort last
    boxes[::41, 2] = boxes[::41, 0]               # zero area: never suppress / suppressed
    boxes[5::53] = boxes[5::53][:, [2, 3, 0, 1]]  # corners given max-first
    boxes[7] = boxes[6]                           # exact duplicate
    for thr, k in ((0.8, 1024), (0.01, 100), (0.5, 3000)):
        got = tf_image.non_max_suppression(boxes, scores, k, thr)
        want = tfops.non_max_suppression_fast(boxes, scores, k, thr)
        assert np.array_equal(got, want)


def test_nms_properties_full_size():
    """A = 14k anchors-like boxes: output is sorted by score, has no pair above
    the threshold, and running NMS on its own output is the identity."""
    rng = np.random.default_rng(23)
    boxes, scores = _nms_case(rng, 14000, 0.04)
    thr = np.float32(0.8)
    sel = tf_image.non_max_suppression(boxes, scores, 1024, thr)
    assert len(sel) == 1024 and len(set(sel.tolist())) == 1024
    assert np.all(np.diff(scores[sel].astype(np.float64)) <= 0)
    again = tf_image.non_max_suppression(boxes[sel], scores[sel], 1024, thr)
    assert np.array_equal(again, np.arange(1024))


def test_nms_rejects_bad_input():
    with pytest.raises(ValueError):
        tf_image.non_max_suppression(np.zeros((4, 3)), np.zeros(4), 2, 0.5)
    with pytest.raises(ValueError):
        tf_image.non_max_suppression(np.zeros((4, 4)), np.zeros(3), 2, 0.5)
    with pytest.raises(ValueError):
        tf_image.non_max_suppression(np.zeros((4, 4)), np.zeros(4), 2, 1.5)
    assert len(tf_image.non_max_suppression(np.zeros((0, 4)), np.zeros(0), 5, 0.5)) == 0


def test_encoders_match_golden_and_oracle(golden_dir):
    enc = np.load(os.path.join(golden_dir, 'encoders.npz'))
    anchors = enc['anchor_ortho'].astype(np.float32)
    off = enc['anchor_offsets'].astype(np.float32)
    got = gpu_anchor_encoder.offset_to_anchor(anchors, off)
    np.testing.assert_allclose(got, enc['regressed_anchors'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got, oboxes.offset_to_anchor(anchors, off, np.float32),
                               rtol=2e-6, atol=1e-6)
    # stage-2 decode chain
    rng = np.random.default_rng(5)
    top = anchors.copy()
    off4c = rng.normal(0, 0.15, size=(len(top), 10)).astype(np.float32)
    plane = C['ground_plane']
    b3, panc, bev = gpu_box_4c.decode_box_4c_predictions(top, off4c, plane, C['bev_extents'])
    o_b3 = oboxes.anchors_to_box_3d(top, fix_lw=True, dtype=np.float32)
    o_4c = oboxes.box_3d_to_box_4c(o_b3, plane, np.float32)
    o_pred = oboxes.box_4c_to_box_3d(oboxes.offsets_to_box_4c(o_4c, off4c), plane, np.float32)
    np.testing.assert_allclose(b3, o_pred, atol=1e-4)        # north_star: 1e-4 fp32
    o_anc = oboxes.box_3d_to_anchor_ortho(o_pred, np.float32)
    np.testing.assert_allclose(panc, o_anc, atol=1e-4)
    o_bev, _ = oboxes.project_to_bev(o_anc, C['bev_extents'], np.float32)
    np.testing.assert_allclose(bev, o_bev[:, [1, 0, 3, 2]], atol=1e-4)
    # the numpy-branch golden (float64) of the same chain without offsets
    b3z, _, _ = gpu_box_4c.decode_box_4c_predictions(
        top, np.zeros_like(off4c), plane, C['bev_extents'])
    want = oboxes.box_4c_to_box_3d(
        oboxes.box_3d_to_box_4c(enc['box_3d_from_anchor'], plane, np.float64), plane, np.float64)
    np.testing.assert_allclose(b3z[:, :6], want[:, :6], atol=1e-4)


def test_softmax_and_gather(ctx):
    rng = np.random.default_rng(9)
    logits = rng.normal(0, 3, size=(5000, 2)).astype(np.float32)
    d_out = ctx.empty((5000,), np.float32)
    ops.softmax_fg(ctx, ctx.array(logits), 5000, None, d_out)
    np.testing.assert_allclose(d_out.download(), tfops.softmax2(logits)[:, 1],
                               rtol=2e-6, atol=1e-7)
    src = rng.normal(size=(700, 6)).astype(np.float32)
    idx = rng.integers(0, 700, 300).astype(np.int32)
    d_g = ctx.empty((300, 6), np.float32)
    ops.gather_rows(ctx, ctx.array(src), 6, ctx.array(idx), 300, None, d_g)
    assert np.array_equal(d_g.download(), src[idx])


def test_box_4ca_records_match_reference_goldens(ctx, golden_dir):
    """dodt_pack_detections with orientations against the reference's own
    get_avod_predicted_boxes_3d_and_scores (avod/core/dt_evaluator.py:1134-1259, box_4ca;
    tests/golden/make_goldens_box4ca.py): heading correction, l/w swap, correlation shift,
    every column -- bit for bit (float32 arithmetic on both sides, thresholds included)."""
    from dodt_amd.core import orientation_encoder as gpu_orient
    g = np.load(os.path.join(golden_dir, 'box4ca.npz'))
    for c in range(int(g['n_cases'])):
        want = g['c%d_records' % c]
        row = 0
        for f in range(2):
            boxes = g['c%d_boxes_3d_%d' % (c, f)]
            ori = g['c%d_orientations_%d' % (c, f)]
            scores = g['c%d_softmax_%d' % (c, f)][:, 1]
            got = gpu_orient.predicted_boxes_3d_and_scores(
                boxes, scores, ori, g['c%d_corr_offsets' % c] if f == 0 else None, f, ctx=ctx)
            assert np.array_equal(got.astype(np.float64), want[row:row + len(boxes)]), (c, f)
            row += len(boxes)
    # box_4c (no orientations): boxes pass through unchanged
    boxes, scores = g['c0_boxes_3d_0'], g['c0_softmax_0'][:, 1]
    plain = gpu_orient.predicted_boxes_3d_and_scores(boxes, scores, None, None, 1, ctx=ctx)
    assert np.array_equal(plain[:, :7], boxes) and not plain[:, 9:16].any()
    # gathered by selection indices, padded with zero rows, count clipped to max_det
    n = len(boxes)
    sel = np.random.default_rng(3).permutation(n)[:40].astype(np.int32)
    d_rec, d_cnt = ctx.empty((50, 17), np.float32), ctx.empty((1,), np.int32)
    ops.pack_detections(ctx, ctx.array(boxes), ctx.array(scores), ctx.array(sel),
                        ctx.array(np.array([40], np.int32)), 50, 0.0, d_rec, d_cnt,
                        d_corr_offsets=ctx.array(g['c0_corr_offsets']),
                        d_orientations=ctx.array(g['c0_orientations_0']))
    rec = d_rec.download()
    assert d_cnt.download()[0] == 40 and not rec[40:].any()
    assert np.array_equal(rec[:40].astype(np.float64), g['c0_records'][sel])


def test_angle_vector_to_orientation(ctx):
    """avod/core/orientation_encoder.py:20-34 and the known answers of
    orientation_encoder_test.py:26-87 (atan2f against numpy's float32 arctan2: 1 ulp)."""
    from dodt_amd.core import orientation_encoder as gpu_orient
    rng = np.random.default_rng(12)
    v = rng.normal(size=(3000, 2)).astype(np.float32)
    v[:6] = [[1, 0], [0, 1], [-1, 0], [0, -1], [0.70710678, 0.70710678], [-1, -1e-9]]
    got = gpu_orient.tf_angle_vector_to_orientation(v, ctx=ctx)
    want = oboxes.angle_vector_to_orientation(v, np.float32)
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-7)
    np.testing.assert_allclose(got[:5], [0, np.pi / 2, np.pi, -np.pi / 2, np.pi / 4], atol=1e-6)
    assert len(gpu_orient.tf_angle_vector_to_orientation(np.zeros((0, 2)), ctx=ctx)) == 0
    with pytest.raises(ValueError):
        gpu_orient.tf_angle_vector_to_orientation(np.zeros((4, 3)), ctx=ctx)


@pytest.mark.parametrize('hwc,crop,stride', [((30, 40, 25), (7, 7), 1248), ((44, 50, 32), (7, 7), 1600),
                                             ((20, 20, 3), (2, 2), 13)])
def test_strided_crop_rows_are_the_packed_crops(ctx, hwc, crop, stride):
    """dodt_crop_and_resize_strided: box b's crop at b * stride floats, the floats between crops untouched
    (the correlation head's rows of 1248 = 7*7*25 + 23 zeros; also a stride that is no multiple of 4)."""
    rng = np.random.default_rng(stride)
    img = rng.normal(size=hwc).astype(np.float32)
    b = _boxes(rng, 200)
    n = crop[0] * crop[1] * hwc[2]
    d_out = ctx.array(np.full((200, stride), 7.5, np.float32))
    d_n = ctx.array(np.array([150], np.int32))
    ops.crop_and_resize(ctx, ctx.array(img), hwc, ctx.array(b), 200, d_n, crop, d_out, out_box_stride=stride)
    got = d_out.download()
    want = tfops.crop_and_resize(img, b[:150], crop[0], crop[1]).reshape(150, n)
    assert np.array_equal(got[:150, :n], want)
    assert np.all(got[:150, n:] == 7.5) and np.all(got[150:] == 7.5)
    with pytest.raises(ValueError):
        ops.crop_and_resize(ctx, ctx.array(img), hwc, ctx.array(b), 200, None, crop, d_out, out_box_stride=n - 1)


def test_mean_fusion_is_the_float32_mean(ctx):
    """dodt_mean_fusion: (a + b) / 2 in float32, exactly, over min(*d_n, rows) rows
    (avod_fc_layer_utils.py:38-41 with both path-drop masks 1)."""
    from oracle import heads as oheads
    rng = np.random.default_rng(77)
    a = (rng.normal(size=(333, 1568)) * 10.0 ** rng.integers(-20, 20, size=(333, 1))).astype(np.float32)
    b = rng.normal(size=(333, 1568)).astype(np.float32)
    d_out = ctx.array(np.full((333, 1568), -3.0, np.float32))
    ops.mean_fusion(ctx, ctx.array(a), ctx.array(b), 333, ctx.array(np.array([300], np.int32)), 1568, d_out)
    got = d_out.download()
    assert np.array_equal(got[:300], oheads.mean_fusion(a[:300], b[:300]))
    assert np.all(got[300:] == -3.0)
    ops.mean_fusion(ctx, ctx.array(a), ctx.array(b), 333, None, 1568, d_out)
    assert np.array_equal(d_out.download(), oheads.mean_fusion(a, b))
    with pytest.raises(ValueError):
        ops.mean_fusion(ctx, ctx.array(a), ctx.array(b), 333, None, 1567, d_out)


@pytest.mark.parametrize('n,n_valid', [(5000, None), (1024, 700), (1, None), (257, 0)])
def test_fused_tail_ops_equal_the_separate_ops_bit_for_bit(ctx, n, n_valid):
    """dodt_rpn_decode, dodt_gather_project and dodt_final_decode (round 4: one launch for each elementwise run of a frame's
    launch chain) against the ops they replace -- which the tests above hold to the oracle and the reference's goldens --
    on the same inputs: every output array equal bit for bit, rows beyond *d_n untouched."""
    rng = np.random.default_rng(n * 7 + (n_valid or 0))
    ae = np.asarray(C['area_extents'], np.float32)
    ext = [float(ae[0, 0]), float(ae[0, 1]), float(ae[2, 0]), float(ae[2, 1])]
    p2 = rng.normal(0, 1, 12).astype(np.float32) * np.asarray([700, 1, 600, 40, 1, 700, 180, 2, 0.01, 0.01, 1, 0.003], np.float32)
    plane = [0.0, -1.0, 0.0, 1.65]
    d_n = ctx.array(np.asarray([n_valid], np.int32)) if n_valid is not None else None
    anchors = np.concatenate([rng.uniform(-40, 40, (n, 1)), rng.uniform(0.5, 2, (n, 1)), rng.uniform(0, 70, (n, 1)),
                              rng.uniform(1, 5, (n, 3))], 1).astype(np.float32)
    offs = rng.normal(0, 0.3, (n, 6)).astype(np.float32)
    logits = rng.normal(0, 3, (n, 2)).astype(np.float32)
    d_a, d_o, d_l = ctx.array(anchors), ctx.array(offs), ctx.array(logits)

    def fresh(*shape):
        return ctx.array(np.full(shape, -7.0, np.float32))

    # --- RPN run
    reg0, bev0, sc0 = fresh(n, 6), fresh(n, 4), fresh(n)
    ops.offset_to_anchor(ctx, d_a, d_o, n, d_n, reg0)
    ops.project_anchors_f32(ctx, reg0, n, d_n, ext, p2, (1242.0, 375.0), d_bev_norm_tf=bev0)
    ops.softmax_fg(ctx, d_l, n, d_n, sc0)
    reg1, bev1, sc1 = fresh(n, 6), fresh(n, 4), fresh(n)
    ops.rpn_decode(ctx, d_a, d_o, d_l, n, d_n, ext, reg1, bev1, sc1)
    for a0, a1 in ((reg0, reg1), (bev0, bev1), (sc0, sc1)):
        assert np.array_equal(a0.download(), a1.download(), equal_nan=True)
    # --- the proposals' run
    m = max(1, n // 3)
    idx = rng.integers(0, n, m).astype(np.int32)
    d_idx = ctx.array(idx)
    d_m = ctx.array(np.asarray([min(m, n_valid)], np.int32)) if n_valid is not None else None
    rows0, tb0, ti0 = fresh(m, 6), fresh(m, 4), fresh(m, 4)
    ops.gather_rows(ctx, reg0, 6, d_idx, m, d_m, rows0)
    ops.project_anchors_f32(ctx, rows0, m, d_m, ext, p2, (1242.0, 375.0), d_bev_norm_tf=tb0, d_img_norm_tf=ti0)
    rows1, tb1, ti1 = fresh(m, 6), fresh(m, 4), fresh(m, 4)
    ops.gather_project(ctx, reg0, d_idx, m, d_m, ext, p2, (1242.0, 375.0), rows1, tb1, ti1)
    for a0, a1 in ((rows0, rows1), (tb0, tb1), (ti0, ti1)):
        assert np.array_equal(a0.download(), a1.download(), equal_nan=True)
    # --- behind the stage-2 head
    off4c = rng.normal(0, 0.2, (m, 10)).astype(np.float32)
    cls = rng.normal(0, 3, (m, 2)).astype(np.float32)
    ang = rng.normal(0, 1, (m, 2)).astype(np.float32)
    d_off, d_cls, d_ang = ctx.array(off4c), ctx.array(cls), ctx.array(ang)
    for with_angle in (True, False):
        b0, pa0, bt0, ns0, ds0, or0 = fresh(m, 7), fresh(m, 6), fresh(m, 4), fresh(m), fresh(m), fresh(m)
        ops.box_4c_decode(ctx, rows0, d_off, m, d_m, plane, ext, b0, pa0, bt0)
        ops.max_fg_logit(ctx, d_cls, 2, m, d_m, ns0)
        ops.softmax_fg(ctx, d_cls, m, d_m, ds0)
        if with_angle:
            ops.angle_vector_to_orientation(ctx, d_ang, m, d_m, or0)
        b1, pa1, bt1, ns1, ds1, or1 = fresh(m, 7), fresh(m, 6), fresh(m, 4), fresh(m), fresh(m), fresh(m)
        ops.final_decode(ctx, rows0, d_off, d_cls, d_ang if with_angle else None, m, d_m, plane, ext, b1, pa1, bt1,
                         ns1, ds1, or1 if with_angle else None)
        for a0, a1 in ((b0, b1), (pa0, pa1), (bt0, bt1), (ns0, ns1), (ds0, ds1), (or0, or1)):
            assert np.array_equal(a0.download(), a1.download(), equal_nan=True)
    with pytest.raises(ValueError):
        ops.final_decode(ctx, rows0, d_off, d_cls, d_ang, m, d_m, plane, ext, b1, pa1, bt1, ns1, ds1, None)
