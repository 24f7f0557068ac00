"""Thin device-level wrappers: DeviceArray in, DeviceArray out, one C-ABI call each.

These are what the frame-pair pipeline is made of; the avod.core-shaped host API
in dodt_amd/core/ wraps them with numpy upload/download.
"""
import ctypes as C
import os

import numpy as np

from dodt_amd import _lib


def _arr(ctype, values):
    values = [float(v) for v in np.asarray(values, dtype=np.float64).reshape(-1)]
    return (ctype * len(values))(*values)


def _p(a):
    return None if a is None else C.c_void_p(a.ptr)


def make_bev_params(cfg, velo_to_cam=None, p2=None, im_wh=None,
                    point_format=_lib.PTS_VELO_XYZI, ground_plane=None,
                    area_extents=None, voxel_size=None):
    """dodt_bev_params from a config dict (dodt_amd.config) and calibration."""
    bp = _lib.BevParams()
    bp.point_format = point_format
    bp.num_slices = int(cfg['num_slices'])
    m = np.zeros(12) if velo_to_cam is None else \
        np.asarray(velo_to_cam, dtype=np.float64).reshape(-1)[:12]
    p = np.zeros(12) if p2 is None else np.asarray(p2, dtype=np.float64).reshape(-1)
    bp.velo_to_cam = (C.c_double * 12)(*m)
    bp.p2 = (C.c_double * 12)(*p)
    bp.im_w, bp.im_h = (0.0, 0.0) if im_wh is None else (float(im_wh[0]), float(im_wh[1]))
    plane = cfg['ground_plane'] if ground_plane is None else ground_plane
    bp.plane = (C.c_double * 4)(*[float(v) for v in plane])
    ext = cfg['area_extents'] if area_extents is None else area_extents
    bp.extents = (C.c_double * 6)(*[float(v) for v in np.asarray(ext).reshape(-1)])
    bp.voxel_size = float(cfg['voxel_size'] if voxel_size is None else voxel_size)
    bp.height_lo = float(cfg['height_lo'])
    bp.height_hi = float(cfg['height_hi'])
    bp.occ_lo = float(cfg['anchor_filter_lo'])
    bp.occ_hi = float(cfg['anchor_filter_hi'])
    bp.has_pre_transform = 0
    return bp


def with_ego_motion(bev_params, trans, matrix):
    """Copy of bev_params that registers a pair's second frame into the first frame's
    coordinates first: p' = (p + trans) @ matrix in the velodyne frame
    (kitti_tracking_dataset.py:303-335; trans, matrix from
    dodt_amd.datasets.kitti.kitti_tracking_utils.coordinate_transform)."""
    bp = _lib.BevParams.from_buffer_copy(bev_params)
    t = np.asarray(trans, np.float64).reshape(3)
    m = np.asarray(matrix, np.float64).reshape(9)
    bp.has_pre_transform = 1
    bp.pre_translate = (C.c_double * 3)(*t)
    bp.pre_rotate = (C.c_double * 9)(*m)
    return bp


def bev_slices(ctx, d_points, n_points, bev_params, d_bev_out, d_occ_bits=None):
    _lib.check(ctx.lib.dodt_bev_slices(ctx.handle, _p(d_points), int(n_points),
                                       C.byref(bev_params), _p(d_bev_out),
                                       _p(d_occ_bits)), 'dodt_bev_slices')


def bev_status(ctx):
    f = C.c_int()
    _lib.check(ctx.lib.dodt_bev_status(ctx.handle, C.byref(f)), 'dodt_bev_status')
    return f.value


def anchor_filter(ctx, d_occ_bits, nx, nz, d_cells, n_anchors, d_keep_idx, d_count,
                  density_threshold=1):
    _lib.check(ctx.lib.dodt_anchor_filter(
        ctx.handle, _p(d_occ_bits), int(nx), int(nz), _p(d_cells), int(n_anchors),
        int(density_threshold), _p(d_keep_idx), _p(d_count)), 'dodt_anchor_filter')


def project_anchors_f64(ctx, d_anchors, d_idx, n, d_n, bev_extents, p2, im_wh,
                        d_bev_norm=None, d_img_norm=None, d_anchors_f32=None):
    _lib.check(ctx.lib.dodt_project_anchors_f64(
        ctx.handle, _p(d_anchors), _p(d_idx), int(n), _p(d_n),
        _arr(C.c_double, bev_extents), _arr(C.c_double, p2),
        float(im_wh[0]), float(im_wh[1]), _p(d_bev_norm), _p(d_img_norm),
        _p(d_anchors_f32)), 'dodt_project_anchors_f64')


def project_anchors_f32(ctx, d_anchors, n, d_n, bev_extents, p2, im_wh,
                        d_bev=None, d_bev_norm_tf=None, d_img_norm_tf=None):
    _lib.check(ctx.lib.dodt_project_anchors_f32(
        ctx.handle, _p(d_anchors), int(n), _p(d_n), _arr(C.c_float, bev_extents),
        _arr(C.c_float, p2), float(im_wh[0]), float(im_wh[1]), _p(d_bev),
        _p(d_bev_norm_tf), _p(d_img_norm_tf)), 'dodt_project_anchors_f32')


def img_preprocess(ctx, d_img_u8, in_hw, out_hw, out_c, mean_rgb, d_out):
    _lib.check(ctx.lib.dodt_img_preprocess(
        ctx.handle, _p(d_img_u8), int(in_hw[0]), int(in_hw[1]), int(out_hw[0]),
        int(out_hw[1]), int(out_c), _arr(C.c_float, mean_rgb), _p(d_out)),
        'dodt_img_preprocess')


def crop_and_resize(ctx, d_image, hwc, d_boxes, n, d_n, crop_hw, d_out, out_box_stride=None):
    """out_box_stride (floats): box b's crop starts at d_out + b * out_box_stride (default: packed)."""
    if out_box_stride is None:
        out_box_stride = int(crop_hw[0]) * int(crop_hw[1]) * int(hwc[2])
    _lib.check(ctx.lib.dodt_crop_and_resize_strided(
        ctx.handle, _p(d_image), int(hwc[0]), int(hwc[1]), int(hwc[2]), _p(d_boxes),
        int(n), _p(d_n), int(crop_hw[0]), int(crop_hw[1]), _p(d_out), int(out_box_stride)),
        'dodt_crop_and_resize')


def nms(ctx, d_boxes, d_scores, n, d_n, max_out, iou_threshold, d_sel, d_count):
    _lib.check(ctx.lib.dodt_nms(
        ctx.handle, _p(d_boxes), _p(d_scores), int(n), _p(d_n), int(max_out),
        float(iou_threshold), _p(d_sel), _p(d_count)), 'dodt_nms')


def offset_to_anchor(ctx, d_anchors, d_offsets, n, d_n, d_out):
    _lib.check(ctx.lib.dodt_offset_to_anchor(
        ctx.handle, _p(d_anchors), _p(d_offsets), int(n), _p(d_n), _p(d_out)),
        'dodt_offset_to_anchor')


def softmax_fg(ctx, d_logits, n, d_n, d_scores):
    _lib.check(ctx.lib.dodt_softmax_fg(
        ctx.handle, _p(d_logits), int(n), _p(d_n), _p(d_scores)), 'dodt_softmax_fg')


def rpn_decode(ctx, d_anchors, d_offsets, d_logits, n, d_n, bev_extents, d_regressed, d_bev_norm_tf, d_scores):
    """offset_to_anchor + project_anchors_f32 (normalised BEV boxes) + softmax_fg in one launch."""
    _lib.check(ctx.lib.dodt_rpn_decode(
        ctx.handle, _p(d_anchors), _p(d_offsets), _p(d_logits), int(n), _p(d_n), _arr(C.c_float, bev_extents),
        _p(d_regressed), _p(d_bev_norm_tf), _p(d_scores)), 'dodt_rpn_decode')


def gather_project(ctx, d_src, d_idx, n, d_n, bev_extents, p2, im_wh, d_rows, d_bev_norm_tf, d_img_norm_tf):
    """gather_rows (width 6) + project_anchors_f32 (both views) in one launch."""
    _lib.check(ctx.lib.dodt_gather_project(
        ctx.handle, _p(d_src), _p(d_idx), int(n), _p(d_n), _arr(C.c_float, bev_extents), _arr(C.c_float, p2),
        float(im_wh[0]), float(im_wh[1]), _p(d_rows), _p(d_bev_norm_tf), _p(d_img_norm_tf)), 'dodt_gather_project')


def final_decode(ctx, d_top_anchors, d_offsets, d_cls_logits, d_angle_vectors, n, d_n, plane, bev_extents,
                 d_boxes_3d, d_pred_anchors, d_bev_tf, d_nms_scores, d_det_scores, d_orientations):
    """box_4c_decode + max_fg_logit + softmax_fg [+ angle_vector_to_orientation] in one launch."""
    _lib.check(ctx.lib.dodt_final_decode(
        ctx.handle, _p(d_top_anchors), _p(d_offsets), _p(d_cls_logits), _p(d_angle_vectors), int(n), _p(d_n),
        _arr(C.c_float, plane), _arr(C.c_float, bev_extents), _p(d_boxes_3d), _p(d_pred_anchors), _p(d_bev_tf),
        _p(d_nms_scores), _p(d_det_scores), _p(d_orientations)), 'dodt_final_decode')


def gather_rows(ctx, d_src, width, d_idx, n, d_n, d_out):
    _lib.check(ctx.lib.dodt_gather_rows(
        ctx.handle, _p(d_src), int(width), _p(d_idx), int(n), _p(d_n), _p(d_out)),
        'dodt_gather_rows')


def box_4c_decode(ctx, d_top_anchors, d_offsets, n, d_n, plane, bev_extents,
                  d_boxes_3d=None, d_pred_anchors=None, d_bev_tf=None):
    _lib.check(ctx.lib.dodt_box_4c_decode(
        ctx.handle, _p(d_top_anchors), _p(d_offsets), int(n), _p(d_n),
        _arr(C.c_float, plane), _arr(C.c_float, bev_extents), _p(d_boxes_3d),
        _p(d_pred_anchors), _p(d_bev_tf)), 'dodt_box_4c_decode')


def max_fg_logit(ctx, d_logits, n_cls, n, d_n, d_scores):
    _lib.check(ctx.lib.dodt_max_fg_logit(
        ctx.handle, _p(d_logits), int(n_cls), int(n), _p(d_n), _p(d_scores)),
        'dodt_max_fg_logit')


def angle_vector_to_orientation(ctx, d_angle_vectors, n, d_n, d_orientations):
    _lib.check(ctx.lib.dodt_angle_vector_to_orientation(
        ctx.handle, _p(d_angle_vectors), int(n), _p(d_n), _p(d_orientations)),
        'dodt_angle_vector_to_orientation')


def pack_detections(ctx, d_boxes_3d, d_scores, d_sel, d_count, max_det, frame_mark,
                    d_rec, d_count_out, d_corr_offsets=None, d_orientations=None):
    _lib.check(ctx.lib.dodt_pack_detections(
        ctx.handle, _p(d_boxes_3d), _p(d_scores), _p(d_orientations), _p(d_corr_offsets),
        _p(d_sel), _p(d_count), int(max_det), float(frame_mark), _p(d_rec), _p(d_count_out)),
        'dodt_pack_detections')


def fetch_i32_begin(ctx, d_src, n, slot):
    _lib.check(ctx.lib.dodt_fetch_i32_begin(ctx.handle, _p(d_src), int(n), int(slot)),
               'dodt_fetch_i32_begin')


def fetch_i32_end(ctx, slot, n):
    buf = (C.c_int32 * n)()
    _lib.check(ctx.lib.dodt_fetch_i32_end(ctx.handle, int(slot), buf, int(n)),
               'dodt_fetch_i32_end')
    return list(buf)


def correlation(ctx, d_a, d_b, hwc, max_displacement, stride_2, pad, d_out):
    _lib.check(ctx.lib.dodt_correlation(
        ctx.handle, _p(d_a), _p(d_b), int(hwc[0]), int(hwc[1]), int(hwc[2]),
        int(max_displacement), int(stride_2), int(pad), _p(d_out)), 'dodt_correlation')


def mean_fusion(ctx, d_a, d_b, rows, d_n, row_floats, d_out):
    """d_out = (d_a + d_b) / 2 over min(*d_n, rows) rows of row_floats floats."""
    _lib.check(ctx.lib.dodt_mean_fusion(ctx.handle, _p(d_a), _p(d_b), int(rows), _p(d_n),
                                        int(row_floats), _p(d_out)), 'dodt_mean_fusion')


def rows_to_bf16(ctx, d_a, d_b, rows, d_n, row_floats, in_ld, d_out, out_ld):
    """d_out (rows, out_ld) bf16 (uint16 storage) = bf16((d_a + d_b) / 2) [bf16(d_a) when d_b is None] over the first
    row_floats elements of min(*d_n, rows) rows (input rows in_ld floats apart), zeros behind them: the first
    layer's rows of a bf16 head (dodt_rows_to_bf16)."""
    _lib.check(ctx.lib.dodt_rows_to_bf16(ctx.handle, _p(d_a), _p(d_b), int(rows), _p(d_n), int(row_floats),
                                         int(in_ld), _p(d_out), int(out_ld)), 'dodt_rows_to_bf16')


def bf16_to_float(a):
    """uint16 array of bf16 bit patterns -> float32 (host side of tests and tools)."""
    return (np.asarray(a, np.uint16).astype(np.uint32) << 16).view(np.float32)


class FullyConnected(object):
    """y = act(x w + b) on the device (dodt_fc_*).  w (K,N) row-major, the layout of the
    TF variable (conv kernels reshaped (kh*kw*cin, cout))."""

    def __init__(self, ctx, w, b, relu, dtype='f32'):
        """dtype 'bf16': bf16 MFMA, fp32 accumulate (DODT_FC_BF16)."""
        if dtype not in ('f32', 'bf16'):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        if w.ndim != 2 or b.shape != (w.shape[1],):
            raise ValueError('weights must be (K,N) and bias (N,)')
        self.ctx, self.K, self.N, self.dtype = ctx, int(w.shape[0]), int(w.shape[1]), dtype
        h = C.c_void_p()
        flags = (_lib.FC_RELU if relu else 0) | (_lib.FC_BF16 if dtype == 'bf16' else 0)
        _lib.check(ctx.lib.dodt_fc_create_ex(ctx.handle, self.K, self.N,
                                             w.ctypes.data_as(C.c_void_p),
                                             b.ctypes.data_as(C.c_void_p), flags, C.byref(h)),
                   'dodt_fc_create_ex')
        self.handle = h

    def forward(self, d_x, M, d_y, ldx=None, ldy=None, d_x2=None, d_m=None, ctx=None):
        c = ctx or self.ctx
        _lib.check(c.lib.dodt_fc_forward(
            self.handle, c.handle, _p(d_x), _p(d_x2), int(self.K if ldx is None else ldx),
            int(M), _p(d_m), _p(d_y), int(self.N if ldy is None else ldy)), 'dodt_fc_forward')

    def can_split(self, ldx=None):
        """Whether forward_split takes this layer (dodt_fc_forward_split's conditions)."""
        return self.N <= 32 and self.K % 16 == 0 and \
            (self.K if ldx is None else ldx) % 4 == 0 and os.environ.get('DODT_FC_SKINNY') != '0'

    def forward_split(self, d_x, M, d_ys, widths, ldx=None, d_m=None, ctx=None):
        """One launch, columns [0, widths[0]) to d_ys[0] (M, widths[0]), the next widths[1] to d_ys[1], ..."""
        c = ctx or self.ctx
        w = (C.c_int * len(widths))(*[int(v) for v in widths])
        ys = (C.c_void_p * len(d_ys))(*[_p(y) for y in d_ys])
        _lib.check(c.lib.dodt_fc_forward_split(
            self.handle, c.handle, _p(d_x), int(self.K if ldx is None else ldx), int(M), _p(d_m),
            len(widths), w, ys), 'dodt_fc_forward_split')

    def bf16_row_elems(self):
        """bf16 elements an input row of forward_bf16 must hold (zeros beyond K); 0: the layer has no such path."""
        return int(self.ctx.lib.dodt_fc_bf16_row_elems(self.handle)) if self.dtype == 'bf16' else 0

    def forward_bf16(self, d_x, M, d_y, ldx, ldy=None, d_m=None, y_bf16=True, ctx=None):
        """The layer on rows that are already bf16 (d_x: uint16 storage, ldx elements per row); d_y bf16 rows
        (y_bf16) or float32."""
        c = ctx or self.ctx
        _lib.check(c.lib.dodt_fc_forward_bf16(self.handle, c.handle, _p(d_x), int(ldx), int(M), _p(d_m), _p(d_y),
                                              int(self.N if ldy is None else ldy), 1 if y_bf16 else 0),
                   'dodt_fc_forward_bf16')

    def forward_split_bf16(self, d_x, M, d_ys, widths, ldx, d_m=None, ctx=None):
        c = ctx or self.ctx
        w = (C.c_int * len(widths))(*[int(v) for v in widths])
        ys = (C.c_void_p * len(d_ys))(*[_p(y) for y in d_ys])
        _lib.check(c.lib.dodt_fc_forward_split_bf16(
            self.handle, c.handle, _p(d_x), int(ldx), int(M), _p(d_m), len(widths), w, ys), 'dodt_fc_forward_split_bf16')

    def flops(self, M):
        return 2.0 * M * self.K * self.N

    def close(self):
        if self.handle is not None:
            self.ctx.lib.dodt_fc_destroy(self.handle)
            self.handle = None
