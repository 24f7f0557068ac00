"""ctypes binding of libdodt_hip.so (include/dodt_hip.h).

The binding style follows the reference's one ctypes precedent,
wavedata/wavedata/tools/core/integral_image.py:22-23,96-122 (cdll.LoadLibrary +
explicit argtypes, caller-allocated outputs).  There is no CPU fallback: if the
shared library is missing, importing a compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DODT_HIP_LIB: another build of the same library (A/B timing by tools/)
LIB_PATH = os.environ.get('DODT_HIP_LIB') or os.path.join(_HERE, 'lib', 'libdodt_hip.so')

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED = 0, 1, 2, 3
PTS_VELO_XYZI, PTS_CAM_3XN = 0, 1
EXTRACTOR_VGG_PYR = 0
EXTRACTOR_VGG = 1
EXTRACTOR_SHARED_GPU = 0x100
EXTRACTOR_BF16 = 0x200
EXTRACTOR_SPLIT = 0x400
FC_RELU = 1
FC_BF16 = 2


class DodtError(RuntimeError):
    pass


class BevParams(C.Structure):
    _fields_ = [
        ('point_format', C.c_int32),
        ('num_slices', C.c_int32),
        ('velo_to_cam', C.c_double * 12),
        ('p2', C.c_double * 12),
        ('im_w', C.c_double),
        ('im_h', C.c_double),
        ('plane', C.c_double * 4),
        ('extents', C.c_double * 6),
        ('voxel_size', C.c_double),
        ('height_lo', C.c_double),
        ('height_hi', C.c_double),
        ('occ_lo', C.c_double),
        ('occ_hi', C.c_double),
        ('has_pre_transform', C.c_int32),
        ('reserved_', C.c_int32),
        ('pre_translate', C.c_double * 3),
        ('pre_rotate', C.c_double * 9),
    ]


class LayerInfo(C.Structure):
    _fields_ = [('name', C.c_char * 32), ('kernel', C.c_char * 48), ('launches', C.c_int32),
                ('items', C.c_int32), ('flops_direct', C.c_double), ('flops_executed', C.c_double),
                ('bytes', C.c_double), ('ms', C.c_float), ('reserved_', C.c_int32)]


_vp = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_pi32 = C.c_void_p      # device int32*
_pf = C.c_void_p        # device float*
_hf = C.c_void_p        # host float*

# name -> (restype, argtypes); every name here is declared in include/dodt_hip.h
SIGNATURES = {
    'dodt_version': (_i, []),
    'dodt_last_error': (C.c_char_p, []),
    'dodt_ctx_create': (_i, [_i, C.POINTER(_vp)]),
    'dodt_mark': (_i, [_vp, _i]),
    'dodt_ctx_wait_mark': (_i, [_vp, _vp, _i]),
    'dodt_mark_elapsed': (_i, [_vp, _i, _vp, _i, C.POINTER(_f)]),
    'dodt_ctx_create_high_priority': (_i, [_i, C.POINTER(_vp)]),
    'dodt_ctx_create_on_stream': (_i, [_i, _vp, C.POINTER(_vp)]),
    'dodt_ctx_destroy': (_i, [_vp]),
    'dodt_ctx_sync': (_i, [_vp]),
    'dodt_ctx_wait_for': (_i, [_vp, _vp]),
    'dodt_malloc': (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    'dodt_free': (_i, [_vp, _vp]),
    'dodt_memcpy_h2d': (_i, [_vp, _vp, _vp, C.c_size_t]),
    'dodt_memcpy_d2h': (_i, [_vp, _vp, _vp, C.c_size_t]),
    'dodt_memset': (_i, [_vp, _vp, _i, C.c_size_t]),
    'dodt_pinned_alloc': (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    'dodt_pinned_free': (_i, [_vp, _vp]),
    'dodt_memcpy_h2d_async': (_i, [_vp, _vp, _vp, C.c_size_t]),
    'dodt_fetch_i32_begin': (_i, [_vp, _pi32, _i, _i]),
    'dodt_fetch_i32_end': (_i, [_vp, _i, C.POINTER(C.c_int32), _i]),
    'dodt_timer_start': (_i, [_vp]),
    'dodt_timer_stop': (_i, [_vp, C.POINTER(_f)]),
    'dodt_bev_slices': (_i, [_vp, _vp, _i, C.POINTER(BevParams), _pf, _vp]),
    'dodt_bev_status': (_i, [_vp, C.POINTER(_i)]),
    'dodt_anchor_filter': (_i, [_vp, _vp, _i, _i, _pi32, _i, _i, _pi32, _pi32]),
    'dodt_project_anchors_f64': (_i, [_vp, _vp, _pi32, _i, _pi32,
                                      C.POINTER(_d), C.POINTER(_d), _d, _d,
                                      _pf, _pf, _pf]),
    'dodt_project_anchors_f32': (_i, [_vp, _pf, _i, _pi32, C.POINTER(_f),
                                      C.POINTER(_f), _f, _f, _pf, _pf, _pf]),
    'dodt_img_preprocess': (_i, [_vp, _vp, _i, _i, _i, _i, _i,
                                 C.POINTER(_f), _pf]),
    'dodt_conv_mode': (_i, []),
    'dodt_extractor_create': (_i, [_vp, _i, _i, _i, _i, _i, _i,
                                   C.POINTER(_vp)]),
    'dodt_extractor_destroy': (_i, [_vp]),
    'dodt_extractor_set_layer': (_i, [_vp, C.c_char_p, _vp, _i, _i, _i, _i,
                                      _vp, _vp, _vp]),
    'dodt_extractor_input': (_i, [_vp, C.POINTER(_vp), C.POINTER(C.c_longlong)]),
    'dodt_extractor_forward': (_i, [_vp, _pf, _pf, _pf]),
    'dodt_extractor_forward_padded': (_i, [_vp, _pf, _pf, _pf]),
    'dodt_extractor_set_input': (_i, [_vp, _pf]),
    'dodt_extractor_read_activation': (_i, [_vp, C.c_char_p, _vp,
                                            C.POINTER(_i), C.POINTER(_i),
                                            C.POINTER(_i)]),
    'dodt_extractor_output_shape': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'dodt_extractor_first_layers_folded': (_i, [_vp]),
    'dodt_extractor_flops': (_d, [_vp]),
    'dodt_extractor_mfma_flops': (_d, [_vp]),
    'dodt_extractor_bytes': (_d, [_vp]),
    'dodt_extractor_layer_count': (_i, [_vp]),
    'dodt_extractor_forward_timed': (_i, [_vp, _pf, _pf, _pf, C.POINTER(LayerInfo), _i]),
    'dodt_crop_and_resize': (_i, [_vp, _pf, _i, _i, _i, _pf, _i, _pi32, _i, _i,
                                  _pf]),
    'dodt_crop_and_resize_strided': (_i, [_vp, _pf, _i, _i, _i, _pf, _i, _pi32, _i, _i, _pf,
                                          C.c_longlong]),
    'dodt_correlation': (_i, [_vp, _pf, _pf, _i, _i, _i, _i, _i, _i, _pf]),
    'dodt_mean_fusion': (_i, [_vp, _pf, _pf, _i, _pi32, _i, _pf]),
    'dodt_fc_create': (_i, [_vp, _i, _i, _hf, _hf, _i, C.POINTER(_vp)]),
    'dodt_fc_create_ex': (_i, [_vp, _i, _i, _hf, _hf, _i, C.POINTER(_vp)]),
    'dodt_fc_destroy': (_i, [_vp]),
    'dodt_fc_forward': (_i, [_vp, _vp, _pf, _pf, _i, _i, _pi32, _pf, _i]),
    'dodt_fc_forward_split': (_i, [_vp, _vp, _pf, _i, _i, _pi32, _i, C.POINTER(_i), C.POINTER(_vp)]),
    'dodt_fc_flops': (_d, [_vp, _i]),
    'dodt_fc_bf16_row_elems': (_i, [_vp]),
    'dodt_fc_forward_bf16': (_i, [_vp, _vp, _vp, _i, _i, _pi32, _vp, _i, _i]),
    'dodt_fc_forward_split_bf16': (_i, [_vp, _vp, _vp, _i, _i, _pi32, _i, C.POINTER(_i), C.POINTER(_vp)]),
    'dodt_rows_to_bf16': (_i, [_vp, _pf, _pf, _i, _pi32, _i, _i, _vp, _i]),
    'dodt_nms': (_i, [_vp, _pf, _pf, _i, _pi32, _i, _f, _pi32, _pi32]),
    'dodt_offset_to_anchor': (_i, [_vp, _pf, _pf, _i, _pi32, _pf]),
    'dodt_softmax_fg': (_i, [_vp, _pf, _i, _pi32, _pf]),
    'dodt_rpn_decode': (_i, [_vp, _pf, _pf, _pf, _i, _pi32, C.POINTER(_f), _pf, _pf, _pf]),
    'dodt_gather_project': (_i, [_vp, _pf, _pi32, _i, _pi32, C.POINTER(_f), C.POINTER(_f), _f, _f, _pf, _pf, _pf]),
    'dodt_final_decode': (_i, [_vp, _pf, _pf, _pf, _pf, _i, _pi32, C.POINTER(_f), C.POINTER(_f),
                               _pf, _pf, _pf, _pf, _pf, _pf]),
    'dodt_gather_rows': (_i, [_vp, _pf, _i, _pi32, _i, _pi32, _pf]),
    'dodt_max_fg_logit': (_i, [_vp, _pf, _i, _i, _pi32, _pf]),
    'dodt_pack_detections': (_i, [_vp, _pf, _pf, _pf, _pf, _pi32, _pi32, _i, _f, _pf, _pi32]),
    'dodt_angle_vector_to_orientation': (_i, [_vp, _pf, _i, _pi32, _pf]),
    'dodt_box_4c_decode': (_i, [_vp, _pf, _pf, _i, _pi32, C.POINTER(_f),
                                C.POINTER(_f), _pf, _pf, _pf]),
    'dodt_comm_unique_id': (_i, [_vp]),
    'dodt_comm_create': (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    'dodt_comm_destroy': (_i, [_vp]),
    'dodt_comm_attach': (_i, [_vp, _vp]),
    'dodt_comm_rank': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'dodt_comm_set_late_peer': (_i, [_vp, _d]),
    'dodt_all_gather_records': (_i, [_vp, _vp, _i, _pf, _pi32, _i, _i, _i, _i, _pf, _pi32]),
    'dodt_comm_join': (_i, [_vp, _i, _vp]),
    'dodt_comm_sync': (_i, [_vp]),
    'dodt_comm_barrier': (_i, [_vp]),
    'dodt_comm_max_f64': (_i, [_vp, C.POINTER(_d)]),
}
COMM_ID_BYTES = 128

_lib = None


def load():
    """Load libdodt_hip.so once and attach argtypes.  Raises DodtError if the
    library has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DodtError(
            'libdodt_hip.so not found at %s -- build it with '
            '`make -C dodt_amd/csrc` (there is no CPU fallback)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    """Map a status code to the exception the reference would raise."""
    if rc == OK:
        return
    msg = load().dodt_last_error().decode('utf-8', 'replace')
    if rc == ERR_INVALID:
        raise ValueError(msg or what)
    raise DodtError('%s: %s (status %d)' % (what, msg, rc))
