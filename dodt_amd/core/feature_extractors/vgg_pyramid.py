"""BevVggPyr / ImgVggPyr on the GPU -- stand where
avod/core/feature_extractors/bev_vgg_pyramid.py:30-178 and
img_vgg_pyramid.py:30-177 stand.  `build` keeps the reference's argument list
but runs eagerly: numpy NHWC in, (feature_maps, end_points) out.  Weights are
given by TF variable name (<scope>/convK/convK_j/weights ... ) through
`load_params`, a dict layer -> dict(w, beta, mean, var)."""
import ctypes as C

import numpy as np

from dodt_amd import _lib, device


class _VggPyr(object):
    PAD_TOP = 0
    KIND = _lib.EXTRACTOR_VGG_PYR

    def __init__(self, extractor_config=None, ctx=None, shared_gpu=False, conv_dtype='f32'):
        """shared_gpu: other streams keep the GPU busy beside this net (the frame-pair
        pipeline): layers run as single launches (include/dodt_hip.h).
        conv_dtype: 'f32' (fp32 MFMA, the reference's arithmetic), 'f32s' (split mode: hi + lo
        bf16 pairs on the bf16 MFMA, fp32-grade, DODT_EXTRACTOR_SPLIT) or 'bf16' (bf16 MFMA,
        fp32 accumulate; DODT_EXTRACTOR_BF16)."""
        if conv_dtype not in ('f32', 'f32s', 'bf16'):
            raise ValueError("conv_dtype must be 'f32', 'f32s' or 'bf16'")
        self._bf16 = conv_dtype == 'bf16'
        self._split = conv_dtype == 'f32s'
        self.config = extractor_config
        self._ctx = ctx
        self._shared_gpu = bool(shared_gpu)
        self._handle = None
        self._shape = None
        self._params = None

    # -- lifetime ---------------------------------------------------------------
    def _ensure(self, batch, h, w, c):
        shape = (batch, h, w, c)
        if self._handle is not None and self._shape == shape:
            return
        self.close()
        self._ctx = self._ctx or device.default_context()
        hnd = C.c_void_p()
        _lib.check(self._ctx.lib.dodt_extractor_create(
            self._ctx.handle,
            self.KIND | (_lib.EXTRACTOR_SHARED_GPU if self._shared_gpu else 0)
            | (_lib.EXTRACTOR_BF16 if self._bf16 else 0)
            | (_lib.EXTRACTOR_SPLIT if self._split else 0),
            h, w, c, self.PAD_TOP,
            batch, C.byref(hnd)), 'dodt_extractor_create')
        self._handle = hnd
        self._shape = shape
        if self._params is not None:
            self._push_params()

    def close(self):
        if self._handle is not None:
            self._ctx.lib.dodt_extractor_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ----------------------------------------------------------------
    def load_params(self, params):
        self._params = params
        if self._handle is not None:
            self._push_params()

    def _push_params(self):
        for name, p in self._params.items():
            w = np.ascontiguousarray(p['w'], dtype=np.float32)
            b = np.ascontiguousarray(p['beta'], dtype=np.float32)
            m = np.ascontiguousarray(p['mean'], dtype=np.float32)
            v = np.ascontiguousarray(p['var'], dtype=np.float32)
            _lib.check(self._ctx.lib.dodt_extractor_set_layer(
                self._handle, name.encode(), w.ctypes.data, w.shape[0], w.shape[1],
                w.shape[2], w.shape[3], b.ctypes.data, m.ctypes.data, v.ctypes.data),
                'dodt_extractor_set_layer(%s)' % name)

    # -- forward ------------------------------------------------------------------
    def forward_device(self, d_in, d_feat, d_bottleneck=None):
        _lib.check(self._ctx.lib.dodt_extractor_forward(
            self._handle, None if d_in is None else C.c_void_p(d_in.ptr),
            C.c_void_p(d_feat.ptr),
            None if d_bottleneck is None else C.c_void_p(d_bottleneck.ptr)),
            'dodt_extractor_forward')

    def forward_device_padded(self, d_x0, d_feat, d_bottleneck=None):
        """forward_device on an input the caller keeps in the extractor's input layout, (batch, PAD_TOP + h, w, c)
        with zero pad rows: read in place, no copy (dodt_extractor_forward_padded)."""
        _lib.check(self._ctx.lib.dodt_extractor_forward_padded(
            self._handle, C.c_void_p(d_x0.ptr), C.c_void_p(d_feat.ptr),
            None if d_bottleneck is None else C.c_void_p(d_bottleneck.ptr)), 'dodt_extractor_forward_padded')

    def set_input(self, d_x0):
        """Forwards without an input argument read d_x0 (extractor input layout) from now on; None: the extractor's
        own buffer again (dodt_extractor_set_input)."""
        _lib.check(self._ctx.lib.dodt_extractor_set_input(
            self._handle, None if d_x0 is None else C.c_void_p(d_x0.ptr)), 'dodt_extractor_set_input')

    def forward_timed(self, d_in, d_feat, d_bottleneck=None):
        """One forward with a HIP event pair around every layer; waits for the stream.  Returns a
        list of dicts in launch order: name, kernel (the __global__ function), launches, items,
        flops_direct, flops_executed, bytes, ms."""
        lib = self._ctx.lib
        n = lib.dodt_extractor_layer_count(self._handle)
        info = (_lib.LayerInfo * n)()
        _lib.check(lib.dodt_extractor_forward_timed(
            self._handle, None if d_in is None else C.c_void_p(d_in.ptr), C.c_void_p(d_feat.ptr),
            None if d_bottleneck is None else C.c_void_p(d_bottleneck.ptr), info, n),
            'dodt_extractor_forward_timed')
        return [dict(name=i.name.decode(), kernel=i.kernel.decode(), launches=i.launches,
                     items=i.items, flops_direct=i.flops_direct, flops_executed=i.flops_executed,
                     bytes=i.bytes, ms=i.ms) for i in info]

    def input_view(self):
        """DeviceArray aliasing the extractor's own input buffer (zero copy)."""
        p = C.c_void_p()
        stride = C.c_longlong()
        _lib.check(self._ctx.lib.dodt_extractor_input(
            self._handle, C.byref(p), C.byref(stride)), 'dodt_extractor_input')
        return p.value, stride.value

    def output_shape(self):
        """(h, w, c) of the feature map a forward returns (needs a built extractor)."""
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._ctx.lib.dodt_extractor_output_shape(
            self._handle, C.byref(h), C.byref(w), C.byref(c)), 'dodt_extractor_output_shape')
        return h.value, w.value, c.value

    @property
    def first_layers_folded(self):
        """True when conv1_1 runs inside conv1_2's launch (bf16 conv path): its map is not stored, and its products are
        hi + lo bf16 pairs on the bf16 MFMA (include/dodt_hip.h dodt_extractor_first_layers_folded)."""
        rc = self._ctx.lib.dodt_extractor_first_layers_folded(self._handle)
        if rc < 0:
            _lib.check(rc, 'dodt_extractor_first_layers_folded')
        return bool(rc)

    def flops(self):
        return self._ctx.lib.dodt_extractor_flops(self._handle)

    def mfma_flops(self):
        """FLOPs the matrix pipe executes (Winograd layers: 16/36 of the direct count)."""
        return self._ctx.lib.dodt_extractor_mfma_flops(self._handle)

    def bytes(self):
        """Algorithmic HBM bytes of one forward."""
        return self._ctx.lib.dodt_extractor_bytes(self._handle)

    def activation(self, name):
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        lib = self._ctx.lib
        _lib.check(lib.dodt_extractor_read_activation(
            self._handle, name.encode(), None, C.byref(h), C.byref(w), C.byref(c)),
            'read_activation')
        out = np.empty((self._shape[0], h.value, w.value, c.value), np.float32)
        _lib.check(lib.dodt_extractor_read_activation(
            self._handle, name.encode(), out.ctypes.data, None, None, None),
            'read_activation')
        return out

    def build(self, inputs, input_pixel_size=None, is_training=False, scope=None,
              with_bottleneck=False):
        """inputs: (batch, H, W, C) float32.  Returns (feature_maps (batch,H,W,32),
        end_points dict) -- end_points holds the bottleneck when requested."""
        if is_training:
            raise NotImplementedError('inference path only (batch-norm in '
                                      'moving-average form)')
        x = np.ascontiguousarray(inputs, dtype=np.float32)
        if x.ndim != 4:
            raise ValueError('inputs must be (batch, H, W, C)')
        b, h, w, c = x.shape
        if c % 2:                       # conv kernels take channel pairs
            x = np.concatenate([x, np.zeros((b, h, w, 1), np.float32)], axis=3)
            c += 1
        self._ensure(b, h, w, c)
        ctx = self._ctx
        d_in = ctx.array(x)
        oh, ow, oc = self.output_shape()
        d_feat = ctx.empty((b, oh, ow, oc), np.float32)
        d_bn = ctx.empty((b, oh, ow, 1), np.float32) if with_bottleneck else None
        self.forward_device(d_in, d_feat, d_bn)
        end_points = {}
        if with_bottleneck:
            end_points['bottleneck'] = d_bn.download()
        return d_feat.download(), end_points


class BevVggPyr(_VggPyr):
    PAD_TOP = 4         # bev_vgg_pyramid.py:58 pads 700 -> 704

    def preprocess_input(self, tensor_in, output_shape):
        """bev_feature_extractor.py:10-26: identity when shapes already match."""
        t = np.asarray(tensor_in)
        if tuple(t.shape[1:3]) != tuple(output_shape):
            raise NotImplementedError('BEV resize is not on the DODT path')
        return t


class ImgVggPyr(_VggPyr):
    PAD_TOP = 0
    _R_MEAN, _G_MEAN, _B_MEAN = 92.8403, 97.7996, 93.5843

    def preprocess_input(self, tensor_in, output_size, ctx=None):
        """img_feature_extractor.py:16-35: bilinear resize + mean subtraction.
        tensor_in (1,H,W,3) or (H,W,3) uint8 RGB -> (1,oh,ow,3) float32."""
        from dodt_amd import ops
        img = np.ascontiguousarray(np.squeeze(np.asarray(tensor_in)), dtype=np.uint8)
        ctx = ctx or self._ctx or device.default_context()
        self._ctx = ctx
        oh, ow = int(output_size[0]), int(output_size[1])
        d_img = ctx.array(img)
        d_out = ctx.empty((oh, ow, 4), np.float32)
        ops.img_preprocess(ctx, d_img, img.shape[:2], (oh, ow), 4,
                           (self._R_MEAN, self._G_MEAN, self._B_MEAN), d_out)
        return d_out.download()[None, :, :, :3]
