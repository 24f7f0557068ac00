"""BevVgg / ImgVgg on the GPU -- stand where avod/core/feature_extractors/bev_vgg.py:34-118
and img_vgg.py:33-120 stand (the AVOD cars_example configuration, BASELINE.json configs[0]):
the VGG encoder conv1_1 .. conv4_3 (conv + batch-norm + ReLU, three VALID 2x2 pools), then
tf.image.resize_bilinear of conv4_3 to input_pixel_size / 8 * upsampling_multiplier (4):
(1,700,800,6) -> (1,350,400,256) and (1,480,1590,3) -> (1,240,795,256).  The 256 -> 1
bottleneck of the RPN (avod/core/models/rpn_model.py:251-267) is computed with the upsampling
when `with_bottleneck` is set.  Same calling conventions as vgg_pyramid.py."""
from dodt_amd import _lib
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr, _VggPyr


class _Vgg(_VggPyr):
    KIND = _lib.EXTRACTOR_VGG
    PAD_TOP = 0

    def __init__(self, extractor_config=None, ctx=None, shared_gpu=False, conv_dtype='f32'):
        if conv_dtype != 'f32':
            raise ValueError("the plain VGG extractors run in 'f32' only")
        super(_Vgg, self).__init__(extractor_config, ctx, shared_gpu, conv_dtype)


class BevVgg(_Vgg):
    preprocess_input = BevVggPyr.preprocess_input


class ImgVgg(_Vgg):
    _R_MEAN, _G_MEAN, _B_MEAN = ImgVggPyr._R_MEAN, ImgVggPyr._G_MEAN, ImgVggPyr._B_MEAN
    preprocess_input = ImgVggPyr.preprocess_input
