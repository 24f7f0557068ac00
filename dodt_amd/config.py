"""Frozen configurations of the hot path.

The reference reads these from protobuf text files; only the two configurations
the path is quoted on are kept, as plain dicts.  Proto `float` fields are
32-bit, and python then does float64 arithmetic on the float32-rounded values,
so they are stored here exactly as python sees them in the reference
(avod/protos/kitti_utils.proto:10,29-34).

  PYRAMID_DODT  avod/configs/pyramid_cars_with_aug_dt_5_tracking.config
  CARS_EXAMPLE  avod/configs/avod_cars_example.config
"""
import numpy as np


def _f32(x):
    return float(np.float32(x))


_COMMON = dict(
    area_extents=np.array([[-40., 40.], [-5., 3.], [0., 70.]]),
    bev_extents=np.array([[-40., 40.], [0., 70.]]),
    voxel_size=_f32(0.1),
    anchor_stride=[_f32(0.5), _f32(0.5)],
    height_lo=_f32(-0.2),
    height_hi=_f32(2.3),
    num_slices=5,
    # obj_utils.get_road_plane overwrites the plane file
    ground_plane=np.array([0., -1., 0., 1.65]),
    # slice used for the empty-anchor filter (kitti_utils.py:212-213)
    anchor_filter_lo=0.2,
    anchor_filter_hi=2.0,
    # The reference clusters car sizes from the training labels
    # (label_cluster_utils.py); those labels are not part of this build, so
    # the published AVOD car clusters stand in (l, w, h).
    anchor_sizes=[[3.514, 1.581, 1.511], [4.236, 1.653, 1.547]],
    bev_dims=(700, 800),
    bev_depth=6,
    rpn_roi_crop_size=3,
    avod_roi_crop_size=7,
    rpn_nms_iou_thresh=_f32(0.8),
    avod_nms_size=100,
    avod_nms_iou_thresh=_f32(0.01),
    rpn_train_nms_size=1024,
    rpn_test_nms_size=300,
    # avod_box_representation (both configs, e.g. ...dt_5_tracking.config:29): box_4c offsets
    # plus a regressed angle vector that fixes the heading (dt_evaluator.py:1166-1212)
    box_representation='box_4ca',
)

# frames_per_sample: 2 = DODT's Siamese model over a (t, t + tau) frame pair with the
# correlation branch (dt_rpn_model.py / dt_avod_model.py); 1 = single-frame AVOD
# (rpn_model.py / avod_model.py).  feat_stride / feat_depth: the extractor's output map
# relative to its input (pyramid: full resolution x 32; plain VGG: H/8*4 x 256).
PYRAMID_DODT = dict(_COMMON,
                    name='pyramid_cars_with_aug_dt_5_tracking',
                    model='dt_avod_model',
                    frames_per_sample=2,
                    extractor='vgg_pyr',
                    img_dims=(360, 1200),
                    img_depth=3)

CARS_EXAMPLE = dict(_COMMON,
                    name='avod_cars_example',
                    model='avod_model',
                    frames_per_sample=1,
                    extractor='vgg',
                    img_dims=(480, 1590),
                    img_depth=3)

# Calibration of the tracking sequence the reference's tests bundle
# (avod/tests/datasets/Kitti/tracking/training/calib/0000.txt, numbers only) and the KITTI
# image size: the default camera of the synthetic benchmarks and parity tests.
KITTI_P2 = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01],
                     [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
                     [0.0, 0.0, 1.0, 2.745884e-03]])
KITTI_R0_RECT = np.array([[9.999239e-01, 9.837760e-03, -7.445048e-03],
                          [-9.869795e-03, 9.999421e-01, -4.278459e-03],
                          [7.402527e-03, 4.351614e-03, 9.999631e-01]])
KITTI_TR_VELO_TO_CAM = np.array(
    [[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
     [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
     [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01]])
KITTI_IMAGE_WH = (1242, 375)


def velo_to_cam(r0_rect=KITTI_R0_RECT, tr=KITTI_TR_VELO_TO_CAM):
    """(3,4) = (R0_rect padded . Tr_velo_to_cam padded)[0:3], float64
    (wavedata/.../calib_utils.py:502-519)."""
    r0 = np.zeros((4, 4)); r0[:3, :3] = r0_rect; r0[3, 3] = 1
    t = np.zeros((4, 4)); t[:3, :4] = tr; t[3, 3] = 1
    return np.dot(r0, t)[:3]
