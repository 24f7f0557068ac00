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

PYRAMID_DODT = dict(_COMMON,
                    name='pyramid_cars_with_aug_dt_5_tracking',
                    extractor='vgg_pyr',
                    img_dims=(360, 1200),
                    img_depth=3)

CARS_EXAMPLE = dict(_COMMON,
                    name='avod_cars_example',
                    extractor='vgg',
                    img_dims=(480, 1590),
                    img_depth=3)
