"""Oracle for the whole frame path: raw points + image + head outputs ->
detections.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Chains the
restatements in the order of the reference's inference call stack
(dt_rpn_model.py:865-1042 then :355-730, dt_avod_model.py:128-711)."""
import numpy as np

from oracle import anchors as oanchors
from oracle import boxes as oboxes
from oracle import extractors as oext
from oracle import points as opoints
from oracle import tfops


def frame_inputs(xyzi, cfg, r0, tr, p2, image_wh, ego_motion=None):
    """The data half: BEV input, kept anchors and their projections.  ego_motion = (trans,
    matrix): the frame is the second one of a pair and is registered into the first frame's
    coordinates first (kitti_tracking_dataset.py:303-335) -- for the BEV maps only: the
    anchor filter's grid is built from the cloud as read from the file
    (kitti_tracking_utils.py:98-126)."""
    cloud = opoints.lidar_in_camera_view(xyzi, r0, tr, p2, image_wh)
    bev_cloud = cloud
    if ego_motion is not None:
        bev_cloud = opoints.lidar_in_camera_view(
            opoints.point_cloud_transform(xyzi, *ego_motion), r0, tr, p2, image_wh)
    bev = opoints.bev_input(bev_cloud, cfg['ground_plane'], cfg['area_extents'],
                            cfg['voxel_size'], cfg['height_lo'], cfg['height_hi'],
                            cfg['num_slices']).astype(np.float32)
    boxes = oanchors.tile_anchors_3d(cfg['area_extents'], cfg['anchor_sizes'],
                                     cfg['anchor_stride'], cfg['ground_plane'])
    anchors = oanchors.box_3d_to_anchor(boxes)
    vox = oanchors.sliced_voxel_grid_2d(cloud, cfg['ground_plane'], cfg['area_extents'],
                                        cfg['voxel_size'], cfg['anchor_filter_lo'],
                                        cfg['anchor_filter_hi'])
    keep = np.nonzero(oanchors.empty_anchor_filter_2d(anchors, vox))[0]
    kept = anchors[keep]
    _, bev_norm = oboxes.project_to_bev(kept, cfg['bev_extents'])
    _, img_norm = oboxes.project_to_image_space(kept, p2, [image_wh[1], image_wh[0]])
    return dict(bev=bev, keep=keep, anchors=kept.astype(np.float32),
                bev_norm_tf=bev_norm.astype(np.float32)[:, [1, 0, 3, 2]],
                img_norm_tf=img_norm[:, [1, 0, 3, 2]])


def frame_detections(inp, heads, cfg, p2, image_wh, rpn_nms_size, bev_feat=None,
                     img_feat=None, bev_bneck=None, img_bneck=None, frame_mark=0):
    """The graph half after the extractors, with the dense-head outputs given
    (heads: rpn_logits, rpn_offsets, cls_logits, offsets_4c [, angle_vectors: box_4ca]
    [, corr_offsets]), and the evaluator's record of the frame
    (dt_evaluator.py:1134-1259 via oracle.postprocess)."""
    A = len(inp['keep'])
    out = {}
    if bev_bneck is not None:
        out['rpn_bev_roi'] = tfops.crop_and_resize(bev_bneck, inp['bev_norm_tf'], 3, 3)
        out['rpn_img_roi'] = tfops.crop_and_resize(img_bneck, inp['img_norm_tf'], 3, 3)
    regressed = oboxes.offset_to_anchor(inp['anchors'], heads['rpn_offsets'][:A], np.float32)
    _, prop_norm = oboxes.project_to_bev(regressed, cfg['bev_extents'], np.float32)
    scores = tfops.softmax2(heads['rpn_logits'][:A])[:, 1]
    top = tfops.non_max_suppression_fast(prop_norm, scores, rpn_nms_size,
                                         cfg['rpn_nms_iou_thresh'])
    top_anchors = regressed[top]
    _, top_bev = oboxes.project_to_bev(top_anchors, cfg['bev_extents'], np.float32)
    _, top_img = oboxes.project_to_image_space(top_anchors, p2, [image_wh[1], image_wh[0]],
                                               dtype=np.float32)
    n = len(top)
    if bev_feat is not None:
        out['bev_rois'] = tfops.crop_and_resize(bev_feat, top_bev[:, [1, 0, 3, 2]], 7, 7)
        out['img_rois'] = tfops.crop_and_resize(img_feat, top_img[:, [1, 0, 3, 2]], 7, 7)
    b3 = oboxes.anchors_to_box_3d(top_anchors, fix_lw=True, dtype=np.float32)
    b4c = oboxes.box_3d_to_box_4c(b3, cfg['ground_plane'], np.float32)
    pred = oboxes.box_4c_to_box_3d(
        oboxes.offsets_to_box_4c(b4c, heads['offsets_4c'][:n]), cfg['ground_plane'], np.float32)
    pred_anchors = oboxes.box_3d_to_anchor_ortho(pred, np.float32)
    bev_m, _ = oboxes.project_to_bev(pred_anchors, cfg['bev_extents'], np.float32)
    s2 = heads['cls_logits'][:n, 1:].max(axis=1)
    det = tfops.non_max_suppression_fast(bev_m[:, [1, 0, 3, 2]], s2, cfg['avod_nms_size'],
                                         cfg['avod_nms_iou_thresh'])
    rec = np.zeros((cfg['avod_nms_size'], 17), np.float32)
    # dt_avod_model.py:547-548,631-634: orientations of all proposals, gathered by NMS #2;
    # dt_evaluator.py:1166-1257: box_4ca correction, score = max non-background softmax,
    # class index (0: one class), the corr-shifted box (frame 0 of the pair; zeros on
    # frame 1), frame mark
    from oracle import postprocess as opost
    final = pred[det].astype(np.float32)
    ori = None
    if heads.get('angle_vectors') is not None:
        ori = oboxes.angle_vector_to_orientation(heads['angle_vectors'][:n], np.float32)
        out['orientations'] = ori
        final = opost.orientation_corrected_boxes_3d(final, ori[det])
    rec[:len(det), :7] = final
    rec[:len(det), 7] = tfops.softmax2(heads['cls_logits'][:n])[det, 1]
    if heads.get('corr_offsets') is not None and frame_mark == 0:
        off = heads['corr_offsets'][:n][det].astype(np.float32)
        shifted = final.copy()
        shifted[:, 0] += off[:, 0]
        shifted[:, 2] += off[:, 1]
        shifted[:, 6] += off[:, 2]
        rec[:len(det), 9:16] = shifted
    rec[:len(det), 16] = frame_mark
    out.update(regressed=regressed, scores=scores, top_idx=top, top_anchors=top_anchors,
               boxes_3d=pred, pred_anchors=pred_anchors, nms2_boxes=bev_m[:, [1, 0, 3, 2]],
               det_idx=det, records=rec)
    return out


def pair_detections_computed(inps, feats, head_params, cfg, p2, image_wh, rpn_nms_size):
    """A frame pair with the dense heads computed by the oracle too (free-running: used as
    the CPU baseline and for smoke checks, not for index-exact parity -- see
    tests/test_gpu_heads.py).  inps / feats: per frame, from frame_inputs / extract."""
    from oracle import heads as oheads
    outs = []
    corr_map = tfops.correlation(feats[0][0], feats[1][0], 5, 2, 5)
    for f in range(2):
        inp = inps[f]
        bev_feat, img_feat, bev_bneck, img_bneck = feats[f]
        obj, off = oheads.rpn_anchor_predictor(
            tfops.crop_and_resize(bev_bneck, inp['bev_norm_tf'], 3, 3),
            tfops.crop_and_resize(img_bneck, inp['img_norm_tf'], 3, 3), head_params['rpn'])
        A = len(inp['keep'])
        regressed = oboxes.offset_to_anchor(inp['anchors'], off, np.float32)
        _, prop_norm = oboxes.project_to_bev(regressed, cfg['bev_extents'], np.float32)
        top = tfops.non_max_suppression_fast(prop_norm, tfops.softmax2(obj)[:, 1], rpn_nms_size,
                                             cfg['rpn_nms_iou_thresh'])
        top_anchors = regressed[top]
        _, top_bev = oboxes.project_to_bev(top_anchors, cfg['bev_extents'], np.float32)
        _, top_img = oboxes.project_to_image_space(top_anchors, p2, [image_wh[1], image_wh[0]],
                                                   dtype=np.float32)
        cls, o4c, ang = oheads.fusion_fc_early(
            tfops.crop_and_resize(bev_feat, top_bev[:, [1, 0, 3, 2]], 7, 7),
            tfops.crop_and_resize(img_feat, top_img[:, [1, 0, 3, 2]], 7, 7), head_params['avod'])
        heads = dict(rpn_logits=obj, rpn_offsets=off, cls_logits=cls, offsets_4c=o4c,
                     angle_vectors=ang)
        if f == 0:
            heads['corr_offsets'] = oheads.corr_fc_early(
                tfops.crop_and_resize(corr_map, top_bev[:, [1, 0, 3, 2]], 7, 7),
                head_params['corr'])
        outs.append(frame_detections(inp, heads, cfg, p2, image_wh, rpn_nms_size, frame_mark=f))
        assert A == len(obj)
    return outs


def extract(bev, img_u8, bev_params, img_params, img_hw, extractor='vgg_pyr'):
    """Both extractors + bottlenecks for one frame.  extractor 'vgg_pyr': the pyramid nets of
    the DODT config; 'vgg': the plain VGG nets of avod_cars_example (bev_vgg.py / img_vgg.py:
    4x-upsampled conv4_3, 256 channels, no top padding)."""
    pre = tfops.img_preprocess(img_u8, img_hw[0], img_hw[1])
    if extractor == 'vgg':
        bev_feat = oext.vgg_plain(bev, bev_params)
        img_feat = oext.vgg_plain(pre, img_params)
    else:
        bev_feat = oext.vgg_pyramid(bev, bev_params, pad_top=4)
        img_feat = oext.vgg_pyramid(pre, img_params, pad_top=0)
    return (bev_feat, img_feat, oext.bottleneck_1x1(bev_feat, bev_params['bottleneck']),
            oext.bottleneck_1x1(img_feat, img_params['bottleneck']))
