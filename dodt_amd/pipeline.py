"""Frame-pair pipeline: the per-frame hot path of DODT chained on one GPU.

Order of work follows the reference's inference call stack:
create_feed_dict (points -> BEV maps, anchor grid -> empty filter -> projections;
avod/core/models/dt_rpn_model.py:732-1042) then the graph
(dt_rpn_model.py:355-730, dt_avod_model.py:128-711).  The dense heads between
crop and NMS (anchor predictor, stage-2 FC, correlation branch: SURVEY 8(f) items 1-2)
run on the device when the pipeline is built with `head_params`; without them their
outputs are inputs of the pipeline (`heads`, resident in HBM), which is what the
index-exact parity tests use.

Frame pairs are independent (batch size 1 in the reference, no cross-pair state),
so a step may carry several pairs: all their frames go through the conv stacks as
one batch.  Streams: the two conv stacks each have their own (they run side by
side); every frame has a side stream for its "prep" (points -> BEV maps, anchors, image
preprocessing) and its "tail" (crops, heads, decode, NMS), so the single-workgroup
stages (NMS scan) of different frames overlap (four streams in all at one pair per
step: the hardware-queue budget of a process).  A frame's prep and tail share its side
stream, so the order on that stream is prep k, tail k-1, prep k+1, ...: the prep of step
k+1 is enqueued by run(k+1) behind the tail of step k-1 and in front of the convs of its own
step (0.1 ms; it does NOT run under the convs of step k); what overlaps is the tail of step
k with the convs of step k+1.  Inputs, feature maps, per-frame buffers and detection records
are double-buffered by step parity for that, and `finish()` drains the last step.
Everything stays on the device; the only host round trip per step is the kept-anchor count
of each frame, fetched one step after it was produced (the host runs one step ahead of the
GPU and otherwise waits in that read).
"""
import os

import numpy as np

from dodt_amd import config as _config
from dodt_amd import device, ops
from dodt_amd.core.anchor_generators import grid_anchor_3d_generator as gen
from dodt_amd.core.avod_fc_layers.fusion_fc_layers import EarlyFusionFcLayers
from dodt_amd.core.feature_extractors.vgg import BevVgg, ImgVgg
from dodt_amd.core.feature_extractors.vgg_pyramid import BevVggPyr, ImgVggPyr
from dodt_amd.core.models.anchor_predictor import AnchorPredictor

# feature_extractor_builder.get_extractor (avod/builders/feature_extractor_builder.py:8-24)
EXTRACTORS = {'vgg_pyr': (BevVggPyr, ImgVggPyr), 'vgg': (BevVgg, ImgVgg)}

MAX_DET = 100            # avod_nms_size
REC_COLS = 17            # dt_evaluator.py:1217-1257
ROI = 7                  # avod_proposal_roi_crop_size
CORR_MAX_DISP, CORR_STRIDE2, CORR_PAD = 5, 2, 5     # correlation_config; correlation.py:7
CORR_CH = (2 * (CORR_MAX_DISP // CORR_STRIDE2) + 1) ** 2


class FramePairPipeline(object):
    """One GPU's pipeline over samples of the configuration: frame pairs for DODT
    (cfg['frames_per_sample'] == 2: Siamese extractors + correlation branch,
    dt_rpn_model.py / dt_avod_model.py) or single frames for plain AVOD
    (frames_per_sample == 1, e.g. config.CARS_EXAMPLE: rpn_model.py / avod_model.py, no
    correlation branch; `pairs_per_step` then counts frames).  The extractor pair follows
    cfg['extractor'] ('vgg_pyr' or 'vgg')."""

    def __init__(self, ctx, cfg, bev_params, img_params, p2=_config.KITTI_P2,
                 r0_rect=_config.KITTI_R0_RECT, tr_velo_to_cam=_config.KITTI_TR_VELO_TO_CAM,
                 image_wh=_config.KITTI_IMAGE_WH, n_points_max=120000, rpn_nms_size=1024,
                 pairs_per_step=1, side_streams=None, head_params=None, conv_dtype='f32',
                 head_dtype='f32', reuse_streams_of=None, tail_sets=None):
        self.ctx = ctx
        self.cfg = cfg
        self.p2 = np.asarray(p2, dtype=np.float64)
        self.image_wh = tuple(image_wh)
        self.P = int(rpn_nms_size)
        self.n_points_max = int(n_points_max)
        self.pairs = int(pairs_per_step)               # samples per step
        self.fps = int(cfg.get('frames_per_sample', 2))
        if self.fps not in (1, 2):
            raise ValueError('frames_per_sample must be 1 or 2')
        self.nf = self.fps * self.pairs                # frames per step
        self.bev_h, self.bev_w = cfg['bev_dims']
        self.img_h, self.img_w = cfg['img_dims']
        self.n_slices = cfg['num_slices']
        self.bev_extents_flat = np.asarray(cfg['bev_extents'], np.float64).reshape(-1)
        self.bp = ops.make_bev_params(cfg, _config.velo_to_cam(r0_rect, tr_velo_to_cam),
                                      self.p2, self.image_wh)
        # streams: conv stacks of the two nets side by side, per-frame work on its own.
        # A second pipeline in the same process takes the first one's streams
        # (`reuse_streams_of`): new ones would share hardware queues with them (see below).
        # `tail_sets` (1, the default, or 2; DODT_PIPE_TAIL_SETS overrides): sets of side streams.  With one set a
        # frame's stream carries prep k, tail k-1, prep k+1, tail k, ... in a row.  With two sets the steps alternate
        # between them by parity (set p: prep k, tail k, prep k+2, tail k+2, ...): the tails of consecutive steps
        # may overlap, every buffer a step touches is still used in stream order by its own parity's streams.
        # Built in round 4 for the bf16 path, whose step looked bound by a tail's dependent launch chain, and
        # measured: SLOWER in every mode (fp32 324 -> 276 pairs/s, bf16 convs + heads 836 -> 522: six streams
        # instead of four) -- opt-in only.
        if tail_sets is None:
            tail_sets = 1
        tail_sets = int(os.environ.get('DODT_PIPE_TAIL_SETS', tail_sets))
        if tail_sets not in (1, 2):
            raise ValueError('tail_sets must be 1 or 2')
        if reuse_streams_of is not None:
            self.img_ctx = reuse_streams_of.img_ctx
            self.stream_sets = list(reuse_streams_of.stream_sets)
            while len(self.stream_sets) < tail_sets:
                self.stream_sets.append(self._make_side_streams(ctx, side_streams))
            self.stream_sets = self.stream_sets[:tail_sets]
        else:
            self.img_ctx = device.Context(ctx.device_id)
            self.stream_sets = [self._make_side_streams(ctx, side_streams) for _ in range(tail_sets)]
        self.sides, self.preps = self.stream_sets[0]      # (the first set: what one-set callers see)
        # ---- constants of the configuration, resident on the device ----------------
        boxes = gen.tile_anchors_3d(cfg['area_extents'], cfg['anchor_sizes'],
                                    cfg['anchor_stride'], cfg['ground_plane'])
        self.anchors_all = gen.box_3d_to_anchor(boxes)            # (N,6) float64
        cells, self.nx, self.nz = gen.anchor_grid_cells(
            self.anchors_all, cfg['area_extents'], cfg['voxel_size'])
        self.n_all = len(self.anchors_all)
        self.d_anchor_table = ctx.array(self.anchors_all)
        self.d_cells = ctx.array(cells)

        # ---- extractors: every frame of the step is one batch ------------------------
        bev_cls, img_cls = EXTRACTORS[cfg.get('extractor', 'vgg_pyr')]
        self.bev_net = bev_cls(ctx=ctx, shared_gpu=True, conv_dtype=conv_dtype)
        self.bev_net.load_params(bev_params)
        self.bev_net._ensure(self.nf, self.bev_h, self.bev_w, cfg['bev_depth'])
        self.img_net = img_cls(ctx=self.img_ctx, shared_gpu=True, conv_dtype=conv_dtype)
        self.img_net.load_params(img_params)
        self.img_net._ensure(self.nf, self.img_h, self.img_w, 4)
        # feature maps the crops read: (700,800,32) / (360,1200,32) for the pyramid,
        # (350,400,256) / (240,795,256) for the plain VGG
        self.bev_fh, self.bev_fw, self.feat_c = self.bev_net.output_shape()
        self.img_fh, self.img_fw, img_c = self.img_net.output_shape()
        if img_c != self.feat_c:
            raise ValueError('mean fusion needs equal feature depths')
        # conv inputs, double-buffered so that step k+1 is prepared under the convs of step k
        # (in the extractors' own input layout -- the BEV maps behind four zero rows, bev_vgg_pyramid.py:58 --, so
        #  that the first conv layer reads them in place: forward_device_padded)
        self.bev_pad = self.bev_net.PAD_TOP
        self.in_bev = [ctx.zeros((self.nf, self.bev_pad + self.bev_h, self.bev_w, cfg['bev_depth']), np.float32)
                       for _ in range(2)]
        self.in_img = [ctx.zeros((self.nf, self.img_net.PAD_TOP + self.img_h, self.img_w, 4), np.float32)
                       for _ in range(2)]
        self.d_bev_in = self._views(self.in_bev[0], (self.bev_h, self.bev_w, cfg['bev_depth']), self.bev_pad)

        # ---- dense heads (weights shared, scratch per side stream) ------------------------
        f32, i32 = np.float32, np.int32
        N, P = self.n_all, self.P
        self.rpn_head = self.avod_head = self.corr_head = None
        rep = cfg.get('box_representation', 'box_4ca')
        if rep not in ('box_4c', 'box_4ca'):
            raise NotImplementedError('Regression not implemented for %s' % rep)
        self.box_4ca = rep == 'box_4ca'
        if head_params is not None:
            self.rpn_head = AnchorPredictor(ctx, head_params['rpn'], dtype=head_dtype)
            self.avod_head = EarlyFusionFcLayers(
                ctx, head_params['avod'], dtype=head_dtype,
                outputs=('cls_out', 'off_out') + (('ang_out',) if self.box_4ca else ()))
            if self.fps == 2:
                self.corr_head = EarlyFusionFcLayers(ctx, head_params['corr'],
                                                     outputs=('off_out',), dtype=head_dtype)
            # one scratch set per side stream of every set (the tails of two sets overlap)
            self.head_scratch_sets = [[dict(rpn=self.rpn_head.make_scratch(N),
                                            fc=self.avod_head.make_scratch(P),
                                            corr_map=ctx.empty((self.bev_fh, self.bev_fw, CORR_CH), f32)
                                            if self.fps == 2 else None)
                                       for _ in sides] for sides, _ in self.stream_sets]
            self.head_scratch = self.head_scratch_sets[0]
            # the pairs' correlation maps, by step parity (written behind the image stack on its stream, see run())
            self.corr_maps = [[ctx.empty((self.bev_fh, self.bev_fw, CORR_CH), f32) for _ in range(self.pairs)]
                              for _ in range(2)] if self.fps == 2 else None

        # ---- work buffers ----------------------------------------------------------------
        FC = self.feat_c
        self.feat = [dict(
            bev_feat=ctx.empty((self.nf, self.bev_fh, self.bev_fw, FC), f32),
            bev_bneck=ctx.empty((self.nf, self.bev_fh, self.bev_fw, 1), f32),
            img_feat=ctx.empty((self.nf, self.img_fh, self.img_fw, FC), f32),
            img_bneck=ctx.empty((self.nf, self.img_fh, self.img_fw, 1), f32)) for _ in range(2)]
        # what a step's prep leaves for its tail, THREE deep (step k: set k % 3): with look-ahead (run(...,
        # lookahead=)) the prep of step k + 1 is enqueued in front of the tail of step k - 1, which still reads
        # the set of its own step
        self.prep3 = [[dict(occ=ctx.empty((self.nz, (self.nx + 31) // 32), np.uint32),
                            keep=ctx.empty((N,), i32), count=ctx.zeros((1,), i32),
                            bev_norm=ctx.empty((N, 4), f32), img_norm=ctx.empty((N, 4), f32),
                            anchors=ctx.empty((N, 6), f32)) for _ in range(self.nf)] for _ in range(3)]
        if 3 * self.nf > 32:
            raise ValueError('at most 10 frames per step (count fetch slots)')
        self.fr2 = [[], []]
        for f in range(2 * self.nf):
            b = dict(
                rpn_bev_roi=ctx.empty((N, 3, 3, 1), f32), rpn_img_roi=ctx.empty((N, 3, 3, 1), f32),
                regressed=ctx.empty((N, 6), f32), prop_bev=ctx.empty((N, 4), f32),
                scores=ctx.empty((N,), f32),
                top_idx=ctx.empty((P,), i32), top_count=ctx.zeros((1,), i32),
                top_anchors=ctx.empty((P, 6), f32),
                top_bev=ctx.empty((P, 4), f32), top_img=ctx.empty((P, 4), f32),
                bev_rois=ctx.empty((P, ROI, ROI, FC), f32),
                img_rois=ctx.empty((P, ROI, ROI, FC), f32),
                boxes_3d=ctx.empty((P, 7), f32), pred_anchors=ctx.empty((P, 6), f32),
                nms2_boxes=ctx.empty((P, 4), f32), nms2_scores=ctx.empty((P,), f32),
                det_idx=ctx.empty((MAX_DET,), i32), det_count=ctx.zeros((1,), i32),
                det_scores=ctx.empty((P,), f32), orientations=ctx.empty((P,), f32))
            if head_params is not None:
                b.update(rpn_logits=ctx.empty((N, 2), f32), rpn_offsets=ctx.empty((N, 6), f32),
                         cls_logits=ctx.empty((P, 2), f32), offsets_4c=ctx.empty((P, 10), f32))
                if self.box_4ca:
                    b.update(angle_vectors=ctx.empty((P, 2), f32))
                if self.fps == 2 and f % 2 == 0:
                    # rows of the correlation head's padded K (zeros behind each 7x7x25 crop)
                    b.update(corr_rois=ctx.zeros((P, self.corr_head.in_ld), f32),
                             corr_offsets=ctx.empty((P, 3), f32))
            self.fr2[f // self.nf].append(b)
        self.fr = [dict(b, **p) for b, p in zip(self.fr2[0], self.prep3[0])]   # buffers of the most recently finished step
        self.prepped = -1              # step whose prep a look-ahead has already enqueued
        self.step_idx = 0
        self.pending = None            # step whose tail has not been enqueued yet
        # detection records of a step: what the all-gather ships.  A ring of R >= 2
        # buffers, step k fills slot k % R (R = 2: by step parity); use_record_ring() makes it
        # longer so that the exchange step can ship several steps at once
        self.rec2 = [ctx.empty((self.pairs, self.fps, MAX_DET, REC_COLS), f32) for _ in range(2)]
        self.cnt2 = [ctx.zeros((self.pairs, self.fps), i32) for _ in range(2)]
        self.d_records, self.d_rec_counts = self.rec2[0], self.cnt2[0]   # last finished step
        self.last_anchor_counts = [0] * self.nf
        # hook(slot, side contexts), called before a tail refills record slot `slot`: the exchange
        # step (sharding.Communicator.join) makes the tails wait for the all-gather that last
        # read it, at least a ring's half earlier
        self.on_records_reuse = None
        self.mark_steps = ()           # tools/pipe_marks.py: steps whose stages get timing marks
        self.early_prep = os.environ.get('DODT_PIPE_EARLY_PREP', '1') != '0'
        self.marks = {}                # name -> (context, slot)
        ctx.sync()

    def _mark(self, c, step, name):
        """Timing mark `name` of step `step` on context c (only for steps in mark_steps)."""
        if step in self.mark_steps:
            slot = sum(1 for (cc, _) in self.marks.values() if cc is c)
            if slot < 64:
                c.mark(slot)
                self.marks['%d:%s' % (step, name)] = (c, slot)

    def _make_side_streams(self, ctx, side_streams):
        """One set of side streams: (tails, preps), one of each per frame of a pair (or `side_streams`)."""
        n_side = min(self.nf, 2) if side_streams is None else int(side_streams)
        hp = os.environ.get('DODT_PIPE_PRIO', '0') == '1'
        sides = [device.Context(ctx.device_id, high_priority=hp)
                 for _ in range(max(n_side, 1))]   # tails
        # ROCm maps a process's streams onto 4 hardware queues; a fifth stream shares a queue
        # with another one and runs behind its launches (measured: 6 streams 164, 5 streams
        # 180, 4 streams 191 pairs/s), so by default a frame's prep and tail share a stream
        mode = os.environ.get('DODT_PIPE_STREAMS', 'shared')
        if mode == 'shared':      # frame f's prep and tail on one stream
            preps = sides
        elif mode == 'conv':      # (round 4, late) the preps on the two conv streams: frame 0's behind the BEV stack, frame 1's
            # behind the image stack -- with look-ahead they are the NEXT step's, which those stacks' successors wait for
            # anyway, and a frame's side stream carries its tails only (no new stream).  Measured: 996 against 1 015 pairs/s
            # (bf16, same box) -- the stacks, not the side streams, then carry the preps' 0.2 ms; opt-in
            preps = [ctx if i % 2 == 0 else self.img_ctx for i in range(len(sides))]
        elif mode == 'one':       # all preps on one extra stream
            one = device.Context(ctx.device_id, high_priority=hp)
            preps = [one for _ in sides]
        else:
            preps = [device.Context(ctx.device_id, high_priority=hp)
                     for _ in range(max(n_side, 1))]
        return sides, preps

    def _streams(self, cur):
        """(tail streams, prep streams) of the steps with parity `cur`."""
        return self.stream_sets[cur % len(self.stream_sets)]

    def _views(self, arr, shape, pad_top=0):
        """Per-frame views of a batch buffer whose frames are pad_top + shape[0] rows tall: the rows behind the pad."""
        row = int(np.prod(shape[1:])) * 4
        frame = (pad_top + shape[0]) * row
        return [arr.offset(frame * f + pad_top * row, shape) for f in range(self.nf)]

    def use_record_buffers(self, rec_ptrs, cnt_ptrs):
        """Write detection records into caller-owned device memory (e.g. buffers registered with
        a communication library): R >= 2 of each, step k fills number k % R."""
        if len(rec_ptrs) < 2 or len(rec_ptrs) != len(cnt_ptrs):
            raise ValueError('use_record_buffers: at least two record and count buffers, as many of each')
        self.rec2 = [self.ctx.wrap(p, (self.pairs, self.fps, MAX_DET, REC_COLS), np.float32)
                     for p in rec_ptrs]
        self.cnt2 = [self.ctx.wrap(p, (self.pairs, self.fps), np.int32) for p in cnt_ptrs]
        self.d_records, self.d_rec_counts = self.rec2[0], self.cnt2[0]

    def use_record_ring(self, d_rec_ring, d_cnt_ring):
        """The same with one contiguous ring: d_rec_ring (R, pairs, fps, MAX_DET, REC_COLS) float32,
        d_cnt_ring (R, pairs, fps) int32 -- so that any run of consecutive slots is one message."""
        R = d_rec_ring.shape[0]
        if tuple(d_rec_ring.shape[1:]) != (self.pairs, self.fps, MAX_DET, REC_COLS) or \
                tuple(d_cnt_ring.shape) != (R, self.pairs, self.fps):
            raise ValueError('use_record_ring: shapes do not match the pipeline')
        nr, nc = 4 * self.pairs * self.fps * MAX_DET * REC_COLS, 4 * self.pairs * self.fps
        self.use_record_buffers([d_rec_ring.ptr + nr * i for i in range(R)],
                                [d_cnt_ring.ptr + nc * i for i in range(R)])

    # ------------------------------------------------------------------------------------
    def _stage_from_host(self, k, h_points, n_points, h_images):
        """Enqueue the copies of step k's raw frames from pinned host memory on its prep streams."""
        cur = k & 1
        _, preps = self._streams(cur)
        ns = len(preps)
        if not hasattr(self, 'stage'):
            H, W = self.image_wh[1], self.image_wh[0]
            self.stage = [[(self.ctx.empty((self.n_points_max, 4), np.float32),
                            self.ctx.empty((H, W, 3), np.uint8)) for _ in range(self.nf)]
                          for _ in range(2)]
        d_pts, d_imgs = [], []
        for f in range(self.nf):
            c = preps[f % ns]
            dp, di = self.stage[cur][f]
            if n_points[f] > self.n_points_max:
                raise ValueError('frame %d has more than n_points_max points' % f)
            dp.upload_async(h_points[f], ctx=c, nbytes=16 * int(n_points[f]))
            di.upload_async(h_images[f], ctx=c)
            d_pts.append(dp)
            d_imgs.append(di)
        return d_pts, d_imgs

    def run_from_host(self, h_points, n_points, h_images, heads=None, ego_motion=None, lookahead=None):
        """run() for raw frames still in (page-locked) host memory: lists of PinnedArray --
        points (n_max,4) float32 of which n_points[f] rows are valid, images (H,W,3) uint8.
        The copies are enqueued on each frame's prep stream in front of its prep kernels, so
        they travel under the kernels of the previous step; the host does not wait for them
        (the caller keeps the pinned buffers untouched until that step's prep has run, e.g.
        by alternating two sets)."""
        if self.prepped == self.step_idx:       # staged and prepared by the previous call's look-ahead
            d_pts = d_imgs = None
        else:
            d_pts, d_imgs = self._stage_from_host(self.step_idx, h_points, n_points, h_images)
        if lookahead is not None:
            # (the copies of step k + 1 go behind what its prep streams hold now, i.e. behind step k's prep)
            la = tuple(lookahead) + (None,) * (4 - len(lookahead))
            nd_pts, nd_imgs = self._stage_from_host(self.step_idx + 1, la[0], la[1], la[2])
            lookahead = (nd_pts, la[1], nd_imgs, la[3])
        return self.run(d_pts, n_points, d_imgs, heads, ego_motion, lookahead)

    PREP_DONE_MARK = 244        # mark slots 244..246 of the prep contexts: end of a step's prep, by step % 3

    def _prep(self, k, d_points, n_points, d_images, ego_motion, ahead):
        """a0-a7 of step k: the data side of the reference's create_feed_dict, one frame per prep stream; its end
        is marked on those streams (PREP_DONE_MARK + k % 3).  `ahead`: enqueued by the previous step's run()
        (look-ahead): the conv inputs of this parity were last read by the convs of step k - 2, which the prep
        streams are told to wait for (with the usual order, behind the tail of step k - 2, that is implied)."""
        nf = self.nf
        mean = (self.img_net._R_MEAN, self.img_net._G_MEAN, self.img_net._B_MEAN)
        cur = k & 1
        sides, preps = self._streams(cur)
        ns = len(sides)
        bev_in = self._views(self.in_bev[cur], (self.bev_h, self.bev_w, self.cfg['bev_depth']), self.bev_pad)
        img_in = self._views(self.in_img[cur], (self.img_h, self.img_w, 4), self.img_net.PAD_TOP)
        if ahead and k >= 2:
            for c in set(preps):
                c.wait_mark(self.ctx, self.CONV_DONE_MARK + cur)
                c.wait_mark(self.img_ctx, self.CONV_DONE_MARK + cur)
        for f in range(nf):
            c, b = preps[f % ns], self.prep3[k % 3][f]
            self._mark(c, k, 'prep%d_start' % f)
            bp = self.bp
            if ego_motion is not None and self.fps == 2 and f % 2 == 1 \
                    and ego_motion[f // 2] is not None:
                bp = ops.with_ego_motion(self.bp, *ego_motion[f // 2])
            ops.bev_slices(c, d_points[f], n_points[f], bp, bev_in[f], b['occ'])
            ops.anchor_filter(c, b['occ'], self.nx, self.nz, self.d_cells, self.n_all,
                              b['keep'], b['count'])
            ops.fetch_i32_begin(c, b['count'], 1, 3 * f + k % 3)
            ops.project_anchors_f64(c, self.d_anchor_table, b['keep'], self.n_all, b['count'],
                                    self.bev_extents_flat, self.p2, self.image_wh,
                                    b['bev_norm'], b['img_norm'], b['anchors'])
            ops.img_preprocess(c, d_images[f], (self.image_wh[1], self.image_wh[0]),
                               (self.img_h, self.img_w), 4, mean, img_in[f])
            self._mark(c, k, 'prep%d_end' % f)
        for c in set(preps):
            c.mark(self.PREP_DONE_MARK + k % 3)
        self.prepped = k

    def run(self, d_points, n_points, d_images, heads=None, ego_motion=None, lookahead=None):
        """Enqueue one step.  Lists of length 2 * pairs_per_step, frame order
        [pair0 f0, pair0 f1, pair1 f0, ...]: d_points[f] (n,4) float32 velodyne xyzi;
        d_images[f] (H,W,3) uint8; heads[f] dict of device arrays rpn_logits (N,2),
        rpn_offsets (N,6), cls_logits (P,2), offsets_4c (P,10), angle_vectors (P,2) (box_4ca)
        [, corr_offsets (P,3) on frame 0 of a pair]; None when the pipeline computes the heads
        itself (head_params).
        ego_motion: None, or one (trans (3,), matrix (3,3)) per pair of the step -- the
        registration of the pair's second frame into the first frame's coordinates
        (datasets.kitti.kitti_tracking_utils.coordinate_transform; applied to the second
        frame's BEV maps, not to its anchor-filter grid, like the reference).
        lookahead: None, or (d_points, n_points, d_images[, ego_motion]) of the NEXT step: its prep is enqueued now,
        in front of the previous step's tail on the side streams, so that the next step's convs do not wait for
        that tail (a caller that knows its next inputs -- a stream of frames -- should pass them; the next call
        must then be made with those inputs).
        Returns the parity (0/1) of the record buffers this step will fill.  The
        detections of the PREVIOUS step are complete on the main stream when this returns
        (self.d_records / self.fr / self.last_anchor_counts then describe that step);
        call finish() after the last step (run(); finish() is the unpipelined form)."""
        main, nf = self.ctx, self.nf
        if (heads is None) != (self.rpn_head is not None):
            raise ValueError('pass `heads` exactly when the pipeline has no head_params')
        cur = self.step_idx & 1
        sides, preps = self._streams(cur)
        ns = len(sides)
        fr, feat = self.fr2[cur], self.feat[cur]
        k = self.step_idx
        if self.prepped != k:      # (else: enqueued by the previous call's look-ahead)
            self._prep(k, d_points, n_points, d_images, ego_motion, ahead=False)
        for c in set(preps):
            main.wait_mark(c, self.PREP_DONE_MARK + k % 3)
            self.img_ctx.wait_mark(c, self.PREP_DONE_MARK + k % 3)
        # -- a8-a10: conv stacks, all frames per launch, the two nets side by side --------
        self._mark(main, k, 'bev_start')
        self._mark(self.img_ctx, k, 'img_start')
        self.bev_net.forward_device_padded(self.in_bev[cur], feat['bev_feat'], feat['bev_bneck'])
        self.img_net.forward_device_padded(self.in_img[cur], feat['img_feat'], feat['img_bneck'])
        self._mark(main, k, 'bev_end')
        self._mark(self.img_ctx, k, 'img_end')
        # The tail of THIS step (next call) starts when these convs are done.  The point is marked now and waited
        # for when the tail is enqueued -- behind the NEXT step's prep on the same side stream, which therefore
        # runs under these convs instead of behind them (DODT_PIPE_EARLY_PREP=0: the wait goes in at the end of this
        # call, in front of that prep, as before round 3)
        main.mark(self.CONV_DONE_MARK + cur)
        self.img_ctx.mark(self.CONV_DONE_MARK + cur)
        # The T branch's correlation map needs the two frames' BEV maps and nothing else: it runs HERE, behind the image
        # stack on its stream (the shorter of the two stacks), not inside a frame's tail, whose dependent launch chain
        # -- with its prep what bounds the bf16 step -- it made 50 us longer (DODT_PIPE_CORR_MAP=f1: round 4's first form,
        # map and crops on frame 1's stream between its crops and its head, the correlation head on frame 0's)
        if self._corr_on_img():
            self.img_ctx.wait_mark(main, self.CONV_DONE_MARK + cur)
            bev_hw, px = (self.bev_fh, self.bev_fw), self.bev_fh * self.bev_fw
            for pair in range(self.pairs):
                fb0 = feat['bev_feat'].offset(4 * px * self.feat_c * (2 * pair), bev_hw + (self.feat_c,))
                fb1 = feat['bev_feat'].offset(4 * px * self.feat_c * (2 * pair + 1), bev_hw + (self.feat_c,))
                ops.correlation(self.img_ctx, fb0, fb1, bev_hw + (self.feat_c,), CORR_MAX_DISP, CORR_STRIDE2, CORR_PAD,
                                self.corr_maps[cur][pair])
            self.img_ctx.mark(self.CORR_MAP_MARK + cur)
        # -- the next step's prep, when the caller has given its inputs: in front of the previous step's tail ----
        if lookahead is not None:
            la = tuple(lookahead) + (None,) * (4 - len(lookahead))
            self._prep(k + 1, la[0], la[1], la[2], la[3], ahead=True)
        # -- the previous step's tail runs under this step's convs --------------------------
        if self.pending is not None:
            self._wait_convs(self.pending)
            self._tail(self.pending)
            p_sides, p_preps = self._streams(self.pending['cur'])
            for i, s in enumerate(p_sides):
                # one event at the tail's end, three waiters: `main` (the previous step's records are complete there, and
                # the BEV stack of the next step of that parity overwrites the maps the tail reads), the image stream (the
                # same for the image net's maps: without look-ahead that wait is implied -- the next prep sits behind the
                # tail on the side stream and the conv streams wait for the prep --, with look-ahead the prep is in front
                # of it), and a prep stream of its own, if any.  (A mark after the tail's LAST READ of the image maps
                # instead -- its stage-2 crops -- was measured: the extra event inside the tail's launch chain costs more
                # than the earlier release returns, 868 against 899 pairs/s with the bf16 path; DODT_PIPE_IMG_WAIT=none is
                # the racy form the first look-ahead build had, 907.)
                s.mark(self.TAIL_DONE_MARK)
                main.wait_mark(s, self.TAIL_DONE_MARK)
                if os.environ.get('DODT_PIPE_IMG_WAIT', 'tail') != 'none':
                    self.img_ctx.wait_mark(s, self.TAIL_DONE_MARK)
                if p_preps[i] is not s:
                    p_preps[i].wait_mark(s, self.TAIL_DONE_MARK)
        self.pending = dict(cur=cur, heads=heads, step=k, rslot=k % len(self.rec2))
        if not self.early_prep:
            for s in sides:
                s.wait_for(main)
                s.wait_for(self.img_ctx)
        self.step_idx += 1
        return cur

    def finish(self):
        """Enqueue the tail of the last step; afterwards self.fr / d_records hold it."""
        if self.pending is not None:
            self._wait_convs(self.pending)
            self._tail(self.pending)
            self.pending = None
        for sides, preps in self.stream_sets:
            for s in set(sides) | set(preps):
                self.ctx.wait_for(s)
        self.ctx.wait_for(self.img_ctx)

    CONV_DONE_MARK = 250        # mark slots 250, 251 of the conv contexts: end of a step's stacks, by parity
    CORR_MAP_MARK = 248         # ... 248, 249 of the image context: the step's correlation maps stand, by parity

    def _corr_on_img(self):
        """The T branch's correlation runs behind the image stack (run()) instead of inside a frame's tail."""
        return (self.rpn_head is not None and self.fps == 2 and len(self.sides) >= 2
                and not os.environ.get('DODT_PIPE_NO_CORR') and os.environ.get('DODT_PIPE_CORR_ON_F1', '1') != '0'
                and os.environ.get('DODT_PIPE_CORR_MAP', 'img') == 'img')

    def _wait_convs(self, st):
        """The side streams wait for the conv stacks of step `st` (marked at the end of its run())."""
        if self.early_prep:
            for s in self._streams(st['cur'])[0]:
                s.wait_mark(self.ctx, self.CONV_DONE_MARK + st['cur'])
                s.wait_mark(self.img_ctx, self.CONV_DONE_MARK + st['cur'])

    def _tail(self, st):
        """Stages after the extractors for every frame of step `st` (a11-a14)."""
        cfg, nf = self.cfg, self.nf
        cur = st['cur']
        sides, preps = self._streams(cur)
        ns = len(sides)
        head_scratch = self.head_scratch_sets[cur % len(self.stream_sets)] if self.rpn_head is not None else None
        fr, feat = self.fr2[cur], self.feat[cur]
        heads = st['heads']
        # kept-anchor counts of that step: fetched by its prep streams, long complete
        k3 = st['step'] % 3
        counts = [ops.fetch_i32_end(preps[f % ns], 3 * f + k3, 1)[0]
                  for f in range(nf)]
        self.last_anchor_counts = counts
        fr = [dict(b, **p) for b, p in zip(fr, self.prep3[k3])]      # the tail's buffers + what its prep left
        self.fr = fr
        self.d_records, self.d_rec_counts = self.rec2[st['rslot']], self.cnt2[st['rslot']]
        self.d_bev_in = self._views(self.in_bev[cur], (self.bev_h, self.bev_w, self.cfg['bev_depth']), self.bev_pad)
        bev_px = self.bev_fh * self.bev_fw
        img_px = self.img_fh * self.img_fw
        FC = self.feat_c
        bev_hw, img_hw = (self.bev_fh, self.bev_fw), (self.img_fh, self.img_fw)
        plane = cfg['ground_plane']
        if os.environ.get('DODT_PIPE_NO_TAIL'):      # (tools/: the step without its tail)
            return
        if self.on_records_reuse is not None:
            self.on_records_reuse(st['rslot'], sides)
        computed = heads is None
        # The T branch of a pair hangs on frame 0's tail, which makes it half as long again as frame 1's.  Its map
        # and crops (not the head) therefore go onto frame 1's stream, between that frame's own crops and head:
        # frame 0's stream marks the point where its proposals stand (slot PROPOSALS_MARK), frame 1's waits for it,
        # correlates, crops and marks CORR_ROIS_MARK, which frame 0's stream waits for in front of the correlation
        # head.  The frames' launches are enqueued in that order: frame 0 up to its head, frame 1 whole, frame
        # 0's rest (DODT_PIPE_CORR_ON_F1=0: all of the branch on frame 0's stream).
        split_t = computed and self.fps == 2 and ns >= 2 and not os.environ.get('DODT_PIPE_NO_CORR') \
            and os.environ.get('DODT_PIPE_CORR_ON_F1', '1') != '0'
        corr_img = split_t and self._corr_on_img()
        fused_tail = os.environ.get('DODT_PIPE_FUSED_TAIL', '1') != '0'

        def t_branch_crops(cc, f0, scratch):
            """Correlation map of pair (f0, f0 + 1) and its 7x7 crops at frame f0's proposals, on context cc."""
            fb0 = feat['bev_feat'].offset(4 * bev_px * FC * f0, bev_hw + (FC,))
            fb1 = feat['bev_feat'].offset(4 * bev_px * FC * (f0 + 1), bev_hw + (FC,))
            ops.correlation(cc, fb0, fb1, bev_hw + (FC,), CORR_MAX_DISP, CORR_STRIDE2, CORR_PAD,
                            scratch['corr_map'])
            ops.crop_and_resize(cc, scratch['corr_map'], bev_hw + (CORR_CH,), fr[f0]['top_bev'], self.P,
                                fr[f0]['top_count'], (ROI, ROI), fr[f0]['corr_rois'],
                                out_box_stride=self.corr_head.in_ld)

        def frame(f):
            c, b, A = sides[f % ns], fr[f], counts[f]
            h = b if computed else heads[f]
            scratch = head_scratch[f % ns] if computed else None
            self._mark(c, st['step'], 'tail%d_start' % f)
            bneck_b = feat['bev_bneck'].offset(4 * bev_px * f, bev_hw + (1,))
            bneck_i = feat['img_bneck'].offset(4 * img_px * f, img_hw + (1,))
            feat_b = feat['bev_feat'].offset(4 * bev_px * FC * f, bev_hw + (FC,))
            feat_i = feat['img_feat'].offset(4 * img_px * FC * f, img_hw + (FC,))
            # -- a11: RPN crops (3x3 on the 1-channel bottlenecks) ------------------------
            ops.crop_and_resize(c, bneck_b, bev_hw + (1,), b['bev_norm'], A, None,
                                (3, 3), b['rpn_bev_roi'])
            ops.crop_and_resize(c, bneck_i, img_hw + (1,), b['img_norm'], A, None,
                                (3, 3), b['rpn_img_roi'])
            self._mark(c, st['step'], 'tail%d_crops' % f)
            if computed and not os.environ.get('DODT_PIPE_NO_RPN'):      # (tools/: timing experiment)
                self.rpn_head.forward(c, b['rpn_bev_roi'], b['rpn_img_roi'], A, b['rpn_logits'],
                                      b['rpn_offsets'], scratch['rpn'])
            self._mark(c, st['step'], 'tail%d_rpn' % f)
            # -- a12, a5, a13: decode, project, NMS #1 --------------------------------------
            # (round 4: the elementwise runs of a frame's launch chain are one launch each -- rpn_decode, gather_project,
            #  final_decode below: the same arithmetic value for value, six launches fewer per frame;
            #  DODT_PIPE_FUSED_TAIL=0: the separate ops)
            if fused_tail:
                ops.rpn_decode(c, b['anchors'], h['rpn_offsets'], h['rpn_logits'], A, None, self.bev_extents_flat,
                               b['regressed'], b['prop_bev'], b['scores'])
            else:
                ops.offset_to_anchor(c, b['anchors'], h['rpn_offsets'], A, None, b['regressed'])
                ops.project_anchors_f32(c, b['regressed'], A, None, self.bev_extents_flat, self.p2,
                                        self.image_wh, d_bev_norm_tf=b['prop_bev'])
                ops.softmax_fg(c, h['rpn_logits'], A, None, b['scores'])
            ops.nms(c, b['prop_bev'], b['scores'], A, None, self.P,
                    cfg['rpn_nms_iou_thresh'], b['top_idx'], b['top_count'])
            if fused_tail:
                # -- stage 2: the kept proposals and their projections, 7x7 crops ---------------
                ops.gather_project(c, b['regressed'], b['top_idx'], self.P, b['top_count'], self.bev_extents_flat,
                                   self.p2, self.image_wh, b['top_anchors'], b['top_bev'], b['top_img'])
                self._mark(c, st['step'], 'tail%d_nms1' % f)
            else:
                ops.gather_rows(c, b['regressed'], 6, b['top_idx'], self.P, b['top_count'],
                                b['top_anchors'])
                self._mark(c, st['step'], 'tail%d_nms1' % f)
                ops.project_anchors_f32(c, b['top_anchors'], self.P, b['top_count'],
                                        self.bev_extents_flat, self.p2, self.image_wh,
                                        d_bev_norm_tf=b['top_bev'], d_img_norm_tf=b['top_img'])
            ops.crop_and_resize(c, feat_b, bev_hw + (FC,), b['top_bev'], self.P,
                                b['top_count'], (ROI, ROI), b['bev_rois'])
            ops.crop_and_resize(c, feat_i, img_hw + (FC,), b['top_img'], self.P,
                                b['top_count'], (ROI, ROI), b['img_rois'])
            yield 'crops'
            pair = self.fps == 2
            corr_offsets = h.get('corr_offsets') if pair and f % 2 == 0 else None
            self._mark(c, st['step'], 'tail%d_crops2' % f)
            if computed:
                self.avod_head.forward(c, b['bev_rois'], b['img_rois'], self.P, b['top_count'],
                                       [b['cls_logits'], b['offsets_4c']]
                                       + ([b['angle_vectors']] if self.box_4ca else []),
                                       scratch['fc'])
                self._mark(c, st['step'], 'tail%d_fc2' % f)
                yield 'head'
                if pair and f % 2 == 0 and not os.environ.get('DODT_PIPE_NO_CORR'):
                    # T branch: correlate the pair's BEV features, crop with frame 0's
                    # proposals (dt_rpn_model.py:324-331, dt_avod_model.py:267-273,300-304)
                    if corr_img:
                        pass        # (crops and head on frame 1's stream, below; the records wait for them)
                    else:
                        if split_t:
                            c.wait_mark(sides[(f + 1) % ns], self.CORR_ROIS_MARK)
                        else:
                            t_branch_crops(c, f, scratch)
                        self._mark(c, st['step'], 'tail%d_corrmap' % f)
                        self.corr_head.forward(c, b['corr_rois'], None, self.P, b['top_count'],
                                               [b['corr_offsets']], scratch['fc'])
            self._mark(c, st['step'], 'tail%d_heads' % f)
            # -- a14, a13: box_4c decode, NMS #2 ---------------------------------------------
            # record score = softmax over [background, class] (dt_evaluator.py:1226-1248)
            # box_4ca: all_orientations = atan2 of the angle vectors (dt_avod_model.py:547-548),
            # gathered with the boxes by NMS #2's indices (:631-634) inside the record kernel,
            # which applies the evaluator's heading correction (dt_evaluator.py:1166-1212)
            if fused_tail:
                ops.final_decode(c, b['top_anchors'], h['offsets_4c'], h['cls_logits'],
                                 h['angle_vectors'] if self.box_4ca else None, self.P, b['top_count'], plane,
                                 self.bev_extents_flat, b['boxes_3d'], b['pred_anchors'], b['nms2_boxes'],
                                 b['nms2_scores'], b['det_scores'], b['orientations'] if self.box_4ca else None)
                ops.nms(c, b['nms2_boxes'], b['nms2_scores'], self.P, b['top_count'], MAX_DET,
                        cfg['avod_nms_iou_thresh'], b['det_idx'], b['det_count'])
            else:
                ops.box_4c_decode(c, b['top_anchors'], h['offsets_4c'], self.P, b['top_count'],
                                  plane, self.bev_extents_flat, b['boxes_3d'], b['pred_anchors'],
                                  b['nms2_boxes'])
                ops.max_fg_logit(c, h['cls_logits'], 2, self.P, b['top_count'], b['nms2_scores'])
                ops.nms(c, b['nms2_boxes'], b['nms2_scores'], self.P, b['top_count'], MAX_DET,
                        cfg['avod_nms_iou_thresh'], b['det_idx'], b['det_count'])
                ops.softmax_fg(c, h['cls_logits'], self.P, b['top_count'], b['det_scores'])
                if self.box_4ca:
                    ops.angle_vector_to_orientation(c, h['angle_vectors'], self.P, b['top_count'],
                                                    b['orientations'])
            yield 'pack'
            ops.pack_detections(
                c, b['boxes_3d'], b['det_scores'], b['det_idx'], b['det_count'], MAX_DET,
                float(f % self.fps),
                self.d_records.offset(4 * MAX_DET * REC_COLS * f, (MAX_DET, REC_COLS)),
                self.d_rec_counts.offset(4 * f, (1,), np.int32), d_corr_offsets=corr_offsets,
                d_orientations=b['orientations'] if self.box_4ca else None)
            self._mark(c, st['step'], 'tail%d_end' % f)

        def drain(g):
            for _ in g:
                pass

        if not split_t:
            for f in range(nf):
                drain(frame(f))
            return
        for f0 in range(0, nf, 2):
            c0, c1 = sides[f0 % ns], sides[(f0 + 1) % ns]
            g0, g1 = frame(f0), frame(f0 + 1)
            while next(g0) != 'crops':          # frame 0 up to its 7x7 crops: its proposals stand
                pass
            c0.mark(self.PROPOSALS_MARK)
            if corr_img:
                # The map stands since the convs ended (run()).  Its crops at frame 0's proposals and the correlation head
                # go onto frame 1's stream, in front of that frame's own head -- frame 0's stream, which carried them,
                # was the longer chain by their 0.1 ms -- and frame 0's records wait for the offsets.
                while next(g0) != 'pack':
                    pass
                while next(g1) != 'crops':
                    pass
                c1.wait_mark(c0, self.PROPOSALS_MARK)
                c1.wait_mark(self.img_ctx, self.CORR_MAP_MARK + cur)
                ops.crop_and_resize(c1, self.corr_maps[cur][f0 // 2], bev_hw + (CORR_CH,), fr[f0]['top_bev'], self.P,
                                    fr[f0]['top_count'], (ROI, ROI), fr[f0]['corr_rois'],
                                    out_box_stride=self.corr_head.in_ld)
                self.corr_head.forward(c1, fr[f0]['corr_rois'], None, self.P, fr[f0]['top_count'],
                                       [fr[f0]['corr_offsets']], head_scratch[(f0 + 1) % ns]['fc'])
                c1.mark(self.CORR_ROIS_MARK)
                c0.wait_mark(c1, self.CORR_ROIS_MARK)
                drain(g0)
                drain(g1)
                continue
            while next(g0) != 'head':           # ... and its stage-2 head
                pass
            while next(g1) != 'crops':          # frame 1 up to its crops, then the T branch's map and crops
                pass
            c1.wait_mark(c0, self.PROPOSALS_MARK)
            t_branch_crops(c1, f0, head_scratch[(f0 + 1) % ns])
            c1.mark(self.CORR_ROIS_MARK)
            drain(g1)
            drain(g0)                           # waits for CORR_ROIS_MARK: correlation head, NMS #2, records

    PROPOSALS_MARK, CORR_ROIS_MARK = 252, 253   # mark slots of the side contexts (the T branch's hand-overs)
    TAIL_DONE_MARK = 247                        # ... : the end of a step's tail on that stream

    def sync(self):
        self.ctx.sync()

    def flops_per_step(self):
        """Conv stacks only (the roofline kernel); see head_flops_per_step."""
        return self.bev_net.flops() + self.img_net.flops()

    def mfma_flops_per_step(self):
        """FLOPs the matrix pipe executes for the two conv stacks (Winograd layers count 16/36)."""
        return self.bev_net.mfma_flops() + self.img_net.mfma_flops()

    def conv_bytes_per_step(self):
        """Algorithmic HBM bytes of the two conv stacks per step."""
        return self.bev_net.bytes() + self.img_net.bytes()

    def head_flops_per_step(self, anchor_counts=None):
        if self.rpn_head is None:
            return 0.0
        counts = anchor_counts or self.last_anchor_counts
        return (sum(self.rpn_head.flops(a) for a in counts)
                + self.nf * self.avod_head.flops(self.P)
                + (self.pairs * self.corr_head.flops(self.P) if self.corr_head else 0.0))

    def flops_per_pair(self):
        return self.flops_per_step() / self.pairs

    def close(self):
        self.bev_net.close()
        self.img_net.close()
        for hd in (self.rpn_head, self.avod_head, self.corr_head):
            if hd is not None:
                hd.close()
