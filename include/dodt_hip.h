/*
 * dodt_hip.h -- C ABI of libdodt_hip.so, the MI355X (gfx950) hot path of DODT/AVOD.
 *
 * The reference has no FFI on this path: its seams are ordinary Python call
 * signatures.  Each entry point below names the reference
 * call it stands behind (paths relative to the reference root).  The ABI follows
 * the one C precedent in the reference, wavedata's integralImage3D
 * (wavedata/wavedata/tools/core/lib/src/integral_images_3d.cpp:66-77, bound in
 * wavedata/wavedata/tools/core/integral_image.py:22-23,96-122): extern "C",
 * caller-allocated outputs, plain pointers and sizes, no ownership transfer --
 * except that every call returns an int status (0 = ok) and the message is
 * available from dodt_last_error() instead of failing silently.
 *
 * Conventions
 *   - One dodt_ctx per GPU per process; it owns one HIP stream and the scratch
 *     memory.  Calls on a ctx are serialised by the caller (the reference is
 *     single threaded, one sess.run at a time).
 *   - Every pointer argument whose name starts with d_ is DEVICE memory
 *     (dodt_malloc, or any hipMalloc'ed / torch-allocated buffer); all other
 *     pointers are host memory, read before the call returns.
 *   - Compute calls are asynchronous on the ctx stream; dodt_ctx_sync() or a
 *     dodt_memcpy_d2h() makes results visible to the host.
 *   - Image-like tensors are NHWC float32 with the batch dimension dropped,
 *     exactly the layout of the reference's TF placeholders.
 *   - No call retains a caller pointer past its return (weights are copied).
 */
#ifndef DODT_HIP_H
#define DODT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DODT_OK 0
#define DODT_ERR_INVALID 1   /* bad argument (the reference raises ValueError) */
#define DODT_ERR_HIP 2       /* a HIP runtime call failed                       */
#define DODT_ERR_UNSUPPORTED 3

typedef struct dodt_ctx dodt_ctx;

/* ---- context, memory --------------------------------------------------- */
int dodt_version(void);
const char* dodt_last_error(void);            /* thread-local, never NULL */
int dodt_ctx_create(int device_id, dodt_ctx** out);
/* Own stream at the device's greatest priority: for short latency-critical chains
 * (per-frame NMS / heads) that share the GPU with long persistent conv launches. */
int dodt_ctx_create_high_priority(int device_id, dodt_ctx** out);
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int dodt_ctx_create_on_stream(int device_id, void* hip_stream, dodt_ctx** out);
int dodt_ctx_destroy(dodt_ctx* ctx);
int dodt_ctx_sync(dodt_ctx* ctx);
/* Timing marks (256 per context) for stream-level timelines: record one on ctx's stream;
 * elapsed GPU time between two marks, possibly of different contexts (waits for `to`). */
int dodt_mark(dodt_ctx* ctx, int slot);
/* Later work on ctx waits for mark `slot` of `other` as last recorded (dodt_ctx_wait_for with the point of
 * `other`'s stream chosen earlier than the call). */
int dodt_ctx_wait_mark(dodt_ctx* ctx, dodt_ctx* other, int slot);
int dodt_mark_elapsed(dodt_ctx* from, int from_slot, dodt_ctx* to, int to_slot, float* ms);
/* Stream-level join: work enqueued on ctx AFTER this call starts only when
 * everything enqueued on `other` BEFORE this call has finished (hipEventRecord on
 * other's stream + hipStreamWaitEvent on ctx's stream; the host does not block).
 * Lets independent frames run on their own contexts/streams and meet again. */
int dodt_ctx_wait_for(dodt_ctx* ctx, dodt_ctx* other);
int dodt_malloc(dodt_ctx* ctx, size_t bytes, void** d_out);
int dodt_free(dodt_ctx* ctx, void* d_ptr);
int dodt_memcpy_h2d(dodt_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int dodt_memcpy_d2h(dodt_ctx* ctx, void* dst, const void* d_src, size_t bytes);
int dodt_memset(dodt_ctx* ctx, void* d_dst, int value, size_t bytes);
/* Page-locked host memory and copies from it that do not block the host: the way raw frames
 * (velodyne .bin contents, camera images -- what KittiUtils hands to create_feed_dict,
 * avod/core/models/dt_rpn_model.py:865-1042) reach the device under the previous step's
 * kernels.  The source must stay untouched until the stream has passed the copy. */
int dodt_pinned_alloc(dodt_ctx* ctx, size_t bytes, void** out);
int dodt_pinned_free(dodt_ctx* ctx, void* ptr);
int dodt_memcpy_h2d_async(dodt_ctx* ctx, void* d_dst, const void* pinned_src, size_t bytes);

/* Small device->host reads that do not stall the stream: begin enqueues a copy of
 * n <= 16 int32 into pinned slot `slot` (0..31) and records an event; end waits for
 * that event only (kernels enqueued after begin keep running) and returns the
 * values.  Used to learn the kept-anchor count while the conv stacks run. */
int dodt_fetch_i32_begin(dodt_ctx* ctx, const int32_t* d_src, int n, int slot);
int dodt_fetch_i32_end(dodt_ctx* ctx, int slot, int32_t* dst, int n);

/* HIP-event stopwatch on the ctx stream: start; ...launches...; stop -> ms
 * (stop waits for the stream). */
int dodt_timer_start(dodt_ctx* ctx);
int dodt_timer_stop(dodt_ctx* ctx, float* ms_out);

/* ---- a0-a3: BEV voxel/slice generator --------------------------------------
 * Stands behind BevSlices.generate_bev (avod/core/bev_generators/bev_slices.py:
 * 33-150) and, with DODT_PTS_VELO_XYZI, also the point path in front of it:
 * lidar_to_cam_frame + z>0 + image-frustum filter
 * (wavedata/.../core/calib_utils.py:484-523, obj_detection/tracking_utils.py:
 * 117-150).  All geometry is evaluated in float64 like the reference. */
#define DODT_PTS_VELO_XYZI 0 /* d_points = (N,4) float32 x,y,z,intensity, velodyne frame */
#define DODT_PTS_CAM_3XN 1   /* d_points = (3,N) float64 rectified camera frame          */

typedef struct dodt_bev_params {
    int32_t point_format;     /* DODT_PTS_*                                         */
    int32_t num_slices;       /* height slices (5); output depth = num_slices + 1   */
    double velo_to_cam[12];   /* (3,4) row major = (R0_rect . Tr_velo_to_cam)[0:3]  */
    double p2[12];            /* (3,4) camera matrix                                 */
    double im_w, im_h;        /* frustum: 0 < u < im_w, 0 < v < im_h (strict)       */
    double plane[4];          /* ground plane a,b,c,d                               */
    double extents[6];        /* x0,x1,y0,y1,z0,z1 (strict on both sides)           */
    double voxel_size;        /* float32-rounded 0.1 as python sees it (F7)         */
    double height_lo, height_hi;
    double occ_lo, occ_hi;    /* slice of the anchor-filter occupancy grid (0.2,2.0) */
    /* Ego-motion registration of the second frame of a pair into the first frame's
     * coordinates, KittiTrackingDataset.point_cloud_transform
     * (avod/datasets/kitti/kitti_tracking_dataset.py:303-335, call site :489):
     * p' = float32((p + pre_translate) @ pre_rotate) in the VELODYNE frame (float64 arithmetic,
     * one rounding into the float32 cloud, as the reference stores it), applied in front of
     * velo_to_cam when has_pre_transform != 0 (DODT_PTS_VELO_XYZI only).  pre_rotate is (3,3)
     * row major = Rz . Rx . Ry of Oxts.get_rotate_matrix (kitti_tracking_utils.py:147-215).
     * The BEV maps come from the warped cloud; the occupancy bits for the anchor filter from
     * the UN-warped one, because the reference re-reads the raw file for that grid
     * (kitti_tracking_utils.py:98-126) -- reproduced, not fixed. */
    int32_t has_pre_transform;
    int32_t reserved_;
    double pre_translate[3];
    double pre_rotate[9];
} dodt_bev_params;

/* d_bev_out: (Z, X, num_slices+1) float32, Z = 700 rows, X = 800 cols for the
 *            KITTI extents; out[r, c, s] = map_s[c, Z-1-r] like the reference.
 * d_occ_bits_out (may be NULL): occupancy of the [occ_lo, occ_hi) slice as a
 *            bit grid, Z rows of ceil(X/32) uint32 words, bit (x & 31) of word
 *            [z * ceil(X/32) + (x >> 5)]; feeds dodt_anchor_filter.
 * Returns DODT_ERR_INVALID when X*Z cells or the y-bin range do not fit the
 * packed 32-bit key (n_points >= 2^25 or more than 127 y-bins). */
int dodt_bev_slices(dodt_ctx* ctx, const void* d_points, int n_points,
                    const dodt_bev_params* params, float* d_bev_out,
                    uint32_t* d_occ_bits_out);
/* Waits for the stream and reports whether the last dodt_bev_slices on this ctx
 * met a point whose cell lies outside the grid (*flags & 1) -- the case in which
 * the reference raises ValueError("Extents are smaller than ...")
 * (voxel_grid_2d.py:130-138) -- or overflowed its work list (*flags & 2). */
int dodt_bev_status(dodt_ctx* ctx, int* flags);

/* ---- a4: empty-anchor filter -------------------------------------------------
 * Stands behind get_empty_anchor_filter_2d (avod/core/anchor_filter.py:64-119)
 * + IntegralImage2D.query (wavedata/.../core/integral_image_2d.py:39-87).
 * d_anchor_cells: (n_anchors,4) int32 [x1,z1,x2,z2] grid indices produced on the
 *   host once per configuration exactly as the reference does (float32 corners,
 *   float32 division, int32 truncation, clip to [0,X] / [0,Z]).
 * d_keep_idx_out: (n_anchors) int32, first *count entries = kept anchor indices in
 *   ascending (generator) order.  d_count_out: (1) int32. */
int dodt_anchor_filter(dodt_ctx* ctx, const uint32_t* d_occ_bits, int nx, int nz,
                       const int32_t* d_anchor_cells, int n_anchors,
                       int density_threshold, int32_t* d_keep_idx_out,
                       int32_t* d_count_out);

/* ---- a5/a6: anchor projection --------------------------------------------------
 * project_to_bev (avod/core/anchor_projector.py:13-69), project_to_image_space
 * (:72-156, numpy branch, float64 in / float32 out) and the [y1,x1,y2,x2] reorder
 * (:254-273; models/dt_rpn_model.py:983-985).
 * d_anchors: (n_total,6) float64 table; d_idx (may be NULL = identity): (n) int32
 * rows to project; *d_n (may be NULL): device count overriding n (<= n).
 * Outputs (n,4) float32 normalised boxes in TF order [y1,x1,y2,x2]. */
int dodt_project_anchors_f64(dodt_ctx* ctx, const double* d_anchors,
                             const int32_t* d_idx, int n, const int32_t* d_n,
                             const double bev_extents[4], const double p2[12],
                             double im_w, double im_h, float* d_bev_norm_out,
                             float* d_img_norm_out, float* d_anchors_f32_out);

/* TF-branch twins, float32 throughout (models/dt_avod_model.py:170-206):
 * project_to_bev + tf_project_to_image_space (anchor_projector.py:159-251).
 * d_bev_out (n,4) metres [x1,z1,x2,z2] (may be NULL), d_bev_norm_tf_out and
 * d_img_norm_tf_out (n,4) in TF order. */
int dodt_project_anchors_f32(dodt_ctx* ctx, const float* d_anchors, int n,
                             const int32_t* d_n, const float bev_extents[4],
                             const float p2[12], float im_w, float im_h,
                             float* d_bev_out, float* d_bev_norm_tf_out,
                             float* d_img_norm_tf_out);

/* ---- a7: image preprocessing ---------------------------------------------------
 * ImgFeatureExtractor.preprocess_input (avod/core/feature_extractors/
 * img_feature_extractor.py:16-35): legacy bilinear resize + per-channel mean
 * subtraction.  d_img_u8: (in_h,in_w,3) uint8 RGB.  d_out: (out_h,out_w,out_c)
 * float32 with out_c >= 3; channels >= 3 are written as 0 (the conv kernels
 * want an even channel count). */
int dodt_img_preprocess(dodt_ctx* ctx, const uint8_t* d_img_u8, int in_h, int in_w,
                        int out_h, int out_w, int out_c, const float mean_rgb[3],
                        float* d_out);

/* ---- a8-a10: feature extractors ---------------------------------------------------
 * Stands behind feature_extractor_builder.get_extractor(cfg).build(...)
 * (avod/core/feature_extractors/bev_vgg_pyramid.py:30-178, img_vgg_pyramid.py:
 * 30-177) plus the 1x1 bottleneck (models/dt_rpn_model.py:298-322). */
typedef struct dodt_extractor dodt_extractor;
#define DODT_EXTRACTOR_VGG_PYR 0
/* The plain-VGG extractors of the AVOD cars_example configuration (BASELINE.json configs[0]):
 * BevVgg / ImgVgg.build (avod/core/feature_extractors/bev_vgg.py:34-118, img_vgg.py:33-120):
 * the same encoder (conv1_1 .. conv4_3, three VALID 2x2 pools that floor odd sizes), then
 * tf.image.resize_bilinear of conv4_3 to (in_h / 8 * 4, in_w / 8 * 4) -- 256 channels -- and the
 * 256 -> 1 bottleneck of avod/core/models/rpn_model.py:251-267.  fp32 only, pad_top = 0. */
#define DODT_EXTRACTOR_VGG 1
/* OR into `kind`: the extractor shares the GPU with other streams (the frame-pair pipeline
 * runs both nets side by side).  Layers are then single launches: the tail launches that
 * even out a layer's last round when it has the GPU to itself only add work when another
 * stream fills the idle CUs anyway. */
#define DODT_EXTRACTOR_SHARED_GPU 0x100
/* OR into `kind`: conv path on the bf16 MFMA (BASELINE.json configs[2]: "bf16 conv path").
 * Inputs and weights of every conv are rounded to bf16 (nearest even), products accumulate
 * in fp32, batch-norm + ReLU run in fp32, activations between layers are stored as bf16;
 * the first layer's arithmetic, the network output and the bottleneck stay fp32.  The
 * reference computes in fp32: with this flag conv outputs agree with an fp32
 * run to ~1e-2 of their scale, not 1e-4 (tests/test_gpu_conv_bf16.py states the bars). */
#define DODT_EXTRACTOR_BF16 0x200
/* OR into `kind`: fp32-grade convs on the bf16 MFMA ("split" mode).  Every stored activation
 * and every weight is a pair hi + lo of bf16 values (16 mantissa bits), a product is
 * w_hi x_hi + w_hi x_lo + w_lo x_hi in three bf16 MFMAs with fp32 accumulation: relative
 * error ~2^-16 per product instead of bf16's 2^-8, within the 1e-4 bar of the fp32 tests
 * (tests/test_gpu_conv_split.py runs them at the fp32 tolerances). */
#define DODT_EXTRACTOR_SPLIT 0x400
/* Form of the fp32 3x3 stride-1 layers (everything else has one kernel): 4 = Winograd F(4x4,3x3),
 * 2 = Winograd F(2x2,3x3), 1 = its one-workgroup-per-CU variants, 0 = direct implicit GEMM.  All are
 * fp32 throughout and within the 1e-4 layer bar; they differ in speed and in how far the rounding
 * noise of a whole stack lies from the exact sums (DESIGN.md 2 / 5.0 give both, and the rule that
 * chose the default).  DODT_CONV_WINO in the environment overrides the default, once per process. */
#define DODT_CONV_MODE_DEFAULT 2
int dodt_conv_mode(void);
/* in_c: channels of the input tensor as stored (6 for BEV; 4 for the padded
 * image); pad_top: zero rows added on top (4 for BEV 700->704, 0 for images);
 * batch: frames processed per forward call (2 = both frames of a pair). */
int dodt_extractor_create(dodt_ctx* ctx, int kind, int in_h, int in_w, int in_c,
                          int pad_top, int batch, dodt_extractor** out);
int dodt_extractor_destroy(dodt_extractor* ex);
/* One conv layer: name as in the TF variable scope ("conv1_1" ... "conv4_3",
 * "upconv3", "pyramid_fusion3", ..., "bottleneck" -- (1,1,32,1), or (1,1,256,1) for
 * DODT_EXTRACTOR_VGG).  w: HWIO (kh,kw,cin,cout)
 * for conv2d, (kh,kw,cout,cin) for conv2d_transpose, as TF stores them.
 * beta/mean/var: slim.batch_norm variables (no gamma, eps = 1e-3).  Host memory. */
int dodt_extractor_set_layer(dodt_extractor* ex, const char* name, const float* w,
                             int kh, int kw, int c_a, int c_b, const float* beta,
                             const float* mean, const float* var);
/* Zero-copy input: *d_ptr = where frame 0 of the (batch, in_h, in_w, in_c) input
 * lives inside the extractor (frame f at + f * *frame_stride_floats).  A producer
 * that writes there (e.g. dodt_bev_slices) can pass d_in = NULL to forward. */
int dodt_extractor_input(dodt_extractor* ex, float** d_ptr, long long* frame_stride_floats);
/* d_in: (batch, in_h, in_w, in_c) or NULL (see above); d_feat_out: (batch, in_h, in_w, 32);
 * d_bottleneck_out (may be NULL): (batch, in_h, in_w, 1).  DODT_EXTRACTOR_VGG: the outputs are
 * (batch, out_h, out_w, 256) and (batch, out_h, out_w, 1), sizes from
 * dodt_extractor_output_shape. */
int dodt_extractor_forward(dodt_extractor* ex, const float* d_in, float* d_feat_out,
                           float* d_bottleneck_out);
/* The same forward on an input the caller keeps in the extractor's own input layout: (batch, pad_top + in_h, in_w,
 * in_c) float32 whose first pad_top rows of every frame are zero (bev_vgg_pyramid.py:58 pads the BEV map by four rows)
 * -- read in place by the first layer, no copy into the extractor's buffer (a pipeline that double-buffers its
 * inputs writes them straight into two such buffers). */
int dodt_extractor_forward_padded(dodt_extractor* ex, const float* d_x0, float* d_feat_out,
                                  float* d_bottleneck_out);
/* Until further notice (d_x0 = NULL: back to its own buffer) forwards without an input argument read d_x0, laid out
 * as above, as the extractor's input buffer; d_x0 must outlive that use (tools: stand-alone timing on a pipeline's inputs). */
int dodt_extractor_set_input(dodt_extractor* ex, const float* d_x0);
/* Size of the feature map forward() returns: (in_h, in_w, 32) for the pyramid,
 * (in_h / 8 * 4, in_w / 8 * 4, 256) for DODT_EXTRACTOR_VGG. */
int dodt_extractor_output_shape(const dodt_extractor* ex, int* h, int* w, int* c);
/* 1 when conv1_1 runs folded into conv1_2's launch (bf16 conv path, the default there; bev_vgg_pyramid.py:63-66's two convs in
 * one kernel): conv1_1's map is then not stored (dodt_extractor_read_activation refuses it) and its arithmetic is the bf16
 * MFMA's at fp32 grade (x and w as hi + lo bf16 pairs, three MFMAs per product term) instead of the fp32 MFMA's.  0 otherwise
 * (fp32 / split conv paths, or DODT_CONV_BF16_FIRST2=0); a negative DODT_ERR_* code for a NULL extractor. */
int dodt_extractor_first_layers_folded(const dodt_extractor* ex);
/* Debug/test access to an intermediate activation by layer name: copies the
 * (batch, h, w, c) float32 tensor to host memory `dst` (NULL to query shape). */
int dodt_extractor_read_activation(dodt_extractor* ex, const char* name, float* dst,
                                   int* h, int* w, int* c);
/* FLOPs of one forward call (2*M*N*K summed over conv layers, all frames): the ALGORITHMIC
 * count of the direct form, whatever kernel computes a layer. */
double dodt_extractor_flops(const dodt_extractor* ex);
/* FLOPs the matrix pipe executes for one forward.  The fp32 3x3 stride-1 layers run as Winograd
 * minimal filtering, fp32 throughout: F(4x4,3x3) (36 multiplications per 4x4 outputs and channel pair
 * where the direct form needs 144) or F(2x2,3x3) (16 per 2x2 outputs instead of 36), selected by
 * DODT_CONV_WINO=4|2 in the environment (0: the direct kernels; DESIGN.md 5.0 states the default and
 * each form's distance to the direct result); split mode issues three bf16 MFMAs per product. */
double dodt_extractor_mfma_flops(const dodt_extractor* ex);
/* Algorithmic HBM bytes of one forward (each map and the weights read / written once). */
double dodt_extractor_bytes(const dodt_extractor* ex);
/* What a forward is made of, layer by layer in launch order, and how long each layer's launches
 * took in one forward (a HIP event pair around every layer on the extractor's stream; the call
 * waits for the stream).  bench.py's roofline object is built from this: `kernel` is the
 * __global__ function a rocprofv3 kernel trace lists, flops_executed what the matrix pipe issues
 * for the layer (see dodt_extractor_mfma_flops), bytes the layer's share of dodt_extractor_bytes. */
typedef struct dodt_layer_info {
    char name[32];          /* TF variable scope of the layer */
    char kernel[48];
    int32_t launches;       /* 1, or 2 when a tail launch evens out the last round */
    int32_t items;          /* work items (tiles x channel tiles x frames) */
    double flops_direct;    /* 2 M N K of the direct form */
    double flops_executed;
    double bytes;
    float ms;
    int32_t reserved_;
} dodt_layer_info;
int dodt_extractor_layer_count(const dodt_extractor* ex);
int dodt_extractor_forward_timed(dodt_extractor* ex, const float* d_in, float* d_feat_out,
                                 float* d_bottleneck_out, dodt_layer_info* info, int n_info);

/* ---- a11: ROI crop ------------------------------------------------------------------
 * tf.image.crop_and_resize(image, boxes, box_ind=0, crop_size) call sites
 * models/dt_rpn_model.py:418-428 and models/dt_avod_model.py:253-273.
 * d_image (H,W,C); d_boxes (n,4) [y1,x1,y2,x2] normalised; d_out (n,ch,cw,C);
 * *d_n (may be NULL) overrides n on the device. */
int dodt_crop_and_resize(dodt_ctx* ctx, const float* d_image, int H, int W, int C,
                         const float* d_boxes, int n, const int32_t* d_n, int crop_h,
                         int crop_w, float* d_out);
/* The same with box b's crop written at d_out + b * out_box_stride floats (>= ch*cw*C; the floats
 * between crops are left untouched): the flattened crops become rows of a matrix whose row length is
 * padded for the fully connected layer that reads them (the correlation head's fc6 has K = 7*7*25 =
 * 1225; rows of 1248 zero-padded floats let it run on the LDS-DMA GEMM). */
int dodt_crop_and_resize_strided(dodt_ctx* ctx, const float* d_image, int H, int W, int C,
                                 const float* d_boxes, int n, const int32_t* d_n, int crop_h,
                                 int crop_w, float* d_out, long long out_box_stride);

/* ---- T branch: correlation of the two frames' BEV feature maps ------------------------
 * Stands behind avod/core/corr_layers/correlation.py:7-27 -> the Correlation custom op
 * (avod/core/ops/correlation/correlation_op.cc:53-62, kernel correlation_kernel.cu.cc:
 * 21-119, padding pad.cu.cc:14-73) with kernel_size 1 and stride_1 1.
 * d_a, d_b: (H,W,C) float32, C = 32; d_out: (H+2*pad-2*max_displacement,
 * W+2*pad-2*max_displacement, (2*(max_displacement/stride_2)+1)^2). */
int dodt_correlation(dodt_ctx* ctx, const float* d_a, const float* d_b, int H, int W, int C,
                     int max_displacement, int stride_2, int pad, float* d_out);

/* ---- dense heads: fully connected layers (fp32 MFMA) ------------------------------------
 * y[M][N] = act(x[M][K] . w[K][N] + bias[N]); stands behind slim.fully_connected / the 1x1
 * and VALID 3x3 slim.conv2d of the heads: avod/core/models/dt_rpn_model.py:445-537 (anchor
 * predictor), avod/core/avod_fc_layers/fusion_fc_layers.py:136-180 (early fusion heads),
 * avod/builders/avod_corr_layers_builder.py (correlation offsets head).
 * w: host, row-major (K,N) = the TF variable's layout (conv kernels reshaped to (kh*kw*cin,
 * cout)); relu != 0 applies ReLU.  d_x2 (may be NULL): x = (x + x2) / 2, the "mean" fusion
 * of BEV and image crops.  ldx, ldy: row strides in floats; *d_m (may be NULL) overrides M
 * on the device.  ctx == NULL in forward uses the creating context's stream. */
/* d_out[r][k] = (d_a[r][k] + d_b[r][k]) / 2 for r < min(*d_n, rows): the "mean" fusion of the BEV and
 * image crops (avod/core/avod_fc_layers/avod_fc_layer_utils.py:38-41, dt_rpn_model.py:434-439 with
 * both path-drop masks 1) as a pass of its own, in front of a layer that then runs without d_x2 (the
 * GEMM with the fusion inside its K loop is half as fast at the stage-2 head's size).  row_floats a
 * multiple of 4, blocks 16-byte aligned and contiguous. */
int dodt_mean_fusion(dodt_ctx* ctx, const float* d_a, const float* d_b, int rows, const int32_t* d_n,
                     int row_floats, float* d_out);
typedef struct dodt_fc dodt_fc;
int dodt_fc_create(dodt_ctx* ctx, int K, int N, const float* w, const float* bias, int relu,
                   dodt_fc** out);
/* flags: DODT_FC_RELU; DODT_FC_BF16 = x rounded to bf16 on load, weights stored as bf16,
 * bf16 MFMA with fp32 accumulation, bias + activation and the output in fp32 (the heads'
 * counterpart of DODT_EXTRACTOR_BF16; not the reference's arithmetic). */
#define DODT_FC_RELU 1
#define DODT_FC_BF16 2
int dodt_fc_create_ex(dodt_ctx* ctx, int K, int N, const float* w, const float* bias, int flags,
                      dodt_fc** out);
int dodt_fc_destroy(dodt_fc* fc);
int dodt_fc_forward(dodt_fc* fc, dodt_ctx* ctx, const float* d_x, const float* d_x2, int ldx,
                    int M, const int32_t* d_m, float* d_y, int ldy);
/* A layer whose columns go to `parts` (1..3) dense arrays d_ys[p] of (M, widths[p]) floats, widths summing to
 * the layer's N: the output layers of a head (fusion_fc_layers.py:94-133 build_output_layers: cls | offsets |
 * angle vectors from the same fc_drop) created as ONE layer with concatenated weights and run as one launch.
 * Layers with N <= 32, K a multiple of 16 and 16-byte aligned rows (ldx % 4 == 0); else
 * DODT_ERR_UNSUPPORTED. */
int dodt_fc_forward_split(dodt_fc* fc, dodt_ctx* ctx, const float* d_x, int ldx, int M, const int32_t* d_m,
                          int parts, const int* widths, float* const* d_ys);
/* bf16 heads with bf16 activations in HBM (round 4).  A DODT_FC_BF16 layer also takes input rows that are
 * ALREADY bf16 (ldx in elements; dodt_fc_bf16_row_elems(fc) of them per row, zeros beyond K; 0 = no such path for
 * this layer) and writes bf16 rows (y_bf16 != 0, ldy in elements; rounded to nearest even -- the rounding the next
 * layer's load would apply, so the arithmetic is the one of dodt_fc_forward on DODT_FC_BF16 layers) or float32.
 * Layers with N % 128 == 0 and K >= 128 (both operands staged by LDS-DMA), and the output layers dodt_fc_forward_split
 * takes (float32 output).  dodt_rows_to_bf16 makes the first layer's rows: bf16((a + b) / 2) -- the heads' mean
 * fusion, avod_fc_layer_utils.py:38-41 -- or bf16(a) when d_b is NULL, zero tail up to out_ld. */
int dodt_fc_bf16_row_elems(const dodt_fc* fc);
int dodt_fc_forward_bf16(dodt_fc* fc, dodt_ctx* ctx, const void* d_x_bf16, int ldx, int M, const int32_t* d_m,
                         void* d_y, int ldy, int y_bf16);
int dodt_fc_forward_split_bf16(dodt_fc* fc, dodt_ctx* ctx, const void* d_x_bf16, int ldx, int M, const int32_t* d_m,
                               int parts, const int* widths, float* const* d_ys);
int dodt_rows_to_bf16(dodt_ctx* ctx, const float* d_a, const float* d_b, int rows, const int32_t* d_n,
                      int row_floats, int in_ld, void* d_out_bf16, int out_ld);
double dodt_fc_flops(const dodt_fc* fc, int M);

/* ---- a13: NMS -------------------------------------------------------------------------
 * tf.image.non_max_suppression(boxes, scores, max_output_size, iou_threshold),
 * call sites models/dt_rpn_model.py:587-591 and models/dt_avod_model.py:606-613.
 * Candidate order is (score descending, index ascending) -- TF leaves ties
 * unspecified.  d_sel_out: (max_out) int32 selected ORIGINAL indices in score
 * order; d_count_out: (1) int32.  *d_n (may be NULL) overrides n. */
int dodt_nms(dodt_ctx* ctx, const float* d_boxes, const float* d_scores, int n,
             const int32_t* d_n, int max_out, float iou_threshold, int32_t* d_sel_out,
             int32_t* d_count_out);

/* ---- a12/a14: box encoders (TF branch, float32) ------------------------------------------ */
/* anchor_encoder.offset_to_anchor (avod/core/anchor_encoder.py:99-150) */
int dodt_offset_to_anchor(dodt_ctx* ctx, const float* d_anchors, const float* d_offsets,
                          int n, const int32_t* d_n, float* d_out);
/* 2-way softmax, column 1 (models/dt_rpn_model.py:581-584) */
int dodt_softmax_fg(dodt_ctx* ctx, const float* d_logits2, int n, const int32_t* d_n,
                    float* d_scores_out);
/* One launch for the three elementwise steps between the RPN head and NMS #1 (models/dt_rpn_model.py:560-591):
 * dodt_offset_to_anchor (-> d_regressed_out (n,6)), dodt_project_anchors_f32's normalised BEV boxes of the regressed anchors
 * (-> d_bev_norm_tf_out (n,4)) and dodt_softmax_fg (-> d_scores_out (n)) -- the same arithmetic, value for value. */
int dodt_rpn_decode(dodt_ctx* ctx, const float* d_anchors, const float* d_offsets, const float* d_logits2, int n,
                    const int32_t* d_n, const float bev_extents[4], float* d_regressed_out,
                    float* d_bev_norm_tf_out, float* d_scores_out);
/* ... for the two behind NMS #1 (models/dt_rpn_model.py:593-612, dt_avod_model.py:176-200): dodt_gather_rows of the kept
 * proposals (width 6 -> d_rows_out (n,6)) and their dodt_project_anchors_f32 boxes in both views. */
int dodt_gather_project(dodt_ctx* ctx, const float* d_src, const int32_t* d_idx, int n, const int32_t* d_n,
                        const float bev_extents[4], const float p2[12], float im_w, float im_h,
                        float* d_rows_out, float* d_bev_norm_tf_out, float* d_img_norm_tf_out);
/* ... and for the four behind the stage-2 head (models/dt_avod_model.py:520-548,600-634): dodt_box_4c_decode,
 * dodt_max_fg_logit of the two-class logits (-> d_nms_scores_out), dodt_softmax_fg (-> d_det_scores_out) and, with
 * d_angle_vectors, dodt_angle_vector_to_orientation (-> d_orientations_out; both NULL for box_4c). */
int dodt_final_decode(dodt_ctx* ctx, const float* d_top_anchors, const float* d_offsets, const float* d_cls_logits2,
                      const float* d_angle_vectors, int n, const int32_t* d_n, const float plane[4],
                      const float bev_extents[4], float* d_boxes_3d_out, float* d_pred_anchors_out,
                      float* d_bev_tf_out, float* d_nms_scores_out, float* d_det_scores_out,
                      float* d_orientations_out);
/* out[i] = src[idx[i]] rows of `width` floats (tf.gather call sites
 * dt_rpn_model.py:593-597, dt_avod_model.py:616-640) */
int dodt_gather_rows(dodt_ctx* ctx, const float* d_src, int width, const int32_t* d_idx,
                     int n, const int32_t* d_n, float* d_out);
/* NMS #2 score: max over the non-background logits, column 1.. of (n,n_cls)
 * (models/dt_avod_model.py:606: tf.reduce_max(all_cls_logits[:, 1:], axis=1)) */
int dodt_max_fg_logit(dodt_ctx* ctx, const float* d_logits, int n_cls, int n,
                      const int32_t* d_n, float* d_scores_out);
/* tf_angle_vector_to_orientation (avod/core/orientation_encoder.py:20-34; call site
 * models/dt_avod_model.py:547-548): atan2(y, x) of the stage-2 head's `ang_out` rows [x, y]
 * (box_4ca: avod/core/avod_fc_layers/avod_fc_layer_utils.py:11-17).  d_angle_vectors (n,2),
 * d_orientations_out (n,). */
int dodt_angle_vector_to_orientation(dodt_ctx* ctx, const float* d_angle_vectors, int n,
                                     const int32_t* d_n, float* d_orientations_out);
/* Detection record of one frame, the 17 columns the evaluator writes per box
 * (get_avod_predicted_boxes_3d_and_scores, avod/core/dt_evaluator.py:1134-1259): box_3d(7),
 * score, class index, the box shifted by the correlation head's offsets (x += dx, z += dz,
 * ry += dry; d_corr_offsets (n,3), frame 0 of a pair) or 7 zeros (d_corr_offsets NULL:
 * frame 1), frame mark.
 * d_orientations (n,) (may be NULL = box_4c): box_4ca's correction of each selected box by
 * the regressed angle (:1166-1212) -- difference wrapped to [-pi, pi], l/w swapped and ry
 * +-pi/2 when it lies in (pi/4, 3pi/4), ry + pi when |difference| >= 3pi/4, ry wrapped above
 * pi -- applied before the correlation shift, exactly as the evaluator orders them.
 * Rows d_sel[0..*d_count) of boxes_3d / scores / orientations (the tf.gather by nms_indices,
 * models/dt_avod_model.py:616-640); remaining rows of the (max_det,17) output are zeroed;
 * d_count_out[0] = *d_count. */
int dodt_pack_detections(dodt_ctx* ctx, const float* d_boxes_3d, const float* d_scores,
                         const float* d_orientations, const float* d_corr_offsets,
                         const int32_t* d_sel, const int32_t* d_count, int max_det,
                         float frame_mark, float* d_rec_out, int32_t* d_count_out);
/* Stage-2 decode (models/dt_avod_model.py:464-469,575-603):
 *   anchors_to_box_3d(fix_lw) -> tf_box_3d_to_box_4c -> + offsets ->
 *   tf_box_4c_to_box_3d -> tf_box_3d_to_anchor -> project_to_bev (metres) ->
 *   reorder.  (avod/core/box_3d_encoder.py:188-322, box_4c_encoder.py:85-165,
 *   369-484.)  d_top_anchors (n,6), d_offsets (n,10), plane[4] host.
 * Outputs: d_boxes_3d_out (n,7), d_pred_anchors_out (n,6), d_bev_tf_out (n,4)
 * [z1,x1,z2,x2] metres (the NMS #2 boxes).  Any output may be NULL. */
int dodt_box_4c_decode(dodt_ctx* ctx, const float* d_top_anchors, const float* d_offsets,
                       int n, const int32_t* d_n, const float plane[4],
                       const float bev_extents[4], float* d_boxes_3d_out,
                       float* d_pred_anchors_out, float* d_bev_tf_out);

/* ---- (e) multi-GPU: the one exchange step of the path, RCCL over xGMI, no PyTorch -----------
 * The reference runs on ONE device (avod/experiments/run_tracking_inference.py:109-128 sets a
 * single CUDA_VISIBLE_DEVICES and walks the sequences in a loop), so there is no reference
 * collective to replace: frame pairs shard over ranks (pair i -> rank i mod N) and the
 * sequential temporal module (avod/core/dt_evaluator_utils.py:189-362) needs every rank's detection
 * records in sequence order.  One process per GPU; librccl.so is dlopen'ed by the first of these
 * calls (a single-GPU process never maps it).
 *   id          DODT_COMM_ID_BYTES opaque bytes (an ncclUniqueId) made by ONE rank with
 *               dodt_comm_unique_id and handed to the others by the launcher's means (a file, an
 *               environment variable, a socket: dodt_amd/sharding.py uses a file next to the
 *               rendezvous port).
 *   create      ncclCommInitRank on ctx's device + a side stream that every collective runs on.
 *   all_gather  waits (event) for what `producer` has enqueued so far, then gathers
 *               float32 [pairs][frames][max_det][cols] + int32 [pairs][frames] of every rank into
 *               d_all_records / d_all_counts ([world][pairs]...; rank major) in one RCCL group, and
 *               signals `slot`'s event (slots 0..3: the caller's ring of send buffers).  The
 *               producer's stream is NOT made to wait: call dodt_comm_join(comm, slot, consumer)
 *               on the context that is about to overwrite that slot's send buffer or read the
 *               gathered data on the device, dodt_comm_sync for the host.
 *   barrier     all ranks have arrived (and the side stream is drained);
 *   max_f64     *value = max over ranks (the bench's max-over-ranks timing). */
typedef struct dodt_comm dodt_comm;
#define DODT_COMM_ID_BYTES 128
int dodt_comm_unique_id(uint8_t* id_out);
int dodt_comm_create(dodt_ctx* ctx, int rank, int world, const uint8_t* id, dodt_comm** out);
int dodt_comm_destroy(dodt_comm* comm);
/* From now on the communicator's collectives run on ctx's stream instead of a stream of their own (every stream
 * beyond four costs this runtime throughput): they then sit in that stream's order, between the
 * caller's kernels.  ctx must outlive the communicator. */
int dodt_comm_attach(dodt_comm* comm, dodt_ctx* ctx);
int dodt_comm_rank(const dodt_comm* comm, int* rank, int* world);
/* Measurement aid: from now on every dodt_all_gather_records is preceded, on the stream that
 * carries the collectives, by a one-wave kernel that idles for `microseconds` -- what that stream sees when a peer
 * reaches the rendezvous late.  0 switches it off.  At most 100 000 us. */
int dodt_comm_set_late_peer(dodt_comm* comm, double microseconds);
int dodt_all_gather_records(dodt_comm* comm, dodt_ctx* producer, int slot, const float* d_records,
                            const int32_t* d_counts, int pairs, int frames, int max_det, int cols,
                            float* d_all_records, int32_t* d_all_counts);
int dodt_comm_join(dodt_comm* comm, int slot, dodt_ctx* consumer);
int dodt_comm_sync(dodt_comm* comm);
int dodt_comm_barrier(dodt_comm* comm);
int dodt_comm_max_f64(dodt_comm* comm, double* value);

#ifdef __cplusplus
}
#endif
#endif /* DODT_HIP_H */
