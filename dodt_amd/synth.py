"""Deterministic synthetic inputs and weights of the shapes the hot path sees
(SURVEY.md section 8d).  No dataset or checkpoint ships with the reference, so
benchmarks and parity tests run on these."""
import numpy as np

from dodt_amd import config as _config

P2 = _config.KITTI_P2
R0_RECT = _config.KITTI_R0_RECT
TR_VELO_TO_CAM = _config.KITTI_TR_VELO_TO_CAM
IMAGE_WH = _config.KITTI_IMAGE_WH

PYRAMID_CHANNELS = {
    'conv1_1': (None, 32), 'conv1_2': (32, 32),
    'conv2_1': (32, 64), 'conv2_2': (64, 64),
    'conv3_1': (64, 128), 'conv3_2': (128, 128), 'conv3_3': (128, 128),
    'conv4_1': (128, 256), 'conv4_2': (256, 256), 'conv4_3': (256, 256),
    'upconv3': (256, 128), 'pyramid_fusion3': (256, 64),
    'upconv2': (64, 64), 'pyramid_fusion2': (128, 32),
    'upconv1': (32, 32), 'pyramid_fusion1': (64, 32),
}
PYRAMID_LAYERS = list(PYRAMID_CHANNELS.keys())
TRANSPOSED = ('upconv3', 'upconv2', 'upconv1')


velo_to_cam = _config.velo_to_cam


def pipeline_weights(cfg):
    """Random-init extractor weights of cfg's architecture for FramePipeline(bev_params=,
    img_params=): seeds 42 (BEV) and 142 (image), the ones every parity test uses."""
    plain = cfg['extractor'] == 'vgg'
    return dict(bev_params=pyramid_params(cfg['bev_depth'], 42, plain=plain),
                img_params=pyramid_params(cfg['img_depth'], 142, plain=plain))


def frame_seed(seq, frame):
    return 0xD0D7 + 1000 * seq + frame


def lidar_frame(seq, frame, n_points=120000, n_boxes=12):
    """(N,4) float32 xyzi in the velodyne frame: 64 rings, ground-plane returns
    capped at 80 m, 12 % of returns on random car-sized boxes."""
    rng = np.random.default_rng(frame_seed(seq, frame))
    az = rng.uniform(-np.pi, np.pi, n_points)
    ring = rng.integers(0, 64, n_points)
    el = np.deg2rad(-24.8 + (26.8 / 63.0) * ring) + rng.normal(0, 1e-3, n_points)
    rng_ground = np.where(np.sin(el) < -1e-3, 1.73 / np.maximum(-np.sin(el), 1e-3), 80.0)
    r = np.minimum(rng_ground, 80.0) * rng.uniform(0.98, 1.0, n_points)
    x = r * np.cos(el) * np.cos(az)
    y = r * np.cos(el) * np.sin(az)
    z = r * np.sin(el)
    k = int(0.12 * n_points)
    centres = np.stack([rng.uniform(5, 65, n_boxes), rng.uniform(-30, 30, n_boxes)], 1)
    which = rng.integers(0, n_boxes, k)
    x[:k] = centres[which, 0] + rng.uniform(-2.0, 2.0, k)
    y[:k] = centres[which, 1] + rng.uniform(-0.85, 0.85, k)
    z[:k] = -1.73 + rng.uniform(0.0, 1.5, k)
    inten = rng.uniform(0, 1, n_points)
    return np.stack([x, y, z, inten], 1).astype(np.float32)


def image_frame(seq, frame, wh=IMAGE_WH):
    rng = np.random.default_rng(frame_seed(seq, frame) + 7)
    return rng.integers(0, 256, size=(wh[1], wh[0], 3), dtype=np.uint8)


def pyramid_params(in_ch, seed=42, plain=False):
    """Per-layer rng(seed + layer index): w ~ N(0, 2/(9 Cin)) in TF layout
    (HWIO; HWOI for the transposed convs), BN mean 0 / var 1 / beta ~ N(0, .01).
    plain: the encoder-only extractors of config 1 (bev_vgg.py / img_vgg.py): conv1_1 ..
    conv4_3 (the same draws as the pyramid's encoder) and a 256 -> 1 bottleneck."""
    params = {}
    for li, name in enumerate(PYRAMID_LAYERS):
        if plain and not name.startswith('conv'):
            continue
        cin, cout = PYRAMID_CHANNELS[name]
        cin = in_ch if cin is None else cin
        rng = np.random.default_rng(seed + li)
        std = np.sqrt(2.0 / (9 * cin))
        shape = (3, 3, cout, cin) if name in TRANSPOSED else (3, 3, cin, cout)
        params[name] = dict(
            w=rng.normal(0, std, size=shape).astype(np.float32),
            beta=rng.normal(0, 0.01, size=cout).astype(np.float32),
            mean=np.zeros(cout, dtype=np.float32),
            var=np.ones(cout, dtype=np.float32))
    rng = np.random.default_rng(seed + 100)
    fc = 256 if plain else 32
    params['bottleneck'] = dict(
        w=rng.normal(0, np.sqrt(2.0 / fc), size=(1, 1, fc, 1)).astype(np.float32),
        beta=rng.normal(0, 0.01, size=1).astype(np.float32),
        mean=np.zeros(1, dtype=np.float32), var=np.ones(1, dtype=np.float32))
    return params


def head_outputs(seq, frame, n_anchors, n_proposals):
    """Seeded stand-ins for the dense heads' outputs (for the index-exact parity tests,
    which inject them): RPN objectness logits (A,2) + offsets (A,6); stage-2 class logits
    (P,2), box_4c offsets (P,10), angle vectors (P,2; box_4ca), correlation offsets (P,3)."""
    rng = np.random.default_rng(frame_seed(seq, frame) + 13)
    return dict(
        rpn_logits=rng.normal(0, 2.0, size=(n_anchors, 2)).astype(np.float32),
        rpn_offsets=rng.normal(0, 0.1, size=(n_anchors, 6)).astype(np.float32),
        cls_logits=rng.normal(0, 2.0, size=(n_proposals, 2)).astype(np.float32),
        offsets_4c=rng.normal(0, 0.1, size=(n_proposals, 10)).astype(np.float32),
        corr_offsets=rng.normal(0, 0.3, size=(n_proposals, 3)).astype(np.float32),
        angle_vectors=rng.normal(0, 1.0, size=(n_proposals, 2)).astype(np.float32))


def _dense(rng, shape, fan_in, bias_std=0.01):
    return dict(w=rng.normal(0, np.sqrt(2.0 / fan_in), size=shape).astype(np.float32),
                b=rng.normal(0, bias_std, size=shape[-1]).astype(np.float32))


def head_params(seed=242, roi=7, feat=32, corr_ch=25, fc_sizes=(2048, 2048, 2048),
                rpn_fc=256, n_classes=2, off_size=10, ang_size=2):
    """Random-init weights of the dense heads in the TF variable shapes
    (dt_rpn_model.py:445-537; fusion_fc_layers.py:94-180; avod_corr_layers_builder.py:126-169;
    sizes from pyramid_cars_with_aug_dt_5_corr_tracking.config:64-90)."""
    rng = np.random.default_rng(seed)
    rpn = {}
    for br, n_out in (('cls', 2), ('reg', 6)):
        rpn[br + '_fc6'] = _dense(rng, (3, 3, 1, rpn_fc), 9)
        rpn[br + '_fc7'] = _dense(rng, (1, 1, rpn_fc, rpn_fc), rpn_fc)
        rpn[br + '_fc8'] = _dense(rng, (1, 1, rpn_fc, n_out), rpn_fc)
        rpn[br + '_fc8']['w'] *= np.float32(0.25 if br == 'reg' else 1.0)

    def stack(k_in, outs):
        p, k = {}, k_in
        for i, n in enumerate(fc_sizes):
            p['fc%d' % (6 + i)] = _dense(rng, (k, n), k)
            k = n
        for name, n in outs:
            p[name] = _dense(rng, (k, n), k)
            p[name]['w'] *= np.float32(0.1 if name == 'off_out' else 1.0)
        return p
    return dict(rpn=rpn,
                avod=stack(roi * roi * feat, (('cls_out', n_classes), ('off_out', off_size))
                           + ((('ang_out', ang_size),) if ang_size else ())),
                corr=stack(roi * roi * corr_ch, (('off_out', 3),)))
