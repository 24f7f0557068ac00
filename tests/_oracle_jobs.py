"""Oracle jobs that run in spawned worker processes (CPU only), so that long oracle computations of a GPU test
proceed side by side: tests/test_gpu_heads.py::test_pair_free_running_by_conv_mode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def free_running_pair(job):
    """(sequence, frames, arithmetic 'f32' | 'exact', BLAS threads, keep feature maps) -> (BEV feature maps of
    the two frames or None, the pair's detections): the oracle end to end on raw inputs."""
    seq, frames, arith, threads, keep = job
    from threadpoolctl import threadpool_limits
    from dodt_amd import config, synth
    from oracle import pipeline as opipe
    from oracle import tfops
    C = config.PYRAMID_DODT
    with threadpool_limits(limits=threads):
        hp = synth.head_params()
        w = synth.pipeline_weights(C)
        pts = [synth.lidar_frame(seq, f) for f in frames]
        imgs = [synth.image_frame(seq, f) for f in frames]
        inps = [opipe.frame_inputs(p, C, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2, synth.IMAGE_WH) for p in pts]

        def extract():
            return [opipe.extract(inps[j]['bev'], imgs[j], w['bev_params'], w['img_params'], C['img_dims'])
                    for j in range(2)]
        if arith == 'exact':
            with tfops.exact_sums():
                feats = extract()
        else:
            feats = extract()
        dets = opipe.pair_detections_computed(inps, feats, hp, C, synth.P2, synth.IMAGE_WH, 1024)
    return ([f[0] for f in feats] if keep else None), dets
