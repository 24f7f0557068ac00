#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec of the DODT hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

One step = one pass of the hot path over one synthetic KITTI-shaped frame pair
(2 x 120k points + 2 x 1242x375 RGB, tau = 2) whose inputs are already resident
in HBM.  Frame pairs shard across ranks (pair i -> rank i mod N, weak scaling);
after each step the ranks all-gather their detection records (RCCL).  Rank 0
prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--points', type=int, default=120000)
    ap.add_argument('--proposals', type=int, default=1024)
    ap.add_argument('--pairs-per-step', type=int, default=1,
                    help='independent frame pairs carried by one step on each GPU')
    ap.add_argument('--heads', choices=('computed', 'injected'), default='computed',
                    help='dense heads + correlation branch on the device (the whole path), '
                         'or their outputs injected from HBM')
    ap.add_argument('--conv-dtype', choices=('f32', 'f32s', 'bf16'), default='f32',
                    help="arithmetic of the conv stacks: 'f32' = the reference's (fp32 MFMA), "
                         "'bf16' = BASELINE configs[2]'s bf16 conv path (bf16 MFMA, fp32 accumulate)")
    ap.add_argument('--head-dtype', choices=('f32', 'bf16'), default='f32',
                    help='arithmetic of the dense heads (fp32 MFMA or bf16 MFMA, fp32 accumulate)')
    ap.add_argument('--config', choices=('dodt', 'cars_example'), default='dodt',
                    help="'dodt' = pyramid_cars_with_aug_dt_5_tracking frame pairs (the metric's "
                         "workload); 'cars_example' = BASELINE.json configs[0]: single frames "
                         'through the plain-VGG AVOD configuration (value is then frames/s)')
    ap.add_argument('--no-alt', action='store_true',
                    help='skip the short extra run with the other conv arithmetic')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-pairs', type=int, default=2,
                    help='frame pairs (= pool workers) of the CPU baseline sample')
    return ap.parse_args()


def _usable_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota and by
    the GPU box's per-GPU share of 16 (DODT_CPU_CORES overrides)."""
    if os.environ.get('DODT_CPU_CORES'):
        return max(1, int(os.environ['DODT_CPU_CORES']))
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 16))


def _cpu_pair_worker(job):
    """One whole frame pair through the oracle with `threads` BLAS threads (a pool worker)."""
    seq, threads, computed = job
    from threadpoolctl import threadpool_limits
    from dodt_amd import config, synth
    from oracle import pipeline as opipe
    cfg = config.PYRAMID_DODT
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        bev_params = synth.pyramid_params(6, 42)
        img_params = synth.pyramid_params(3, 142)
        inps, feats = [], []
        for f in (0, 2):
            xyzi = synth.lidar_frame(seq, f)
            img = synth.image_frame(seq, f)
            inp = opipe.frame_inputs(xyzi, cfg, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2,
                                     synth.IMAGE_WH)
            inps.append(inp)
            feats.append(opipe.extract(inp['bev'], img, bev_params, img_params, cfg['img_dims']))
        if computed:
            opipe.pair_detections_computed(inps, feats, synth.head_params(), cfg, synth.P2,
                                           synth.IMAGE_WH, 1024)
        else:
            for k, f in enumerate((0, 2)):
                opipe.frame_detections(inps[k], synth.head_outputs(seq, f, 89600, 1024), cfg,
                                       synth.P2, synth.IMAGE_WH, 1024, *feats[k], frame_mark=k)
        return time.perf_counter() - t0


def _cpu_points_worker(job):
    """The numpy point path a0-a6 of `n` frames on one thread (the part of the path the
    reference itself runs on the CPU): seconds spent."""
    seq, n = job
    from threadpoolctl import threadpool_limits
    from dodt_amd import config, synth
    from oracle import pipeline as opipe
    cfg = config.PYRAMID_DODT
    clouds = [synth.lidar_frame(seq, f) for f in range(n)]
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for xyzi in clouds:
            opipe.frame_inputs(xyzi, cfg, synth.R0_RECT, synth.TR_VELO_TO_CAM, synth.P2,
                               synth.IMAGE_WH)
        return time.perf_counter() - t0


def cpu_baseline(computed_heads=True, pairs=2):
    """The oracle ('port' of the reference's algorithm) timed on this host's cores, before
    anything touches the GPU (the pool's workers are fresh processes).  Test infrastructure
    used as a yardstick, never as the product.
      value        whole frame pairs / s: a frame-parallel pool of `pairs` workers, each one
                   pair with cores // pairs BLAS threads (restatement of the TF
                   half, thread count stated)
      point_path   the numpy half a0-a6 alone, which is how far the reference itself runs on
                   a CPU: 1 process x 1 thread, and a pool of `cores` single-thread workers"""
    import multiprocessing as mp
    cores = _usable_cores()
    workers = max(1, min(pairs, cores))
    threads = max(1, cores // workers)
    ctx = mp.get_context('spawn')
    with ctx.Pool(workers) as pool:
        t0 = time.perf_counter()
        pool.map(_cpu_pair_worker, [(20 + i, threads, computed_heads) for i in range(workers)])
        wall = time.perf_counter() - t0
    one = _cpu_points_worker((30, 4))
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_points_worker, [(40 + i, 1) for i in range(cores)])     # spawn + import
        t0 = time.perf_counter()
        pool.map(_cpu_points_worker, [(50 + i, 4) for i in range(cores)])
        pool_wall = time.perf_counter() - t0
    return dict(value=workers / wall, unit='frame-pairs/s', cores=int(workers * threads),
                kind='port',
                sample='%d synthetic frame pairs through oracle/pipeline.py (numpy point path + '
                       'numpy/BLAS fp32 conv stacks + crop + NMS%s), frame-parallel pool of %d '
                       'processes x %d BLAS threads, %.1f s wall'
                       % (workers, ' + correlation + dense heads' if computed_heads else '',
                          workers, threads, wall),
                point_path={'stages': 'a0-a6 (points -> BEV maps, anchor filter, projections)',
                            'one_thread_frames_per_s': round(4 / one, 3),
                            'pool_frames_per_s': round(4 * cores / pool_wall, 3),
                            'pool_workers': cores, 'unit': 'frames/s'})


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started by hand without a launcher: run the ranks as children (nothing has touched
        # the GPU yet in this process) and leave with their exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    n_gpus = max(args.gpus, 1)
    baseline = None
    if not args.no_cpu_baseline and world == 1:
        # first, while the GPU is still untouched: the workers are spawned processes
        baseline = cpu_baseline(args.heads == 'computed', args.cpu_pairs)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the HIP path')
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from dodt_amd import config, device, sharding, synth
    from dodt_amd.pipeline import FramePairPipeline, MAX_DET, REC_COLS
    cfg = config.PYRAMID_DODT if args.config == 'dodt' else config.CARS_EXAMPLE
    stream = torch.cuda.current_stream().cuda_stream
    ctx = device.Context(local_rank, stream=stream)
    pps = args.pairs_per_step
    computed = args.heads == 'computed'
    made = []   # pipelines built so far: later ones reuse the first one's streams

    def measure(conv_dtype, steps, warmup, head_dtype='f32', cfg=cfg, proposals=args.proposals,
                from_host=False):
        """`steps` timed steps of the pipeline built for conv_dtype, then the conv stacks
        alone (roofline).  Returns a dict of raw measurements."""
        fps = cfg['frames_per_sample']
        feat_c = 256 if cfg['extractor'] == 'vgg' else 32
        pipe = FramePairPipeline(ctx, cfg, **synth.pipeline_weights(cfg),
                                 n_points_max=args.points, rpn_nms_size=proposals,
                                 pairs_per_step=pps,
                                 head_params=synth.head_params(feat=feat_c) if computed else None,
                                 conv_dtype=conv_dtype, head_dtype=head_dtype,
                                 reuse_streams_of=made[0] if made and made[0].fps == fps else None)
        made.append(pipe)

        # detection records live in torch memory so that RCCL can ship them
        # (two of each: the pipeline alternates them by step parity)
        rec = [torch.zeros((pps, fps, MAX_DET, REC_COLS), dtype=torch.float32, device='cuda')
               for _ in range(2)]
        cnt = [torch.zeros((pps, fps), dtype=torch.int32, device='cuda') for _ in range(2)]
        pipe.use_record_buffers([t.data_ptr() for t in rec], [t.data_ptr() for t in cnt])
        gathered = torch.zeros((world * pps, fps, MAX_DET, REC_COLS), dtype=torch.float32,
                               device='cuda')
        gathered_cnt = torch.zeros((world * pps, fps), dtype=torch.int32, device='cuda')

        # a small ring of distinct synthetic batches, resident in HBM before timing starts;
        # every pair of a batch comes from a different sequence (they are independent)
        n_batches = 2
        batches = []
        for i in range(n_batches):
            pts, imgs, heads = [], [], []
            for j in range(pps):
                seq = (rank * n_batches + i) * pps + j
                for f in ((2 * i, 2 * i + 2) if fps == 2 else (2 * i,)):      # tau = 2
                    pts.append(synth.lidar_frame(seq, f, args.points))
                    imgs.append(ctx.array(synth.image_frame(seq, f)))
                    if not computed:
                        heads.append({k: ctx.array(v) for k, v in
                                      synth.head_outputs(seq, f, pipe.n_all, pipe.P).items()})
            b = dict(pts=[ctx.array(p) for p in pts], n=[len(p) for p in pts],
                     imgs=imgs, heads=None if computed else heads)
            if from_host:   # the same frames in page-locked host memory (PCIe-inclusive run)
                b['h_pts'], b['h_imgs'] = [], []
                for p_, d_img in zip(pts, imgs):
                    hp = ctx.pinned((args.points, 4), np.float32)
                    hp.a[:len(p_)] = p_
                    hi = ctx.pinned(d_img.shape, np.uint8)
                    hi.a[...] = d_img.download()
                    b['h_pts'].append(hp)
                    b['h_imgs'].append(hi)
            batches.append(b)

        state = {'n': 0, 'par': 0}

        fake = os.environ.get('DODT_BENCH_FAKE_GATHER') == '1'   # rehearsal of the N > 1 stream
        side = torch.cuda.Stream() if fake else None               # pattern on one GPU

        def gather(par):
            if world > 1:
                sharding.all_gather_records(dist, rec[par], cnt[par], gathered, gathered_cnt)
            elif fake:     # like ProcessGroupNCCL: own stream, joined with the current one
                cur_s = torch.cuda.current_stream()
                side.wait_stream(cur_s)
                with torch.cuda.stream(side):
                    gathered[:pps].copy_(rec[par])
                    gathered_cnt[:pps].copy_(cnt[par])
                cur_s.wait_stream(side)

        def step(i):
            p = batches[i % n_batches]
            if from_host:
                par = pipe.run_from_host(p['h_pts'], p['n'], p['h_imgs'], p['heads'])
            else:
                par = pipe.run(p['pts'], p['n'], p['imgs'], p['heads'])
            if state['n'] > 0:     # records of the previous step are complete on this stream
                gather(1 - par)
            state['n'] += 1
            state['par'] = par

        def drain():
            pipe.finish()
            if state['n'] > 0:
                gather(state['par'])
            state['n'] = 0

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # steps are pipelined two deep inside pipe.run(); finish() drains the last one, so
        # exactly `steps` complete steps (convs AND tails) lie inside the timed region
        for i in range(warmup):
            step(i)
        drain()
        barrier()
        # one HIP event per step on the main stream (torch's current stream = ctx's stream):
        # event i fires when step i's convs and step i-1's tail are done, so the deltas show
        # a slow fill / drain step or a clock ramp that the single wall-clock window hides
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 2)]
        host_ms = []
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            th = time.perf_counter()
            step(i)
            evs[i + 1].record()
            host_ms.append((time.perf_counter() - th) * 1e3)
        host_enqueue_ms = (time.perf_counter() - t0) / steps * 1e3   # host side of a step
        drain()
        evs[steps + 1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps + 1)]
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- roofline of the dominant kernel family: the conv stacks -----------------------
        # measured live with HIP events on the stream the kernels run on
        # (each net alone on its own stream, so the kernel durations do not overlap)
        reps = max(3, min(steps, 10))
        barrier()
        conv_ms = 0.0
        for net, c, f, b in ((pipe.bev_net, ctx, pipe.feat[0]['bev_feat'], pipe.feat[0]['bev_bneck']),
                             (pipe.img_net, pipe.img_ctx, pipe.feat[0]['img_feat'],
                              pipe.feat[0]['img_bneck'])):
            c.sync()
            c.timer_start()
            for _ in range(reps):
                net.forward_device(None, f, b)
            conv_ms += c.timer_stop() / reps
        # both nets side by side, as in the timed steps (but nothing else on the GPU)
        ctx.sync()
        pipe.img_ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            pipe.img_ctx.wait_for(ctx)
            pipe.bev_net.forward_device(None, pipe.feat[0]['bev_feat'], pipe.feat[0]['bev_bneck'])
            pipe.img_net.forward_device(None, pipe.feat[0]['img_feat'], pipe.feat[0]['img_bneck'])
            ctx.wait_for(pipe.img_ctx)
        both_ms = ctx.timer_stop() / reps
        res = dict(elapsed=elapsed, host_enqueue_ms=host_enqueue_ms, conv_ms=conv_ms, reps=reps,
                   step_ms=step_ms, host_ms=host_ms,
                   both_ms=both_ms,
                   flops=pipe.flops_per_step(), mfma_flops=pipe.mfma_flops_per_step(),
                   head_gflop=pipe.head_flops_per_step() / 1e9,
                   conv_bytes=pipe.conv_bytes_per_step(),
                   anchors=list(pipe.last_anchor_counts), steps=steps)
        pipe.close()
        return res

    import gc
    if os.environ.get('DODT_BENCH_GC', 'freeze') == 'freeze':
        # everything allocated so far (torch, numpy, the package) leaves the collector's
        # generations: a full collection of that heap in the middle of a 0.1 s timed window
        # would stall the enqueueing thread for tens of ms
        gc.collect()
        gc.freeze()
    m = measure(args.conv_dtype, args.steps, args.warmup, args.head_dtype)
    elapsed, host_enqueue_ms, conv_ms, reps = m['elapsed'], m['host_enqueue_ms'], m['conv_ms'], m['reps']
    alt = None
    if not args.no_alt:
        # the other conv arithmetic, same workload, a shorter run: reported beside the main
        # measurement, never part of `value`
        k = max(5, args.steps // 2)

        def short(conv_dtype, head_dtype, **kw):
            a = measure(conv_dtype, k, 2, head_dtype, **kw)
            return {'conv_dtype': conv_dtype, 'head_dtype': head_dtype,
                    'value': round(world * k * pps / a['elapsed'], 3), 'unit': 'frame-pairs/s',
                    'steps': k, 'ms_per_step': round(a['elapsed'] / k * 1e3, 4),
                    'conv_stacks_tflops': round(a['flops'] / (a['conv_ms'] * 1e-3) / 1e12, 2),
                    'conv_stacks_ms': round(a['conv_ms'], 4)}
        alt = {'note': "f32s = split mode: hi + lo bf16 pairs, three bf16 MFMAs per product term, "
                       'fp32 accumulate -- passes the fp32 layer tests at 1e-4 '
                       '(tests/test_gpu_conv_split.py); bf16 conv = BASELINE.json configs[2]\'s '
                       'bf16 conv path (bars in tests/test_gpu_conv_bf16.py); bf16 heads = the '
                       "same for the FC layers; f32 = the reference's arithmetic on the fp32 MFMA",
               'runs': [short(c, h) for c, h in (('f32', 'f32'), ('f32s', 'f32'), ('bf16', 'f32'),
                                                  ('bf16', 'bf16'))
                        if (c, h) != (args.conv_dtype, args.head_dtype) and (computed or h == 'f32')]}
        if args.config == 'dodt':
            # BASELINE.json configs[0]: single frames through the AVOD cars_example
            # configuration (plain VGG extractors, 480 x 1590 image, 300 proposals, fp32)
            r = short('f32', 'f32', cfg=config.CARS_EXAMPLE,
                      proposals=config.CARS_EXAMPLE['rpn_test_nms_size'])
            r.update(unit='frames/s', config='avod_cars_example: bev_vgg + img_vgg, 1 frame '
                                             'per step (BASELINE.json configs[0])')
            alt['cars_example'] = r
            # PCIe-inclusive: raw frames start in page-locked host memory and are copied by
            # hipMemcpyAsync on the prep streams inside every step (never `value`)
            r = short(args.conv_dtype, args.head_dtype, from_host=True)
            r['inputs'] = ('2 x (%d x 16 B points + 1242x375x3 B image) = %.1f MB per pair from '
                           'pinned host memory inside each step'
                           % (args.points, 2 * (args.points * 16 + 1242 * 375 * 3) / 1e6))
            alt['pcie_inclusive'] = r
    flops = m['flops']
    achieved = flops / (conv_ms * 1e-3) / 1e12
    # HBM bytes per conv launch: PMC counters cannot be read from inside this process; the
    # figure comes from the rocprofv3 --pmc passes over this same command (profiles/)
    traffic, traffic_src = None, None
    tj = os.path.join(ROOT, 'profiles', {'f32': 'r2_conv_traffic.json',
                                         'f32s': 'r2f32s_conv_traffic.json',
                                         'bf16': 'r2bf16_conv_traffic.json'}[args.conv_dtype])
    if os.path.exists(tj):
        t = json.load(open(tj))
        traffic = round(t['fetch_bytes_per_launch'] + t['write_bytes_per_launch'])
        traffic_src = t['source']
    common = dict(traffic=traffic, traffic_unit='bytes/launch', traffic_source=traffic_src,
                  kernel='wino43_f32_kernel (24 launches per step) + deconv3x3_f32_kernel (6 transposed convs) '
                         '+ 2 first-layer launches',
                  launch_ms=round(conv_ms, 4), algorithmic_gflop=round(flops / 1e9, 2),
                  algorithmic_mbytes=round(m['conv_bytes'] / 1e6, 1),
                  launches_per_step=32, avg_launch_us=round(conv_ms * 1e3 / 32, 2),
                  side_by_side_ms=round(m['both_ms'], 4),
                  side_by_side_tflops=round(flops / (m['both_ms'] * 1e-3) / 1e12, 2),
                  measured='HIP events, each net alone on its stream, %d reps after the '
                           'timed region' % reps)
    if args.conv_dtype == 'f32':
        # fp32 MFMA: 157.3 TFLOP/s dense (MI355X_MICROARCH.md chip table); every layer is
        # MFMA-bound at fp32.  `achieved` prices the ALGORITHMIC FLOPs (2 M N K of the direct
        # form); the 3x3 stride-1 layers run as Winograd F(4x4,3x3), which executes
        # 36/144 of them (so `frac` can exceed 1: the direct form's roof is not this algorithm's):
        # what the matrix pipe really does is given beside it.
        executed = m['mfma_flops'] / (conv_ms * 1e-3) / 1e12
        roofline = dict(bound='mfma', achieved=round(achieved, 2), peak=157.3, unit='TFLOP/s',
                        frac=round(achieved / 157.3, 4),
                        executed_gflop=round(m['mfma_flops'] / 1e9, 2),
                        executed_tflops=round(executed, 2),
                        executed_frac=round(executed / 157.3, 4),
                        algorithm='Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32 for the 3x3 '
                                  'stride-1 layers (fp32 throughout; 4x fewer multiplications than '
                                  'the direct form priced by `achieved`), an LDS-DMA staged direct kernel for the transposed '
                                  'convs, direct implicit GEMM for the first layers', **common)
    elif args.conv_dtype == 'f32s':
        # split mode: three bf16 MFMAs per product term -> 3x the algorithmic FLOPs on the
        # bf16 pipe (2.5 PFLOP/s dense); the fp32-equivalent rate is given beside it
        roofline = dict(bound='mfma', achieved=round(3 * achieved, 2), peak=2500.0, unit='TFLOP/s',
                        frac=round(3 * achieved / 2500.0, 4), executed_flops='3 x algorithmic',
                        fp32_equivalent_tflops=round(achieved, 2), **common)
    else:
        # bf16 MFMA (2.5 PFLOP/s) makes the stacks 16x cheaper in matrix time than in fp32:
        # they are bound by moving the maps (HBM ~8 TB/s), which is what is priced here;
        # the matrix-pipe fraction is given beside it
        gbs = m['conv_bytes'] / (conv_ms * 1e-3) / 1e9
        roofline = dict(bound='hbm', achieved=round(gbs, 1), peak=8000.0, unit='GB/s',
                        frac=round(gbs / 8000.0, 4), mfma_tflops=round(achieved, 2),
                        mfma_frac_of_2500=round(achieved / 2500.0, 4), **common)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            'metric': 'frame-pairs/sec (whole node) KITTI-shape tau=2',
            'value': round(world * args.steps * pps / elapsed, 3),
            'unit': 'frame-pairs/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 4),
            'host_enqueue_ms_per_step': round(host_enqueue_ms, 3),
            # HIP-event time between consecutive steps' completion on the main stream (the
            # last entry is the drain: the final step's tail with no convs beside it)
            'step_ms': {'min': round(min(m['step_ms'][:-1]), 3),
                        'median': round(float(np.median(m['step_ms'][:-1])), 3),
                        'max': round(max(m['step_ms'][:-1]), 3),
                        'drain': round(m['step_ms'][-1], 3),
                        'all': [round(v, 2) for v in m['step_ms']],
                        'host_max': round(max(m['host_ms']), 3),
                        'host_median': round(float(np.median(m['host_ms'])), 3)},
            'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.conv_dtype, 'data': 'synthetic',
            'head_dtype': args.head_dtype,
            'config': {'workload': ('DODT tau=2 frame pair: 2 x %dk pts + 2 x 1242x375 RGB, '
                                    'pyramid_cars_with_aug_dt_5_tracking (box_4ca), %d proposals, '
                                    '%s' % (args.points // 1000, args.proposals,
                                            'S+T path: correlation + dense heads on the device'
                                            if computed else 'S path (heads injected)'))
                       if args.config == 'dodt' else
                       ('AVOD single frame: %dk pts + 1242x375 RGB, avod_cars_example (plain '
                        'VGG extractors, box_4ca), %d proposals; value counts FRAMES/s'
                        % (args.points // 1000, args.proposals)),
                       'head_gflop_per_step': round(m['head_gflop'], 2),
                       'pairs_per_step_per_gpu': pps, 'parallelism': 'pair-shard x%d' % world,
                       'anchors_kept': m['anchors']},
            'roofline': roofline,
        }
        if alt is not None:
            out['alt'] = alt
        if baseline is not None:
            out['cpu_baseline'] = baseline
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
