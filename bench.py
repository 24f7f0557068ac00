#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec of the DODT hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU -- started by torch.distributed.run, which only serves as the process
   launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT are read from the environment), or by
   this script itself when it is run by hand)

One step = one pass of the hot path over one synthetic KITTI-shaped frame pair
(2 x 120k points + 2 x 1242x375 RGB, tau = 2) whose inputs are already resident
in HBM.  Frame pairs shard across ranks (pair i -> rank i mod N, weak scaling);
after each step the ranks all-gather their detection records with RCCL through the
package's own C-ABI (dodt_comm_*, include/dodt_hip.h) on a side stream.  No PyTorch is
imported.  Rank 0 prints ONE JSON line.
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
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
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
    ap.add_argument('--gather-every', type=int, default=8,
                    help='steps whose detection records travel in one all-gather (the temporal module '
                         'consumes them in sequence order whenever they arrive; fewer, larger messages)')
    ap.add_argument('--comm', action='store_true',
                    help='make an RCCL communicator even with one rank (exercises the exchange '
                         'step on a single GPU)')
    ap.add_argument('--cpu-pairs', type=int, default=2,
                    help='frame pairs (= pool workers) of the CPU baseline sample')
    ap.add_argument('--tau', type=int, default=2, help='frame stride of a pair (frames f, f + tau)')
    ap.add_argument('--boxes', type=int, default=12,
                    help='car-sized boxes in a synthetic cloud (SURVEY 8d: 12; 40 in the dense scene)')
    ap.add_argument('--sustained-steps', type=int, default=2000,
                    help='steps of alt.sustained, the headline workload over seconds (0: skip)')
    ap.add_argument('--allow-no-exchange', action='store_true',
                    help='N > 1 only: when RCCL cannot make the communicator, still time the ranks (host barrier, '
                         'NO records exchanged, said in config.exchange) instead of leaving with an error')
    ap.add_argument('--late-peer-ms', type=float, default=0.0,
                    help='experiment (with --comm): a device-side delay of this many ms in front of every gather on '
                         "the stream that carries the collectives -- a peer that arrives late at the rendezvous")
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


def _temporal_worker(job):
    """The host's temporal module ("M" of S+T+M) over one synthetic sequence on ONE thread: per pair the
    association + interpolation of the non-keyframes (interpolate_non_keyframe_predictions), then the
    sequence's tracker (encode_tracking_dets + track_through_ious with the reference's thresholds,
    avod_stack_tracking.config:137-140).  job: (records of the sequence's pairs, tau, repeats); returns
    seconds per repeat and the number of tracks."""
    records, tau, repeats = job
    if records is None:
        import dodt_amd.core.dt_evaluator_utils  # noqa: F401  (warm-up: spawn + imports)
        import dodt_amd.core.dt_inference_utils  # noqa: F401
        return 0.0, 0
    from threadpoolctl import threadpool_limits
    from dodt_amd import synth
    from dodt_amd.core import dt_evaluator_utils as M
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for _ in range(repeats):
            for rec in records:
                M.interpolate_non_keyframe_predictions(rec, tau + 1, 0.1, on_conflict='next_best')
            dt, di = M.encode_tracking_dets([(k * tau, k * tau + tau, rec) for k, rec in enumerate(records)],
                                            synth.P2, synth.IMAGE_WH, ['Car'], 0.1)
            tracks = M.track_through_ious(dt, di, 0.5, 0.005, 3)
        return (time.perf_counter() - t0) / repeats, len(tracks)


def temporal_host(pool, cores, records, tau, gpu_pairs_per_s):
    """alt.temporal_host: pairs/s of the host temporal module on the records the timed steps produced."""
    n_det = float(np.mean([len(r) for r in records])) / 2.0
    one, n_tracks = _temporal_worker((records, tau, 3))
    res = dict(stage='M of S+T+M on the host: interpolate_non_keyframe_predictions per pair + encode_tracking_dets + '
                     'track_through_ious per sequence (dodt_amd/core/dt_evaluator_utils.py; reference '
                     'dt_evaluator_utils.py:212-296,368-511)',
               records='%d consecutive steps of the timed run as one sequence, %.0f detections per frame on average, '
                       'tau = %d; %d tracks' % (len(records), n_det, tau, n_tracks),
               one_thread_pairs_per_s=round(len(records) / one, 1), ms_per_pair_one_thread=round(one / len(records) * 1e3, 3),
               unit='frame-pairs/s')
    if pool is not None:
        t0 = time.perf_counter()
        out = pool.map(_temporal_worker, [(records, tau, 6)] * cores)
        wall = time.perf_counter() - t0
        res.update(pool_pairs_per_s=round(cores * 6 * len(records) / wall, 1), pool_workers=cores,
                   pool_note='%d single-thread worker processes, one sequence each (sequences are independent)' % cores)
        res['keeps_up_with_gpu'] = bool(res['pool_pairs_per_s'] >= gpu_pairs_per_s)
        res['gpu_pairs_per_s'] = round(gpu_pairs_per_s, 1)
        res['threads_needed_for_gpu_rate'] = int(np.ceil(gpu_pairs_per_s / max(res['one_thread_pairs_per_s'], 1e-9)))
    return res


FP32_MFMA_PEAK = 157.3     # TFLOP/s dense, MI355X_MICROARCH.md chip table (v_mfma_f32_*_f32)
BF16_MFMA_PEAK = 2500.0    # TFLOP/s dense
HBM_PEAK = 8000.0          # GB/s


def _spawn_ranks(args):
    """Started by hand with --gpus N and no launcher: run the ranks as plain child processes (nothing has
    touched the GPU in this one), SUPERVISED: the first rank that leaves with an error ends the others (a rank
    blocked in an RCCL collective whose peer is gone would otherwise wait for ever), and that code is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DODT_RUN_ID=str(os.getpid()))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env))
    return _supervise(procs)


def _supervise(procs, poll_s=0.2, grace_s=10.0):
    """Wait for child processes; on the first non-zero exit terminate (then kill) the rest.  Returns the first
    failure's code, else 0."""
    failed = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and not failed:
                failed = rc if rc > 0 else 128 - rc
                sys.stderr.write('[bench] a rank left with code %d: ending the other %d\n' % (rc, len(live)))
                for q in live:
                    q.terminate()
                t_end = time.time() + grace_s
                for q in live:
                    try:
                        q.wait(timeout=max(0.0, t_end - time.time()))
                    except Exception:
                        q.kill()
                        q.wait()
                live = []
                break
        if live:
            time.sleep(poll_s)
    return failed


def _group_by_kernel(layers):
    """Per-kernel totals of a list of per-layer measurements (dicts of forward_timed)."""
    out = {}
    for l in layers:
        k = out.setdefault(l['kernel'], dict(kernel=l['kernel'], launches=0, ms=0.0, flops_executed=0.0,
                                             flops_direct=0.0, bytes=0.0, layers=[]))
        k['launches'] += l['launches']
        k['ms'] += l['ms'] if l['launches'] else 0.0      # (a folded layer's event pair brackets nothing)
        k['flops_executed'] += l['flops_executed']
        k['flops_direct'] += l['flops_direct']
        k['bytes'] += l['bytes']
        k['layers'].append(l['name'])
    return out


def _profile_json(name):
    path = os.path.join(ROOT, 'profiles', name)
    return json.load(open(path)) if os.path.exists(path) else None


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(_spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    n_gpus = max(args.gpus, 1)
    baseline = None
    if not args.no_cpu_baseline and world == 1:
        # first, while the GPU is still untouched: the workers are spawned processes
        baseline = cpu_baseline(args.heads == 'computed', args.cpu_pairs)

    m_pool, m_cores = None, _usable_cores()
    if not args.no_alt and world == 1 and args.config == 'dodt' and args.heads == 'computed':
        # the temporal module's workers: spawned now, while the GPU is untouched, idle until alt.temporal_host
        import multiprocessing as mp
        m_pool = mp.get_context('spawn').Pool(m_cores)
        m_pool.map(_temporal_worker, [(None, 0, 0)] * m_cores)

    from dodt_amd import _lib, config, device, ops, sharding, synth
    from dodt_amd.pipeline import (CORR_CH, CORR_MAX_DISP, CORR_PAD, CORR_STRIDE2, MAX_DET, REC_COLS,
                                   ROI, FramePairPipeline)
    cfg = config.PYRAMID_DODT if args.config == 'dodt' else config.CARS_EXAMPLE
    # DODT_BENCH_SAME_GPU=1 (rehearsal on a one-GPU box): every rank on device 0 -- RCCL refuses two
    # ranks on one device, so this exercises the spawner, the rendezvous and the degraded mode
    dev_id = 0 if os.environ.get('DODT_BENCH_SAME_GPU') == '1' else local_rank
    try:
        ctx = device.Context(dev_id)
    except _lib.DodtError as e:
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the HIP path (%s)' % e)
    comm, host_sync, comm_error = None, None, None
    # A rank whose peers are gone would wait in an RCCL collective for ever (and one that never reaches the
    # communicator's set-up leaves the others in ncclCommInitRank): a watchdog, re-armed around the set-up, every
    # barrier and every timed region, ends this rank instead -- the launcher / _supervise then ends the job.
    import signal
    watch = {'what': ''}

    def _stuck(signum, frame):
        sys.stderr.write('[bench] rank %d: no progress in %s within its limit; giving up\n' % (rank, watch['what']))
        sys.stderr.flush()
        os._exit(3)

    def arm(seconds, what):
        if world > 1:
            watch['what'] = what
            signal.alarm(int(seconds))

    def disarm():
        if world > 1:
            signal.alarm(0)
    if world > 1:
        signal.signal(signal.SIGALRM, _stuck)
        arm(600, 'RCCL set-up and the first barrier')
    if world > 1 or args.comm:
        try:
            # librccl prints a version banner on stdout when it starts: the job's stdout is ONE JSON line, so the
            # process's fd 1 points at stderr while the communicator is made
            sys.stdout.flush()
            saved_out = os.dup(1)
            os.dup2(2, 1)
            try:
                comm = sharding.Communicator(ctx, rank, world)
                comm.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved_out, 1)
                os.close(saved_out)
        except Exception as e:      # RCCL missing, the ranks could not meet, or the first collective failed
            if comm is not None:    # made, but its first barrier failed: not usable
                try:
                    comm.close()
                except Exception:
                    pass
                comm = None
            comm_error = '%s: %s' % (type(e).__name__, e)
            if world == 1:
                raise
            if not args.allow_no_exchange:
                # a whole-node number without the exchange step in it is not the metric: leave with an error
                # (every rank fails the same way; the launcher / _supervise ends the job)
                sys.stderr.write('[bench] rank %d: RCCL exchange unavailable (%s); no result (--allow-no-exchange '
                                 'times the ranks without it)\n' % (rank, comm_error))
                if rank == 0:
                    print(json.dumps({'metric': 'frame-pairs/sec (whole node) KITTI-shape tau=2', 'value': None,
                                      'unit': 'frame-pairs/s', 'n_gpus': n_gpus,
                                      'error': 'no detection records exchanged: ' + comm_error}))
                    sys.stdout.flush()
                os._exit(4)
            # Degraded mode, asked for and said loudly in config.exchange: the ranks keep the timing protocol
            # (barriers and max over ranks through files), but NO records are exchanged
            sys.stderr.write('[bench] rank %d: RCCL exchange unavailable (%s); host barrier only\n'
                             % (rank, comm_error))
            host_sync = sharding.HostBarrier(rank, world)
        disarm()
    if comm is not None and args.late_peer_ms > 0:
        comm.set_late_peer(args.late_peer_ms * 1e3)
    computed = args.heads == 'computed'
    comm_stream = os.environ.get('DODT_BENCH_COMM_STREAM', 'own' if world > 1 else 'side')
    made = []   # pipelines built so far: later ones reuse the first one's streams

    def measure(conv_dtype, steps, warmup, head_dtype='f32', cfg=cfg, proposals=args.proposals,
                from_host=False, pps=args.pairs_per_step, detail=False, hbm_detail=False, points=args.points,
                tau=args.tau, n_boxes=args.boxes, computed=computed, keep_records=0):
        """`steps` timed steps of the pipeline built for conv_dtype; with detail, then the conv
        stacks alone layer by layer (roofline); with hbm_detail the HBM-bound kernels alone.
        keep_records: download the records of that many of the last steps (for the host's temporal module)."""
        fps = cfg['frames_per_sample']
        feat_c = 256 if cfg['extractor'] == 'vgg' else 32
        pipe = FramePairPipeline(ctx, cfg, **synth.pipeline_weights(cfg),
                                 n_points_max=points, rpn_nms_size=proposals,
                                 pairs_per_step=pps,
                                 head_params=synth.head_params(feat=feat_c) if computed else None,
                                 conv_dtype=conv_dtype, head_dtype=head_dtype,
                                 reuse_streams_of=made[0] if made and min(made[0].nf, 2) == min(fps * pps, 2) else None)
        made.append(pipe)
        # The exchange step ships G consecutive steps' records in one all-gather: the pipeline
        # writes step k into slot k % 2G of a contiguous ring, and a half of the ring (a block of G
        # steps) is one message.  gathered[b]: every rank's block, rank major.
        G = max(1, args.gather_every)
        rec_ring = ctx.zeros((2 * G, pps, fps, MAX_DET, REC_COLS), np.float32)
        cnt_ring = ctx.zeros((2 * G, pps, fps), np.int32)
        pipe.use_record_ring(rec_ring, cnt_ring)
        nr, nc = 4 * pps * fps * MAX_DET * REC_COLS, 4 * pps * fps
        blocks = [(rec_ring.offset(b * G * nr, (G * pps, fps, MAX_DET, REC_COLS)),
                   cnt_ring.offset(b * G * nc, (G * pps, fps), np.int32)) for b in range(2)]
        gathered = [ctx.zeros((world * G * pps, fps, MAX_DET, REC_COLS), np.float32) for _ in range(2)]
        gathered_cnt = [ctx.zeros((world * G * pps, fps), np.int32) for _ in range(2)]
        if comm is not None:
            # the tail that refills a slot first joins the gather that last read its block
            pipe.on_records_reuse = lambda slot, sides: [comm.join(slot // G, s) for s in sides]

        # a small ring of distinct synthetic batches, resident in HBM before timing starts;
        # every pair of a batch comes from a different sequence (they are independent)
        # (three per rank: at N = 8 the ranks together draw 24 sequences -- BASELINE.json configs[3]'s
        #  "synthetic 21-sequence stream")
        n_batches = 3
        batches = []
        for i in range(n_batches):
            pts, imgs, heads = [], [], []
            for j in range(pps):
                seq = (rank * n_batches + i) * pps + j
                for f in ((tau * i, tau * i + tau) if fps == 2 else (tau * i,)):
                    pts.append(synth.lidar_frame(seq, f, points, n_boxes))
                    imgs.append(ctx.array(synth.image_frame(seq, f)))
                    if not computed:
                        heads.append({k: ctx.array(v) for k, v in
                                      synth.head_outputs(seq, f, pipe.n_all, pipe.P).items()})
            b = dict(pts=[ctx.array(p) for p in pts], n=[len(p) for p in pts],
                     imgs=imgs, heads=None if computed else heads)
            if from_host:   # the same frames in page-locked host memory (PCIe-inclusive run)
                b['h_pts'], b['h_imgs'] = [], []
                for p_, d_img in zip(pts, imgs):
                    hp = ctx.pinned((points, 4), np.float32)
                    hp.a[:len(p_)] = p_
                    hi = ctx.pinned(d_img.shape, np.uint8)
                    hi.a[...] = d_img.download()
                    b['h_pts'].append(hp)
                    b['h_imgs'].append(hi)
            batches.append(b)

        state = {'sent': 0, 'last_block': None}     # steps whose records have been shipped
        # Which stream carries the collectives (measured with a late peer injected): with
        # peers (world > 1) the communicator's OWN stream -- a rank that arrives late at a gather then holds that
        # stream only, not a frame's prep and tail; with one rank (--comm) frame 1's side stream, which saves the
        # fifth stream.  DODT_BENCH_COMM_STREAM=own|side overrides.
        if comm is not None and comm_stream == 'side':
            comm.attach(pipe.sides[-1])

        def gather_block(b):
            # on the communicator's side stream, behind what the main stream holds so far (the
            # records of the block's steps are complete there: FramePairPipeline.run / finish);
            # nothing waits for it until the block's first slot is written again, G + 1 steps later
            if comm is not None:
                comm.all_gather_records(ctx, b, blocks[b][0], blocks[b][1], gathered[b], gathered_cnt[b])
                state['last_block'] = b

        def ship(complete):      # records of steps < complete are complete on the main stream
            while state['sent'] + G <= complete:
                gather_block((state['sent'] // G) % 2)
                state['sent'] += G

        # the stream of frames is known ahead: a step may hand the pipeline the next step's inputs as well, whose prep
        # then goes in front of the previous step's tail (FramePairPipeline.run: lookahead).  Measured (same box,
        # gpurun_out/r4_exp1.txt and DESIGN.md section 8): +2.4 % with the bf16 conv path, whose convs would otherwise
        # wait for that tail, -0.5 % with the fp32 convs, whose step is not bound there -- so it follows the conv
        # arithmetic; DODT_BENCH_LOOKAHEAD=0 / 1 forces it
        ahead = os.environ.get('DODT_BENCH_LOOKAHEAD', '1' if conv_dtype == 'bf16' else '0') != '0'

        def step(i):
            p = batches[pipe.step_idx % n_batches]
            q = batches[(pipe.step_idx + 1) % n_batches]
            if from_host:
                pipe.run_from_host(p['h_pts'], p['n'], p['h_imgs'], p['heads'],
                                   lookahead=(q['h_pts'], q['n'], q['h_imgs']) if ahead else None)
            else:
                pipe.run(p['pts'], p['n'], p['imgs'], p['heads'], lookahead=(q['pts'], q['n'], q['imgs']) if ahead else None)
            ship(pipe.step_idx - 1)

        def drain():
            pipe.finish()
            ship(pipe.step_idx)
            if state['sent'] < pipe.step_idx:       # a block that is not full yet: ship it as it is
                gather_block((state['sent'] // G) % 2)

        def barrier():
            arm(300, 'a barrier')
            ctx.sync()                  # finish() joined every stream of the pipeline into this one
            if comm is not None:
                comm.barrier()          # drains the side stream, then all ranks meet
            elif host_sync is not None:
                host_sync.barrier()
            disarm()

        # steps are pipelined two deep inside pipe.run(); finish() drains the last one, so
        # exactly `steps` complete steps (convs AND tails) lie inside the timed region
        # Set-up steps in front of the W warm-up steps, untimed like them and reported as `setup_steps`: the first ~15 steps of a
        # fresh pipeline run 5-10 % slow (clock ramp, first touches of code and buffers), and a caller that
        # asks for a short warm-up (the driver: W = 5) would otherwise time the tail of that ramp
        setup_steps = max(0, 15 - warmup)
        arm(300 + 0.1 * (setup_steps + warmup + steps), 'the warm-up steps')
        for i in range(setup_steps + warmup):
            step(i)
        drain()
        barrier()
        arm(300 + 0.1 * steps, 'the timed steps')
        # one HIP event per step on the main stream: event i fires when step i's convs and step
        # i-1's tail are done, so the deltas show a slow fill / drain step or a clock ramp that
        # the single wall-clock window hides
        n_ev = min(steps, 240)      # (mark slots 250.. of the main context belong to the pipeline)
        host_ms = []
        t0 = time.perf_counter()
        ctx.mark(0)
        for i in range(steps):
            th = time.perf_counter()
            step(i)
            if i < n_ev:
                ctx.mark(i + 1)
            host_ms.append((time.perf_counter() - th) * 1e3)
        host_enqueue_ms = (time.perf_counter() - t0) / steps * 1e3   # host side of a step
        drain()
        ctx.mark(n_ev + 1)
        barrier()
        elapsed = time.perf_counter() - t0
        step_ms = [ctx.elapsed_ms(i, ctx, i + 1) for i in range(n_ev + 1)]
        arm(300, 'the max over ranks')
        if comm is not None:
            elapsed = comm.max_over_ranks(elapsed)
        elif host_sync is not None:
            elapsed = host_sync.max_over_ranks(elapsed)
        disarm()
        res = dict(elapsed=elapsed, host_enqueue_ms=host_enqueue_ms, step_ms=step_ms, host_ms=host_ms,
                   flops=pipe.flops_per_step(), mfma_flops=pipe.mfma_flops_per_step(),
                   head_gflop=pipe.head_flops_per_step() / 1e9, conv_bytes=pipe.conv_bytes_per_step(),
                   anchors=list(pipe.last_anchor_counts), steps=steps, pps=pps, setup_steps=setup_steps)
        if keep_records:
            # what the temporal module gets: the last steps' records in step order (slot k % 2G of the ring)
            ring, cring = rec_ring.download(), cnt_ring.download()
            res['records'] = []
            for k_ in range(max(0, pipe.step_idx - min(keep_records, 2 * G)), pipe.step_idx):
                for j in range(pps):
                    c_ = cring[k_ % (2 * G), j]
                    res['records'].append(np.concatenate([ring[k_ % (2 * G), j, f, :c_[f]] for f in range(fps)]))
        if comm is not None and rank == 0 and state['last_block'] is not None:
            # the exchange really happened: rank 0's own part of the last message equals its records
            b = state['last_block']
            g = gathered[b].download()[rank * G * pps:(rank + 1) * G * pps]
            res['gather_ok'] = bool(np.array_equal(g, blocks[b][0].download()) and np.abs(g).max() > 0)
            if not res['gather_ok']:
                raise RuntimeError("the all-gather's output does not hold rank 0's own records: the exchange step "
                                   'did not do its work')

        # ---- the conv stacks alone (each net by itself on its own stream, so that kernel
        #      durations do not overlap), layer by layer: HIP events on the stream the kernels run on
        reps = max(3, min(steps, 10))
        # (the stand-alone forwards read the inputs the last steps left in the pipeline's buffers)
        pipe.bev_net.set_input(pipe.in_bev[0])
        pipe.img_net.set_input(pipe.in_img[0])
        nets = ((pipe.bev_net, ctx, pipe.feat[0]['bev_feat'], pipe.feat[0]['bev_bneck']),
                (pipe.img_net, pipe.img_ctx, pipe.feat[0]['img_feat'], pipe.feat[0]['img_bneck']))
        conv_ms = 0.0
        for net, c, f, b in nets:
            c.sync()
            c.timer_start()
            for _ in range(reps):
                net.forward_device(None, f, b)
            conv_ms += c.timer_stop() / reps
        res.update(conv_ms=conv_ms, reps=reps)
        if detail:
            layers = []
            for net, c, f, b in nets:
                acc = None
                for _ in range(reps):
                    cur = net.forward_timed(None, f, b)
                    if acc is None:
                        acc = cur
                    else:
                        for a_, c_ in zip(acc, cur):
                            a_['ms'] += c_['ms']
                for a_ in acc:
                    a_['ms'] /= reps
                layers += acc
            res['layers'] = layers
            # both nets side by side, as in the timed steps (but nothing else on the GPU)
            ctx.sync()
            pipe.img_ctx.sync()
            ctx.timer_start()
            for _ in range(reps):
                pipe.img_ctx.wait_for(ctx)
                pipe.bev_net.forward_device(None, pipe.feat[0]['bev_feat'], pipe.feat[0]['bev_bneck'])
                pipe.img_net.forward_device(None, pipe.feat[0]['img_feat'], pipe.feat[0]['img_bneck'])
                ctx.wait_for(pipe.img_ctx)
            res['both_ms'] = ctx.timer_stop() / reps
        # ---- the HBM-bound kernels alone, on the inputs the last step left behind -----------
        if hbm_detail and fps == 2 and computed and cfg['extractor'] != 'vgg':
            hreps = 20
            fr, feat = pipe.fr, pipe.feat[(pipe.step_idx - 1) & 1]
            n_pts = batches[0]['n'][0]

            def timed(fn):
                fn()
                ctx.sync()
                ctx.timer_start()
                for _ in range(hreps):
                    fn()
                return ctx.timer_stop() / hreps * 1e3      # us
            bev_hw, FC = (pipe.bev_fh, pipe.bev_fw), pipe.feat_c
            feat_b0 = feat['bev_feat'].offset(0, bev_hw + (FC,))
            feat_b1 = feat['bev_feat'].offset(4 * pipe.bev_fh * pipe.bev_fw * FC, bev_hw + (FC,))
            d_bev = ctx.empty((pipe.bev_h, pipe.bev_w, cfg['bev_depth']), np.float32)
            n_top = int(fr[0]['top_count'].download()[0])
            out_b = n_top * ROI * ROI * FC * 4
            hbm = []
            us = timed(lambda: ops.bev_slices(ctx, batches[0]['pts'][0], n_pts, pipe.bp, d_bev, fr[0]['occ']))
            hbm.append(dict(kernel='hipMemsetAsync + vox_scatter + vox_finalize',
                            stage='a0-a3 voxeliser, %d points -> (700,800,6) maps' % n_pts,
                            algorithmic_bytes=16 * n_pts + pipe.bev_h * pipe.bev_w * cfg['bev_depth'] * 4,
                            us=us))
            us = timed(lambda: ops.crop_and_resize(ctx, feat_b0, bev_hw + (FC,), fr[0]['top_bev'], pipe.P,
                                                   fr[0]['top_count'], (ROI, ROI), fr[0]['bev_rois']))
            hbm.append(dict(kernel='crop_kernel<4>', stage='a11 stage-2 ROI crop, %d proposals x 7x7x%d, BEV map'
                            % (n_top, FC),
                            algorithmic_bytes=16 * n_top + min(4 * out_b, pipe.bev_fh * pipe.bev_fw * FC * 4) + out_b,
                            us=us))
            corr_map = pipe.head_scratch[0]['corr_map']
            us = timed(lambda: ops.correlation(ctx, feat_b0, feat_b1, bev_hw + (FC,), CORR_MAX_DISP,
                                               CORR_STRIDE2, CORR_PAD, corr_map))
            hbm.append(dict(kernel='correlation_sp_kernel', stage='f1 correlation of the pair\'s BEV features',
                            algorithmic_bytes=(2 * FC + CORR_CH) * pipe.bev_fh * pipe.bev_fw * 4, us=us))
            # NMS #1 alone on the last step's candidates (n(n-1)/2 pair tests, bytes =
            # 20 n + 8 n ceil(n/64); latency-bound, the bandwidth fraction is informative only)
            n_c = int(pipe.last_anchor_counts[0])
            d_sel, d_cnt = ctx.empty((pipe.P,), np.int32), ctx.zeros((1,), np.int32)
            us = timed(lambda: ops.nms(ctx, fr[0]['prop_bev'], fr[0]['scores'], n_c, None, pipe.P,
                                       cfg['rpn_nms_iou_thresh'], d_sel, d_cnt))
            hbm.append(dict(kernel='nms_*', stage='a13 NMS #1, %d candidates -> %d at IoU %.2f'
                            % (n_c, int(d_cnt.download()[0]), cfg['rpn_nms_iou_thresh']),
                            algorithmic_bytes=20 * n_c + 8 * n_c * ((n_c + 63) // 64), us=us,
                            pair_tests=n_c * (n_c - 1) // 2))
            for h in hbm:
                h['us'] = round(h['us'], 2)
                h['gbps'] = round(h['algorithmic_bytes'] / h['us'] / 1e3, 1)
                h['frac_of_hbm_peak'] = round(h['gbps'] / HBM_PEAK, 4)
                if 'pair_tests' in h:
                    h['pair_tests_per_s'] = float('%.4g' % (h['pair_tests'] / (h['us'] * 1e-6)))
            res['hbm'] = hbm
        pipe.close()
        return res

    import gc
    if os.environ.get('DODT_BENCH_GC', 'freeze') == 'freeze':
        # everything allocated so far (numpy, the package) leaves the collector's generations: a
        # full collection of that heap in the middle of a 0.1 s timed window would stall the
        # enqueueing thread for tens of ms
        gc.collect()
        gc.freeze()
    pps = args.pairs_per_step
    m = measure(args.conv_dtype, args.steps, args.warmup, args.head_dtype, detail=True, hbm_detail=True,
                keep_records=16)
    elapsed, host_enqueue_ms, conv_ms, reps = m['elapsed'], m['host_enqueue_ms'], m['conv_ms'], m['reps']
    alt = None
    if not args.no_alt and world == 1:      # (N > 1 runs measure the sharded path only)
        # other arithmetics / batchings of the same workload, shorter runs: reported beside the
        # main measurement, never part of `value`
        # (at least 60 steps behind 10 warm-up steps each: round 3's alt runs were 10 steps behind 2 and read 8 % low)
        k = max(60, args.steps)

        def short(conv_dtype, head_dtype, **kw):
            a = measure(conv_dtype, k, 10, head_dtype, **kw)
            return {'conv_dtype': conv_dtype, 'head_dtype': head_dtype,
                    'pairs_per_step': a['pps'],
                    'value': round(world * k * a['pps'] / a['elapsed'], 3), 'unit': 'frame-pairs/s',
                    'steps': k, 'ms_per_step': round(a['elapsed'] / k * 1e3, 4),
                    'conv_stacks_tflops': round(a['flops'] / (a['conv_ms'] * 1e-3) / 1e12, 2),
                    'conv_stacks_ms': round(a['conv_ms'], 4)}
        alt = {'note': "f32s = split mode: hi + lo bf16 pairs, three bf16 MFMAs per product term, "
                       'fp32 accumulate -- passes the fp32 layer tests at 1e-4 '
                       '(tests/test_gpu_conv_split.py); bf16 conv = BASELINE.json configs[2]\'s '
                       'bf16 conv path (bars in tests/test_gpu_conv_bf16.py); bf16 heads = the '
                       "same for the FC layers; f32 = the reference's arithmetic on the fp32 MFMA; "
                       'pairs_per_step > 1 = that many independent frame pairs batched through every launch',
               'runs': [short(c, h) for c, h in (('f32', 'f32'), ('f32s', 'f32'), ('bf16', 'f32'),
                                                  ('bf16', 'bf16'))
                        if (c, h) != (args.conv_dtype, args.head_dtype) and (computed or h == 'f32')]}
        if args.config == 'dodt':
            alt['batched'] = [short(c, h, pps=n) for c, h in (('f32', 'f32'), ('bf16', 'bf16'))
                              for n in (2, 4) if computed and n != pps]
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
            if computed:
                # BASELINE.json configs[4]: tau = 3, ~300k points per frame, 4096 proposals, heads computed
                # (S + T), with its own HBM-side kernels alone
                a = measure(args.conv_dtype, k, 10, args.head_dtype, points=300000, proposals=4096, tau=3,
                            n_boxes=40, hbm_detail=True)
                dj = _profile_json('dense_hbm_traffic.json')
                for h in a.get('hbm', []):
                    t = dj['kernels'].get(h['kernel']) if dj else None
                    h['traffic'] = round(t['fetch_bytes'] + t['write_bytes']) if t else None
                alt['dense_scene'] = {
                    'config': 'BASELINE.json configs[4] on one GPU: tau = 3, 2 x 300k points (40 boxes), 4096 proposals, '
                              'S+T path, %s convs / %s heads' % (args.conv_dtype, args.head_dtype),
                    'value': round(k / a['elapsed'], 3), 'unit': 'frame-pairs/s', 'steps': k,
                    'ms_per_step': round(a['elapsed'] / k * 1e3, 4), 'anchors_kept': a['anchors'],
                    'head_gflop_per_step': round(a['head_gflop'], 2), 'hbm': a.get('hbm'),
                    'traffic_head': dj.get('head') if dj else None,
                    'traffic_source': dj.get('source') if dj else None}
            # BASELINE.json configs[1]: tau = 1 Siamese (S) path, heads' outputs injected from HBM
            a = measure(args.conv_dtype, k, 10, 'f32', tau=1, computed=False)
            alt['s_path'] = {'config': 'BASELINE.json configs[1]: DODT tau = 1 Siamese (S), batch = 1 frame pair, '
                                       '%s; correlation branch and dense heads NOT run, their outputs injected' % args.conv_dtype,
                             'value': round(k / a['elapsed'], 3), 'unit': 'frame-pairs/s', 'steps': k,
                             'ms_per_step': round(a['elapsed'] / k * 1e3, 4)}
            if args.sustained_steps > 0:
                # the headline workload over seconds instead of a fraction of one: steady-state clocks and
                # temperature, step-time percentiles from the host's per-step enqueue-to-enqueue times (the host
                # runs one step ahead of the GPU, so over a long run they are the GPU's step times)
                n_s = args.sustained_steps
                a = measure(args.conv_dtype, n_s, args.warmup, args.head_dtype)
                ev = np.asarray(a['step_ms'][:-1])
                hs = np.asarray(a['host_ms'][len(a['host_ms']) // 10:])
                alt['sustained'] = {
                    'config': 'the headline workload, %d timed steps' % n_s,
                    'value': round(n_s * a['pps'] / a['elapsed'], 3), 'unit': 'frame-pairs/s', 'steps': n_s,
                    'seconds': round(a['elapsed'], 3), 'ms_per_step': round(a['elapsed'] / n_s * 1e3, 4),
                    'step_ms_percentiles_host': {q: round(float(np.percentile(hs, p_)), 3) for q, p_ in
                                                 (('p01', 1), ('p50', 50), ('p90', 90), ('p99', 99), ('max', 100))},
                    'step_ms_first_%d_events' % len(ev): {'median': round(float(np.median(ev)), 3),
                                                          'max': round(float(ev.max()), 3)},
                    'ms_per_step_by_quarter': [round(float(np.mean(q_)), 4) for q_ in np.array_split(np.asarray(a['host_ms']), 4)]}
            if computed and m.get('records'):
                alt['temporal_host'] = temporal_host(m_pool, m_cores, m['records'], args.tau,
                                                     world * args.steps * pps / elapsed)
        if m_pool is not None:
            m_pool.close()
            m_pool.join()

    # ---- roofline of the dominant kernel ---------------------------------------------------------
    # Everything below is measured in this run with HIP events on the streams the kernels run on;
    # only the PMC byte counts come from the rocprofv3 --pmc passes over this same command
    # (profiles/), because counters cannot be read from inside the process.
    by_kernel = _group_by_kernel(m['layers'])
    dom = max(by_kernel.values(), key=lambda k_: k_['ms'])
    n_launch = sum(k_['launches'] for k_ in by_kernel.values())
    layers_ms = sum(k_['ms'] for k_ in by_kernel.values())
    peak = FP32_MFMA_PEAK if args.conv_dtype == 'f32' else BF16_MFMA_PEAK
    # (written by tools/summarize_profile.py with the head they were taken at: `traffic_head`)
    tj = _profile_json('conv_traffic_%s.json' % args.conv_dtype)
    traffic = tj['kernels'].get(dom['kernel']) if tj else None
    kernels = []
    for k_ in sorted(by_kernel.values(), key=lambda k_: -k_['ms']):
        tf = k_['flops_executed'] / (k_['ms'] * 1e-3) / 1e12
        kernels.append(dict(kernel=k_['kernel'], launches_per_step=k_['launches'], ms_per_step=round(k_['ms'], 4),
                            avg_launch_us=round(k_['ms'] * 1e3 / k_['launches'], 2),
                            executed_gflop=round(k_['flops_executed'] / 1e9, 2), executed_tflops=round(tf, 2),
                            frac=round(tf / peak, 4),
                            direct_equivalent_tflops=round(k_['flops_direct'] / (k_['ms'] * 1e-3) / 1e12, 2),
                            algorithmic_gbps=round(k_['bytes'] / (k_['ms'] * 1e-3) / 1e9, 1),
                            layers=k_['layers']))
    d = kernels[0]
    all_tf = m['mfma_flops'] / (layers_ms * 1e-3) / 1e12
    roofline = dict(
        bound='mfma', kernel=d['kernel'], launches_per_step=d['launches_per_step'],
        avg_launch_us=d['avg_launch_us'],
        # executed MFMA FLOPs of the dominant kernel's launches / their stand-alone duration
        achieved=d['executed_tflops'], peak=peak, unit='TFLOP/s', frac=d['frac'],
        algorithmic_gflop_per_launch=round(d['executed_gflop'] / d['launches_per_step'], 3),
        direct_equivalent_tflops=d['direct_equivalent_tflops'],
        traffic=(round(traffic['fetch_bytes_per_launch'] + traffic['write_bytes_per_launch'])
                 if traffic else None),
        traffic_unit='bytes/launch', traffic_source=tj['source'] if tj else None,
        traffic_head=tj.get('head') if tj else None,
        algorithmic_bytes_per_launch=round(by_kernel[d['kernel']]['bytes'] / d['launches_per_step']),
        arithmetic={'f32': 'fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2), peak 157.3 TFLOP/s dense',
                    'f32s': 'three bf16 MFMAs per product term (split mode), fp32 accumulate',
                    'bf16': 'bf16 MFMA, fp32 accumulate'}[args.conv_dtype],
        measured='HIP event pair around every layer, each net alone on its stream, %d forwards after the '
                 'timed region; frac = executed MFMA FLOPs / time / peak' % reps,
        conv_stacks=dict(launches_per_step=n_launch, ms=round(conv_ms, 4), layers_ms=round(layers_ms, 4),
                         executed_gflop=round(m['mfma_flops'] / 1e9, 2), executed_tflops=round(all_tf, 2),
                         frac=round(all_tf / peak, 4),
                         direct_gflop=round(m['flops'] / 1e9, 2),
                         direct_equivalent_tflops=round(m['flops'] / (conv_ms * 1e-3) / 1e12, 2),
                         algorithmic_mbytes=round(m['conv_bytes'] / 1e6, 1),
                         side_by_side_ms=round(m['both_ms'], 4),
                         side_by_side_direct_equivalent_tflops=round(m['flops'] / (m['both_ms'] * 1e-3) / 1e12, 2)),
        kernels=kernels,
        # (a layer with no launch of its own -- conv1_1 folded into conv1_2's launch on the bf16 conv path -- lists its
        #  FLOPs with `us` null: its time is inside the next row's)
        layers=[dict(net='bev' if i < len(m['layers']) // 2 else 'img', name=l['name'], kernel=l['kernel'],
                     items=l['items'], us=round(l['ms'] * 1e3, 1) if l['launches'] else None,
                     executed_tflops=round(l['flops_executed'] / (l['ms'] * 1e-3) / 1e12, 1) if l['launches'] else None,
                     **({} if l['launches'] else {'folded_into_next': True}))
                for i, l in enumerate(m['layers'])])
    hbm = m.get('hbm')
    if hbm:
        hj = _profile_json('hbm_traffic.json')
        for h in hbm:
            t = hj['kernels'].get(h['kernel']) if hj else None
            h['traffic'] = round(t['fetch_bytes'] + t['write_bytes']) if t else None
            h['traffic_source'] = hj['source'] if (hj and t) else None
            h['traffic_head'] = hj.get('head') if (hj and t) else None
        roofline['hbm'] = hbm

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            'metric': 'frame-pairs/sec (whole node) KITTI-shape tau=2',
            'value': round(world * args.steps * pps / elapsed, 3),
            'unit': 'frame-pairs/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': args.warmup, 'setup_steps': m['setup_steps'], 'ms_per_step': round(ms, 4),
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
            'config': {'workload': ('DODT tau=%d frame pair: 2 x %dk pts + 2 x 1242x375 RGB, '
                                    'pyramid_cars_with_aug_dt_5_tracking (box_4ca), %d proposals, '
                                    '%s' % (args.tau, args.points // 1000, args.proposals,
                                            'S+T path: correlation + dense heads on the device'
                                            if computed else 'S path (heads injected)'))
                       if args.config == 'dodt' else
                       ('AVOD single frame: %dk pts + 1242x375 RGB, avod_cars_example (plain '
                        'VGG extractors, box_4ca), %d proposals; value counts FRAMES/s'
                        % (args.points // 1000, args.proposals)),
                       'head_gflop_per_step': round(m['head_gflop'], 2),
                       'pairs_per_step_per_gpu': pps, 'parallelism': 'pair-shard x%d' % world,
                       'exchange': ('RCCL all-gather of (%d steps x pairs,2,100,17) f32 + counts every %d steps, on %s, '
                                    'C-ABI (no PyTorch)%s%s' % (max(1, args.gather_every), max(1, args.gather_every),
                                                               "the communicator's own stream" if comm_stream == 'own'
                                                               else "frame 1's side stream",
                                                               "; rank 0's part verified" if m.get('gather_ok') else '',
                                                               '; a %.1f ms late peer injected in front of every gather'
                                                               % args.late_peer_ms if args.late_peer_ms > 0 else '')
                                    ) if comm is not None else ('none (one rank)' if comm_error is None else
                                                               'FAILED, no records exchanged, host barrier only: ' + comm_error),
                       'conv_mode': os.environ.get('DODT_CONV_WINO', 'default'),
                       'anchors_kept': m['anchors']},
            'roofline': roofline,
        }
        if alt is not None:
            out['alt'] = alt
        if baseline is not None:
            out['cpu_baseline'] = baseline
        print(json.dumps(out))
        sys.stdout.flush()
    arm(600, 'the closing barrier')
    if comm is not None:
        comm.barrier()
        comm.close()
    if host_sync is not None:
        host_sync.close()
    disarm()


if __name__ == '__main__':
    main()
