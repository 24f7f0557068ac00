"""The exchange step of the multi-GPU path on real hardware: an RCCL communicator
made through the C-ABI (dodt_comm_*, no PyTorch) gathers the detection records a pipeline step
produced.  The GPU box has one card, so the communicator has one rank: what runs is RCCL's
ncclCommInitRank / grouped ncclAllGather / ncclAllReduce on the side stream with the event
hand-offs of the real pattern; the N > 1 merge logic is covered on CPU over gloo
(tests/test_sharding_gloo.py).  Needs an MI355X."""
import numpy as np
import pytest

from dodt_amd import config, device, sharding, synth
from dodt_amd.pipeline import MAX_DET, REC_COLS, FramePairPipeline

pytestmark = pytest.mark.gpu
C = config.PYRAMID_DODT


def test_one_rank_communicator_gathers_pipeline_records(tmp_path):
    ctx = device.default_context()
    comm = sharding.Communicator(ctx, 0, 1, id_path=str(tmp_path / 'id'))
    pipe = FramePairPipeline(ctx, C, **synth.pipeline_weights(C), rpn_nms_size=300, n_points_max=40000)
    pipe.on_records_reuse = lambda par, sides: [comm.join(par, s) for s in sides]
    gathered = [ctx.zeros((1, 2, MAX_DET, REC_COLS), np.float32) for _ in range(2)]
    gathered_cnt = [ctx.zeros((1, 2), np.int32) for _ in range(2)]
    steps, kept = 4, []
    prev = None
    for k in range(steps):
        pts = [synth.lidar_frame(7 + k, f, 40000) for f in (0, 2)]
        heads = [{n: ctx.array(v) for n, v in synth.head_outputs(7 + k, f, pipe.n_all, pipe.P).items()}
                 for f in (0, 2)]
        par = pipe.run([ctx.array(p) for p in pts], [len(p) for p in pts],
                       [ctx.array(synth.image_frame(7 + k, f)) for f in (0, 2)], heads)
        if prev is not None:      # the previous step's records are complete on the main stream
            comm.all_gather_records(ctx, prev, pipe.rec2[prev], pipe.cnt2[prev], gathered[prev],
                                    gathered_cnt[prev])
            comm.sync()
            kept.append((gathered[prev].download().copy(), gathered_cnt[prev].download().copy(),
                         pipe.rec2[prev].download().copy(), pipe.cnt2[prev].download().copy()))
        prev = par
    pipe.finish()
    comm.all_gather_records(ctx, prev, pipe.rec2[prev], pipe.cnt2[prev], gathered[prev], gathered_cnt[prev])
    comm.barrier()
    kept.append((gathered[prev].download(), gathered_cnt[prev].download(), pipe.rec2[prev].download(),
                 pipe.cnt2[prev].download()))
    assert len(kept) == steps
    for g, gc, r, c in kept:
        assert np.array_equal(g, r) and np.array_equal(gc, c)
        assert c.min() > 0 and np.abs(r).max() > 0          # real detections, not an empty buffer
    merged = sharding.merge_step(kept[-1][0], kept[-1][1], steps - 1, 1, 1)
    assert [(p, f) for p, f, _ in merged] == [(steps - 1, 0), (steps - 1, 1)]
    assert comm.max_over_ranks(3.25) == 3.25
    with pytest.raises(ValueError):
        comm.all_gather_records(ctx, 9, pipe.rec2[0], pipe.cnt2[0], gathered[0], gathered_cnt[0])
    with pytest.raises(ValueError):
        comm.all_gather_records(ctx, 0, pipe.rec2[0], pipe.cnt2[0], gathered_cnt[0], gathered_cnt[0])
    pipe.close()
    comm.close()
