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


@pytest.mark.parametrize('attach', [False, True])
def test_one_rank_communicator_gathers_pipeline_records(tmp_path, attach):
    """(attach: the collectives on frame 1's side stream instead of a stream of the communicator's own --
    bench.py's default since a fifth stream costs 2 % of the pairs/s.)
    bench.py's pattern: the pipeline writes step k into slot k % 2G of a contiguous ring, a block of
    G steps is one grouped all-gather on the side stream, the tail that refills a slot joins the
    gather that last read its block."""
    ctx = device.default_context()
    comm = sharding.Communicator(ctx, 0, 1, id_path=str(tmp_path / 'id'))
    pipe = FramePairPipeline(ctx, C, **synth.pipeline_weights(C), rpn_nms_size=300, n_points_max=40000)
    G, steps = 2, 7
    rec_ring = ctx.zeros((2 * G, 1, 2, MAX_DET, REC_COLS), np.float32)
    cnt_ring = ctx.zeros((2 * G, 1, 2), np.int32)
    pipe.use_record_ring(rec_ring, cnt_ring)
    if attach:
        comm.attach(pipe.sides[-1])
    nr, nc = 4 * 2 * MAX_DET * REC_COLS, 4 * 2
    blocks = [(rec_ring.offset(b * G * nr, (G, 2, MAX_DET, REC_COLS)),
               cnt_ring.offset(b * G * nc, (G, 2), np.int32)) for b in range(2)]
    gathered = [ctx.zeros((G, 2, MAX_DET, REC_COLS), np.float32) for _ in range(2)]
    gathered_cnt = [ctx.zeros((G, 2), np.int32) for _ in range(2)]
    joins = []
    pipe.on_records_reuse = lambda slot, sides: [joins.append(slot) or comm.join(slot // G, s) for s in sides]
    per_step, shipped, sent = [], [], 0
    for k in range(steps + 1):
        if k < steps:
            pts = [synth.lidar_frame(7 + k, f, 40000) for f in (0, 2)]
            heads = [{n: ctx.array(v) for n, v in synth.head_outputs(7 + k, f, pipe.n_all, pipe.P).items()}
                     for f in (0, 2)]
            pipe.run([ctx.array(p) for p in pts], [len(p) for p in pts],
                     [ctx.array(synth.image_frame(7 + k, f)) for f in (0, 2)], heads)
            complete = pipe.step_idx - 1
        else:
            pipe.finish()
            complete = pipe.step_idx
        if complete > len(per_step):       # the records of step complete-1 are final on the main stream
            ctx.sync()
            per_step.append((pipe.d_records.download().copy(), pipe.d_rec_counts.download().copy()))
        while sent + G <= complete:
            b = (sent // G) % 2
            comm.all_gather_records(ctx, b, blocks[b][0], blocks[b][1], gathered[b], gathered_cnt[b])
            comm.sync()
            shipped.append((sent, gathered[b].download().copy(), gathered_cnt[b].download().copy()))
            sent += G
    comm.barrier()
    assert len(per_step) == steps and [s0 for s0, _, _ in shipped] == [0, 2, 4]
    assert sorted(set(joins)) == [0, 1, 2, 3]
    for first, g, gc in shipped:
        merged = sharding.merge_block(g, gc, first, G, 1, 1)
        assert [(p, f) for p, f, _ in merged] == [(first + i, f) for i in range(G) for f in range(2)]
        for p, f, rec in merged:
            want_rec, want_cnt = per_step[p]
            assert len(rec) == want_cnt[0, f] > 0
            assert np.array_equal(rec, want_rec[0, f, :len(rec)]) and np.abs(rec).max() > 0
    assert comm.max_over_ranks(3.25) == 3.25
    with pytest.raises(ValueError):
        comm.all_gather_records(ctx, 9, blocks[0][0], blocks[0][1], gathered[0], gathered_cnt[0])
    with pytest.raises(ValueError):
        comm.all_gather_records(ctx, 0, blocks[0][0], blocks[0][1], gathered_cnt[0], gathered_cnt[0])
    with pytest.raises(ValueError):
        pipe.use_record_ring(gathered[0], cnt_ring)
    pipe.close()
    comm.close()
