"""N > 1 path on CPU: two gloo ranks shard a pair stream round-robin, all-gather their
detection records and reassemble them in sequence order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dodt_amd import sharding


def _fake_records(pair_id):
    """Deterministic stand-in for a pair's detections (the GPU pipeline is not
    needed to test the exchange)."""
    rng = np.random.default_rng(1000 + pair_id)
    cnt = rng.integers(0, sharding.MAX_DET + 1, size=2).astype(np.int32)
    rec = np.zeros((2, sharding.MAX_DET, sharding.REC_COLS), np.float32)
    for f in range(2):
        rec[f, :cnt[f]] = rng.normal(size=(cnt[f], sharding.REC_COLS))
        rec[f, :cnt[f], 16] = f
    return rec, cnt


def _worker(rank, world, port, pps, steps, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    merged = []
    for step in range(steps):
        recs, cnts = zip(*[_fake_records(p) for p in
                           sharding.step_pairs(step, pps, rank, world)])
        rec = torch.from_numpy(np.stack(recs))
        cnt = torch.from_numpy(np.stack(cnts))
        g = torch.zeros((world * pps,) + tuple(rec.shape[1:]), dtype=torch.float32)
        gc = torch.zeros((world * pps,) + tuple(cnt.shape[1:]), dtype=torch.int32)
        sharding.all_gather_records(dist, rec, cnt, g, gc)
        merged += sharding.merge_step(g.numpy(), gc.numpy(), step, pps, world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)       # the bench's max-over-ranks timing
    dist.barrier()
    q.put((rank, [(p, f, r.copy()) for p, f, r in merged], float(t.item())))
    dist.destroy_process_group()


def _block_worker(rank, world, port, pps, G, blocks, q):
    """The layout bench.py ships: a rank's records of G consecutive steps are ONE message, step major
    (the pipeline writes step k into slot k % 2G of a contiguous ring; dodt_amd/pipeline.py), gathered
    rank major; sharding.merge_block puts them back into global pair order."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    merged = []
    for blk in range(blocks):
        rec = np.zeros((G, pps, 2, sharding.MAX_DET, sharding.REC_COLS), np.float32)
        cnt = np.zeros((G, pps, 2), np.int32)
        for i in range(G):
            for j, pid in enumerate(sharding.step_pairs(blk * G + i, pps, rank, world)):
                rec[i, j], cnt[i, j] = _fake_records(pid)
        # the exchange sees (G * pps, 2, ...) per rank, exactly the shape Communicator.all_gather_records takes
        g = torch.zeros((world * G * pps, 2, sharding.MAX_DET, sharding.REC_COLS), dtype=torch.float32)
        gc = torch.zeros((world * G * pps, 2), dtype=torch.int32)
        sharding.all_gather_records(dist, torch.from_numpy(rec.reshape(G * pps, 2, sharding.MAX_DET, sharding.REC_COLS)),
                                    torch.from_numpy(cnt.reshape(G * pps, 2)), g, gc)
        merged += sharding.merge_block(g.numpy(), gc.numpy(), blk * G, G, pps, world)
    dist.barrier()
    q.put((rank, [(p, f, r.copy()) for p, f, r in merged]))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_round_robin_assignment():
    assert sharding.pairs_for_rank(10, 0, 4) == [0, 4, 8]
    assert sharding.pairs_for_rank(10, 3, 4) == [3, 7]
    seen = sorted(p for r in range(8) for s in range(3) for p in sharding.step_pairs(s, 2, r, 8))
    assert seen == list(range(48))                 # every pair exactly once
    assert sharding.step_pairs(1, 2, 3, 8) == [19, 27]


def test_two_rank_all_gather_reassembles_sequence_order():
    world, pps, steps = 2, 2, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, pps, steps, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_pairs = world * pps * steps
    for rank, merged, tmax in results:
        assert tmax == float(world)                 # MAX over ranks reached every rank
        assert [(p, f) for p, f, _ in merged] == [(p, f) for p in range(n_pairs) for f in range(2)]
        for p, f, rec in merged:
            want_rec, want_cnt = _fake_records(p)
            assert rec.shape == (want_cnt[f], sharding.REC_COLS)
            assert np.array_equal(rec, want_rec[f, :want_cnt[f]])


@pytest.mark.parametrize('world,pps,G,blocks', [(2, 2, 2, 2), (3, 1, 4, 1), (2, 1, 8, 1)])
def test_block_of_steps_per_message_reassembles_sequence_order(world, pps, G, blocks):
    """merge_block at world > 1: the rank-major x step-major reshape puts every pair exactly once,
    in order, with its own records."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_block_worker, args=(r, world, port, pps, G, blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_pairs = world * pps * G * blocks
    for rank, merged in results:
        assert [(p, f) for p, f, _ in merged] == [(p, f) for p in range(n_pairs) for f in range(2)]
        for p, f, rec in merged:
            want_rec, want_cnt = _fake_records(p)
            assert np.array_equal(rec, want_rec[f, :want_cnt[f]])


def test_merge_block_equals_merge_step_per_step():
    """No process group needed: merge_block over a hand-built gathered buffer = merge_step of every step."""
    world, pps, G = 3, 2, 4
    g = np.zeros((world, G, pps, 2, sharding.MAX_DET, sharding.REC_COLS), np.float32)
    c = np.zeros((world, G, pps, 2), np.int32)
    for r in range(world):
        for i in range(G):
            for j, pid in enumerate(sharding.step_pairs(5 + i, pps, r, world)):
                g[r, i, j], c[r, i, j] = _fake_records(pid)
    got = sharding.merge_block(g.reshape(world * G * pps, 2, sharding.MAX_DET, sharding.REC_COLS),
                               c.reshape(-1, 2), 5, G, pps, world)
    want = []
    for i in range(G):
        want += sharding.merge_step(g[:, i], c[:, i], 5 + i, pps, world)
    assert [(p, f) for p, f, _ in got] == [(p, f) for p, f, _ in want]
    assert [(p, f) for p, f, _ in got] == [(p, f) for p in range(5 * world * pps, (5 + G) * world * pps)
                                           for f in range(2)]
    assert all(np.array_equal(a[2], b[2]) for a, b in zip(got, want))
