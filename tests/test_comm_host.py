"""CPU-side checks of the exchange step's host logic (no GPU): rendezvous path, argument errors."""
import os

import pytest

from dodt_amd import _lib, sharding


def test_rendezvous_path_is_per_job():
    a = sharding.rendezvous_path({'MASTER_PORT': '29500', 'DODT_RUN_ID': 'run/1'})
    b = sharding.rendezvous_path({'MASTER_PORT': '29501', 'DODT_RUN_ID': 'run/1'})
    c = sharding.rendezvous_path({'MASTER_PORT': '29500'})
    assert a != b and a != c and '/' not in os.path.basename(a)
    assert str(os.getppid()) in c            # ranks of one launcher meet, jobs of another do not
    assert sharding.rendezvous_path({'MASTER_PORT': '1', 'TMPDIR': '/dev/shm'}).startswith('/dev/shm/')


def test_comm_entry_points_reject_null_arguments():
    lib = _lib.load()
    assert lib.dodt_comm_create(None, 0, 1, None, None) == _lib.ERR_INVALID
    assert lib.dodt_comm_sync(None) == _lib.ERR_INVALID
    assert lib.dodt_comm_barrier(None) == _lib.ERR_INVALID
    assert lib.dodt_comm_destroy(None) == _lib.OK
    with pytest.raises(ValueError):
        _lib.check(lib.dodt_comm_join(None, 0, None), 'join')


def _host_rank(rank, world, path, q):
    hb = sharding.HostBarrier(rank, world, path)
    hb.barrier()
    m = [hb.max_over_ranks(10.0 * k + rank) for k in range(5)]
    hb.close()
    q.put((rank, m))


def test_host_barrier_fallback_two_processes(tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    path = str(tmp_path / 'rdzv')
    procs = [ctx.Process(target=_host_rank, args=(r, 2, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == got[1] == [10.0 * k + 1 for k in range(5)]
    assert len([f for f in os.listdir(str(tmp_path)) if f.startswith('rdzv.host')]) <= 2   # the last meeting's
