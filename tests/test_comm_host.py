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


def _fetch_rank(path, q):
    q.put(sharding.fetch_id(path, 128, timeout_s=30.0, who='rank 1'))


def test_id_file_rendezvous_two_processes(tmp_path):
    """The file logic of Communicator.__init__ alone (no RCCL): a second process polls for the id, rank 0
    publishes it atomically, private to the user; wrong-sized files, symlinks and stale temporaries are not
    accepted as / do not divert the id."""
    import multiprocessing as mp
    import stat
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    path = str(tmp_path / 'id.0')
    p = ctx.Process(target=_fetch_rank, args=(path, q))
    p.start()
    # a short file at the path must not be taken for the id; nor must a symlink to a 128-byte file
    with open(path, 'wb') as fh:
        fh.write(b'x' * 5)
    import time
    time.sleep(0.2)
    os.remove(path)
    victim = tmp_path / 'victim'
    victim.write_bytes(b'v' * 128)
    os.symlink(str(victim), path)
    time.sleep(0.2)
    os.remove(path)
    # a symlink planted at the temporary name does not divert the write
    decoy = tmp_path / 'decoy'
    decoy.write_bytes(b'')
    os.symlink(str(decoy), '%s.%d.tmp' % (path, os.getpid()))
    payload = bytes(range(128))
    sharding.publish_id(path, payload)
    assert decoy.read_bytes() == b''
    assert stat.S_IMODE(os.stat(path).st_mode) == 0o600
    assert q.get(timeout=30) == payload
    p.join(timeout=30)
    assert p.exitcode == 0
    with pytest.raises(_lib.DodtError):
        sharding.fetch_id(str(tmp_path / 'nobody'), 128, timeout_s=0.05)


def test_every_communicator_of_a_process_gets_its_own_id_file():
    """Generation counter: the n-th Communicator of each rank meets at <base>.<n>, so a second communicator of
    the same job never reads the first one's id."""
    sharding.Communicator._generation.pop('/tmp/x_base', None)
    seen = []
    for _ in range(3):
        gen = sharding.Communicator._generation.get('/tmp/x_base', 0)
        sharding.Communicator._generation['/tmp/x_base'] = gen + 1
        seen.append(gen)
    assert seen == [0, 1, 2]
