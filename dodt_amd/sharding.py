"""Multi-GPU sharding of the frame-pair stream.

Frame pairs are independent through NMS #2 (batch size 1, no cross-pair state:
avod/core/models/dt_rpn_model.py:733-735), so they shard round-robin over ranks with
no data-path collective; the only exchange is an all-gather of the fixed-size
detection records, which the sequential temporal module consumes
(avod/core/dt_evaluator_utils.py:212-367).  On the GPUs the exchange is `Communicator`:
RCCL through the C-ABI (dodt_comm_* / dodt_all_gather_records, include/dodt_hip.h), no PyTorch;
the assignment and merge logic below is backend-free and is also exercised over
torch.distributed's gloo backend in the CPU tests (`all_gather_records`).
"""
import ctypes as C
import os
import time

import numpy as np

MAX_DET = 100
REC_COLS = 17


def pairs_for_rank(n_pairs, rank, world):
    """Global pair ids of `rank`: pair i -> rank i mod world (static round-robin)."""
    return list(range(rank, n_pairs, world))


def step_pairs(step, pairs_per_step, rank, world):
    """Global ids of the pairs rank `rank` processes in step `step` (weak scaling:
    every step the ranks together consume world * pairs_per_step consecutive pairs)."""
    base = step * world * pairs_per_step
    return [base + j * world + rank for j in range(pairs_per_step)]


def all_gather_records(dist, rec, cnt, gathered, gathered_cnt):
    """One exchange per step: (pairs,2,MAX_DET,REC_COLS) float32 + (pairs,2) int32 per
    rank -> rank-major concatenation (world*pairs, ...) on every rank (the output form
    both the gloo and the nccl/RCCL backends accept)."""
    dist.all_gather_into_tensor(gathered, rec)
    dist.all_gather_into_tensor(gathered_cnt, cnt)


class Communicator(object):
    """One rank's RCCL communicator over the C-ABI.  `ctx`: the rank's device.Context.

    The ncclUniqueId travels through a file: rank 0 writes `<id_path>` atomically, the others
    poll for it (`rendezvous_path()` derives a per-job path from MASTER_PORT / the launcher's run
    id, so that concurrent jobs on one host do not meet)."""

    SLOTS = 4
    _generation = {}      # rendezvous base path -> communicators made so far by this process

    def __init__(self, ctx, rank, world, id_path=None, timeout_s=120.0):
        from dodt_amd import _lib
        self._lib, self.lib, self.ctx = _lib, ctx.lib, ctx
        self.rank, self.world = int(rank), int(world)
        uid = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        # one file per communicator of a job: the n-th Communicator of every rank meets at `<path>.<n>`
        base = id_path or rendezvous_path()
        gen = Communicator._generation.get(base, 0)
        Communicator._generation[base] = gen + 1
        path = '%s.%d' % (base, gen)
        if self.rank == 0:
            _lib.check(self.lib.dodt_comm_unique_id(uid), 'dodt_comm_unique_id')
            if self.world > 1:
                publish_id(path, bytes(uid))
        else:
            uid = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(
                fetch_id(path, _lib.COMM_ID_BYTES, timeout_s, 'rank %d' % self.rank))
        h = C.c_void_p()
        try:
            _lib.check(self.lib.dodt_comm_create(ctx.handle, self.rank, self.world, uid, C.byref(h)),
                       'dodt_comm_create')
        finally:
            # ncclCommInitRank returns once every rank has joined, i.e. has read the id: the file goes now
            if self.rank == 0 and self.world > 1:
                _remove(path)
        self.handle = h

    def all_gather_records(self, producer, slot, d_rec, d_cnt, d_all_rec, d_all_cnt):
        """Enqueue the step's exchange on the communicator's side stream, behind what
        `producer` has enqueued so far.  d_rec (pairs, frames, MAX_DET, REC_COLS) float32 and
        d_cnt (pairs, frames) int32 device arrays; d_all_* their (world * pairs, ...) twins."""
        pairs, frames, max_det, cols = d_rec.shape
        if tuple(d_all_rec.shape) != (self.world * pairs, frames, max_det, cols) or \
                tuple(d_cnt.shape) != (pairs, frames) or \
                tuple(d_all_cnt.shape) != (self.world * pairs, frames):
            raise ValueError('all_gather_records: buffer shapes do not match the world size')
        self._lib.check(self.lib.dodt_all_gather_records(
            self.handle, producer.handle, int(slot), d_rec.ptr, d_cnt.ptr, pairs, frames, max_det,
            cols, d_all_rec.ptr, d_all_cnt.ptr), 'dodt_all_gather_records')

    def attach(self, ctx):
        """Run the collectives on ctx's stream from now on (no stream of the communicator's own)."""
        self._lib.check(self.lib.dodt_comm_attach(self.handle, ctx.handle), 'dodt_comm_attach')

    def set_late_peer(self, microseconds):
        """Measurement aid: every gather is preceded on its stream by an idle kernel of this length (a peer that
        reaches the rendezvous late); 0 switches it off."""
        self._lib.check(self.lib.dodt_comm_set_late_peer(self.handle, float(microseconds)), 'dodt_comm_set_late_peer')

    def join(self, slot, consumer):
        """`consumer`'s later work waits for the gather last enqueued with `slot`."""
        self._lib.check(self.lib.dodt_comm_join(self.handle, int(slot), consumer.handle),
                        'dodt_comm_join')

    def sync(self):
        self._lib.check(self.lib.dodt_comm_sync(self.handle), 'dodt_comm_sync')

    def barrier(self):
        self._lib.check(self.lib.dodt_comm_barrier(self.handle), 'dodt_comm_barrier')

    def max_over_ranks(self, value):
        v = C.c_double(float(value))
        self._lib.check(self.lib.dodt_comm_max_f64(self.handle, C.byref(v)), 'dodt_comm_max_f64')
        return v.value

    def close(self):
        if self.handle:
            self.lib.dodt_comm_destroy(self.handle)
            self.handle = None


def _remove(path):
    try:
        os.remove(path)
    except OSError:
        pass


def publish_id(path, payload):
    """Rank 0's half of the id hand-over: the bytes appear at `path` atomically, in a file only this user can
    read, created exclusively (a file or symlink somebody else put at the temporary name is an error, not a
    target)."""
    tmp = '%s.%d.tmp' % (path, os.getpid())
    _remove(tmp)
    fd = os.open(tmp, os.O_CREAT | os.O_EXCL | os.O_WRONLY | getattr(os, 'O_NOFOLLOW', 0), 0o600)
    try:
        os.write(fd, payload)
    finally:
        os.close(fd)
    os.replace(tmp, path)


def fetch_id(path, nbytes, timeout_s=120.0, who='rank'):
    """The other ranks' half: poll for a regular file of exactly `nbytes` owned by this user."""
    from dodt_amd import _lib
    t0 = time.time()
    while True:
        try:
            st = os.lstat(path)
            import stat
            if stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid() and st.st_size == nbytes:
                with open(path, 'rb') as fh:
                    data = fh.read()
                if len(data) == nbytes:
                    return data
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise _lib.DodtError('%s: no RCCL id at %s after %.0f s' % (who, path, timeout_s))
        time.sleep(0.01)


class HostBarrier(object):
    """Degraded stand-in for Communicator's barrier / max_over_ranks through files next to the
    rendezvous path: what bench.py falls back to -- loudly, `config.exchange` says so -- when RCCL
    cannot make a communicator, so that the ranks' timing protocol (barrier, K steps, barrier,
    max over ranks) still holds.  It exchanges no detection records."""

    def __init__(self, rank, world, path=None, timeout_s=300.0):
        self.rank, self.world = int(rank), int(world)
        self.base = (path or rendezvous_path()) + '.host'
        self.timeout_s = timeout_s
        self.n = 0

    def _meet(self, payload):
        self.n += 1
        mine = '%s.%d.%d' % (self.base, self.n, self.rank)
        tmp = mine + '.tmp'
        with open(tmp, 'w') as fh:
            fh.write(repr(float(payload)))
        os.replace(tmp, mine)
        vals, t0 = [], time.time()
        for r in range(self.world):
            f = '%s.%d.%d' % (self.base, self.n, r)
            while not os.path.exists(f):
                if time.time() - t0 > self.timeout_s:
                    raise RuntimeError('rank %d: rank %d did not reach meeting %d' % (self.rank, r, self.n))
                time.sleep(0.002)
            vals.append(float(open(f).read()))
        if self.n > 2:       # everyone has read meeting n - 2 by now (they wrote n - 1 and n after it)
            try:
                os.remove('%s.%d.%d' % (self.base, self.n - 2, self.rank))
            except OSError:
                pass
        return vals

    def barrier(self):
        self._meet(0.0)

    def max_over_ranks(self, value):
        return max(self._meet(value))

    def close(self):
        """One last meeting, after which every rank has read everything up to the one before it: those
        files go; the last meeting's own (a few bytes per rank) stay, a peer may still be reading them."""
        self.barrier()
        for k in (self.n - 2, self.n - 1):
            try:
                os.remove('%s.%d.%d' % (self.base, k, self.rank))
            except OSError:
                pass


def rendezvous_path(env=None):
    """Where the ranks of one job meet: keyed by the launcher's rendezvous port and by the
    launcher's process id (the ranks of one node are children of one launcher -- torch.distributed.run's
    agent or bench.py's own spawner --, so a file left behind by a crashed job is not picked up
    by the next one; DODT_RUN_ID overrides the latter)."""
    env = os.environ if env is None else env
    key = '%s_%s' % (env.get('MASTER_PORT', '0'), env.get('DODT_RUN_ID') or os.getppid())
    key = ''.join(ch if ch.isalnum() or ch in '_-' else '_' for ch in key)
    return os.path.join(env.get('TMPDIR', '/tmp'), 'dodt_rccl_id_%s' % key)


def merge_block(gathered, gathered_cnt, first_step, n_steps, pairs_per_step, world):
    """Records of `n_steps` consecutive steps shipped as one message (each rank's block is
    (n_steps, pairs_per_step, 2, ...), step major) in global pair order."""
    g = np.asarray(gathered).reshape(world, n_steps, pairs_per_step, 2, MAX_DET, REC_COLS)
    c = np.asarray(gathered_cnt).reshape(world, n_steps, pairs_per_step, 2)
    out = []
    for i in range(n_steps):
        out += merge_step(g[:, i], c[:, i], first_step + i, pairs_per_step, world)
    return out


def merge_step(gathered, gathered_cnt, step, pairs_per_step, world):
    """Records of one step in global pair order: list of (pair_id, frame, (n,17) array)."""
    g = np.asarray(gathered).reshape(world, pairs_per_step, 2, MAX_DET, REC_COLS)
    c = np.asarray(gathered_cnt).reshape(world, pairs_per_step, 2)
    out = []
    for rank in range(world):
        for j, pid in enumerate(step_pairs(step, pairs_per_step, rank, world)):
            for f in range(2):
                out.append((pid, f, g[rank, j, f, :c[rank, j, f]]))
    out.sort(key=lambda t: (t[0], t[1]))
    return out
