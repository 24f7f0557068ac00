"""Host-side logic of bench.py that needs no GPU: the supervision of the ranks it spawns (a rank that
leaves with an error must end its siblings, which would otherwise wait in an RCCL collective for ever) and the
temporal-module worker of alt.temporal_host."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _child(code):
    return subprocess.Popen([sys.executable, '-c', code])


def test_first_failing_rank_ends_the_others():
    t0 = time.time()
    procs = [_child('import time; time.sleep(60)'), _child('import time, sys; time.sleep(0.3); sys.exit(3)'),
             _child('import time; time.sleep(60)')]
    rc = bench._supervise(procs, poll_s=0.05, grace_s=5.0)
    assert rc == 3
    assert time.time() - t0 < 20
    assert all(p.poll() is not None for p in procs)          # nobody left behind
    assert procs[0].returncode != 0 and procs[2].returncode != 0


def test_all_ranks_fine_and_killed_rank():
    assert bench._supervise([_child('pass'), _child('import time; time.sleep(0.2)')], poll_s=0.05) == 0
    # a rank killed by a signal counts as a failure (128 + signal number), and a rank that ignores SIGTERM is killed
    stubborn = _child('import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(60)')
    dying = _child('import os, signal, time; time.sleep(0.3); os.kill(os.getpid(), signal.SIGKILL)')
    rc = bench._supervise([stubborn, dying], poll_s=0.05, grace_s=1.0)
    assert rc == 128 + 9
    assert stubborn.poll() is not None


def test_temporal_worker_on_synthetic_records():
    """One small sequence through the M stage's worker: it runs, reports a time and finds the cars' tracks."""
    recs = []
    for k in range(5):
        rec = np.zeros((6, 17), np.float32)
        for f in range(2):
            for c in range(3):
                z = 12.0 + 8 * c + 0.8 * (2 * k + 2 * f)
                rec[3 * f + c] = [-4.0 + 4 * c, 1.65, z, 3.9, 1.6, 1.5, 0.1, 0.9 - 0.1 * c, 0,
                                  *((-4.0 + 4 * c, 1.65, z + 1.6, 3.9, 1.6, 1.5, 0.1) if f == 0 else (0,) * 7), f]
        recs.append(rec)
    t, n_tracks = bench._temporal_worker((recs, 2, 1))
    assert t > 0 and n_tracks >= 3
    assert bench._temporal_worker((None, 0, 0)) == (0.0, 0)
    out = bench.temporal_host(None, 1, recs, 2, 100.0)
    assert out['one_thread_pairs_per_s'] > 0 and 'pool_pairs_per_s' not in out
