"""A hang of the sharded loop becomes a diagnosis and a non-zero exit (pycusdr_amd.dist.StepWatchdog): world-2 gloo
processes on the CPU, one rank stops joining the collectives."""
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, 'tests', 'children', 'watchdog_child.py')


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(rank, port, withheld, stall_at, single):
    return subprocess.Popen([sys.executable, CHILD, str(rank), '2', str(port), str(withheld), str(stall_at), str(single)],
                            cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.timeout(120)
@pytest.mark.parametrize('single', [0, 1])
def test_all_ranks_present_finishes_cleanly(single):
    port = _port()
    procs = [_spawn(r, port, -1, -1, single) for r in range(2)]
    outs = [p.communicate(timeout=90) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert all('six steps done' in o[0] for o in outs)


@pytest.mark.timeout(120)
@pytest.mark.parametrize('withheld,single', [(1, 0), (0, 1)])
def test_withheld_rank_makes_the_other_exit_nonzero_naming_the_collective(withheld, single):
    port = _port()
    procs = [_spawn(r, port, withheld, 3, single) for r in range(2)]
    waiting = procs[1 - withheld]
    t0 = time.time()
    try:
        out, err = waiting.communicate(timeout=60)
    finally:
        procs[withheld].kill()
        procs[withheld].communicate()
    assert waiting.returncode == 3, (out, err)
    assert time.time() - t0 < 45
    assert '[mfb watchdog] rank %d: no progress' % (1 - withheld) in err and 'after step 2' in err
    # the rank without the stream waits in the block broadcast, the stream's owner in the exchange of the scores
    assert ('broadcast of a block' in err) if withheld == 0 else ('all-gather' in err or 'all-reduce' in err or 'broadcast' in err), err
    assert 'six steps done' not in out
