"""The pipelined form of the streaming loop (three threads joined by bounded queues, the in-process counterpart of the
reference's source / Demodulator_process / decoder processes): same results in the same order as the sequential loop,
and a failure in any stage surfaces in the caller without hanging.  CPU only: the device stage is a stub."""
import threading
import time

import numpy as np
import pytest

from pycusdr_amd.demodulator_process import DemodulatorRunner


class _Runner(DemodulatorRunner):
    def __init__(self, fail_at=None):          # no device: only what run() touches
        self.samplesPerSlice, self.count, self.timeMA, self.radioName = 64, 0, 0.0, 'stub'
        self.fail_at = fail_at
        self.threads = set()

    def feed(self, new_samples):
        self.threads.add(threading.current_thread().name)
        if self.fail_at == self.count:
            raise RuntimeError('device stage failed')
        time.sleep(0.002)
        d = {'count': self.count, 'data': (np.abs(new_samples) > 0.5).astype(np.uint8), 'doppler': float(new_samples[0].real),
             'SNR': 0.0, 'spSymEst': 16.0, 'time_ms': 0.0, 'rate_ksps': 0.0, 'rate_ksps_avg': 0.0}
        self.count += 1
        return d


class _Decoder:
    def __init__(self, fail_at=None):
        self.seen, self.fail_at = 0, fail_at

    def findFrames(self, bits, frameStartIdx):
        if self.fail_at == self.seen:
            raise KeyError('decoder stage failed')
        self.seen += 1
        time.sleep(0.001)
        return [int(bits.sum()) + 1000 * self.seen], bits, 3


def _blocks(n, fail_at=None):
    rs = np.random.RandomState(0)
    buf = np.zeros(64, np.complex64)
    for i in range(n):
        if fail_at == i:
            raise OSError('source failed')
        buf[:] = rs.standard_normal(64) + 1j * rs.standard_normal(64)       # storage reused, as the ring buffer does
        yield buf


def test_pipelined_run_equals_sequential_run():
    seq, pk_seq = _Runner().run(_blocks(40), decoder=_Decoder())
    run = _Runner()
    pip, pk_pip = run.run(_blocks(40), decoder=_Decoder(), pipelined=True)
    assert len(seq) == len(pip) == 40 and pk_seq == pk_pip
    for a, b in zip(seq, pip):
        assert a['count'] == b['count'] and a['doppler'] == b['doppler'] and a['numSyncSig'] == b['numSyncSig'] == 3
        assert np.array_equal(a['data'], b['data'])
    assert run.threads == {'MainThread'}          # the device stage stays on the caller's thread (a handle is single-threaded)
    sunk = []
    res, _ = _Runner().run(_blocks(5), sink=sunk.append, pipelined=True)
    assert res == [] and [d['count'] for d in sunk] == [0, 1, 2, 3, 4]


@pytest.mark.timeout(60)
@pytest.mark.parametrize('where, exc', [('source', OSError), ('device', RuntimeError), ('decoder', KeyError)])
def test_pipelined_run_surfaces_failures_without_hanging(where, exc):
    run = _Runner(fail_at=7 if where == 'device' else None)
    with pytest.raises(exc):
        run.run(_blocks(30, fail_at=9 if where == 'source' else None), decoder=_Decoder(fail_at=5 if where == 'decoder' else None),
                pipelined=True)
    assert threading.active_count() <= 2          # the stage threads have ended


def test_marked_source_puts_a_marker_where_the_transport_would_block():
    """DemodulatorRunner.drain_marked(poll, wait): every chunk that is there, a ``None`` before each blocking wait, the end of the
    stream when ``wait`` returns None; the object is re-iterable state-free glue (no device)."""
    from pycusdr_amd.demodulator_process import MarkedSource
    ready = [[1, 2], [], [3], [], []]          # what poll() finds before each time it comes back empty
    waits = iter([10, 20, 30, None])           # what the blocking receive returns
    batches = iter(ready)
    cur = list(next(batches))
    calls = []

    def poll():
        calls.append('poll')
        return cur.pop(0) if cur else None

    def wait():
        calls.append('wait')
        nonlocal cur
        v = next(waits)
        cur = list(next(batches, []))
        return v
    src = DemodulatorRunner.drain_marked(poll, wait)
    assert isinstance(src, MarkedSource)
    assert list(src) == [1, 2, None, 10, None, 20, 3, None, 30, None]
    # every wait() is preceded by a poll() that came back empty: the loop has flushed before it blocks
    for i, c in enumerate(calls):
        assert c != 'wait' or calls[i - 1] == 'poll'
