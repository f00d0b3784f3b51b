"""The host copy worker of the library (mfb_hostcopy_*, pycusdr_amd.mfbank.HostCopy): chunk -> sample-window copies on a thread
of their own.  Plain memory, no GPU: runs everywhere the library loads."""
import threading
import time

import numpy as np
import pytest

from pycusdr_amd import _lib
from pycusdr_amd.mfbank import HostCopy


def test_copies_arrive_in_order_and_complete_at_drain():
    rs = np.random.RandomState(0)
    hc = HostCopy()
    try:
        dst = np.zeros(1 << 20, np.complex64)
        want = np.zeros_like(dst)
        for rep in range(20):
            pos = 0
            while pos < len(dst):
                n = int(min(len(dst) - pos, rs.randint(1, 40000)))
                src = (rs.randn(n) + 1j * rs.randn(n)).astype(np.complex64)
                hc.submit(dst, pos, src)
                want[pos:pos + n] = src
                pos += n
            # overlapping destinations: the later copy wins (submission order)
            late = np.full(1000, rep + 2j, np.complex64)
            hc.submit(dst, 500, late)
            want[500:1500] = late
            hc.drain()
            assert np.array_equal(dst, want), rep
        hc.drain()                       # nothing pending: returns at once
    finally:
        hc.close()
    hc.close()                           # idempotent


def test_other_dtypes_slices_and_errors():
    hc = HostCopy()
    try:
        dst = np.zeros(1000, np.complex64)
        hc.submit(dst, 10, np.arange(20, dtype=np.float64))            # converted to the destination's type, as slice assignment does
        hc.submit(dst, 100, np.arange(50, dtype=np.complex64)[::2])    # not contiguous: copied first
        hc.submit(dst, 0, np.zeros(0, np.complex64))
        hc.drain()
        assert np.array_equal(dst[10:30], np.arange(20).astype(np.complex64))
        assert np.array_equal(dst[100:125], np.arange(50, dtype=np.complex64)[::2])
        with pytest.raises(IndexError):
            hc.submit(dst, 990, np.zeros(20, np.complex64))
        with pytest.raises(ValueError):
            hc.submit(dst[::2], 0, np.zeros(2, np.complex64))
    finally:
        hc.close()


def test_the_submitting_thread_is_free_while_the_worker_copies():
    """256 MiB of copies are queued in well under the time they take, and the caller's own work goes on beside them."""
    hc = HostCopy()
    try:
        src = np.ones(1 << 22, np.complex64)                             # 32 MiB
        dsts = [np.empty_like(src) for _ in range(8)]
        t0 = time.perf_counter()
        for d in dsts:
            hc.submit(d, 0, src)
        t_submit = time.perf_counter() - t0
        hc.drain()
        t_all = time.perf_counter() - t0
        assert all(np.array_equal(d[::4097], src[::4097]) for d in dsts)
        assert t_submit < 0.25 * t_all or t_all < 2e-3, (t_submit, t_all)
    finally:
        hc.close()


def test_destroy_finishes_what_was_submitted():
    lib = _lib.load()
    import ctypes as C
    h = C.c_void_p()
    assert lib.mfb_hostcopy_create(C.byref(h)) == 0
    src = np.arange(1 << 20, dtype=np.float32)
    dst = np.zeros_like(src)
    for i in range(0, len(src), 4096):
        assert lib.mfb_hostcopy_submit(h, dst.ctypes.data + 4 * i, src.ctypes.data + 4 * i, 4096 * 4) == 0
    lib.mfb_hostcopy_destroy(h)
    assert np.array_equal(dst, src)
    assert lib.mfb_hostcopy_submit(None, None, None, 0) != 0 and lib.mfb_hostcopy_drain(None) != 0


def test_two_workers_side_by_side():
    out = []

    def body(seed):
        rs = np.random.RandomState(seed)
        hc = HostCopy()
        try:
            dst = np.zeros(200000, np.float32)
            want = np.zeros_like(dst)
            for _ in range(200):
                n = int(rs.randint(1, 5000))
                at = int(rs.randint(0, len(dst) - n))
                src = rs.randn(n).astype(np.float32)
                hc.submit(dst, at, src)
                want[at:at + n] = src
            hc.drain()
            out.append(np.array_equal(dst, want))
        finally:
            hc.close()
    ts = [threading.Thread(target=body, args=(s,)) for s in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out == [True] * 4


@pytest.mark.parametrize('with_copier', [False, True, 'read-only'])
def test_window_assembler_pull_form_builds_the_same_windows_as_the_push_form(with_copier):
    """sigFIFO.WindowAssembler.take / full / retarget (the batched receive loop's pull form, copies queued for the copy thread)
    against its push generator and against slicing the stream directly: every window = B blocks of stride samples behind the
    carried overlap."""
    from pycusdr_amd.sigFIFO import WindowAssembler
    rs = np.random.RandomState(3)
    ov, stride, B = 64, 1000, 4
    size = B * stride + ov
    stream = (rs.randn(31 * stride + 17) + 1j * rs.randn(31 * stride + 17)).astype(np.complex64)
    head = (rs.randn(ov) + 1j * rs.randn(ov)).astype(np.complex64)
    full_stream = np.concatenate((head, stream))
    want = [full_stream[w * B * stride:w * B * stride + size].copy() for w in range((len(stream)) // (B * stride))]

    # push form
    a, b = np.zeros(size, np.complex64), np.zeros(size, np.complex64)
    a[:ov] = head
    asm = WindowAssembler(a, ov, stride, B)
    got, other = [], b
    for i in range(0, len(stream), 777):
        for nb in asm.push(stream[i:i + 777]):
            assert nb == B and len(asm.stamps) == B
            got.append(asm.buf.copy())
            cur = asm.buf
            asm.retarget(other, nb)
            other = cur
    assert len(got) == len(want) and all(np.array_equal(x, y) for x, y in zip(got, want))

    # pull form
    hc = HostCopy() if with_copier else None
    try:
        a[:], b[:] = 0, 0
        a[:ov] = head
        # a read-only stream is queued for the copy thread as it is; a writable one only when told so (copy_all_async)
        if with_copier == 'read-only':
            stream.flags.writeable = False
        asm = WindowAssembler(a, ov, stride, B, copier=hc, copy_all_async=with_copier is True)
        queued = []
        if hc is not None:
            inner = hc.submit
            hc.submit = lambda *args: (queued.append(1), inner(*args))[1]
        got, other, rest = [], b, None
        chunks = iter(stream[i:i + 1234] for i in range(0, len(stream), 1234))
        while True:
            if rest is None:
                rest = next(chunks, None)
                if rest is None:
                    break
            n = asm.take(rest)
            rest = rest[n:] if n < len(rest) else None
            if asm.full():
                if hc is not None:
                    hc.drain()
                assert asm.complete_blocks() == B and len(asm.stamps) == B
                got.append(asm.buf.copy())
                cur = asm.buf
                asm.retarget(other, B)
                other = cur
        assert len(got) == len(want) and all(np.array_equal(x, y) for x, y in zip(got, want))
        # the partly filled last window: its complete blocks, in place
        if hc is not None:
            hc.drain()
        assert hc is None or len(queued) >= len(stream) // 1234
        nb = asm.complete_blocks()
        tail = full_stream[len(want) * B * stride:]
        assert nb == (len(tail) - ov) // stride and np.array_equal(asm.buf[:asm.fill], tail)
        if hc is not None:
            # a writable chunk without copy_all_async is copied on the spot
            plain = WindowAssembler(np.zeros(size, np.complex64), ov, stride, B, copier=hc)
            del queued[:]
            plain.take(np.ones(10, np.complex64))
            assert not queued and np.array_equal(plain.buf[ov:ov + 10], np.ones(10, np.complex64))
    finally:
        if hc is not None:
            hc.close()
