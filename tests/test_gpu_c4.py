"""BASELINE config C4 at its real geometry on ONE device: 2048 Doppler bins sharded 256 per rank over EIGHT ranks,
N = 2^20, GMSK bank, through DopplerShard -- as eight threads of one process (a GPU box admits at most six processes on
its card), each with its own library handle, stream and exchange buffers, joined by tests/thread_comm.py.  The exchanged
table and the pick must equal, bit for bit, those of ONE unsharded 2048-bin handle: a score depends on the block, the shift
and the filter only, never on how many bins a handle holds (seg_kernels.hpp: one partial sum per slot, fixed order)."""
import numpy as np
import pytest

from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table
from pycusdr_amd.dist import DopplerShard
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol
from thread_comm import run_ranks

pytestmark = pytest.mark.gpu


def _c4(log2N, D):
    N = 1 << log2N
    conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=D, rangeRateMax=60000)
    proto = loadProtocol('bench_GMSK')(conf=conf)
    M, masks = proto.get_filter(N, 16, 3)
    _, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], 60000, N)
    x = sg.s1_stream(2, N, 1 << 10, 'GMSK', snr_db=10.0, seed=1)
    return N, M, masks, shifts, [x[:N], x[N - 1024:2 * N - 1024]]


@pytest.mark.parametrize('exchange', ['allgather', 'allreduce'])
def test_c4_eight_ranks_equal_the_unsharded_handle(exchange):
    import torch
    log2N, D, world = 20, 2048, 8
    N, M, masks, shifts, blocks = _c4(log2N, D)
    assert len(np.unique(shifts)) == D
    dev = torch.device('cuda', 0)
    one = MFBank(log2N, D, M, sum_all_masks=True)
    one.set_filters(masks)
    one.set_shifts(shifts)
    want = []
    for x in blocks:
        one.upload(x)
        idx, metric = one.find_carrier()
        want.append((idx, metric, one.get_scores().copy()))
    one.close()
    assert abs(np.interp(float(want[0][0]), np.arange(D), shifts) - N / 4) < 2 * np.median(np.diff(shifts))
    dev_blocks = [torch.from_numpy(np.ascontiguousarray(x).view(np.float32)).to(dev) for x in blocks]
    torch.cuda.synchronize()

    def body(comm):
        rank = comm.get_rank()
        shard = DopplerShard(device=dev, comm=comm)
        lo, hi = shard.bin_range(D)
        assert hi - lo == 256
        bank = MFBank(log2N, hi - lo, M, sum_all_masks=True)
        try:
            bank.set_filters(masks)
            bank.set_shifts(shifts[lo:hi])
            shard.attach(bank, D, M, sum_all=True, exchange=exchange)
            assert shard.even == (exchange == 'allgather')
            got = []
            for i, blk in enumerate(dev_blocks):
                nxt = dev_blocks[i + 1] if i + 1 < len(dev_blocks) else None
                idx, metric = shard.step(bank, lo, blk if rank == 0 else None, next_block=nxt if rank == 0 else None,
                                         prefetch_next=nxt is not None)
                got.append((idx, metric, shard.full_scores()))
            return got
        finally:
            bank.close()
    for got in run_ranks(world, body):
        for (i0, m0, s0), (i1, m1, s1) in zip(want, got):
            assert i0 == i1 and m0 == m1
            assert np.array_equal(s0, s1)


def test_scores_do_not_depend_on_the_number_of_bins_or_the_grid():
    """The property the sharded path rests on: the same (block, shift, filter) gives the same bits from a 3-bin handle,
    a 300-bin handle and any grid tuning."""
    log2N = 18
    N, M, masks, shifts, blocks = _c4(log2N, 300)
    tables = []
    for sel, tune in ((slice(0, 300), ()), (slice(100, 103), ()), (slice(0, 300), (0, 8, 4)), (slice(97, 200), (0, 3, 0))):
        b = MFBank(log2N, len(shifts[sel]), M, sum_all_masks=True)
        b.set_filters(masks)
        b.set_shifts(shifts[sel])
        if tune:
            b.set_search_path('segment', *tune)
        b.upload(blocks[0])
        b.find_carrier()
        full = np.full(300, np.nan, np.float32)
        full[sel] = b.get_scores()[:, 0]
        tables.append(full)
        b.close()
    for t in tables[1:]:
        keep = ~np.isnan(t)
        assert np.array_equal(t[keep], tables[0][keep])
