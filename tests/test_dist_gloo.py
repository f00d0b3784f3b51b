"""Multi-rank path on CPU: world_size-2 gloo processes shard the Doppler bins, each computes its
slice of the score matrix (oracle), all-reduce, and every rank must pick exactly what a single
process picks on the full table."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from pycusdr_amd.dist import bin_slice


def test_bin_slice_partitions():
    for D in (1, 7, 256, 2048, 1001):
        for G in (1, 2, 3, 8):
            parts = [bin_slice(D, r, G) for r in range(G)]
            assert parts[0][0] == 0 and parts[-1][1] == D
            assert all(parts[i][1] == parts[i + 1][0] for i in range(G - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, concurrent=True):
    """One rank: DopplerShard itself (CPU tensors over gloo) around the OracleBank stand-in: rank 0 broadcasts the
    block, every rank searches its bin slice, one all-reduce, the pick on every rank."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import torch
    import torch.distributed as dist
    from oracle import mfbank_oracle as orc
    from oracle_bank import OracleBank
    from pycusdr_amd.dist import DopplerShard, allreduce_scores_host, bin_owner
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rs = np.random.RandomState(5)              # same filters on every rank; the block lives on rank 0 only
    log2N, M, D = 12, 4, 10
    N = 1 << log2N
    x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
    shifts = rs.randint(0, N, D)
    X = orc.forward_fft(x)
    ok = True
    for sum_all, D in ((True, 10), (False, 10), (True, 7), (False, 7)):    # even slices: all-gather; uneven: all-reduce
        shifts = rs.randint(0, N, D)
        shard = DopplerShard(device=torch.device('cpu'), concurrent_broadcast=concurrent)
        ok &= (shard.bcast_group is shard.group) == (not concurrent)        # single-communicator mode: ONE communicator
        lo, hi = shard.bin_range(D)
        bank = OracleBank(log2N, hi - lo, M, sum_all_masks=sum_all)
        bank.set_filters(masks)
        bank.set_shifts(shifts[lo:hi])
        shard.attach(bank, D, M, sum_all=sum_all)
        block = torch.from_numpy(x.view(np.float32).copy()) if rank == 0 else None
        idx, metric = shard.step(bank, lo, block)
        ok &= shard.phase == 'between steps' and 'phase:' in shard.describe()
        single = orc.doppler_scores(X, masks, shifts, sum_all).astype(np.float32)
        sidx, smetric = orc.find_doppler_est(single, D, 0, sum_all)
        ok &= bool(np.array_equal(shard.full_scores(), single))       # adding exact zeros
        ok &= bool(idx == sidx and metric == smetric)
        ok &= shard.owner(int(float(idx))) == bin_owner(D, world, int(float(idx)))
        if sum_all:
            ok &= shard.scores.numel() == D                            # only the populated column travels
        # a stream of blocks: the next block is distributed while the current one is searched
        b1 = torch.from_numpy(x.view(np.float32).copy()) if rank == 0 else None
        b2 = torch.from_numpy((x * np.complex64(2)).view(np.float32).copy()) if rank == 0 else None
        i1, _ = shard.step(bank, lo, b1, next_block=b2, prefetch_next=True)
        ok &= bool(np.array_equal(shard.full_scores(), single))
        i2, _ = shard.step(bank, lo, None, next_block=b1, prefetch_next=True)      # block 2 is already there
        ok &= bool(np.array_equal(shard.full_scores(), single * np.float32(4))) and i1 == sidx and i2 == sidx
        i3, _ = shard.step(bank, lo, None)                                         # and block 1 again
        ok &= bool(np.array_equal(shard.full_scores(), single)) and i3 == sidx
    # the documented numpy statement of the same exchange
    local = orc.doppler_scores(X, masks, shifts[lo:hi], True).astype(np.float32)
    ok &= bool(np.array_equal(allreduce_scores_host(local, lo, D), orc.doppler_scores(X, masks, shifts, True).astype(np.float32)))
    q.put((rank, bool(ok), float(idx)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize('concurrent', [True, False])
def test_sharded_scores_allreduce_world2(concurrent):
    """Both broadcast modes: a communicator of its own for the block broadcast (default), or the exchange's (single-communicator
    mode: one program order of collectives per rank)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, concurrent)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]


def _worker_noise(rank, world, port, q):
    """Noise-reference bin under sharding (reference DB:148-159, CU:550-554): every rank searches the noise bin beside its
    slice; the exchanged table [noise | D bins] and the pick equal the single-process ones, even and uneven slices."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import torch
    import torch.distributed as dist
    from oracle import mfbank_oracle as orc
    from oracle_bank import OracleBank
    from pycusdr_amd.dist import DopplerShard
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rs = np.random.RandomState(11)
    log2N, M = 12, 4
    N = 1 << log2N
    x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
    X = orc.forward_fft(x)
    ok = True
    for sum_all, D in ((True, 10), (False, 10), (True, 7), (False, 7)):
        shifts = rs.randint(0, N, D + 1)                 # row 0 = the noise bin
        shard = DopplerShard(device=torch.device('cpu'))
        lo, hi = shard.bin_range(D)
        bank = OracleBank(log2N, hi - lo, M, sum_all_masks=sum_all, doppler_offset=1)
        bank.set_filters(masks)
        bank.set_shifts(np.concatenate((shifts[:1], shifts[1 + lo:1 + hi])))
        shard.attach(bank, D, M, sum_all=sum_all, noise_rows=1)
        block = torch.from_numpy(x.view(np.float32).copy()) if rank == 0 else None
        idx, metric = shard.step(bank, lo, block)
        single = orc.doppler_scores(X, masks, shifts, sum_all).astype(np.float32)
        sidx, smetric = orc.find_doppler_est(single, D, 1, sum_all)
        ok &= bool(np.array_equal(shard.full_scores(), single))
        ok &= bool(idx == sidx and metric == smetric)
        ok &= shard.owner(0) == 0 and shard.owner(1) == 0 and shard.owner(D) == world - 1
    q.put((rank, bool(ok), float(idx)))
    dist.destroy_process_group()


def _spawn(worker, world=2, timeout=100, extra=()):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=1.0))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:      # a rank died (its peers would wait for it for ever) or time is up
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f'ranks exited with {dead}' if dead else f'no result after {timeout} s')
    for p in procs:
        p.join(timeout=30)
    return sorted(res)


@pytest.mark.timeout(120)
def test_noise_reference_bin_under_sharding_world2():
    res = _spawn(_worker_noise)
    assert all(ok for _, ok, _ in res) and res[0][2] == res[1][2]


def _hopping_stream(N, ov, nblocks, offsets_hz, fs=153600):
    """The GMSK bench packet, tiled; the carrier sits at fs/4 + offsets_hz[b] during block b's new samples: the picked
    Doppler bin moves through the bin table from block to block."""
    from pycusdr_amd import signals as sg
    sig = sg.s1_stream(nblocks, N, ov, 'GMSK', snr_db=20.0, seed=3).astype(np.complex128)
    n = np.arange(len(sig))
    f = np.zeros(len(sig))
    for b, hz in enumerate(offsets_hz):
        f[ov + b * (N - ov): ov + (b + 1) * (N - ov)] = hz
    return (sig * np.exp(2j * np.pi * np.cumsum(f) / fs)).astype(np.complex64), n


def _worker_stream(rank, world, port, q):
    """The sharded streaming loop against the single-process loop on a carrier that drifts from rank 0's bins into rank
    1's while a packet is on the air: every rank must hand the decoder the same contiguous bit stream, with the same
    block-to-block alignment state, as the unsharded Demodulator -- and the same packets come out."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import torch
    import torch.distributed as dist
    from oracle import mfbank_oracle as orc
    from oracle_bank import OracleBank
    import pycusdr_amd.demodulator.demodulator_base as dbm
    from pycusdr_amd import config as cfg
    from pycusdr_amd.decoder import Decoder
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from pycusdr_amd.dist import DopplerShard
    from pycusdr_amd.protocol import loadProtocol
    dbm.MFBank = OracleBank
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    bs, ov, D = 13, 1 << 10, 9                     # uneven slices: 5 + 4 bins, and a noise-reference bin in front
    N = 1 << bs
    nblocks = 27                                   # the whole bench packet: 10 000 symbols in blocks of 448
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D)
    conf['Radios']['Rx']['UHF-H']['noise_measure_offset_Hz'] = 60000
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig, _ = _hopping_stream(N, ov, nblocks, np.linspace(-7000, 7000, nblocks))
    step = N - ov
    chunks = [sig[ov + i * step: ov + (i + 1) * step] for i in range(nblocks)]
    shard = DopplerShard(device=torch.device('cpu'))
    run = DemodulatorRunner(conf, p, 'UHF-H', shard=shard)
    dec = Decoder({}, p, correlator=orc.sync_correlate)
    res, packets = run.run([c if rank == 0 else np.zeros(step, np.complex64) for c in chunks], decoder=dec)
    owners = [shard.owner(run.demod.doppIdxArrayOffset) for _ in range(1)]
    picks = [float(d['doppler']) for d in res]
    ok = True
    if rank == 0:
        plain = DemodulatorRunner(conf, p, 'UHF-H')
        ref, ref_packets = plain.run(chunks, decoder=Decoder({}, p, correlator=orc.sync_correlate))
        for a, b in zip(res, ref):
            ok &= bool(np.array_equal([a['doppler'], a['doppler_std'], a['SNR']], [b['doppler'], b['doppler_std'], b['SNR']], equal_nan=True))
            ok &= bool(np.array_equal(a['data'], b['data']) and np.array_equal(a['trust'], b['trust'])) and a['spSymEst'] == b['spSymEst']
        ok &= bool(np.array_equal(run.demod.poswinP, plain.demod.poswinP) and np.array_equal(run.demod.posSymEnd, plain.demod.posSymEnd))
        ok &= len(packets) == len(ref_packets) and all(np.array_equal(u.bits if hasattr(u, 'bits') else u.rawData, v.bits if hasattr(v, 'bits') else v.rawData)
                                                        for u, v in zip(packets, ref_packets))
    # every rank: the same stream (hash of all bits), and the pick really moved across the slice boundary
    bits = np.concatenate([d['data'] for d in res])
    t = torch.tensor([float(bits.sum()), float(len(bits)), float(len(packets))] + picks, dtype=torch.float64)
    g = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(g, t)
    ok &= all(bool(torch.equal(g[0], u)) for u in g)
    hz = np.array(picks)
    q.put((rank, bool(ok), float(hz.min()), float(hz.max()), len(bits), len(packets)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_stream_is_contiguous_when_the_owner_changes():
    res = _spawn(_worker_stream, timeout=280)
    assert all(r[1] for r in res), res
    assert res[0][2] < -3000 and res[0][3] > 3000          # the carrier crossed from rank 0's bins into rank 1's
    assert res[0][4] == res[1][4] > 10000 and res[0][5] == res[1][5] == 1      # one packet, on every rank


def _plant_slips(run, slips):
    """Make the symbol grid of chosen blocks start one symbol early / late (centres moved by -+ one symbol): what a timing
    estimate that lands in the neighbouring symbol does.  The alignment against the previous block (DB:938-957) must repair
    it -- drop or re-insert one bit at the window's start -- or the bit stream gains / loses a symbol there."""
    orig = run.demod.demodulateDevice

    def slipped():
        rec = orig()
        k = slips.get(run.count, 0)                # demodulateDevice runs before the runner counts the block
        if k:
            rec['centres'] = rec['centres'] + np.int32(k * 16)
        return rec
    run.demod.demodulateDevice = slipped


def _worker_blockshard(rank, world, port, q, slips=None, root_rank=0, cut=None):
    """Time-chunk sharding: rank r runs the device AND host stages of blocks r, r + G, ... (the previous block's tail comes
    from its owner); the root runs the decoder in block order.  The bit stream, the alignment state and the packets must
    equal those of one process running the whole stream."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import torch.distributed as dist
    from oracle import mfbank_oracle as orc
    from oracle_bank import OracleBank
    import pycusdr_amd.demodulator.demodulator_base as dbm
    from pycusdr_amd import config as cfg
    from pycusdr_amd.decoder import Decoder
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from pycusdr_amd.dist import BlockShard, StepWatchdog
    from pycusdr_amd.protocol import loadProtocol
    dbm.MFBank = OracleBank
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    bs, ov, D = 13, 1 << 10, 8
    N = 1 << bs
    nblocks = 27
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig, _ = _hopping_stream(N, ov, nblocks, np.linspace(-3000, 3000, nblocks))
    step = N - ov
    chunks = [sig[ov + i * step: ov + (i + 1) * step] for i in range(nblocks)]
    shard = BlockShard(root=root_rank)
    run = DemodulatorRunner(conf, p, 'UHF-H')
    run.raw[:ov] = sig[:ov]
    if slips:
        _plant_slips(run, slips)
    dog = StepWatchdog(120.0, rank=rank, describe=shard.describe)
    dec = Decoder({}, p, correlator=orc.sync_correlate)
    if cut is None:
        res, packets = shard.run(run, chunks, decoder=dec, watchdog=dog)
    else:           # the stream in two calls: the second call's first block continues the first call's last
        res, packets = shard.run(run, chunks[:cut], decoder=dec, watchdog=dog)
        more, pk = shard.run(run, chunks[cut:], decoder=dec, watchdog=dog)
        res, packets = res + more, packets + pk
    dog.stop()
    ok = True
    repaired = 0
    if rank == root_rank:
        plain = DemodulatorRunner(conf, p, 'UHF-H')
        plain.raw[:ov] = sig[:ov]
        if slips:
            _plant_slips(plain, slips)
        ref, ref_packets = plain.run(chunks, decoder=Decoder({}, p, correlator=orc.sync_correlate))
        ok &= len(res) == len(ref) == nblocks
        for a, b in zip(res, ref):
            ok &= a['count'] == b['count'] and a['numSyncSig'] == b['numSyncSig']
            ok &= bool(np.array_equal([a['doppler'], a['doppler_std'], a['SNR'], a['spSymEst']],
                                      [b['doppler'], b['doppler_std'], b['SNR'], b['spSymEst']], equal_nan=True))
            ok &= bool(np.array_equal(a['data'], b['data']) and np.array_equal(a['trust'], b['trust']))
        ok &= bool(np.array_equal(run.demod.poswinP, plain.demod.poswinP) and np.array_equal(run.demod.posSymEnd, plain.demod.posSymEnd))
        ok &= len(packets) == len(ref_packets) == 1 and bool(np.array_equal(packets[0].bits, ref_packets[0].bits))
        ok &= packets[0].checkPacketData() == ref_packets[0].checkPacketData()
        if slips:
            # the planted slips were really there and really repaired: blocks hand over a bit more or less than they do on
            # the undisturbed grid, and yet the concatenated stream is the undisturbed stream, bit for bit
            clean = DemodulatorRunner(conf, p, 'UHF-H')
            clean.raw[:ov] = sig[:ov]
            base, _ = clean.run(chunks)
            moved = sum(1 for a, b in zip(res, base) if len(a['data']) != len(b['data']))
            same = np.array_equal(np.concatenate([d['data'] for d in res]), np.concatenate([d['data'] for d in base]))
            repaired = moved if same else -1
    else:
        ok &= res == [] and packets == [] and run.count == nblocks
    q.put((rank, bool(ok), len(res), repaired))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 3])
def test_block_round_robin_equals_single_process_stream(world):
    res = _spawn(_worker_blockshard, world=world, timeout=280)
    assert all(r[1] for r in res), res
    assert res[0][2] == 27


@pytest.mark.timeout(300)
@pytest.mark.parametrize('root,cut', [(0, 13), (1, 14), (2, 10)])
def test_block_round_robin_stream_in_two_calls_any_root(root, cut):
    """The stream handed over in two ``run`` calls (the second call's block 0 continues the first call's last block, whoever owned
    it), with the root on any rank: the same blocks, alignment state and packet as one process on the whole stream."""
    res = _spawn(_worker_blockshard, world=3, timeout=280, extra=(None, root, cut))
    assert all(r[1] for r in res), res
    assert [r[2] for r in res] == [27 if r[0] == root else 0 for r in res]


@pytest.mark.timeout(400)
def test_block_round_robin_world8_repairs_slips_at_ownership_changes():
    """Eight owners (C2-shaped work on a whole node), the alignment done by the owners: a symbol slip planted in the first
    block of rank 0's second turn (block 8: its predecessor is rank 7's), in blocks 15, 16 (both sides of the next change from rank 7 to rank 0) and 23 is repaired exactly as
    one process repairs it -- the tails forwarded between owners carry everything the repair needs."""
    slips = {8: 1, 15: 1, 16: 1, 23: 1}
    res = _spawn(_worker_blockshard, world=8, timeout=380, extra=(slips,))
    assert all(r[1] for r in res), res
    root = [r for r in res if r[0] == 0][0]
    assert root[2] == 27 and root[3] >= len(slips)        # every slip moved at least one block boundary, and all were repaired


def _worker_grid(rank, world, port, q, bin_ranks):
    """Doppler bins x time chunks: groups of ``bin_ranks`` processes shard the bins, the groups take the blocks round-robin.
    Process 0's results must equal one process on the whole stream with the whole bin table."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import torch.distributed as dist
    from oracle import mfbank_oracle as orc
    from oracle_bank import OracleBank
    import pycusdr_amd.demodulator.demodulator_base as dbm
    from pycusdr_amd import config as cfg
    from pycusdr_amd.decoder import Decoder
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from pycusdr_amd.dist import GridShard
    from pycusdr_amd.protocol import loadProtocol
    dbm.MFBank = OracleBank
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    bs, ov, D = 13, 1 << 10, 12
    N = 1 << bs
    nblocks = 27
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig, _ = _hopping_stream(N, ov, nblocks, np.linspace(-3000, 3000, nblocks))
    step = N - ov
    chunks = [sig[ov + i * step: ov + (i + 1) * step] for i in range(nblocks)]
    grid = GridShard(bin_ranks)
    assert (grid.g, grid.b) == divmod(rank, bin_ranks) and grid.T * grid.B == world
    run = DemodulatorRunner(conf, p, 'UHF-H', shard=grid.doppler)
    assert run.demod.bank.D == len(range(*grid.doppler.bin_range(D)))          # this process holds its slice only
    run.raw[:ov] = sig[:ov]
    res, packets = grid.run(run, chunks, decoder=Decoder({}, p, correlator=orc.sync_correlate))
    ok = True
    picked = set()
    if rank == 0:
        plain = DemodulatorRunner(conf, p, 'UHF-H')
        plain.raw[:ov] = sig[:ov]
        ref, ref_packets = plain.run(chunks, decoder=Decoder({}, p, correlator=orc.sync_correlate))
        ok &= len(res) == len(ref) == nblocks
        for a, b in zip(res, ref):
            ok &= a['count'] == b['count'] and a['numSyncSig'] == b['numSyncSig']
            ok &= bool(np.array_equal([a['doppler'], a['doppler_std'], a['SNR'], a['spSymEst']],
                                      [b['doppler'], b['doppler_std'], b['SNR'], b['spSymEst']], equal_nan=True))
            ok &= bool(np.array_equal(a['data'], b['data']) and np.array_equal(a['trust'], b['trust']))
            picked.add(round(a['doppler'], -2))
        ok &= bool(np.array_equal(run.demod.poswinP, plain.demod.poswinP) and np.array_equal(run.demod.posSymEnd, plain.demod.posSymEnd))
        ok &= len(packets) == len(ref_packets) == 1 and bool(np.array_equal(packets[0].bits, ref_packets[0].bits))
        ok &= packets[0].checkPacketData() == ref_packets[0].checkPacketData()
        ok &= len(picked) > 3                      # the carrier really moved through the slices
    else:
        ok &= res == [] and packets == [] and run.count == nblocks
    q.put((rank, bool(ok), len(res)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize('world,bin_ranks', [(4, 2), (6, 3), (6, 2)])
def test_bins_by_blocks_grid_equals_single_process_stream(world, bin_ranks):
    res = _spawn(_worker_grid, world=world, timeout=380, extra=(bin_ranks,))
    assert all(r[1] for r in res), res
    assert res[0][2] == 27 and all(r[2] == 0 for r in res[1:])


def _worker_blockshard_failure(rank, world, port, q):
    """Rank 1 fails in the middle of the stream (its device call raises): it must leave ``run`` with that exception and
    nothing in flight, and the other ranks must leave too -- with an exception that names the failed rank -- instead of
    sitting in a receive for ever; a suspended watchdog does not kill a caller that idles afterwards."""
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import torch.distributed as dist
    from oracle_bank import OracleBank
    import pycusdr_amd.demodulator.demodulator_base as dbm
    from pycusdr_amd import config as cfg
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from pycusdr_amd.dist import BlockShard, StepWatchdog
    from pycusdr_amd.protocol import loadProtocol
    dbm.MFBank = OracleBank
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    bs, ov, D = 13, 1 << 10, 8
    N = 1 << bs
    nblocks = 12
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig, _ = _hopping_stream(N, ov, nblocks, np.zeros(nblocks))
    step = N - ov
    chunks = [sig[ov + i * step: ov + (i + 1) * step] for i in range(nblocks)]
    shard = BlockShard()
    run = DemodulatorRunner(conf, p, 'UHF-H')
    run.raw[:ov] = sig[:ov]
    if rank == 1:
        inner, calls = run.feed_device, [0]

        def failing(x=None):
            calls[0] += 1
            if calls[0] == 3:
                raise OSError('planted device failure')
            return inner(x)
        run.feed_device = failing
    dog = StepWatchdog(1.5, rank=rank, describe=shard.describe, first_grace_s=60.0)
    what = 'returned'
    t0 = time.time()
    try:
        shard.run(run, chunks, watchdog=dog)
    except OSError as e:
        what = f'OSError: {e}'
    except RuntimeError as e:
        what = f'RuntimeError: {e}'
    took = time.time() - t0
    time.sleep(2.5)            # longer than the watchdog's limit: run() suspended it on the way out
    dog.stop()
    q.put((rank, what, took < 60.0))
    try:
        dist.destroy_process_group()
    except Exception:       # noqa: BLE001
        pass


@pytest.mark.timeout(200)
def test_block_round_robin_failure_leaves_no_rank_waiting():
    res = _spawn(_worker_blockshard_failure, world=3, timeout=150)
    by = {r[0]: r for r in res}
    assert by[1][1].startswith('OSError: planted') and all(r[2] for r in res), res
    # the root waits for rank 1's finished blocks, rank 2 for rank 1's tails: both are told
    assert by[0][1].startswith('RuntimeError') and ('failure' in by[0][1] or 'arrived where' in by[0][1]), res
    assert by[2][1].startswith('RuntimeError'), res


def test_block_shard_wire_format_round_trip():
    """What travels under time-chunk sharding: the tail of a block (owner -> next owner) and the finished block (owner -> root);
    every array comes back with its dtype and values."""
    from pycusdr_amd.dist import BlockShard
    rs = np.random.RandomState(3)
    S = 777
    for dt in (np.uint8, np.bool_, np.float64):
        tail = {'post': rs.randint(0, 2, 33).astype(dt), 'end': rs.randint(0, 2, 21).astype(dt), 'exact': True}
        th, tb = BlockShard.pack_tail(17, tail)
        assert th.dtype == np.float64 and len(th) == BlockShard.TAIL_HEADER and tb.dtype == np.uint8 and len(tb) == 54
        i, back = BlockShard.unpack_tail(th, tb)
        assert i == 17 and back['exact'] and all(back[k].dtype == dt and np.array_equal(back[k], tail[k]) for k in ('post', 'end'))
    with pytest.raises(TypeError):
        BlockShard.pack_tail(0, {'post': np.zeros(3, np.int32), 'end': np.zeros(3, np.int32), 'exact': True})
    d = {'count': 41, 'doppler': -123.456, 'doppler_std': 7.5, 'SNR': float('nan'), 'spSymEst': 15.987654321, 'time_ms': 2.5,
         'timestamp': 1759560000.123456,        # the owner's block stamp travels with the block (epoch seconds, float64: exact to ~0.2 us)
         'data': rs.randint(0, 2, S).astype(np.uint8), 'trust': rs.randint(0, 256, S).astype(np.uint8)}
    sh = BlockShard.__new__(BlockShard)
    head, body = BlockShard.pack(sh, d, tail)
    assert head.dtype == np.float64 and len(head) == BlockShard.HEADER and body.dtype == np.uint8 and len(body) == 2 * S + 54
    back = BlockShard.unpack(head, body)
    assert back['count'] == 41 and back['doppler'] == -123.456 and np.isnan(back['SNR']) and back['spSym'] == d['spSymEst']
    assert np.array_equal(back['bits'], d['data']) and np.array_equal(back['trust'], d['trust']) and back['spent'] == 2.5e-3
    assert back['timestamp'] == d['timestamp']
    assert all(np.array_equal(back['tail'][k], tail[k]) and back['tail'][k].dtype == tail[k].dtype for k in ('post', 'end'))
    empty = dict(d, data=d['data'][:0], trust=d['trust'][:0])
    h2, b2 = BlockShard.pack(sh, empty, {'post': tail['post'][:0], 'end': tail['end'][:0], 'exact': True})
    assert len(b2) == 0 and len(BlockShard.unpack(h2, b2)['bits']) == 0
