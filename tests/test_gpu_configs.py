"""BASELINE.json configurations at their own workload: C5 (two concurrent demodulator instances, 512 bins
each, N=2^20; both handles in one process and as two processes on one device), C3 (1024 bins, real GMSK bank,
widened Doppler span), the multi-GPU sharded search (runs when >= 2 devices are visible), and the C-ABI's
resource handling (failed creation leaks nothing, concurrent sync-correlator calls)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.mfbank import MFBank, sync_correlate, sync_find
from pycusdr_amd.protocol import loadProtocol
from ports import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ok(res, name):
    assert res['parseval'] < 1e-5, (name, res)           # north_star tolerance on magnitudes, all 512 bins
    assert res['real'] < 1e-5, (name, res)                # real oracle IFFTs on a sample of bins
    assert res['pick_exact'] and res['zeros_ok'], (name, res)
    assert res['carrier_err_bins'] <= 1.0, (name, res)    # the carrier sits where the stimulus puts it


def test_c5_two_instances_one_process():
    """C5: CC11xx FSK-2 (sps 128, M=8) and BPSK (M=32, 16 unique up to sign), D=512 each, N=2^20, matching
    stimuli, both handles alive and used alternately."""
    from c5_common import c5_instance, check_instance
    a = c5_instance('CC11xx', 20, 512)
    b = c5_instance('bench_BPSK', 20, 512)
    ba = MFBank(20, 512, a['M'])
    bb = MFBank(20, 512, b['M'])
    try:
        ba.set_filters(a['masks'])
        bb.set_filters(b['masks'])
        ba.set_shifts(a['shifts'])
        bb.set_shifts(b['shifts'])
        assert ba.get_search_path()['path'] == 'segment' and ba.get_search_path()['taps'] == 384
        assert bb.get_search_path()['path'] == 'segment' and bb.get_search_path()['taps'] == 80 and bb.get_info()[2] == 16
        ba.upload(a['x'])
        bb.upload(b['x'])
        ia = ba.find_carrier()
        ib = bb.find_carrier()
        ra = check_instance(ba, a)
        rb = check_instance(bb, b)
        assert float(ia[0]) == ra['idx'] and float(ib[0]) == rb['idx']      # interleaving changes nothing
    finally:
        ba.close()
        bb.close()
    _ok(ra, 'CC11xx')
    _ok(rb, 'bench_BPSK')


def test_c5_instances_on_shares_of_the_compute_units():
    """mfb_set_cu_share: the two C5 instances on the even and the odd compute units of one device, searching at the same time in two
    threads -- every table bit for bit the whole-device one (a score is a function of the block, the shift and the filter only), the
    oracle gates as above; the share can be changed back, bad shares and a block in flight are refused."""
    from c5_common import c5_instance, check_instance
    from pycusdr_amd._lib import MFBankError
    insts = [c5_instance('CC11xx', 20, 512), c5_instance('bench_BPSK', 20, 512)]
    banks = [MFBank(20, 512, i['M']) for i in insts]
    try:
        whole = []
        for bk, inst in zip(banks, insts):
            bk.set_filters(inst['masks'])
            bk.set_shifts(inst['shifts'])
            bk.upload(inst['x'])
            bk.find_carrier()
            whole.append(bk.get_scores().copy())
        for k, bk in enumerate(banks):
            bk.set_cu_share(k, 2)
        got = [None, None]

        def search(k):
            for _ in range(6):
                banks[k].upload(insts[k]['x'])
                banks[k].find_carrier()
            got[k] = banks[k].get_scores().copy()
        ths = [threading.Thread(target=search, args=(k,)) for k in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for k in range(2):
            assert np.array_equal(got[k], whole[k]), k
        res = [check_instance(bk, inst) for bk, inst in zip(banks, insts)]
        with pytest.raises(ValueError):
            banks[0].set_cu_share(2, 2)
        with pytest.raises(ValueError):
            banks[0].set_cu_share(0, 0)
        banks[0].set_cu_share(0, 1)                      # the whole device again
        banks[0].upload(insts[0]['x'])
        banks[0].find_carrier()
        assert np.array_equal(banks[0].get_scores(), whole[0])
        k_off, k_len = 1000, 4000
        banks[0].begin_block(0, k_off, k_len, 8, source='uploaded')
        with pytest.raises(MFBankError):
            banks[0].set_cu_share(0, 2)                  # a block is in flight
        banks[0].end_block(0)
    finally:
        for bk in banks:
            bk.close()
    _ok(res[0], 'CC11xx')
    _ok(res[1], 'bench_BPSK')


def test_c5_two_processes_one_device():
    """C5 as the reference deploys it: one OS process (and device context) per demodulator instance, both on
    the same device at the same time."""
    child = os.path.join(ROOT, 'tests', 'children', 'c5_child.py')
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, child, name, '20', '512', '6'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              env=env, text=True) for name in ('CC11xx', 'bench_BPSK')]
    outs = [p.communicate(timeout=500) for p in procs]
    for p, (so, se), name in zip(procs, outs, ('CC11xx', 'bench_BPSK')):
        assert p.returncode == 0, se[-2000:]
        res = json.loads(so.strip().splitlines()[-1])
        _ok(res, name)
        assert res['picks_equal'] and res['path']['path'] == 'segment'


@pytest.mark.parametrize('path', ['segment', 'twopass'])
def test_c3_1024_bins_gmsk_bank(path):
    """C3: D=1024 at N=2^20 with the real GMSK bank and the Doppler span widened until the 1024 shifts are
    distinct, S1 stimulus; both search paths."""
    from bench import widen_range_rate
    log2N, D = 20, 1024
    N = 1 << log2N
    conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=D)
    rr, shifts = widen_range_rate(conf, 'UHF-H', N, D)
    assert len(np.unique(shifts)) == D
    M, masks = loadProtocol('bench_GMSK')(conf=conf).get_filter(N, 16, 3)
    x = sg.s1_stream(1, N, 1 << 10, 'GMSK', snr_db=10.0, seed=1)[:N]
    bank = MFBank(log2N, D, M)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.set_search_path(path)
        bank.upload(x)
        idx, metric = bank.find_carrier()
        ds = bank.get_scores()
        X = bank.get_spectrum()
    finally:
        bank.close()
    pv = orc.doppler_scores_parseval(X, masks, shifts)
    assert np.abs(ds[:, 0] - pv).max() / pv.max() < 1e-5
    sel = [0, 1, 300, 511, 512, 1023]
    ref = orc.doppler_scores(X, masks, shifts[sel], True)[:, 0]
    assert np.abs(ds[sel, 0] - ref).max() / ref.max() < 1e-5
    oidx, _ = orc.find_doppler_est(ds, D, 0, True)
    assert idx == oidx
    pick = orc.interpolate_doppler(idx, shifts, np.zeros(D))
    assert abs(pick['dopplerIdxlast'] - N // 4) <= np.median(np.diff(np.sort(shifts)))


def _run_ranks(n, args, timeout=900, child='dist_child.py'):
    import re
    child = os.path.join(ROOT, 'tests', 'children', child)
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), child] + [str(a) for a in args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads(m) for m in re.findall(r'\{[^{}]*\}', r.stdout)]      # the ranks' lines may interleave
    assert len(res) == n and all(q['ok'] for q in res), (res, r.stderr[-2000:])
    return res


def test_multi_gpu_sharded_pick_equals_unsharded():
    """Sharded search on every visible GPU (one fresh process per rank via torch.distributed.run, RCCL):
    broadcast of rank 0's block, search, one all-gather / all-reduce, pick -- equal, bit for bit, to the unsharded
    search; every rank demodulates the same bits.  On an 8-GPU node also BASELINE config C4 itself: 2048 bins, 256
    per GPU, N = 2^20, and the same with a noise-reference bin."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip('needs >= 2 GPUs (the driver runs it on the 8-GPU node)')
    _run_ranks(min(n, 4), ['nccl'])
    if n >= 8:
        assert all(q['even'] for q in _run_ranks(8, ['nccl', 20, 2048]))
        _run_ranks(8, ['nccl', 20, 2048, 1])
        _run_ranks(8, ['nccl', 20, 512], child='c5_dist_child.py')        # C5: two instances, 512 bins each, 8 GPUs


def test_two_ranks_sharing_one_gpu_over_gloo():
    """The sharded path with TWO ranks on the GPU box's single device (gloo moves the device tensors): broadcast of
    rank 0's block, search of each rank's bin slice on its own handle, exchange, pick on both ranks -- equal, bit
    for bit, to the unsharded search; the same bits demodulated on both ranks.  Everything but RCCL itself."""
    _run_ranks(2, ['gloo'])


def test_two_ranks_uneven_slices_and_noise_bin_over_gloo():
    """World 2, 129 bins (65 + 64: the all-reduce form of the exchange on device tensors) plus the reference's
    noise-reference bin (DB:148-159), which every rank searches beside its slice and rank 0's copy of which enters the
    table the pick runs on (CU:550-554)."""
    res = _run_ranks(2, ['gloo', 16, 129, 1])
    assert all(q['noise_rows'] == 1 and not q['even'] for q in res)


def test_block_round_robin_two_ranks_one_gpu():
    """Time-chunk sharding (dist.BlockShard) with two ranks on the one device: blocks dealt round-robin, device stages on
    the owner's handle, host stages and decoder on rank 0 in block order -- the same bit stream, alignment state and packet
    (zero bit errors) as one process on the whole stream."""
    res = _run_ranks(2, [15], child='block_child.py')
    assert res[0]['blocks'] > 6
    _run_ranks(3, [16], child='block_child.py')


def test_c5_two_sharded_instances_per_rank():
    """C5's multi-GPU shape on one device: two ranks, each holding BOTH demodulator instances (CC11xx FSK-2, M=8, 384-tap
    filters; BPSK, M=32) with the instance's bins sharded over the ranks on a communicator of its own; the two streams
    alternate.  Per instance: sharded table and pick equal the unsharded handle's, every rank demodulates the same bits."""
    _run_ranks(2, ['gloo', 17, 64], child='c5_dist_child.py')


def test_c4_slices_four_gloo_ranks_full_size():
    """C4's per-rank geometry in four real processes on the one device (the box admits at most six): N = 2^20, 1024 bins,
    256 per rank, GMSK bank -- sharded pick and table equal the unsharded 1024-bin handle's bit for bit."""
    _run_ranks(4, ['gloo', 20, 1024])


def test_bins_by_blocks_grid_on_one_gpu():
    """north_star's "Doppler-bin x time-chunk grid" (dist.GridShard): four processes on the one device (the box admits six
    with this one) as 2 block groups x 2 bin slices, with even slices (all-gather) and with uneven ones (65 bins: all-reduce)
    -- results, alignment state and the packet (zero bit errors) equal one process with the whole table on the whole stream.
    Over RCCL when the node has the devices."""
    import torch
    res = _run_ranks(4, ['gloo', 2, 15, 64], child='grid_child.py')
    assert sorted((q['group'], q['bin_rank']) for q in res) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    _run_ranks(4, ['gloo', 2, 15, 65], child='grid_child.py')
    if torch.cuda.device_count() >= 8:
        _run_ranks(8, ['nccl', 4, 18, 256], child='grid_child.py')


def test_failed_create_frees_everything():
    """mfb_create that fails at ANY of its allocations must hand back every byte it took -- the caller gets no
    handle to destroy (reference teardown DB:517-530).  mfb_debug_fail_alloc makes the n-th allocation fail."""
    import torch
    from pycusdr_amd import _lib
    lib = _lib.load()
    MFBank(14, 8, 4).close()                 # code objects, context: resident before the baseline is read
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    failed = 0
    for nth in range(1, 40):
        lib.mfb_debug_fail_alloc(nth)
        try:
            MFBank(18, 64, 8).close()        # past the last allocation: creation succeeds
            break
        except MemoryError:
            failed += 1
        finally:
            lib.mfb_debug_fail_alloc(0)
    assert failed >= 15                      # every buffer of the handle was walked
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < (32 << 20), (free0, free1)
    # the same walk through mfb_set_filters (segment spectra, intermediate): a failed call leaves a usable handle
    rs = np.random.RandomState(1)
    bank = MFBank(14, 4, 2)
    masks = (rs.standard_normal((2, 1 << 14)) + 1j * rs.standard_normal((2, 1 << 14))).astype(np.complex64)
    lib.mfb_debug_fail_alloc(1)
    with pytest.raises(MemoryError):
        bank.set_filters(masks)
    lib.mfb_debug_fail_alloc(0)
    bank.set_filters(masks)
    bank.set_shifts([1, 2, 3, 4])
    bank.upload(masks[0])
    bank.find_carrier()
    want = bank.get_scores().copy()
    # ... and the same with a C++ exception where the allocation was (nth < 0: std::bad_alloc, what a failed HOST allocation in the
    # filter analysis looks like): a status code, not a terminated process; nothing left behind by a failed create; the handle takes
    # the filters again and gives the same table
    bank.close()
    bank = MFBank(14, 4, 2)                  # (a fresh handle: the first mfb_set_filters is the one that allocates)
    lib.mfb_debug_fail_alloc(-1)
    try:
        with pytest.raises(MemoryError):
            bank.set_filters(masks)
    finally:
        lib.mfb_debug_fail_alloc(0)
    bank.set_shifts([1, 2, 3, 4])
    with pytest.raises(RuntimeError):
        bank.upload(masks[0])                # no filters in force after half a bank: a state error, not a crash
    with pytest.raises(RuntimeError):
        bank.find_carrier()
    bank.set_filters(masks)
    bank.upload(masks[0])
    bank.find_carrier()
    assert np.array_equal(bank.get_scores(), want)
    bank.close()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    thrown = 0
    for nth in range(1, 40):
        lib.mfb_debug_fail_alloc(-nth)
        try:
            MFBank(18, 64, 8).close()
            break
        except MemoryError:
            thrown += 1
        finally:
            lib.mfb_debug_fail_alloc(0)
    assert thrown == failed
    torch.cuda.synchronize()
    free3, _ = torch.cuda.mem_get_info()
    assert abs(free2 - free3) < (32 << 20), (free2, free3)


def test_two_handles_and_concurrent_sync_calls():
    """Two handles on one device used from two threads, each interleaving searches with sync-correlator
    calls: per-call workspaces and streams, results equal to the single-threaded ones."""
    rs = np.random.RandomState(8)
    log2N, D, M = 14, 9, 4
    N = 1 << log2N
    jobs = []
    for t in range(2):
        masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
        x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
        shifts = rs.randint(0, N, D).astype(np.int32)
        bits = rs.randint(0, 2, (64, 30000 + 777 * t)).astype(np.uint8)
        tmpl = (2 * rs.randint(0, 2, 64) - 1).astype(np.int8)
        jobs.append((masks, x, shifts, bits, tmpl))
    expect = []
    for masks, x, shifts, bits, tmpl in jobs:
        X = orc.forward_fft(x)
        expect.append((orc.doppler_scores(X, masks, shifts, True), np.stack([np.convolve(b.astype(np.int64), tmpl) for b in bits])))
    errors = []

    def work(t):
        try:
            masks, x, shifts, bits, tmpl = jobs[t]
            bank = MFBank(log2N, D, M)
            bank.set_filters(masks)
            bank.set_shifts(shifts)
            for _ in range(6):
                bank.upload(x)
                bank.find_carrier()
                ds = bank.get_scores()
                sc = sync_correlate(bits, tmpl)
                hits = sync_find(bits, tmpl, 20)
                assert np.abs(ds - expect[t][0]).max() / expect[t][0].max() < 1e-5
                assert np.array_equal(sc, expect[t][1])
                for b in range(0, 64, 13):
                    assert np.array_equal(hits[b][0], np.where(expect[t][1][b] >= 20)[0])
            bank.close()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_three_channels_in_three_threads_one_device():
    """The reference's 3-channel receive config (config/benchmark/bench_3_chan_rx_base.json) runs one Demodulator_process
    per channel; here three runners, each with its own handle, stream and decoder, work concurrently from three threads
    of ONE process on one device.  Every channel must produce exactly what it produces alone."""
    import threading
    from pycusdr_amd import signals as sg
    from pycusdr_amd.decoder import Decoder
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    bs, ov = 15, 1 << 10
    N = 1 << bs
    mods = [('GMSK', 'bench_GMSK'), ('FSK', 'bench_FSK'), ('BPSK', 'bench_BPSK')]
    streams = {}
    for k, (mod, _) in enumerate(mods):
        s = sg.awgn(np.concatenate((sg.get_padded_packet(mod)[0], np.zeros(2 * N))), 12.0, rng=np.random.RandomState(40 + k))
        streams[mod] = s.astype(np.complex64)

    def run_channel(mod, pname, out):
        try:
            conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=48)
            proto = loadProtocol(pname)(conf=conf)
            run = DemodulatorRunner(conf, proto, 'UHF-H')
            sig = streams[mod]
            res, packets = run.run_stream((sig[i:i + 4096] for i in range(0, len(sig), 4096)), decoder=Decoder(conf, proto))
            run.close()
            out[mod] = ([(r['doppler'], r['spSymEst'], r['data'].tobytes(), r['trust'].tobytes()) for r in res],
                        [p.checkPacketData() for p in packets])
        except BaseException as e:      # noqa: BLE001 -- reported by the assertion below
            out[mod] = e

    alone, together = {}, {}
    for mod, pname in mods:
        run_channel(mod, pname, alone)
    threads = [threading.Thread(target=run_channel, args=(mod, pname, together)) for mod, pname in mods]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for mod, _ in mods:
        assert not isinstance(alone[mod], BaseException), alone[mod]
        assert not isinstance(together.get(mod), BaseException), together.get(mod)
        assert together[mod] == alone[mod], mod
        assert alone[mod][1] == [0], (mod, alone[mod][1])          # one packet, no bit errors
