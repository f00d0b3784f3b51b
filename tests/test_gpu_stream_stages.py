"""The stream-stage kernels (csrc/stream_kernels.hpp: A12 bit lookup / NRZ-S decode, A13 block-overlap alignment, A14 the decoder's
searches on the stream without a stash, the would-be stash edges) driven through their test seam on INJECTED symbol decisions,
against the host code they replace -- ``Demodulator.demodulateHost`` (extractBits / extractBitsNRZs / checkSymbolOverlap, reference
DB:863-1051) and ``np.convolve`` (decoder.py:96-113) -- bit for bit: planted +-1 symbol slips in both directions (the repair moves the
window), bit errors in the overlap (the match thresholds), impossible NRZ-S transitions, and irregular blocks (symbol indices outside
the LUT, missing first / last centres, tiny blocks), which the device must hand to the host."""
import numpy as np
import pytest

from pycusdr_amd import config as cfg
from pycusdr_amd.demodulator import UHF
from pycusdr_amd.protocol import loadProtocol

pytestmark = pytest.mark.gpu

BS, OV, SPS = 15, 1 << 10, 16
N = 1 << BS
STEP_SYM = (N - OV) // SPS            # symbols a block advances by
NSYM = N // SPS                       # symbols decided per block


def _blocks(rs, mode, nblocks, lut, slips, irregular, p_err=0.04):
    """Symbol decisions of `nblocks` consecutive blocks of one symbol stream.  Block b sees global symbols S_b + x + slip_b at
    centres x * SPS + phase: a slip shifts WHICH symbol sits at a position, not the positions -- what a symbol-clock slip at a
    block edge looks like to checkSymbolOverlap."""
    total = nblocks * STEP_SYM + NSYM + 8
    if mode == 'lut':
        gbits = rs.randint(0, 2, total)
        classes = [np.where(lut == v)[0] for v in (0, 1)]
    else:                             # NRZ-S: a random walk through the LUT's successor sets, a few impossible steps
        rows, _, succ = lut.shape
        g = np.zeros(total, dtype=np.int64)
        g[0] = rs.randint(0, rows)
        for i in range(1, total):
            if rs.rand() < 0.01:
                g[i] = rs.randint(0, rows)
            else:
                g[i] = lut[g[i - 1], rs.randint(0, 2), rs.randint(0, succ)]
    out = []
    for b in range(nblocks):
        count = NSYM - int(rs.randint(0, 3))
        x = np.arange(NSYM)
        phase = rs.randint(0, SPS)
        cen = (x * SPS + phase + rs.randint(-1, 2, NSYM)).astype(np.int32)
        idx = np.clip(b * STEP_SYM + x + slips.get(b, 0) + 4, 0, total - 1)
        if mode == 'lut':
            bits = gbits[idx].copy()
            flip = rs.rand(NSYM) < p_err
            bits[flip] ^= 1
            sym = np.array([classes[v][rs.randint(0, len(classes[v]))] for v in bits], dtype=np.int32)
        else:
            sym = g[idx].astype(np.int32)
            bad = rs.rand(NSYM) < p_err / 4
            sym[bad] = rs.randint(0, lut.shape[0], int(bad.sum()))
        kind = irregular.get(b)
        if kind == 'negative':
            sym[rs.randint(100, 1900)] = -1
        elif kind == 'large':
            sym[rs.randint(100, 1900)] = 99
        elif kind == 'no_end':
            cen[cen > N - OV // 2] = N - OV // 2
        elif kind == 'no_start':
            cen[:] = 3
        elif kind == 'tiny':
            count = 9
        mag = (rs.rand(NSYM).astype(np.float32) * 1e6).astype(np.float32)
        out.append((count, sym, cen, mag))
    return out


def _host_block(host, blk):
    count, sym, cen, mag = blk
    rec = {'spSym': 16.0, 'symbols': sym[:count], 'centres': cen[:count], 'trust': mag.view(np.int8)[:count].copy(),
           'clipped': np.zeros(0, np.int64)}
    try:
        bits, c8, t8, _ = host.demodulateHost(rec)
    except (IndexError, ValueError) as e:          # what the reference does with such a block: the exception leaves the method
        return None, str(e)
    return (np.asarray(bits), np.asarray(c8), np.asarray(t8)), None


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
@pytest.mark.parametrize('mode,pname', [('lut', 'bench_GMSK'), ('nrzs', 'bench_BPSK')])
def test_alignment_and_bits_equal_the_host_code_on_planted_slips(mode, pname, seed):
    rs = np.random.RandomState(11 + 2 * seed + (mode == 'nrzs'))
    conf = cfg.bench_config(pname, blockSize=BS, doppCarrierSteps=8)
    p = loadProtocol(pname)(conf=conf)
    dev, host = UHF.Demodulator(conf, p, 'UHF-H'), UHF.Demodulator(conf, p, 'UHF-H')
    try:
        assert dev.enableStreamStages(None)
        lut = dev._bitLUT_u8 if mode == 'lut' else dev.symbolLUT
        nblocks, B = 40, 5
        if seed == 0:
            slips = {3: 1, 7: -1, 8: -1, 14: 1, 15: 1, 21: -1, 26: 1, 31: -1, 36: 1}           # incl. the first block of a batch (5 | 15)
            irregular = {11: 'negative', 18: 'large', 23: 'no_end', 28: 'tiny', 33: 'no_start'}
        else:               # random places, both directions, runs of slips, every kind of irregular block
            slips = {int(b): int(rs.choice([-1, 1])) for b in rs.choice(np.arange(1, nblocks), 12, replace=False)}
            kinds = ['negative', 'large', 'no_end', 'tiny', 'no_start']
            irregular = {int(b): kinds[int(rs.randint(0, 5))] for b in rs.choice(np.arange(2, nblocks), 4, replace=False)}
        blocks = _blocks(rs, mode, nblocks, lut, slips, irregular)
        moved, device_blocks, host_blocks = 0, 0, 0
        for b0 in range(0, nblocks, B):
            # the device starts every batch from the host's state (a batch that follows an all-regular one could chain on the
            # device's own carry: exercised by the streaming tests; here every batch is checked against an explicit seed)
            dev.poswinP, dev.posSymEnd = host.poswinP, getattr(host, 'posSymEnd', [])
            assert dev.seedStreamStages()
            group = blocks[b0:b0 + B]
            R = dev.bank.debug_stream_stages([g[0] for g in group], np.stack([g[1] for g in group]), np.stack([g[2] for g in group]),
                                             np.stack([g[3] for g in group]))
            s = R.s
            for i, blk in enumerate(group):
                b = b0 + i
                got, err = _host_block(host, blk)
                st = s['a13_status'][i]
                if b in irregular:
                    assert st == 0, (b, irregular[b], st)            # the device hands every irregular block to the host
                if st == 0:
                    host_blocks += 1
                    continue
                assert err is None, (b, err)
                device_blocks += 1
                nw = s['a13_nwin'][i]
                assert nw == len(got[0]), (b, nw, len(got[0]), s['a13_start'][i])
                assert np.array_equal(R.bits[i, :nw], got[0]), b
                assert np.array_equal(R.cen8[i, :nw], got[1]) and np.array_equal(R.trust[i, :nw], got[2]), b
                assert np.array_equal(R.post[i, :s['a13_npost'][i]], np.asarray(host.poswinP).astype(np.uint8)), b
                assert np.array_equal(R.end[i, :s['a13_nend'][i]], np.asarray(host.posSymEnd).astype(np.uint8)), b
                # did the repair move this window?  (first centre >= ov/2 is the unrepaired start)
                cnt, _, cen, _ = blk
                first = int(np.argmax(cen[:cnt] >= OV // 2))
                moved += int(s['a13_start'][i] != first)
        # the planted slips were really repaired on the device (a slip right behind an irregular block goes to the host with it)
        assert moved >= 4, moved
        assert device_blocks >= nblocks - 2 * len(irregular) - 2 and host_blocks >= len(irregular), (device_blocks, host_blocks)
    finally:
        dev.close()
        host.close()


@pytest.mark.parametrize('taps,thr,seed', [((40, 12), (9, 5), 5), ((128, 16), (20, 7), 6), ((33, 64), (8, 11), 7), ((256, 31), (23, 8), 8),
                                           ((300, 12), (30, 5), 9)])
def test_sync_hits_ring_and_edges_equal_np_convolve(taps, thr, seed):
    """A14 on injected decisions with templates that fire often: every block's hits on the stream without a stash, the ring carried
    from batch to batch on the device, and the leading positions of the would-be stash streams -- against np.convolve on the bit
    sequence the host code produces.  Templates of up to 256 taps take the packed kernel (k_stream_search: popcounts on a
    bit-packed stream, word boundaries at 32 / 33 / 64 / 256 taps); 300 taps the byte kernels (no edges there)."""
    rs = np.random.RandomState(seed)
    conf = cfg.bench_config('bench_GMSK', blockSize=BS, doppCarrierSteps=8)
    p = loadProtocol('bench_GMSK')(conf=conf)
    dev, host = UHF.Demodulator(conf, p, 'UHF-H'), UHF.Demodulator(conf, p, 'UHF-H')
    try:
        t0 = (2 * rs.randint(0, 2, taps[0]) - 1).astype(np.int8)
        t1 = (2 * rs.randint(0, 2, taps[1]) - 1).astype(np.int8)
        t1[rs.randint(0, taps[1])] = 0                   # a tap of 0 belongs to neither mask
        nOv = 96 if taps[0] < 100 else 320
        dev._stages, dev._stage_decoder = True, None
        dev.bank.set_stream_stages(dev.sigOverlap, dev.overlapOffset, dev.symbol_check_match_threshold, dev.symbol_check_error_threshold,
                                   bit_lut=dev._bitLUT_u8, templates=(t0, t1), thresholds=thr, bits_overlap=nOv)
        nblocks, B = 12, 4
        blocks = _blocks(rs, 'lut', nblocks, dev._bitLUT_u8, {5: 1}, {})
        seq = np.zeros(nOv, dtype=np.int64)              # the decoder's stream so far (it starts with numBitsOverlap zeros, DEC:42)
        dev.bank.stream_seed(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(nOv, np.uint8))
        checked_edges = 0
        for b0 in range(0, nblocks, B):
            group = blocks[b0:b0 + B]
            vstart = len(seq) - nOv              # the oldest bit the device holds for this batch (its ring)
            R = dev.bank.debug_stream_stages([g[0] for g in group], np.stack([g[1] for g in group]), np.stack([g[2] for g in group]),
                                             np.stack([g[3] for g in group]))
            for i, blk in enumerate(group):
                got, err = _host_block(host, blk)
                assert err is None and R.s['a13_status'][i] == 1 and R.s['sync_valid'][i] == 1, (b0 + i, err)
                bits = got[0].astype(np.int64)
                window = np.concatenate((seq[-nOv:], bits))
                for k, (t, h) in enumerate(((t0, thr[0]), (t1, thr[1]))):
                    score = np.convolve(window, t.astype(np.int64))
                    want = np.where(score >= h)[0]
                    c = R.s['sync_count'][i][k]
                    assert c == len(want), (b0 + i, k, c, len(want), R.hits[i, k, 0, :min(c, R.max_hits)].tolist(), want.tolist(), len(window))
                    c = min(c, R.max_hits)               # (a count beyond what the record holds is reported, the first max_hits kept)
                    assert np.array_equal(R.hits[i, k, 0, :c], want[:c]) and np.array_equal(R.hits[i, k, 1, :c], score[want[:c]]), (b0 + i, k)
                # the leading positions of the stream a FIXED-mode decoder would restart at, for the first header hits
                base = len(seq) - nOv                     # position of the window's first bit in `full`
                full = np.concatenate((seq, bits))
                hdr = np.where(np.convolve(window, t0.astype(np.int64)) >= thr[0])[0]
                E, eh = R.edges[i], R.edge_hits
                for cidx in range(min(4, len(hdr))):
                    a_rel = int(hdr[cidx]) - len(t0) + 1 - 20
                    assert E[cidx, 0] == a_rel
                    start = base + a_rel
                    if not E[cidx, 1]:
                        n_lead = [int((np.convolve(full[max(start, 0):max(start, 0) + max(len(t0), len(t1)) - 1], t.astype(np.int64))[:len(t) - 1] >= h).sum())
                                  for t, h in ((t0, thr[0]), (t1, thr[1]))]
                        # not computed only where it cannot be: outside what the device holds, too many hits, templates too long
                        assert start < vstart or start + max(len(t0), len(t1)) - 1 > len(full) or max(n_lead) > eh or max(taps) > 256, (b0 + i, cidx)
                        continue
                    assert start >= 0
                    lead = full[start:start + max(len(t0), len(t1)) - 1]
                    for k, (t, h) in enumerate(((t0, thr[0]), (t1, thr[1]))):
                        sc = np.convolve(lead, t.astype(np.int64))[:len(t) - 1]
                        want = np.where(sc >= h)[0]
                        n = E[cidx, 2 + k]
                        assert n == len(want), (b0 + i, cidx, k, n, len(want))
                        assert np.array_equal(E[cidx, 4 + k * eh:4 + k * eh + n], want)
                        assert np.array_equal(E[cidx, 4 + 2 * eh + k * eh:4 + 2 * eh + k * eh + n], sc[want])
                    checked_edges += 1
                seq = full
        assert checked_edges >= (8 if max(taps) < 200 else 4) or max(taps) > 256
    finally:
        dev.close()
        host.close()
