#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the *importable numpy parts* of the
reference (pyCuSDR, mounted read-only at /root/reference in the authoring container).

This script is test infrastructure.  It runs ONLY in the authoring container (the reference does
not exist on the GPU box); its outputs (``*.npz`` next to this file) are committed and are what
the tests read.  Nothing of the reference's source text is stored -- only inputs and outputs.

What is captured (SURVEY.md section 8c, G1..G5 + host-logic KATs):
  G1  filter banks            protocol.*.get_filter          (templates, small banks, digests)
  G2  symbol LUTs             protocol.*.get_symbolLUT2
  G3  decoder templates       protocol.*.get_mask / get_syncFlag (+ tolerances, counts)
  G4  decoder KATs            decoder.Decoder.findFrames on seeded bit streams
  G5  stimulus                examples/benchmark/create_signals.get_padded_packet
  G7  host-logic KATs         Demodulator.checkSymbolOverlap / extractBitsNRZs (pure numpy
                              methods, called unbound on a plain namespace object)
  G8  PN9 whitening bytes     lib.shift_registers.PN9 (CC11xx framing, next-scope row N2)
  G9  peak clipping           Demodulator.__thresholdInput (name-mangled, unbound on a namespace)
  G10 SNR estimate            Demodulator.computeSNR on a numpy spectrum (cuda.Context.synchronize stubbed)
  G11 FLAGS-mode decoder      Decoder.findFrames with packetEndDetectMode = FLAGS (frame inside one call,
                              across calls, overflow past maxPacketLenBits)
  G12 CC11xx packet parsing   PacketCC11xx: length cut, PN9 de-whitening, CRC flag
  G14 ring buffer             sigFIFO.RingBuffer insert / popBlock traces (zmq stubbed: the class is pure numpy)
  G13 bit-stream alignment    lib.customXCorr.customXCorr on the reference's own unit-test streams
                              (test/test_trustProcessor/bitData_test.npz) -- next-scope row N4

Harness-side shims (the same ones SURVEY.md 8c lists): numpy aliases removed in numpy>=1.24
(np.float, np.int); a ``crcmod`` module whose ``mkCrcFun`` is this harness's own bitwise CRC (crcmod
1.7 is a third-party dependency absent from this image; its published algorithm -- polynomial with the
leading 1, MSB-first when rev=False, initial value, final XOR -- is restated below and checked against
the catalogue value 0xAEE7 of CRC-16/CMS); and, for G7/G9/G10 only, empty ``pycuda`` / ``lib.cufft``
modules so that the module holding the pure-numpy methods imports (both are glue to closed-source CUDA
libraries that cannot exist here; ``cuda.Context.synchronize`` is a no-op stub).
No GPU code of the reference is executed (none can be: there is no CUDA here).

Usage:  python tests/golden/make_golden.py      (writes tests/golden/*.npz)
"""
import hashlib
import os
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _install_shims():
    sys.dont_write_bytecode = True
    np.float = float  # noqa: removed aliases the reference still uses
    np.int = int
    crc = types.ModuleType('crcmod')

    def mkCrcFun(poly, initCrc=0, rev=True, xorOut=0):
        """crcmod.mkCrcFun restated for the non-reflected 16-bit case the reference uses."""
        assert not rev and poly >> 16 == 1, 'only the rev=False, 16-bit form is restated'
        p16 = poly & 0xFFFF

        def fun(data):
            c = initCrc
            for byte in bytes(data):
                c ^= byte << 8
                for _ in range(8):
                    c = ((c << 1) ^ p16) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
            return np.int64(c ^ xorOut)      # numpy scalar: np.uint8(...) of it wraps as it did under numpy 1.x
        return fun
    assert mkCrcFun(0x18005, rev=False, initCrc=0xFFFF, xorOut=0)(b'123456789') == 0xAEE7
    crc.mkCrcFun = mkCrcFun
    sys.modules['crcmod'] = crc
    # pycuda is imported at module level by demodulator_base; only needed so the module object
    # exists -- none of its functions are called by the pure-numpy methods we exercise.
    pc = types.ModuleType('pycuda')
    drv = types.ModuleType('pycuda.driver')
    drv.Context = types.SimpleNamespace(synchronize=lambda: None)     # computeSNR calls it before reading the spectrum
    comp = types.ModuleType('pycuda.compiler')
    comp.SourceModule = object
    pc.driver = drv
    pc.compiler = comp
    sys.modules['pycuda'] = pc
    sys.modules['pycuda.driver'] = drv
    sys.modules['pycuda.compiler'] = comp
    sys.modules['zmq'] = types.ModuleType('zmq')      # sigFIFO imports it at module level; RingBuffer never touches it
    sys.path.insert(0, os.path.join(REF, 'pyCuSDR'))
    sys.path.insert(0, os.path.join(REF, 'examples', 'benchmark'))
    # lib/cufft.py raises OSError at import when libcufft is absent (lib/cufft.py:45-46); give the
    # package an empty attribute of that name instead (G7 only; no FFT of the reference is run).
    import lib as reflib
    fake = types.ModuleType('lib.cufft')
    sys.modules['lib.cufft'] = fake
    reflib.cufft = fake


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def time_templates(bank, tlen):
    """Recover the time-domain template each bank row was made from: row = conj(fft(t, N))."""
    t = np.fft.ifft(np.conj(bank.astype(np.complex128)), axis=1)
    return t[:, :tlen]


def g1_g2_g3(out):
    from protocol.benchmark.bench_GMSK import Bench_GMSK
    from protocol.benchmark.bench_FSK import Bench_FSK
    from protocol.benchmark.bench_GFSK import Bench_GFSK
    from protocol.benchmark.bench_BPSK import Bench_BPSK
    from protocol.CC11xx import CC11xx
    from protocol.GFSK2_base import GFSK2
    from protocol.benchmark.bench_base import Bench_base

    bconf = {'Main': {'PacketLen': 10000, 'RandSeed': 123}}
    cconf = {'Radios': {'Protocol': {
        'rx_preamble': ['0xaa'] * 4, 'rx_sync_seq': ['0xd6', '0xba', '0xd6', '0xba'],
        'tx_preamble': ['0xaa'], 'tx_num_preambles': 10,
        'tx_sync_seq': ['0xd6', '0xba', '0xd6', '0xba']}}}

    class _GFSK2(Bench_base, GFSK2):      # GFSK2.get_filter itself (CC11xx's alternative base)
        def get_filter(self, *a):
            return GFSK2.get_filter(self, *a)

    cases = [
        ('bench_GMSK', Bench_GMSK(conf=bconf), 16, 3),
        ('bench_FSK', Bench_FSK(conf=bconf), 16, 3),
        ('bench_GFSK', Bench_GFSK(conf=bconf), 16, 3),
        ('bench_BPSK', Bench_BPSK(conf=bconf), 16, 5),
        ('CC11xx_sps16', CC11xx(conf=cconf), 16, 3),
        ('CC11xx_sps128', CC11xx(conf=cconf), 128, 3),
        ('GFSK2_sps16', _GFSK2(conf=bconf), 16, 3),
    ]
    for name, p, sps, ms in cases:
        nsmall = 1 << 10
        M, bank = p.get_filter(nsmall, sps, ms)
        out[f'g1/{name}/M'] = np.int64(M)
        out[f'g1/{name}/sps'] = np.int64(sps)
        out[f'g1/{name}/maskSize'] = np.int64(ms)
        out[f'g1/{name}/bank_n1024'] = bank
        # time-domain templates recovered at a size that holds them without aliasing
        M2, bank2 = p.get_filter(1 << 12, sps, ms)
        out[f'g1/{name}/templates'] = time_templates(bank2, ms * sps)
        for lg in (16, 20):
            if name not in ('bench_GMSK', 'CC11xx_sps128', 'bench_BPSK') and lg == 20:
                continue
            Mb, big = p.get_filter(1 << lg, sps, ms)
            out[f'g1/{name}/n{lg}_head'] = big[:, :64].copy()
            out[f'g1/{name}/n{lg}_tail'] = big[:, -64:].copy()
            out[f'g1/{name}/n{lg}_stride'] = big[:, ::4099].copy()
            out[f'g1/{name}/n{lg}_sha256'] = np.array(sha(big))
            del big
        if name == 'GFSK2_sps16':      # helper class: filter generator only, no LUT of its own
            continue
        lut = p.get_symbolLUT2(ms)
        out[f'g2/{name}/bitLUT_is_none'] = np.bool_(lut[0] is None)
        if lut[0] is not None:
            out[f'g2/{name}/bitLUT'] = np.asarray(lut[0])
        out[f'g2/{name}/symbolLUT'] = np.asarray(lut[1])
        out[f'g2/{name}/sum_all_masks'] = np.bool_(getattr(p, 'SUM_ALL_MASKS_PYTHON', False))
        if name in ('bench_GMSK', 'CC11xx_sps16'):
            key = 'bench' if name == 'bench_GMSK' else 'CC11xx'
            out[f'g3/{key}/mask'] = np.asarray(p.get_mask())
            out[f'g3/{key}/syncFlag'] = np.asarray(p.get_syncFlag())
            out[f'g3/{key}/numOnesHeader'] = np.float64(p.numOnesHeader)
            out[f'g3/{key}/numOnesSyncSig'] = np.float64(p.numOnesSyncSig)
            out[f'g3/{key}/headerTol'] = np.int64(p.headerTol)
            out[f'g3/{key}/syncSigTol'] = np.int64(p.syncSigTol)
            out[f'g3/{key}/numBitsOverlap'] = np.int64(p.numBitsOverlap)
            out[f'g3/{key}/packetLen'] = np.int64(p.packetLen)
    return cases


def g4_decoder(out, cases):
    """KATs for Decoder.findFrames: planted headers with k flipped bits, stream cut in 3 calls."""
    import decoder as refdec
    protos = {'bench': cases[0][1], 'CC11xx': cases[4][1]}
    for key, p in protos.items():
        tmpl = ((np.flipud(np.asarray(p.get_mask())) + 1) // 2).astype(np.int64)  # header bits
        rs = np.random.RandomState(7 if key == 'bench' else 8)
        plen = int(p.packetLen)
        L = 3 * plen + 9000
        stream = rs.randint(0, 2, L).astype(np.float64)
        offsets = [700, 700 + plen + 3100, 700 + 2 * plen + 6000]
        flips = [0, int(p.headerTol), int(p.headerTol) + 1]
        payloads = []
        for off, k in zip(offsets, flips):
            pkt = rs.randint(0, 2, plen).astype(np.float64)
            pkt[:len(tmpl)] = tmpl
            if key == 'CC11xx':   # length byte (index 8), de-whitened with PN9[0]=0xff -> len 20
                lb = 20 ^ 0xff
                pkt[64:72] = [(lb >> (7 - i)) & 1 for i in range(8)]
            fl = rs.choice(len(tmpl), size=k, replace=False)
            pkt[fl] = 1 - pkt[fl]
            stream[off:off + plen] = pkt
            payloads.append(pkt)
        cuts = [0, offsets[1] + 300, offsets[2] + plen - 40, L]   # 2nd & 3rd packet straddle calls
        d = refdec.Decoder({}, p)
        out[f'g4/{key}/stream'] = stream.astype(np.uint8)
        out[f'g4/{key}/cuts'] = np.array(cuts)
        out[f'g4/{key}/offsets'] = np.array(offsets)
        out[f'g4/{key}/flips'] = np.array(flips)
        for ci in range(3):
            seg = stream[cuts[ci]:cuts[ci + 1]]
            pk, bits, nsync = d.findFrames(seg, 0)
            out[f'g4/{key}/call{ci}/numSyncSig'] = np.int64(nsync)
            out[f'g4/{key}/call{ci}/npackets'] = np.int64(len(pk))
            out[f'g4/{key}/call{ci}/overlapBuf_after'] = np.asarray(d.bitsOverlapBuf).astype(np.uint8)
            for i, q in enumerate(pk):
                if key == 'bench':
                    out[f'g4/{key}/call{ci}/p{i}/start'] = np.int64(q.frameStartIdx)
                    out[f'g4/{key}/call{ci}/p{i}/maskBitErrors'] = np.float64(q.maskBitErrors)
                out[f'g4/{key}/call{ci}/p{i}/bits'] = np.asarray(q.bits).astype(np.uint8)


def g5_stimulus(out):
    import create_signals as cs
    out['g5/payload_bits'] = cs.packetData().astype(np.uint8)
    for mod in ('GMSK', 'FSK', 'GFSK', 'BPSK'):
        sig, bits = cs.get_padded_packet(mod, 16, 153600)
        sig = np.asarray(sig)
        out[f'g5/{mod}/len'] = np.int64(len(sig))
        out[f'g5/{mod}/dtype'] = np.array(str(sig.dtype))
        out[f'g5/{mod}/head'] = sig[9990:10200].copy()
        out[f'g5/{mod}/tail'] = sig[-10200:-9990].copy()
        out[f'g5/{mod}/stride'] = sig[::397].copy()
        out[f'g5/{mod}/sha256_c64'] = np.array(sha(sig.astype(np.complex64)))
    # awgn semantics (legacy global RNG): seed, then one call
    sig, _ = cs.get_padded_packet('GMSK', 16, 153600)
    np.random.seed(1)
    n = cs.awgn(sig[:4096], 10.0)
    out['g5/awgn/in_head'] = sig[:4096].copy()
    out['g5/awgn/out_seed1_snr10'] = n
    # NRZ-S helper used by the BPSK stimulus
    b = np.random.RandomState(3).randint(0, 2, 64)
    out['g5/nrzs/in'] = b.astype(np.uint8)
    out['g5/nrzs/out'] = cs.encodeNRZS(b)


def g7_hostlogic(out, cases):
    """Pure-numpy methods of the reference's Demodulator, called unbound on a namespace."""
    import demodulator.demodulator_base as db
    D = db.Demodulator
    rs = np.random.RandomState(11)
    N, sps = 1 << 12, 16.0

    def mk_self(ov):
        s = types.SimpleNamespace()
        s.sigOverlapWin = ov // 2
        s.Nfft = N
        s.overlapOffset = 20
        s.symbol_check_error_threshold = 1000
        s.symbol_check_match_threshold = 10
        s.poswinP = []
        return s

    # a continuous symbol stream observed through overlapping blocks, with a +/-1 slip injected
    total = rs.randint(0, 2, 4000).astype(np.float64)
    # 'short': overlap window (8 symbols) shorter than the 20-symbol comparison -> the reference's
    # own try/except swallows a broadcast error and skips the alignment (DB:907-967)
    scen = {'aligned': ([0, 0, 0, 0], 1 << 10), 'early': ([0, 0, 1, 0], 1 << 10),
            'late': ([0, -1, 0, 0], 1 << 10), 'both': ([0, 1, -1, 0], 1 << 10),
            'short': ([0, 0, 1, 0], 1 << 8)}
    for name, (slips, ov) in scen.items():
        s = mk_self(ov)
        out[f'g7/overlap/{name}/ov'] = np.int64(ov)
        step = (N - ov) / sps          # symbols advanced per block
        for b, slip in enumerate(slips):
            first = int(round(b * step)) + slip
            nsym = int(N / sps)
            centres = (np.arange(nsym) * sps + 5).astype(np.int32)
            bits = total[first:first + nsym].copy()
            trust = rs.randint(-128, 127, nsym).astype(np.int8)
            cw, bw, tw, iw = D.checkSymbolOverlap(s, 0, centres, bits.astype(np.int32), bits, trust)
            out[f'g7/overlap/{name}/b{b}/centres'] = centres
            out[f'g7/overlap/{name}/b{b}/bits'] = bits
            out[f'g7/overlap/{name}/b{b}/trust'] = trust
            out[f'g7/overlap/{name}/b{b}/centresWin'] = np.asarray(cw)
            out[f'g7/overlap/{name}/b{b}/bitsWin'] = np.asarray(bw)
            out[f'g7/overlap/{name}/b{b}/trustWin'] = np.asarray(tw)
    # NRZ-S extraction with the BPSK 3-D LUT
    lut = np.asarray(cases[3][1].get_symbolLUT2(5)[1])
    s = types.SimpleNamespace(symbolLUT=lut, bitLUT=None)
    syms = rs.randint(0, 16, 500).astype(np.int32)
    res, err = D.extractBitsNRZs(s, None, syms)
    out['g7/nrzs/symbols'] = syms
    out['g7/nrzs/bits'] = np.asarray(res)
    out['g7/nrzs/symError'] = np.asarray(err, dtype=np.int64)


def g8_pn9(out):
    """PN9 whitening bytes of the CC11xx framing (lib/shift_registers.PN9, pure numpy)."""
    from lib.shift_registers import PN9
    out['g8/pn9_300'] = np.asarray(PN9()).astype(np.int64)


def g9_threshold(out):
    """Demodulator.__thresholdInput (DB:670-707): two clipping rounds, gap filling below 100 samples."""
    import demodulator.demodulator_base as db
    fn = db.Demodulator._Demodulator__thresholdInput
    rs = np.random.RandomState(21)
    N = 1 << 12
    cases = {'none': [], 'one': [(700, 5, 60.0)], 'close': [(500, 4, 80.0), (560, 3, 50.0), (1900, 6, 200.0)],
             'far': [(300, 2, 90.0), (450, 2, 70.0), (3000, 8, 40.0)], 'loud': [(1000, 20, 5000.0), (1050, 1, 30.0)]}
    for name, bursts in cases.items():
        x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
        for pos, ln, amp in bursts:
            x[pos:pos + ln] *= amp
        for scale in (4.5, 40.5):
            s = types.SimpleNamespace(peakThresholdScale=scale, Nfft=N)
            y = x.copy()
            fn(s, y)
            out[f'g9/{name}/s{scale}/in'] = x
            out[f'g9/{name}/s{scale}/out'] = y
            out[f'g9/{name}/s{scale}/clippedPeakIPure'] = np.asarray(s.clippedPeakIPure, dtype=np.int64)
            out[f'g9/{name}/s{scale}/clippedPeakI'] = np.asarray(s.clippedPeakI, dtype=np.int64)


def g10_snr(out):
    """Demodulator.computeSNR (DB:635-667) on a numpy spectrum; bands that wrap around bin 0 included."""
    import demodulator.demodulator_base as db
    rs = np.random.RandomState(22)
    N = 1 << 12
    X = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    X[1000:1040] *= 30
    X[N - 8:] *= 12
    X[:12] *= 12
    shifts = np.array([990, 1010, 1030, N - 6, 4, 2046, 2050, 20], dtype=np.int32)
    s = types.SimpleNamespace(doppCyperSymNorm=shifts, Nfft=N, GPU_bufSignalFreq_cpu_handle=X)
    out['g10/X'] = X
    out['g10/shifts'] = shifts
    pairs = [(0, 1, 5), (1, 2, 5), (3, 4, 5), (5, 6, 5), (0, 0, 3), (7, 7, 5), (4, 7, 2)]
    out['g10/pairs'] = np.array(pairs)
    with np.errstate(all='ignore'):
        out['g10/snr'] = np.array([db.Demodulator.computeSNR(s, lo, hi, w) for lo, hi, w in pairs], dtype=np.float64)


def g11_flags(out, cases):
    """Decoder.findFrames in FLAGS mode (DEC:122-243).  The bench plugin (128-bit header, 16-bit sync flag)
    with packetEndDetectMode switched to FLAGS on the instance."""
    import decoder as refdec
    from protocol.protocolBase import PacketEndDetect
    p = cases[0][1]
    hdr = ((np.flipud(np.asarray(p.get_mask())) + 1) // 2).astype(np.float64)
    flag = ((np.flipud(np.asarray(p.get_syncFlag())) + 1) // 2).astype(np.float64)
    p.packetEndDetectMode = PacketEndDetect.FLAGS
    try:
        scen = {
            # header, two flags behind it (the frame ends at the first, the reference needs a later one too)
            'inside': dict(L=6000, hdr=[900], flags=[2500, 2900], cuts=[0, 6000]),
            # header in call 0 without any flag; flags arrive in call 1
            'across': dict(L=9000, hdr=[3800], flags=[5600, 6100], cuts=[0, 4500, 9000]),
            # header, then silence for more than maxPacketLenBits = 8192 bits: forced end
            'overflow': dict(L=16000, hdr=[700], flags=[], cuts=[0, 3000, 6000, 9000, 12000, 16000]),
            # two frames in one call, the second one too short a tail (< 120 bits to its flag)
            'two': dict(L=9000, hdr=[400, 4000], flags=[1500, 2100, 4090, 6000, 6600], cuts=[0, 9000]),
        }
        for name, sc in scen.items():
            rs = np.random.RandomState(sum(map(ord, name)))
            stream = np.zeros(sc['L'])
            stream[rs.choice(sc['L'], sc['L'] // 10, replace=False)] = 1      # sparse background: no chance matches
            for h in sc['hdr']:
                stream[h:h + len(hdr)] = hdr
            for f in sc['flags']:
                stream[f:f + len(flag)] = flag
            d = refdec.Decoder({}, p)
            out[f'g11/{name}/stream'] = stream.astype(np.uint8)
            out[f'g11/{name}/cuts'] = np.array(sc['cuts'])
            for ci in range(len(sc['cuts']) - 1):
                pk, bits, nsync = d.findFrames(stream[sc['cuts'][ci]:sc['cuts'][ci + 1]], 1000 * ci)
                out[f'g11/{name}/call{ci}/numSyncSig'] = np.int64(nsync)
                out[f'g11/{name}/call{ci}/npackets'] = np.int64(len(pk))
                out[f'g11/{name}/call{ci}/pending'] = np.int64(-1 if d.headerFrameStartIdx is None else d.headerFrameStartIdx)
                for i, q in enumerate(pk):
                    out[f'g11/{name}/call{ci}/p{i}/bits'] = np.asarray(q.bits).astype(np.uint8)
                    out[f'g11/{name}/call{ci}/p{i}/start'] = np.int64(q.frameStartIdx)
                    out[f'g11/{name}/call{ci}/p{i}/maskBitErrors'] = np.float64(q.maskBitErrors)
                    out[f'g11/{name}/call{ci}/p{i}/split'] = np.int64(q.frameSplitIdx)
    finally:
        del p.packetEndDetectMode      # back to the class attribute (FIXED)


def g12_cc11xx_packet(out, cases):
    """PacketCC11xx (CC11xx.py:216-300): length cut, de-whitening, the CRC flag as the reference computes it."""
    from lib.shift_registers import PN9
    p = cases[4][1]
    pn9 = np.asarray(PN9()).astype(np.uint8)
    crcf = sys.modules['crcmod'].mkCrcFun(0x18005, rev=False, initCrc=0xFFFF, xorOut=0)
    rs = np.random.RandomState(31)
    hdr = np.array([0xAA] * 4 + [0xD6, 0xBA, 0xD6, 0xBA], dtype=np.uint8)
    frames = {}
    # (a) the reference TX framer's layout: len = n + 2 counts the CRC
    pay = rs.randint(0, 256, 10).astype(np.uint8)
    body = np.r_[len(pay) + 2, pay].astype(np.uint8)
    c = crcf(body.tobytes())
    body = np.r_[body, c & 0xFF, c >> 8].astype(np.uint8)
    frames['framer'] = np.r_[hdr, body ^ pn9[:len(body)], rs.randint(0, 256, 40).astype(np.uint8)]
    # (b) hardware packet mode: len counts the payload only; CRC over [len | payload] follows, whitened
    pay = rs.randint(0, 256, 17).astype(np.uint8)
    body = np.r_[len(pay), pay].astype(np.uint8)
    c = crcf(body.tobytes())
    body = np.r_[body, c & 0xFF, c >> 8].astype(np.uint8)
    frames['hardware'] = np.r_[hdr, body ^ pn9[:len(body)], rs.randint(0, 256, 40).astype(np.uint8)]
    # (c) the layout for which the reference's own flag reads False: the two bytes behind the data are the
    #     RAW (un-whitened) CRC of [len | data], low byte first
    pay = rs.randint(0, 256, 9).astype(np.uint8)
    body = np.r_[len(pay), pay].astype(np.uint8)
    c = crcf(body.tobytes())
    frames['rawcrc'] = np.r_[hdr, body ^ pn9[:len(body)], c & 0xFF, c >> 8, rs.randint(0, 256, 40).astype(np.uint8)].astype(np.uint8)
    # (d) (c) with one payload bit flipped
    bad = frames['rawcrc'].copy()
    bad[12] ^= 0x10
    frames['rawcrc_bad'] = bad
    for name, fr in frames.items():
        bits = np.unpackbits(fr.astype(np.uint8)).astype(np.float64)
        pk = p.Packet(bits, 5, 1.0)
        data, flag, again = pk.getBinaryData()
        out[f'g12/{name}/bits_in'] = bits.astype(np.uint8)
        out[f'g12/{name}/packetLen'] = np.int64(pk.packetLen)
        out[f'g12/{name}/bits_cut'] = np.asarray(pk.bits).astype(np.uint8)
        out[f'g12/{name}/data'] = np.asarray(data).astype(np.uint8)
        out[f'g12/{name}/flag'] = np.bool_(flag)


def g13_xcorr(out):
    """lib.customXCorr on the streams the reference's own soft-combiner tests use, with the call shape of
    softCombiner.py:701-706 (a zero-padded to the next power of two, b truncated to len(a))."""
    from lib.customXCorr import customXCorr
    z = np.load(os.path.join(REF, 'test', 'test_trustProcessor', 'bitData_test.npz'))
    for tag, (ka, kb, cut) in {'r1_vs_d1': ('ddR1', 'dd1', 58834), 'r2_vs_d2_16k': ('ddR2', 'dd2', 16000),
                               'r1_shifted': ('ddR1', 'ddR1', 20000)}.items():
        a = np.asarray(z[ka], dtype=np.float64)[:cut]
        b = np.asarray(z[kb], dtype=np.float64)
        if tag == 'r1_shifted':
            b = np.asarray(z[kb], dtype=np.float64)[1234:1234 + cut]
        n = len(a)
        nAdd = int(2 ** (np.ceil(np.log2(n))))
        ax = np.r_[a, np.zeros(nAdd - n)]
        r = customXCorr(ax, b[:n])
        out[f'g13/{tag}/a'] = ax.astype(np.uint8)
        out[f'g13/{tag}/b'] = b[:n].astype(np.uint8)
        out[f'g13/{tag}/abs_c64'] = np.abs(r).astype(np.float32)
        top = np.argsort(-np.abs(r), kind='stable')[:15]
        out[f'g13/{tag}/top15_idx'] = top.astype(np.int64)
        out[f'g13/{tag}/head'] = r[:64].astype(np.complex128)


def g14_ringbuffer(out):
    """sigFIFO.RingBuffer (sigFIFO.py:13-103) driven by a seeded trace of inserts and pops: chunk sizes that
    divide nothing, wrap-around of head and tail, pops while too little is buffered, the overflow flush."""
    import sigFIFO as ref
    rs = np.random.RandomState(41)
    for name, (outLen, bufLen, chunks) in {'grc': (15360, 30720, [4095, 4096] * 12), 'bench': (15360, 30720, [16384] * 6),
                                            'odd': (1000, 2500, list(rs.randint(1, 900, 40))),
                                            'tight': (1000, 2000, [900, 900, 900, 300, 700])}.items():
        rb = ref.RingBuffer(outLen, bufLen=bufLen, dtype=np.complex64)
        sizes, popped, state = [], [], []
        v = 0
        for n in chunks:
            n = int(n)
            data = (np.arange(v, v + n) + 1j * (np.arange(v, v + n) % 7)).astype(np.complex64)
            v += n
            sizes.append(rb.insert(data))
            blk = rb.popBlock(outLen)
            popped.append(np.array(blk, dtype=np.complex64).copy() if len(blk) else np.empty(0, np.complex64))
            state.append((rb.headIdx, rb.tailIdx, rb.currentBufSize))
        out[f'g14/{name}/outLen'] = np.int64(outLen)
        out[f'g14/{name}/bufLen'] = np.int64(bufLen)
        out[f'g14/{name}/chunks'] = np.array(chunks, dtype=np.int64)
        out[f'g14/{name}/sizes'] = np.array(sizes, dtype=np.int64)
        out[f'g14/{name}/state'] = np.array(state, dtype=np.int64)
        out[f'g14/{name}/npop'] = np.array([len(q) for q in popped], dtype=np.int64)
        out[f'g14/{name}/popped'] = np.concatenate(popped) if popped else np.empty(0, np.complex64)
    # overflow: the reference logs 'buffer full: Flush', flushes -- and then stores with the end index it computed
    # BEFORE the flush (sigFIFO.py:49-65), which raises ValueError.  Recorded as a fact, not reproduced.
    rb = ref.RingBuffer(1900, bufLen=2000, dtype=np.complex64)
    rb.insert(np.zeros(900, np.complex64))
    rb.insert(np.zeros(900, np.complex64))
    try:
        rb.insert(np.zeros(900, np.complex64))
        raised = False
    except ValueError:
        raised = True
    out['g14/overflow_raises_in_reference'] = np.bool_(raised)


def main():
    if not os.path.isdir(REF):
        raise SystemExit('reference not mounted; fixtures can only be regenerated in the authoring container')
    _install_shims()
    out = {}
    cases = g1_g2_g3(out)
    g4_decoder(out, cases)
    g5_stimulus(out)
    g7_hostlogic(out, cases)
    g8_pn9(out)
    g9_threshold(out)
    g10_snr(out)
    g11_flags(out, cases)
    g12_cc11xx_packet(out, cases)
    g13_xcorr(out)
    g14_ringbuffer(out)
    flat = {k.replace('/', '__'): v for k, v in out.items()}
    path = os.path.join(HERE, 'ref_goldens.npz')
    np.savez_compressed(path, **flat)
    print(f'wrote {path}: {len(flat)} arrays, {os.path.getsize(path)/1e6:.2f} MB')


if __name__ == '__main__':
    main()
