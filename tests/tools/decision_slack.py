"""How much of the end-to-end decision tolerance is really needed: HIP Demodulator vs the oracle-backed one on
the same blocks (fp32 device transforms vs fp64 oracle transforms): differing symbol decisions and moved
symbol centres per block, for every modulation of the BER bench."""
import sys
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.demodulator import UHF
import pycusdr_amd.demodulator.demodulator_base as db
from oracle_bank import OracleBank

for mod, pname, snr in [('GMSK', 'bench_GMSK', 10.0), ('GFSK', 'bench_GFSK', 12.0), ('FSK', 'bench_FSK', 12.0), ('BPSK', 'bench_BPSK', 12.0),
                        ('GMSK', 'bench_GMSK', 6.0)]:
    bs, ov, D = 15, 1 << 10, 32
    N = 1 << bs
    conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol(pname)(conf=conf)
    gpu = UHF.Demodulator(conf, p, 'UHF-H')
    real = db.MFBank
    db.MFBank = OracleBank
    cpu = UHF.Demodulator(conf, p, 'UHF-H')
    db.MFBank = real
    for dm in (gpu, cpu):
        def tapped(*a, _orig=dm.checkSymbolOverlap, _dm=dm, **k):
            out = _orig(*a, **k)
            _dm._centresWin = np.asarray(out[0])
            return out
        dm.checkSymbolOverlap = tapped
    sig, payload = sg.get_padded_packet(mod, 16, 153600)
    sig = sg.awgn(np.concatenate((sig, np.zeros(N))), snr, rng=np.random.RandomState(1)).astype(np.complex64)
    nblocks = (len(sig) - ov) // (N - ov)
    rg, rc = gpu.get_signalBufferHostPointer(), cpu.get_signalBufferHostPointer()
    rg[:ov] = rc[:ov] = sig[:ov]
    tot = nsym = ncen = mx = 0
    for b in range(nblocks):
        rg[ov:] = rc[ov:] = sig[ov + b * (N - ov): ov + (b + 1) * (N - ov)]
        gpu.uploadAndFindCarrier(rg)
        cpu.uploadAndFindCarrier(rc)
        bg = gpu.demodulate()[0]
        bc = cpu.demodulate()[0]
        if len(bg) == len(bc):
            nsym += int(np.count_nonzero(bg != bc))
            d = gpu._centresWin - cpu._centresWin
            ncen += int(np.count_nonzero(d))
            mx = max(mx, int(np.abs(d).max()))
            tot += len(bg)
        else:
            print('  length differs', len(bg), len(bc))
        rg[:ov] = rg[-ov:]
        rc[:ov] = rc[-ov:]
    print(f'{mod} snr {snr}: path {gpu.bank.get_search_path()["path"]}, {tot} symbols: {nsym} decisions differ, {ncen} centres moved (max {mx} samples)', flush=True)
    gpu.close()
