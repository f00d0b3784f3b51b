"""Shared by tests/test_gpu_configs.py and tests/children/c5_child.py: the two demodulator instances of
BASELINE config C5 (CC11xx FSK-2 at 128 samples/symbol, M=8; custom BPSK filter set, M=32), each with a
matching stimulus, and the checks run on an instance's bank.  Test infrastructure."""
import numpy as np

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table
from pycusdr_amd.protocol import loadProtocol


def c5_instance(name, log2N, D):
    N = 1 << log2N
    if name == 'CC11xx':
        conf = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D)
        sps, ms = 128, 3
        from pycusdr_amd.protocol.CC11xx import frame_bits
        rs = np.random.RandomState(4)
        fs = 7416 * sps
        bits = np.concatenate([frame_bits(rs.randint(0, 256, 200).astype(np.uint8), preamble=(0xAA,) * 10) for _ in range(1 + N // (sps * 1800))])
        sig = sg.modulateFSK(bits, sps)[:N]
        sig = sig * np.exp(1j * 2 * np.pi * 148320 / fs * np.arange(len(sig)))       # config/CC11xx.json IF offset
        x = sg.awgn(sig, 15.0, rng=np.random.RandomState(2)).astype(np.complex64)
        expect = int(round(148320 / fs * N))
    else:
        conf = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D)
        sps, ms = 16, 5
        x = sg.s1_stream(1, N, 1 << 10, 'BPSK', snr_db=12.0, seed=3)[:N]
        expect = N // 4
    proto = loadProtocol(name)(conf=conf)
    M, masks = proto.get_filter(N, sps, ms)
    _, hz, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
    return dict(name=name, N=N, D=D, M=M, masks=masks, shifts=shifts, x=x, expect=expect, conf=conf, proto=proto)


def check_instance(bank, inst, oracle_bins=6):
    """Parseval on all bins, real oracle IFFTs on a few, exact pick, carrier where the stimulus puts it."""
    N, D, masks, shifts = inst['N'], inst['D'], inst['masks'], inst['shifts']
    bank.upload(inst['x'])
    idx, metric = bank.find_carrier()
    ds = bank.get_scores()
    X = bank.get_spectrum()
    pv = orc.doppler_scores_parseval(X, masks, shifts)
    parseval = float(np.abs(ds[:, 0] - pv).max() / pv.max())
    sel = np.unique(np.r_[0, D - 1, int(float(idx)), np.linspace(1, D - 2, max(oracle_bins - 3, 1)).astype(int)])[:oracle_bins]
    ref = orc.doppler_scores(X, masks, shifts[sel], True)[:, 0]
    real = float(np.abs(ds[sel, 0] - ref).max() / ref.max())
    oidx, ometric = orc.find_doppler_est(ds, D, 0, True)
    pick = orc.interpolate_doppler(idx, shifts, np.zeros(len(shifts)))
    sh = np.where(shifts > N // 2, shifts - N, shifts).astype(np.int64)
    spacing = float(np.median(np.diff(np.sort(sh))))
    found = pick['dopplerIdxlast'] if pick['dopplerIdxlast'] <= N // 2 else pick['dopplerIdxlast'] - N
    return dict(parseval=parseval, real=real, pick_exact=bool(idx == oidx), idx=float(idx),
                carrier_err_bins=float(abs(found - inst['expect']) / spacing), zeros_ok=bool(np.all(ds[:, 1:] == 0)))
