import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.demodulator import UHF
from pycusdr_amd.protocol import loadProtocol
import pycusdr_amd.demodulator.demodulator_base as dbm
from oracle_bank import OracleBank
from oracle import mfbank_oracle as orc
bs, ov, D = 15, 1<<10, 32
N = 1<<bs
conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D)
p = loadProtocol('bench_GMSK')(conf=conf)
gpu = UHF.Demodulator(conf, p, 'UHF-H')
real = dbm.MFBank
dbm.MFBank = OracleBank
cpu = UHF.Demodulator(conf, p, 'UHF-H')
dbm.MFBank = real
sig,_ = sg.get_padded_packet('GMSK',16,153600)
sig = np.concatenate((sig,np.zeros(N)))
sig = sg.awgn(sig, 60.0, rng=np.random.RandomState(1)).astype(np.complex64)
rg, rc = gpu.get_signalBufferHostPointer(), cpu.get_signalBufferHostPointer()
rg[:] = sig[:N]; rc[:] = sig[:N]
print(gpu.uploadAndFindCarrier(rg), cpu.uploadAndFindCarrier(rc))
sg_, sc_ = gpu.findCodeRateAndPhaseGPU(), cpu.findCodeRateAndPhaseGPU()
print('gpu', sg_, gpu._codeRateResult, 'cpu', sc_, cpu._codeRateResult)
xg, xc = gpu.bank.get_xcorr(), cpu.bank.get_xcorr()
print('xc rel err', np.abs(xg-xc).max()/np.abs(xc).max())
a = gpu.cudaFindCentres(*sg_); b = cpu.cudaFindCentres(*sc_)
d = np.where(a[0]!=b[0])[0]; print('sym diffs', len(d), d[:20], a[0][d[:10]], b[0][d[:10]])
d2 = np.where(a[2]!=b[2])[0]; print('centre diffs', len(d2), d2[:20], a[2][d2[:10]], b[2][d2[:10]])
# oracle with gpu's params on gpu xc
o = orc.find_centres(xg, sg_[0], sg_[1], 7, 0)
print('gpu vs oracle(gpu inputs): sym', np.array_equal(a[0], o[0]), 'cen', np.array_equal(a[2], o[1]))
mags = gpu.magnitudes
print('mag at diffs', mags[d[:10]], 'median mag', np.median(mags))
