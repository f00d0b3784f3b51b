import sys, numpy as np
sys.path.insert(0,'.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()          # numpy's BLAS workers must not spend the container's CPU quota: a throttled host starves the device
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table
log2N=20; N=1<<log2N; D=256
conf=cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D, samplesPerSym=128)
_,_,shifts,_=doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
M,masks=loadProtocol('CC11xx')(conf=conf).get_filter(N,128,3)
x=sg.s1_stream(1,N,1<<10,'GMSK',snr_db=10.0,seed=1)[:N]
bank=MFBank(log2N,D,M); bank.set_filters(masks); bank.set_shifts(shifts); bank.upload(x)
def settled():
    bank.find_carrier(); prev=None
    for _ in range(12):
        bank.timer_start()
        for _ in range(5): bank.search_async()
        ms=bank.timer_stop()/5
        if prev is not None and ms>0.995*prev: break
        prev=ms
    return ms
for l in (11,12):
    for fpp in (8,4,2,1):
        bank.set_search_path('segment', l, 32, fpp)
        print(f'L=2^{l} filters per pass {fpp}: {settled():.3f} ms', flush=True)
bank.close()
