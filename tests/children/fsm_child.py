"""Child of tests/test_gpu_segment.py::test_the_shift_on_the_filters_side_equals_the_shift_on_the_samples: the doppSum table of one seeded
block in THIS process's form of the segment search (MFB_SEG_FSM in the environment is read once per process).  Writes an .npz.
usage: fsm_child.py <protocol> <log2N> <D> <noise bins> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pycusdr_amd import config as cfg, signals as sg                                  # noqa: E402
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table                 # noqa: E402
from pycusdr_amd.mfbank import MFBank                                                  # noqa: E402
from pycusdr_amd.protocol import loadProtocol                                          # noqa: E402

name, log2N, D, Doff, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
N = 1 << log2N
if name == 'CC11xx':
    conf, sps, ms = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D), 128, 3
else:
    conf, sps, ms = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D), 16, (5 if name == 'bench_BPSK' else 3)
_, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
if Doff:                                        # noise-reference rows in front of the table (DB:148-159)
    shifts = np.concatenate(((shifts[:Doff] + N // 3) % N, shifts)).astype(np.int32)
M, masks = loadProtocol(name)(conf=conf).get_filter(N, sps, ms)
x = sg.s1_stream(1, N, 1 << 10, 'GMSK', snr_db=8.0, seed=41)[:N]
bank = MFBank(log2N, D, M, doppler_offset=Doff)
bank.set_filters(masks)
bank.set_shifts(shifts)
bank.upload(x)
pick = bank.find_carrier()
np.savez(out, scores=bank.get_scores(), pick=np.asarray(pick, dtype=np.float64), filter_side=int(bank.get_search_info()['filter_side']),
         bins_per_forward=bank.get_search_info()['bins_per_forward'], log2L=bank.get_search_path()['log2L'])
bank.close()
