"""Child process of the C5 test: ONE demodulator instance of the multi-protocol configuration (its own
process and device context, reference demodulator_base.py:177-181) running next to another on the same
device.  Prints one JSON line.  usage: c5_child.py <CC11xx|bench_BPSK> <log2N> <D> <blocks>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from c5_common import c5_instance, check_instance   # noqa: E402

name, log2N, D, blocks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
inst = c5_instance(name, log2N, D)
from pycusdr_amd.mfbank import MFBank   # noqa: E402
bank = MFBank(log2N, D, inst['M'])
bank.set_filters(inst['masks'])
bank.set_shifts(inst['shifts'])
picks = []
for b in range(blocks):                  # keep the device busy while the sibling process runs
    bank.upload(inst['x'])
    picks.append(float(bank.find_carrier()[0]))
res = check_instance(bank, inst, oracle_bins=3)
res['picks_equal'] = len(set(picks)) == 1
res['path'] = bank.get_search_path()
bank.close()
print(json.dumps(res))
