import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def goldens():
    import numpy as np
    path = os.path.join(ROOT, 'tests', 'golden', 'ref_goldens.npz')
    z = np.load(path, allow_pickle=False)
    return {k.replace('__', '/'): z[k] for k in z.files}
