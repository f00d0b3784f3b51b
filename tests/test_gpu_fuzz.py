"""Seeded random sweep of both search paths (and the span basis) against the oracle: the tool's loop, 25 cases."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_geometries_match_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'fuzz_seg.py'), '25', '7'], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '25 random cases ok' in out.stdout


@pytest.mark.gpu
def test_batched_loop_equals_one_block_loop_on_random_cases():
    """tests/tools/fuzz_batches.py, 12 seeded cases: run_stream with B blocks per device call (stream stages on the device or on
    the host) against the one-block loop over random block sizes, bins, modulations, SNRs down to where packets are lost, chunk
    sizes, zero stretches (skipped blocks, irregular blocks, re-seeding of the device's state) and two calls per runner."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'fuzz_batches.py'), '12', '5'], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert 'all equal' in out.stdout and out.stdout.count('\nok ') + out.stdout.startswith('ok ') == 12
