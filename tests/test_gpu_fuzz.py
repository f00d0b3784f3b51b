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
