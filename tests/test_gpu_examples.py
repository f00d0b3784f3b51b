"""examples/benchmark/live_latency.py runs: a paced producer thread through ``drain_marked`` -- a slow source gets its blocks out one by
one, long before the next block is complete (loose bounds: this is a demonstration, the exact figures are in profiles/r05_chain.md)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_live_latency_example_runs_and_a_slow_source_is_not_kept_waiting():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'benchmark', 'live_latency.py'), '15', '48'], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if '|' in ln and '(48 blocks)' in ln]
    assert len(rows) == 6, r.stdout
    pace, per_call, lat, period = [c.strip() for c in rows[0].split('|')]
    assert pace == '2' and float(per_call) <= 2.0
    median = float(lat.split('/')[0])
    assert median < 0.5 * float(period.split()[0]), rows[0]          # out well before the next block is complete (15.9 ms apart)
