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
    assert len(rows) == 12, r.stdout               # six paces with a source that marks where it would block, six with a plain iterator
    for first in (0, 6):
        for row, want_pace in ((rows[first], '2'), (rows[first + 1], '20')):
            pace, per_call, lat, period = [c.strip() for c in row.split('|')]
            assert pace == want_pace and float(per_call) <= 2.0, row
            median = float(lat.split('/')[0])
            assert median < 0.5 * float(period.split()[0]), row         # out well before the next block is complete (15.9 / 1.6 ms apart)
