"""bench.py end to end on the GPU at a small geometry: one JSON line with the contract's keys, parity gate included."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('extra', [[], ['--path', 'twopass']])
def test_bench_runs_and_prints_one_json_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2', '--log2n', '16', '--bins', '32'] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['value'] > 0
    assert abs(d['value'] - (2 ** 16 - 2 ** 10) / (d['ms_per_step'] * 1e-3) / 1e6) / d['value'] < 1e-3
    assert d['roofline']['bound'] == ('hbm' if extra else 'valu_fp32') and 0 < d['roofline']['frac'] < 1
    assert d['cpu_baseline']['max_rel_diff_vs_gpu'] < 1e-5 and d['cpu_baseline']['cores'] >= 1
    assert 'energy_search' in d['config'] and d['config']['energy_search']['max_rel_diff_vs_default_search'] < 1e-5
