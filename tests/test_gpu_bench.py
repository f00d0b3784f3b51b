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


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [[], ['--single-comm'], ['--single-comm', '--no-prefetch']])
def test_bench_sharded_path_on_one_rank(mode):
    """`bench.py --gpus 1 --force-dist`: the whole sharded step (block hand-over, search, RCCL exchange on a one-rank
    communicator, pick on the gathered table) in both broadcast modes; the line names the communicator's size and the
    device behind every rank."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '4', '--warmup', '2',
           '--log2n', '16', '--bins', '32', '--repeats', '2', '--watchdog', '60'] + mode
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    c = d['config']
    assert d['n_gpus'] == 1 and c['rccl_world'] == 1 and c['backend'].startswith('nccl') and c['carrier_found'] is True
    assert len(c['rank_devices']) == 1 and c['rank_devices'][0]['rank'] == 0 and c['distinct_devices'] == 1
    assert ('single communicator' in c['broadcast_mode']) == ('--single-comm' in mode)
    assert ('no prefetch' in c['broadcast_mode']) == ('--no-prefetch' in mode)
    assert d['value'] > 0 and c['repeats'] == 2
