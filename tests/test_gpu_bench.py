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


def _lead_is_flat(c):
    """The driver keeps the first twenty scalar keys of `config`: they must be `workload` and the flat figures, names <= 40
    characters, strings <= 120."""
    keys = list(c)
    assert keys[0] == 'workload'
    lead = keys[:min(20, len(keys))]
    n_scalar = 0
    for k in keys:
        if isinstance(c[k], (dict, list)):
            break
        n_scalar += 1
    assert n_scalar >= min(8, len(lead)), keys[:12]
    for k in keys[:n_scalar][:20]:
        assert len(k) <= 40 and (not isinstance(c[k], str) or len(c[k]) <= 120), k
    return keys[:n_scalar]


@pytest.mark.gpu
@pytest.mark.parametrize('launcher', [False, True])
def test_bench_two_ranks_fall_back_to_the_conservative_mode(launcher):
    """One-shot safety of the N > 1 bench: the first attempt (default mode: block broadcast on a communicator and stream of its
    own) is made to fail -- the last rank never joins the first collective, the other rank's watchdog names it and exits 3 --,
    fresh ranks are started with --single-comm --no-prefetch, the run ends with status 0 and the line says which mode produced
    it.  Two gloo ranks on one GPU; once started by bench.py itself, once by the launcher the driver uses (then every rank
    process supervises its own worker)."""
    args = ['--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1', '--log2n', '16', '--bins', '32', '--repeats', '2',
            '--watchdog', '8', '--no-cpu-baseline']
    bench = os.path.join(ROOT, 'bench.py')
    if launcher:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
               '--master-port', str(port), bench] + args
    else:
        cmd = [sys.executable, bench] + args
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'BENCH_WORKER'):
        env.pop(k, None)
    env['BENCH_FAIL_FIRST_ATTEMPT'] = '1'
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 2 and d['value'] > 0 and c['world_size'] == 2
    assert c['dist_mode'].startswith('single-comm') and 'no-prefetch' in c['dist_mode']
    assert c['fallback_from'].startswith('default mode') and 'exit' in c['fallback_from']
    assert 'blocks_stream_msamples' not in c          # the fallback attempt runs without the time-chunk-sharded leg
    lead = _lead_is_flat(c)
    for k in ('roofline_frac', 'dist_mode', 'fallback_from', 'stream_msamples'):
        assert k in lead[:20], (k, lead)
    assert 'fresh' in r.stderr          # the supervisor's notice that the default-mode ranks had left with a non-zero status


@pytest.mark.gpu
def test_bench_two_ranks_report_both_sharding_axes():
    """An undisturbed two-rank job (gloo on one GPU): the default mode produces the line, and the same job carries the
    time-chunk-sharded leg (SURVEY 8e's other axis) as flat keys among the first twenty."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '1', '--log2n', '16',
           '--bins', '32', '--repeats', '2', '--watchdog', '60', '--no-cpu-baseline']
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'BENCH_WORKER', 'BENCH_FAIL_FIRST_ATTEMPT'):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    c = d['config']
    assert d['n_gpus'] == 2 and c['dist_mode'].startswith('concurrent-broadcast') and 'fallback_from' not in c
    assert c['blocks_stream_msamples'] > 0 and 0 < c['blocks_efficiency_vs_1gpu'] < 1.5
    lead = _lead_is_flat(c)
    for k in ('roofline_frac', 'dist_mode', 'blocks_stream_msamples', 'blocks_efficiency_vs_1gpu', 'stream_msamples'):
        assert k in lead[:20], (k, lead)
