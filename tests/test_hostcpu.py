"""pycusdr_amd/hostcpu.py: the CPU share the entry scripts report and the BLAS pool limit they apply (the stall it
prevents is measured on the GPU box, profiles/r03_ber.md)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_share_is_bounded_by_affinity_and_host():
    from pycusdr_amd.hostcpu import cpu_share
    n = cpu_share()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, 'sched_getaffinity'):
        assert n <= len(os.sched_getaffinity(0))


def test_quiet_blas_limits_loaded_pools_and_children():
    code = ('import numpy as np, json, os\n'
            'from pycusdr_amd.hostcpu import quiet_blas\n'
            'n = quiet_blas()\n'
            'from threadpoolctl import threadpool_info\n'
            'np.linalg.norm(np.ones(1 << 16))\n'
            'print(json.dumps(dict(changed=n, threads=[p["num_threads"] for p in threadpool_info() if p["user_api"] == "blas"],'
            ' env=os.environ["OPENBLAS_NUM_THREADS"])))\n')
    env = {k: v for k, v in os.environ.items() if k not in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS')}
    env['PYTHONPATH'] = ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    import json
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['env'] == '1'
    assert d['changed'] >= 1 and all(t == 1 for t in d['threads'])


def test_quiet_blas_keeps_an_explicit_setting():
    code = ('import os\nfrom pycusdr_amd.hostcpu import quiet_blas\nquiet_blas()\nprint(os.environ["OPENBLAS_NUM_THREADS"])\n')
    env = dict(os.environ, PYTHONPATH=ROOT, OPENBLAS_NUM_THREADS='3')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1] == '3'
