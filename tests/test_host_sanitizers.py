"""The host-only half of libmfbank (filter analysis, segment spectra, span basis: pycusdr_amd/csrc/filter_taps.hpp; the copy
worker of the receive loop: pycusdr_amd/csrc/hostcopy.hpp) under AddressSanitizer + UndefinedBehaviorSanitizer and under
ThreadSanitizer (the analysis runs one thread per filter row; the copy worker is a thread with a queue).  The GPU pool cannot run
sanitizers on device code; this is the native code that can be checked this way, on the CPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'csrc', 'taps_sanitize.cpp')
SRC_COPY = os.path.join(ROOT, 'tests', 'csrc', 'hostcopy_sanitize.cpp')


@pytest.mark.timeout(600)
@pytest.mark.parametrize('name,flags', [('asan_ubsan', ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined']),
                                        ('tsan', ['-fsanitize=thread'])])
def test_filter_analysis_under_sanitizers(tmp_path, name, flags):
    _run_under(tmp_path, name, flags, SRC, 'banks ok')


@pytest.mark.timeout(600)
@pytest.mark.parametrize('name,flags', [('asan_ubsan', ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined']),
                                        ('tsan', ['-fsanitize=thread'])])
def test_copy_worker_under_sanitizers(tmp_path, name, flags):
    _run_under(tmp_path, name, flags, SRC_COPY, 'copies ok')


def _run_under(tmp_path, name, flags, src, last_line):
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = tmp_path / f'{os.path.basename(src)[:-4]}_{name}'
    build = subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-pthread', '-Wall', '-Wextra'] + flags + [src, '-o', str(exe)],
                           capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and ('cannot find' in build.stderr or 'unrecognized' in build.stderr):
        pytest.skip(f'sanitizer runtime not installed: {build.stderr[-200:]}')
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1',
               TSAN_OPTIONS='halt_on_error=1')
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=500, env=env)
    if run.returncode != 0 and ('unexpected memory mapping' in run.stderr or 'Shadow memory range interleaves' in run.stderr
                                or 'ReserveShadowMemoryRange failed' in run.stderr):
        pytest.skip('this kernel\'s address-space layout does not admit the sanitizer runtime: ' + run.stderr[-200:])
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert run.stdout.strip().endswith(last_line)
