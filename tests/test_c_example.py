"""examples/c/receive_block.c: a host program that uses libmfbank.so through include/mfbank.h alone (no Python, no
torch).  CPU: the header is valid C and the program links against every symbol it uses.  GPU: it runs one block
through search, pick, demodulation and symbol decisions and checks carrier, symbol rate and bits itself."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'examples', 'c', 'receive_block.c')
SRC_BATCH = os.path.join(ROOT, 'examples', 'c', 'receive_blocks.c')
LIBDIR = os.path.join(ROOT, 'pycusdr_amd')


def _compile(out, SRC=SRC):
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    if not os.path.exists(os.path.join(LIBDIR, 'libmfbank.so')):
        import __graft_entry__
        __graft_entry__.build()
    cmd = ['gcc', '-O2', '-std=gnu99', '-Wall', '-Wextra', '-Werror', SRC, '-I', os.path.join(ROOT, 'include'), '-L', LIBDIR,
           '-lmfbank', '-lm', f'-Wl,-rpath,{LIBDIR}', '-o', str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_c_host_program_compiles_and_links(tmp_path):
    _compile(tmp_path / 'receive_block')


@pytest.mark.gpu
def test_c_host_program_receives_a_block(tmp_path):
    exe = _compile(tmp_path / 'receive_block')
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith('ok') and '0 errors' in r.stdout and 'identical to the stage-by-stage calls' in r.stdout


def test_c_host_program_for_batches_compiles_and_links(tmp_path):
    _compile(tmp_path / 'receive_blocks', SRC_BATCH)


@pytest.mark.gpu
def test_c_host_program_receives_a_batch_of_blocks(tmp_path):
    """examples/c/receive_blocks.c: four consecutive blocks of one window per device call, the bit lookup and the block-overlap
    alignment on the device, from plain C; every block equals the one-block call, the handed-over bit stream equals the bits sent."""
    exe = _compile(tmp_path / 'receive_blocks', SRC_BATCH)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith('ok') and '4 blocks aligned on the device' in r.stdout and ' 0 differ' in r.stdout
    assert 'DIFFERENT' not in r.stdout
