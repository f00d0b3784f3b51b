"""The C-ABI shared library loads and exports every symbol include/mfbank.h declares (no compute
calls: there is no GPU in the authoring container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib_path():
    import __graft_entry__
    return __graft_entry__.build()


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'mfbank.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mfb_[a-z_0-9]+)\s*\(', text)))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ('mfb_create', 'mfb_destroy', 'mfb_set_filters', 'mfb_set_shifts', 'mfb_input_buffer', 'mfb_upload',
                 'mfb_find_carrier', 'mfb_demodulate', 'mfb_find_centres', 'mfb_sync_correlate', 'mfb_get_scores'):
        assert must in names


def test_header_is_self_contained_c_and_cpp():
    """include/mfbank.h alone, as C99 and as C++17 (a binding generator or a cgo preamble includes nothing before it)."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, 'include', 'mfbank.h')
    for cc, lang, std in (('gcc', 'c', '-std=gnu99'), ('g++', 'c++', '-std=c++17')):
        if shutil.which(cc) is None:
            pytest.skip(f'no {cc}')
        r = subprocess.run([cc, '-fsyntax-only', '-x', lang, std, '-Wall', '-Wextra', '-Werror', hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_every_declaration_names_what_it_replaces():
    """Every entry point's comment cites the reference interface it replaces (file:line), or says that there is none."""
    text = open(os.path.join(ROOT, 'include', 'mfbank.h')).read()
    cite = re.compile(r'(?:DB|CU|DP|DEC|CUFFT|\w+\.py|cuda_kernels\.cu):\s*\d+|[Nn]o reference counterpart|reference has no')
    last, seen, missing = '', set(), []
    # walk the header: a declaration belongs to the last comment block before it (declarations may share one)
    for m in re.finditer(r'(/\*.*?\*/)|\b(mfb_[a-z_0-9]+)\s*\(', text, re.S):
        if m.group(1):
            last = m.group(1)
        elif m.group(2) not in seen:
            seen.add(m.group(2))
            if not cite.search(last):
                missing.append(m.group(2))
    assert not missing, f'no citation in the comment above: {missing}'
    assert seen >= set(_declared_functions())


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared_functions():
        assert hasattr(lib, name), f'{name} declared in include/mfbank.h but not exported'


def test_python_binding_covers_the_header(lib_path):
    from pycusdr_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _declared_functions()
    lib = _lib.load()
    assert lib.mfb_abi_version() == 9
    assert lib.mfb_strerror(0) == b'ok' and b'argument' in lib.mfb_strerror(1)


def test_status_codes_map_to_python_exceptions(lib_path):
    from pycusdr_amd import _lib
    _lib.load()
    with pytest.raises(ValueError):
        _lib.check(_lib.MFB_ERR_ARG, 'x')
    with pytest.raises(TypeError):
        _lib.check(_lib.MFB_ERR_DTYPE, 'x')
    with pytest.raises(MemoryError):
        _lib.check(_lib.MFB_ERR_ALLOC, 'x')
    with pytest.raises(RuntimeError):
        _lib.check(_lib.MFB_ERR_HIP, 'x')
    with pytest.raises(ValueError):
        _lib.check(_lib.MFB_ERR_UNSUPPORTED, 'x')


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pycusdr_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.MFBankLibraryError):
        _lib.load()


def test_product_path_never_imports_the_oracle():
    """pycusdr_amd/ must not reference oracle/ (the oracle is the checker, never the product)."""
    pkg = os.path.join(ROOT, 'pycusdr_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert '/root/reference' not in src, f


def test_bank_rank_analysis_host_only():
    """mfb_analyze_rank (host-only): the shipped banks span far fewer dimensions than they have filters -- the
    basis of the opt-in span search."""
    import numpy as np
    from pycusdr_amd import config as cfg
    from pycusdr_amd.mfbank import analyze_rank
    from pycusdr_amd.protocol import loadProtocol
    expect = {'bench_GMSK': (8, 6), 'bench_FSK': (8, 4), 'bench_GFSK': (8, 4), 'bench_BPSK': (32, 5), 'CC11xx': (8, 4)}
    for name, (M0, r0) in expect.items():
        conf = cfg.cc11xx_config(blockSize=14) if name == 'CC11xx' else cfg.bench_config(name, blockSize=14)
        sps, ms = (128, 3) if name == 'CC11xx' else (16, 5 if name == 'bench_BPSK' else 3)
        M, masks = loadProtocol(name)(conf=conf).get_filter(1 << 14, sps, ms)
        assert (M, analyze_rank(masks)) == (M0, r0), name
    rs = np.random.RandomState(0)
    assert analyze_rank((rs.standard_normal((3, 1024)) + 1j * rs.standard_normal((3, 1024))).astype(np.complex64)) == 3
