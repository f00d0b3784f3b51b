"""ctypes binding of libmfbank.so (the C ABI declared in include/mfbank.h).

The product path has no CPU fallback: if the shared library is missing or cannot be loaded, every
entry point raises ``MFBankLibraryError``.  Status codes are turned into Python exceptions the way
the reference maps cuFFT statuses to exceptions (reference lib/cufft.py:90-116).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFBANK_LIB lets a developer point at an alternative build of the same ABI (kernel experiments)
LIB_PATH = os.environ.get('MFBANK_LIB') or os.path.join(_HERE, 'libmfbank.so')

MFB_OK, MFB_ERR_ARG, MFB_ERR_DTYPE, MFB_ERR_ALLOC, MFB_ERR_HIP, MFB_ERR_STATE, MFB_ERR_UNSUPPORTED = range(7)


class MFBankLibraryError(ImportError):
    """libmfbank.so is not built / not loadable; the HIP path is mandatory."""


class MFBankError(RuntimeError):
    """HIP runtime / state error reported by libmfbank (MFB_ERR_HIP, MFB_ERR_STATE)."""


_EXC = {
    MFB_ERR_ARG: ValueError,
    MFB_ERR_DTYPE: TypeError,
    MFB_ERR_ALLOC: MemoryError,
    MFB_ERR_HIP: MFBankError,
    MFB_ERR_STATE: MFBankError,
    MFB_ERR_UNSUPPORTED: ValueError,
}

_vp, _i, _fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class BlockParams(C.Structure):
    """mfb_block_params of include/mfbank.h."""
    _fields_ = [('mode', C.c_int32), ('input', C.c_int32), ('device_block', C.c_void_p), ('fixed_shift', C.c_int32),
                ('k_offset', C.c_int32), ('k_len', C.c_int32), ('spsym_min', C.c_int32), ('op', C.c_int32),
                ('snr_window', C.c_int32), ('max_symbols', C.c_int32), ('band_capacity', C.c_int32), ('block_stride', C.c_int32)]


class BlockResult(C.Structure):
    """mfb_block_result of include/mfbank.h."""
    _fields_ = [('pick', C.c_float * 2), ('pick_valid', C.c_int32), ('shift', C.c_int32), ('low', C.c_int32), ('high', C.c_int32),
                ('frac', C.c_double), ('cr', C.c_float * 3), ('spSym', C.c_double), ('codeOffset', C.c_double),
                ('count', C.c_int32), ('rate_fallback', C.c_int32), ('band_len', C.c_int32 * 2)]


class StreamParams(C.Structure):
    """mfb_stream_params of include/mfbank.h."""
    _fields_ = [('overlap_samples', C.c_int32), ('overlap_offset', C.c_int32), ('match_threshold', C.c_int32),
                ('error_threshold', C.c_int32), ('lut_mode', C.c_int32), ('lut_rows', C.c_int32), ('lut_successors', C.c_int32),
                ('lut', C.c_void_p), ('num_templates', C.c_int32), ('bits_overlap', C.c_int32), ('template_taps', C.c_int32 * 2),
                ('template_thresholds', C.c_int32 * 2), ('templates', C.c_void_p)]


class RecordLayout(C.Structure):
    """mfb_record_layout of include/mfbank.h."""
    _fields_ = [(k, C.c_int32) for k in ('nblocks', 'scalars_bytes', 'symbols', 'band_capacity', 'mode', 'fixed_shift', 'stream_stages',
                                         'max_hits', 'templates', 'reserved', 'edge_candidates', 'edge_hits')] + \
               [(k, C.c_int64) for k in ('record_bytes', 'off_bands', 'off_sym', 'off_cen', 'off_mag', 'off_bits', 'off_centres_u8',
                                         'off_trust', 'off_post', 'off_end', 'off_hits', 'off_edges')]


# name -> (restype, argtypes); exactly the prototypes of include/mfbank.h
PROTOTYPES = {
    'mfb_strerror': (C.c_char_p, [_i]),
    'mfb_abi_version': (_i, []),
    'mfb_create': (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i, _i]),
    'mfb_destroy': (_i, [_vp]),
    'mfb_set_stream': (_i, [_vp, _vp]),
    'mfb_set_tuning': (_i, [_vp, _i, _i, _i, _i]),
    'mfb_get_tuning': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'mfb_get_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'mfb_set_search_path': (_i, [_vp, _i, _i, _i, _i]),
    'mfb_get_search_path': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'mfb_set_search_basis': (_i, [_vp, _i]),
    'mfb_get_search_basis': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'mfb_set_search_mode': (_i, [_vp, _i]),
    'mfb_get_search_mode': (_i, [_vp, C.POINTER(_i)]),
    'mfb_analyze_rank': (_i, [_vp, _i, _i, C.POINTER(_i)]),
    'mfb_debug_fail_alloc': (_i, [_i]),
    'mfb_analyze_filters': (_i, [_vp, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    'mfb_set_filters': (_i, [_vp, _vp, _i, _i]),
    'mfb_set_shifts': (_i, [_vp, _vp, _i]),
    'mfb_input_buffer': (_i, [_vp, C.POINTER(_fp)]),
    'mfb_upload': (_i, [_vp]),
    'mfb_upload_from': (_i, [_vp, _vp, _i]),
    'mfb_upload_device': (_i, [_vp, _vp]),
    'mfb_search_async': (_i, [_vp]),
    'mfb_export_scores_async': (_i, [_vp, _vp, _i]),
    'mfb_pick': (_i, [_vp, _vp, _i, _i, _fp]),
    'mfb_export_column_async': (_i, [_vp, _vp, _i]),
    'mfb_export_rows_async': (_i, [_vp, _vp, _i, _i, _i, _i]),
    'mfb_receive_block': (_i, [_vp, C.POINTER(BlockParams), C.POINTER(BlockResult), _vp, _vp, _vp, _vp]),
    'mfb_receive_block_begin': (_i, [_vp, C.POINTER(BlockParams), _i]),
    'mfb_receive_block_end': (_i, [_vp, _i, C.POINTER(BlockResult), _vp, _vp, _vp, _vp]),
    'mfb_input_buffer2': (_i, [_vp, C.POINTER(_fp)]),
    'mfb_window_buffer': (_i, [_vp, _i, _i, _i, C.POINTER(_fp)]),
    'mfb_receive_blocks_begin': (_i, [_vp, C.POINTER(BlockParams), _i, _i]),
    'mfb_set_batch_overlap': (_i, [_vp, _i]),
    'mfb_get_batch_scores': (_i, [_vp, _i, _vp]),
    'mfb_get_search_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'mfb_set_cu_share': (_i, [_vp, _i, _i]),
    'mfb_receive_blocks_end': (_i, [_vp, _i, C.POINTER(BlockResult), _vp, _vp, _vp, _i, _vp]),
    'mfb_receive_blocks_end_record': (_i, [_vp, _i, _vp, C.c_size_t, C.POINTER(RecordLayout)]),
    'mfb_debug_stream_stages': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.POINTER(RecordLayout)]),
    'mfb_set_stream_stages': (_i, [_vp, C.POINTER(StreamParams)]),
    'mfb_stream_seed': (_i, [_vp, _vp, _i, _vp, _i, _vp, _i]),
    'mfb_debug_block_scalars': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, C.POINTER(BlockResult), _vp, _vp]),
    'mfb_pick_column': (_i, [_vp, _vp, _i, _i, _fp]),
    'mfb_find_carrier': (_i, [_vp, _fp]),
    'mfb_get_scores': (_i, [_vp, _vp]),
    'mfb_get_spectrum': (_i, [_vp, _vp, _i, _i]),
    'mfb_demodulate': (_i, [_vp, _i, _i, _i, _fp]),
    'mfb_find_centres': (_i, [_vp, C.c_float, C.c_float, _i, _i, _vp, _vp, _vp]),
    'mfb_get_xcorr': (_i, [_vp, _vp]),
    'mfb_get_envelope': (_i, [_vp, _vp]),
    'mfb_sync_correlate': (_i, [_i, _vp, _i, _i, _vp, _i, _vp]),
    'mfb_sync_find': (_i, [_i, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'mfb_sync_find_multi': (_i, [_i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'mfb_syncfinder_create': (_i, [C.POINTER(_vp), _i, _vp, _vp, _vp, _i, _i, _i]),
    'mfb_syncfinder_destroy': (_i, [_vp]),
    'mfb_syncfinder_begin': (_i, [_vp, _vp, _i]),
    'mfb_syncfinder_end': (_i, [_vp, _vp, _vp, _vp]),
    'mfb_sync_find_packed': (_i, [_i, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    'mfb_sync_pinned_buffer': (_i, [_i, C.c_size_t, C.POINTER(_vp)]),
    'mfb_xcorr': (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    'mfb_timer_start': (_i, [_vp]),
    'mfb_timer_stop': (_i, [_vp, _fp]),
    'mfb_profile_enable': (_i, [_vp, _i]),
    'mfb_profile_read': (_i, [_vp, C.POINTER(_i), _fp]),
    'mfb_sync': (_i, [_vp]),
    'mfb_hostcopy_create': (_i, [C.POINTER(_vp)]),
    'mfb_hostcopy_submit': (_i, [_vp, _vp, _vp, C.c_size_t]),
    'mfb_hostcopy_drain': (_i, [_vp]),
    'mfb_hostcopy_destroy': (None, [_vp]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64/libhsa-runtime64 and
    loads them by full path; libmfbank.so asks for libamdhip64.so.7 by name.  If libmfbank came
    first it would pull in /opt/rocm's copy, and a later `import torch` would bring a SECOND runtime
    into the process, which then finds no GPU ("No HIP GPUs are available").  Loading torch's copy
    (without importing torch) before libmfbank makes both use the same one, in either order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    for d in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        path = os.path.join(d, 'lib', 'libamdhip64.so')
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return path
    return None


def load():
    """Load libmfbank.so once; raise MFBankLibraryError loudly if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MFBankLibraryError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback for the matched-filter bank.')
    try:
        _preload_hip_runtime()
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise MFBankLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MFBankLibraryError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what=''):
    if status == MFB_OK:
        return
    msg = load().mfb_strerror(status).decode()
    raise _EXC.get(status, MFBankError)(f'libmfbank {what}: {msg} (status {status})')
