"""Config dictionaries with the reference's key layout (config/base.json, config/benchmark/*.json,
config/CC11xx.json): builders for the geometries the tests and benches use, and ``load_config`` for a user's
own files in the reference's format (JSON with // comments, layered through "configBase"; the reference
reads them with rjsmin + pyLoadModularJson, pyCuSDR.py:61)."""
import copy
import json
import os

BASE_GPU_UHF = {
    'blockSize': 16, 'overlap': 10, 'bitWindowWidth': 7, 'xcorrMaskSize': 3, 'clippedPeakSpan': 20,
    'symbol_check_overlap_offset': 20, 'symbol_check_error_threshold': 1000,
    'symbol_check_match_num_errors_allowed': 10, 'doppCarrierSteps': 64, 'peakThresholdScale': 40.5,
    'CUDA': {'device': 0, 'numThreads': 64, 'numThreadsS': 1024, 'batchSize': 0, 'streams': 3, 'num_SMS': 0},
}
BASE_GPU_STX = dict({k: v for k, v in BASE_GPU_UHF.items() if k != 'doppCarrierSteps'},      # config/base.json:32-49
                    blockSize=17, overlap=11, bitWindowWidth=3, clippedPeakSpan=40, peakThresholdScale=4.5)


def bench_config(protocol='bench_GMSK', blockSize=15, overlap=10, doppCarrierSteps=64, rangeRateMax=7500,
                 xcorrMaskSize=None, device=0, radio='UHF-H'):
    """config/benchmark/bench_base.json geometry: Fc 437.3 MHz, IF offset 38.4 kHz (= fs/4),
    9600 baud x 16 samples/symbol."""
    gpu = copy.deepcopy(BASE_GPU_UHF)
    gpu.update(blockSize=blockSize, overlap=overlap)
    gpu['CUDA']['device'] = device
    if xcorrMaskSize is None:
        xcorrMaskSize = 5 if protocol == 'bench_BPSK' else 3
    gpu['xcorrMaskSize'] = xcorrMaskSize
    return {
        'Main': {'workerId': protocol, 'PacketLen': 10000, 'RandSeed': 123, 'protocols': {'UHF': protocol}},
        'GPU': {'UHF': gpu, 'STX': copy.deepcopy(BASE_GPU_STX)},
        'Radios': {
            'rangeRateMax': rangeRateMax,
            'Rx': {radio: {'CUDA_settings': 'UHF', 'frequency_Hz': 437.3e6, 'frequencyOffset_Hz': 38400,
                           'baud': 9600, 'samplesPerSym': 16, 'doppCarrierSteps': doppCarrierSteps,
                           'Protocol': 'UHF', 'radioBackend': 'UHF'}},
        },
    }


def cc11xx_config(blockSize=16, overlap=10, doppCarrierSteps=64, rangeRateMax=27500, samplesPerSym=128,
                  device=0, radio='UHF-H'):
    """config/CC11xx.json geometry: 401.538 MHz, IF offset 148.32 kHz, 7416 baud."""
    gpu = copy.deepcopy(BASE_GPU_UHF)
    gpu.update(blockSize=blockSize, overlap=overlap)
    gpu['CUDA']['device'] = device
    return {
        'Main': {'workerId': 'gpu-sdr', 'protocols': {'UHF': 'CC11xx'}},
        'GPU': {'UHF': gpu, 'STX': copy.deepcopy(BASE_GPU_STX)},
        'Radios': {
            'rangeRateMax': rangeRateMax,
            'Protocol': {'rx_preamble': ['0xaa'] * 4, 'rx_sync_seq': ['0xd6', '0xba', '0xd6', '0xba'],
                         'tx_preamble': ['0xaa'], 'tx_num_preambles': 10,
                         'tx_sync_seq': ['0xd6', '0xba', '0xd6', '0xba']},
            'Rx': {radio: {'name': 'UHF', 'CUDA_settings': 'UHF', 'frequency_Hz': 401.538e6,
                           'frequencyOffset_Hz': 148320, 'baud': 7416, 'samplesPerSym': samplesPerSym,
                           'doppCarrierSteps': doppCarrierSteps, 'Protocol': 'UHF', 'radioBackend': 'UHF'}},
        },
    }


def strip_json_comments(text):
    """Remove // line comments and /* */ block comments that are outside string literals
    (``"tcp://*:5512"`` must survive)."""
    out, i, n = [], 0, len(text)
    in_str = False
    while i < n:
        ch = text[i]
        if in_str:
            out.append(ch)
            if ch == '\\' and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
            out.append(ch)
        elif ch == '/' and i + 1 < n and text[i + 1] == '/':
            while i < n and text[i] != '\n':
                i += 1
            continue
        elif ch == '/' and i + 1 < n and text[i + 1] == '*':
            end = text.find('*/', i + 2)
            i = n if end < 0 else end + 2
            continue
        else:
            out.append(ch)
        i += 1
    return ''.join(out)


def merge_config(base, override):
    """Nested dictionaries: ``override`` wins key by key, dictionaries are merged recursively."""
    out = copy.deepcopy(base)
    for k, v in override.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge_config(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def load_config(path, _seen=()):
    """Load a config file in the reference's format: JSON with comments whose optional "configBase" names a base
    file (relative to the file that names it) that is loaded first and overridden key by key."""
    path = os.path.abspath(path)
    if path in _seen:
        raise ValueError(f'configBase cycle through {path}')
    with open(path) as f:
        conf = json.loads(strip_json_comments(f.read()))
    base = conf.pop('configBase', None)
    if base is not None:
        conf = merge_config(load_config(os.path.join(os.path.dirname(path), base), _seen + (path,)), conf)
    return conf
