"""Config dictionaries with the reference's key layout (config/base.json, config/benchmark/*.json,
config/CC11xx.json).  The layered JSON-with-comments loader of the reference is out of scope; these
helpers just build the dicts the Demodulator / protocol plugins read."""
import copy

BASE_GPU_UHF = {
    'blockSize': 16, 'overlap': 10, 'bitWindowWidth': 7, 'xcorrMaskSize': 3, 'clippedPeakSpan': 20,
    'symbol_check_overlap_offset': 20, 'symbol_check_error_threshold': 1000,
    'symbol_check_match_num_errors_allowed': 10, 'doppCarrierSteps': 64, 'peakThresholdScale': 40.5,
    'CUDA': {'device': 0, 'numThreads': 64, 'numThreadsS': 1024, 'batchSize': 0, 'streams': 3, 'num_SMS': 0},
}
BASE_GPU_STX = dict(BASE_GPU_UHF, blockSize=17, overlap=11, bitWindowWidth=3, clippedPeakSpan=40, peakThresholdScale=4.5)


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
