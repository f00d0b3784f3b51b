"""Protocol plugin loader -- same names as the reference's protocol/loadProtocol.py:3-20."""


def loadProtocol(protocolName):
    if protocolName == 'CC11xx':
        from .CC11xx import CC11xx as cls
    elif protocolName == 'bench_GMSK':
        from .benchmark.bench_GMSK import Bench_GMSK as cls
    elif protocolName == 'bench_BPSK':
        from .benchmark.bench_BPSK import Bench_BPSK as cls
    elif protocolName == 'bench_FSK':
        from .benchmark.bench_FSK import Bench_FSK as cls
    elif protocolName == 'bench_GFSK':
        from .benchmark.bench_GFSK import Bench_GFSK as cls
    else:
        raise ImportError('Protocol %s does not exist' % (protocolName,))
    return cls
