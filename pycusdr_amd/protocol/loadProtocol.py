"""Protocol plugin registry.

``loadProtocol(name)`` returns the plugin class for the names the reference's loader knows
(reference protocol/loadProtocol.py:3-20) and raises ImportError for anything else.  Plugins are
imported lazily so that a receiver only pays for the protocol it runs.
"""
import importlib

# name -> (module relative to this package, class name)
_REGISTRY = {
    'CC11xx': ('.CC11xx', 'CC11xx'),
    'CC11xx_GFSK2': ('.CC11xx', 'CC11xx_GFSK2'),     # the reference's modIDX = 1 flavour (CC11xx.py:30-32)
    'bench_GMSK': ('.benchmark.bench_GMSK', 'Bench_GMSK'),
    'bench_BPSK': ('.benchmark.bench_BPSK', 'Bench_BPSK'),
    'bench_FSK': ('.benchmark.bench_FSK', 'Bench_FSK'),
    'bench_GFSK': ('.benchmark.bench_GFSK', 'Bench_GFSK'),
}


def known_protocols():
    return sorted(_REGISTRY)


def loadProtocol(protocolName):
    try:
        module, cls = _REGISTRY[protocolName]
    except KeyError:
        raise ImportError('Protocol %s does not exist' % (protocolName,)) from None
    return getattr(importlib.import_module(module, __package__), cls)
