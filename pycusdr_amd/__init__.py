"""pycusdr_amd -- MI355X-native Doppler matched-filter-bank hot path of pyCuSDR.

Host side mirrors the reference's Demodulator / protocol-plugin / Decoder call shapes; the device
side is libmfbank.so (hand-written HIP for gfx950) behind the C ABI in include/mfbank.h.
"""
__version__ = '0.1.0'
