"""Synthetic stimulus for parity tests and bench.py -- our generator of the reference's BER-bench
packets (reference examples/benchmark/create_signals.py:10-201), pinned by fixtures G5.

S1 (parity + timing): the seed-123 10 000-bit packet, modulated, zero-padded by 10 000 samples on
either side and mixed to +fs/4.  S2 (pure throughput): unit-variance complex Gaussian noise.
"""
import numpy as np

from .lib.filters import gaussianFilter, rrcosfilter


def createBitSequence(n_bits, seed=None):
    """Legacy-RNG bit sequence; a truthy seed gives the deterministic sequence without disturbing
    the global RNG state."""
    if seed:
        return np.random.RandomState(seed).randint(0, 2, n_bits)
    return np.random.randint(0, 2, n_bits)


def packetData():
    return createBitSequence(10000, seed=123)


def zeropad(sig, n):
    z = np.zeros(n)
    return np.concatenate((z, sig, z))


def encodeNRZS(bitData):
    """NRZ-S: a 1 keeps the level, a 0 toggles it; the first output equals the first bit."""
    bits = np.asarray(bitData).astype(np.uint8)
    toggles = np.concatenate(([0], (bits[1:] == 0).astype(np.uint8)))
    return (bits[0] ^ (np.cumsum(toggles) & 1)).astype(np.uint8)


def modulateBPSK(raw_bits, sps):
    levels = encodeNRZS(np.concatenate(([1, 0, 1], raw_bits))).astype(float) * 2 - 1
    taps = rrcosfilter(0.5, 6, sps)
    taps = taps / np.sum(taps)
    return np.convolve(taps, np.repeat(levels, sps)).astype(np.complex64)


def modulateFSK(raw_bits, sps):
    raw_bits = np.asarray(raw_bits)
    step = np.ones(sps) / sps * np.pi
    lut = np.array([-step, step])
    phase = np.cumsum(lut[raw_bits]) - (raw_bits[0] * 2 - 1) * np.pi / 2
    return np.exp(1j * np.mod(phase, 2 * np.pi)).astype(np.complex64)


def modulateGFSK2(raw_bits, sps):
    taps = gaussianFilter(1, 1, sps, 4 * sps)
    phase = np.convolve(taps, np.repeat(np.asarray(raw_bits) * 2 - 1, sps))
    return np.exp(1j * np.cumsum(phase) / sps * np.pi).astype(np.complex64)


def modulateGMSK(raw_bits, sps):
    taps = gaussianFilter(1, 0.5, sps, 4 * sps)
    phase = np.convolve(taps, np.repeat(np.asarray(raw_bits) * 2 - 1, sps))
    return np.exp(1j * np.cumsum(phase) / sps * np.pi / 2).astype(np.complex64)


_MODS = {'BPSK': modulateBPSK, 'GMSK': modulateGMSK, 'FSK': modulateFSK, 'GFSK': modulateGFSK2}


def get_padded_packet(modulation, spSym=16, fs=9600 * 16, offset_freq=None, raw_bits=()):
    """(signal complex128, payload bits): modulated packet, 10 000 zero samples either side, mixed
    to ``offset_freq`` (default fs/4)."""
    if offset_freq is None:
        offset_freq = fs / 4
    if len(raw_bits) == 0:
        raw_bits = packetData()
    if modulation not in _MODS:
        raise TypeError('Only supports GMSK, FSK, GFSK and BPSK')
    sig = zeropad(_MODS[modulation](raw_bits, spSym), 10000)
    sig = sig * np.exp(1j * 2 * np.pi * offset_freq / fs * np.arange(len(sig)))
    return sig, raw_bits


def awgn(sig, snr, measured=True, rng=None):
    """AWGN channel with the reference's semantics (create_signals.py:115-141); ``rng`` is a
    RandomState (default: the legacy global RNG, as the reference uses)."""
    rng = np.random if rng is None else rng
    if measured:
        sigp = 10 * np.log10(np.linalg.norm(np.abs(sig), 2) ** 2 / len(sig))
        snr = snr - sigp
    noiseP = 10 ** (-snr / 10)
    if np.iscomplexobj(sig):
        return sig + np.sqrt(noiseP / 2) * (rng.randn(len(sig)) + 1j * rng.randn(len(sig)))
    return sig + np.sqrt(noiseP) * rng.randn(len(sig))


def s1_stream(nblocks, N, ov, modulation='GMSK', spSym=16, fs=153600, snr_db=10.0, seed=1):
    """S1: the padded packet tiled to cover ``nblocks`` blocks of N samples advancing by N-ov,
    plus AWGN from RandomState(seed).  Returns complex64 [ov + nblocks*(N-ov)]."""
    pkt, _ = get_padded_packet(modulation, spSym, fs)
    need = ov + nblocks * (N - ov)
    reps = -(-need // len(pkt))
    clean = np.tile(pkt, reps)[:need]
    if snr_db is None:
        return clean.astype(np.complex64)
    return awgn(clean, snr_db, rng=np.random.RandomState(seed)).astype(np.complex64)


def s2_noise(nblocks, N, seed=0):
    """S2: RandomState(seed) standard-normal I and Q, complex64 [nblocks, N]."""
    rs = np.random.RandomState(seed)
    out = np.empty((nblocks, N), dtype=np.complex64)
    for b in range(nblocks):
        out[b] = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    return out
