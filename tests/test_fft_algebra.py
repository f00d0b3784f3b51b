"""Numpy model of the index algebra used by the HIP FFT passes (pycusdr_amd/csrc/fft_core.hpp):
same radix plan, same read/write positions, same twiddles, same natural-index formula, and the
two-pass N = N1*N2 split with the inter-pass twiddle.  Catches algebra bugs without a GPU."""
import numpy as np
import pytest


def plan(l):
    r = [16] * (l // 4)
    if l % 4:
        r.append(1 << (l % 4))
    return r


def small_dft(v):
    R = len(v)
    k = np.arange(R)
    return np.array([np.sum(v * np.exp(2j * np.pi * k * p / R)) for p in range(R)])


def fft_group(L, load):
    radices, NT = plan(int(np.log2(L))), L // 16
    lds, out, Lcur = np.zeros(L, complex), np.zeros(L, complex), L
    for s, R in enumerate(radices):
        Lnext, PC = Lcur // R, L // Lcur
        new = np.zeros(L, complex)
        for g in range(NT):
            for u in range(16 // R):
                beta = g + NT * u
                prefix, t = beta // Lnext, beta % Lnext
                v = np.array([(load(prefix * Lcur + t + Lnext * i) if s == 0 else lds[prefix * Lcur + t + Lnext * i])
                              for i in range(R)])
                y = small_dft(v)
                for p in range(R):
                    pos = (p * PC + prefix) * Lnext + t
                    if s == len(radices) - 1:
                        assert pos == p * PC + NT * u + g            # n = nu + g, nu compile-time
                        out[pos] = y[p]
                    else:
                        new[pos] = y[p] * np.exp(2j * np.pi * t * p / Lcur)
        lds, Lcur = new, Lnext
    return out


@pytest.mark.parametrize('L', [32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_pass_chain_is_an_inverse_dft(L):
    rs = np.random.RandomState(L)
    x = rs.standard_normal(L) + 1j * rs.standard_normal(L)
    y = fft_group(L, lambda p: x[p])
    ref = np.fft.ifft(x) * L
    assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.parametrize('N1,N2', [(32, 32), (32, 64), (64, 128)])
def test_two_pass_split(N1, N2):
    N = N1 * N2
    rs = np.random.RandomState(N)
    x = rs.standard_normal(N) + 1j * rs.standard_normal(N)
    Z = np.zeros((N1, N2), complex)
    for k2 in range(N2):
        Z[:, k2] = fft_group(N1, lambda k1: x[N2 * k1 + k2]) * np.exp(2j * np.pi * k2 * np.arange(N1) / N)
    y = np.zeros(N, complex)
    for n1 in range(N1):
        y[n1 + N1 * np.arange(N2)] = fft_group(N2, lambda k2: Z[n1, k2])
    ref = np.fft.ifft(x) * N
    assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-11


def test_two_level_twiddle_table():
    b = 12
    N, lo = 1 << b, (b + 1) // 2
    tl = np.exp(2j * np.pi * np.arange(1 << lo) / N)
    th = np.exp(2j * np.pi * np.arange(N >> lo) * (1 << lo) / N)
    t = np.arange(N)
    assert np.abs(th[t >> lo] * tl[t & ((1 << lo) - 1)] - np.exp(2j * np.pi * t / N)).max() < 1e-13
