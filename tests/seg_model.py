"""numpy model of the single-pass overlap-save search (test infrastructure).

Restates, index for index, what ``k_seg`` in pycusdr_amd/csrc/mfbank.hip does with L-point
segments, so that the algebra (support window, rotated taps, validity ranges, output offset,
power-of-two scaling) is checked on the CPU against the reference formulation
``IFFT_N(X[(k+s) mod N] * H_m[k])`` (reference cuda_kernels.cu:339-373 + demodulator_base.py:578-591)
before the kernel runs on a GPU.
"""
import numpy as np


def filter_support(masks, thr=1e-13):
    """Common circular support window [a, a+T) of the impulse responses h_m = ifft(H_m): every
    sample with |h_m[n]|^2 > thr * E_m / N for any m lies inside (mfb_analyze_filters)."""
    masks = np.asarray(masks)
    M, N = masks.shape
    h = np.fft.ifft(masks.astype(np.complex128), axis=1)
    e = np.abs(h) ** 2
    E = e.sum(axis=1, keepdims=True)
    sig = (e > thr * E / N).any(axis=0)
    idx = np.where(sig)[0]
    if len(idx) == 0:
        return 0, 1, h
    gaps = np.diff(np.r_[idx, idx[0] + N])
    k = int(np.argmax(gaps))                 # first largest gap, like the C++ scan
    a = int(idx[(k + 1) % len(idx)])
    T = int(N - gaps[k] + 1)
    return a, T, h


def segment_spectra(h, a, T, L):
    """G[m][k] = (N/L) * FFT_L(c'_m) with c'_m the T taps h_m[(a+r) mod N] rotated so that the valid
    outputs of every segment are i = 0 .. L-T."""
    M, N = h.shape
    c = h[:, (a + np.arange(T)) % N]
    cp = np.zeros((M, L), dtype=np.complex128)
    cp[:, (np.arange(T) - (T - 1)) % L] = c
    return (N / L) * np.fft.fft(cp, axis=1)


def segment_search(x, masks, shifts, L, sum_all=True, dtype=np.complex128, thr=1e-13):
    """doppSum [D][M] by overlap-save over L-point segments; also returns (a, T, V, Q)."""
    x = np.asarray(x, dtype=np.complex128)
    N = len(x)
    a, T, h = filter_support(masks, thr)
    V = L - T + 1
    Q = -(-N // V)
    G = segment_spectra(h, a, T, L).astype(dtype)
    M = G.shape[0]
    n = np.arange(N)
    out = np.zeros((len(shifts), M))
    for j, s in enumerate(shifts):
        xs = (x * np.exp(-2j * np.pi * ((int(s) * n) % N) / N)).astype(dtype)
        e = np.zeros(M)
        for q in range(Q):
            b0 = q * V
            u = xs[(b0 + np.arange(L)) % N]
            U = np.fft.fft(u)
            v = np.fft.ifft(U[None, :] * G, axis=1) * L
            nv = min(V, N - b0)
            e += (np.abs(v[:, :nv]) ** 2).sum(axis=1)
        e /= 262144.0
        if sum_all:
            out[j, 0] = e.sum()
        else:
            out[j] = e
    return out, (a, T, V, Q)


def segment_xcorr(x, masks, shift, L, thr=1e-13):
    """y[m][n] in natural order (the STORE mode of k_seg): what IFFT_N(X[(k+s) mod N] H_m[k]) gives."""
    x = np.asarray(x, dtype=np.complex128)
    N = len(x)
    a, T, h = filter_support(masks, thr)
    V = L - T + 1
    Q = -(-N // V)
    G = segment_spectra(h, a, T, L)
    n = np.arange(N)
    xs = x * np.exp(-2j * np.pi * ((int(shift) * n) % N) / N)
    y = np.zeros((G.shape[0], N), dtype=np.complex128)
    off = (a + T - 1) % N
    for q in range(Q):
        b0 = q * V
        u = xs[(b0 + np.arange(L)) % N]
        v = np.fft.ifft(np.fft.fft(u)[None, :] * G, axis=1) * L
        nv = min(V, N - b0)
        y[:, (b0 + np.arange(nv) + off) % N] = v[:, :nv]
    return y
