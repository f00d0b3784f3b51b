"""CPU oracle for the Doppler matched-filter-bank hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain numpy restatement of what the reference (pyCuSDR) computes on the GPU for
the receive hot path.  It exists to *check* the HIP implementation; nothing in the product path
(``pycusdr_amd/``) imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.

PARITY PINNING.  The reference has no test, fixture or golden vector for its GPU kernels
 and its device code cannot be compiled or run here (CUDA, PyCUDA and
closed-source cuFFT).  For the device stages (A3..A11) this oracle is therefore **parity
unpinned** against the reference: it follows the reference's source text (citations below) and
is checked by invariants (Parseval, known carrier bin, zero bit errors on the reference's own
bench packet).  The host-side stages that *can* be pinned are pinned by fixtures generated from
the importable numpy parts of the reference (tests/golden/make_golden.py): filter banks, LUTs,
decoder templates, Decoder.findFrames, bench stimulus, checkSymbolOverlap, extractBitsNRZs -- and, since round 4,
by fixtures recorded from the reference's own constructor / uploadAndFindCarrier / demodulate run under a recording
fake of its driver (tests/golden/make_golden_host.py, G15-G17): ``doppler_table``, ``interpolate_doppler``,
``compute_snr`` and ``code_rate_host`` below reproduce those bit for bit (tests/test_oracle.py).

Citations are file:line under /root/reference/pyCuSDR/ with
  DB = demodulator/demodulator_base.py,  CU = demodulator/cuda_kernels.cu,  DEC = decoder.py.

FFT conventions (lib/cufft.py:128-129): forward sign -1, inverse sign +1, both unnormalised --
i.e. ``np.fft.fft`` and ``N * np.fft.ifft``.

Floating-point evaluation order.  Where the reference's fp32 index arithmetic decides an integer
result (findCentres, findDopplerEst) the evaluation order is fixed here and used verbatim by the
HIP kernels:
  * ``a*b + c`` / ``a*b - c`` written in one CUDA expression is taken as ONE fused multiply-add
    (nvcc's default -fmad=true contracts it), everything else rounds after each operation.
"""
import numpy as np

SCALE_2_18 = 262144.0  # CU:442
C_LIGHT = 299792458.0  # scipy.constants.speed_of_light, DB:143

f32 = np.float32


# --------------------------------------------------------------------------------------------
# A1  Doppler-bin table                                                       DB:130-165
# --------------------------------------------------------------------------------------------
def doppler_table(frequency_Hz, frequencyOffset_Hz, baud, spsym, rangeRateMax, num_dopplers, N,
                  noise_measure_offset_Hz=False):
    """Returns dict(shifts int32[Dtot], doppHzLUT f64[Dtot], doppIdxNorm, offset_count,
    doppOffsetIdx).  Follows DB:130-165 line by line (float64 host arithmetic)."""
    Fc = frequency_Hz - frequencyOffset_Hz                                  # DB:132
    doppOffset = frequencyOffset_Hz / baud / spsym                          # DB:136
    doppOffsetIdx = np.int32(doppOffset * N)                                # DB:137 (truncation)
    if doppOffsetIdx < 0:
        doppOffsetIdx += N                                                  # DB:138-139
    sampleRate = baud * spsym                                               # DB:106
    doppMax = rangeRateMax * Fc / C_LIGHT                                   # DB:143
    doppMaxNorm = doppMax / sampleRate                                      # DB:144
    lo, hi = doppOffset - doppMaxNorm, doppOffset + doppMaxNorm             # DB:145-146
    grid = np.linspace(lo, hi, num_dopplers)
    if noise_measure_offset_Hz:                                             # DB:150-155
        grid = np.concatenate((np.array([noise_measure_offset_Hz / baud / spsym]), grid))
    hz = grid * spsym * baud                                                # DB:162
    shifts = np.round(grid * N).astype(np.int32)                            # DB:164
    shifts[shifts < 0] += N                                                 # DB:165
    return dict(shifts=shifts, doppHzLUT=hz, doppIdxNorm=grid,
                offset_count=len(grid) - num_dopplers, doppOffsetIdx=int(doppOffsetIdx))


# --------------------------------------------------------------------------------------------
# A3  forward FFT                                                             DB:548-558
# --------------------------------------------------------------------------------------------
def forward_fft(x, dtype=np.complex64):
    """Unnormalised forward FFT; the reference keeps the spectrum as complex64 (DB:459)."""
    X = np.fft.fft(np.asarray(x, dtype=np.complex128))
    return X.astype(dtype)


# --------------------------------------------------------------------------------------------
# A4-A6  shift-multiply, batched inverse FFT, |.|^2 row sums     CU:339-373, DB:578-588, CU:421-480
# --------------------------------------------------------------------------------------------
def shifted_product(X, mask_row, shift):
    """xc[k] = X[(k+shift) mod N] * mask[k]                                  CU:370, CU:933-940"""
    return np.roll(X, -int(shift)) * mask_row


def doppler_scores(X, masks, shifts, sum_all_masks=True, dtype=np.complex128):
    """doppSum as the reference's blockAbsSumAtomic leaves it (CU:421-480), in float64.

    Returns float64 [Dtot, M].  With SUM_ALL_MASKS only column 0 is populated (CU:453-464,
    quirk Q2); otherwise column m holds mask m's sum (CU:472-475).
    ``dtype`` complex64 gives a float32-arithmetic run via scipy (used as the CPU baseline).
    """
    X = np.asarray(X)
    masks = np.asarray(masks)
    N = X.shape[0]
    M = masks.shape[0]
    out = np.zeros((len(shifts), M), dtype=np.float64)
    if dtype == np.complex128:
        ifft = lambda a: np.fft.ifft(a, axis=-1) * N
        Xw, Mw = X.astype(np.complex128), masks.astype(np.complex128)
    else:
        import scipy.fft as sfft
        ifft = lambda a: sfft.ifft(a, axis=-1, norm='forward')   # unnormalised inverse
        Xw, Mw = X.astype(np.complex64), masks.astype(np.complex64)
    for j, s in enumerate(shifts):
        prod = np.roll(Xw, -int(s))[None, :] * Mw                            # A4
        y = ifft(prod)                                                       # A5
        e = (y.real.astype(np.float64) ** 2 + y.imag.astype(np.float64) ** 2).sum(axis=1)
        e /= SCALE_2_18                                                      # A6, CU:442
        if sum_all_masks:
            out[j, 0] = e.sum()
        else:
            out[j, :] = e
    return out


def doppler_scores_parseval(X, masks, shifts):
    """The IFFT-free identity  sum_n |IFFT(P)[n]|^2 = N * sum_k |P[k]|^2  (cross-check only)."""
    X = np.asarray(X, dtype=np.complex128)
    w = (np.abs(np.asarray(masks, dtype=np.complex128)) ** 2).sum(axis=0)
    p = np.abs(X) ** 2
    N = len(X)
    return np.array([N * np.dot(np.roll(p, -int(s)), w) / SCALE_2_18 for s in shifts])


# --------------------------------------------------------------------------------------------
# A7  Doppler pick                                                   CU:502-597, DB:604-632
# --------------------------------------------------------------------------------------------
def _top2_column(col, num_elements, element_offset):
    """One thread of findDopplerEst (CU:527-554) in fp32.  Returns (idxL, valL)."""
    maxVal = [f32(0), f32(0)]
    maxIdx = [0, 0]
    cur = 0
    for i in range(element_offset, num_elements + element_offset):           # CU:534
        tmp = f32(col[i])
        if tmp > maxVal[cur]:                                                # strict, CU:537
            maxVal[cur] = tmp
            maxIdx[cur] = i
            cur = 1 if maxVal[0] >= maxVal[1] else 0                         # CU:541
    # CU:546  tmp = i0*v0 + i1*v1  -> mul, then fused mul-add (see module docstring)
    prod1 = f32(f32(maxIdx[1]) * maxVal[1])
    num = f32(np.float64(f32(maxIdx[0])) * np.float64(maxVal[0]) + np.float64(prod1))
    with np.errstate(divide='ignore', invalid='ignore'):
        idxL = f32(num / f32(maxVal[0] + maxVal[1]))                         # CU:547
        valL = f32(num / f32(maxIdx[0] + maxIdx[1]))                         # CU:548 (int sum)
        if element_offset > 0:                                               # CU:550-554
            valL = f32(maxVal[(cur + 1) % 2] / f32(col[0]))
    return idxL, valL


def find_doppler_est(dopp_sum, num_elements, element_offset=0, sum_all_masks=True):
    """[idx, metric] as written to ``res`` by findDopplerEst.  ``dopp_sum`` float32 [Dtot, M].

    SUM_ALL_MASKS: thread 0 only (CU:560-567).  Otherwise the mean over masks of the per-mask
    estimates (CU:576-593); the reference reduces with warp shuffles over a partially active
    warp, which is undefined in CUDA for M < 32 -- the documented intent (mean) is restated,
    summing in a fixed xor-butterfly tree order.
    """
    ds = np.asarray(dopp_sum, dtype=np.float32)
    M = ds.shape[1]
    with np.errstate(divide='ignore', invalid='ignore'):
        if sum_all_masks:
            idxL, valL = _top2_column(ds[:, 0], num_elements, element_offset)
            return f32(idxL), f32(f32(10) * np.log10(valL, dtype=np.float32))
        per = [_top2_column(ds[:, m], num_elements, element_offset) for m in range(M)]

        def tree(vals):
            vals = [f32(v) for v in vals]
            n = 1
            while n < len(vals):
                n *= 2
            vals += [f32(0)] * (n - len(vals))
            step = n // 2
            while step >= 1:
                vals = [f32(vals[i] + vals[i ^ step]) for i in range(n)]
                step //= 2
            return vals[0]
        idx = f32(tree([p[0] for p in per]) / f32(M))
        val = f32(tree([p[1] for p in per]) / f32(M))
        return idx, f32(f32(10) * np.log10(val, dtype=np.float32))


def interpolate_doppler(best_idx, shifts, dopp_hz_lut, centre_freq_offset=0.0):
    """Host part of __findUHF (DB:609-632).  Returns dict or None when the block is skipped
    (NaN index -> ValueError in int(), DB:625-630)."""
    b = float(best_idx)
    try:
        low = int(b)                                                         # DB:610
        high = int(np.ceil(b))                                               # DB:611
    except (ValueError, OverflowError):
        return None
    frac = b % 1                                                             # DB:615
    hz = dopp_hz_lut[low] + (dopp_hz_lut[high] - dopp_hz_lut[low]) * frac
    s_lo, s_hi = int(shifts[low]), int(shifts[high])
    idx_last = np.int32(np.round(s_lo + (s_hi - s_lo) * frac))               # DB:618
    return dict(low=low, high=high, hz=hz, dopplerIdxlast=int(idx_last),
                freqOffset=hz - centre_freq_offset)                          # DB:622


# --------------------------------------------------------------------------------------------
# A8  SNR                                                                    DB:635-667
# --------------------------------------------------------------------------------------------
def compute_snr(X, shifts, low, high, width, N):
    X = np.asarray(X)
    lo_i, hi_i = int(shifts[low]), int(shifts[high])                         # DB:644-645
    nlo, nhi = (lo_i + N // 2) % N, (hi_i + N // 2) % N                      # DB:647-648

    def band(a, b):
        if a > b:                                                            # DB:653-656
            return np.mean(np.concatenate((np.abs(X[a - width:]), np.abs(X[:b + width]))))
        return np.mean(np.abs(X[a - width:b + width]))
    with np.errstate(divide='ignore', invalid='ignore'):
        return 20 * np.log10(band(lo_i, hi_i) / band(nlo, nhi) - 1)          # DB:663


# --------------------------------------------------------------------------------------------
# A9  demodulation matched filters at the chosen shift         CU:174-185, DB:776-785
# --------------------------------------------------------------------------------------------
def demod_xcorr(X, masks, shift):
    """xc[m][n] = N*ifft(X[(k+shift) mod N] * masks[m][k]), complex128."""
    X = np.asarray(X, dtype=np.complex128)
    prod = np.roll(X, -int(shift))[None, :] * np.asarray(masks, dtype=np.complex128)
    return np.fft.ifft(prod, axis=1) * X.shape[0]


# --------------------------------------------------------------------------------------------
# A10  symbol rate / phase                       CU:191-205, DB:721, CU:236-320, DB:730-752
# --------------------------------------------------------------------------------------------
def abs2_f32(z):
    """ComplexAbsSquared (CU:1022-1026) in fp32: fma(x, x, y*y)."""
    z = np.asarray(z, dtype=np.complex64)
    x = z.real.astype(np.float64)
    y2 = (z.imag * z.imag).astype(np.float32).astype(np.float64)
    return (x * x + y2).astype(np.float32)


def envelope(xc, code_search_mask_offset=0):
    """p[n] = sum_m |xc[m][n]|^2 over masks [off, M-off)  (CU:191-205), float64."""
    xc = np.asarray(xc)
    M = xc.shape[0]
    sel = xc[code_search_mask_offset:M - code_search_mask_offset]
    return (sel.real.astype(np.float64) ** 2 + sel.imag.astype(np.float64) ** 2).sum(axis=0)


def code_rate_window(N, spsym):
    """(offset, length) handed to findCodeRateAndPhase (DB:508-512, 726)."""
    lo = int(N / (0.9 * spsym))
    hi = int(N / (1.1 * spsym))
    return hi, lo - hi


def code_rate_and_phase(env, offset, length):
    """[k*, arg(P[k*]), |P[k*]|^2] (CU:236-320): P = rfft(env); argmax of |P|^2 over
    [offset, offset+length); ties -> lowest index.  float64: the true maximum.  The reference squares in fp32
    (ComplexAbsSquared CU:1022-1026), which overflows to +inf for strong signals in long blocks with long filters
    (|P|^2 ~ M^2 N^6 T^4 a^8: amplitude-1 samples, 384 taps, N = 2^20); what its warp butterfly returns among several +inf
    is an artefact of the shuffle pattern and is NOT restated here.  Wherever the fp32 squares are finite the fp32 argmax
    and this one pick the same bin (the HIP kernel compares exactly scaled fp32 squares: small_kernels.hpp)."""
    P = np.fft.rfft(np.asarray(env, dtype=np.float64))
    w = np.abs(P[offset:offset + length]) ** 2
    k = int(np.argmax(w)) + offset
    return k, float(np.angle(P[k])), float(np.abs(P[k]) ** 2)


def code_rate_host(k, arg, N):
    """spSym, codeOffset (DB:733-752), float64 host arithmetic on fp32 inputs."""
    kf = float(f32(k))
    if kf == 0.0:
        spSym = 10.0                                                         # DB:737-740
    else:
        spSym = N / kf                                                       # DB:735
    codeOffset = -float(f32(arg)) / np.pi * spSym / 2                        # DB:745
    if codeOffset < 0:
        codeOffset += spSym - 1                                              # DB:746-747
    return spSym, codeOffset


# --------------------------------------------------------------------------------------------
# A11  symbol centres                                             CU:78-146, DB:991-1009
# --------------------------------------------------------------------------------------------
def _fma32(a, b, c):
    """fmaf(a, b, c) for fp32 inputs: exact in float64 for the magnitudes used here."""
    return (np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64)
            + np.asarray(c, np.float32).astype(np.float64)).astype(np.float32)


def find_centres(xc, spSym, offset, window_width=7, op=0, spsym_min=None):
    """findCentres for every symbol index x with a start inside the signal.

    ``xc`` complex64 [M, N] (the device's matched-filter outputs, so that symbol decisions are
    compared on identical inputs); ``spSym``/``offset`` are rounded to fp32 as DB:997 does.
    Returns (sym int32[S], centre int32[S], mag float32[S]) with S = int(N/spSym) (DB:999).
    op: 0 |.|^2, 1 |re|, 2 |im|  (CU:108-121).
    """
    xc = np.asarray(xc, dtype=np.complex64)
    M, N = xc.shape
    if spsym_min is not None and spSym < spsym_min:                          # DB:994-995
        spSym = spsym_min
    S = int(N / spSym)                                                       # DB:999
    sp = f32(spSym)
    off = f32(offset)
    W = int(window_width)
    half = f32(W // 2)                                                       # WINDOW_WIDTH/2, DB:407
    x = np.arange(S, dtype=np.int64)
    xf = x.astype(np.float32)
    base = _fma32(xf, sp, -half)                                             # (float)x*spSym - 3
    start = (base + off).astype(np.float32)                                  # ... + offset
    arrayIdx = np.trunc(start).astype(np.int64)                              # (int) cast, CU:88
    maxArrayIdx = arrayIdx + W                                               # CU:89
    offsetComp = np.full(S, int(np.trunc(off)), dtype=np.int64)              # CU:91
    neg = arrayIdx < 0
    offsetComp[neg] -= arrayIdx[neg]                                         # CU:96
    arrayIdx = np.where(neg, 0, arrayIdx)                                    # CU:97
    maxArrayIdx = np.minimum(maxArrayIdx, N)                                 # CU:99-100
    wlen = maxArrayIdx - arrayIdx                                            # CU:102
    valid = arrayIdx < N                                                     # CU:105

    if op == 0:
        val = abs2_f32(xc)
    elif op == 1:
        val = np.abs(xc.real).astype(np.float32)
    else:
        val = np.abs(xc.imag).astype(np.float32)

    sym = np.full(S, -1, dtype=np.int32)
    kbest = np.full(S, -1, dtype=np.int64)
    mag = np.zeros(S, dtype=np.float32)
    # row-major scan (mask outer, window inner) with strict '>' (CU:125-139): the winner is the
    # first occurrence of the maximum in that scan order, provided it is > 0.
    kk = np.arange(W)
    idx = arrayIdx[:, None] + kk[None, :]                                    # [S, W]
    inwin = (kk[None, :] < wlen[:, None]) & valid[:, None]
    idxc = np.clip(idx, 0, N - 1)
    cand = val[:, idxc]                                                      # [M, S, W]
    cand = np.where(inwin[None, :, :], cand, f32(-1))
    flat = np.transpose(cand, (1, 0, 2)).reshape(S, M * W)                   # scan order m, k
    am = np.argmax(flat, axis=1)
    best = flat[np.arange(S), am]
    hit = best > 0
    sym[hit] = (am[hit] // W).astype(np.int32)
    kbest[hit] = am[hit] % W
    mag[hit] = best[hit]
    # CU:142  (int)(x*spSym - 3 + maxCentreIdx + offsetComp)
    c = (base + kbest.astype(np.float32)).astype(np.float32)
    c = (c + offsetComp.astype(np.float32)).astype(np.float32)
    centre = np.trunc(c).astype(np.int32)
    # threads whose start lies outside the signal write nothing (CU:105): mark with sentinel
    sym[~valid] = np.iinfo(np.int32).min
    centre[~valid] = np.iinfo(np.int32).min
    return sym, centre, mag


# --------------------------------------------------------------------------------------------
# A14  sync / preamble correlation                                          DEC:96-113
# --------------------------------------------------------------------------------------------
def sync_correlate(bits, template):
    """score = np.convolve(bits, template) (full), exact integers."""
    return np.convolve(np.asarray(bits).astype(np.int64), np.asarray(template).astype(np.int64))


def header_candidates(score, num_ones, tol, template_len):
    """idxCand, packetIdx (DEC:101-104)."""
    idx = np.where(score >= num_ones - tol)[0]
    return idx, idx - template_len + 1


# --------------------------------------------------------------------------------------------
# whole find_carrier on the CPU (used as cpu_baseline 'port' and by the parity tests)
# --------------------------------------------------------------------------------------------
def find_carrier(x, masks, shifts, num_dopplers, element_offset=0, sum_all_masks=True,
                 dtype=np.complex128):
    """A3..A7 device part: returns (X complex64, doppSum float32 [Dtot, M], idx, metric)."""
    X = forward_fft(x)
    ds = doppler_scores(X, masks, shifts, sum_all_masks, dtype=dtype).astype(np.float32)
    idx, metric = find_doppler_est(ds, num_dopplers, element_offset, sum_all_masks)
    return X, ds, idx, metric


# --------------------------------------------------------------------------------------------
# N4  bit-stream alignment cross-correlation                       lib/customXCorr.py:5-18
# --------------------------------------------------------------------------------------------
def custom_xcorr(a, b, N=None):
    """ifft(fft(a, N) * conj(fft(b, N))), N = max(len) by default (fft truncates or zero-pads to N).
    Pinned by fixture G13 (the reference's own soft-combiner test streams)."""
    a, b = np.asarray(a), np.asarray(b)
    if N is None:
        N = max(len(a), len(b))
    return np.fft.ifft(np.fft.fft(a, N) * np.conj(np.fft.fft(b, N)), N)
