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


# ---- the wave-local 2048-point transform (fft_core.hpp: bfly32, fft_w32; seg_kernels.hpp MFB_SEG_W32) ----------------------
# Register-level model: 64 lanes x 32 registers, the same slot maps, exchange addresses, lane-pair combination and twiddles.
def _b2(v, a, b):
    t = v[a]
    v[a], v[b] = t + v[b], t - v[b]


def _b2_bi(v, a, b):
    t, u = v[a], v[b]
    v[a], v[b] = t + 1j * u, t - 1j * u


def _b4(v, i0, i1, i2, i3):
    a0, a1, a2, a3 = v[i0], v[i1], v[i2], v[i3]
    t0, t1, t2, t3 = a0 + a2, a0 - a2, a1 + a3, a1 - a3
    v[i0], v[i2], v[i1], v[i3] = t0 + t2, t0 - t2, t1 + 1j * t3, t1 - 1j * t3


def _r8(v, O):
    _b4(v, O, O + 2, O + 4, O + 6)
    _b4(v, O + 1, O + 3, O + 5, O + 7)
    v[O + 3] = v[O + 3] * np.exp(1j * np.pi / 4)
    v[O + 7] = v[O + 7] * np.exp(3j * np.pi / 4)
    _b2(v, O, O + 1)
    _b2(v, O + 2, O + 3)
    _b2_bi(v, O + 4, O + 5)
    _b2(v, O + 6, O + 7)


def _slot32(p):
    """Register slot of output p = p1 + 4 p2 (slot32 of fft_core.hpp; rev(8, p2) = 2 (p2 % 4) + p2 / 4)."""
    p1, p2 = p % 4, p // 4
    return 8 * p1 + 2 * (p2 % 4) + p2 // 4


# the 21 twiddles between the radix-4 and the radix-8 butterflies, as the kernel writes them: (slot, variant of c1/c2/c3/W8)
_C = {1: np.exp(2j * np.pi / 32), 2: np.exp(4j * np.pi / 32), 3: np.exp(6j * np.pi / 32)}
_TW32 = {(1, 1): _C[1], (1, 2): _C[2], (1, 3): _C[3], (2, 1): _C[2], (2, 2): np.exp(1j * np.pi / 4), (2, 3): 1j * np.conj(_C[2]),
         (3, 1): _C[3], (3, 2): 1j * np.conj(_C[2]), (3, 3): 1j * _C[1], (4, 1): np.exp(1j * np.pi / 4), (4, 2): 1j,
         (4, 3): np.exp(3j * np.pi / 4), (5, 1): 1j * np.conj(_C[3]), (5, 2): 1j * _C[2], (5, 3): -np.conj(_C[1]),
         (6, 1): 1j * np.conj(_C[2]), (6, 2): np.exp(3j * np.pi / 4), (6, 3): -_C[2], (7, 1): 1j * np.conj(_C[1]),
         (7, 2): -np.conj(_C[2]), (7, 3): -1j * np.conj(_C[3])}


def _bfly32(v):
    for t in range(8):
        _b4(v, t, t + 8, t + 16, t + 24)
    for (t, p1), w in _TW32.items():
        assert abs(w - np.exp(2j * np.pi * t * p1 / 32)) < 1e-15       # every variant IS W_32^(t p1)
        v[t + 8 * p1] = v[t + 8 * p1] * w
    for p1 in range(4):
        _r8(v, 8 * p1)


def test_in_register_32_point_butterfly():
    rs = np.random.RandomState(32)
    x = rs.standard_normal(32) + 1j * rs.standard_normal(32)
    v = [np.array([x[i]]) for i in range(32)]
    _bfly32(v)
    out = np.array([v[_slot32(p)][0] for p in range(32)])
    assert np.abs(out - np.fft.ifft(x) * 32).max() < 1e-13
    assert sorted(_slot32(p) for p in range(32)) == list(range(32))


def w32_transform(x):
    L, ROW = 2048, 66
    lane = np.arange(64)
    pos = (lane >> 1) + 32 * (lane & 1)                       # pi(lane)
    v = [x[pos + 64 * i].astype(complex) for i in range(32)]  # register i of every lane
    _bfly32(v)
    lds = np.zeros(32 * ROW, complex)
    for p in range(32):
        lds[p * ROW + lane] = v[_slot32(p)] * np.exp(2j * np.pi * pos * p / L)       # tw1, lane-contiguous row p
    row, h = lane >> 1, lane & 1
    w = []
    for j in range(32):
        c1, c2 = lds[row * ROW + 2 * j], lds[row * ROW + 2 * j + 1]                  # ONE 16-byte read: lanes 2j and 2j + 1 wrote them
        sig = np.where(h == 0, 1.0, -1.0)
        tw2 = np.where(h == 0, 1.0, np.exp(2j * np.pi * j / 64))
        w.append((c1 + sig * c2) * tw2)
    _bfly32(w)
    out = np.zeros(L, complex)
    for m in range(32):
        out[pos + 64 * m] = w[_slot32(m)]                     # output m' of lane l is element pi(l) + 64 m': the input layout
    return out


def test_wave_local_2048_point_transform_is_an_inverse_dft():
    rs = np.random.RandomState(2048)
    x = rs.standard_normal(2048) + 1j * rs.standard_normal(2048)
    ref = np.fft.ifft(x) * 2048
    assert np.abs(w32_transform(x) - ref).max() / np.abs(ref).max() < 1e-13
    # exchange image: rows of 66 elements keep the 16-lane store groups and the 16-lane b128 read groups on distinct banks
    ROW = 66
    for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31]):
        slots = {(((l >> 1) * ROW * 2) % 64) // 4 for l in grp}          # 16-byte slot (of 16 per 256-byte line) each pair reads, j = 0
        assert len(slots) == len({l >> 1 for l in grp})


# ---- the fused-twiddle transforms of round 6 (fft_core.hpp: bf_ct, dit_fused, fft256_fused, fft_w32_fused; mfbank.hip: ct_pair) ----
# Register-level model with fp32 rounding after every packed operation: radix-2 decimation in time, in place, every twiddled
# butterfly in tangent form -- b' = b + t (i b), a +- c b' -- the same slots, stage twiddles, table order and exchange addresses.
def _r32(x):
    return np.asarray(x).astype(np.complex64).astype(np.complex128)


def _ct_pair(u, L):
    """(cos, tan) of 2 pi u / L as the host builds them (mfbank.hip ct_pair): fp32 values; exactly pi/2 becomes (2^-40, 2^40)."""
    ang = 2.0 * np.pi * u / L
    c, s = np.cos(ang), np.sin(ang)
    if abs(c) < 1e-9:
        c = 2.0 ** -40
    return float(np.float32(c)), float(np.float32(s / c))


def _brev(x, bits):
    y = 0
    for _ in range(bits):
        y, x = (y << 1) | (x & 1), x >> 1
    return y


def _ct_index(s, qb):
    return (0 if s == 1 else 1 << (s - 2)) + qb


def _bf_ct(v, a, b, c, t, rot):
    bp = _r32(v[b] + t * (1j * v[b]))                 # v_pk_fma: one rounding per component
    w = (1j if rot else 1.0) * c
    v[a], v[b] = _r32(v[a] + w * bp), _r32(v[a] - w * bp)


def _dit_fused(v, levels, tables=None, s0=1):
    """tables: per-lane arrays [lanes][2^(levels-1)][2] of (c, t) in ct_index order (LANE_TW); None: the plain transform with its
    constant twiddles (cos, tan of k pi / 16)."""
    n = 1 << levels
    for s in range(s0, levels + 1):
        off = n >> s
        half = 1 << (s - 2) if s >= 2 else 1
        for a in range(n):
            if a & off:
                continue
            q = _brev(a >> (levels - s + 1), s - 1)
            rot = s >= 2 and q >= half
            qb = q - half if rot else q
            if tables is not None:
                c, t = tables[:, _ct_index(s, qb), 0], tables[:, _ct_index(s, qb), 1]
                _bf_ct(v, a, a + off, c, t, rot)
            elif qb == 0:
                w = 1j if rot else 1.0
                v[a], v[a + off] = _r32(v[a] + w * v[a + off]), _r32(v[a] - w * v[a + off])
            else:
                k = qb * (32 >> s)                     # CtK<k>: angle k pi / 16
                _bf_ct(v, a, a + off, float(np.float32(np.cos(k * np.pi / 16))), float(np.float32(np.tan(k * np.pi / 16))), rot)


def _f256_tables():
    # append_fused_tables, L = 256: [16 lanes][8]
    return np.array([[_ct_pair(u, 256) for u in (8 * g, 4 * g, 2 * g, 2 * g + 32, g, g + 16, g + 32, g + 48)] for g in range(16)])


def _f2048_tables():
    rows = []
    for g in range(64):
        us = [16 * g, 8 * g] + [4 * g + 256 * q for q in range(2)] + [2 * g + 128 * q for q in range(4)] + [g + 64 * q for q in range(8)]
        rows.append([_ct_pair(u, 2048) for u in us])
    return np.array(rows), np.array([_ct_pair(32 * p, 2048) for p in range(32)])


def fft256_fused_model(x):
    g = np.arange(16)
    v = [_r32(x[g + 16 * i]) for i in range(16)]
    _dit_fused(v, 4)
    lds = np.zeros(16 * 17, complex)
    for p in range(16):
        lds[p * 17 + g] = v[_brev(p, 4)]                  # padi(16 p + g) = 17 p + g
    w = [lds[g * 17 + i] for i in range(16)]              # padi(16 g + i)
    _dit_fused(w, 4, _f256_tables())
    out = np.zeros(256, complex)
    for q in range(16):
        out[g + 16 * q] = w[_brev(q, 4)]
    return out


def fft2048_fused_model(x):
    lane = np.arange(64)
    pos = (lane >> 1) + 32 * (lane & 1)
    v = [_r32(x[pos + 64 * i]) for i in range(32)]
    _dit_fused(v, 5)
    ROW = 66
    lds = np.zeros(32 * ROW, complex)
    for p in range(32):
        lds[p * ROW + lane] = v[_brev(p, 5)]
    ct, t0 = _f2048_tables()
    row, h = lane >> 1, lane & 1
    c0 = np.where(h == 0, t0[row, 0], -t0[row, 0])        # f2048_setup: minus on odd lanes
    e = []
    for j in range(32):
        b1, b2 = lds[row * ROW + 2 * j], lds[row * ROW + 2 * j + 1]
        bp = _r32(b2 + t0[row, 1] * (1j * b2))
        e.append(_r32(b1 + c0 * bp))                      # half_ct2
    _dit_fused(e, 5, ct[pos])
    out = np.zeros(2048, complex)
    for m in range(32):
        out[pos + 64 * m] = e[_brev(m, 5)]
    return out


@pytest.mark.parametrize('L,model', [(256, fft256_fused_model), (2048, fft2048_fused_model)])
def test_fused_twiddle_transforms_are_inverse_dfts_to_fp32_rounding(L, model):
    worst = 0.0
    for seed in range(6):
        rs = np.random.RandomState(1000 * L + seed)
        x = _r32(rs.standard_normal(L) + 1j * rs.standard_normal(L))
        ref = np.fft.ifft(x) * L
        worst = max(worst, np.abs(model(x) - ref).max() / np.abs(ref).max())
    # one rounding fewer per twiddled output than multiply-then-add; the radix-16 chain's model reads 1.4e-7 on the same inputs
    assert worst < 3e-7, worst


def test_fused_twiddle_tables():
    # no angle of a lane's set reaches pi/2 except stage 1 (and the lane-pair level of the 2048-point transform), where exactly
    # one lane holds w = i as (2^-40, 2^40): c t = 1 exactly
    t256 = _f256_tables()
    assert t256.shape == (16, 8, 2) and np.isfinite(t256).all()
    assert t256[8, 0, 0] == 2.0 ** -40 and t256[8, 0, 0] * t256[8, 0, 1] == 1.0
    assert np.abs(t256[:, 1:, 1]).max() < 41.0             # tan(2 pi 63 / 256)
    ct, t0 = _f2048_tables()
    assert ct.shape == (64, 16, 2) and t0.shape == (32, 2) and np.isfinite(ct).all() and np.isfinite(t0).all()
    assert ct[32, 0, 0] == 2.0 ** -40 and t0[16, 0] == 2.0 ** -40
    assert np.abs(ct[:, 1:, 1]).max() < 330.0              # tan(2 pi 511 / 2048)
    # c (1 + i t) IS the twiddle: every pair against exp(i angle), relative to fp32 rounding
    for g in range(64):
        us = [16 * g, 8 * g] + [4 * g + 256 * q for q in range(2)] + [2 * g + 128 * q for q in range(4)] + [g + 64 * q for q in range(8)]
        w = ct[g, :, 0] * (1 + 1j * ct[g, :, 1])
        assert np.abs(w - np.exp(2j * np.pi * np.array(us) / 2048)).max() < 2e-7
