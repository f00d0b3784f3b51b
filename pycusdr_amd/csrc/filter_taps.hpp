// filter_taps.hpp -- host-side (double precision, init-time only) analysis of the filter bank.
//
// The boundary hands the bank over as spectra H_m[k] = conj(fft(template_m, N)) (complex64 [M][N],
// reference demodulator_base.py:196,246-263).  Every shipped protocol builds them from templates of
// maskSize * samplesPerSym = 48 ... 640 samples (reference protocol/FSK2_base.py:17-46,
// benchmark/bench_GMSK.py:40-61, bench_BPSK.py:47-73), so the impulse response h_m = ifft(H_m) is
// non-zero on a short circular window only.  That is what makes the single-pass overlap-save search
// possible: X[(k+s) mod N] * H_m[k] (reference cuda_kernels.cu:339-373) is the spectrum of the
// circular convolution of x[n] e^{-2 pi i s n / N} with h_m, and a T-tap convolution can be evaluated
// on short segments without ever writing a length-N intermediate.
//
// This file recovers the window and the taps from the spectra: a plain radix-2 double-precision FFT
// (run once per filter row, rows in parallel on host threads), a significance scan, and the L-point
// segment spectra the device kernel multiplies with.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <complex>
#include <thread>
#include <vector>

namespace taps {

typedef std::complex<double> cd;

// in-place radix-2 FFT of length n = 2^k; sign = -1 forward, +1 inverse; unnormalised
struct Fft {
    int n = 0;
    std::vector<cd> w;          // w[j] = e^{+2 pi i j / n}, j < n/2
    std::vector<uint32_t> rev;
    explicit Fft(int n_) : n(n_), w(n_ / 2 > 0 ? n_ / 2 : 1), rev(n_) {
        for (int j = 0; j < n / 2; ++j) {
            const double a = 2.0 * M_PI * (double)j / (double)n;
            w[j] = cd(cos(a), sin(a));
        }
        int lg = 0;
        while ((1 << lg) < n) ++lg;
        rev[0] = 0;
        for (int i = 1; i < n; ++i) rev[i] = (rev[i >> 1] >> 1) | ((uint32_t)(i & 1) << (lg - 1));
    }
    void run(cd *a, int sign) const {
        for (int i = 0; i < n; ++i)
            if ((uint32_t)i < rev[i]) std::swap(a[i], a[rev[i]]);
        for (int len = 2; len <= n; len <<= 1) {
            const int half = len >> 1, step = n / len;
            for (int i = 0; i < n; i += len) {
                for (int j = 0; j < half; ++j) {
                    cd tw = w[(size_t)j * step];
                    if (sign < 0) tw = std::conj(tw);
                    const cd u = a[i + j], v = a[i + j + half] * tw;
                    a[i + j] = u + v;
                    a[i + j + half] = u - v;
                }
            }
        }
    }
};

struct RowTaps {
    int start = 0;            // first sample of the row's own window (circular)
    std::vector<cd> c;        // h[(start + r) mod N], r < c.size(); empty for an all-zero row
};

struct Bank {
    int N = 0, M = 0;
    int start = 0;            // common window [start, start + T) mod N
    int T = 0;                // N when some row has no short support
    std::vector<RowTaps> rows;
};

// A sample belongs to the support of row m when |h_m[n]|^2 > REL * E_m / N (E_m = sum |h_m|^2).  The
// complex64 rounding of H_m puts ~1.2e-15 * E_m / N of noise energy on every sample; 1e-13 is 80x
// that (never crossed by the noise) while everything left outside sums to < 1e-13 * E_m, i.e. < 3.2e-7
// of the output amplitude in the worst case and ~3e-8 (the rounding noise itself) in practice.
// (A spectrum that is periodic in k -- a pure delay by a multiple of a large power of two -- has periodic rounding
// errors, whose energy lands on a few samples and can cross the threshold there: the window then comes out longer than
// the support.  Slower, never wrong: the window only ever grows.  tests/csrc/taps_sanitize.cpp has the case.)
static constexpr double REL = 1e-13;
static constexpr int ROW_TAPS_MAX = 16384;

// largest circular gap between marked samples -> complement window.  Returns false if nothing is marked.
static inline bool window_of(const std::vector<uint8_t> &mark, int N, int *start, int *len) {
    int first = -1, prev = -1, best_gap = -1, best_next = 0;
    for (int n = 0; n < N; ++n) {
        if (!mark[n]) continue;
        if (first < 0) first = n;
        if (prev >= 0 && n - prev > best_gap) {
            best_gap = n - prev;
            best_next = n;
        }
        prev = n;
    }
    if (first < 0) return false;
    const int wrap_gap = first + N - prev;     // from the last marked sample round to the first
    if (wrap_gap > best_gap) {
        best_gap = wrap_gap;
        best_next = first;
    }
    *start = best_next;
    *len = N - best_gap + 1;
    return true;
}

// work(t) for stripes t = 0 ... n - 1, one host thread each beside the caller's (stripe 0).  A thread that cannot be created (a process
// at its thread limit) is not an error: its stripe runs on the caller's thread -- nothing is thrown across the C ABI for it.
template <class Work>
static inline void run_striped(int n, Work &&work) {
    std::vector<std::thread> th;
    int started = 0;
    try {
        th.reserve((size_t)(n > 1 ? n - 1 : 0));
        for (int t = 1; t < n; ++t) {
            th.emplace_back(work, t);
            ++started;
        }
    } catch (...) {
    }
    work(0);
    for (int t = started + 1; t < n; ++t) work(t);
    for (auto &x : th) x.join();
}

static inline void analyse_row(const Fft &plan, const float *H, int N, RowTaps *out, std::vector<cd> &buf,
                               std::vector<uint8_t> &mark) {
    buf.resize(N);
    for (int k = 0; k < N; ++k) buf[k] = cd((double)H[2 * k], (double)H[2 * k + 1]);
    plan.run(buf.data(), +1);
    const double inv = 1.0 / (double)N;
    double E = 0.0;
    for (int n = 0; n < N; ++n) {
        buf[n] *= inv;
        E += std::norm(buf[n]);
    }
    mark.assign(N, 0);
    const double thr = REL * E / (double)N;
    for (int n = 0; n < N; ++n) mark[n] = std::norm(buf[n]) > thr ? 1 : 0;
    int st = 0, len = 0;
    out->c.clear();
    out->start = 0;
    if (!window_of(mark, N, &st, &len)) return;       // all-zero row
    out->start = st;
    if (len > ROW_TAPS_MAX) {                         // no short support: only the length matters
        out->c.resize((size_t)len);                   // (values unused; the caller falls back)
        return;
    }
    out->c.resize((size_t)len);
    for (int r = 0; r < len; ++r) out->c[r] = buf[(st + r) & (N - 1)];
}

// masks: complex64 [M][N] interleaved; N a power of two
static inline void analyse(const float *masks, int M, int N, Bank *bank) {
    bank->N = N;
    bank->M = M;
    bank->rows.assign(M, RowTaps());
    const Fft plan(N);
    unsigned hw = std::thread::hardware_concurrency();
    int nthr = (int)std::min<unsigned>(hw ? hw : 1, 16);
    if (nthr > M) nthr = M;
    if (nthr < 1) nthr = 1;
    auto work = [&](int t) {
        std::vector<cd> buf;
        std::vector<uint8_t> mark;
        for (int m = t; m < M; m += nthr) analyse_row(plan, masks + (size_t)m * 2 * N, N, &bank->rows[m], buf, mark);
    };
    run_striped(nthr, work);
    // common window: union of the rows' windows
    std::vector<uint8_t> cover(N, 0);
    bool any = false, too_long = false;
    for (const auto &r : bank->rows) {
        if (r.c.empty()) continue;
        any = true;
        if ((int)r.c.size() > ROW_TAPS_MAX) too_long = true;
        for (size_t i = 0; i < r.c.size(); ++i) cover[(r.start + (int)i) & (N - 1)] = 1;
    }
    if (!any) {
        bank->start = 0;
        bank->T = 1;
        return;
    }
    int st = 0, len = 0;
    window_of(cover, N, &st, &len);
    bank->start = st;
    bank->T = too_long ? N : len;
}

// tap r of row m in the COMMON window: h_m[(start + r) mod N]; zero outside the row's own window
static inline cd tap(const Bank &b, int m, int r) {
    const RowTaps &row = b.rows[m];
    if (row.c.empty()) return cd(0.0, 0.0);
    const int n = (b.start + r) & (b.N - 1);
    const int off = (n - row.start) & (b.N - 1);
    return off < (int)row.c.size() ? row.c[off] : cd(0.0, 0.0);
}

// L-point segment spectra, complex64 [M][L]:  G_m = (N/L) * FFT_L(c'_m), c'_m[(r - (Te-1)) mod L] = tap r,
// for a layout length Te >= T (the kernel rounds the valid count L - Te + 1 down to whole register slots).
// With this rotation the valid outputs of a segment that starts at sample b0 are i = 0 .. L-Te and
// output i is y[(b0 + i + start + Te - 1) mod N] of the length-N formulation.
// Storage order: [m][ii][g][e] = G_m[g + NT*(2*ii + e)], NT = L/ppl (ppl = points per transform lane: 16, or 32 for the
// wave-local 2048-point kernel) -- the two register slots 2*ii, 2*ii+1 of transform lane g side by side, so that the kernel
// fetches them with one 16-byte load.
static inline void segment_spectra(const Bank &b, int L, int Te, std::vector<float> *out, int ppl = 16) {
    const Fft plan(L);
    out->assign((size_t)b.M * 2 * L, 0.f);
    std::vector<cd> buf(L);
    const double scale = (double)b.N / (double)L;
    for (int m = 0; m < b.M; ++m) {
        std::fill(buf.begin(), buf.end(), cd(0.0, 0.0));
        for (int r = 0; r < b.T; ++r) buf[(r - (Te - 1)) & (L - 1)] = tap(b, m, r);
        plan.run(buf.data(), -1);
        float *o = out->data() + (size_t)m * 2 * L;
        const int NT = L / ppl;
        for (int ii = 0; ii < ppl / 2; ++ii)
            for (int g = 0; g < NT; ++g)
                for (int e = 0; e < 2; ++e) {
                    const cd v = buf[g + NT * (2 * ii + e)] * scale;
                    const size_t pos = ((size_t)(ii * NT + g) * 2 + e) * 2;
                    o[pos] = (float)v.real();
                    o[pos + 1] = (float)v.imag();
                }
    }
}

// The same spectra with the Doppler shift on the FILTER side (seg_kernels.hpp, segf_body): the search needs |y_s[n]|^2 only, and
//     y_s[n] = sum_k h[k] x[n-k] e^{-2 pi i s (n-k) / N} = e^{-2 pi i s n / N} sum_k (h[k] e^{+2 pi i s k / N}) x[n-k],
// so the segment of x can be transformed ONCE for all bins and each bin brings its own spectra of h_s[k] = h[k] e^{+2 pi i s k / N}
// (k counted from the window start: a constant phase drops out of |y|^2).  Storage: [shift][row][ii][g][e], `rows` = the bank
// rows wanted (nullptr: all of them), the slot-pair order of segment_spectra.
// `qout` (optional, with `weights` per row): Q[shift][jj][g][4] = sum over the rows of weights[row] * |spectrum as stored|^2 -- register
// slots 4 jj ... 4 jj + 3 of lane g side by side (segf_body, SUMQ).
static inline void segment_spectra_shifted(const Bank &b, const int *rows, int nrows, const int *shifts, int nshifts, int L, int Te,
                                           std::vector<float> *out, int ppl = 16, std::vector<float> *qout = nullptr,
                                           const double *weights = nullptr) {
    const Fft plan(L);
    out->assign((size_t)nshifts * nrows * 2 * L, 0.f);
    if (qout) qout->assign((size_t)nshifts * L, 0.f);
    const double scale = (double)b.N / (double)L;
    const int NT = L / ppl;
    // one host thread per stripe of bins (as analyse does per filter row): D * MU transforms of L points -- 2048 of 256 points at C2,
    // 2048 of 2048 points for the 384-tap bank at 256 bins
    unsigned hw = std::thread::hardware_concurrency();
    int nthr = (int)std::min<unsigned>(hw ? hw : 1, 16);
    if ((long long)nshifts * nrows * L < (1 << 18)) nthr = 1;
    if (nthr > nshifts) nthr = nshifts;
    if (nthr < 1) nthr = 1;
    auto work = [&](int t) {
        std::vector<cd> buf(L);
        std::vector<double> qacc(qout ? (size_t)L : 0);
        for (int j = t; j < nshifts; j += nthr) {
            const long long s = shifts[j];
            std::fill(qacc.begin(), qacc.end(), 0.0);
            for (int u = 0; u < nrows; ++u) {
                const int m = rows ? rows[u] : u;
                std::fill(buf.begin(), buf.end(), cd(0.0, 0.0));
                for (int r = 0; r < b.T; ++r) {
                    const double ang = 2.0 * M_PI * (double)((s * r) % b.N) / (double)b.N;
                    buf[(r - (Te - 1)) & (L - 1)] = tap(b, m, r) * cd(cos(ang), sin(ang));
                }
                plan.run(buf.data(), -1);
                float *o = out->data() + ((size_t)j * nrows + u) * 2 * L;
                for (int ii = 0; ii < ppl / 2; ++ii)
                    for (int g = 0; g < NT; ++g)
                        for (int e = 0; e < 2; ++e) {
                            const cd v = buf[g + NT * (2 * ii + e)] * scale;
                            const size_t pos = ((size_t)(ii * NT + g) * 2 + e) * 2;
                            o[pos] = (float)v.real();
                            o[pos + 1] = (float)v.imag();
                            if (qout)          // of the values the kernel multiplies with, i.e. after the rounding to float
                                qacc[(size_t)g + (size_t)NT * (2 * ii + e)] +=
                                    (weights ? weights[u] : 1.0) * ((double)o[pos] * (double)o[pos] + (double)o[pos + 1] * (double)o[pos + 1]);
                        }
            }
            if (qout) {
                float *q = qout->data() + (size_t)j * L;
                for (int jj = 0; jj < ppl / 4; ++jj)
                    for (int g = 0; g < NT; ++g)
                        for (int e = 0; e < 4; ++e) q[((size_t)jj * NT + g) * 4 + e] = (float)qacc[(size_t)g + (size_t)NT * (4 * jj + e)];
            }
        }
    };
    run_striped(nthr, work);
}

// ---- span basis (opt-in, SUM_ALL_MASKS search only) ---------------------------------------------------------
// With SUM_ALL_MASKS the search needs  sum_m |y_m[n]|^2 = x_n^H (C C^H) x_n  only, C = [c_1 ... c_M] the taps.
// Any F with F F^H = C C^H gives the same number from rank(C) filters instead of M.  The shipped banks are
// strongly rank-deficient (all 2^k bit patterns of a smooth modulation: GMSK 6 of 8, FSK-2 / CC11xx 4 of 8,
// BPSK 5 of 32).  F = Q L with C = Q R (pivoted modified Gram-Schmidt, Q orthonormal) and R R^H = L L^H
// (Cholesky): F F^H = Q R R^H Q^H = C C^H.  Directions whose residual energy is below SPAN_REL = 1e-10 of the
// largest column are dropped: exact dependencies (the complex64 rounding of the spectra leaves ~1e-15 there) and
// the two directions of the GMSK bank that carry 3e-12 of the energy -- they change doppSum by < 1e-10 relative,
// three orders below the fp32 rounding of the transforms themselves.
static constexpr double SPAN_REL = 1e-10;
static inline int span_basis(const Bank &b, Bank *out) {
    const int M = b.M, T = b.T;
    std::vector<std::vector<cd>> C(M, std::vector<cd>(T));
    double emax = 0.0;
    for (int m = 0; m < M; ++m) {
        double e = 0.0;
        for (int r = 0; r < T; ++r) {
            C[m][r] = tap(b, m, r);
            e += std::norm(C[m][r]);
        }
        emax = std::max(emax, e);
    }
    out->N = b.N;
    out->start = b.start;
    out->T = b.T;
    out->rows.clear();
    if (emax == 0.0) {
        out->M = 0;
        return 0;
    }
    std::vector<std::vector<cd>> Q;                 // orthonormal columns
    std::vector<std::vector<cd>> work = C;          // residuals
    std::vector<uint8_t> used(M, 0);
    for (;;) {
        int best = -1;
        double be = SPAN_REL * emax;
        for (int m = 0; m < M; ++m) {
            if (used[m]) continue;
            double e = 0.0;
            for (int r = 0; r < T; ++r) e += std::norm(work[m][r]);
            if (e > be) {
                be = e;
                best = m;
            }
        }
        if (best < 0) break;
        used[best] = 1;
        std::vector<cd> q = work[best];
        for (int pass = 0; pass < 2; ++pass)          // re-orthogonalise once (twice is enough)
            for (const auto &p : Q) {
                cd d(0.0, 0.0);
                for (int r = 0; r < T; ++r) d += std::conj(p[r]) * q[r];
                for (int r = 0; r < T; ++r) q[r] -= d * p[r];
            }
        double n2 = 0.0;
        for (int r = 0; r < T; ++r) n2 += std::norm(q[r]);
        const double inv = 1.0 / sqrt(n2);
        for (int r = 0; r < T; ++r) q[r] *= inv;
        for (int m = 0; m < M; ++m) {
            if (used[m]) continue;
            cd d(0.0, 0.0);
            for (int r = 0; r < T; ++r) d += std::conj(q[r]) * work[m][r];
            for (int r = 0; r < T; ++r) work[m][r] -= d * q[r];
        }
        Q.push_back(q);
    }
    const int R = (int)Q.size();
    // Rm = Q^H C (R x M), S = Rm Rm^H (R x R), Cholesky S = L L^H
    std::vector<std::vector<cd>> Rm(R, std::vector<cd>(M));
    for (int k = 0; k < R; ++k)
        for (int m = 0; m < M; ++m) {
            cd d(0.0, 0.0);
            for (int r = 0; r < T; ++r) d += std::conj(Q[k][r]) * C[m][r];
            Rm[k][m] = d;
        }
    std::vector<std::vector<cd>> S(R, std::vector<cd>(R)), Lc(R, std::vector<cd>(R, cd(0.0, 0.0)));
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) {
            cd d(0.0, 0.0);
            for (int m = 0; m < M; ++m) d += Rm[i][m] * std::conj(Rm[j][m]);
            S[i][j] = d;
        }
    for (int j = 0; j < R; ++j) {
        double d = S[j][j].real();
        for (int k = 0; k < j; ++k) d -= std::norm(Lc[j][k]);
        const double ljj = sqrt(d > 0.0 ? d : 0.0);
        Lc[j][j] = cd(ljj, 0.0);
        for (int i = j + 1; i < R; ++i) {
            cd v = S[i][j];
            for (int k = 0; k < j; ++k) v -= Lc[i][k] * std::conj(Lc[j][k]);
            Lc[i][j] = ljj > 0.0 ? v / ljj : cd(0.0, 0.0);
        }
    }
    // F = Q L: column k of F = sum_i Q_i * L[i][k]
    out->M = R;
    out->rows.assign(R, RowTaps());
    for (int k = 0; k < R; ++k) {
        out->rows[k].start = b.start;
        out->rows[k].c.assign(T, cd(0.0, 0.0));
        for (int i = k; i < R; ++i)
            for (int r = 0; r < T; ++r) out->rows[k].c[r] += Q[i][r] * Lc[i][k];
    }
    return R;
}

}  // namespace taps
