// Sanitizer driver for the host-only half of libmfbank (pycusdr_amd/csrc/filter_taps.hpp: impulse-response analysis,
// segment spectra, span basis).  Built by tests/test_host_sanitizers.py with g++ -fsanitize=address,undefined and with
// -fsanitize=thread (the analysis runs one thread per filter row); exits non-zero on a wrong answer, the sanitizers abort
// on a bad access.  The GPU pool cannot run sanitizers on device code, so this is the part that can be checked this way.
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../../pycusdr_amd/csrc/filter_taps.hpp"

using taps::cd;

static std::mt19937 rng(12345);
static double gauss() { return std::normal_distribution<double>(0.0, 1.0)(rng); }

// spectra of M filters with T taps starting at sample `start` (circular), complex64 [M][N] = conj(FFT(h)) like the plug-ins
static std::vector<float> bank_of(int M, int N, int T, int start, std::vector<std::vector<cd>> *h_out, bool dup_last) {
    std::vector<float> masks((size_t)M * 2 * N);
    taps::Fft plan(N);
    h_out->assign(M, std::vector<cd>(N, cd(0, 0)));
    for (int m = 0; m < M; ++m) {
        std::vector<cd> &h = (*h_out)[m];
        if (dup_last && m == M - 1 && M > 1) {
            h = (*h_out)[0];
        } else {
            for (int r = 0; r < T; ++r) h[(start + r) & (N - 1)] = cd(gauss(), gauss());
            // the first and last tap never vanish: the window is exactly T long
            h[start & (N - 1)] += cd(3.0, 0.0);
            h[(start + T - 1) & (N - 1)] += cd(0.0, 3.0);
        }
        std::vector<cd> H = h;
        plan.run(H.data(), -1);
        for (int k = 0; k < N; ++k) {
            masks[((size_t)m * N + k) * 2] = (float)H[k].real();
            masks[((size_t)m * N + k) * 2 + 1] = (float)H[k].imag();
        }
    }
    return masks;
}

static int fail(const char *what, int a, int b, int c) {
    fprintf(stderr, "FAILED %s (%d %d %d)\n", what, a, b, c);
    return 1;
}

int main() {
    int checked = 0;
    const int Ns[] = {64, 1024, 4096, 1 << 15};
    for (int N : Ns) {
        for (int M : {1, 2, 5, 8, 17}) {
            for (int Tsel = 0; Tsel < 4; ++Tsel) {
                const int T = Tsel == 0 ? 1 : Tsel == 1 ? 7 : Tsel == 2 ? N / 8 + 1 : N / 2;
                const int start = Tsel == 3 ? N - 5 : (int)(rng() % (unsigned)N);          // windows that wrap round
                std::vector<std::vector<cd>> h;
                const bool dup = (M + Tsel) % 3 == 0 && M > 1;
                std::vector<float> masks = bank_of(M, N, T, start, &h, dup);
                taps::Bank b;
                taps::analyse(masks.data(), M, N, &b);
                if (b.M != M || b.N != N) return fail("geometry", N, M, T);
                // The window must CONTAIN the support (a longer one is slower, never wrong); it is exact for everything but a
                // pure delay, whose periodic spectrum concentrates the complex64 rounding noise on a few samples.
                {
                    const int lead = (start - b.start) & (N - 1);
                    const bool contains = b.T == N || lead + T <= b.T;
                    if (!contains || (T >= 7 && (b.T != T || b.start != (start & (N - 1))))) return fail("window", N, b.T, b.start);
                }
                for (int m = 0; m < M && b.T < N; ++m)
                    for (int r = 0; r < b.T; ++r) {
                        const cd want = h[m][(b.start + r) & (N - 1)];
                        if (std::abs(taps::tap(b, m, r) - want) > 1e-5 * (1.0 + std::abs(want))) return fail("tap", N, m, r);
                    }
                // segment spectra for every segment length that holds the window twice over
                for (int L = 256; L <= 4096 && 4 * L <= N; L <<= 1) {
                    if (2 * (b.T - 1) > L) continue;
                    const int NT = L / 16, Te = L - ((L - b.T + 1) / NT) * NT + 1;
                    std::vector<float> G;
                    taps::segment_spectra(b, L, Te, &G);
                    if (G.size() != (size_t)M * 2 * L) return fail("segment spectra size", N, L, T);
                    // DC bin = (N/L) * sum of the taps; stored at [ii=0][g=0][e=0]
                    cd dc(0, 0);
                    for (int r = 0; r < b.T; ++r) dc += taps::tap(b, 0, r);
                    dc *= (double)N / L;
                    if (std::abs(cd(G[0], G[1]) - dc) > 1e-4 * (1.0 + std::abs(dc))) return fail("segment spectra DC", N, L, T);
                }
                if (T <= 64 && b.T == T) {
                    taps::Bank f;
                    const int R = taps::span_basis(b, &f);
                    const int want = std::min(dup ? M - 1 : M, T);            // T complex taps: at most T independent rows
                    if (R != want) return fail("span rank", N, M, R);
                    // F F^H == C C^H on the diagonal: sum_k |f_k[r]|^2 == sum_m |c_m[r]|^2
                    for (int r = 0; r < T; ++r) {
                        double a = 0, c = 0;
                        for (int k = 0; k < R; ++k) a += std::norm(taps::tap(f, k, r));
                        for (int m = 0; m < M; ++m) c += std::norm(taps::tap(b, m, r));
                        if (fabs(a - c) > 1e-8 * (1.0 + c)) return fail("span energy", N, M, r);
                    }
                }
                ++checked;
            }
        }
    }
    // a bank without a short support (white spectrum) and an all-zero bank
    {
        const int N = 2048, M = 3;
        std::vector<float> masks((size_t)M * 2 * N);
        for (auto &v : masks) v = (float)gauss();
        taps::Bank b;
        taps::analyse(masks.data(), M, N, &b);
        if (b.T < N - 8) return fail("white bank", N, b.T, 0);
        std::fill(masks.begin(), masks.end(), 0.f);
        taps::analyse(masks.data(), M, N, &b);
        if (b.T != 1) return fail("zero bank", N, b.T, 0);
        taps::Bank f;
        if (taps::span_basis(b, &f) != 0) return fail("zero span", N, 0, 0);
    }
    printf("%d banks ok\n", checked);
    return 0;
}
