/* receive_block.c -- the hot path through the C ABI alone, no Python: what a host program that binds
 * libmfbank.so directly (cgo, JNI, a C daemon ...) does per block.
 *
 *   gcc -O2 -std=gnu99 examples/c/receive_block.c -Iinclude -Lpycusdr_amd -lmfbank -lm \
 *       -Wl,-rpath,$PWD/pycusdr_amd -o examples/c/receive_block && examples/c/receive_block
 *
 * A 2-FSK stream (16 samples per symbol, tones at -/+ half a cycle per symbol around a carrier near fs/4) is
 * received with a two-filter bank (one symbol-long tone each, stored as conj(FFT(template, N)) like the
 * reference's protocol plug-ins do, protocol/FSK2_base.py:17-46): Doppler search over 33 bins, pick,
 * matched filtering at the picked shift, symbol rate and phase, symbol decisions -- first with one call per stage and the
 * reference's host arithmetic in between, then with mfb_receive_block (one call for the whole block).  The program checks
 * the carrier bin, the symbol rate, the bits it sent and that the two forms agree bit for bit; it exits non-zero on any
 * mismatch.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfbank.h"

#define LOG2N 14
#define N (1 << LOG2N)
#define SPS 16
#define D 33
#define M 2

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != MFB_OK) {                                                          \
            fprintf(stderr, "%s -> %s\n", #call, mfb_strerror(rc_));                  \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

static unsigned lcg(unsigned *s) { return *s = *s * 1664525u + 1013904223u; }
static double uniform(unsigned *s) { return (lcg(s) >> 8) * (1.0 / 16777216.0); }
static double gauss(unsigned *s) { return sqrt(-2.0 * log(uniform(s) + 1e-12)) * cos(2.0 * M_PI * uniform(s)); }

int main(void) {
    const double PI2 = 2.0 * M_PI;
    const int nsym = N / SPS;
    const int carrier = N / 4 + 24;                 /* in FFT bins of the block */
    unsigned seed = 12345u;

    /* filter bank: conj(FFT_N(template_m)), template_m[n] = exp(+/- i pi n / SPS), n < SPS */
    float *masks = (float *)malloc(sizeof(float) * 2 * M * N);
    for (int m = 0; m < M; ++m) {
        const double sign = m ? 1.0 : -1.0;
        for (int k = 0; k < N; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < SPS; ++n) {
                const double ph = sign * M_PI * n / SPS - PI2 * (double)k * n / N;
                re += cos(ph);
                im += sin(ph);
            }
            masks[2 * ((size_t)m * N + k)] = (float)re;
            masks[2 * ((size_t)m * N + k) + 1] = (float)(-im);
        }
    }
    int32_t shifts[D];
    for (int j = 0; j < D; ++j) shifts[j] = N / 4 - 64 + 4 * j;      /* candidate carrier bins, 4 apart */

    mfb_ctx *h = NULL;
    CHECK(mfb_create(&h, 0, LOG2N, D, 0, M, 7, 1, 0));
    CHECK(mfb_set_filters(h, masks, M, N));
    CHECK(mfb_set_shifts(h, shifts, D));
    int path, log2L, taps;
    CHECK(mfb_get_search_path(h, &path, &log2L, &taps, NULL, NULL));
    printf("search path %s, %d taps, segments of %d points\n", path == MFB_PATH_SEGMENT ? "segment" : "two-pass", taps, 1 << log2L);

    /* the block: continuous-phase 2-FSK on the carrier, plus noise, written into the pinned input buffer */
    float *in = NULL;
    CHECK(mfb_input_buffer(h, &in));
    unsigned char *bits = (unsigned char *)malloc(nsym);
    double phase = 0.0;
    for (int s = 0; s < nsym; ++s) {
        bits[s] = (lcg(&seed) >> 16) & 1;
        const double step = PI2 * carrier / N + (bits[s] ? 1.0 : -1.0) * M_PI / SPS;
        for (int n = 0; n < SPS; ++n) {
            const int i = s * SPS + n;
            in[2 * i] = (float)(cos(phase) + 0.05 * gauss(&seed));
            in[2 * i + 1] = (float)(sin(phase) + 0.05 * gauss(&seed));
            phase += step;
        }
    }

    /* A3..A7: forward FFT, Doppler search, pick */
    float pick[2];
    CHECK(mfb_upload(h));
    CHECK(mfb_find_carrier(h, pick));
    const int lo = (int)floor(pick[0]), hi = (int)ceil(pick[0]);
    const double frac = pick[0] - lo;
    const int shift = (int)lround(shifts[lo] + frac * (shifts[hi] - shifts[lo]));       /* host interpolation, DB:609-632 */
    printf("picked bin %.3f -> shift %d (sent %d), metric %.2f dB\n", pick[0], shift, carrier, pick[1]);

    /* A9..A10: the filters at the picked shift, symbol rate and phase */
    const int k_off = (int)(N / (1.1 * SPS)), k_len = (int)(N / (0.9 * SPS)) - k_off;
    float cr[3];
    CHECK(mfb_demodulate(h, shift, k_off, k_len, cr));
    const double spSym = (double)N / cr[0];
    double offset = -cr[1] / M_PI * spSym / 2.0;                                      /* DB:733-752 */
    if (offset < 0) offset += spSym - 1;
    printf("samples per symbol %.4f, code phase %.2f\n", spSym, offset);

    /* A11: symbol decisions */
    const int count = (int)(N / spSym);
    int32_t *sym = (int32_t *)malloc(sizeof(int32_t) * count), *cen = (int32_t *)malloc(sizeof(int32_t) * count);
    float *mag = (float *)malloc(sizeof(float) * count);
    CHECK(mfb_find_centres(h, (float)spSym, (float)offset, MFB_CENTRES_ABS, count, sym, cen, mag));
    /* the correlation with a one-symbol template peaks where that symbol starts: a decision belongs to the symbol
     * whose first sample is nearest to its centre */
    int errors = 0, compared = 0;
    for (int i = 2; i < count - 2; ++i) {
        const int s = (cen[i] + SPS / 2) / SPS;
        if (s < 0 || s >= nsym) continue;
        errors += (sym[i] != bits[s]);
        ++compared;
    }
    printf("%d symbol decisions compared, %d errors\n", compared, errors);

    /* The same block once more through the ONE-call form: forward FFT, search, pick, shift interpolation, matched filters,
     * rate/phase arithmetic and symbol decisions as one stream of launches with one synchronisation.  Every number must
     * equal what the five calls above produced (the float64 host arithmetic of DB:609-616 and DB:733-752 runs on the
     * device in the same operations; np.round there, lround here: the interpolated shift is not a tie). */
    mfb_block_params bp;
    mfb_block_result br;
    memset(&bp, 0, sizeof(bp));
    bp.mode = MFB_BLOCK_SEARCH;
    bp.input = MFB_INPUT_PINNED;
    bp.k_offset = k_off;
    bp.k_len = k_len;
    bp.spsym_min = SPS / 2;
    bp.op = MFB_CENTRES_ABS;
    bp.snr_window = 5;
    bp.max_symbols = N / 2;
    bp.band_capacity = 256;
    int32_t *sym2 = (int32_t *)malloc(sizeof(int32_t) * (N / 2)), *cen2 = (int32_t *)malloc(sizeof(int32_t) * (N / 2));
    float *mag2 = (float *)malloc(sizeof(float) * (N / 2)), *bands = (float *)malloc(sizeof(float) * 2 * 2 * 256);
    CHECK(mfb_receive_block(h, &bp, &br, sym2, cen2, mag2, bands));
    int same = br.pick_valid && br.pick[0] == pick[0] && br.pick[1] == pick[1] && br.shift == shift && br.cr[0] == cr[0] &&
               br.cr[1] == cr[1] && br.spSym == spSym && br.codeOffset == offset && br.count == count;
    for (int i = 0; same && i < count; ++i) same = sym2[i] == sym[i] && cen2[i] == cen[i] && mag2[i] == mag[i];
    printf("one call: shift %d, samples per symbol %.4f, %d symbols, SNR windows of %d and %d bins: %s\n", br.shift, br.spSym, br.count,
           br.band_len[0], br.band_len[1], same ? "identical to the stage-by-stage calls" : "DIFFERENT");
    free(sym2); free(cen2); free(mag2); free(bands);
    CHECK(mfb_destroy(h));

    int bad = 0;
    if (abs(shift - carrier) > 2) bad |= 1;
    if (fabs(spSym - SPS) > 0.05) bad |= 2;
    if (compared < nsym - 8 || errors != 0) bad |= 4;
    if (!same) bad |= 8;
    free(masks); free(bits); free(sym); free(cen); free(mag);
    if (bad) {
        fprintf(stderr, "FAILED (%d)\n", bad);
        return 1;
    }
    printf("ok\n");
    return 0;
}
