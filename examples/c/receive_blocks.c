/* receive_blocks.c -- B consecutive blocks of a stream per device call, through the C ABI alone (no Python): what a host program
 * does that wants the device busy at the reference's block sizes (2^15 ... 2^17 samples: config/base.json:13,33), where one block
 * per call is a few tens of microseconds of device work inside a host-bound loop (Demodulator_process.run, DP:284-338).
 *
 *   gcc -O2 -std=gnu99 examples/c/receive_blocks.c -Iinclude -Lpycusdr_amd -lmfbank -lm \
 *       -Wl,-rpath,$PWD/pycusdr_amd -o examples/c/receive_blocks && examples/c/receive_blocks
 *
 * A continuous 2-FSK stream (the signal of receive_block.c) is cut into B = 4 overlapping blocks of N = 2^14 samples (overlap 2^10)
 * that sit in ONE page-locked window -- block b at b * (N - overlap), neighbours sharing their overlap -- and go through one set of
 * launches (mfb_receive_blocks_begin / _end); with mfb_set_stream_stages the bit lookup and the block-overlap alignment
 * (extractBits / checkSymbolOverlap, DB:1012-1023, 863-988) run on the device too and every block's record carries the bits the
 * caller returns (DB:859).  The program checks that every number of every block equals what mfb_receive_block returns for that
 * block alone, that the kept bits are the bit LUT of the kept symbols, and that consecutive blocks hand over a gap-free bit stream
 * that equals the bits sent; it exits non-zero on any mismatch.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfbank.h"

#define LOG2N 14
#define N (1 << LOG2N)
#define OV 1024
#define STRIDE (N - OV)
#define SPS 16
#define D 33
#define M 2
#define B 4

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

/* struct BlockScalars of csrc/small_kernels.hpp, as far as this program reads it (mfb_record_layout.scalars_bytes long) */
typedef struct {
    double frac, spSym, codeOffset;
    float pick[2], cr[3], spSymF, offsetF;
    int32_t shift, low, high, pick_valid, count, rate_fallback, band[8], band_len[2];
    int32_t a13_status, a13_start, a13_end, a13_nwin, a13_noerr, a13_npost, a13_nend, sync_valid, sync_count[2], a13_prev_npost, pad;
} scalars_t;

int main(void) {
    const double PI2 = 2.0 * M_PI;
    const int total = B * STRIDE + OV, nsym = total / SPS;
    const int carrier = N / 4 + 24;
    unsigned seed = 4242u;

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
    for (int j = 0; j < D; ++j) shifts[j] = N / 4 - 64 + 4 * j;

    mfb_ctx *h = NULL;
    CHECK(mfb_create(&h, 0, LOG2N, D, 0, M, 7, 1, 0));
    CHECK(mfb_set_filters(h, masks, M, N));
    CHECK(mfb_set_shifts(h, shifts, D));

    /* the stream, written straight into the window the device reads: ONE copy per sample, the overlap is shared storage */
    float *win = NULL, *win2 = NULL;
    /* a C caller is not bound by per-block host work: run a batch as two parts on two streams, so that the NEXT batch's search would
     * run beside this batch's matched filters, rate estimate, centres and alignment (same records, bit for bit; include/mfbank.h) */
    CHECK(mfb_set_batch_overlap(h, 1));
    CHECK(mfb_window_buffer(h, 0, B, STRIDE, &win));
    CHECK(mfb_window_buffer(h, 1, B, STRIDE, &win2));
    unsigned char *sent = (unsigned char *)malloc(nsym);
    double phase = 0.0;
    for (int s = 0; s < nsym; ++s) {
        sent[s] = (lcg(&seed) >> 16) & 1;
        const double step = PI2 * carrier / N + (sent[s] ? 1.0 : -1.0) * M_PI / SPS;
        for (int n = 0; n < SPS; ++n) {
            const int i = s * SPS + n;
            win[2 * i] = (float)(cos(phase) + 0.05 * gauss(&seed));
            win[2 * i + 1] = (float)(sin(phase) + 0.05 * gauss(&seed));
            phase += step;
        }
    }

    /* A12 / A13 on the device: filter m decides bit m; 20-symbol alignment windows, 10 errors allowed (config/base.json:18-20) */
    const uint8_t lut[M] = {0, 1};
    mfb_stream_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.overlap_samples = OV;
    sp.overlap_offset = 20;
    sp.match_threshold = 10;
    sp.error_threshold = 1000;
    sp.lut_mode = 1;
    sp.lut_rows = M;
    sp.lut = lut;
    CHECK(mfb_set_stream_stages(h, &sp));
    CHECK(mfb_stream_seed(h, NULL, 0, NULL, 0, NULL, 0));       /* start of a stream: nothing in front of the first block */

    mfb_block_params bp;
    memset(&bp, 0, sizeof(bp));
    bp.mode = MFB_BLOCK_SEARCH;
    bp.input = MFB_INPUT_WINDOW;
    bp.k_offset = (int)(N / (1.1 * SPS));
    bp.k_len = (int)(N / (0.9 * SPS)) - bp.k_offset;
    bp.spsym_min = SPS / 2;
    bp.op = MFB_CENTRES_ABS;
    bp.snr_window = 5;
    bp.max_symbols = N / 2;
    bp.band_capacity = 256;
    CHECK(mfb_receive_blocks_begin(h, &bp, B, 0));
    /* (a real loop fills the other window here while the device works) */
    const size_t cap = (size_t)B * (1 << 17);
    unsigned char *recs = (unsigned char *)malloc(cap);
    mfb_record_layout lay;
    CHECK(mfb_receive_blocks_end_record(h, 0, recs, cap, &lay));
    if (lay.nblocks != B || !lay.stream_stages || (size_t)lay.scalars_bytes != sizeof(scalars_t)) {
        fprintf(stderr, "unexpected record layout (%d blocks, stages %d, %d scalar bytes against %zu)\n", lay.nblocks, lay.stream_stages,
                lay.scalars_bytes, sizeof(scalars_t));
        return 1;
    }

    /* every block once more on its own: mfb_receive_block on the same samples */
    float *in = NULL;
    CHECK(mfb_input_buffer(h, &in));
    int32_t *sym = (int32_t *)malloc(sizeof(int32_t) * (N / 2)), *cen = (int32_t *)malloc(sizeof(int32_t) * (N / 2));
    float *mag = (float *)malloc(sizeof(float) * (N / 2)), *bands = (float *)malloc(sizeof(float) * 2 * 2 * 256);
    unsigned char *stream = (unsigned char *)malloc(nsym + 64);
    int nstream = 0, same = 1, stage_blocks = 0, first_symbol = -1;
    for (int b = 0; b < B; ++b) {
        const unsigned char *r = recs + (size_t)b * lay.record_bytes;
        scalars_t sc;
        memcpy(&sc, r, sizeof(sc));
        memcpy(in, win + 2 * (size_t)b * STRIDE, sizeof(float) * 2 * N);
        bp.input = MFB_INPUT_PINNED;
        mfb_block_result br;
        CHECK(mfb_receive_block(h, &bp, &br, sym, cen, mag, bands));
        const int32_t *bsym = (const int32_t *)(r + lay.off_sym), *bcen = (const int32_t *)(r + lay.off_cen);
        const float *bmag = (const float *)(r + lay.off_mag);
        int ok = sc.pick_valid == br.pick_valid && sc.pick[0] == br.pick[0] && sc.pick[1] == br.pick[1] && sc.shift == br.shift &&
                 sc.cr[0] == br.cr[0] && sc.cr[1] == br.cr[1] && sc.spSym == br.spSym && sc.codeOffset == br.codeOffset && sc.count == br.count;
        for (int i = 0; ok && i < br.count; ++i) ok = bsym[i] == sym[i] && bcen[i] == cen[i] && bmag[i] == mag[i];
        same = same && ok;
        printf("block %d: shift %d, %.4f samples per symbol, %d symbols%s", b, sc.shift, sc.spSym, sc.count, ok ? "" : "  DIFFERENT from the one-block call");
        if (sc.a13_status) {
            /* the kept window: bits = lut[sym], centres mod 256, and the stream goes on where the previous block stopped */
            const unsigned char *kb = r + lay.off_bits, *kc = r + lay.off_centres_u8;
            for (int i = 0; i < sc.a13_nwin; ++i) {
                same = same && kb[i] == lut[bsym[sc.a13_start + i]] && kc[i] == (unsigned char)(bcen[sc.a13_start + i] & 0xff);
                stream[nstream++] = kb[i];
            }
            if (first_symbol < 0) first_symbol = (b * STRIDE + bcen[sc.a13_start] + SPS / 2) / SPS;
            ++stage_blocks;
            printf(", kept symbols [%d, %d) on the device\n", sc.a13_start, sc.a13_end);
        } else {
            printf(", alignment left to the host\n");
        }
    }
    /* the blocks' kept bits, concatenated, are the bits sent: no gap and no repeat at the block edges */
    int errors = 0;
    for (int i = 0; i < nstream; ++i) errors += stream[i] != sent[first_symbol + i];
    printf("%d blocks aligned on the device, %d bits handed over in a row, %d differ from the bits sent\n", stage_blocks, nstream, errors);
    CHECK(mfb_destroy(h));
    free(masks); free(sent); free(recs); free(sym); free(cen); free(mag); free(bands); free(stream);
    (void)win2;
    if (!same || stage_blocks != B || errors != 0 || nstream < (B * STRIDE - OV) / SPS - 4) {
        fprintf(stderr, "FAILED\n");
        return 1;
    }
    printf("ok\n");
    return 0;
}
