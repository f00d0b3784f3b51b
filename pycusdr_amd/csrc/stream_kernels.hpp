// stream_kernels.hpp -- the integer stages behind the symbol decisions, for batches of blocks (mfb_receive_blocks_*):
//   A12  symbol -> bit            bit LUT (extractBits, demodulator_base.py:1012-1023) or NRZ-S transition decode with the 3-D
//                                 LUT symbolLUT[sym][0|1][successors] (extractBitsNRZs, DB:1026-1051)
//   A13  block-overlap alignment  checkSymbolOverlap (DB:863-988): keep the symbols whose centre lies in [ov/2, N - ov/2], repair
//                                 a +-1 symbol slip against the previous block (20-symbol windows, match threshold), uint8 casts
//                                 of the caller's three arrays (DB:859) incl. the trust bytes of quirk Q3 (DB:472,1005-1006)
//   A14  sync / preamble search   np.convolve(stream, template) >= threshold of the decoder (decoder.py:96-113) on the stream the
//                                 decoder stitches when no candidate is stashed: the last numBitsOverlap bits before the block +
//                                 the block's bits (DEC:89-90)
// All integer / byte work: results equal the host code's bit for bit, or the block is flagged and the host does it (every
// irregular case -- a symbol index outside the LUT, a window or a previous tail too short for the comparisons numpy would make,
// a first / last centre that does not exist -- is left to the host path, which reproduces whatever the reference does there).
// The previous block's tail (the bits behind its window, the last offset + 1 bits inside it) is a function of that block's
// DEVICE results alone (the +-1 repair moves a window's start, never its last bits), so every block of a batch is aligned in
// parallel; the first block's predecessor comes from a device-resident carry the previous batch left (double-buffered).
// Kernels: k_stream_align (A12 + A13, one workgroup per block), k_stream_search (A14 with the stream bit-packed in LDS, its
// would-be stash edges and the ring for the next batch: one launch, templates of <= 256 taps in {-1, 0, +1}); k_stream_sync /
// k_stream_ring / k_stream_edges are the byte forms of the same three steps for templates the packing does not take.
#pragma once
#include <stdint.h>

#include "small_kernels.hpp"

#define STREAM_POST_MAX 512      // bits behind a block's window kept in its record (ov/2 / samples per symbol + a few)
#define STREAM_END_MAX 32        // overlapOffset + 1 <= 32
#define STREAM_MAX_HITS 64       // sync hits per (block, template) kept in the record
#define STREAM_MAX_TMPL 2
#define STREAM_NOV_MAX 4096      // numBitsOverlap <= 4096 (CC11xx: 2048)
#define STREAM_EDGE_CANDS 4      // header hits per block whose would-be stash start gets its first T - 1 positions searched
#define STREAM_EDGE_HITS 8
#define STREAM_PACK_TAPS 256    // longest template the packed search takes (bench: 128 / 16 taps, CC11xx: 64 / 32)
#define STREAM_EDGE_BACK 20      // the decoder keeps 20 bits in front of a stashed candidate (decoder.py:258)
struct StreamEdge {              // the first T - 1 positions of the stream that would start at the stash of one header hit
    int a_rel;                   // its start relative to the block's stream without a stash (may be negative)
    int valid;                   // 0: not computed (outside what the device holds, template too long)
    int n[2];
    int idx[2][STREAM_EDGE_HITS], score[2][STREAM_EDGE_HITS];
};

#define A13_DEVICE 1             // a13_status: the record holds the block's final arrays
#define A13_HOST 0               //             the host runs A12 / A13 for this block (irregular case)
#define A13_RAISED 2             //             final arrays too; the alignment's first comparison has operands of unequal length
                                 //             (fewer than `offset` bits behind the previous window: 128 samples per symbol with a 2^10-sample
                                 //             overlap, config/CC11xx.json) -- numpy raises, the reference logs and carries on (DB:965-967)

struct StreamCarry {             // what the next batch's first block needs of this batch's last one
    int valid;                   // 0: unknown (the last block was irregular, or nothing was seeded): the first block goes to the host
    int npost, nend;
    int ring_valid, ring_len;    // A14: the last numBitsOverlap bits of the stream so far
    int pad[3];
    uint8_t post[STREAM_POST_MAX];
    uint8_t end[STREAM_END_MAX];
    uint8_t ring[STREAM_NOV_MAX];
};

struct StreamArgs {
    uint8_t *rec0;               // result records [nb][rec]
    size_t rec;
    size_t off_sym, off_cen, off_mag;        // int32[count] | int32[count] | float32[count] inside a record
    size_t off_bits, off_cenw, off_trust;    // uint8[nwin] each (A13 outputs)
    size_t off_post, off_end, off_hits, off_edges;
    int nb, N, ovw;              // blocks, block length, overlap / 2
    int nsym;                    // entries the symbol / centre / magnitude arrays of a record hold
    int o, thr, err_thr;         // overlapOffset, match threshold, symbol_check_error_threshold
    int mode;                    // 1: bit LUT (uint8[rows]); 2: NRZ-S LUT (int32[rows][2][succ])
    int rows, succ;
    const uint8_t *lut8;
    const int *lut3;
    const StreamCarry *carry_in;
    StreamCarry *carry_out;
    // A14
    int K, nOv, max_hits;
    int T[STREAM_MAX_TMPL], thrs[STREAM_MAX_TMPL], toff[STREAM_MAX_TMPL];
    const int8_t *tmpls;
    // templates of at most STREAM_PACK_TAPS taps in {-1, 0, +1} (every shipped decoder: 2 * bits - 1): bit masks of the +1 / -1
    // taps in WINDOW order (bit j of word k = tap T - 1 - (32 k + j)), for k_stream_search
    int packed;
    uint32_t P[STREAM_MAX_TMPL][STREAM_PACK_TAPS / 32], Q[STREAM_MAX_TMPL][STREAM_PACK_TAPS / 32];
};

// A12 on the fly: dataBits[x] of a block from its symbol index s = sym[x] (and, NRZ-S, its successor n = sym[x + 1]); the LUTs are
// the workgroup's LDS copies.  *irregular = an index is outside the LUT (the host path then raises or wraps as numpy does);
// *mismatch = the transition is impossible (NRZ-S).
DEVI int stream_bit(const StreamArgs &a, const uint8_t *l8, const int *l3, int s, int n, bool *irregular, bool *mismatch) {
    if (s < 0 || s >= a.rows) {
        *irregular = true;
        return 0;
    }
    if (a.mode == 1) return l8[s];
    if (n < 0 || n >= a.rows) {          // the successor indexes the LUT itself one symbol later
        *irregular = true;
        return 0;
    }
    bool one = false, zero = false;
    for (int q = 0; q < a.succ; ++q) {
        one = one || (n == l3[(s * 2 + 0) * a.succ + q]);
        zero = zero || (n == l3[(s * 2 + 1) * a.succ + q]);
    }
    if (!one && !zero) {
        *mismatch = true;
        return 0;                         // SYMBOL_MISMATCHVAL
    }
    return one ? 1 : 0;
}

struct BlockWindow {
    int count, nbits;            // symbols decided; dataBits available (count, or count - 1 for NRZ-S)
    int start, end;              // first centre >= ov/2, first centre > N - ov/2 (count: none)
    int noerr;                   // impossible transitions (NRZ-S)
    bool ok;                     // both exist, every index inside the LUT, window >= o + 2 symbols, end <= nbits
};

#define STREAM_ALIGN_THREADS 1024
#define STREAM_LUT8_MAX 256      // rows of a bit LUT (2^xcorrMaskSize: 8 ... 32 in the shipped protocols)
#define STREAM_LUT3_MAX 2048     // rows * 2 * successors of an NRZ-S LUT (bench_BPSK: 16 * 2 * 4)

// Workgroup of 1024 threads = one block of the batch.  The symbol / centre arrays are swept ONCE per window (16-byte loads,
// four symbols per thread and step; the LUTs sit in LDS), for the block and for its predecessor at the same time; the bits the
// +-1 comparison looks at are staged in LDS by 2 o + 3 threads before one thread compares them.
__global__ void __launch_bounds__(STREAM_ALIGN_THREADS) k_stream_align(StreamArgs a) {
    __shared__ int s_w[2][4];            // [me | previous block][first start, first end, irregular, impossible transitions]
    __shared__ int s_start;
    __shared__ uint8_t s_l8[STREAM_LUT8_MAX];
    __shared__ int s_l3[STREAM_LUT3_MAX];
    __shared__ uint8_t s_near[2 * STREAM_END_MAX + 4];
    __shared__ uint8_t p_post[STREAM_END_MAX + 1], p_end[STREAM_END_MAX];
    __shared__ int p_npost, p_nend, p_known;
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
    uint8_t *rec = a.rec0 + (size_t)b * a.rec;
    const uint8_t *prec = a.rec0 + (size_t)(b > 0 ? b - 1 : 0) * a.rec;
    BlockScalars *sc = reinterpret_cast<BlockScalars *>(rec);
    const int o = a.o;
    if (a.mode == 1)
        for (int q = tid; q < a.rows; q += nth) s_l8[q] = a.lut8[q];
    else
        for (int q = tid; q < a.rows * 2 * a.succ; q += nth) s_l3[q] = a.lut3[q];
    auto counts = [&](const uint8_t *r, BlockWindow &w) {
        // (a count beyond what the record holds -- the rate fallback k* = 0, DB:737-740 -- is the host's business)
        const int n = reinterpret_cast<const BlockScalars *>(r)->count;
        w.count = n <= a.nsym ? n : 0;
        w.nbits = a.mode == 1 ? w.count : w.count - 1;
    };
    BlockWindow me, pw;
    counts(rec, me);
    counts(prec, pw);
    if (tid < 8) s_w[tid >> 2][tid & 3] = (tid & 3) < 2 ? ((tid >> 2) ? pw.count : me.count) : 0;
    __syncthreads();

    auto sweep = [&](const uint8_t *r, const BlockWindow &w, int *slot) {
        const int *sym = reinterpret_cast<const int *>(r + a.off_sym), *cen = reinterpret_cast<const int *>(r + a.off_cen);
        int lstart = w.count, lend = w.count, lbad = 0;
        bool lirr = false;
        for (int g = tid; 4 * g < w.count; g += nth) {
            const int x0 = 4 * g;
            const int4 c4 = *reinterpret_cast<const int4 *>(cen + x0), s4 = *reinterpret_cast<const int4 *>(sym + x0);
            const int nx = (a.mode == 2 && x0 + 4 < w.count) ? sym[x0 + 4] : 0;
            const int cs[4] = {c4.x, c4.y, c4.z, c4.w}, ss[5] = {s4.x, s4.y, s4.z, s4.w, nx};
#pragma unroll
            for (int u = 3; u >= 0; --u) {
                const int x = x0 + u;
                if (x < w.count && cs[u] >= a.ovw) lstart = x < lstart ? x : lstart;
                if (x < w.count && cs[u] > a.N - a.ovw) lend = x < lend ? x : lend;
                if (x < w.nbits) {
                    bool m = false;
                    (void)stream_bit(a, s_l8, s_l3, ss[u], ss[u + 1], &lirr, &m);
                    lbad += m ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            lstart = min(lstart, __shfl_xor(lstart, d, 64));
            lend = min(lend, __shfl_xor(lend, d, 64));
            lbad += __shfl_xor(lbad, d, 64);
        }
        const bool anyirr = __ballot(lirr) != 0ull;
        if (lane == 0) {
            if (lstart < w.count) atomicMin(&slot[0], lstart);
            if (lend < w.count) atomicMin(&slot[1], lend);
            if (anyirr) atomicOr(&slot[2], 1);
            if (lbad) atomicAdd(&slot[3], lbad);
        }
    };
    sweep(rec, me, s_w[0]);
    if (b > 0) sweep(prec, pw, s_w[1]);
    __syncthreads();
    auto finish = [&](BlockWindow &w, const int *slot) {
        w.start = slot[0];
        w.end = slot[1];
        w.noerr = slot[3];
        w.ok = slot[2] == 0 && w.start < w.count && w.end < w.count && w.end - w.start >= o + 2 && w.end <= w.nbits && w.nbits > 0;
    };
    finish(me, s_w[0]);
    finish(pw, s_w[1]);
    const int noerr = me.noerr;
    const int *sym = reinterpret_cast<const int *>(rec + a.off_sym), *cen = reinterpret_cast<const int *>(rec + a.off_cen);
    const uint8_t *magb = rec + a.off_mag;         // trust = the raw bytes of the leading fp32 magnitudes (quirk Q3)
    bool i1 = false, m1 = false;                  // (irregular indices were found by the sweep; a window with one is not ok)
    auto bit = [&](const int *sy, int x) { return stream_bit(a, s_l8, s_l3, sy[x], a.mode == 2 ? sy[x + 1] : 0, &i1, &m1); };

    // ---- the previous block's tail: first o + 1 bits behind its window, last o + 1 bits inside it ----
    if (b > 0) {
        const int *psym = reinterpret_cast<const int *>(prec + a.off_sym);
        if (tid == 0) {
            p_known = pw.ok ? 1 : 0;
            p_npost = pw.ok ? pw.nbits - pw.end : 0;
            p_nend = pw.ok ? o + 1 : 0;
        }
        if (pw.ok) {
            if (tid < o + 1 && tid < pw.nbits - pw.end) p_post[tid] = (uint8_t)bit(psym, pw.end + tid);
            if (tid >= 64 && tid - 64 < o + 1) p_end[tid - 64] = (uint8_t)bit(psym, pw.end - (o + 1) + tid - 64);
        }
    } else {
        const StreamCarry *c = a.carry_in;
        if (tid == 0) {
            p_known = c->valid;
            p_npost = c->npost;
            p_nend = c->nend;
        }
        if (tid < o + 1 && tid < c->npost) p_post[tid] = c->post[tid];
        if (tid >= 64 && tid - 64 < c->nend && tid - 64 < STREAM_END_MAX) p_end[tid - 64] = c->end[tid - 64];
    }
    // ---- the bits the comparison looks at: dataBits[start - o - 1 ... start + o + 1] ----
    if (me.ok && tid >= 128 && tid - 128 < 2 * o + 3) {
        const int x = me.start - o - 1 + tid - 128;
        s_near[tid - 128] = (x >= 0 && x < me.nbits) ? (uint8_t)bit(sym, x) : (uint8_t)0;
    }
    __syncthreads();

    // ---- alignment (one thread; a few dozen byte compares in LDS) ----
    if (tid == 0) {
        int status = A13_DEVICE, start = me.start;
        // regular case only: everything numpy would slice exists at full length
        const bool have_prev = p_npost > 0;
        if (!me.ok || !p_known) status = A13_HOST;
        else if (noerr > a.err_thr) {
            // "pass": no alignment (DB:925)
        } else if (have_prev && min(o, p_npost) != min(o, me.end - me.start)) {
            status = A13_RAISED;          // prev_post[:o] == win[:o] cannot be formed: exception, nothing is adjusted
        } else if (have_prev) {
            if (p_npost < o + 1 || p_nend != o + 1 || me.start < o + 1 || me.count - me.end > STREAM_POST_MAX) status = A13_HOST;
            else {
                // win[i] = near[o + 1 + i]; pre[-o:][i] = near[1 + i]; pre[-o-1:-1][i] = near[i]; prev_end[-k] = p_end[o + 1 - k]
                bool all_pre = true, all_pos = true;
                int m_pre[3] = {0, 0, 0}, m_pos[3] = {0, 0, 0};
                for (int i = 0; i < o; ++i) {
                    const int w0 = s_near[o + 1 + i], w1 = s_near[o + 2 + i];
                    all_pre = all_pre && (p_post[i] == w0);
                    m_pre[0] += p_post[i] == w0;
                    m_pre[1] += p_post[i] == w1;
                    m_pre[2] += p_post[i + 1] == w0;
                    const int q0 = s_near[1 + i], q1 = s_near[i];
                    const int e0 = p_end[1 + i], e1 = p_end[i];        // prev_end[-o:][i], prev_end[-o-1:-1][i]
                    all_pos = all_pos && (e0 == q0);
                    m_pos[0] += e0 == q0;
                    m_pos[1] += e1 == q0;
                    m_pos[2] += e0 == q1;
                }
                if (!(all_pre || all_pos)) {
                    const int mx_pre = max(m_pre[0], max(m_pre[1], m_pre[2])), mx_pos = max(m_pos[0], max(m_pos[1], m_pos[2]));
                    if (a.thr < m_pre[1] && m_pre[1] == mx_pre) {
                        if (a.thr < m_pos[1] && m_pos[1] == mx_pos) start += 1;
                    } else if (a.thr < m_pre[2] && m_pre[2] == mx_pre) {
                        if (a.thr < m_pos[2] && m_pos[2] == mx_pos) start -= 1;
                    }
                }
            }
        }
        if (status != A13_HOST && me.nbits - me.end > STREAM_POST_MAX) status = A13_HOST;
        s_start = start;
        sc->a13_status = status;
        sc->a13_start = start;
        sc->a13_end = me.end;
        sc->a13_nwin = status != A13_HOST ? me.end - start : 0;
        sc->a13_noerr = noerr;
        sc->a13_npost = status != A13_HOST ? me.nbits - me.end : 0;
        sc->a13_nend = status != A13_HOST ? o + 1 : 0;
        sc->a13_prev_npost = p_npost;
        sc->sync_valid = 0;              // (the search kernel, when it runs, says otherwise)
        sc->sync_count[0] = sc->sync_count[1] = 0;
        p_known = status != A13_HOST;    // (reused: does the record get the block's arrays?)
    }
    __syncthreads();
    const bool dev = p_known != 0;
    const int start = s_start;
    // ---- the caller's three arrays, the block's own tail ----
    if (dev) {
        uint8_t *ob = rec + a.off_bits, *oc = rec + a.off_cenw, *ot = rec + a.off_trust, *op = rec + a.off_post, *oe = rec + a.off_end;
        for (int x = start + tid; x < me.end; x += nth) {
            ob[x - start] = (uint8_t)bit(sym, x);
            oc[x - start] = (uint8_t)(cen[x] & 0xff);
            ot[x - start] = magb[x];
        }
        for (int x = me.end + tid; x < me.nbits; x += nth) op[x - me.end] = (uint8_t)bit(sym, x);
        if (tid < o + 1) oe[tid] = (uint8_t)bit(sym, me.end - (o + 1) + tid);
    }
    // ---- the carry for the next batch's first block ----
    if (b == a.nb - 1) {
        StreamCarry *c = a.carry_out;
        if (tid == 0) {
            c->valid = dev ? 1 : 0;
            c->npost = dev ? me.nbits - me.end : 0;
            c->nend = dev ? o + 1 : 0;
        }
        if (dev) {
            for (int x = me.end + tid; x < me.nbits; x += nth) c->post[x - me.end] = (uint8_t)bit(sym, x);
            if (tid < o + 1) c->end[tid] = (uint8_t)bit(sym, me.end - (o + 1) + tid);
        }
    }
}

// ---- A14 on the windows without a stash: the byte kernels (templates of > 256 taps or other tap values) -------------------------
// V = ring (the numBitsOverlap bits before the batch) ++ bitsWin_0 ++ bitsWin_1 ++ ...; block b's stream is
// V[cum_b : cum_b + nOv + nwin_b] (cum_b = kept bits of the blocks before it).  Workgroup (template, block) walks over its stream
// in segments with a running offset, so hits come out ordered without atomics (as k_sync_small does for the decoder's own call).
// Valid only while every block up to b was aligned on the device and the ring is known; sync_valid says so per block.
DEVI int stream_v(const StreamArgs &a, int p, const int *cum) {           // V[p]
    if (p < a.nOv) return a.carry_in->ring[p];
    p -= a.nOv;
    int i = 0;
    while (i + 1 < a.nb && p >= cum[i + 1]) ++i;
    return (a.rec0 + (size_t)i * a.rec + a.off_bits)[p - cum[i]];
}

__global__ void __launch_bounds__(256) k_stream_sync(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) int8_t sm[];
    __shared__ int wsum[4];
    __shared__ int cum[65];
    __shared__ int s_valid;
    const int t = blockIdx.x, b = blockIdx.y;
    uint8_t *rec = a.rec0 + (size_t)b * a.rec;
    BlockScalars *sc = reinterpret_cast<BlockScalars *>(rec);
    if (threadIdx.x == 0) {
        int run = 0, ok = a.carry_in->ring_valid && a.carry_in->ring_len == a.nOv && a.nb <= 64;
        for (int i = 0; i < a.nb && i < 64; ++i) {
            cum[i] = run;
            const BlockScalars *s = reinterpret_cast<const BlockScalars *>(a.rec0 + (size_t)i * a.rec);
            if (i <= b) ok = ok && s->a13_status != A13_HOST;
            run += s->a13_nwin;
        }
        cum[a.nb < 64 ? a.nb : 64] = run;
        s_valid = ok;
    }
    __syncthreads();
    const int T = a.T[t], thr = a.thrs[t];
    int32_t *oi = reinterpret_cast<int32_t *>(rec + a.off_hits) + (size_t)t * 2 * a.max_hits, *os = oi + a.max_hits;
    if (!s_valid) {
        if (threadIdx.x == 0) {
            sc->sync_valid = 0;
            sc->sync_count[t] = 0;
        }
        return;
    }
    int8_t *st = sm, *sb = sm + T;
    for (int q = threadIdx.x; q < T; q += 256) st[q] = a.tmpls[a.toff[t] + q];
    const int L = a.nOv + sc->a13_nwin, base = cum[b];
    const int outLen = L + T - 1;
    const int nseg = (outLen + SYNC_SEG - 1) / SYNC_SEG;
    int run = 0;
    for (int seg = 0; seg < nseg; ++seg) {
        const int i0 = seg * SYNC_SEG;
        __syncthreads();
        for (int q = threadIdx.x; q < SYNC_SEG + T - 1; q += 256) {
            const int src = i0 - (T - 1) + q;
            sb[q] = (src >= 0 && src < L) ? (int8_t)stream_v(a, base + src, cum) : (int8_t)0;
        }
        __syncthreads();
        int scv[4];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int li = threadIdx.x * 4 + u;
            int acc = 0;
            const int8_t *p = sb + li + T - 1;
            for (int q = 0; q < T; ++q) acc += (int)st[q] * (int)p[-q];
            scv[u] = acc;
            cnt += (i0 + li < outLen && acc >= thr) ? 1 : 0;
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if ((int)(threadIdx.x & 63) >= o) incl += v;
        }
        const int wid = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 63) wsum[wid] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wid) wbase += wsum[w];
            total += wsum[w];
        }
        int pos = run + wbase + incl - cnt;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int li = threadIdx.x * 4 + u;
            if (i0 + li < outLen && scv[u] >= thr) {
                if (pos < a.max_hits) {
                    oi[pos] = i0 + li;
                    os[pos] = scv[u];
                }
                ++pos;
            }
        }
        run += total;
    }
    if (threadIdx.x == 0) {
        sc->sync_valid = 1;
        sc->sync_count[t] = run;
    }
}

// the ring for the next batch: the last numBitsOverlap bits of V (one workgroup)
__global__ void __launch_bounds__(256) k_stream_ring(StreamArgs a) {
    __shared__ int cum[65];
    __shared__ int s_ok, s_total;
    if (threadIdx.x == 0) {
        int run = 0, ok = a.carry_in->ring_valid && a.carry_in->ring_len == a.nOv && a.nb <= 64;
        for (int i = 0; i < a.nb && i < 64; ++i) {
            cum[i] = run;
            const BlockScalars *s = reinterpret_cast<const BlockScalars *>(a.rec0 + (size_t)i * a.rec);
            ok = ok && s->a13_status != A13_HOST;
            run += s->a13_nwin;
        }
        cum[a.nb < 64 ? a.nb : 64] = run;
        s_ok = ok;
        s_total = run;
        a.carry_out->ring_valid = ok;
        a.carry_out->ring_len = a.nOv;
    }
    __syncthreads();
    if (!s_ok) return;
    for (int q = threadIdx.x; q < a.nOv; q += blockDim.x) a.carry_out->ring[q] = (uint8_t)stream_v(a, s_total + q, cum);
}

// A FIXED-mode decoder that finds a header whose packet is not complete yet restarts its next stream 20 bits in front of that
// header (DEC:254-263).  The first T - 1 positions of such a stream are full-convolution positions whose window hangs over the
// stream's start -- they belong to no other window and would cost the decoder a search of their own, once per packet.  Every
// header hit of a block's stream is a possible restart: its T - 1 leading positions are searched here, for both templates.
// Workgroup = block of the batch; hits of at most STREAM_EDGE_CANDS header hits, in order.
__global__ void __launch_bounds__(256) k_stream_edges(StreamArgs a) {
    __shared__ int cum[65];
    __shared__ int s_flag[2][256];
    const int b = blockIdx.x;
    uint8_t *rec = a.rec0 + (size_t)b * a.rec;
    const BlockScalars *sc = reinterpret_cast<const BlockScalars *>(rec);
    StreamEdge *out = reinterpret_cast<StreamEdge *>(rec + a.off_edges);
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < a.nb && i < 64; ++i) {
            cum[i] = run;
            run += reinterpret_cast<const BlockScalars *>(a.rec0 + (size_t)i * a.rec)->a13_nwin;
        }
        cum[a.nb < 64 ? a.nb : 64] = run;
    }
    __syncthreads();
    const bool usable = sc->sync_valid && a.K == 2 && a.T[0] <= 256 && a.T[1] <= 256;
    const int nh = usable ? min(sc->sync_count[0], a.max_hits) : 0;
    const int32_t *hidx = reinterpret_cast<const int32_t *>(rec + a.off_hits);
    const int vend = a.nOv + cum[b] + sc->a13_nwin;              // end of block b's stream in V
    __shared__ int8_t lead[256];          // the first Tmax - 1 bits of the would-be stream
    __shared__ int8_t taps[2][256];
    if (usable)
        for (int k = 0; k < 2; ++k)
            if ((int)threadIdx.x < a.T[k]) taps[k][threadIdx.x] = a.tmpls[a.toff[k] + threadIdx.x];
    for (int c = 0; c < STREAM_EDGE_CANDS; ++c) {
        StreamEdge *e = out + c;
        const bool have = c < nh;
        if (!have) {                     // (uniform over the workgroup)
            if (threadIdx.x == 0) e->valid = 0;
            continue;
        }
        const int a_rel = have ? hidx[c] - a.T[0] + 1 - STREAM_EDGE_BACK : 0;
        const int av = cum[b] + a_rel;                           // start in V
        const int Tm = max(a.T[0], a.T[1]);
        const bool ok = have && av >= 0 && av + Tm - 1 <= vend;
        __syncthreads();
        if (ok && (int)threadIdx.x < Tm - 1) lead[threadIdx.x] = (int8_t)stream_v(a, av + (int)threadIdx.x, cum);
        __syncthreads();
        for (int k = 0; k < 2; ++k) {
            const int i = threadIdx.x;
            int score = 0;
            bool hit = false;
            if (ok && i < a.T[k] - 1) {
                const int8_t *tp = taps[k];
                for (int q = 0; q <= i; ++q) score += (int)tp[q] * (int)lead[i - q];
                hit = score >= a.thrs[k];
            }
            // ordered compaction: position = hits in the lower lanes of the wave + hits of the lower waves
            const unsigned long long bal = __ballot(hit);
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            if (lane == 0) s_flag[k][wv] = __popcll(bal);
            __syncthreads();
            int pos = __popcll(bal & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv; ++w) pos += s_flag[k][w];
            if (hit && pos < STREAM_EDGE_HITS) {
                e->idx[k][pos] = i;
                e->score[k][pos] = score;
            }
            if (threadIdx.x == 0) e->n[k] = s_flag[k][0] + s_flag[k][1] + s_flag[k][2] + s_flag[k][3];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            e->a_rel = a_rel;
            e->valid = ok ? 1 : 0;
            if (e->n[0] > STREAM_EDGE_HITS || e->n[1] > STREAM_EDGE_HITS) e->valid = 0;
        }
    }
}

// ---- A14, packed: search, would-be stash edges and ring of a block in ONE workgroup -------------------------------------------
// The block's stream (with the T0 - 1 + 20 bits in front of it that a would-be stash may start in) is packed into LDS, one bit per
// bit, by wave ballots; a score is then popcount(window & P) - popcount(window & Q) over ceil(T / 32) words (exact for taps in
// {-1, 0, +1}) instead of T multiply-adds on bytes.  Every thread owns a contiguous run of positions: one count pass, one
// workgroup scan, one write pass -- hits in order, no atomics, no segment loop.  Same records as k_stream_sync + k_stream_ring +
// k_stream_edges (which stay for templates the packing does not take).
// position x of block b's stream (x < 0: the bits in front of it): the block's own kept bits behind the first numBitsOverlap
// positions, else the tail of the blocks before it (normally of block b - 1 alone) or the ring
DEVI int stream_at(const StreamArgs &a, int b, int base, int x, const int *cum) {
    if (x >= a.nOv) return (a.rec0 + (size_t)b * a.rec + a.off_bits)[x - a.nOv];
    int p = base + x;
    if (p < a.nOv) return a.carry_in->ring[p];
    p -= a.nOv;
    int i = b - 1;                        // (p < cum[b]: there is a block in front)
    while (i > 0 && p < cum[i]) --i;
    return (a.rec0 + (size_t)i * a.rec + a.off_bits)[p - cum[i]];
}

DEVI int packed_score(const uint32_t *zw, int s, const uint32_t *P, const uint32_t *Q, int K, int valid_from) {
    const int w0 = s >> 5, sh = s & 31;
    int plus = 0, minus = 0;
    uint32_t lo = zw[w0];
    for (int k = 0; k < K; ++k) {
        const uint32_t hi = zw[w0 + k + 1];
        uint32_t w = __builtin_amdgcn_alignbit(hi, lo, sh);          // bits s + 32 k ... s + 32 k + 31 of the packed stream
        const int m = valid_from - 32 * k;            // window elements below valid_from lie in front of the stream's first bit: absent
        if (m > 0) w &= m >= 32 ? 0u : (~0u << m);
        plus += __popc(w & P[k]);
        minus += __popc(w & Q[k]);
        lo = hi;
    }
    return plus - minus;
}

#define STREAM_SEARCH_THREADS 1024
__global__ void __launch_bounds__(STREAM_SEARCH_THREADS) k_stream_search(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t zw[];
    __shared__ int cum[65];
    __shared__ int s_valid, s_nh, s_h0[STREAM_EDGE_CANDS];
    __shared__ int wsum[STREAM_SEARCH_THREADS / 64];
    __shared__ uint32_t s_P[STREAM_MAX_TMPL][STREAM_PACK_TAPS / 32], s_Q[STREAM_MAX_TMPL][STREAM_PACK_TAPS / 32];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nth = blockDim.x, nwv = nth >> 6;
    if (tid < STREAM_MAX_TMPL * (STREAM_PACK_TAPS / 32)) {
        s_P[tid / (STREAM_PACK_TAPS / 32)][tid % (STREAM_PACK_TAPS / 32)] = a.P[tid / (STREAM_PACK_TAPS / 32)][tid % (STREAM_PACK_TAPS / 32)];
        s_Q[tid / (STREAM_PACK_TAPS / 32)][tid % (STREAM_PACK_TAPS / 32)] = a.Q[tid / (STREAM_PACK_TAPS / 32)][tid % (STREAM_PACK_TAPS / 32)];
    }
    uint8_t *rec = a.rec0 + (size_t)b * a.rec;
    BlockScalars *sc = reinterpret_cast<BlockScalars *>(rec);
    StreamEdge *edges = reinterpret_cast<StreamEdge *>(rec + a.off_edges);
    // kept bits in front of every block (64 loads side by side, one short prefix in LDS), and: was every block up to this one
    // aligned on the device?
    if (tid < 64) {
        int nw = 0, host = 0;
        if (tid < a.nb) {
            const BlockScalars *s = reinterpret_cast<const BlockScalars *>(a.rec0 + (size_t)tid * a.rec);
            nw = s->a13_nwin;
            host = (tid <= b && s->a13_status == A13_HOST) ? 1 : 0;
        }
        int incl = nw;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        cum[tid] = incl - nw;
        if (tid == 63) cum[64] = incl;
        const bool anyhost = __ballot(host != 0) != 0ull;
        if (tid == 0) s_valid = a.carry_in->ring_valid && a.carry_in->ring_len == a.nOv && a.nb <= 64 && !anyhost;
    }
    __syncthreads();
    if (!s_valid) {
        if (tid == 0) {
            sc->sync_valid = 0;
            for (int t = 0; t < a.K; ++t) sc->sync_count[t] = 0;
            if (b == a.nb - 1) {
                a.carry_out->ring_valid = 0;
                a.carry_out->ring_len = a.nOv;
            }
        }
        if (tid < STREAM_EDGE_CANDS) edges[tid].valid = 0;
        return;
    }
    const int nwin = sc->a13_nwin, L = a.nOv + nwin, base = cum[b];
    const int back = a.T[0] - 1 + STREAM_EDGE_BACK;
    const int exb = base < back ? base : back;                   // bits in front of the stream that exist in V
    const int org = STREAM_PACK_TAPS + exb;                      // packed index of the stream's position 0
    const int nwords = ((org + L + 31) >> 5) + STREAM_PACK_TAPS / 32 + 2;
    for (int g0 = wv; 2 * g0 < nwords; g0 += nwv * 4) {            // four loads in flight per lane before the first ballot
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = 64 * (g0 + nwv * u) + lane - org;
            v[u] = (x >= -exb && x < L) ? stream_at(a, b, base, x, cum) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + nwv * u;
            const unsigned long long bal = __ballot(v[u] != 0);
            if (lane == 0 && 2 * g < nwords) {
                zw[2 * g] = (uint32_t)bal;
                zw[2 * g + 1] = (uint32_t)(bal >> 32);
            }
        }
    }
    __syncthreads();
    for (int t = 0; t < a.K; ++t) {
        const int T = a.T[t], thr = a.thrs[t], K = (T + 31) >> 5;
        int32_t *oi = reinterpret_cast<int32_t *>(rec + a.off_hits) + (size_t)t * 2 * a.max_hits, *os = oi + a.max_hits;
        const int outLen = L + T - 1, ppt = (outLen + nth - 1) / nth;
        const int i0 = tid * ppt, i1 = min(i0 + ppt, outLen);
        int cnt = 0;
        for (int i = i0; i < i1; ++i) cnt += packed_score(zw, org + i - (T - 1), s_P[t], s_Q[t], K, T - 1 - i) >= thr ? 1 : 0;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        __syncthreads();                                          // (wsum of the previous template has been read)
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int pos = incl - cnt, total = 0;
        for (int w = 0; w < nwv; ++w) {
            if (w < wv) pos += wsum[w];
            total += wsum[w];
        }
        if (cnt)
            for (int i = i0; i < i1; ++i) {
                const int v = packed_score(zw, org + i - (T - 1), s_P[t], s_Q[t], K, T - 1 - i);
                if (v >= thr) {
                    if (pos < a.max_hits) {
                        oi[pos] = i;
                        os[pos] = v;
                    }
                    if (t == 0 && pos < STREAM_EDGE_CANDS) s_h0[pos] = i;
                    ++pos;
                }
            }
        if (tid == 0) {
            sc->sync_valid = 1;
            sc->sync_count[t] = total;
        }
        if (t == 0 && tid == 0) s_nh = total;                         // header hits, for the edges below
        __syncthreads();
    }
    // ---- the leading T - 1 positions of the streams a FIXED-mode decoder would restart at (see k_stream_edges) ----
    const bool usable = a.K == 2;
    const int nh = usable ? min(s_nh, a.max_hits) : 0;
    const int Tm = max(a.T[0], a.K == 2 ? a.T[1] : 0);
    for (int c = 0; c < STREAM_EDGE_CANDS; ++c) {
        StreamEdge *e = edges + c;
        if (c >= nh) {                   // (uniform over the workgroup)
            if (tid == 0) e->valid = 0;
            continue;
        }
        const int a_rel = s_h0[c] - a.T[0] + 1 - STREAM_EDGE_BACK;
        const bool ok = base + a_rel >= 0 && a_rel + Tm - 1 <= L;
        int n[2];
        for (int k = 0; k < 2; ++k) {
            const int T = a.T[k];
            int score = 0;
            bool hit = false;
            if (ok && tid < T - 1) {
                // position tid of the stream that starts at a_rel: its window ends at a_rel + tid and has tid + 1 elements
                score = packed_score(zw, org + a_rel + tid - (T - 1), s_P[k], s_Q[k], (T + 31) >> 5, T - 1 - tid);
                hit = score >= a.thrs[k];
            }
            const unsigned long long bal = __ballot(hit);            // (threads 0 ... T - 2: the first four waves)
            __syncthreads();
            if (lane == 0) wsum[wv] = __popcll(bal);
            __syncthreads();
            int pos = __popcll(bal & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv && w < 4; ++w) pos += wsum[w];
            if (hit && pos < STREAM_EDGE_HITS) {
                e->idx[k][pos] = tid;
                e->score[k][pos] = score;
            }
            n[k] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        }
        if (tid == 0) {
            e->a_rel = a_rel;
            e->n[0] = n[0];
            e->n[1] = n[1];
            e->valid = ok && n[0] <= STREAM_EDGE_HITS && n[1] <= STREAM_EDGE_HITS ? 1 : 0;
        }
    }
    // ---- the ring for the next batch: the last numBitsOverlap bits of the stream so far = of the last block's stream ----
    if (b == a.nb - 1) {
        if (tid == 0) {
            a.carry_out->ring_valid = 1;
            a.carry_out->ring_len = a.nOv;
        }
        for (int q = tid; q < a.nOv; q += nth) {
            const int idx = org + nwin + q;
            a.carry_out->ring[q] = (uint8_t)((zw[idx >> 5] >> (idx & 31)) & 1u);
        }
    }
}
