// sync_kernels.hpp -- batched sync/preamble correlation of the decoder (reference decoder.py:96-113): exact int32.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

// batched sync-word correlation (decoder.py:96,112): full convolution, exact int32
__global__ void k_sync_corr(const uint8_t *bits, const int8_t *tmpl, int32_t *out, int L, int T) {
    extern __shared__ __attribute__((aligned(16))) int8_t sm[];
    int8_t *st = sm;            // T taps
    int8_t *sb = sm + T;        // blockDim.x + T - 1 bits
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * blockDim.x;
    const int outLen = L + T - 1;
    for (int t = threadIdx.x; t < T; t += blockDim.x) st[t] = tmpl[t];
    const uint8_t *row = bits + (size_t)b * L;
    for (int q = threadIdx.x; q < (int)blockDim.x + T - 1; q += blockDim.x) {
        const int src = i0 - (T - 1) + q;
        sb[q] = (src >= 0 && src < L) ? (int8_t)row[src] : (int8_t)0;
    }
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i < outLen) {
        int acc = 0;
        // out[i] = sum_t tmpl[t]*bits[i-t];  bits[i-t] sits at sb[threadIdx.x + T-1 - t]
        const int8_t *p = sb + threadIdx.x + T - 1;
        for (int t = 0; t < T; ++t) acc += (int)st[t] * (int)p[-t];
        out[(size_t)b * outLen + i] = acc;
    }
}

// Thresholded form of the sync correlation: what decoder.py:101,113 does with np.where on the full
// score array.  Three tiny kernels keep the hits of every stream in ascending position order
// without atomics: (1) hits per 1024-position segment, (2) exclusive scan over the segments of a
// stream, (3) recompute and write (position, score) at segment offset + rank inside the segment.
#define SYNC_SEG 1024
template <bool WRITE>
__global__ void __launch_bounds__(256) k_sync_find(const uint8_t *bits, const int8_t *tmpl, int L, int T, int thr,
                                                   int nseg, int *segcnt, const int *segoff, int max_hits,
                                                   int32_t *hit_idx, int32_t *hit_score) {
    extern __shared__ __attribute__((aligned(16))) int8_t sm[];
    __shared__ int wsum[4];
    int8_t *st = sm;      // T taps
    int8_t *sb = sm + T;  // SYNC_SEG + T - 1 bits
    const int b = blockIdx.y, seg = blockIdx.x;
    const int i0 = seg * SYNC_SEG;
    const int outLen = L + T - 1;
    for (int t = threadIdx.x; t < T; t += 256) st[t] = tmpl[t];
    const uint8_t *row = bits + (size_t)b * L;
    for (int q = threadIdx.x; q < SYNC_SEG + T - 1; q += 256) {
        const int src = i0 - (T - 1) + q;
        sb[q] = (src >= 0 && src < L) ? (int8_t)row[src] : (int8_t)0;
    }
    __syncthreads();
    int sc[4];
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {      // thread owns 4 consecutive positions -> order = (thread, u)
        const int li = threadIdx.x * 4 + u;
        int acc = 0;
        const int8_t *p = sb + li + T - 1;
        for (int t = 0; t < T; ++t) acc += (int)st[t] * (int)p[-t];
        sc[u] = acc;
        cnt += (i0 + li < outLen && acc >= thr) ? 1 : 0;
    }
    // exclusive scan of cnt over the 256 threads: wave prefix (64 lanes) + 4 wave totals
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += v;
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[wid] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wid; ++w) wbase += wsum[w];
    if constexpr (!WRITE) {
        if (threadIdx.x == 255) segcnt[(size_t)b * nseg + seg] = wbase + incl;
    } else {
        int pos = segoff[(size_t)b * nseg + seg] + wbase + incl - cnt;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int li = threadIdx.x * 4 + u;
            if (i0 + li < outLen && sc[u] >= thr) {
                if (pos < max_hits) {
                    hit_idx[(size_t)b * max_hits + pos] = i0 + li;
                    hit_score[(size_t)b * max_hits + pos] = sc[u];
                }
                ++pos;
            }
        }
    }
}

// one thread per stream: exclusive scan of the segment counts, total per stream
__global__ void k_sync_scan(const int *segcnt, int *segoff, int *counts, int B, int nseg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int run = 0;
    for (int s = 0; s < nseg; ++s) {
        segoff[(size_t)b * nseg + s] = run;
        run += segcnt[(size_t)b * nseg + s];
    }
    counts[b] = run;
}

// ---- packed form: 8 bits per byte (np.packbits layout: stream bit i is bit 7 - i%8 of byte i/8) ------------------------
// score[i] = sum_t tmpl[t] * bits[i - t] with taps in {-1, 0, +1} is  popcount(W & P) - popcount(W & Q)  on the window
// W (bit j = bits[i - j]) and the masks P / Q of the +1 / -1 taps: exact, 64 taps per pair of popcounts instead of 64
// multiply-adds.  A thread owns one 64-bit word of the stream (big-endian: oldest bit on top) and walks over its 64 output
// positions, so hits come out in ascending position order per thread; the same three-step scheme as above (count per
// segment, scan, recompute and write) keeps them ordered without atomics.  Results are a FLAT list over all streams
// (stream b's hits start at the exclusive scan of the stream totals): only hits travel back to the host.
#define SYNCP_THREADS 256
#define SYNCP_SEG (SYNCP_THREADS * 64)      // output positions per workgroup
#define SYNCP_MAXK 64                        // 64-bit words per template: T <= 4096

// KT: words per template when small (1 or 2: the window slides in registers, one shift per position), 0 = any K <= SYNCP_MAXK
// (the window words are rebuilt from the staged stream words at every position).
// Pass 1 (WRITE = false) sweeps the positions once, leaves the number of hits of every thread in `tcnt` (one byte) and of every
// segment in `segcnt`; pass 2 scans those and only threads that own a hit sweep again to write it: hits are sparse, so the
// whole search costs about one sweep.
template <bool WRITE, int KT>
__global__ void __launch_bounds__(SYNCP_THREADS) k_sync_packed(const uint8_t *packed, int row_bytes, int L, int T, int K,
                                                              const unsigned long long *masks /* P[K] | Q[K] */, int thr, int nseg,
                                                              int *segcnt, uint8_t *tcnt, const int *segoff, const int *streamoff,
                                                              int max_total, int32_t *hit_idx, int32_t *hit_score) {
    __shared__ unsigned long long sw[SYNCP_THREADS + SYNCP_MAXK + 1];    // stream words q0 - K - 1 ... q0 + 255, big-endian
    __shared__ unsigned long long sp[2 * SYNCP_MAXK];
    __shared__ int wsum[SYNCP_THREADS / 64];
    const int b = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
    const long long q0 = (long long)seg * SYNCP_THREADS;                 // first stream word of this segment
    const size_t tslot = ((size_t)b * nseg + seg) * SYNCP_THREADS + tid;
    int cnt = 0;
    if constexpr (WRITE) {
        cnt = tcnt[tslot];
        if (__syncthreads_or(cnt) == 0) return;                          // no hit in this segment: nothing to stage
    }
    const uint8_t *row = packed + (size_t)b * row_bytes;
    const bool aligned = (reinterpret_cast<unsigned long long>(row) & 7ull) == 0;
    const int nwords = SYNCP_THREADS + K + 1;
    for (int w = tid; w < nwords; w += SYNCP_THREADS) {
        const long long q = q0 - K - 1 + w;                              // stream word index (bits 64q ... 64q + 63)
        unsigned long long v = 0;
        if (q >= 0) {
            const long long byte0 = q * 8;
            if (aligned && byte0 + 8 <= row_bytes) {          // one 8-byte load, byte-swapped: the oldest bit on top
                v = __builtin_bswap64(*reinterpret_cast<const unsigned long long *>(row + byte0));
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long long by = byte0 + u;
                    const unsigned long long x = by < row_bytes ? row[by] : 0;
                    v = (v << 8) | x;
                }
            }
            // bits past the end of the stream (padding of the last byte, or garbage the caller left there) do not exist
            const long long first = q * 64;
            if (first + 64 > L) v = first >= L ? 0ull : (v & (~0ull << (64 - (L - first))));
        }
        sw[w] = v;
    }
    for (int w = tid; w < 2 * K; w += SYNCP_THREADS) sp[w] = masks[w];
    __syncthreads();
    const int outLen = L + T - 1;
    const long long i0 = (q0 + tid) * 64;                                // first output position of this thread
    const int base = K + 1 + tid;                                        // sw index of word q = q0 + tid
    int pos = 0;
    if constexpr (WRITE) {
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += v;
        }
        const int wid = tid >> 6;
        if ((tid & 63) == 63) wsum[wid] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += wsum[w];
        pos = streamoff[b] + segoff[(size_t)b * nseg + seg] + wbase + incl - cnt;
        if (cnt == 0) return;
        cnt = 0;
    }
    auto visit = [&](int r, int s) {
        const long long i = i0 + r;
        if (i < outLen && s >= thr) {
            if constexpr (WRITE) {
                if (pos < max_total) {
                    hit_idx[pos] = (int32_t)i;
                    hit_score[pos] = s;
                }
                ++pos;
            } else {
                ++cnt;
            }
        }
    };
    if constexpr (KT == 1 || KT == 2) {
        // window word k, bit j = bits[i - 64k - j]; from position i to i + 1 every word shifts up by one and takes the next
        // older word's top bit -- word 0 takes the new stream bit
        const unsigned long long cur = sw[base];
        unsigned long long W0 = sw[base - 1], W1 = KT == 2 ? sw[base - 2] : 0ull;       // the windows of position i0 - 1
        const unsigned long long P0 = sp[0], Q0 = sp[K], P1 = KT == 2 ? sp[1] : 0ull, Q1 = KT == 2 ? sp[K + 1] : 0ull;
        for (int r = 0; r < 64; ++r) {
            if constexpr (KT == 2) W1 = (W1 << 1) | (W0 >> 63);
            W0 = (W0 << 1) | ((cur >> (63 - r)) & 1ull);
            int s = __popcll(W0 & P0) - __popcll(W0 & Q0);
            if constexpr (KT == 2) s += __popcll(W1 & P1) - __popcll(W1 & Q1);
            visit(r, s);
        }
    } else {
        for (int r = 0; r < 64; ++r) {
            // (Z[q-k-1] << (r+1)) | (Z[q-k] >> (63-r))
            int s = 0;
            for (int k = 0; k < K; ++k) {
                const unsigned long long hi = sw[base - k - 1], lo = sw[base - k];
                const unsigned long long W = (r == 63) ? lo : ((hi << (r + 1)) | (lo >> (63 - r)));
                s += __popcll(W & sp[k]) - __popcll(W & sp[K + k]);
            }
            visit(r, s);
        }
    }
    if constexpr (!WRITE) {
        tcnt[tslot] = (uint8_t)cnt;
        int incl = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) incl += __shfl_xor(incl, o, 64);
        if ((tid & 63) == 0) wsum[tid >> 6] = incl;
        __syncthreads();
        if (tid == 0) segcnt[(size_t)b * nseg + seg] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// exclusive scan of the stream totals (one workgroup; B is a few thousand at most): streamoff[b], total in streamoff[B]
__global__ void __launch_bounds__(256) k_sync_stream_scan(const int *counts, int *streamoff, int B) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256;
    int s = 0;
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < B) s += counts[b];
    }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) {
            const int v = part[t];
            part[t] = run;
            run += v;
        }
        streamoff[B] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int k = 0; k < per; ++k) {
        const int b = tid * per + k;
        if (b < B) {
            streamoff[b] = run;
            run += counts[b];
        }
    }
}

// ---- small inputs (the decoder's call: one block's bit stream, a couple of templates) ------------------------------------
// ONE launch for all templates and streams: workgroup (template, stream) walks over the stream's segments in order, so a
// running offset keeps the hits ordered without the count / scan / write passes -- at this size the three passes per
// template were six dependent launches of a few microseconds of work each.
struct SyncSmallArgs {
    int T[16], thr[16], toff[16];
};
__global__ void __launch_bounds__(256) k_sync_small(const uint8_t *bits, const int8_t *tmpls, SyncSmallArgs a, int B, int L, int max_hits,
                                                    int32_t *counts, int32_t *hit_idx, int32_t *hit_score) {
    extern __shared__ __attribute__((aligned(16))) int8_t sm[];
    __shared__ int wsum[4];
    const int t = blockIdx.x, b = blockIdx.y;
    const int T = a.T[t], thr = a.thr[t];
    int8_t *st = sm;      // T taps
    int8_t *sb = sm + T;  // SYNC_SEG + T - 1 bits
    for (int q = threadIdx.x; q < T; q += 256) st[q] = tmpls[a.toff[t] + q];
    const uint8_t *row = bits + (size_t)b * L;
    const int outLen = L + T - 1;
    const int nseg = (outLen + SYNC_SEG - 1) / SYNC_SEG;
    int32_t *oi = hit_idx + ((size_t)t * B + b) * max_hits, *os = hit_score + ((size_t)t * B + b) * max_hits;
    int run = 0;
    for (int seg = 0; seg < nseg; ++seg) {
        const int i0 = seg * SYNC_SEG;
        __syncthreads();                                  // the previous segment's readers are done
        for (int q = threadIdx.x; q < SYNC_SEG + T - 1; q += 256) {
            const int src = i0 - (T - 1) + q;
            sb[q] = (src >= 0 && src < L) ? (int8_t)row[src] : (int8_t)0;
        }
        __syncthreads();
        int sc[4];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int li = threadIdx.x * 4 + u;
            int acc = 0;
            const int8_t *p = sb + li + T - 1;
            for (int q = 0; q < T; ++q) acc += (int)st[q] * (int)p[-q];
            sc[u] = acc;
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
            if (i0 + li < outLen && sc[u] >= thr) {
                if (pos < max_hits) {
                    oi[pos] = i0 + li;
                    os[pos] = sc[u];
                }
                ++pos;
            }
        }
        run += total;
    }
    if (threadIdx.x == 0) counts[(size_t)t * B + b] = run;
}
