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
        if ((threadIdx.x & 63) >= o) incl += v;
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
