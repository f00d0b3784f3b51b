// small_kernels.hpp -- the kernels around the transforms: partial-sum finalisation, transpose, Doppler pick,
// envelope, symbol-rate search, symbol centres.  Reference semantics reproduced (file:line under pyCuSDR/):
//   |.|^2 / 2^18 row sums  demodulator/cuda_kernels.cu:421-480      pick        cuda_kernels.cu:502-597
//   envelope               cuda_kernels.cu:191-205                  rate/phase  cuda_kernels.cu:236-320
//   centres                cuda_kernels.cu:78-146
#pragma once
#include <stdint.h>

#include "fft_core.hpp"

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// doppSum[j][m] from the partial sums.  One workgroup per Doppler bin, one wavefront per transformed filter row (up to
// 16 at a time): lane l adds partials l, l+64, ... of the row in four interleaved chains, then a fixed butterfly over the
// 64 lanes -- the order depends on nothing but `parts`, so results are bit-reproducible (no float atomics, unlike
// cuda_kernels.cu:463,474).
// SUM_ALL_MASKS: column 0 gets the sum over masks (added in filter order by one thread), the other columns stay 0
// (cuda_kernels.cu:453-464); else per mask (472-475).
// The partials hold MU <= M rows per bin (filters that are exact copies or exact negatives of an
// earlier filter are transformed once: |.|^2 is identical bit for bit); rep[m] names filter m's row.
#define FIN_MAX_ROWS 64
__global__ void __launch_bounds__(1024) k_finalize(const float *partials, float *dsum, int D, int M, int MU, const int *rep, int parts,
                                                   int sum_all) {
    __shared__ float srow[FIN_MAX_ROWS];
    const int j = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (j >= D) return;
    for (int u = wave; u < MU; u += nw) {
        const float *p = partials + ((size_t)j * MU + u) * parts;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = lane;
        for (; q + 192 < parts; q += 256) {
            a0 += p[q];
            a1 += p[q + 64];
            a2 += p[q + 128];
            a3 += p[q + 192];
        }
        for (; q < parts; q += 64) a0 += p[q];
        float s = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) srow[u] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nrows = rep ? M : MU;       // rep == nullptr: sum every transformed row (span basis, SUM_ALL only)
        if (sum_all) {
            float tot = 0.f;
            for (int m = 0; m < nrows; ++m) tot += srow[rep ? rep[m] : m];
            dsum[j * M] = tot;
            for (int m = 1; m < M; ++m) dsum[j * M + m] = 0.f;
        } else {
            for (int m = 0; m < M; ++m) dsum[j * M + m] = srow[rep ? rep[m] : m];
        }
    }
}

// Z[row][n1][n2] -> out[row][n1 + N1*n2] (natural order), optional conjugation (forward transform).
// 32x32 tiles through LDS: coalesced reads along n2, coalesced writes along n1.
__global__ void __launch_bounds__(256) k_transpose(const cf *Z, cf *out, int N1, int N2, int conj) {
    __shared__ cf tile[32][33];
    const size_t rowoff = (size_t)blockIdx.z * N1 * N2;
    const int n2_0 = blockIdx.x * 32, n1_0 = blockIdx.y * 32;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    for (int r = r0; r < 32; r += 8) {
        if (n1_0 + r < N1 && n2_0 + c < N2) tile[r][c] = Z[rowoff + (size_t)(n1_0 + r) * N2 + n2_0 + c];
    }
    __syncthreads();
    const float sgn = conj ? -1.f : 1.f;
    for (int r = r0; r < 32; r += 8) {
        if (n2_0 + r < N2 && n1_0 + c < N1) {
            const cf z = tile[c][r];
            out[rowoff + (size_t)(n2_0 + r) * N1 + n1_0 + c] = mkc(z.x, sgn * z.y);
        }
    }
}

// findDopplerEst (cuda_kernels.cu:502-597).  The reference scans every column sequentially: a value
// replaces the smaller of the two kept maxima when it is strictly greater (first come wins ties), and
// which of the two slots a maximum sits in -- it matters for the fp32 rounding of the weighted index --
// depends on the whole history.  One wavefront reproduces the outcome of that scan exactly:
//   * closed form (the usual case).  Without a value that EQUALS the running maximum when it arrives, the scan ends
//     with the largest value (first occurrence) and the largest of the rest (first occurrence); every new running
//     maximum ("record") lands in the slot that does not hold the previous one, so the largest value sits in slot
//     (records - 1) mod 2.  Records, first occurrences and the tie test are prefix-maximum scans over the column,
//     staged in LDS: a handful of wave operations instead of one dependent step per rising row.
//   * otherwise (a tie with the running maximum, or a column too long for the staging buffer) the scan itself,
//     skipping what cannot change the state: 64 rows are compared against the current threshold at once (ballot);
//     only lanes that beat it are visited, in row order.
// fp32 evaluation order is pinned with explicit intrinsics (see oracle/mfbank_oracle.py).
#define PICK_LDS 4096
DEVI void pick_body(const float *in, float *res, int num, int offset, int M, int sum_all, float *sIdx, float *sVal, float *scol) {
    const int lane = threadIdx.x;
    const int ncol = sum_all ? 1 : M;
    for (int x = 0; x < ncol; ++x) {
        float mv0 = 0.f, mv1 = 0.f;
        int mi0 = 0, mi1 = 0, cur = 0;
        bool closed = false;
        if (num <= PICK_LDS) {
            __syncthreads();                              // the previous column's readers are done
            for (int i = lane; i < num; i += 64) {
                const float v = in[(size_t)(offset + i) * M + x];
                scol[i] = (v > 0.f) ? v : 0.f;            // zeros, negatives and NaN never enter the scan (strict '>' against >= 0)
            }
            __syncthreads();
            float carry = 0.f, gmax = 0.f;
            int records = 0, gidx = 0;
            bool tie = false;
            for (int r0 = 0; r0 < num; r0 += 64) {
                const int i = r0 + lane;
                const float vv = (i < num) ? scol[i] : 0.f;
                float pm = vv;                            // inclusive prefix maximum over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const float t = __shfl_up(pm, o, 64);
                    if (lane >= o) pm = fmaxf(pm, t);
                }
                float ex = __shfl_up(pm, 1, 64);          // exclusive: everything before this row
                ex = lane ? fmaxf(ex, carry) : carry;
                const unsigned long long rb = __ballot(vv > ex);
                tie = tie || (__ballot(vv == ex && vv > 0.f) != 0ull);
                if (rb) {
                    records += __popcll(rb);
                    const int l = 63 - __builtin_clzll(rb);           // the chunk's last record = the running maximum so far
                    gmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), l));
                    gidx = r0 + l;
                }
                carry = fmaxf(carry, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm), 63)));
            }
            if (!tie) {
                float best = 0.f;                         // largest of the rest, first occurrence
                int bidx = 0;
                for (int i = lane; i < num; i += 64) {
                    const float vv = (i == gidx) ? 0.f : scol[i];
                    if (vv > best) {
                        best = vv;
                        bidx = i;
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bidx, o, 64);
                    if (ov > best || (ov == best && oi < bidx)) {
                        best = ov;
                        bidx = oi;
                    }
                }
                if (records > 0) {
                    const int gi = offset + gidx, bi = best > 0.f ? offset + bidx : 0;
                    if ((records - 1) & 1) {
                        mv1 = gmax, mi1 = gi, mv0 = best, mi0 = bi;
                    } else {
                        mv0 = gmax, mi0 = gi, mv1 = best, mi1 = bi;
                    }
                    cur = (mv0 >= mv1) ? 1 : 0;
                }
                closed = true;
            }
        }
        if (!closed) {
            mv0 = mv1 = 0.f;
            mi0 = mi1 = cur = 0;
            for (int r0 = offset; r0 < num + offset; r0 += 64) {
                const int r = r0 + lane;
                const bool inside = r < num + offset;
                const float v = inside ? in[(size_t)r * M + x] : 0.f;
                unsigned long long todo = ~0ull;
                while (true) {
                    const float thr = cur ? mv1 : mv0;
                    const unsigned long long hit = __ballot(inside && v > thr) & todo;     // strict '>' (NaN never enters)
                    if (!hit) break;
                    const int l = __builtin_ctzll(hit);
                    const float tmp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));   // l is wave-uniform
                    if (cur) {
                        mv1 = tmp;
                        mi1 = r0 + l;
                    } else {
                        mv0 = tmp;
                        mi0 = r0 + l;
                    }
                    cur = (mv0 >= mv1) ? 1 : 0;          // the slot of the smaller value is replaced next
                    todo = (l == 63) ? 0ull : (~0ull << (l + 1));
                }
            }
        }
        const float numr = __fmaf_rn((float)mi0, mv0, __fmul_rn((float)mi1, mv1));
        const float idxL = __fdiv_rn(numr, __fadd_rn(mv0, mv1));
        float valL = __fdiv_rn(numr, (float)(mi0 + mi1));
        if (offset > 0) valL = __fdiv_rn(cur ? mv0 : mv1, in[x]);     // maxVal[(cur + 1) % 2] / noise-bin score
        if (sum_all) {
            if (lane == 0) {
                res[0] = idxL;
                res[1] = 10.f * log10f(valL);
            }
            return;
        }
        if (lane == 0) {
            sIdx[x] = idxL;
            sVal[x] = valL;
        }
    }
    int n = 1;
    while (n < M) n <<= 1;
    __syncthreads();
    if (lane >= M) {
        sIdx[lane] = 0.f;
        sVal[lane] = 0.f;
    }
    __syncthreads();
    for (int step = n >> 1; step >= 1; step >>= 1) {
        float a = 0.f, b = 0.f;
        if (lane < n) {
            a = __fadd_rn(sIdx[lane], sIdx[lane ^ step]);
            b = __fadd_rn(sVal[lane], sVal[lane ^ step]);
        }
        __syncthreads();
        if (lane < n) {
            sIdx[lane] = a;
            sVal[lane] = b;
        }
        __syncthreads();
    }
    if (lane == 0) {
        res[0] = __fdiv_rn(sIdx[0], (float)M);
        res[1] = 10.f * log10f(__fdiv_rn(sVal[0], (float)M));
    }
}
__global__ void __launch_bounds__(64) k_pick(const float *in, float *res, int num, int offset, int M, int sum_all) {
    __shared__ float sIdx[64], sVal[64];
    __shared__ float scol[PICK_LDS];
    pick_body(in, res, num, offset, M, sum_all, sIdx, sVal, scol);
}

// |z|^2 the way nvcc contracts in.x*in.x+in.y*in.y (cuda_kernels.cu:1022-1026): fma(x,x,y*y)
DEVI float abs2c(cf z) { return __fmaf_rn(z.x, z.x, __fmul_rn(z.y, z.y)); }

// sumXCorrBuffMasks (cuda_kernels.cu:191-205)
// (blockIdx.y = block of a batch: its M rows and its envelope follow the previous block's)
__global__ void k_envelope(const cf *xc, float *env, int N, int M, int off) {
    xc += (size_t)blockIdx.y * (size_t)M * (size_t)N;
    env += (size_t)blockIdx.y * (size_t)N;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < N; x += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int m = off; m < M - off; ++m) s = __fadd_rn(s, abs2c(xc[(size_t)m * N + x]));
        env[x] = s;
    }
}

// findCodeRateAndPhase (cuda_kernels.cu:236-320): argmax |P[k]|^2 over [offset, offset+len);
// ties resolve to the lowest k (deterministic).  out = {k, atan2(im,re), |P|^2}
// The reference squares in fp32 (ComplexAbsSquared, CU:1022-1026): for a strong signal in a long block with long filters
// (|P| ~ M N^3 T^2 a^4: amplitude-1 samples, 384 taps, N = 2^20) |P[k]|^2 exceeds the fp32 range, every candidate near the
// peak reads +inf, and which of them the reference's warp butterfly then returns is an artefact of its shuffle pattern.  Here
// the comparison runs on P scaled by a power of two that brings the window's largest component to [1, 2): the scaling is exact,
// so wherever the reference's squares are finite and normal the same k wins with the same ties, and where they overflow the
// true maximum still wins (candidates 2^-63 below the largest flush to zero: they could not win either way).  out[2] stays
// the reference's own fp32 square (+inf when it overflows; the host never reads it, DB:730-752).
DEVI void code_rate_body(const cf *P, float *out, int offset, int len, float *sv, int *si) {
    const int tid = threadIdx.x;
    unsigned emax = 0;             // largest finite |component| of the window, as its fp32 bit pattern (monotonic for >= 0)
    for (int x = tid; x < len; x += blockDim.x) {
        const cf z = P[x + offset];
        const unsigned a = __float_as_uint(fabsf(z.x)), b = __float_as_uint(fabsf(z.y));
        if (a < 0x7f800000u && a > emax) emax = a;
        if (b < 0x7f800000u && b > emax) emax = b;
    }
    si[tid] = (int)emax;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s && (unsigned)si[tid + s] > (unsigned)si[tid]) si[tid] = si[tid + s];
        __syncthreads();
    }
    const int shift = 127 - (int)(((unsigned)si[0] >> 23) & 0xffu);      // 2^shift * largest component in [1, 2) (0 for a zero window)
    __syncthreads();
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int x = tid; x < len; x += blockDim.x) {
        const cf z = P[x + offset];
        const float v = abs2c(mkc(ldexpf(z.x, shift), ldexpf(z.y, shift)));
        if (v > best) {
            best = v;
            bi = x + offset;
        }
    }
    sv[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s) {
            const float ov = sv[tid + s];
            const int oi = si[tid + s];
            if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) {
                sv[tid] = ov;
                si[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int k = (len > 0) ? si[0] : 0;
        const cf z = P[k];
        out[0] = (float)k;
        out[1] = atan2f(z.y, z.x);
        out[2] = (len > 0) ? abs2c(z) : -1.f;
    }
}
__global__ void k_code_rate(const cf *P, float *out, int offset, int len) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    code_rate_body(P, out, offset, len, sv, si);
}
// findCentres (cuda_kernels.cu:78-146); thread = symbol index.
DEVI void centres_body(int *outSym, int *outIdx, float *mag, const cf *sig, float spSym, float offset, int lenSig, int M, int W, int op,
                       int capacity, const int x) {
    if (x >= capacity) return;
    const float half = (float)(W / 2);  // WINDOW_WIDTH/2 is an integer division (demodulator_base.py:407)
    const float base = __fmaf_rn((float)x, spSym, -half);
    int arrayIdx = (int)__fadd_rn(base, offset);
    int maxArrayIdx = arrayIdx + W;
    int offsetComp = (int)offset;
    if (arrayIdx < 0) {
        offsetComp -= arrayIdx;
        arrayIdx = 0;
    }
    if (maxArrayIdx > lenSig) maxArrayIdx = lenSig;
    maxArrayIdx -= arrayIdx;
    int maxIdx = -1, maxCentreIdx = -1;
    if (arrayIdx < lenSig) {
        float maxVal = 0.f;
        for (int m = 0; m < M; ++m) {
            const cf *row = sig + (size_t)m * lenSig + arrayIdx;
            for (int k = 0; k < maxArrayIdx; ++k) {
                const cf z = row[k];
                const float tmp = (op == 0) ? abs2c(z) : (op == 1 ? fabsf(z.x) : fabsf(z.y));
                if (tmp > maxVal) {
                    maxVal = tmp;
                    maxIdx = m;
                    maxCentreIdx = k;
                }
            }
        }
        outSym[x] = maxIdx;
        outIdx[x] = (int)__fadd_rn(__fadd_rn(base, (float)maxCentreIdx), (float)offsetComp);
        mag[x] = maxVal;
    } else {            // a symbol past the end of the block: the reference leaves its buffers untouched there
        outSym[x] = INT32_MIN;
        outIdx[x] = INT32_MIN;
        mag[x] = 0.f;
    }
}
__global__ void k_centres(int *outSym, int *outIdx, float *mag, const cf *sig, float spSym, float offset,
                          int lenSig, int M, int W, int op, int capacity) {
    centres_body(outSym, outIdx, mag, sig, spSym, offset, lenSig, M, W, op, capacity, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// ---- one-call block path (mfb_receive_block): the scalar host arithmetic of the receive loop, on the device ----------
// The reference reads the pick back, interpolates the shift on the host in float64 (DB:609-620), launches the matched
// filters, reads the rate/phase triple back, derives samples per symbol and code phase on the host in float64
// (DB:733-752) and launches findCentres -- two round trips in the middle of a block.  Here the same float64 arithmetic,
// operation for operation (explicit round-to-nearest intrinsics: no contraction), runs in two single-thread kernels
// between the stages, so that a block is ONE stream of launches and one read-back.
struct BlockScalars {
    double frac, spSym, codeOffset;
    float pick[2];         // {index, metric} of findDopplerEst
    float cr[3];           // {k*, arg P[k*], |P[k*]|^2}
    float spSymF, offsetF; // what findCentres gets (DB:999: float32 casts)
    int shift;             // dopplerIdxlast
    int low, high;         // int(idx), ceil(idx)
    int pick_valid;        // 0: NaN index -> block skipped (DB:625-630), shift = 0
    int count;             // int(N / max(spSym, spsym_min))
    int rate_fallback;     // k* == 0 -> spSym = 10 (DB:737-740)
    int band[2][2][2];     // SNR windows (DB:635-667): [signal | noise][piece][start, length] in the spectrum
    int band_len[2];
    // batches with the stream stages on the device (stream_kernels.hpp): A12 / A13 ...
    int a13_status;        // 1: the record holds the block's kept bits / centres / trust bytes and its tail; 0: the host does it
    int a13_start, a13_end, a13_nwin;      // the kept window [start, end) after the +-1 repair; its length
    int a13_noerr;         // impossible NRZ-S transitions
    int a13_npost, a13_nend;               // bits behind the window / the last offset + 1 bits inside it, as delivered
    // ... and A14 on the stream without a stash
    int sync_valid;        // 1: sync_count / the hit lists are those of the last numBitsOverlap bits before the block + its bits
    int sync_count[2];
    int a13_prev_npost;    // bits the previous block left behind its window (a13_status 2: the operand numpy could not broadcast)
    int pad_;
};
static_assert(sizeof(BlockScalars) == 168 && sizeof(BlockScalars) <= 256, "BlockScalars: the record head (the Python side mirrors this layout)");

// Python's slice(a, b).indices(N) for step 1; has_a / has_b = 0 stand for None
DEVI void slice_indices(int has_a, int a, int has_b, int b, int N, int *start, int *len) {
    int s = has_a ? (a < 0 ? (a + N < 0 ? 0 : a + N) : (a > N ? N : a)) : 0;
    int e = has_b ? (b < 0 ? (b + N < 0 ? 0 : b + N) : (b > N ? N : b)) : N;
    *start = s;
    *len = e > s ? e - s : 0;
}

// shift interpolation (DB:609-616) and the spectrum windows computeSNR reads (DB:635-667); one thread
DEVI void block_pick_body(const float *res, const int *shifts, int Dtot, int N, int w, BlockScalars *out) {
    const float idx = res[0];
    out->pick[0] = idx;
    out->pick[1] = res[1];
    out->band_len[0] = out->band_len[1] = 0;
    for (int q = 0; q < 8; ++q) (&out->band[0][0][0])[q] = 0;
    // int(NaN) raises in the reference (ValueError -> block skipped); +-inf cannot come out of findDopplerEst
    if (!(idx == idx) || idx < 0.f || idx > (float)(Dtot - 1)) {
        out->pick_valid = 0;
        out->shift = 0;
        out->low = out->high = 0;
        out->frac = 0.0;
        return;
    }
    const int low = (int)idx;                    // int() truncates
    const int high = (int)ceilf(idx);
    const double frac = (double)idx - floor((double)idx);      // float(idx) % 1 for idx >= 0: exact
    const int s_lo = shifts[low], s_hi = shifts[high];
    const double pos = __dadd_rn((double)s_lo, __dmul_rn((double)(s_hi - s_lo), frac));
    out->pick_valid = 1;
    out->low = low;
    out->high = high;
    out->frac = frac;
    out->shift = (int)rint(pos);                 // np.round: half to even
    const int half = N / 2;
    const int lo = s_lo, hi = s_hi;
    const int nlo = (lo + half) % N, nhi = (hi + half) % N;
    const int ab[2][2] = {{lo, hi}, {nlo, nhi}};
    for (int b = 0; b < 2; ++b) {
        const int a = ab[b][0], e = ab[b][1];
        int (*piece)[2] = out->band[b];
        if (a > e) {         // the band wraps around 0 Hz: X[a-w:] then X[:e+w]
            slice_indices(1, a - w, 0, 0, N, &piece[0][0], &piece[0][1]);
            slice_indices(0, 0, 1, e + w, N, &piece[1][0], &piece[1][1]);
        } else {
            slice_indices(1, a - w, 1, e + w, N, &piece[0][0], &piece[0][1]);
            piece[1][0] = piece[1][1] = 0;
        }
        out->band_len[b] = piece[0][1] + piece[1][1];
    }
}

// Block path, search mode: findDopplerEst, then (one thread) the shift interpolation, then (all 64 threads) the two SNR
// windows copied piece after piece into bands[band][cap] -- one launch instead of three; longer bands are flagged by
// band_len > cap on the host.
// A batch of blocks (mfb_receive_blocks): workgroup b takes block b -- its score table (num + offset rows on), its spectrum (N
// elements on), its result record (rec_stride bytes on: scalars and bands live in the record).
__global__ void __launch_bounds__(64) k_pick_block(const float *in, float *res, int num, int offset, int M, int sum_all, const int *shifts,
                                                   int Dtot, int N, int w, const cf *X, BlockScalars *out, cf *bands, int cap,
                                                   size_t rec_stride) {
    __shared__ float sIdx[64], sVal[64];
    __shared__ float scol[PICK_LDS];
    {
        const size_t b = blockIdx.x;
        in += b * (size_t)(num + offset) * (size_t)M;
        res += 2 * b;
        X += b * (size_t)N;
        out = reinterpret_cast<BlockScalars *>(reinterpret_cast<uint8_t *>(out) + b * rec_stride);
        bands = reinterpret_cast<cf *>(reinterpret_cast<uint8_t *>(bands) + b * rec_stride);
    }
    pick_body(in, res, num, offset, M, sum_all, sIdx, sVal, scol);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        out->rate_fallback = 0;
        out->count = 0;
        block_pick_body(res, shifts, Dtot, N, w, out);
    }
    __threadfence_block();
    __syncthreads();
    for (int b = 0; b < 2; ++b) {
        const int n0 = out->band[b][0][1], n = n0 + out->band[b][1][1];
        for (int i = threadIdx.x; i < n && i < cap; i += 64)
            bands[(size_t)b * cap + i] = i < n0 ? X[out->band[b][0][0] + i] : X[out->band[b][1][0] + (i - n0)];
    }
}
// Block path, fixed shift (STX): no pick -- the scalars of the search are cleared
__global__ void k_block_clear(BlockScalars *out, size_t rec_stride) {
    if (threadIdx.x) return;
    out = reinterpret_cast<BlockScalars *>(reinterpret_cast<uint8_t *>(out) + (size_t)blockIdx.x * rec_stride);
    out->pick[0] = out->pick[1] = 0.f;
    out->pick_valid = 0;
    out->shift = out->low = out->high = 0;
    out->frac = 0.0;
    out->band_len[0] = out->band_len[1] = 0;
    for (int q = 0; q < 8; ++q) (&out->band[0][0][0])[q] = 0;
}

// samples per symbol and code phase (DB:733-752), the clamp and the symbol count of cudaFindCentres (DB:994-999), in
// float64 exactly as the host wrote them; one thread
DEVI void block_rate_body(const float *cr, int N, int spsym_min, int capacity, BlockScalars *out) {
    out->cr[0] = cr[0];
    out->cr[1] = cr[1];
    out->cr[2] = cr[2];
    const double k = (double)cr[0];
    double spSym;
    if (k == 0.0) {
        spSym = 10.0;
        out->rate_fallback = 1;
    } else {
        spSym = __ddiv_rn((double)N, k);
        out->rate_fallback = 0;
    }
    // codeOffset = -arg / pi * spSym / 2, left to right
    double off = __ddiv_rn(__dmul_rn(__ddiv_rn(-(double)cr[1], 3.141592653589793), spSym), 2.0);
    if (off < 0.0) off = __dadd_rn(off, __dadd_rn(spSym, -1.0));
    out->spSym = spSym;
    out->codeOffset = off;
    double sp = spSym;
    if (sp < (double)spsym_min) sp = (double)spsym_min;
    out->spSymF = (float)sp;
    out->offsetF = (float)off;
    int count = (int)__ddiv_rn((double)N, sp);
    if (count > capacity) count = capacity;
    out->count = count;
}

// Test seam (mfb_debug_block_scalars): the two single-thread stages above on n INJECTED device results -- thread i takes pick i
// and rate triple i -- so that fixtures recorded from the reference's own host code gate this arithmetic bit for bit.
__global__ void k_block_scalars_debug(int n, const float *picks, const float *triples, const int *shifts, int Dtot, int N, int w,
                                      int spsym_min, int capacity, BlockScalars *out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    BlockScalars sc;
    block_pick_body(picks + 2 * i, shifts, Dtot, N, w, &sc);
    block_rate_body(triples + 3 * i, N, spsym_min, capacity, &sc);
    out[i] = sc;
}

// findCentres with its two float arguments taken from the block scalars
// (blockIdx.y = block of a batch: record rec_stride bytes on, matched-filter outputs M * lenSig elements on)
__global__ void k_centres_block(int *outSym, int *outIdx, float *mag, const cf *sig, const BlockScalars *sc, int lenSig, int M, int W,
                                int op, int capacity, size_t rec_stride) {
    {
        const size_t off = (size_t)blockIdx.y * rec_stride;
        outSym = reinterpret_cast<int *>(reinterpret_cast<uint8_t *>(outSym) + off);
        outIdx = reinterpret_cast<int *>(reinterpret_cast<uint8_t *>(outIdx) + off);
        mag = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(mag) + off);
        sc = reinterpret_cast<const BlockScalars *>(reinterpret_cast<const uint8_t *>(sc) + off);
        sig += (size_t)blockIdx.y * (size_t)M * (size_t)lenSig;
    }
    centres_body(outSym, outIdx, mag, sig, sc->spSymF, sc->offsetF, lenSig, M, W, op, capacity, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// Block path: the same, and thread 0 goes on to the float64 rate/phase arithmetic (one launch instead of two)
// (blockIdx.x = block of a batch)
__global__ void k_code_rate_block(const cf *P, float *out, int offset, int len, int N, int spsym_min, int capacity, BlockScalars *sc,
                                  size_t rec_stride) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    P += (size_t)blockIdx.x * (size_t)N;
    out += 3 * (size_t)blockIdx.x;
    sc = reinterpret_cast<BlockScalars *>(reinterpret_cast<uint8_t *>(sc) + (size_t)blockIdx.x * rec_stride);
    code_rate_body(P, out, offset, len, sv, si);
    if (threadIdx.x == 0) {
        __threadfence_block();
        block_rate_body(out, N, spsym_min, capacity, sc);
    }
}

