// energy_kernels.hpp -- opt-in spectral-energy Doppler search (MFB_SEARCH_ENERGY).
//
// The search needs only doppSum[j][m] = sum_n |y_jm[n]|^2 / 2^18 over ALL N outputs of the circular correlation
// y_jm = IFFT_N(X[(k+s_j) mod N] . H_m[k]) (reference: cuda_kernels.cu:421-480 sums whole rows of the unnormalised
// inverse transforms, DB:578-591).  By Parseval's identity that is
//     doppSum[j][m] = N/2^18 . sum_k |X[(k+s_j) mod N]|^2 . |H_m[k]|^2
// and, under SUM_ALL_MASKS, sum_m of it = N/2^18 . sum_k P[(k+s_j) mod N] . W[k] with W = sum_m |H_m|^2:
// D.N multiply-adds on two real vectors instead of D.M inverse transforms.  Same numbers to fp32 rounding, but no
// transform runs, so this is an algorithmic shortcut: OFF by default, never the benchmarked path.
// The matched filtering at the chosen shift (A9) is untouched.
#pragma once
#include "fft_core.hpp"

// P[k] = |X[k]|^2
__global__ void __launch_bounds__(256) k_power(const cf *X, float *P, int N) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < N) {
        const cf z = X[k];
        P[k] = z.x * z.x + z.y * z.y;
    }
}

// W[r][k]: sum over the filters of |H_m[k]|^2 (sum_all: one row, added in filter order) or |H_r[k]|^2 per filter
__global__ void __launch_bounds__(256) k_filter_energy(const cf *H, float *W, int N, int M, int sum_all) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    if (sum_all) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) {
            const cf z = H[(size_t)m * N + k];
            s += z.x * z.x + z.y * z.y;
        }
        W[k] = s;
    } else {
        const cf z = H[(size_t)blockIdx.y * N + k];
        W[(size_t)blockIdx.y * N + k] = z.x * z.x + z.y * z.y;
    }
}

// partials[j][r][c] = scale . sum_{k in chunk c} P[(k + s_j) mod N] . W[r][k]
// grid (N / EN_CHUNK, R); 4 waves; every wave keeps the chunk of W in registers (lane l: k0 + l + 64 i) and takes
// the bins j = wave, wave + 4, ...: per bin EN_VPL coalesced loads of the rotated window of P (8 MiB for P and W
// together: served by L2 / Infinity Cache), EN_VPL multiply-adds in index order, one fixed butterfly over the lanes.
#define EN_VPL 16
#define EN_CHUNK (64 * EN_VPL)
__global__ void __launch_bounds__(256) k_energy(const float *P, const float *W, const int *shifts, float *partials, int N, int D, int R,
                                                int parts, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x, r = blockIdx.y;
    const int k0 = c * EN_CHUNK + lane;
    float w[EN_VPL];
#pragma unroll
    for (int i = 0; i < EN_VPL; ++i) w[i] = W[(size_t)r * N + k0 + 64 * i];
    const int mask = N - 1;
    for (int j = wave; j < D; j += 4) {
        const int s = shifts[j];
        float p[EN_VPL];
#pragma unroll
        for (int i = 0; i < EN_VPL; ++i) p[i] = P[(k0 + 64 * i + s) & mask];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < EN_VPL; ++i) acc = fmaf(p[i], w[i], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) partials[((size_t)j * R + r) * parts + c] = acc * scale;
    }
}
