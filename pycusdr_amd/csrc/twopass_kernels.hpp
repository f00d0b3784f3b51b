// twopass_kernels.hpp -- the length-N two-pass transforms (N = N1 * N2) through an HBM intermediate: the fallback search
// for filters without a short impulse response, and the plain forward transforms (A3, the envelope spectrum of A10,
// mfb_xcorr).  Layout and index maps are described at the top of mfbank.hip.
#pragma once
#include "fft_core.hpp"

#define KIND_BANK 0  // X shifted * mask, inverse
#define KIND_FWDC 1  // complex input, forward (conjugate in)
#define KIND_FWDR 2  // real input, forward
#define MODE_REDUCE 0
#define MODE_STORE 1

#define TILE 16

struct P1Args {
    const cf *X;        // spectrum (BANK) or time-domain complex input (FWDC)
    const float *Xr;    // real input (FWDR)
    const cf *masks;    // [M][N]
    cf *Z;              // [rows][N1][N2]
    const int *shifts;  // device shift table
    const cf *tw1;      // W_N1
    const cf *twLo;     // W_N^x, x < 2^lo
    const cf *twHi;     // W_N^(y*2^lo)
    int N, N2, lo;
    int M;        // filter rows this launch transforms per Doppler bin (all M, or the unique ones)
    const int *rows;  // bank row of each of them (nullptr: identity)
    int mpb;      // masks per block
    int j0;       // first Doppler index of this chunk (into shifts)
    int dc;       // Doppler bins in this chunk
    int ntiles, mgroups, jsplit;  // 1-D grid = ntiles * mgroups * jsplit workgroups
    int fixed_shift;  // used when shifts == nullptr
    int in_stride;    // FWDC / FWDR with jsplit = rows > 1: elements between the inputs of consecutive rows (row r -> Z row r)
};

struct P2Args {
    const cf *Z;     // [rows][N1][N2]
    const cf *tw2;   // W_N2
    float *partials; // REDUCE: [..][PARTS]
    int N, N1, N2;
    int srb;         // sub-rows (n1 values) per block
    int parts;       // N1 / srb
    int part_row0;   // REDUCE: first (j*M+m) row of this chunk in the partials array
    float scale;     // REDUCE: 1/2^18
};

// ------------------------------------------------------------------------------------------------
// pass 1: strided N1-point FFTs on a tile of 16 adjacent columns
//   grid = (N2/16 tiles, mask groups, jsplit); a workgroup keeps its tile and mask group and walks
//   over the Doppler bins jl = blockIdx.z, blockIdx.z + gridDim.z, ... of the chunk, so the
//   per-thread twiddles (inner W_N1 and inter-pass W_N) are built once and stay in registers.
// ------------------------------------------------------------------------------------------------
// Kernel-shape switches, chosen by interleaved A/B runs on MI355X at C2 (tools/probe_var.sh):
//   ping-pong LDS halves (one barrier per exchange instead of two) bought nothing measurable,
//   while the single buffer (35 KiB) lets three workgroups share a CU;
//   pass 2 runs best at 3 waves/SIMD (168 VGPRs, 3 dwords spilled): 3.55 -> 3.17 ms;
//   pass 1 runs best at 2 waves/SIMD WITH the one-filter-ahead register prefetch (3.46 ms; without
//   it at 3 waves/SIMD: 3.85 ms).
#ifndef MFB_P1_PP
#define MFB_P1_PP 0
#endif
#ifndef MFB_P2_PP
#define MFB_P2_PP 0
#endif
#ifndef MFB_P1_TILEMAP
#define MFB_P1_TILEMAP 0
#endif
#ifndef MFB_P1_PREFETCH
#define MFB_P1_PREFETCH 1
#endif
#ifndef MFB_P1_WAVES
#define MFB_P1_WAVES 2
#endif
#ifndef MFB_P2_WAVES
#define MFB_P2_WAVES 3
#endif

template <int L1>
struct P1Cfg {
    static constexpr int NT = L1 / 16;
    static constexpr int HALF = padlen(L1) * TILE;
    static constexpr bool PP = MFB_P1_PP;  // exchange buffer: 34 KiB at L1 = 256 (x2 with ping-pong)
    static constexpr size_t lds_bytes = (size_t)HALF * (PP ? 2 : 1) * sizeof(cf);
};

template <int L1, int KIND>
__global__ void __launch_bounds__((L1 / 16) * TILE, (L1 == 256 && KIND == KIND_BANK) ? MFB_P1_WAVES : 1) k_pass1(P1Args a) {
    using Cfg = P1Cfg<L1>;
    constexpr int NT = Cfg::NT;
    constexpr int l1 = ilog2c(L1);
    constexpr int NP = npass(l1);
    constexpr int RL = radix_of(l1, NP - 1);  // radix of the last pass
    constexpr int NBL = 16 / RL;
    constexpr int PCL = L1 / RL;  // prefixCount of the last pass
    extern __shared__ __attribute__((aligned(16))) cf lds[];

    const int tid = threadIdx.x;
    const int col = tid & (TILE - 1);
    const int g = tid >> 4;
    // XCD-aware decode of the 1-D grid.  Workgroups are dealt round-robin over the 8 XCDs
    // (b % 8 names the XCD-sharing group), so inside one XCD consecutive workgroups are made to
    // share a TILE and differ in the Doppler stream: the tile's filter rows (M x N1 x 128 B) are
    // then served from that XCD's L2 for every Doppler bin instead of being re-fetched.
    // Placement only affects speed, never results.
    int tile, mg, zj;
    {
        const int b = blockIdx.x;
        const int per_tile = a.mgroups * a.jsplit;
        if ((a.ntiles & 7) == 0) {
            const int x = b & 7, q = b >> 3;
#if MFB_P1_TILEMAP
            tile = x * (a.ntiles >> 3) + (q / per_tile);   // each XCD owns a contiguous range of tiles
#else
            tile = (q / per_tile) * 8 + x;
#endif
            const int r = q % per_tile;
            mg = r / a.jsplit;
            zj = r % a.jsplit;
        } else {
            tile = b / per_tile;
            const int r = b % per_tile;
            mg = r / a.jsplit;
            zj = r % a.jsplit;
        }
    }
    const int k2 = tile * TILE + col;
    const int N = a.N, N2 = a.N2;
    int ebuf = 0;

    // inter-pass twiddles W_N^(k2*n1) for the 16 outputs this thread owns, and the inner twiddles:
    // both depend on the thread only -> built once per workgroup, kept in registers.
    cf twN[16];
    sfor<0, NBL>([&](auto u) {
        sfor<0, RL>([&](auto p) {
            const int n1 = decltype(p)::value * PCL + g + NT * decltype(u)::value;
            const unsigned t = (unsigned)k2 * (unsigned)n1;
            twN[decltype(u)::value * RL + decltype(p)::value] = cmul(a.twHi[t >> a.lo], a.twLo[t & ((1u << a.lo) - 1u)]);
        });
    });
    TwRegs<L1> twr;
    load_twiddles<L1>(twr, a.tw1, g);

    const int ebase = N2 * g + k2;  // element index of slot i: ebase + i * (N2*NT)
    const int estride = N2 * NT;
    const unsigned rowbytes = (unsigned)N * sizeof(cf);
    const int vo_in = ebase * (int)sizeof(cf);   // per-thread byte offset of input slot 0
    const int vo_out = ebase * (int)sizeof(cf);  // output (n1 = g, k2) sits at the same offset in a Z row
    const int so_in = estride * (int)sizeof(cf);
    const int so_out = N2 * (int)sizeof(cf);     // per unit of n1

    // last pass emits n1 = p*PCL + g + NT*u  ->  scalar offset (p*PCL + NT*u) * N2 * 8
    auto z_store = [&](__amdgpu_buffer_rsrc_t zr, cf val, auto slot, auto nu) {
        buf_store_cf<MFB_AUX_ZSTORE>(zr, vo_out, decltype(nu)::value * so_out, cmul(val, twN[decltype(slot)::value]));
    };

    if constexpr (KIND != KIND_BANK) {
        // plain forward transforms: one row per value of zj (a batch of blocks: mfb_receive_blocks)
        cf v[16];
        if constexpr (KIND == KIND_FWDC) {
            const auto xr = mk_rsrc(a.X + (size_t)zj * (size_t)a.in_stride, rowbytes);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = cconj(buf_load_cf(xr, vo_in, i * so_in));
        } else {
            const auto xr = mk_rsrc(a.Xr + (size_t)zj * (size_t)a.in_stride, (unsigned)N * sizeof(float));
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = mkc(buf_load_f(xr, vo_in / 2, i * (so_in / 2)), 0.f);
        }
        const auto zr = mk_rsrc(a.Z + (size_t)zj * (size_t)N, rowbytes);
        auto store = [&](int, cf val, auto slot, auto nu) { z_store(zr, val, slot, nu); };
        fft_passes<L1, TILE, 0, true, Cfg::PP, Cfg::HALF>(v, lds, ebuf, g, col, twr, a.tw1, store);
    } else {
        const int m0 = mg * a.mpb;
        const int m1 = min(m0 + a.mpb, a.M);
        const auto xr = mk_rsrc(a.X, rowbytes);
        cf mk[16];  // mask values of the NEXT transform (software prefetch)
        if constexpr (MFB_P1_PREFETCH) {
            const int r0 = a.rows ? a.rows[m0] : m0;
            const auto mr = mk_rsrc(a.masks + (size_t)r0 * N, rowbytes);
#pragma unroll
            for (int i = 0; i < 16; ++i) mk[i] = buf_load_cf(mr, vo_in, i * so_in);
        }
        for (int jl = zj; jl < a.dc; jl += a.jsplit) {
            const int shift = a.shifts ? a.shifts[a.j0 + jl] : a.fixed_shift;
            cf xv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                xv[i] = buf_load_cf(xr, ((ebase + i * estride + shift) & (N - 1)) * (int)sizeof(cf), 0);
            for (int m = m0; m < m1; ++m) {
                cf v[16];
                if constexpr (MFB_P1_PREFETCH) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = cmul(xv[i], mk[i]);
                    const int mn = (m + 1 < m1) ? (m + 1) : m0;  // always valid: branch-free prefetch
                    const int rn = a.rows ? a.rows[mn] : mn;
                    const auto mr = mk_rsrc(a.masks + (size_t)rn * N, rowbytes);
#pragma unroll
                    for (int i = 0; i < 16; ++i) mk[i] = buf_load_cf(mr, vo_in, i * so_in);
                } else {
                    const int rm = a.rows ? a.rows[m] : m;
                    const auto mr = mk_rsrc(a.masks + (size_t)rm * N, rowbytes);
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = buf_load_cf(mr, vo_in, i * so_in);
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = cmul(xv[i], v[i]);
                }
                const auto zr = mk_rsrc(a.Z + ((size_t)jl * a.M + m) * (size_t)N, rowbytes);
                auto store = [&](int, cf val, auto slot, auto nu) { z_store(zr, val, slot, nu); };
                fft_passes<L1, TILE, 0, true, Cfg::PP, Cfg::HALF>(v, lds, ebuf, g, col, twr, a.tw1, store);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: contiguous N2-point FFTs; RB rows side by side when N2/16 < 256 threads
// ------------------------------------------------------------------------------------------------
DEVI float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

template <int L2>
struct P2Cfg {
    static constexpr int NT = L2 / 16;
    static constexpr int RB = NT >= 256 ? 1 : 256 / NT;
    static constexpr int HALF = padlen(L2) * RB;
    static constexpr bool PP = MFB_P2_PP && (size_t)HALF * 2 * sizeof(cf) <= 72 * 1024;
    static constexpr bool HOIST = L2 <= 8192;  // (16384-point rows would run 1024 threads on a 128-VGPR budget and spill: N = 2^22 is split 512 x 8192 instead)
    static constexpr size_t lds_bytes = (size_t)HALF * (PP ? 2 : 1) * sizeof(cf);
};

template <int L2, int MODE>
__global__ void __launch_bounds__((L2 / 16) * P2Cfg<L2>::RB, (L2 == 4096 && MODE == MODE_REDUCE) ? MFB_P2_WAVES : 1) k_pass2(P2Args a) {
    using Cfg = P2Cfg<L2>;
    constexpr int NT = Cfg::NT;
    constexpr int RB = Cfg::RB;
    constexpr int NTHREADS = NT * RB;
    extern __shared__ __attribute__((aligned(16))) cf lds[];
    __shared__ float red[16];

    const int tid = threadIdx.x;
    const int g = tid % NT;
    const int rb = tid / NT;
    cf *mylds = lds + rb * padlen(L2);  // both ping-pong halves are laid out [half][rb][padlen]
    int ebuf = 0;

    const int row = blockIdx.y;  // (jl*M + m) for REDUCE, output row for STORE
    const int sr0 = blockIdx.x * a.srb;
    const cf *zrow = a.Z + (size_t)row * a.N;
    const unsigned fftbytes = (unsigned)L2 * sizeof(cf);
    const int vo = g * (int)sizeof(cf);
    constexpr int so = NT * (int)sizeof(cf);
    float acc = 0.f;

    TwRegs<L2> twr;
    if constexpr (Cfg::HOIST) load_twiddles<L2>(twr, a.tw2, g);

    // Every thread runs the same number of iterations (the passes contain workgroup barriers).
    // When RB does not divide the block's row range, the surplus row slots recompute the block's
    // last row: their sums are weighted 0 and their stores rewrite identical values.
    // For RB == 1 every row descriptor is wave-uniform (SGPRs); for RB > 1 (small transforms) rows
    // differ inside a wave and the row offset goes into the per-thread byte offset instead.
    cf nv[16];  // next row (software prefetch)
    auto load_row = [&](int rsel) {
        if constexpr (RB == 1) {
            const auto zr = mk_rsrc(zrow + (size_t)(sr0 + rsel) * L2, fftbytes);
#pragma unroll
            for (int i = 0; i < 16; ++i) nv[i] = buf_load_cf<MFB_AUX_ZLOAD>(zr, vo, i * so);
        } else {
            const auto zr = mk_rsrc(zrow + (size_t)sr0 * L2, (unsigned)a.srb * fftbytes);
            const int vr = vo + rsel * (int)fftbytes;
#pragma unroll
            for (int i = 0; i < 16; ++i) nv[i] = buf_load_cf<MFB_AUX_ZLOAD>(zr, vr, i * so);
        }
    };
    load_row(min(rb, a.srb - 1));
    for (int base = 0; base < a.srb; base += RB) {
        const int rsel = min(base + rb, a.srb - 1);
        const float okf = (base + rb < a.srb) ? 1.f : 0.f;
        cf v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = nv[i];
        if (base + RB < a.srb) load_row(min(base + RB + rb, a.srb - 1));  // wave-uniform condition
        if constexpr (MODE == MODE_REDUCE) {
            cf racc = mkc(0.f, 0.f);   // (sum re^2, sum im^2): one packed FMA per point
            auto store = [&](int, cf val, auto, auto) { racc = __builtin_elementwise_fma(val, val, racc); };
            fft_passes<L2, 1, 0, Cfg::HOIST, Cfg::PP, Cfg::HALF>(v, mylds, ebuf, g, 0, twr, a.tw2, store);
            const float rs = racc.x + racc.y;
            acc += (RB == 1) ? rs : rs * okf;
        } else {
            // the transformed row goes back IN PLACE as Z[row][n1][n2] (coalesced; this workgroup is
            // the only reader and writer of the row and has it in registers); k_transpose then
            // produces the natural order y[n1 + N1*n2]
            cf *zw = const_cast<cf *>(zrow) + (size_t)sr0 * L2;
            const auto orr = mk_rsrc(zw, (unsigned)a.srb * fftbytes);
            const int vw = vo + rsel * (int)fftbytes;
            auto store = [&](int, cf val, auto, auto nu) {
                buf_store_cf(orr, vw, decltype(nu)::value * (int)sizeof(cf), val);
            };
            fft_passes<L2, 1, 0, Cfg::HOIST, Cfg::PP, Cfg::HALF>(v, mylds, ebuf, g, 0, twr, a.tw2, store);
        }
    }

    if constexpr (MODE == MODE_REDUCE) {
        // registers -> wavefront (64 lanes) -> workgroup, fixed order
        acc = wave_sum(acc);
        const int wid = tid >> 6;
        if ((tid & 63) == 0) red[wid] = acc;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < NTHREADS / 64; ++w) s += red[w];
            a.partials[(size_t)(a.part_row0 + row) * a.parts + blockIdx.x] = s * a.scale;
        }
    }
}
