// seg_kernels.hpp -- single-pass overlap-save Doppler search / matched filtering for gfx950.
//
// The reference forms  y[s][m][n] = IFFT_N( X[(k + s) mod N] * H_m[k] )  (cuda_kernels.cu:339-373 +
// batched cuFFT, demodulator_base.py:578-591) -- a length-N circular convolution of
//     x_s[n] = x[n] * e^{-2 pi i s n / N}          (the spectrum shift, done in time)
// with the impulse response h_m = ifft(H_m).  When h_m is T taps long (every shipped protocol: 48...640,
// filter_taps.hpp) the same y is obtained from L-point segments, L >= 2T:
//     segment q covers samples b0 = q*V ... b0 + L - 1 (circular in N),  V <= L - T + 1 outputs each
//     U = FFT_L(x_s[b0 ...]),  v_m = IFFT_L(U * G_m),  y[s][m][(b0 + i + off) mod N] = v_m[i], 0 <= i < V
// with G_m the (N/L)-scaled L-point spectrum of the rotated taps.  Nothing of length N is ever written:
// a segment lives in registers + LDS from the x load to the |.|^2 sum (Doppler search, SEG_REDUCE) or
// to the natural-order store of y (demodulation, SEG_STORE).  HBM traffic of the whole search is the
// 8 MiB block (served from L2 / Infinity Cache thereafter) plus a few KiB of partial sums, so the kernel
// is bound by the fp32 vector rate, not by HBM.
//
// Work layout.  One transform is run by NT = L/P threads holding P points each (fft_core.hpp): P = 16 and radix-16 passes chained
// through one LDS exchange each, or -- L = 2048 -- P = 32 and two 32-point in-register butterflies around ONE exchange (fft_w32).  A
// "team" is the set of threads that must synchronise for a transform: one wavefront carrying CT = 64/NT segments side by side
// for L <= 2048 (no s_barrier anywhere: LDS exchanges are wave-local), or NT threads behind workgroup barriers for L = 4096 (4
// waves) and 8192 (8 waves, one team per CU).  Teams never talk to each other: every wave writes one partial sum per slot, added up
// in fixed order by k_finalize (bit-reproducible, and independent of the grid decomposition: see the store below).
// The grid is decoded XCD-aware: workgroups with equal blockIdx % nsg share a contiguous range of
// segments, i.e. one eighth of the block, which stays in that XCD's L2 while every Doppler bin passes
// over it (placement affects speed only).
//
// Issue-slot economy (the kernel is bound by instruction issue, measured: rocprofv3 SQ counters).
//   * V is a multiple of NT, so the valid outputs of a complete segment are exactly the register slots
//     k < PV = V / NT of every lane: PV is a TEMPLATE parameter, the |y|^2 accumulation has no masks and
//     no branches, and the butterflies feeding dead slots are pruned by the compiler.  Incomplete
//     segments (the last one of the block, padding of barrier teams) go through the PV = -1 instantiation,
//     which masks per lane; the host launches it separately over the few slots that need it.
//   * The mixing phasor of point i of a lane is  W_N^(s*e0) * W_N^(s*NT*i): one gathered base phasor per
//     lane and unit, times 15 step phasors that depend on the Doppler bin only and are kept in LDS
//     (broadcast reads) -- instead of two table gathers and ~9 integer ops per point.  The Doppler
//     search goes further: a phase that is constant over a segment multiplies every output of that
//     segment by a unit phasor and drops out of |y|^2, so only the phase RELATIVE to the segment start
//     is applied, W_N^(s*(g + NT*i)) -- a function of the lane and the bin alone, built once per bin
//     (per-wave table in LDS for L = 256, lane phasor in registers otherwise): no gathers per segment.
//   * x is read with immediate offsets (no per-point index arithmetic) unless the slot wraps past N.
#pragma once
#include "fft_core.hpp"

#define SEG_REDUCE 0
#define SEG_STORE 1
#define SEG_MPB_MAX 16   // filters per team pass
// Per-wave stash of the per-lane |.|^2 sums of a slot, one row per filter of the pass: row stride 68 floats keeps both the
// row-wise stores and the column-wise reads of the reduction (lane = (filter, j): elements j, j+4, ...) off shared banks.
#define SEG_ACC_STRIDE 68

// waves per SIMD the register budget is pinned to: 3 for L = 256 (one twiddled pass: 149 VGPRs), 2 for the
// longer transforms (two twiddled passes: ~210 VGPRs; 3 waves would spill)
#ifndef MFB_SEG_WAVES_SHORT
#define MFB_SEG_WAVES_SHORT 3
#endif
#ifndef MFB_SEG_WAVES_MID
#define MFB_SEG_WAVES_MID 2      // wave-local transforms of 512 / 1024 points
#endif
#ifndef MFB_SEG_WAVES_LONG
#define MFB_SEG_WAVES_LONG 2     // barrier teams: 2048 / 4096 points
#endif
#ifndef MFB_SEG_PREFETCH
#define MFB_SEG_PREFETCH 1
#endif
// issue the first filter's spectrum loads BEFORE the forward transform, so that they land while it runs
// barrier teams (L = 2048 / 4096): two exchange buffers used alternately -> one workgroup barrier per exchange instead of two
#ifndef MFB_SEG_PP
#define MFB_SEG_PP 1
#endif
// threads per workgroup of the kernels whose teams are single, independent waves (L <= 1024): 64, 128 or 256
// Barrier teams: two filters of a segment through their inverse transforms in lock-step (fft_passes2), so that a wave has one
// transform's butterflies to run while the other's exchange is in flight.  Measured on the CC11xx bank: 3.48-3.50 ms against
// 3.40-3.43 (L = 4096), 3.33 against 3.29 (L = 2048): the kernel does not wait for its own exchanges either.  Off.
#ifndef MFB_SEG_DUAL
#define MFB_SEG_DUAL 0
#endif
#ifndef MFB_SEG_BLOCK
#define MFB_SEG_BLOCK 256
#endif
static_assert(MFB_SEG_BLOCK == 64 || MFB_SEG_BLOCK == 128 || MFB_SEG_BLOCK == 256, "MFB_SEG_BLOCK: 64, 128 or 256 threads");
// Threads per workgroup of the 2048-point kernel (two-wave barrier teams): 256 = two teams behind one s_barrier, 128 = one.
// One team per workgroup: the s_barrier of a 256-thread workgroup keeps its two teams in step, so that all four waves
// compute and exchange at the same moments; four independent pairs per CU instead of two coupled quadruples run the
// 384-tap bank 5 % faster (3.05 against 3.22 ms at D = 256, same device, profiles/r03_long_filter.md).
#ifndef MFB_SEG_BLOCK_2048
#define MFB_SEG_BLOCK_2048 128
#endif
static_assert(MFB_SEG_BLOCK_2048 == 128 || MFB_SEG_BLOCK_2048 == 256, "MFB_SEG_BLOCK_2048: 128 or 256 threads");
// (8192 points: one eight-wave team, 512 threads)
constexpr int seg_block_threads(int NT) { return NT <= 64 ? MFB_SEG_BLOCK : (NT == 128 ? MFB_SEG_BLOCK_2048 : (NT == 512 ? 512 : 256)); }
#ifndef MFB_SEG_G0EARLY
#define MFB_SEG_G0EARLY 1
#endif
// half of the next filter's spectrum fetched during the running transform of the fused 2048-point kernel: measured 1 % SLOWER on
// the 384-tap bank (2.353 against 2.332 ms, three interleaved runs; profiles/r06_fft_ops.md) -- as in round 4, the spectrum loads
// cost bandwidth and registers, not latency.  Off.
// measurement only, never shipped on: every (bin, filter) of k_segf<256> reads the spectra of row 0 (profiles/r06_fft_ops.md section 6)
#ifndef MFB_SEG_FSM_PROBE_ONE_ROW
#define MFB_SEG_FSM_PROBE_ONE_ROW 0
#endif
// ... and the filters of a bin summed in registers, one partial sum per (bin, slot), reduced over the wave by DPP (no LDS round trip):
// for banks whose rows k_finalize counts equally often (every shipped one)
#ifndef MFB_SEG_PRESUM
#define MFB_SEG_PRESUM 1
#endif
// SUM_ALL searches: the Parseval half of the complement form once per bin from a table of sum_f |G_f|^2 instead of once per (bin, filter)
#ifndef MFB_SEG_SUMQ
#define MFB_SEG_SUMQ 1
#endif
// valid energy = total energy (Parseval on the product) - energy of the invalid outputs (segf_body): 1 on, 0 = every valid output
#ifndef MFB_SEG_COMPLEMENT
#define MFB_SEG_COMPLEMENT 1
#endif
#ifndef MFB_SEG_PREHALF
#define MFB_SEG_PREHALF 0
#endif
// 2048-point segments as ONE wave per segment, 32 points per lane, one LDS exchange per transform (fft_core.hpp, fft_w32)
// instead of two-wave barrier teams with 16 points per lane and two exchanges.
#ifndef MFB_SEG_W32
#define MFB_SEG_W32 1
#endif
// points per lane of the L-point segment kernel
constexpr int seg_ppl(int L) { return (L == 2048 && MFB_SEG_W32) ? 32 : 16; }

struct SegArgs {
    const cf *x;         // time-domain block, complex64 [N]
    const cf *G;         // segment spectra, complex64 [M][8][NT][2]: element (ii, g, e) = G_m[g + NT*(2*ii + e)],
                         // i.e. register slots 2*ii and 2*ii+1 of lane g side by side (one 16-byte load)
    const int *rows;     // bank row of filter slot m (nullptr: identity)
    const int *shifts;   // device shift table (nullptr: fixed_shift)
    const cf *twL;       // W_L^j, j < L (inverse sign)
    const cf *twLo;      // W_N^j, j < 2^lo
    const cf *twHi;      // W_N^(j * 2^lo)
    float *partials;     // REDUCE: [rows][parts]
    cf *out;             // STORE: complex64 [M][N], natural order
    float *env;          // STORE, optional: float32 [N] symbol-energy envelope sum_{env_lo <= m < env_hi} |y_m[n]|^2 (needs mgroups == 1)
    int env_lo, env_hi;
    int N, lo;
    int V;               // valid outputs per complete segment (multiple of NT)
    int slot0, nslots;   // this launch covers team-iterations (slots) [slot0, slot0 + nslots); a slot = CT segments
    int MU;              // filter slots per Doppler bin
    int Grows;           // rows of G (all M filters)
    int mpb, mgroups;    // filters per team pass, passes (workgroup groups) needed for MU
    int igroups;         // REDUCE: > 1 = the filter groups are walked INSIDE a team (one forward transform per segment for all of them; mgroups == 1)
    int nsg;             // segment groups (blockIdx % nsg)
    int bsplit, ssplit;  // a group's teams = bsplit Doppler streams x ssplit segment sub-ranges
    int j0, dc;          // Doppler bins [j0, j0 + dc) of the shift table
    int fixed_shift;
    int out_off;         // STORE: (window start + T_eff - 1) mod N
    int part_row0, parts;   // REDUCE: row of bin 0, row stride (= slots of the block x waves per team)
    float scale;         // REDUCE: 1 / 2^18
    // A batch of blocks in one launch (mfb_receive_blocks): stream jl of the launch is bin jl % dper of block jl / dper; that
    // block's samples start xstride elements behind the previous block's (consecutive blocks of a stream share their overlap),
    // its shift is shifts[j0 + jl % dper + (jl / dper) * sstride] and its STORE outputs go ostride (out) / N (env) elements on.
    // dper = 0: one block (every stream reads x, shifts[j0 + jl]).  Partial-sum rows are part_row0 + jl either way.
    int dper, xstride, sstride;
    long long ostride;
};

template <int L>
struct SegCfg {
    static constexpr int PPL = seg_ppl(L);        // points per lane
    static constexpr bool W32 = PPL == 32;
    static constexpr int NT = L / PPL;
    static constexpr int TEAM = NT < 64 ? 64 : NT;
    static constexpr int CT = TEAM / NT;          // segments side by side in a team
    static constexpr int BLOCK = seg_block_threads(NT);   // threads per workgroup
    static constexpr int TPW = BLOCK / TEAM;      // teams per workgroup
    static constexpr int WPT = TEAM / 64;         // waves per team
    static constexpr int SYNC = NT <= 64 ? 1 : 0;
    // Barrier teams (L = 2048 / 4096) at THREE waves per SIMD: measured on the CC11xx bank (384 taps, L = 4096), two waves
    // leave the vector pipe 57 % busy and the LDS pipe 38 % -- neither is saturated, the two waves of a SIMD simply cannot
    // cover each other's exchange latency (alternating exchange buffers, i.e. half the barriers, changes nothing).  A third
    // wave needs <= 168 VGPRs: the second twiddle set moves to LDS (fft_core.hpp, LTW), the spectrum prefetch registers go
    // (a third wave hides those loads instead), and one exchange buffer per team instead of two keeps three teams in LDS.
    static constexpr bool LONG3 = !SYNC && MFB_SEG_WAVES_LONG >= 3;
    // Barrier teams, Doppler search: two filters of a segment go through their inverse transforms in lock-step
    // (fft_passes2), each with an exchange buffer of its own, so that a wave has butterflies of one to run while the
    // exchange of the other is in flight.
    static constexpr bool DUAL = !SYNC && MFB_SEG_DUAL && !LONG3;
    static constexpr bool PP = !SYNC && MFB_SEG_PP && !LONG3 && !DUAL;
    static constexpr bool TWO_BUFFERS = PP || DUAL;
    // (8192 points: three twiddle sets are 90 registers; the spectrum prefetch registers would spill)
    static constexpr bool PREFETCH = MFB_SEG_PREFETCH && !LONG3 && !DUAL && !W32 && L < 8192;
    static constexpr bool LTW = LONG3;
    static constexpr int LTW_ELEMS = W32 ? W32Cfg::TW2_ELEMS : TwRegs<L, LTW>::LDS_ELEMS;
    static constexpr int HALF = W32 ? W32Cfg::XCHG : padlen(L) * CT;
    static constexpr int LDS_PER_TEAM = HALF * (TWO_BUFFERS ? 2 : 1);
    static constexpr int LDS_ELEMS = LDS_PER_TEAM * TPW;
    static constexpr int STEP_ELEMS = (BLOCK / 64) * PPL;    // one step phasor per register slot, per wave
    // Doppler search, L = 256: per-wave table of the relative mixing phasors W_N^(s*j), j < L (2 KiB a wave)
    static constexpr bool PHASE_TABLE = L <= 256;
    static constexpr int PHASE_ELEMS = PHASE_TABLE ? (BLOCK / 64) * L : 0;
    static constexpr int WAVES = L <= 256 ? MFB_SEG_WAVES_SHORT : (W32 ? 2 : (SYNC ? MFB_SEG_WAVES_MID : MFB_SEG_WAVES_LONG));   // per SIMD
    static constexpr size_t lds_bytes(int mpb) {   // mpb = 0: STORE mode (no phase table, no reduction stash)
        return (size_t)(LDS_ELEMS + STEP_ELEMS + LTW_ELEMS + (mpb ? PHASE_ELEMS : 0)) * sizeof(cf) +
               (size_t)(BLOCK / 64) * mpb * SEG_ACC_STRIDE * sizeof(float);
    }
};

// PV >= 0: every segment this role touches is complete; valid outputs = register slots k < PV.
// PV < 0 : per-lane masking (incomplete / padding segments, any V).
// `blk` is the workgroup's index in the launch.
template <int L, int MODE, int PV>
DEVI void seg_body(const SegArgs &a, const int blk) {
    using Cfg = SegCfg<L>;
    constexpr int NT = Cfg::NT, CT = Cfg::CT, TPW = Cfg::TPW, TEAM = Cfg::TEAM, SYNC = Cfg::SYNC, PPL = Cfg::PPL;
    constexpr bool W32 = Cfg::W32;
    constexpr bool MASKED = PV < 0;
    static_assert(MODE == SEG_REDUCE || MASKED, "the STORE mode masks per lane");
    extern __shared__ __attribute__((aligned(16))) cf lds[];
    constexpr bool REL = MODE == SEG_REDUCE;                  // phase relative to the segment start suffices
    constexpr bool PTAB = REL && Cfg::PHASE_TABLE;
    cf *lstep = lds + Cfg::LDS_ELEMS;                                         // [4 waves][16]
    cf *ltw = lstep + Cfg::STEP_ELEMS;                                        // LTW: twiddles of the later passes
    cf *lphase = ltw + Cfg::LTW_ELEMS;                                        // [4 waves][L] (PTAB)

    const int tid = threadIdx.x;
    const int team = __builtin_amdgcn_readfirstlane(tid / TEAM);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lt = tid % TEAM;
    // position of this lane inside a transform (W32: lane pairs stand 32 apart, fft_core.hpp)
    const int g = W32 ? ((lane >> 1) + 32 * (lane & 1)) : lt % NT;
    const int col = W32 ? 0 : lt / NT;
    cf *mylds = lds + team * Cfg::LDS_PER_TEAM + col * padlen(L);
    cf *mystep = lstep + wave * PPL;
    [[maybe_unused]] cf *myphase = lphase + wave * L;
    [[maybe_unused]] float *lacc = reinterpret_cast<float *>(lphase + (REL ? Cfg::PHASE_ELEMS : 0)) + wave * (a.mpb * SEG_ACC_STRIDE);
    int ebuf = 0;

    // fused-twiddle transforms (fft_core.hpp, round 6): 256 points and the wave-local 2048 points; their (cos, tan) tables sit
    // behind the W_L table
    constexpr bool F256 = MFB_FFT_FUSED && L == 256;
    constexpr bool F2048 = MFB_FFT_FUSED && W32;
    TwRegs<(W32 || F256) ? 16 : L, Cfg::LTW> twr;
    [[maybe_unused]] W32Regs w32;
    [[maybe_unused]] F256Regs f256;
    [[maybe_unused]] F2048Regs f2048;
    if constexpr (F256) {
        f256_setup(f256, a.twL + L, g);
    } else if constexpr (F2048) {
        f2048_setup(f2048, a.twL + L, lane);
    } else if constexpr (W32) {
        w32_setup(w32, a.twL, ltw, lane, tid, Cfg::BLOCK);
        __syncthreads();
    } else {
        load_twiddles<L, Cfg::LTW>(twr, a.twL, g);
        twr.ltab = ltw;
        if constexpr (Cfg::LTW) {
            fill_lds_twiddles<L>(ltw, a.twL, tid, Cfg::BLOCK);
            __syncthreads();
        }
    }
    // one transform of this team on v; store(n, value, slot, nu) gets natural output n = nu + g; only the outputs of the first
    // `live` register slots are wanted (the fused transforms skip the dead halves of their last butterflies)
    auto transform = [&](cf (&vv)[PPL], auto &store, auto live) {
        constexpr int LIVE = decltype(live)::value;
        if constexpr (F256) fft256_fused<0, LIVE>(vv, mylds, g, f256, store);
        else if constexpr (F2048) fft_w32_fused<0, LIVE>(vv, mylds, lane, f2048, store);
        else if constexpr (W32) fft_w32(vv, mylds, lane, w32, store);
        else fft_passes<L, 1, 0, true, Cfg::PP, Cfg::HALF, SYNC>(vv, mylds, ebuf, g, 0, twr, a.twL, store);
    };
    constexpr std::integral_constant<int, PPL> all_live{};
    constexpr std::integral_constant<int, (MASKED || MODE != SEG_REDUCE) ? PPL : PV> sum_live{};

    // ---- which Doppler bins and which slots this team owns -----------------------------------
    int grp, mg, bstream, ssub;
    {
        const int b = blk;
        grp = b % a.nsg;
        int r = b / a.nsg;
        mg = r % a.mgroups;
        r /= a.mgroups;
        if constexpr (SYNC) {            // independent waves: neighbours take neighbouring Doppler streams
            const int tt = r * TPW + team;
            bstream = tt % a.bsplit;
            ssub = tt / a.bsplit;
        } else {                         // barrier teams of one workgroup share the stream (equal trip counts)
            bstream = r % a.bsplit;
            ssub = (r / a.bsplit) * TPW + team;
        }
    }
    const int gs0 = (int)((long long)grp * a.nslots / a.nsg);
    const int glen = (int)((long long)(grp + 1) * a.nslots / a.nsg) - gs0;
    const int s0 = a.slot0 + gs0 + (int)((long long)ssub * glen / a.ssplit);
    const int s1 = a.slot0 + gs0 + (int)((long long)(ssub + 1) * glen / a.ssplit);
    // barrier teams run the longest range of the group (surplus iterations contribute nothing)
    const int niter = __builtin_amdgcn_readfirstlane(SYNC ? (s1 - s0) : (glen + a.ssplit - 1) / a.ssplit);
    // filter groups: one per workgroup (mg from the grid), or -- Doppler search with several groups -- all of them in turn on the
    // forward spectrum of the segment, which is then computed once (the BPSK bank: 16 unique filters, 8 per pass)
    // Only the 256-point kernel has the loop (the bank that needs it, BPSK, has 80 taps): around the longer transforms the extra
    // loop level makes the compiler serialise the spectrum loads behind s_waitcnt vmcnt(0) (CC11xx: 3.04 against 2.49 ms).
    constexpr bool GROUP_LOOP = MODE == SEG_REDUCE && L <= 256;
    const int ngrp = GROUP_LOOP ? __builtin_amdgcn_readfirstlane(a.igroups > 1 ? a.igroups : 1) : 1;

    const unsigned nmask = (unsigned)a.N - 1u;
    const unsigned lomask = (1u << a.lo) - 1u;
    const auto lor = mk_rsrc(a.twLo, (unsigned)(1u << a.lo) * sizeof(cf));
    const auto hir = mk_rsrc(a.twHi, (unsigned)(a.N >> a.lo) * sizeof(cf));
    const auto gr = mk_rsrc(a.G, (unsigned)a.Grows * (unsigned)(L * sizeof(cf)));   // whole G; rows via the scalar offset
    constexpr int so_g = NT * (int)sizeof(cf);
    const int vo_g2 = g * 2 * (int)sizeof(cf);          // slot-pair layout of G
    constexpr int so_g2 = NT * 2 * (int)sizeof(cf);
    auto phasor = [&](unsigned t) {
        return cmul(buf_load_cf(hir, (int)((t >> a.lo) * sizeof(cf)), 0), buf_load_cf(lor, (int)((t & lomask) * sizeof(cf)), 0));
    };

    for (int jl = bstream; jl < a.dc; jl += a.bsplit) {
        // (wave-uniform: bstream comes from the workgroup index and the wave's team)
        const int bk = a.dper ? jl / a.dper : 0;               // block of the batch
        const int jb = jl - bk * a.dper;                       // = jl for a single block
        const int shift = a.shifts ? a.shifts[a.j0 + jb + bk * a.sstride] : a.fixed_shift;
        const auto xr = mk_rsrc(a.x + (size_t)bk * (size_t)a.xstride, (unsigned)a.N * sizeof(cf));
        [[maybe_unused]] cf pg = mkc(1.f, 0.f);
        if constexpr (PTAB) {
            // relative mixing phasors of all L positions of a segment, shared by the wave's columns
#pragma unroll
            for (int r = 0; r < L / 64; ++r) myphase[lane + 64 * r] = phasor(((unsigned)shift * (unsigned)(lane + 64 * r)) & nmask);
        } else {
            // step phasors W_N^(shift * NT * i), i < PPL: the same for every lane, segment and filter of this bin
            if (lane < PPL) mystep[lane] = phasor(((unsigned)shift * (unsigned)(NT * lane)) & nmask);
            if constexpr (REL) pg = phasor(((unsigned)shift * (unsigned)g) & nmask);
        }
        xsync<1>();     // wave-local: every wave fills and reads its own copy

        // x of a slot -> 16 registers (immediate offsets unless the slot wraps past the end of the block)
        auto load_x = [&](cf (&dst)[PPL], int slot_) {
            const unsigned e0_ = (unsigned)(slot_ * CT + col) * (unsigned)a.V + (unsigned)g;
            const unsigned last = (unsigned)(slot_ * CT + CT - 1) * (unsigned)a.V + (unsigned)L;   // team-uniform
            if (__builtin_amdgcn_readfirstlane(last <= (unsigned)a.N ? 1 : 0)) {
                const int vo_x = (int)(e0_ * sizeof(cf));
#pragma unroll
                for (int i = 0; i < PPL; ++i) dst[i] = buf_load_cf(xr, vo_x, i * so_g);
            } else {
#pragma unroll
                for (int i = 0; i < PPL; ++i) dst[i] = buf_load_cf(xr, (int)(((e0_ + (unsigned)(NT * i)) & nmask) * sizeof(cf)), 0);
            }
        };
        auto load_g = [&](cf (&dst)[PPL], int row) {
#pragma unroll
            for (int ii = 0; ii < PPL / 2; ++ii) buf_load_cf2(gr, vo_g2, row * (L * (int)sizeof(cf)) + ii * so_g2, dst[2 * ii], dst[2 * ii + 1]);
        };
        // One set of 16 prefetch registers (MFB_SEG_PREFETCH) receives the first filter's spectrum while the forward
        // transform runs and the next filter's spectrum during every inverse transform.  (Also tried, one MI355X,
        // A/B of two builds: the next slot's x in the same registers during the last filter -- no gain, 20 spilled
        // VGPRs at the 3-wave budget; the next filter's product in the shadow of the LDS exchange -- 7 % slower than
        // the compiler's own schedule.)
        // The fused 2048-point kernel (round 6: 212 instead of 254 VGPRs) has room for HALF of the next filter's spectrum: register
        // slots 0 ... 15 are fetched while the running transform computes, slots 16 ... 31 at the top of the product, behind which
        // the first half's multiplies hide part of their latency (MFB_SEG_PREHALF; profiles/r06_fft_ops.md)
        constexpr bool PREHALF = F2048 && MFB_SEG_PREHALF && !MASKED && MODE == SEG_REDUCE && !Cfg::PREFETCH;
        [[maybe_unused]] cf gk[Cfg::PREFETCH ? PPL : (PREHALF ? PPL / 2 : 1)];
        [[maybe_unused]] auto load_g_half = [&](cf (&dst)[PPL / 2], int row, auto first) {
            constexpr int I0 = decltype(first)::value;
#pragma unroll
            for (int ii = 0; ii < PPL / 4; ++ii)
                buf_load_cf2(gr, vo_g2, row * (L * (int)sizeof(cf)) + (I0 + ii) * so_g2, dst[2 * ii], dst[2 * ii + 1]);
        };

        for (int it = 0; it < niter; ++it) {
            const int slot = s0 + it;
            const bool active = slot < s1;        // team-uniform; false only on padding iterations of barrier teams
            const int seg = slot * CT + col;
            const unsigned b0 = (unsigned)seg * (unsigned)a.V;
            const unsigned e0 = b0 + (unsigned)g;

            // ---- the segment, mixed with e^{-2 pi i s n / N} and conjugated (forward via inverse) ----
            cf v[PPL];
            {
                load_x(v, slot);
                // conj(x) * W_N^{+t} = conj(x * e^{-2 pi i t / N})
                if constexpr (PTAB) {
#pragma unroll
                    for (int i = 0; i < PPL; ++i) v[i] = cmul_cj(v[i], myphase[g + NT * i]);
                } else {
                    cf ph0 = pg;
                    if constexpr (!REL) ph0 = phasor(((unsigned)shift * e0) & nmask);   // N | 2^32: wrap-around is harmless
                    v[0] = cmul_cj(v[0], ph0);
#pragma unroll
                    for (int i = 1; i < PPL; ++i) v[i] = cmul_cj(v[i], cmul(ph0, mystep[i]));
                }
            }
            const int mfirst = (ngrp > 1 ? 0 : mg) * a.mpb;
            const int r0 = a.rows ? a.rows[mfirst] : mfirst;
            if constexpr (Cfg::PREFETCH && MFB_SEG_G0EARLY) load_g(gk, r0);    // lands while the forward transform runs
            if constexpr (PREHALF) load_g_half(gk, r0, std::integral_constant<int, 0>{});
            cf A[PPL];                            // A[k] = conj(U[g + NT*k])
            {
                auto keep = [&](int, cf val, auto, auto nu) { A[decltype(nu)::value / NT] = val; };
                if constexpr (Cfg::DUAL) xsync<SYNC>();        // the previous slot's last pair may still be read by other waves
                transform(v, keep, all_live);
            }
            if constexpr (Cfg::PREFETCH && !MFB_SEG_G0EARLY) load_g(gk, r0);

            // ---- per filter: multiply by the segment spectrum, inverse transform, reduce or store ----
            [[maybe_unused]] int lim = 0;
            if constexpr (MASKED) {
                const int vseg = active ? min(a.V, a.N - (int)b0) : 0;     // <= 0 beyond the last segment
                lim = vseg - g;                   // output slot k is valid for this lane iff k*NT < lim
            }
            [[maybe_unused]] float envacc[PPL];
            if constexpr (MODE == SEG_STORE) {
#pragma unroll
                for (int k = 0; k < PPL; ++k) envacc[k] = 0.f;
            }
            for (int gi = 0; gi < ngrp; ++gi) {
            const int m0 = (ngrp > 1 ? gi : mg) * a.mpb;
            const int nm = __builtin_amdgcn_readfirstlane(min(a.mpb, a.MU - m0));
            if constexpr (Cfg::PREFETCH) {
                if (gi > 0) load_g(gk, a.rows ? a.rows[m0] : m0);
            }
            int mi_first = 0;
            if constexpr (Cfg::DUAL && MODE == SEG_REDUCE) {
                for (; mi_first + 1 < nm; mi_first += 2) {
                    const int rA = a.rows ? a.rows[m0 + mi_first] : (m0 + mi_first);
                    const int rB = a.rows ? a.rows[m0 + mi_first + 1] : (m0 + mi_first + 1);
                    cf wA[16], wB[16];
#pragma unroll
                    for (int ii = 0; ii < 8; ++ii) {
                        cf g0, g1, h0, h1;
                        buf_load_cf2(gr, vo_g2, rA * (L * (int)sizeof(cf)) + ii * so_g2, g0, g1);
                        buf_load_cf2(gr, vo_g2, rB * (L * (int)sizeof(cf)) + ii * so_g2, h0, h1);
                        wA[2 * ii] = cmul_cj(A[2 * ii], g0);
                        wA[2 * ii + 1] = cmul_cj(A[2 * ii + 1], g1);
                        wB[2 * ii] = cmul_cj(A[2 * ii], h0);
                        wB[2 * ii + 1] = cmul_cj(A[2 * ii + 1], h1);
                    }
                    cf raccA[4] = {mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f)};
                    cf raccB[4] = {mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f)};
                    auto accA = [&](int, cf val, auto, auto nu) {
                        constexpr int k = decltype(nu)::value / NT;
                        if constexpr (MASKED) {
                            const float wgt = (k * NT < lim) ? 1.f : 0.f;
                            raccA[k & 3] = __builtin_elementwise_fma(val * wgt, val, raccA[k & 3]);
                        } else if constexpr (k < PV) {
                            raccA[k & 3] = __builtin_elementwise_fma(val, val, raccA[k & 3]);
                        }
                    };
                    auto accB = [&](int, cf val, auto, auto nu) {
                        constexpr int k = decltype(nu)::value / NT;
                        if constexpr (MASKED) {
                            const float wgt = (k * NT < lim) ? 1.f : 0.f;
                            raccB[k & 3] = __builtin_elementwise_fma(val * wgt, val, raccB[k & 3]);
                        } else if constexpr (k < PV) {
                            raccB[k & 3] = __builtin_elementwise_fma(val, val, raccB[k & 3]);
                        }
                    };
                    fft_passes2<L, 1, 0, true, Cfg::HALF, SYNC>(wA, wB, mylds, g, 0, twr, a.twL, accA, accB);
                    const cf sA = (raccA[0] + raccA[1]) + (raccA[2] + raccA[3]);
                    const cf sB = (raccB[0] + raccB[1]) + (raccB[2] + raccB[3]);
                    lacc[mi_first * SEG_ACC_STRIDE + lane] = sA.x + sA.y;
                    lacc[(mi_first + 1) * SEG_ACC_STRIDE + lane] = sB.x + sB.y;
                }
                if (mi_first < nm) xsync<SYNC>();          // an odd filter follows on the first buffer
            }
            for (int mi = mi_first; mi < nm; ++mi) {
                const int rm = a.rows ? a.rows[m0 + mi] : (m0 + mi);
                cf w[PPL];
                if constexpr (Cfg::PREFETCH) {
#pragma unroll
                    for (int i = 0; i < PPL; ++i) w[i] = cmul_cj(A[i], gk[i]);    // conj(A) * G = U * G
                    if (mi + 1 < nm) load_g(gk, a.rows ? a.rows[m0 + mi + 1] : (m0 + mi + 1));
                } else if constexpr (PREHALF) {
                    cf gh[PPL / 2];
                    load_g_half(gh, rm, std::integral_constant<int, PPL / 4>{});
#pragma unroll
                    for (int i = 0; i < PPL / 2; ++i) w[i] = cmul_cj(A[i], gk[i]);
                    if (mi + 1 < nm) load_g_half(gk, a.rows ? a.rows[m0 + mi + 1] : (m0 + mi + 1), std::integral_constant<int, 0>{});
#pragma unroll
                    for (int i = 0; i < PPL / 2; ++i) w[PPL / 2 + i] = cmul_cj(A[PPL / 2 + i], gh[i]);
                } else {
#pragma unroll
                    for (int ii = 0; ii < PPL / 2; ++ii) {
                        cf g0, g1;
#ifdef MFB_SEG_NOG     // timing-only variant (wrong numbers): no spectrum loads at all = the bound of a perfect prefetch
                        g0 = A[(2 * ii + 1) % PPL];
                        g1 = A[(2 * ii + 2) % PPL];
#else
                        buf_load_cf2(gr, vo_g2, rm * (L * (int)sizeof(cf)) + ii * so_g2, g0, g1);
#endif
                        w[2 * ii] = cmul_cj(A[2 * ii], g0);
                        w[2 * ii + 1] = cmul_cj(A[2 * ii + 1], g1);
                    }
                }
                if constexpr (MODE == SEG_REDUCE) {
                    // (sum re^2, sum im^2) in four independent chains: back-to-back dependent packed ops cost a
                    // wait state each (the compiler pads them with s_nop)
                    cf racc[4] = {mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f)};
                    auto acc = [&](int, cf val, auto, auto nu) {
                        constexpr int k = decltype(nu)::value / NT;
                        if constexpr (MASKED) {
                            const float wgt = (k * NT < lim) ? 1.f : 0.f;
                            racc[k & 3] = __builtin_elementwise_fma(val * wgt, val, racc[k & 3]);
                        } else if constexpr (k < PV) {
                            racc[k & 3] = __builtin_elementwise_fma(val, val, racc[k & 3]);
                        }
                    };
                    transform(w, acc, sum_live);
                    const cf rsum = (racc[0] + racc[1]) + (racc[2] + racc[3]);
                    lacc[mi * SEG_ACC_STRIDE + lane] = rsum.x + rsum.y;       // reduced over the lanes after the last filter
                } else {
                    const auto orr = mk_rsrc(a.out + (size_t)bk * (size_t)a.ostride + (size_t)rm * a.N, (unsigned)a.N * sizeof(cf));
                    const unsigned o0 = e0 + (unsigned)a.out_off;
                    const bool in_env = a.env && rm >= a.env_lo && rm < a.env_hi;      // team-uniform
                    auto put = [&](int, cf val, auto, auto nu) {
                        constexpr int k = decltype(nu)::value / NT;
                        if (k * NT < lim) buf_store_cf(orr, (int)(((o0 + (unsigned)(k * NT)) & nmask) * sizeof(cf)), 0, val);
                        // same fp32 order as k_envelope: s = s + fma(re, re, im * im), filters ascending
                        if (in_env) envacc[k] = __fadd_rn(envacc[k], __fmaf_rn(val.x, val.x, __fmul_rn(val.y, val.y)));
                    };
                    transform(w, put, all_live);
                }
            }
            if constexpr (MODE == SEG_REDUCE) {
                // One partial per (bin, filter, slot, wave of the team), at an index that depends on the slot alone, summed
                // over the wave's lanes in ONE fixed order: lane (f, j) = (lane / 4, lane % 4) adds elements j, j+4, ... of
                // filter f's row as a balanced tree, two quad steps finish.  k_finalize adds the slots in a fixed order.
                // A score is therefore a function of the block, the shift and the filter only -- not of how many bins the
                // handle holds, of the grid decomposition or of the tuning: shards of any size reproduce the unsharded
                // table bit for bit.  (All filters of the pass share the 16 + 2 steps; a butterfly per filter cost 8-10 %.)
                xsync<1>();
                const int f = lane >> 2, j = lane & 3;
                const float *row = lacc + min(f, nm - 1) * SEG_ACC_STRIDE + j;
                float t[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = row[4 * k];
                float s = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                          (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
                s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                if (j == 0 && f < nm && active)
                    a.partials[((size_t)(a.part_row0 + jl) * a.MU + (m0 + f)) * a.parts + (slot * Cfg::WPT + (lt >> 6))] = s * a.scale;
                xsync<1>();     // the rows are rewritten by the next group / slot
            }
            }       // filter groups
            if constexpr (MODE == SEG_STORE) {
                if (a.env) {
                    const unsigned o0 = e0 + (unsigned)a.out_off;
#pragma unroll
                    for (int k = 0; k < PPL; ++k)
                        if (k * NT < lim) a.env[(size_t)bk * (size_t)a.N + ((o0 + (unsigned)(k * NT)) & nmask)] = envacc[k];
                }
            }
        }
        xsync<1>();     // the step phasors are rewritten for the next bin
    }
}

// ---- the search with the Doppler shift on the FILTER side (round 6; 256- and wave-local 2048-point segments) --------------------------------------------
// seg_body mixes the SAMPLES of a segment with the bin's frequency and transforms them once per (bin, segment): 1 forward + MU
// inverse transforms and a phasor multiply per point.  The search only needs |y|^2, and the shift can sit on the filter instead
// (filter_taps.hpp, segment_spectra_shifted): the segment is transformed ONCE -- unmixed -- for all bins a wave takes it through,
// and every (bin, filter) brings spectra of its own, Gs[bin][filter] (D * MU * 2 KiB: 4 MiB at C2, served by L2).  Per (bin, slot):
// MU x (product + inverse transform + |.|^2) and nothing else -- 1/9 of the transforms and all the mixing multiplies of the
// 8-filter banks gone, for L2 reads of 2 KiB per (slot, bin, filter) that used to hit in L1.
//   * A wave owns a rectangle: `fb` neighbouring bins x `fs` slots of one block (a slot = the four segments it carries side by
//     side), slots outside, bins inside; the grid deals rectangles XCD-aware as seg_body does (blockIdx % nsg = the eighth of
//     the block whose samples stay in that XCD's L2).
//   * The forward transform runs as the inverse one on swapped samples: (im, re) = i conj(x), so the registers hold i conj(U),
//     and conj(.) G = -i U G -- a constant unit factor, gone in |.|^2.  The swap is a renaming of registers, not an instruction.
//   * The same partial sums at the same indices in the same lane order as seg_body (k_finalize does not know which kernel ran):
//     a score is still a function of the block, the shift and the filter only.
struct SegFArgs {
    const cf *x;         // block (or window of a batch)
    const cf *Gs;        // [Dtot][MU][PPL / 2][NT][2]: spectra of filter slot u at bin j, slot-pair layout
    const float *Qs;     // [Dtot][PPL / 4][NT][4]: sum over the filter slots of |Gs|^2 at bin j, weighted by how often k_finalize counts
                         // each slot relative to slot 0 (SUM_ALL searches, SUMQ instantiations; else unused)
    const cf *twL;       // W_L table with the fused (cos, tan) pairs behind it
    float *partials;
    int N, V;
    int slot0, nslots;   // slots [slot0, slot0 + nslots) of every block
    int MU;              // filter slots per bin (<= SEG_MPB_MAX)
    int dper;            // bins per block (Dtot)
    int nblk, xstride;   // blocks in the launch; block b's samples start at x + b * xstride
    int nsg;             // groups (blockIdx % nsg = the XCD under round-robin placement)
    int gbins;           // 1: a group is an nsg-th of the BINS (its share of Gs stays in that XCD's L2, the samples stream through
                         // every XCD); 0: an nsg-th of the slots (the samples stay, all of Gs passes through every L2)
    int fb, fs;          // bins / slots per rectangle
    int nbc, nsc;        // bin chunks per block (gbins: per group), slot chunks per group (gbins: per block)
    int part_row0, parts;
    int presum;          // SUMQ searches: the filter rows count equally, one partial sum per (bin, slot) in row 0 (SUMQ = 2)
    float scale;
};
// SUMQ: 0 = the Parseval total per (bin, filter); 1 = once per bin, one partial sum per filter row; 2 = once per bin and the rows summed in
// registers (256-point form, rows that count equally)
template <int L, int PV, int SUMQ>
DEVI void segf_body(const SegFArgs &a, const int blk) {
    static_assert(MFB_FFT_FUSED && (L == 256 || (L == 2048 && MFB_SEG_W32)), "the fused 256- and 2048-point transforms");
    using Cfg = SegCfg<L>;
    constexpr int NT = Cfg::NT, CT = Cfg::CT, PPL = Cfg::PPL;
    constexpr bool W32 = Cfg::W32;
    extern __shared__ __attribute__((aligned(16))) cf lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int g = W32 ? ((lane >> 1) + 32 * (lane & 1)) : lane % NT;
    const int col = W32 ? 0 : lane / NT;
    cf *mylds = lds + wave * Cfg::LDS_PER_TEAM + col * padlen(L);
    float *lacc = reinterpret_cast<float *>(lds + Cfg::LDS_ELEMS) + wave * (a.MU * SEG_ACC_STRIDE);
    [[maybe_unused]] F256Regs f256;
    [[maybe_unused]] F2048Regs f2048;
    if constexpr (W32) f2048_setup(f2048, a.twL + L, lane);
    else f256_setup(f256, a.twL + L, g);
    auto transform = [&](cf (&vv)[PPL], auto &store, auto lo, auto hi) {          // only the outputs of register slots [lo, hi) are wanted
        constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
        if constexpr (W32) fft_w32_fused<LO, HI>(vv, mylds, lane, f2048, store);
        else fft256_fused<LO, HI>(vv, mylds, g, f256, store);
    };
    // The energy of the VALID outputs of a segment as the energy of ALL its outputs minus that of the invalid ones: sum_{i < L} |v[i]|^2
    // = L sum_k |W[k]|^2 (Parseval; W = the product the inverse transform starts from: PPL packed FMAs per lane, no transform), and the
    // invalid outputs -- register slots PV ... PPL - 1 of every lane -- are few: with 13 of 16 slots valid the second pass computes 3
    // outputs per lane instead of 13 (62 instead of 93 packed operations, 3 instead of 13 for the squares, 16 for Parseval: 81
    // instead of 106 per filter; 210 instead of 260 in the 2048-point form at 26 of 32).  Worth it from 11 (22) valid slots up.
    // Same number to fp32 rounding: the difference is 4/5 of the minuend, nothing cancels (profiles/r06_fft_ops.md section 4).
    constexpr bool COMPL = MFB_SEG_COMPLEMENT && PV < PPL && (W32 ? PV >= 22 : PV >= 11);
    // SUM_ALL searches need the sum over the filters only, and the Parseval half of it factors: sum_f L sum_k |A[k] G_f[k]|^2 =
    // L sum_k |A[k]|^2 Q[k] with Q = sum_f |G_f|^2, a table per bin (host, filter_taps.hpp): PPL / 2 packed FMAs per BIN instead of
    // PPL per (bin, filter).  The bin's total goes into filter slot 0's partial sum, the other slots carry minus their invalid
    // outputs' energy; k_finalize adds the slots as it always did (Q carries the weights with which it counts them).
    constexpr bool QSUM = SUMQ != 0 && COMPL;
    constexpr bool PRESUM = QSUM && SUMQ == 2;

    // ---- this wave's rectangle ----
    const int grp = blk % a.nsg;
    const int u = (blk / a.nsg) * (Cfg::BLOCK / 64) + wave;        // rectangle index inside the group: bin chunk fastest
    const int per_sc = a.nblk * a.nbc;
    const int sc = u / per_sc, rest = u - sc * per_sc;
    const int bk = rest / a.nbc, bc = rest - bk * a.nbc;
    // the group's share: of the slots (all bins), or of the bins (all slots)
    const int gs0 = a.gbins ? 0 : (int)((long long)grp * a.nslots / a.nsg);
    const int glen = a.gbins ? a.nslots : (int)((long long)(grp + 1) * a.nslots / a.nsg) - gs0;
    const int gb0 = a.gbins ? (int)((long long)grp * a.dper / a.nsg) : 0;
    const int gblen = a.gbins ? (int)((long long)(grp + 1) * a.dper / a.nsg) - gb0 : a.dper;
    const int s0 = __builtin_amdgcn_readfirstlane(a.slot0 + gs0 + sc * a.fs);
    const int s1 = __builtin_amdgcn_readfirstlane(min(a.slot0 + gs0 + glen, s0 + a.fs));
    const int jb0 = __builtin_amdgcn_readfirstlane(gb0 + bc * a.fb);
    const int jb1 = __builtin_amdgcn_readfirstlane(min(gb0 + gblen, jb0 + a.fb));
    if (sc >= a.nsc || s0 >= s1 || jb0 >= jb1 || bk >= a.nblk) return;           // (wave-uniform)

    const unsigned nmask = (unsigned)a.N - 1u;
    const auto xr = mk_rsrc(a.x + (size_t)bk * (size_t)a.xstride, (unsigned)a.N * sizeof(cf));
    const auto gr = mk_rsrc(a.Gs, (unsigned)a.dper * (unsigned)a.MU * (unsigned)(L * sizeof(cf)));
    [[maybe_unused]] const auto qr = mk_rsrc(QSUM ? (const void *)a.Qs : (const void *)a.Gs, (unsigned)a.dper * (unsigned)(L * sizeof(float)));
    [[maybe_unused]] auto load_q = [&](cf (&dst)[PPL / 2], int bin) {
#pragma unroll
        for (int jj = 0; jj < PPL / 4; ++jj)
            buf_load_cf2(qr, g * 4 * (int)sizeof(float), bin * (L * (int)sizeof(float)) + jj * NT * 4 * (int)sizeof(float), dst[2 * jj], dst[2 * jj + 1]);
    };
    constexpr int so_x = NT * (int)sizeof(cf);
    const int vo_g2 = g * 2 * (int)sizeof(cf);
    constexpr int so_g2 = NT * 2 * (int)sizeof(cf);
    // (one set of prefetch registers for the next filter's spectra, as in seg_body -- the 2048-point form has no room for it and
    // fetches inside the product)
    constexpr bool PREFETCH = !W32;
    [[maybe_unused]] auto load_g = [&](cf (&dst)[PPL], int row) {
#if MFB_SEG_FSM_PROBE_ONE_ROW        // measurement only (wrong scores): every (bin, filter) reads row 0 -- what the L2 reads of Gs cost
        row = 0;
#endif
#pragma unroll
        for (int ii = 0; ii < PPL / 2; ++ii) buf_load_cf2(gr, vo_g2, row * (L * (int)sizeof(cf)) + ii * so_g2, dst[2 * ii], dst[2 * ii + 1]);
    };
    const int MU = __builtin_amdgcn_readfirstlane(a.MU);
    constexpr std::integral_constant<int, PPL> all_live{};
    constexpr std::integral_constant<int, PV> sum_live{};
    [[maybe_unused]] cf gk[PREFETCH ? PPL : 1];

    for (int slot = s0; slot < s1; ++slot) {
        // ---- the segments of the slot, unmixed, swapped: the forward transform via the inverse one ----
        cf v[PPL];
        {
            const unsigned e0 = (unsigned)(slot * CT + col) * (unsigned)a.V + (unsigned)g;
            const unsigned last = (unsigned)(slot * CT + CT - 1) * (unsigned)a.V + (unsigned)L;      // wave-uniform
            if (__builtin_amdgcn_readfirstlane(last <= (unsigned)a.N ? 1 : 0)) {
                const int vo_x = (int)(e0 * sizeof(cf));
#pragma unroll
                for (int i = 0; i < PPL; ++i) {
                    const cf t = buf_load_cf(xr, vo_x, i * so_x);
                    v[i] = mkc(t.y, t.x);
                }
            } else {
#pragma unroll
                for (int i = 0; i < PPL; ++i) {
                    const cf t = buf_load_cf(xr, (int)(((e0 + (unsigned)(NT * i)) & nmask) * sizeof(cf)), 0);
                    v[i] = mkc(t.y, t.x);
                }
            }
        }
        if constexpr (PREFETCH) load_g(gk, jb0 * MU);      // lands while the forward transform runs
        cf A[PPL];                                         // A[k] = i conj(U[g + NT k])
        {
            auto keep = [&](int, cf val, auto, auto nu) { A[decltype(nu)::value / NT] = val; };
            transform(v, keep, std::integral_constant<int, 0>{}, all_live);
        }
        // (|A[2j]|^2, |A[2j + 1]|^2): kept per slot by the 256-point form; the 2048-point form has no 32 registers for them and squares
        // again for every bin (80 instead of 16 operations per bin, against the 32 per (bin, filter) they replace)
        constexpr bool PP_REGS = QSUM && !W32;
        [[maybe_unused]] cf pp[PP_REGS ? PPL / 2 : 1];
        [[maybe_unused]] cf qq[QSUM ? PPL / 2 : 1];
        auto power_pair = [&](int j) {
            const cf s0 = A[2 * j] * A[2 * j], s1 = A[2 * j + 1] * A[2 * j + 1];
            return mkc(s0.x + s0.y, s1.x + s1.y);
        };
        if constexpr (QSUM) {
            if constexpr (PP_REGS) {
#pragma unroll
                for (int j = 0; j < PPL / 2; ++j) pp[j] = power_pair(j);
            }
            if constexpr (PP_REGS) load_q(qq, jb0);
        }
        for (int jb = jb0; jb < jb1; ++jb) {
            [[maybe_unused]] float tot = 0.f;
            if constexpr (QSUM) {
                if constexpr (!PP_REGS) load_q(qq, jb);            // (no registers to hold the next bin's table across the reduce)
                cf t2[2] = {mkc(0.f, 0.f), mkc(0.f, 0.f)};
#pragma unroll
                for (int j = 0; j < PPL / 2; ++j) {
                    if constexpr (PP_REGS) t2[j & 1] = __builtin_elementwise_fma(pp[j], qq[j], t2[j & 1]);
                    else t2[j & 1] = __builtin_elementwise_fma(power_pair(j), qq[j], t2[j & 1]);
                }
                tot = (t2[0].x + t2[0].y) + (t2[1].x + t2[1].y);
            }
            [[maybe_unused]] float binacc = 0.f;        // minus the invalid outputs' energy, summed over the bin's filters (PRESUM)
            for (int mi = 0; mi < MU; ++mi) {
                cf w[PPL];
                if constexpr (PREFETCH) {
#pragma unroll
                    for (int i = 0; i < PPL; ++i) w[i] = cmul_cj(A[i], gk[i]);
                    // the next filter's spectra -- or the next bin's first -- during this transform
                    const int nxt = jb * MU + mi + 1;
                    if (nxt < jb1 * MU) load_g(gk, nxt);
                } else {
                    const int row = jb * MU + mi;
#pragma unroll
                    for (int ii = 0; ii < PPL / 2; ++ii) {
                        cf g0, g1;
                        buf_load_cf2(gr, vo_g2, row * (L * (int)sizeof(cf)) + ii * so_g2, g0, g1);
                        w[2 * ii] = cmul_cj(A[2 * ii], g0);
                        w[2 * ii + 1] = cmul_cj(A[2 * ii + 1], g1);
                    }
                }
                cf racc[4] = {mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f)};
                if constexpr (QSUM) {
                    auto acc = [&](int, cf val, auto, auto nu) {
                        constexpr int k = decltype(nu)::value / NT;
                        if constexpr (k >= PV) racc[k & 3] = __builtin_elementwise_fma(val, val, racc[k & 3]);
                    };
                    transform(w, acc, sum_live, all_live);                 // slots [PV, PPL): the invalid outputs
                    const cf inv = (racc[0] + racc[1]) + (racc[2] + racc[3]);
                    if constexpr (PRESUM) binacc -= inv.x + inv.y;
                    else lacc[mi * SEG_ACC_STRIDE + lane] = __builtin_fmaf((float)L, mi == 0 ? tot : 0.f, -(inv.x + inv.y));
                } else if constexpr (COMPL) {
                    cf pacc[4] = {mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f), mkc(0.f, 0.f)};
#pragma unroll
                    for (int i = 0; i < PPL; ++i) pacc[i & 3] = __builtin_elementwise_fma(w[i], w[i], pacc[i & 3]);
                    auto acc = [&](int, cf val, auto, auto nu) {
                        constexpr int k = decltype(nu)::value / NT;
                        if constexpr (k >= PV) racc[k & 3] = __builtin_elementwise_fma(val, val, racc[k & 3]);
                    };
                    transform(w, acc, sum_live, all_live);                 // slots [PV, PPL): the invalid outputs
                    const cf all = (pacc[0] + pacc[1]) + (pacc[2] + pacc[3]);
                    const cf inv = (racc[0] + racc[1]) + (racc[2] + racc[3]);
                    lacc[mi * SEG_ACC_STRIDE + lane] = __builtin_fmaf((float)L, all.x + all.y, -(inv.x + inv.y));
                } else {
                    auto acc = [&](int, cf val, auto, auto nu) {
                        constexpr int k = decltype(nu)::value / NT;
                        if constexpr (k < PV) racc[k & 3] = __builtin_elementwise_fma(val, val, racc[k & 3]);
                    };
                    transform(w, acc, std::integral_constant<int, 0>{}, sum_live);
                    const cf rsum = (racc[0] + racc[1]) + (racc[2] + racc[3]);
                    lacc[mi * SEG_ACC_STRIDE + lane] = rsum.x + rsum.y;
                }
            }
            if constexpr (PP_REGS) {
                if (jb + 1 < jb1) load_q(qq, jb + 1);              // lands during the reduce below
            }
            if constexpr (PRESUM) {
                // the wave's 64 values in a fixed order: quads, the four quads of a row, the four rows
                float sv = __builtin_fmaf((float)L, tot, binacc);
                sv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sv), 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
                sv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sv), 0x4E, 0xF, 0xF, true));       // quad_perm [2,3,0,1]
                sv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sv), 0x124, 0xF, 0xF, true));      // row_ror:4
                sv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sv), 0x128, 0xF, 0xF, true));      // row_ror:8
                const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), 0));
                const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), 16));
                const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), 32));
                const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), 48));
                const float total = (r0 + r1) + (r2 + r3);
                // row 0 carries the bin's sum, the other rows zero: k_finalize adds the rows as it always did
                if (lane < MU)
                    a.partials[((size_t)(a.part_row0 + bk * a.dper + jb) * MU + lane) * a.parts + slot] = lane == 0 ? total * a.scale : 0.f;
                continue;
            }
            // the wave's lanes in seg_body's fixed order: lane (f, j) adds elements j, j + 4, ... of filter f's row, two quad steps
            xsync<1>();
            const int f = lane >> 2, j = lane & 3;
            const float *row = lacc + min(f, MU - 1) * SEG_ACC_STRIDE + j;
            float t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = row[4 * k];
            float sm = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                       (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
            sm += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sm), 0xB1, 0xF, 0xF, true));
            sm += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sm), 0x4E, 0xF, 0xF, true));
            if (j == 0 && f < MU)
                a.partials[((size_t)(a.part_row0 + bk * a.dper + jb) * MU + f) * a.parts + slot] = sm * a.scale;
            xsync<1>();
        }
    }
}
template <int L, int PV, int SUMQ = 0>
__global__ void __launch_bounds__(SegCfg<L>::BLOCK) __attribute__((amdgpu_waves_per_eu(SegCfg<L>::WAVES, SegCfg<L>::WAVES)))
k_segf(SegFArgs a) {
    segf_body<L, PV, SUMQ>(a, (int)blockIdx.x);
}
// which (L, PV) have a SUMQ form worth instantiating: those that take the complement branch
template <int L, int PV>
constexpr bool segf_has_sumq() {
    return MFB_SEG_SUMQ && MFB_SEG_COMPLEMENT && PV < SegCfg<L>::PPL && (SegCfg<L>::W32 ? PV >= 22 : PV >= 11);
}
// ... and the rows summed in registers: the 256-point form (the 2048-point one has no registers left for it: spills, measured 0.3 % slower)
template <int L, int PV>
constexpr bool segf_has_presum() {
    return MFB_SEG_PRESUM && segf_has_sumq<L, PV>() && !SegCfg<L>::W32;
}

// amdgpu_waves_per_eu pins the register budget: without the upper bound the scheduler chases a fourth
// wave per SIMD (which the LDS footprint does not admit anyway) by serialising every load behind an
// s_waitcnt vmcnt(0).
// The masked instantiations (tail of the search, demodulation) carry the per-lane limits and are not on the
// critical path: they get the 2-wave budget (no spills).
// (the masked forms of the wave-local 2048-point kernel hold 32 points, the spectrum, 31 twiddles AND the per-lane limits:
// they get the whole register file -- one wave per SIMD -- instead of spilling 50 registers at the two-wave budget)
// (so do the masked forms of the 4096-point kernel -- four-wave teams, one wave per SIMD is enough: STORE needed 260 registers and
// spilled four of them to scratch at the two-wave budget; the 8192-point kernel's eight-wave team needs two waves per SIMD to
// fit a CU at all, its masked REDUCE form keeps 26 registers in scratch)
template <int L, int PV>
struct SegWaves {
    static constexpr int value = PV < 0 ? ((SegCfg<L>::W32 || L == 4096) ? 1 : 2) : SegCfg<L>::WAVES;
};
#define SEG_KERNEL_ATTRS(L_, PV_) \
    __launch_bounds__(SegCfg<L_>::BLOCK) __attribute__((amdgpu_waves_per_eu(SegWaves<L_, PV_>::value, SegWaves<L_, PV_>::value)))

// Variant builds only (-DMFB_SEG_TRACE, tools/xcd_trace.py): start / end time and XCC of every workgroup of the
// branch-free search kernel, to see how evenly the grid drains over the XCDs.
#ifndef MFB_SEG_PRIO
#define MFB_SEG_PRIO 0
#endif
#ifdef MFB_SEG_TRACE
__device__ unsigned long long g_seg_trace[3 * 65536];
#endif

template <int L, int MODE, int PV>
__global__ void SEG_KERNEL_ATTRS(L, PV) k_seg(SegArgs a) {
#ifdef MFB_SEG_TRACE
    unsigned long long t0 = 0;
    if (PV >= 0 && MODE == SEG_REDUCE && threadIdx.x == 0) t0 = wall_clock64();
#endif
#if MFB_SEG_PRIO
    // experiment: the two barrier teams that share a CU get different issue priorities by their wave slot
    if (L >= 2048) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));
#if MFB_SEG_PRIO == 2
        if (blockIdx.x & 1)
#else
        if (hw & 1)
#endif
            __builtin_amdgcn_s_setprio(3);
        else
            __builtin_amdgcn_s_setprio(0);
    }
#endif
    seg_body<L, MODE, PV>(a, (int)blockIdx.x);
#ifdef MFB_SEG_TRACE
    if (PV >= 0 && MODE == SEG_REDUCE && threadIdx.x == 0 && blockIdx.x < 65536) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_seg_trace[3 * blockIdx.x] = t0;
        g_seg_trace[3 * blockIdx.x + 1] = wall_clock64();
        g_seg_trace[3 * blockIdx.x + 2] = xcc;
    }
#endif
}
