// mfbank.hip -- hand-written gfx950 kernels + C ABI for the Doppler matched-filter bank.
//
// Data layout in HBM (N = N1*N2 samples per block, D Doppler bins, M matched filters):
//   d_x     complex64 [N]        time-domain block (or caller's device pointer)
//   d_X     complex64 [N]        spectrum, natural order
//   d_masks complex64 [M][N]     filter bank as protocol.get_filter returns it (row-major)
//   d_Z     complex64 [Dc*MU][N1][N2] intermediate of the two-pass inverse FFT for ONE chunk of
//                                Dc Doppler bins x MU unique filters: Z[row][n1][k2], already multiplied
//                                by the inter-pass twiddle W_N^(k2*n1).  Never holds the D*M*N cube.
//   d_part  float32 [D][MU][PARTS] per-workgroup partial |.|^2 sums (fixed-order second reduction,
//                                no float atomics -> bit-reproducible)
//   d_sum   float32 [D][M]       doppSum, same meaning/layout as the reference's GPU_bufDoppSum
//   d_xc    complex64 [M][N]     matched-filter outputs at the chosen shift (natural order)
//
// Two-pass inverse FFT of length N = N1*N2 with input index k = N2*k1 + k2 and output index
// n = n1 + N1*n2:
//   pass 1 (k_pass1): for 16 adjacent k2 ("tile"), all k1: load X[(k+shift) mod N]*mask[m][k] on the
//           fly (128-byte coalesced segments), N1-point FFT over k1 in registers+LDS, multiply
//           by W_N^(k2*n1), store Z[n1][k2] (128-byte segments).
//   pass 2 (k_pass2): for every n1: N2-point FFT over k2 of the contiguous row Z[n1][:], then either
//           reduce |y|^2 in registers -> wave -> workgroup (Doppler search; the time-domain rows
//           are never written), or write the row back in place and let k_transpose produce
//           y[n1 + N1*n2] (demodulation / forward FFT).
//
// Short filters (every shipped protocol) take the single-pass overlap-save path instead (seg_kernels.hpp):
// no intermediate at all; the two-pass path below stays as the fallback for filters without a short
// impulse response and for blocks too small to segment, and serves the plain forward transforms.
//
// Reference semantics reproduced (file:line in the reference tree, pyCuSDR/):
//   shift-multiply  demodulator/cuda_kernels.cu:339-373, 174-185
//   |.|^2 / 2^18 row sums  cuda_kernels.cu:421-480      pick  cuda_kernels.cu:502-597
//   envelope        cuda_kernels.cu:191-205             rate/phase  cuda_kernels.cu:236-320
//   centres         cuda_kernels.cu:78-146
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mfbank.h"
#include "fft_core.hpp"
#include "filter_taps.hpp"
#include "seg_kernels.hpp"
#include "twopass_kernels.hpp"
#include "small_kernels.hpp"
#include "sync_kernels.hpp"
#include "energy_kernels.hpp"
#include "stream_kernels.hpp"
#include "hostcopy.hpp"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct BlockFlight {      // one block -- or one batch of nb blocks -- between mfb_receive_block(s)_begin and _end
    bool active;
    int mode, nthreads, bcap, shift, op;
    int nb;               // 0: a single block (mfb_receive_block_begin); > 0: a batch (mfb_receive_blocks_begin)
    size_t rec;           // bytes per block record in the staging buffer (batches)
    size_t ext;           // offset of the stream-stage outputs inside a record (0: none)
    size_t off[5];
    unsigned long long seq;
};

constexpr int WG_NB = 32;
struct BlockGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    mfb_block_params params;
    int nb;               // blocks of the batch the graph was recorded for (0: single block)
    unsigned long long epoch;
    bool seen, failed;
};

struct mfb_ctx {
    int device;
    int log2N, N, N1, N2, l1, l2, lo;
    int D, Doff, Dtot, M, W, sum_all, cs_off;
    int chunk, mpb, jsplit;
    int MU;               // unique filter rows (up to sign)
    bool chunk_auto;      // chunk still at its default (recomputed when the bank turns out to have duplicates)
    int *d_uniq, *d_rep;  // [MU] bank row of each unique filter; [M] unique index of each filter
    int parts, srb;
    int num_cus;          // compute units of the handle's device
    hipStream_t own_stream, stream;
    cf *h_in;  // pinned
    // page-locked landing zone of the small read-backs (pick, rate/phase, spectrum windows, symbol decisions): a
    // copy into pageable memory is staged by the runtime with a round trip of its own per copy
    uint8_t *h_back;
    size_t back_cap;
    // Host mirror of the spectrum for small blocks.  The reference keeps the spectrum in host-mapped memory and reads
    // its SNR windows straight from there (DB:456-457, 651-661); here the first mfb_get_spectrum on a handle turns on
    // an asynchronous copy of every new spectrum (own stream, beside the search), so that later window reads cost no
    // host-device round trip.  Blocks above MIRROR_MAX_N keep fetching windows on demand.
    cf *h_X;
    bool mirror, mirror_valid;
    hipStream_t copy_stream;
    hipEvent_t ev_fft, ev_X;
    cf *d_x, *d_X, *d_masks, *d_Z, *d_xc, *d_P;
    const cf *d_in;  // current time-domain input (d_x or caller's device pointer)
    float *d_env;
    int *d_shifts;
    cf *d_tw1, *d_tw2, *d_twLo, *d_twHi;
    float *d_part, *d_sum, *d_res, *d_cr;
    int *d_sym, *d_cen;
    float *d_mag;
    int cap;  // capacity of the centres buffers
    size_t z_rows;  // rows allocated in d_Z
    bool have_filters, have_shifts, have_input, have_xc;
    hipEvent_t t0, t1;
    bool prof;
    // single-pass overlap-save path (seg_kernels.hpp)
    int path_req, segl_req;   // requested: MFB_PATH_*, log2 L (0 = automatic)
    int path, segl;           // resolved
    int T, win_start, V, Q;   // taps, window start, valid outputs per segment, segments
    int seg_wpc, seg_mpb;     // workgroups per CU, filters per team pass (0 = automatic)
    cf *d_G, *d_twL;
    int twL_len;
    size_t part_cap;          // floats allocated in d_part
    taps::Bank *bank;         // host copy of the taps (kept so that L can be changed)
    int basis_req, basis;     // MFB_BASIS_*: requested / in force
    int MB;                   // filters of the span basis
    cf *d_Gb;                 // their segment spectra [MB][L] (slot-pair layout)
    int gb_l;                 // segment length d_Gb was built for
    int search_mode;          // MFB_SEARCH_*: transforms (default) or the opt-in spectral-energy shortcut
    float *d_pow, *d_W;       // |X|^2 [N]; filter energy [R][N] (R = 1 under SUM_ALL_MASKS, else M)
    cf *h_in2;                // second page-locked input buffer (blocks in flight alternate between the two)
    cf *d_x2;                 // its device copy: the next block's samples arrive (own stream) while the current block is searched
    hipStream_t in_stream;    // host-to-device copies of the block path
    hipEvent_t ev_h2d[2];     // [input buffer]: the copy has landed
    hipEvent_t ev_xfree[2];   // [input buffer]: the last block that read this device copy has been enqueued and finished
    uint8_t *h_blk[2];        // page-locked staging of the two block flights
    size_t blk_cap[2];
    hipEvent_t ev_blk[2];
    BlockFlight flight[2];
    unsigned long long blk_seq;
    BlockGraph bgraph[2][2];  // [input buffer][slot]: the block's launches as a HIP graph
    unsigned long long epoch; // bumped by every change that invalidates them
    uint8_t *d_blkout;        // mfb_receive_block: the block's result record (scalars | SNR windows | symbols), one copy to the host
    uint8_t *d_blkout2;       // ... of flight slot 1 when a block runs as two parts on two streams (mfb_set_batch_overlap)
    BlockGraph bgraph2[2][2]; // part 2's recorded graphs ([input buffer][slot]; bgraph holds part 1's, or the whole block)
    BlockScalars *d_scal;     // = d_blkout
    int band_cap;
    bool W_valid;
    // batches of blocks (mfb_receive_blocks_*): B consecutive blocks of ONE contiguous window of samples per launch set
    int win_blocks, win_stride;   // capacity in blocks and block advance (N - overlap) the windows were made for
    cf *h_win[2], *d_win[2];      // page-locked windows of win_blocks * win_stride + (N - win_stride) samples, and their device copies
    hipEvent_t ev_wh2d[2], ev_wfree[2];
    int bat_cap;                  // blocks the batch work buffers below hold
    cf *d_Xb, *d_xcb, *d_Pb;      // spectra [B][N], matched-filter outputs [B][M][N], envelope spectra [B][N]
    float *d_envb, *d_sumb, *d_resb, *d_crb;   // envelopes [B][N], doppSum [B][Dtot][M], picks [B][2], rate triples [B][3]
    uint8_t *d_batout;            // result records [B][rec] of the batch in flight slot 0 ...
    uint8_t *d_batout2;           // ... and slot 1 (the next batch's search writes its picks while this one's records are still read)
    size_t batout_cap;
    // A batch in two parts on two streams (round 6).  Part 1 -- forward transforms, search, pick: the kernels that fill the chip --
    // stays on `stream`; part 2 -- matched filters at the picked shift, envelope, its spectrum, rate, centres, the integer stages,
    // the read-back: a chain of small launches that leaves most of the chip idle -- goes to `s2`, behind an event, with an
    // intermediate of its own for its transforms (d_Z2).  The next batch's part 1 then runs beside this batch's part 2.
    hipStream_t s2;
    hipEvent_t ev_p1[2];          // [slot]: part 1 of the batch in that flight has been enqueued and finished
    cf *d_Z2;
    size_t z2_rows;
    bool use_z2;                  // the transforms being enqueued use d_Z2 (part 2)
    bool s2_busy;                 // something was enqueued on s2 since the last wait for it
    bool batch_overlap;           // mfb_set_batch_overlap
    int last_batch_blocks;        // blocks of the batch begun last (mfb_get_batch_scores)
    int cu_part, cu_parts;        // mfb_set_cu_share (0, 0: the whole device)
    BlockGraph wgraph2[2][2][2][WG_NB + 1];     // part 2's recorded graphs (wgraph holds part 1's)
    // [window][slot][carry parity][blocks of the batch]: a receive loop that takes whatever is complete (1 ... B blocks per call)
    // keeps one recorded graph per batch size it has met twice; sizes above WG_NB share entry 0 (recorded again when the size changes)
    BlockGraph wgraph[2][2][2][WG_NB + 1];
    // the integer stages behind the symbol decisions on the device (stream_kernels.hpp; mfb_set_stream_stages): A12 bit lookup,
    // A13 block-overlap alignment, A14 sync search on the stream without a stash -- batches only
    bool st_on;
    StreamArgs st;                // constant part (LUTs, thresholds, templates); records, carries filled per batch
    uint8_t *d_lut8;
    int *d_lut3;
    int8_t *d_sttmpl;
    StreamCarry *d_carry[2];      // [parity]: the previous batch's tail and bit ring / this batch's
    int carry_cur;                // buffer that holds the state the next batch starts from
    StreamCarry *h_seed;          // page-locked staging of mfb_stream_seed
    std::vector<hipEvent_t> ev[2];
    std::vector<hipEvent_t> ev_pool;
    // the search with the Doppler shift on the filter side (seg_kernels.hpp, segf_body; 256-point segments): per-bin spectra
    std::vector<int> h_shifts, h_uniq;   // host copies of the shift table and of the unique-filter map
    std::vector<int> h_mult;             // how many of the M filters each unique one stands for (k_finalize counts its row that often)
    float *d_Qs = nullptr;               // [Dtot][L]: per-bin sum over the rows of |d_Gs|^2 (SUM_ALL searches; segf_body SUMQ), built with d_Gs
    cf *d_Gs = nullptr;                  // [Dtot][rows][L]: rows = the unique filters, or the span basis
    int gs_rows = 0;                     // rows per bin d_Gs was built for (0: not built)
    int gs_span = -1;                    // ... and for which basis
    int gs_l = 0;                        // ... and segment length (log2)
    int fsm_fb = 0, fsm_fs = 0;          // rectangle of a wave: bins x slots (0: default)
};

#define HIPCHK(x)                                                                            \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "mfbank: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? MFB_ERR_ALLOC : MFB_ERR_HIP;               \
        }                                                                                    \
    } while (0)

static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
// the handle's stream and -- when a batch has put work there since the last wait -- the second stream of the batch path
static hipError_t sync_streams(mfb_ctx *c) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && c->s2 && c->s2_busy) {
        e = hipStreamSynchronize(c->s2);
        if (e == hipSuccess) c->s2_busy = false;
    }
    return e;
}

// Device allocations of a handle go through dev_alloc so that a test can make the n-th one fail
// (mfb_debug_fail_alloc): the only way to exercise the free-on-error path of mfb_create without driving
// a 288 GB device out of memory.
static thread_local int g_fail_alloc_countdown = 0;
static thread_local int g_throw_alloc_countdown = 0;      // nth < 0: the |nth|-th allocation throws std::bad_alloc instead (see guarded)
static hipError_t dev_alloc(void **p, size_t bytes) {
    if (g_fail_alloc_countdown > 0 && --g_fail_alloc_countdown == 0) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    if (g_throw_alloc_countdown > 0 && --g_throw_alloc_countdown == 0) {
        *p = nullptr;
        throw std::bad_alloc();
    }
    return hipMalloc(p, bytes);
}
// No C++ exception leaves the library: the entry points that do host-side preparation (filter analysis, segment spectra, tables:
// std::vector, std::thread) run inside this; a failed host allocation is MFB_ERR_ALLOC like a failed device allocation.
template <class F>
static int guarded(F &&f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return MFB_ERR_ALLOC;
    } catch (...) {
        return MFB_ERR_STATE;
    }
}
extern "C" int mfb_debug_fail_alloc(int nth) {
    g_fail_alloc_countdown = nth > 0 ? nth : 0;
    g_throw_alloc_countdown = nth < 0 ? -nth : 0;
    return MFB_OK;
}

extern "C" const char *mfb_strerror(int s) {
    switch (s) {
        case MFB_OK: return "ok";
        case MFB_ERR_ARG: return "invalid argument or shape";
        case MFB_ERR_DTYPE: return "invalid element type";
        case MFB_ERR_ALLOC: return "allocation failed";
        case MFB_ERR_HIP: return "HIP runtime error";
        case MFB_ERR_STATE: return "filters, shifts or input not set";
        case MFB_ERR_UNSUPPORTED: return "transform size not supported";
        default: return "unknown status";
    }
}
extern "C" int mfb_abi_version(void) { return 9; }

static void make_twiddles(std::vector<cf> &v, int count, double denom, double stepmul) {
    v.resize(count);
    for (int i = 0; i < count; ++i) {
        const double ang = 2.0 * M_PI * (double)i * stepmul / denom;
        v[i] = (cf){(float)cos(ang), (float)sin(ang)};
    }
}

// (cos, tan) pairs of the fused-twiddle transforms (fft_core.hpp, MFB_FFT_FUSED), appended to the W_L table: angle 2 pi u / L.
// An angle of exactly pi/2 (one lane of stage 1) becomes (2^-40, 2^40): c t = 1 exactly, and c b vanishes in the rounding of the sum.
static cf ct_pair(long u, int L) {
    const double ang = 2.0 * M_PI * (double)u / (double)L;
    double c = cos(ang);
    const double s = sin(ang);
    if (fabs(c) < 1e-9) c = 0x1p-40;
    return (cf){(float)c, (float)(s / c)};
}
static void append_fused_tables(std::vector<cf> &v, int L) {
    if (L == 256) {            // F256Regs: [16 lanes][8]
        for (int g = 0; g < 16; ++g) {
            const long u[8] = {8 * g, 4 * g, 2 * g, 2 * g + 32, g, g + 16, g + 32, g + 48};
            for (int k = 0; k < 8; ++k) v.push_back(ct_pair(u[k], L));
        }
    } else if (L == 2048) {    // F2048Regs: [64 positions][16], then W_64^p, p < 32
        for (int g = 0; g < 64; ++g) {
            v.push_back(ct_pair(16 * g, L));
            v.push_back(ct_pair(8 * g, L));
            for (int q = 0; q < 2; ++q) v.push_back(ct_pair(4 * g + 256 * q, L));
            for (int q = 0; q < 4; ++q) v.push_back(ct_pair(2 * g + 128 * q, L));
            for (int q = 0; q < 8; ++q) v.push_back(ct_pair(g + 64 * q, L));
        }
        for (int p = 0; p < 32; ++p) v.push_back(ct_pair(32 * p, L));
    }
}

static int upload_tw(cf **dst, const std::vector<cf> &v) {
    HIPCHK(dev_alloc((void **)dst, v.size() * sizeof(cf)));
    HIPCHK(hipMemcpy(*dst, v.data(), v.size() * sizeof(cf), hipMemcpyHostToDevice));
    return MFB_OK;
}

// pass-2 work split: sub-rows (n1 values) per workgroup; a power of two in [RB, N1]
static void set_rows_per_block(mfb_ctx *c, int srb) {
    const int NT = c->N2 / 16;
    const int RB = NT >= 256 ? 1 : 256 / NT;
    int p = 1;
    while (p * 2 <= srb) p *= 2;
    srb = p;
    if (srb < RB) srb = RB;
    if (srb > c->N1) srb = c->N1;
    c->srb = srb;
    c->parts = c->N1 / srb;
}

static int default_chunk(const mfb_ctx *c) {
    const size_t row_bytes = (size_t)c->N * sizeof(cf) * c->MU;
    size_t ch = ((size_t)8 << 30) / row_bytes;
    if (ch < 1) ch = 1;
    if (ch > (size_t)c->Dtot) ch = c->Dtot;
    if (ch * c->MU > 65535) ch = 65535 / c->MU;
    return (int)ch;
}

// Intermediate of the two-pass transforms.  The segment path needs one row only (the plain forward
// transforms of A3 / A10 still run two-pass); the two-pass search needs chunk * MU rows.
static size_t z_rows_needed(const mfb_ctx *c) {
    if (!c->have_filters || c->path == MFB_PATH_SEGMENT) return 1;
    if (c->search_mode == MFB_SEARCH_ENERGY) return (size_t)c->M;   // the demodulation stage only
    const size_t rows = (size_t)c->chunk * c->MU;   // the search stores only the unique filter rows
    return rows > (size_t)c->M ? rows : (size_t)c->M;
}
static int alloc_Z(mfb_ctx *c) {
    const size_t need = z_rows_needed(c);
    if (c->d_Z && c->z_rows >= need) return MFB_OK;
    if (c->d_Z) {
        HIPCHK(sync_streams(c));
        HIPCHK(hipFree(c->d_Z));
        c->d_Z = nullptr;
        c->z_rows = 0;
    }
    HIPCHK(dev_alloc((void **)&c->d_Z, need * c->N * sizeof(cf)));
    c->z_rows = need;
    return MFB_OK;
}

static int reserve_partials(mfb_ctx *c, size_t floats) {
    if (c->part_cap >= floats) return MFB_OK;
    if (c->d_part) {
        HIPCHK(sync_streams(c));
        ++c->epoch;          // recorded block graphs hold the old address
        HIPCHK(hipFree(c->d_part));
        c->d_part = nullptr;
        c->part_cap = 0;
    }
    HIPCHK(dev_alloc((void **)&c->d_part, floats * sizeof(float)));
    c->part_cap = floats;
    return MFB_OK;
}

// Device-side result record of a block: [BlockScalars, 256 B][bands 2 x bcap complex64][sym][cen][mag] (nthreads each),
// contiguous so that ONE copy brings it to the host.
#define BLK_HEAD 256
static size_t blkout_bytes(int bcap, int nthreads) {
    return BLK_HEAD + align16((size_t)2 * bcap * sizeof(cf)) + 3 * align16((size_t)nthreads * sizeof(int));
}
static int blkout_reserve(mfb_ctx *c, int bcap) {
    if (c->d_blkout && bcap <= c->band_cap) return MFB_OK;
    HIPCHK(sync_streams(c));
    if (c->d_blkout) {
        ++c->epoch;          // captured block graphs hold the old record's address (and d_scal): none of them may replay
        HIPCHK(hipFree(c->d_blkout));
    }
    c->d_blkout = nullptr;
    c->band_cap = 0;
    HIPCHK(dev_alloc((void **)&c->d_blkout, blkout_bytes(bcap, c->cap)));
    if (c->d_blkout2) HIPCHK(hipFree(c->d_blkout2));
    c->d_blkout2 = nullptr;
    HIPCHK(dev_alloc((void **)&c->d_blkout2, blkout_bytes(bcap, c->cap)));        // flight slot 1 (two streams: mfb_set_batch_overlap)
    c->band_cap = bcap;
    c->d_scal = (BlockScalars *)c->d_blkout;
    return MFB_OK;
}

// ---- per-device kernel attributes ------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): it is set for every
// instantiation that can need more than 48 KiB whenever a handle is created, on that handle's device.
template <int L1>
static int attr_p1() {
    if (P1Cfg<L1>::lds_bytes > 48 * 1024) {
        const int b = (int)P1Cfg<L1>::lds_bytes;
        HIPCHK(hipFuncSetAttribute((const void *)k_pass1<L1, KIND_BANK>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
        HIPCHK(hipFuncSetAttribute((const void *)k_pass1<L1, KIND_FWDC>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
        HIPCHK(hipFuncSetAttribute((const void *)k_pass1<L1, KIND_FWDR>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
    }
    return MFB_OK;
}
template <int L2>
static int attr_p2() {
    if (P2Cfg<L2>::lds_bytes > 48 * 1024) {
        const int b = (int)P2Cfg<L2>::lds_bytes;
        HIPCHK(hipFuncSetAttribute((const void *)k_pass2<L2, MODE_REDUCE>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
        HIPCHK(hipFuncSetAttribute((const void *)k_pass2<L2, MODE_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
    }
    return MFB_OK;
}
template <int L>
static int attr_seg() {
    const int b = (int)SegCfg<L>::lds_bytes(L == 8192 ? 8 : SEG_MPB_MAX);     // (8192 points: 157 KiB with 8 filters per pass is all a CU has)
    HIPCHK(hipFuncSetAttribute((const void *)k_seg<L, SEG_STORE, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
#define MFB_ATTR_PV(PV_) HIPCHK(hipFuncSetAttribute((const void *)k_seg<L, SEG_REDUCE, PV_>, hipFuncAttributeMaxDynamicSharedMemorySize, b))
    MFB_ATTR_PV(-1);
    if constexpr (seg_ppl(L) == 32) {
        MFB_ATTR_PV(16); MFB_ATTR_PV(17); MFB_ATTR_PV(18); MFB_ATTR_PV(19); MFB_ATTR_PV(20); MFB_ATTR_PV(21); MFB_ATTR_PV(22);
        MFB_ATTR_PV(23); MFB_ATTR_PV(24); MFB_ATTR_PV(25); MFB_ATTR_PV(26); MFB_ATTR_PV(27); MFB_ATTR_PV(28); MFB_ATTR_PV(29);
        MFB_ATTR_PV(30); MFB_ATTR_PV(31); MFB_ATTR_PV(32);
    } else {
        MFB_ATTR_PV(8);
        MFB_ATTR_PV(9);
        MFB_ATTR_PV(10);
        MFB_ATTR_PV(11);
        MFB_ATTR_PV(12);
        MFB_ATTR_PV(13);
        MFB_ATTR_PV(14);
        MFB_ATTR_PV(15);
        MFB_ATTR_PV(16);
    }
#undef MFB_ATTR_PV
    return MFB_OK;
}
static int set_kernel_attributes(const mfb_ctx *c) {
    int rc = MFB_OK;
    switch (c->l1) {
        case 5: rc = attr_p1<32>(); break;
        case 6: rc = attr_p1<64>(); break;
        case 7: rc = attr_p1<128>(); break;
        case 8: rc = attr_p1<256>(); break;
        case 9: rc = attr_p1<512>(); break;
    }
    if (rc) return rc;
    switch (c->l2) {
        case 13: rc = attr_p2<8192>(); break;
        default: break;   // <= 4096 points: 35 KiB
    }
    if (rc) return rc;
    if ((rc = attr_seg<256>())) return rc;
    {
        const int b = (int)(SegCfg<256>::LDS_ELEMS * sizeof(cf) + (size_t)(SegCfg<256>::BLOCK / 64) * SEG_MPB_MAX * SEG_ACC_STRIDE * sizeof(float));
#define MFB_ATTR_F(PV_) HIPCHK(hipFuncSetAttribute((const void *)k_segf<256, PV_>, hipFuncAttributeMaxDynamicSharedMemorySize, b))
        MFB_ATTR_F(8); MFB_ATTR_F(9); MFB_ATTR_F(10); MFB_ATTR_F(11); MFB_ATTR_F(12); MFB_ATTR_F(13); MFB_ATTR_F(14); MFB_ATTR_F(15); MFB_ATTR_F(16);
#undef MFB_ATTR_F
#define MFB_ATTR_Q(L_, PV_, B_)                                                                                                       \
    if constexpr (segf_has_sumq<L_, PV_>())                                                                                           \
        HIPCHK(hipFuncSetAttribute((const void *)k_segf<L_, PV_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, B_));                    \
    if constexpr (segf_has_presum<L_, PV_>())                                                                                         \
        HIPCHK(hipFuncSetAttribute((const void *)k_segf<L_, PV_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, B_))
        MFB_ATTR_Q(256, 11, b); MFB_ATTR_Q(256, 12, b); MFB_ATTR_Q(256, 13, b); MFB_ATTR_Q(256, 14, b); MFB_ATTR_Q(256, 15, b);
        const int b2 = (int)(SegCfg<2048>::LDS_ELEMS * sizeof(cf) + (size_t)(SegCfg<2048>::BLOCK / 64) * 8 * SEG_ACC_STRIDE * sizeof(float));
#define MFB_ATTR_F(PV_) HIPCHK(hipFuncSetAttribute((const void *)k_segf<2048, PV_>, hipFuncAttributeMaxDynamicSharedMemorySize, b2))
        MFB_ATTR_F(16); MFB_ATTR_F(17); MFB_ATTR_F(18); MFB_ATTR_F(19); MFB_ATTR_F(20); MFB_ATTR_F(21); MFB_ATTR_F(22); MFB_ATTR_F(23); MFB_ATTR_F(24);
        MFB_ATTR_F(25); MFB_ATTR_F(26); MFB_ATTR_F(27); MFB_ATTR_F(28); MFB_ATTR_F(29); MFB_ATTR_F(30); MFB_ATTR_F(31); MFB_ATTR_F(32);
#undef MFB_ATTR_F
        MFB_ATTR_Q(2048, 22, b2); MFB_ATTR_Q(2048, 23, b2); MFB_ATTR_Q(2048, 24, b2); MFB_ATTR_Q(2048, 25, b2); MFB_ATTR_Q(2048, 26, b2);
        MFB_ATTR_Q(2048, 27, b2); MFB_ATTR_Q(2048, 28, b2); MFB_ATTR_Q(2048, 29, b2); MFB_ATTR_Q(2048, 30, b2); MFB_ATTR_Q(2048, 31, b2);
#undef MFB_ATTR_Q
    }
    if ((rc = attr_seg<512>())) return rc;
    if ((rc = attr_seg<1024>())) return rc;
    if ((rc = attr_seg<2048>())) return rc;
    if ((rc = attr_seg<4096>())) return rc;
    return attr_seg<8192>();
}

static int create_impl(mfb_ctx *c) {
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(hipEventCreate(&c->t0));
    HIPCHK(hipEventCreate(&c->t1));
    int rc = set_kernel_attributes(c);
    if (rc) return rc;
    const int M = c->M;
    const size_t nb = (size_t)c->N * sizeof(cf);
    HIPCHK(hipHostMalloc((void **)&c->h_in, nb, hipHostMallocDefault));
    memset(c->h_in, 0, nb);
    // three arrays of at most N/2 symbol decisions, the block scalars and the two SNR windows of mfb_receive_block
    c->back_cap = (size_t)3 * (c->N / 2) * sizeof(int) + ((size_t)256 << 10);
    if (c->back_cap < ((size_t)64 << 10)) c->back_cap = (size_t)64 << 10;
    HIPCHK(hipHostMalloc((void **)&c->h_back, c->back_cap, hipHostMallocDefault));
    HIPCHK(dev_alloc((void **)&c->d_x, nb));
    HIPCHK(dev_alloc((void **)&c->d_X, nb));
    HIPCHK(dev_alloc((void **)&c->d_masks, nb * M));
    HIPCHK(dev_alloc((void **)&c->d_xc, nb * M));
    HIPCHK(dev_alloc((void **)&c->d_P, nb));
    HIPCHK(dev_alloc((void **)&c->d_env, (size_t)c->N * sizeof(float)));
    HIPCHK(dev_alloc((void **)&c->d_shifts, (size_t)c->Dtot * sizeof(int)));
    if ((rc = reserve_partials(c, (size_t)c->Dtot * M * c->N1))) return rc;
    HIPCHK(dev_alloc((void **)&c->d_sum, (size_t)c->Dtot * M * sizeof(float)));
    HIPCHK(hipMemset(c->d_sum, 0, (size_t)c->Dtot * M * sizeof(float)));
    HIPCHK(dev_alloc((void **)&c->d_res, 2 * sizeof(float)));
    HIPCHK(dev_alloc((void **)&c->d_uniq, (size_t)M * sizeof(int)));
    HIPCHK(dev_alloc((void **)&c->d_rep, (size_t)M * sizeof(int)));
    HIPCHK(dev_alloc((void **)&c->d_cr, 3 * sizeof(float)));
    c->cap = c->N / 2;  // symbols never shorter than 2 samples
    HIPCHK(dev_alloc((void **)&c->d_sym, (size_t)c->cap * sizeof(int)));
    HIPCHK(dev_alloc((void **)&c->d_cen, (size_t)c->cap * sizeof(int)));
    HIPCHK(dev_alloc((void **)&c->d_mag, (size_t)c->cap * sizeof(float)));
    // one row for the plain forward transforms; the search's share is sized when the filters arrive
    if ((rc = alloc_Z(c))) return rc;
    // block path (mfb_receive_block*): everything it needs exists before the first block -- page-locking memory and creating
    // events inside a stream of blocks costs milliseconds
    if ((rc = blkout_reserve(c, 1024))) return rc;
    for (int i = 0; i < 2; ++i) {
        // scalars + three arrays of the symbols a rate window of +-10 % around 4 samples per symbol admits + the two windows
        c->blk_cap[i] = blkout_bytes(c->band_cap, c->N / 3);
        HIPCHK(hipHostMalloc((void **)&c->h_blk[i], c->blk_cap[i], hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&c->ev_blk[i], hipEventDisableTiming));
    }

    std::vector<cf> t;
    make_twiddles(t, c->N1, (double)c->N1, 1.0);
    if ((rc = upload_tw(&c->d_tw1, t))) return rc;
    make_twiddles(t, c->N2, (double)c->N2, 1.0);
    if ((rc = upload_tw(&c->d_tw2, t))) return rc;
    make_twiddles(t, 1 << c->lo, (double)c->N, 1.0);
    if ((rc = upload_tw(&c->d_twLo, t))) return rc;
    make_twiddles(t, c->N >> c->lo, (double)c->N, (double)(1 << c->lo));
    if ((rc = upload_tw(&c->d_twHi, t))) return rc;
    c->d_in = c->d_x;
    return MFB_OK;
}

static int create_body(mfb_ctx **out, int device, int log2N, int num_dopplers, int doppler_offset, int M,
                       int window_width, int sum_all_masks, int code_search_mask_offset) {
    if (!out) return MFB_ERR_ARG;
    *out = nullptr;
    if (num_dopplers < 1 || doppler_offset < 0 || M < 1 || M > 64 || window_width < 1 || (window_width & 1) == 0 ||
        code_search_mask_offset < 0 || 2 * code_search_mask_offset >= M)
        return MFB_ERR_ARG;
    if (log2N < 10 || log2N > 22) return MFB_ERR_UNSUPPORTED;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(device));
    int ncu = 0;
    HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));

    mfb_ctx *c = new mfb_ctx();
    memset((void *)c, 0, offsetof(mfb_ctx, ev));  // everything before the vectors
    c->device = device;
    c->num_cus = ncu > 0 ? ncu : 256;
    c->log2N = log2N;
    c->N = 1 << log2N;
    // N = N1 * N2: columns of at most 256 points, rows of at most 8192 (2^22 = 512 x 8192: a 16384-point row would
    // need 1024 threads at 128 VGPRs and spilled > 100 registers)
    c->l1 = log2N / 2 < 8 ? log2N / 2 : 8;
    if (log2N - c->l1 > 13) c->l1 = log2N - 13;
    c->l2 = log2N - c->l1;
    c->N1 = 1 << c->l1;
    c->N2 = 1 << c->l2;
    c->lo = (log2N + 1) / 2;
    c->D = num_dopplers;
    c->Doff = doppler_offset;
    c->Dtot = num_dopplers + doppler_offset;
    c->M = M;
    c->W = window_width;
    c->sum_all = sum_all_masks ? 1 : 0;
    c->cs_off = code_search_mask_offset;
    // Two-pass search, Doppler bins per launch: as many as an 8 GiB intermediate holds.  Measured on
    // MI355X (C2: D=256, M=8, N=2^20): long persistent loops amortise the per-workgroup twiddle setup
    // and keep each XCD's L2 serving the filter tile to many Doppler streams, so big chunks win (chunk
    // 16: 9.8 ms, 64: 6.65 ms, 128: 6.55 ms per block) -- but a 16 GiB intermediate (chunk 256) costs
    // 0.4 ms more than two 8 GiB launches.  Keeping the intermediate inside the 256 MiB Infinity
    // Cache (chunk 1-2) does not pay: the cache streams no faster than HBM (tools/ubench/mall.hip).
    c->MU = M;
    c->chunk_auto = true;
    c->chunk = default_chunk(c);
    c->mpb = M < 8 ? M : 8;
    c->jsplit = 32;
    set_rows_per_block(c, 64);
    c->path_req = MFB_PATH_AUTO;
    c->path = MFB_PATH_TWOPASS;

    // any failure below frees what was allocated so far (mfb_destroy tolerates partial handles):
    // the counterpart of the reference's context teardown, demodulator_base.py:517-530
    int rc;
    try {
        rc = create_impl(c);
    } catch (...) {
        (void)mfb_destroy(c);
        throw;
    }
    if (rc) {
        (void)mfb_destroy(c);
        return rc;
    }
    *out = c;
    return MFB_OK;
}
extern "C" int mfb_create(mfb_ctx **out, int device, int log2N, int num_dopplers, int doppler_offset, int M,
                          int window_width, int sum_all_masks, int code_search_mask_offset) {
    return guarded([&] { return create_body(out, device, log2N, num_dopplers, doppler_offset, M, window_width, sum_all_masks, code_search_mask_offset); });
}

static void graph_drop(BlockGraph &g);
extern "C" int mfb_destroy(mfb_ctx *c) {
    if (!c) return MFB_ERR_ARG;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->s2) (void)hipStreamSynchronize(c->s2);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    void *bufs[] = {c->d_uniq, c->d_rep, c->d_x,  c->d_X,    c->d_masks, c->d_Z,   c->d_xc,  c->d_P,   c->d_env, c->d_shifts, c->d_tw1,
                    c->d_tw2, c->d_twLo, c->d_twHi,  c->d_part, c->d_sum, c->d_res, c->d_cr,  c->d_sym,    c->d_cen, c->d_mag,
                    c->d_G, c->d_twL, c->d_Gb, c->d_pow, c->d_W, c->d_blkout, c->d_blkout2, c->d_x2,
                    c->d_win[0], c->d_win[1], c->d_Xb, c->d_xcb, c->d_Pb, c->d_envb, c->d_sumb, c->d_resb, c->d_crb, c->d_batout, c->d_batout2, c->d_Z2,
                    c->d_lut8, c->d_lut3, c->d_sttmpl, c->d_carry[0], c->d_carry[1]};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_in2) (void)hipHostFree(c->h_in2);
    for (int i = 0; i < 2; ++i) {
        if (c->h_win[i]) (void)hipHostFree(c->h_win[i]);
        if (c->ev_wh2d[i]) (void)hipEventDestroy(c->ev_wh2d[i]);
        if (c->ev_wfree[i]) (void)hipEventDestroy(c->ev_wfree[i]);
    }
    for (auto &row : c->bgraph)
        for (auto &g : row) graph_drop(g);
    for (auto &row : c->bgraph2)
        for (auto &g : row) graph_drop(g);
    for (auto &row : c->wgraph)
        for (auto &col : row)
            for (auto &par : col)
                for (auto &g : par) graph_drop(g);
    for (auto &row : c->wgraph2)
        for (auto &col : row)
            for (auto &par : col)
                for (auto &g : par) graph_drop(g);
    for (int i = 0; i < 2; ++i) {
        if (c->h_blk[i]) (void)hipHostFree(c->h_blk[i]);
        if (c->ev_blk[i]) (void)hipEventDestroy(c->ev_blk[i]);
        if (c->ev_p1[i]) (void)hipEventDestroy(c->ev_p1[i]);
    }
    if (c->s2) (void)hipStreamDestroy(c->s2);
    if (c->h_back) (void)hipHostFree(c->h_back);
    if (c->h_seed) (void)hipHostFree(c->h_seed);
    if (c->h_X) (void)hipHostFree(c->h_X);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->in_stream) {
        (void)hipStreamSynchronize(c->in_stream);
        (void)hipStreamDestroy(c->in_stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->ev_h2d[i]) (void)hipEventDestroy(c->ev_h2d[i]);
        if (c->ev_xfree[i]) (void)hipEventDestroy(c->ev_xfree[i]);
    }
    if (c->ev_fft) (void)hipEventDestroy(c->ev_fft);
    if (c->ev_X) (void)hipEventDestroy(c->ev_X);
    for (auto &v : c->ev)
        for (auto e : v) (void)hipEventDestroy(e);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->d_Gs) (void)hipFree(c->d_Gs);
    if (c->d_Qs) (void)hipFree(c->d_Qs);
    delete c->bank;
    delete c;
    return MFB_OK;
}

extern "C" int mfb_set_stream(mfb_ctx *c, void *s) {
    if (!c) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    ++c->epoch;
    return MFB_OK;
}

// A share of the device's compute units for this handle's launches (include/mfbank.h): the handle's own stream is replaced by one
// created with a CU mask -- compute unit i belongs to part i % parts.
extern "C" int mfb_set_cu_share(mfb_ctx *c, int part, int parts) {
    if (!c || parts < 1 || parts > 16 || part < 0 || part >= parts) return MFB_ERR_ARG;
    for (int s = 0; s < 2; ++s)
        if (c->flight[s].active) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));
    hipStream_t fresh = nullptr;
    if (parts == 1) {
        HIPCHK(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    } else {
        const int words = (c->num_cus + 31) / 32;
        std::vector<uint32_t> mask((size_t)words, 0u);
        int mine = 0;
        for (int i = 0; i < c->num_cus; ++i)
            if (i % parts == part) {
                mask[(size_t)i / 32] |= 1u << (i % 32);
                ++mine;
            }
        if (!mine) return MFB_ERR_ARG;
        HIPCHK(hipExtStreamCreateWithCUMask(&fresh, (uint32_t)words, mask.data()));
    }
    const bool was_own = c->stream == c->own_stream;
    if (c->own_stream) HIPCHK(hipStreamDestroy(c->own_stream));
    c->own_stream = fresh;
    if (was_own) c->stream = fresh;
    c->cu_part = part;
    c->cu_parts = parts;
    ++c->epoch;
    return MFB_OK;
}

extern "C" int mfb_set_tuning(mfb_ctx *c, int chunk, int mpb, int rows_per_block, int jsplit) {
    if (!c || chunk < 0 || mpb < 0 || rows_per_block < 0 || jsplit < 0) return MFB_ERR_ARG;
    ++c->epoch;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));
    if (rows_per_block > 0) set_rows_per_block(c, rows_per_block);
    if (jsplit > 0) c->jsplit = jsplit;
    if (chunk > 0) {
        c->chunk = chunk > c->Dtot ? c->Dtot : chunk;
        c->chunk_auto = false;
    }
    if (c->chunk * c->MU > 65535) c->chunk = 65535 / c->MU;  // grid.y limit of pass 2
    if (mpb > 0) c->mpb = mpb > c->M ? c->M : mpb;
    if (!c->have_filters) return MFB_OK;   // the intermediate is sized when the filters arrive
    return alloc_Z(c);
}
extern "C" int mfb_get_tuning(mfb_ctx *c, int *chunk, int *mpb, int *rows_per_block, int *jsplit) {
    if (!c) return MFB_ERR_ARG;
    if (chunk) *chunk = c->chunk;
    if (mpb) *mpb = c->mpb;
    if (rows_per_block) *rows_per_block = c->srb;
    if (jsplit) *jsplit = c->jsplit;
    return MFB_OK;
}

extern "C" int mfb_get_info(mfb_ctx *c, int *N1, int *N2, int *unique_filters) {
    if (!c) return MFB_ERR_ARG;
    if (N1) *N1 = c->N1;
    if (N2) *N2 = c->N2;
    if (unique_filters) *unique_filters = c->MU;
    return MFB_OK;
}

// ---- search path ---------------------------------------------------------------------------------
// Valid outputs per complete segment: the largest multiple of NT = L / (points per lane) not above L - T + 1, so that
// they are whole register slots (seg_kernels.hpp); 0 if fewer than half of a segment would be valid.
#define SEG_LOG2L_MAX 13      // 8192-point segments: filters of up to 4097 taps
#define SEG_TWOPASS_COST 6.45  // the two-pass path in the units of seg_cost (ms per block at C2's geometry)
static int seg_valid(int l, int T) {
    const int L = 1 << l, ppl = seg_ppl(L), NT = L / ppl;
    const int pv = (L - T + 1) / NT;
    return pv >= ppl / 2 ? pv * NT : 0;
}
// Relative cost of one segment point per filter, MEASURED on MI355X at C2 (time x V / L, 48 taps, D = 256,
// M = 8; tools/seg_probe.py): the 256-point kernel is by far the cheapest per point (one twiddled pass,
// three waves per SIMD, phase table in LDS), so it wins whenever at least ~2/3 of a segment is valid; the
// cost of a valid output is this divided by the segment efficiency V / L.  Re-measured late in round 3 at the settled
// clock and with a quiet host (profiles/r03_ramp.md: the first ~30 ms after an idle spell run 8-20 % slow, and a probe whose
// own BLAS threads get the process throttled starves the device the same way -- some earlier tables carried both effects):
// GMSK bank, 256 ... 4096 points: 1.597 / 2.128 / 2.070 / 2.443 / 2.648 ms; CC11xx bank (384 taps) on five devices:
// 2048 points 2.69-2.77 ms, 4096 points 2.90-2.96 ms (the 2048-point kernel gained 5 % from one team per workgroup);
// BPSK bank (80 taps), 256 / 512 / 1024 points: 3.74 / 4.28 / 4.15 ms.  Figures with 32 workgroups per CU in the grid.
// Round 4: the 2048-point kernel became one wave per segment with one LDS exchange per transform (seg_kernels.hpp, W32):
// 2.51 against 2.72 ms on one device for the 384-tap bank (profiles/r04_long_filter.md), hence 2.20 -> 2.03.  8192 points (one
// eight-wave team per CU, four passes): 4.67 ms at 2049 taps, 5.05 at 2050, 5.48 at 3000, 6.61 at 4097 -> 3.45 per point; in
// these units the two-pass path costs 6.45 whatever the filter (6.5 ms), so the longest filters of the range stay there.
static double seg_cost(int l, int T, int MU = 8) {
    // Round 6 (fused transforms, the shift on the filters' side for 256 and 2048 points; gpurun_out/r06_taps.txt: random banks at C2,
    // lengths forced beside the chosen one): 256 points 1.49 ms at 64 taps, 1.74 at 96 -> 1.10; 2048 points 1.90 at 100 taps, 2.28 at 512,
    // 2.68 at 768, 3.24 at 1025 -> 1.75 (1.62 ... 1.78: dead register slots are pruned as the valid share shrinks); 4096 points 2.98 at
    // 768 taps, 3.53 at 1150 -> 2.43.  The 2048-point kernel now keeps 650 ... ~850 taps that went to 4096 points (2.68 against 2.98 ms
    // at 768 taps); the hand-over from 256 points stays just under 100 taps (1.91 against 1.90 ms at 100).
    // ... and once more with the complement second pass (gpurun_out/r06_taps2.txt), whose cost falls with the share of valid slots:
    // 256 points 1.28 ms at 48 taps, 1.44 at 64, 1.60 at 80, 1.75 at 88 / 96 (10 valid slots: no complement) -> 1.08; 2048 points 1.45 ms
    // at 64 taps, 1.60 at 72 ... 100, 2.45 at 640, 2.69 at 768, 3.16 at 900 -> 1.60.  The hand-over from 256 to 2048 points moves from
    // just under 100 taps to 81 (1.75 against 1.60 ms at 88 and 96 taps; a tie at 64 ... 80), 2048 points keep everything up to 1025.
    static const double per_point[14] = {0, 0, 0, 0, 0, 0, 0, 0, MFB_FFT_FUSED ? (MFB_SEG_COMPLEMENT ? 1.08 : 1.10) : 1.35, 1.94, 2.02,
                                         MFB_SEG_W32 ? (MFB_FFT_FUSED ? (MFB_SEG_COMPLEMENT ? 1.60 : 1.75) : 2.03) : 2.20,
                                         MFB_FFT_FUSED ? 2.43 : 2.55, 3.45};
    const int V = seg_valid(l, T);
    if (!V) return 1e30;
    // (the 2048-point kernel takes the filter-side shift and the complement pass for up to 8 filters per bin only: with more it runs
    // round 6's fused transforms on seg_body -- 2.32 ms on the 384-tap bank, 1.89 per point)
    const double pp = (l == 11 && MFB_SEG_W32 && MFB_FFT_FUSED && MU > 8) ? 1.89 : per_point[l];
    return pp * (double)(1 << l) / (double)V;
}

static int choose_segl(const mfb_ctx *c, int T, bool may_decline) {
    int best = 0;
    double bc = 1e30;
    for (int l = 8; l <= SEG_LOG2L_MAX; ++l) {
        if (!seg_valid(l, T) || (4 << l) > c->N) continue;    // at least half of every segment valid, >= 4 segments
        const double cost = seg_cost(l, T, c->MU);
        if (cost < bc) {
            bc = cost;
            best = l;
        }
    }
    // barely half of an 8192-point segment valid: the two-pass path is no slower (unless the caller insists on segments)
    if (may_decline && bc > SEG_TWOPASS_COST) return 0;
    return best;
}

static int fsm_prepare_eager(mfb_ctx *c);
// settle path and segment length from the request and the analysed bank; (re)build G and W_L
static int resolve_path(mfb_ctx *c) {
    ++c->epoch;               // whatever changes here changes the launches of a block
    if (!c->have_filters || !c->bank) return MFB_OK;
    const int T = c->bank->T;
    int l = 0;
    if (c->path_req != MFB_PATH_TWOPASS) {
        if (c->segl_req) {
            if (c->segl_req >= 8 && c->segl_req <= SEG_LOG2L_MAX && seg_valid(c->segl_req, T) && (2 << c->segl_req) <= c->N) l = c->segl_req;
        } else {
            l = choose_segl(c, T, c->path_req == MFB_PATH_AUTO);
        }
    }
    if (c->path_req == MFB_PATH_SEGMENT && !l) return MFB_ERR_UNSUPPORTED;
    HIPCHK(sync_streams(c));
    if (!l) {
        c->path = MFB_PATH_TWOPASS;
        c->segl = 0;
        c->basis = MFB_BASIS_FILTERS;
    } else {
        const int L = 1 << l;
        if (c->segl != l || !c->d_G) {
            std::vector<float> G;
            taps::segment_spectra(*c->bank, L, L - seg_valid(l, T) + 1, &G, seg_ppl(L));
            if (c->d_G) HIPCHK(hipFree(c->d_G));
            c->d_G = nullptr;
            HIPCHK(dev_alloc((void **)&c->d_G, G.size() * sizeof(float)));
            HIPCHK(hipMemcpy(c->d_G, G.data(), G.size() * sizeof(float), hipMemcpyHostToDevice));
            if (c->twL_len != L) {
                if (c->d_twL) HIPCHK(hipFree(c->d_twL));
                c->d_twL = nullptr;
                c->twL_len = 0;
                std::vector<cf> t;
                make_twiddles(t, L, (double)L, 1.0);
                append_fused_tables(t, L);
                int rc = upload_tw(&c->d_twL, t);
                if (rc) return rc;
                c->twL_len = L;
            }
        }
        // opt-in span basis of the SUM_ALL search (filter_taps.hpp): rank(C) filters instead of M
        c->basis = MFB_BASIS_FILTERS;
        if (c->basis_req == MFB_BASIS_SPAN && c->sum_all) {
            if (c->gb_l != l || !c->d_Gb) {
                taps::Bank sb;
                c->MB = taps::span_basis(*c->bank, &sb);
                std::vector<float> G;
                if (c->MB > 0) taps::segment_spectra(sb, L, L - seg_valid(l, T) + 1, &G, seg_ppl(L));
                if (c->d_Gb) HIPCHK(hipFree(c->d_Gb));
                c->d_Gb = nullptr;
                c->gb_l = 0;
                if (c->MB > 0) {
                    HIPCHK(dev_alloc((void **)&c->d_Gb, G.size() * sizeof(float)));
                    HIPCHK(hipMemcpy(c->d_Gb, G.data(), G.size() * sizeof(float), hipMemcpyHostToDevice));
                    c->gb_l = l;
                }
            }
            if (c->MB > 0) c->basis = MFB_BASIS_SPAN;
        }
        c->path = MFB_PATH_SEGMENT;
        c->segl = l;
        c->V = seg_valid(l, T);
        c->T = L - c->V + 1;          // taps the segments are laid out for (>= the bank's T)
        c->win_start = c->bank->start;
        c->Q = (c->N + c->V - 1) / c->V;
    }
    if (c->path == MFB_PATH_TWOPASS) {
        if (c->chunk_auto) c->chunk = default_chunk(c);
        if (c->chunk * c->MU > 65535) c->chunk = 65535 / c->MU;
        int rc = reserve_partials(c, (size_t)c->Dtot * c->M * c->N1);
        if (rc) return rc;
    } else if (c->d_Z && c->z_rows > 1) {   // give the big intermediate back
        HIPCHK(hipFree(c->d_Z));
        c->d_Z = nullptr;
        c->z_rows = 0;
    }
    int rc = alloc_Z(c);
    if (rc) return rc;
    return fsm_prepare_eager(c);
}

static int set_search_path_body(mfb_ctx *c, int path, int log2L, int wg_per_cu, int filters_per_pass) {
    if (!c || path < 0 || path > MFB_PATH_SEGMENT || log2L < 0 || wg_per_cu < 0 || wg_per_cu > 64 || filters_per_pass < 0 ||
        filters_per_pass > SEG_MPB_MAX)
        return MFB_ERR_ARG;
    if (log2L && (log2L < 8 || log2L > SEG_LOG2L_MAX)) return MFB_ERR_UNSUPPORTED;
    HIPCHK(hipSetDevice(c->device));
    const int old_path = c->path_req, old_l = c->segl_req, old_wpc = c->seg_wpc, old_mpb = c->seg_mpb;
    c->path_req = path;
    c->segl_req = log2L;
    c->seg_wpc = wg_per_cu;
    c->seg_mpb = filters_per_pass;
    const int rc = resolve_path(c);
    if (rc) {           // keep the previous, working configuration
        c->path_req = old_path;
        c->segl_req = old_l;
        c->seg_wpc = old_wpc;
        c->seg_mpb = old_mpb;
        (void)resolve_path(c);
    }
    return rc;
}
extern "C" int mfb_set_search_path(mfb_ctx *c, int path, int log2L, int wg_per_cu, int filters_per_pass) {
    return guarded([&] { return set_search_path_body(c, path, log2L, wg_per_cu, filters_per_pass); });
}

static int set_search_basis_body(mfb_ctx *c, int basis) {
    if (!c || (basis != MFB_BASIS_FILTERS && basis != MFB_BASIS_SPAN)) return MFB_ERR_ARG;
    if (basis == MFB_BASIS_SPAN && !c->sum_all) return MFB_ERR_STATE;   // per-filter sums need every filter
    HIPCHK(hipSetDevice(c->device));
    c->basis_req = basis;
    return resolve_path(c);
}
extern "C" int mfb_set_search_basis(mfb_ctx *c, int basis) {
    return guarded([&] { return set_search_basis_body(c, basis); });
}
static int set_search_mode_body(mfb_ctx *c, int mode) {
    if (!c || (mode != MFB_SEARCH_TRANSFORMS && mode != MFB_SEARCH_ENERGY)) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    c->search_mode = mode;
    return resolve_path(c);      // the two-pass intermediate is sized for what the search needs
}
extern "C" int mfb_set_search_mode(mfb_ctx *c, int mode) {
    return guarded([&] { return set_search_mode_body(c, mode); });
}
extern "C" int mfb_get_search_mode(mfb_ctx *c, int *mode) {
    if (!c || !mode) return MFB_ERR_ARG;
    *mode = c->search_mode;
    return MFB_OK;
}
extern "C" int mfb_get_search_basis(mfb_ctx *c, int *basis, int *transformed_filters) {
    if (!c) return MFB_ERR_ARG;
    if (basis) *basis = c->basis;
    if (transformed_filters) *transformed_filters = (c->path == MFB_PATH_SEGMENT && c->basis == MFB_BASIS_SPAN) ? c->MB : c->MU;
    return MFB_OK;
}
static int analyze_rank_body(const float *masks, int M, int N, int *rank) {
    if (!masks || !rank || M < 1 || N < 2 || (N & (N - 1))) return MFB_ERR_ARG;
    taps::Bank b, sb;
    taps::analyse(masks, M, N, &b);
    if (b.T > taps::ROW_TAPS_MAX) {
        *rank = M;          // no short support: not analysed
        return MFB_OK;
    }
    *rank = taps::span_basis(b, &sb);
    return MFB_OK;
}
extern "C" int mfb_analyze_rank(const float *masks, int M, int N, int *rank) {
    return guarded([&] { return analyze_rank_body(masks, M, N, rank); });
}

extern "C" int mfb_get_search_path(mfb_ctx *c, int *path, int *log2L, int *taps_out, int *valid_per_segment, int *segments) {
    if (!c) return MFB_ERR_ARG;
    if (path) *path = c->path;
    if (log2L) *log2L = c->segl;
    if (taps_out) *taps_out = c->bank ? c->bank->T : 0;
    if (valid_per_segment) *valid_per_segment = c->path == MFB_PATH_SEGMENT ? c->V : 0;
    if (segments) *segments = c->path == MFB_PATH_SEGMENT ? c->Q : 0;
    return MFB_OK;
}

static int analyze_filters_body(const float *masks, int M, int N, int *support_start, int *support_len) {
    if (!masks || M < 1 || N < 2 || (N & (N - 1))) return MFB_ERR_ARG;
    taps::Bank b;
    taps::analyse(masks, M, N, &b);
    if (support_start) *support_start = b.start;
    if (support_len) *support_len = b.T;
    return MFB_OK;
}
extern "C" int mfb_analyze_filters(const float *masks, int M, int N, int *support_start, int *support_len) {
    return guarded([&] { return analyze_filters_body(masks, M, N, support_start, support_len); });
}

static int set_filters_body(mfb_ctx *c, const float *masks, int M, int N) {
    if (!c || !masks) return MFB_ERR_ARG;
    if (M != c->M || N != c->N) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_masks, masks, (size_t)M * N * sizeof(cf), hipMemcpyHostToDevice, c->stream));
    // Filters that are exact copies or exact negatives of an earlier one (the BPSK bank: pattern p
    // and its complement) give bit-identical |.|^2; the Doppler search transforms each such class once.
    std::vector<int> uniq, rep(M);
    for (int m = 0; m < M; ++m) {
        const float *rm = masks + (size_t)m * 2 * N;
        int found = -1;
        for (size_t u = 0; u < uniq.size() && found < 0; ++u) {
            const float *ru = masks + (size_t)uniq[u] * 2 * N;
            bool same = true, neg = true;
            for (size_t i = 0; i < (size_t)2 * N && (same || neg); ++i) {
                same = same && (rm[i] == ru[i]);
                neg = neg && (rm[i] == -ru[i]);
            }
            if (same || neg) found = (int)u;
        }
        if (found < 0) {
            found = (int)uniq.size();
            uniq.push_back(m);
        }
        rep[m] = found;
    }
    c->MU = (int)uniq.size();
    c->h_uniq = uniq;
    c->h_mult.assign(uniq.size(), 0);
    for (int m = 0; m < M; ++m) ++c->h_mult[(size_t)rep[m]];
    c->gs_rows = 0;
    HIPCHK(hipMemcpyAsync(c->d_uniq, uniq.data(), uniq.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_rep, rep.data(), (size_t)M * sizeof(int), hipMemcpyHostToDevice, c->stream));
    // impulse-response window and taps of every filter (double precision, host threads)
    if (!c->bank) c->bank = new taps::Bank();
    taps::analyse(masks, M, N, c->bank);
    HIPCHK(sync_streams(c));
    c->have_filters = true;
    c->have_xc = false;
    c->W_valid = false;
    ++c->epoch;
    c->segl = 0;    // force G to be rebuilt for the new bank
    c->gb_l = 0;
    int rc = resolve_path(c);
    if (rc == MFB_ERR_UNSUPPORTED) {   // a segment path was demanded but this bank has no short support
        c->path_req = MFB_PATH_AUTO;
        rc = resolve_path(c);
    }
    if (rc) c->have_filters = false;
    return rc;
}
extern "C" int mfb_set_filters(mfb_ctx *c, const float *masks, int M, int N) {
    bool thrown = true;
    const int rc = guarded([&] {
        const int r = set_filters_body(c, masks, M, N);
        thrown = false;
        return r;
    });
    if (thrown && c) c->have_filters = false;      // half a bank: the next mfb_set_filters starts over
    return rc;
}

static int set_shifts_body(mfb_ctx *c, const int32_t *shifts, int count) {
    if (!c || !shifts || count != c->Dtot) return MFB_ERR_ARG;
    for (int i = 0; i < count; ++i)
        if (shifts[i] < 0 || shifts[i] >= c->N) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_shifts, shifts, (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(sync_streams(c));
    // (a caller that sets the same table again -- a tracking loop, a re-attached shard -- keeps the per-bin spectra: 35 ms at C2)
    const bool same = (int)c->h_shifts.size() == count && std::equal(shifts, shifts + count, c->h_shifts.begin());
    c->h_shifts.assign(shifts, shifts + count);
    if (!same) c->gs_rows = 0;    // the per-bin spectra belong to the old table
    c->have_shifts = true;
    ++c->epoch;
    return fsm_prepare_eager(c);
}
extern "C" int mfb_set_shifts(mfb_ctx *c, const int32_t *shifts, int count) {
    return guarded([&] { return set_shifts_body(c, shifts, count); });
}

extern "C" int mfb_input_buffer(mfb_ctx *c, float **p) {
    if (!c || !p) return MFB_ERR_ARG;
    *p = (float *)c->h_in;
    return MFB_OK;
}

// ---- launch helpers ------------------------------------------------------------------------------
static int fin_threads(int rows) { return 64 * (rows < 1 ? 1 : (rows > 16 ? 16 : rows)); }   // k_finalize: one wave per row
static hipEvent_t get_event(mfb_ctx *c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
// Profiling marks come in pairs (begin, end); a mark that cannot be created or recorded switches profiling off and drops
// the marks taken so far, so that mfb_profile_read never pairs events of different launches.
static void prof_mark(mfb_ctx *c, int which) {
    if (!c->prof) return;
    hipEvent_t e = get_event(c);
    if (!e || hipEventRecord(e, c->stream) != hipSuccess) {
        fprintf(stderr, "mfbank: profiling event unavailable, profiling disabled\n");
        if (e) c->ev_pool.push_back(e);
        for (auto &v : c->ev) {
            for (auto q : v) c->ev_pool.push_back(q);
            v.clear();
        }
        c->prof = false;
        return;
    }
    c->ev[which].push_back(e);
}

template <int L1, int KIND>
static int launch_p1_t(mfb_ctx *c, const P1Args &a, dim3 grid) {
    const size_t lds = P1Cfg<L1>::lds_bytes;   // > 48 KiB cases: set_kernel_attributes (per device)
    hipLaunchKernelGGL((k_pass1<L1, KIND>), grid, dim3((L1 / 16) * TILE), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return MFB_OK;
}
template <int KIND>
static int launch_p1(mfb_ctx *c, const P1Args &a, dim3 grid) {
    switch (c->l1) {
        case 5: return launch_p1_t<32, KIND>(c, a, grid);
        case 6: return launch_p1_t<64, KIND>(c, a, grid);
        case 7: return launch_p1_t<128, KIND>(c, a, grid);
        case 8: return launch_p1_t<256, KIND>(c, a, grid);
        case 9: return launch_p1_t<512, KIND>(c, a, grid);
    }
    return MFB_ERR_UNSUPPORTED;
}

template <int L2, int MODE>
static int launch_p2_t(mfb_ctx *c, const P2Args &a, dim3 grid) {
    constexpr int NT = L2 / 16;
    constexpr int RB = P2Cfg<L2>::RB;
    const size_t lds = P2Cfg<L2>::lds_bytes;   // > 48 KiB cases: set_kernel_attributes (per device)
    hipLaunchKernelGGL((k_pass2<L2, MODE>), grid, dim3(NT * RB), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return MFB_OK;
}
template <int MODE>
static int launch_p2(mfb_ctx *c, const P2Args &a, dim3 grid) {
    switch (c->l2) {
        case 5: return launch_p2_t<32, MODE>(c, a, grid);
        case 6: return launch_p2_t<64, MODE>(c, a, grid);
        case 7: return launch_p2_t<128, MODE>(c, a, grid);
        case 8: return launch_p2_t<256, MODE>(c, a, grid);
        case 9: return launch_p2_t<512, MODE>(c, a, grid);
        case 10: return launch_p2_t<1024, MODE>(c, a, grid);
        case 11: return launch_p2_t<2048, MODE>(c, a, grid);
        case 12: return launch_p2_t<4096, MODE>(c, a, grid);
        case 13: return launch_p2_t<8192, MODE>(c, a, grid);
    }
    return MFB_ERR_UNSUPPORTED;
}

static P1Args p1_base(mfb_ctx *c) {
    P1Args a;
    memset(&a, 0, sizeof(a));
    a.masks = c->d_masks;
    a.Z = c->use_z2 ? c->d_Z2 : c->d_Z;
    a.tw1 = c->d_tw1;
    a.twLo = c->d_twLo;
    a.twHi = c->d_twHi;
    a.N = c->N;
    a.N2 = c->N2;
    a.lo = c->lo;
    a.M = c->M;
    a.mpb = c->mpb;
    return a;
}
// rows per workgroup for the STORE transforms (1 or M rows only): small, so the launch fills the chip
static void p2_store_split(mfb_ctx *c, P2Args &b, int rows) {
    const int NT = c->N2 / 16;
    const int RB = NT >= 256 ? 1 : 256 / NT;
    int srb = rows >= 4 ? 2 : 1;
    if (srb < RB) srb = RB;
    if (srb > c->N1) srb = c->N1;
    b.srb = srb;
    b.parts = c->N1 / srb;
}

static P2Args p2_base(mfb_ctx *c) {
    P2Args a;
    memset(&a, 0, sizeof(a));
    a.Z = c->use_z2 ? c->d_Z2 : c->d_Z;
    a.tw2 = c->d_tw2;
    a.N = c->N;
    a.N1 = c->N1;
    a.N2 = c->N2;
    a.srb = c->srb;
    a.parts = c->parts;
    a.scale = 1.0f / 262144.0f;
    return a;
}

static int launch_transpose(mfb_ctx *c, cf *dst, int rows, int conj) {
    dim3 grid((c->N2 + 31) / 32, (c->N1 + 31) / 32, rows);
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, c->stream, (const cf *)(c->use_z2 ? c->d_Z2 : c->d_Z), dst, c->N1, c->N2, conj);
    HIPCHK(hipGetLastError());
    return MFB_OK;
}

// forward FFT of `rows` rows (a batch of blocks: row r reads its input in_stride elements behind row r - 1 and goes to
// dst + r * N): complex (src_c) or real (src_r) input -> dst natural order.  d_Z must hold `rows` rows.
static int forward_fft(mfb_ctx *c, const cf *src_c, const float *src_r, cf *dst, int conj_out = 1, int rows = 1, size_t in_stride = 0);
static int forward_fft(mfb_ctx *c, const cf *src_c, const float *src_r, cf *dst, int conj_out, int rows, size_t in_stride) {
    if (!(c->use_z2 ? c->z2_rows : c->z_rows)) return MFB_ERR_STATE;       // the transforms' intermediate arrives with the filters
    if (rows < 1 || (size_t)rows > (c->use_z2 ? c->z2_rows : c->z_rows) || in_stride > 0x7fffffffu) return MFB_ERR_ARG;
    P1Args a = p1_base(c);
    a.X = src_c;
    a.Xr = src_r;
    a.ntiles = c->N2 / TILE;
    a.mgroups = 1;
    a.jsplit = rows;
    a.in_stride = (int)in_stride;
    dim3 g1(a.ntiles * rows, 1, 1);
    int rc = src_c ? launch_p1<KIND_FWDC>(c, a, g1) : launch_p1<KIND_FWDR>(c, a, g1);
    if (rc) return rc;
    P2Args b = p2_base(c);
    p2_store_split(c, b, 1);
    dim3 g2(b.parts, rows, 1);
    rc = launch_p2<MODE_STORE>(c, b, g2);
    if (rc) return rc;
    return launch_transpose(c, dst, rows, conj_out);   // conj_out = 0 leaves conj(fft(src)) (real input only)
}


// ---- single-pass overlap-save launches -------------------------------------------------------------
template <int L, int MODE, int PV>
static int launch_seg_k(mfb_ctx *c, const SegArgs &a, int grid) {
    const size_t lds = SegCfg<L>::lds_bytes(MODE == SEG_REDUCE ? a.mpb : 0);
    hipLaunchKernelGGL((k_seg<L, MODE, PV>), dim3(grid), dim3(SegCfg<L>::BLOCK), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return MFB_OK;
}
// pv: valid register slots of a complete segment (half of the points per lane ... all of them) for the branch-free
// instantiation, -1 = masked
template <int L, int PV0, int PV1>
static int launch_seg_pv(mfb_ctx *c, const SegArgs &a, int grid, int pv) {
    if constexpr (PV0 > PV1) {
        return MFB_ERR_UNSUPPORTED;
    } else {
        if (pv == PV0) return launch_seg_k<L, SEG_REDUCE, PV0>(c, a, grid);
        return launch_seg_pv<L, PV0 + 1, PV1>(c, a, grid, pv);
    }
}
template <int L>
static int launch_seg_l(mfb_ctx *c, const SegArgs &a, int grid, int mode, int pv) {
    if (mode == SEG_STORE) return launch_seg_k<L, SEG_STORE, -1>(c, a, grid);
    if constexpr (seg_ppl(L) == 32) {
        if (pv == -1) return launch_seg_k<L, SEG_REDUCE, -1>(c, a, grid);
        return launch_seg_pv<L, 16, 32>(c, a, grid, pv);
    } else {
        if (pv == -1) return launch_seg_k<L, SEG_REDUCE, -1>(c, a, grid);
        return launch_seg_pv<L, 8, 16>(c, a, grid, pv);
    }
}
static int launch_seg(mfb_ctx *c, const SegArgs &a, int grid, int mode, int pv) {
    switch (c->segl) {
        case 8: return launch_seg_l<256>(c, a, grid, mode, pv);
        case 9: return launch_seg_l<512>(c, a, grid, mode, pv);
        case 10: return launch_seg_l<1024>(c, a, grid, mode, pv);
        case 11: return launch_seg_l<2048>(c, a, grid, mode, pv);
        case 12: return launch_seg_l<4096>(c, a, grid, mode, pv);
        case 13: return launch_seg_l<8192>(c, a, grid, mode, pv);
    }
    return MFB_ERR_UNSUPPORTED;
}

// Decomposition of one launch over `nslots` slots (a slot = CT neighbouring segments, one team
// iteration): nsg segment groups (8 = one per XCD under round-robin placement), each with wpg workgroups
// = bsplit Doppler streams x ssplit slot sub-ranges (x TPW teams).
struct SegGeom {
    int NT, TEAM, CT, TPW, WPT;
    bool wave_sync;
};
static SegGeom seg_geom(const mfb_ctx *c) {
    SegGeom g;
    const int L = 1 << c->segl;
    g.NT = L / seg_ppl(L);
    g.TEAM = g.NT < 64 ? 64 : g.NT;
    g.CT = g.TEAM / g.NT;
    g.TPW = seg_block_threads(g.NT) / g.TEAM;
    g.WPT = g.TEAM / 64;
    g.wave_sync = g.NT <= 64;
    return g;
}
struct SegPlan {
    int nsg, wpg, bsplit, ssplit, mpb, mgroups, grid;
    int igroups;      // > 1: the filter groups are walked inside a team (REDUCE launches of large problems), mgroups == 1
};
static SegPlan plan_seg(const mfb_ctx *c, int dc, int nfilters, int nslots, int mpb_want, bool inner_groups = false) {
    const SegGeom g = seg_geom(c);
    SegPlan p;
    p.nsg = nslots >= 64 ? 8 : 1;
    // L = 256 runs three workgroups per CU only while a workgroup's LDS stays under 53 KiB: 8 filters per pass
    // (the wave-local 2048-point kernel keeps two workgroups per CU only while a workgroup's LDS stays under 80 KiB: 8 per pass)
    // (so does the 8192-point kernel: its one team per CU has 157 KiB with 8 filters per pass)
    int mpb = mpb_want > 0 ? mpb_want : ((c->segl <= 8 || c->segl == 13 || seg_ppl(1 << c->segl) == 32) ? 8 : SEG_MPB_MAX);
    if (c->segl == 13 && mpb > 8) mpb = 8;
    if (mpb > SEG_MPB_MAX) mpb = SEG_MPB_MAX;
    if (mpb > nfilters) mpb = nfilters;
    p.mgroups = (nfilters + mpb - 1) / mpb;
    p.mpb = (nfilters + p.mgroups - 1) / p.mgroups;   // balanced
    p.igroups = 1;
    // Workgroups in the grid per CU.  More than are resident at once (3 / 2 per CU): the surplus is dealt out as workgroups
    // retire, which evens out CUs that finish at different times.  Measured at C2, L = 256, settled clock, quiet host
    // (tools/seg_probe.py): 8 per CU 1.615 ms, 12: 1.615, 16: 1.613, 24: 1.604, 32: 1.596, 48: 1.595 -- about 1 %.  (Round 2's
    // table here -- 12: 1.90, 16: 1.71, 32: 1.655 "on an uneven device" -- was mostly the probe: its first variant ran in the
    // clock ramp after the handle was built.)  Small blocks are unaffected (the grid never has more teams than (bin, slot) units).
    const int wpc = c->seg_wpc > 0 ? c->seg_wpc : 32;
    // workgroups per group: wpc 256-thread workgroups' worth of teams per CU, over the device's CUs
    const int teams = wpc * c->num_cus * 256 / g.TEAM;         // teams in the whole grid
    int wpg = teams / g.TPW / p.nsg;
    if (wpg < 1) wpg = 1;
    // never more teams than (bin, slot) units in a group
    const long long units = (long long)dc * ((nslots + p.nsg - 1) / p.nsg);
    while (wpg > 1 && (long long)wpg * g.TPW > units) wpg >>= 1;
    // Several filter groups and plenty of (bin, slot) units: a team walks the groups itself on ONE forward transform per
    // segment instead of a workgroup per group each repeating it (1 of 18 transforms of the BPSK bank).  REDUCE launches only.
    if (inner_groups && c->segl <= 8 && p.mgroups > 1 && units >= 4LL * wpg * g.TPW) {      // (the 256-point kernel only: seg_kernels.hpp)
        p.igroups = p.mgroups;
        p.mgroups = 1;
    }
    p.wpg = wpg;
    // bsplit * ssplit = wpg * TPW teams; barrier teams of one workgroup must share the Doppler stream.
    // Of all factorisations take the one whose busiest team has the least (bins x slots) to do;
    // ties go to more Doppler streams (fewer partial-sum columns).
    const int tg = g.wave_sync ? wpg * g.TPW : wpg;
    const int glen = (nslots + p.nsg - 1) / p.nsg;
    int bs = 1;
    long long best = -1;
    for (int d = 1; d <= tg; ++d) {
        if (tg % d) continue;
        const int ss = g.wave_sync ? tg / d : (wpg / d) * g.TPW;
        const long long work = (long long)((dc + d - 1) / d) * ((glen + ss - 1) / ss);
        if (best < 0 || work <= best) {
            best = work;
            bs = d;
        }
    }
    p.bsplit = bs;
    p.ssplit = g.wave_sync ? tg / bs : (wpg / bs) * g.TPW;
    p.grid = p.nsg * p.mgroups * p.wpg;
    return p;
}

static SegArgs seg_base(mfb_ctx *c, const SegPlan &p) {
    SegArgs a;
    memset(&a, 0, sizeof(a));
    a.x = c->d_in;
    a.G = c->d_G;
    a.Grows = c->M;
    a.twL = c->d_twL;
    a.twLo = c->d_twLo;
    a.twHi = c->d_twHi;
    a.N = c->N;
    a.lo = c->lo;
    a.V = c->V;
    a.mpb = p.mpb;
    a.mgroups = p.mgroups;
    a.igroups = p.igroups;
    a.nsg = p.nsg;
    a.bsplit = p.bsplit;
    a.ssplit = p.ssplit;
    a.scale = 1.0f / 262144.0f;
    return a;
}

// slots whose segments are all complete (branch-free kernel) and the rest (masked kernel)
static void seg_slots(const mfb_ctx *c, int *full, int *total) {
    const SegGeom g = seg_geom(c);
    const int qfull = c->N / c->V;                    // complete segments
    *full = qfull / g.CT;
    *total = (c->Q + g.CT - 1) / g.CT;
}


#define MIRROR_MAX_N (1 << 18)
// d_X is about to be overwritten: the copy of the previous spectrum must have left it
static int before_fft(mfb_ctx *c) {
    if (c->mirror_valid) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_X, 0));
    c->mirror_valid = false;
    return MFB_OK;
}
// d_X holds a new spectrum: start its copy to the host mirror beside whatever the handle's stream does next
static int after_fft(mfb_ctx *c) {
    if (!c->mirror) return MFB_OK;
    HIPCHK(hipEventRecord(c->ev_fft, c->stream));
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_fft, 0));
    HIPCHK(hipMemcpyAsync(c->h_X, c->d_X, (size_t)c->N * sizeof(cf), hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipEventRecord(c->ev_X, c->copy_stream));
    c->mirror_valid = true;
    return MFB_OK;
}
static int enable_mirror(mfb_ctx *c) {
    if (c->mirror || c->N > MIRROR_MAX_N) return MFB_OK;
    if (!c->h_X) HIPCHK(hipHostMalloc((void **)&c->h_X, (size_t)c->N * sizeof(cf), hipHostMallocDefault));
    if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->ev_fft) HIPCHK(hipEventCreateWithFlags(&c->ev_fft, hipEventDisableTiming));
    if (!c->ev_X) HIPCHK(hipEventCreateWithFlags(&c->ev_X, hipEventDisableTiming));
    c->mirror = true;
    return MFB_OK;
}

extern "C" int mfb_upload(mfb_ctx *c) {
    if (!c) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_x, c->h_in, (size_t)c->N * sizeof(cf), hipMemcpyHostToDevice, c->stream));
    c->d_in = c->d_x;
    int rc = before_fft(c);
    if (rc) return rc;
    rc = forward_fft(c, c->d_in, nullptr, c->d_X);
    if (rc) return rc;
    if ((rc = after_fft(c))) return rc;
    c->have_input = true;
    c->have_xc = false;
    return MFB_OK;
}

extern "C" int mfb_upload_from(mfb_ctx *c, const float *host, int N) {
    if (!c || !host || N != c->N) return MFB_ERR_ARG;
    if ((const void *)host != (const void *)c->h_in) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(sync_streams(c));  // pinned buffer may still be in flight
        memcpy(c->h_in, host, (size_t)N * sizeof(cf));
    }
    return mfb_upload(c);
}

extern "C" int mfb_upload_device(mfb_ctx *c, const void *dev) {
    if (!c || !dev) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    c->d_in = (const cf *)dev;
    int rc = before_fft(c);
    if (rc) return rc;
    rc = forward_fft(c, c->d_in, nullptr, c->d_X);
    if (rc) return rc;
    if ((rc = after_fft(c))) return rc;
    c->have_input = true;
    c->have_xc = false;
    return MFB_OK;
}

// ---- the search with the shift on the filter side (seg_kernels.hpp, segf_body) ------------------------------------------------------
// MFB_SEG_FSM=0 in the environment keeps the search on seg_body (the A/B switch of profiles/r06_fft_ops.md); MFB_SEG_FSM_RECT="fb,fs"
// sets the rectangle a wave takes (bins x slots).
static bool fsm_enabled() {
    static const int on = getenv("MFB_SEG_FSM") ? atoi(getenv("MFB_SEG_FSM")) : 1;
    return on != 0 && MFB_FFT_FUSED;
}
// per-bin spectra of the bank in force (the unique filters, or the span basis), built when the shifts or the filters have changed
static int fsm_prepare(mfb_ctx *c, bool span, int rows) {
    if (c->d_Gs && c->gs_rows == rows && c->gs_span == (span ? 1 : 0) && c->gs_l == c->segl) return MFB_OK;
    if ((int)c->h_shifts.size() != c->Dtot || !c->bank) return MFB_ERR_STATE;
    std::vector<float> G, Q;
    // SUM_ALL searches: the per-bin table of sum_rows w |G|^2 (segf_body, SUMQ); w = how often k_finalize counts the row, relative to
    // row 0, whose partial sums carry the bin's total
    std::vector<float> *q = (c->sum_all && MFB_SEG_SUMQ) ? &Q : nullptr;
    const int L = 1 << c->segl, Te = L - seg_valid(c->segl, c->bank->T) + 1;
    if (span) {
        taps::Bank sb;
        if (taps::span_basis(*c->bank, &sb) != rows) return MFB_ERR_STATE;
        taps::segment_spectra_shifted(sb, nullptr, rows, c->h_shifts.data(), c->Dtot, L, Te, &G, seg_ppl(L), q, nullptr);
    } else {
        if ((int)c->h_uniq.size() != rows || (int)c->h_mult.size() != rows || c->h_mult[0] < 1) return MFB_ERR_STATE;
        std::vector<double> wts((size_t)rows);
        for (int u = 0; u < rows; ++u) wts[(size_t)u] = (double)c->h_mult[(size_t)u] / (double)c->h_mult[0];
        taps::segment_spectra_shifted(*c->bank, c->h_uniq.data(), rows, c->h_shifts.data(), c->Dtot, L, Te, &G, seg_ppl(L), q, wts.data());
    }
    HIPCHK(sync_streams(c));
    if (c->d_Gs) HIPCHK(hipFree(c->d_Gs));
    c->d_Gs = nullptr;
    c->gs_rows = 0;
    if (c->d_Qs) HIPCHK(hipFree(c->d_Qs));
    c->d_Qs = nullptr;
    HIPCHK(dev_alloc((void **)&c->d_Gs, G.size() * sizeof(float)));
    HIPCHK(hipMemcpy(c->d_Gs, G.data(), G.size() * sizeof(float), hipMemcpyHostToDevice));
    if (q) {
        HIPCHK(dev_alloc((void **)&c->d_Qs, Q.size() * sizeof(float)));
        HIPCHK(hipMemcpy(c->d_Qs, Q.data(), Q.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    c->gs_rows = rows;
    c->gs_span = span ? 1 : 0;
    c->gs_l = c->segl;
    return MFB_OK;
}
// A wave's rectangle (bins x slots) and what a group of the grid is.  The forward transform is shared by fb bins -- 1 / (fb MU) of
// the inverse work --; small rectangles keep the grid over-subscribed (the surplus evens out CUs that finish at different times, as
// in plan_seg).  Measured (profiles/r06_fft_ops.md): 128 (bin, filter) transforms of 256 points per forward transform -- 16 bins
// of the 8-filter banks, 8 of the BPSK bank's 16 --, 64 of 2048 points.  Groups (blockIdx % 8 = XCD): an eighth of the BINS when
// there are enough of them -- every XCD then keeps its share of the per-bin spectra in its L2 (C3: +4.5 %; the 2048-point kernel,
// 32 MiB of spectra at 256 bins: 2.09 against 2.49 ms) and the samples stream through all of them --, else an eighth of the slots.
struct FsmPlan {
    bool ok;
    int nsg, gbins, fb, fs, nbc, nsc;
};
static FsmPlan fsm_plan(const mfb_ctx *c, int MU, int nfull, int nb = 1) {
    static int env_fb = 0, env_fs = 0;
    static const bool env_read = [] {
        const char *e = getenv("MFB_SEG_FSM_RECT");
        if (e) sscanf(e, "%d,%d", &env_fb, &env_fs);
        return true;
    }();
    (void)env_read;
    static const int env_gb = getenv("MFB_SEG_FSM_GROUP") ? atoi(getenv("MFB_SEG_FSM_GROUP")) : -1;
    FsmPlan p;
    memset(&p, 0, sizeof(p));
    const bool w32 = c->segl == 11;
    p.nsg = nfull >= 64 ? 8 : 1;
    const int per_fwd = w32 ? 64 : 128;
    int fb = c->fsm_fb > 0 ? c->fsm_fb : (env_fb > 0 ? env_fb : (per_fwd / MU > 2 ? per_fwd / MU : 2));
    int fs = c->fsm_fs > 0 ? c->fsm_fs : (env_fs > 0 ? env_fs : 1);
    if (fb > c->Dtot) fb = c->Dtot;
    p.gbins = env_gb >= 0 ? (env_gb != 0) : (p.nsg > 1 && c->Dtot >= p.nsg * fb ? 1 : 0);
    if (p.gbins && c->Dtot < p.nsg) p.gbins = 0;
    if (w32 && !p.gbins && p.nsg > 1 && env_gb < 0) {
        // the long kernel's spectra (16 KiB per (bin, filter)) must stay in an L2: fewer bins per rectangle, or not this kernel
        fb = c->Dtot / p.nsg;
        if (fb < 1) return p;
        p.gbins = 1;
    }
    const int glen = p.gbins ? nfull : (nfull + p.nsg - 1) / p.nsg;
    const int gblen = p.gbins ? (c->Dtot + p.nsg - 1) / p.nsg : c->Dtot;
    if (fs > glen) fs = glen;
    if (fb > gblen) fb = gblen;
    if (fb < 1 || fs < 1) return p;
    // Small problems (one block of 2^15 samples x 64 bins: 2496 (bin, slot) units) must not be folded into a few long waves: the
    // rectangle shrinks until the launch has twice the waves the device holds at once -- at one bin per rectangle the forward
    // transform is no longer shared, and what is left of the gain is the mixing multiply (measured: bench.py's one-block loop at
    // 2^15 x 64 fell from 250 to 155 Msamples/s with 16-bin rectangles: 156 waves on 1024 SIMDs)
    if (c->fsm_fb <= 0 && env_fb <= 0) {
        const long long want = 2LL * c->num_cus * 4 * (w32 ? 2 : 3);
        while (fb > 1 && (long long)nb * p.nsg * ((gblen + fb - 1) / fb) * ((glen + fs - 1) / fs) < want) fb >>= 1;
    }
    p.fb = fb;
    p.fs = fs;
    p.nbc = (gblen + fb - 1) / fb;
    p.nsc = (glen + fs - 1) / fs;
    p.ok = true;
    return p;
}
template <int L, int PV0, int PV1>
static int launch_segf_pv(mfb_ctx *c, const SegFArgs &a, int grid, size_t lds, int pv) {
    if constexpr (PV0 > PV1) {
        return MFB_ERR_UNSUPPORTED;
    } else {
        if (pv == PV0) {
            bool done = false;
            if constexpr (segf_has_presum<L, PV0>()) {
                if (a.Qs && a.presum) {      // ... and the filter rows summed in registers
                    hipLaunchKernelGGL((k_segf<L, PV0, 2>), dim3(grid), dim3(SegCfg<L>::BLOCK), lds, c->stream, a);
                    done = true;
                }
            }
            if constexpr (segf_has_sumq<L, PV0>()) {
                if (a.Qs && !done) {         // SUM_ALL: the Parseval half of the complement form once per bin
                    hipLaunchKernelGGL((k_segf<L, PV0, 1>), dim3(grid), dim3(SegCfg<L>::BLOCK), lds, c->stream, a);
                    done = true;
                }
            }
            if (!done) hipLaunchKernelGGL((k_segf<L, PV0, 0>), dim3(grid), dim3(SegCfg<L>::BLOCK), lds, c->stream, a);
            HIPCHK(hipGetLastError());
            return MFB_OK;
        }
        return launch_segf_pv<L, PV0 + 1, PV1>(c, a, grid, lds, pv);
    }
}
// the complete slots of nb blocks through k_segf; the caller has the partial sums reserved
static int launch_fsm(mfb_ctx *c, int nb, const cf *x, int xstride, int MU, int nfull, int parts, int pv) {
    SegFArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.Gs = c->d_Gs;
    a.Qs = c->sum_all ? c->d_Qs : nullptr;
    a.presum = c->gs_span == 1 || std::all_of(c->h_mult.begin(), c->h_mult.end(), [&](int m) { return m == c->h_mult[0]; }) ? 1 : 0;
    a.twL = c->d_twL;
    a.partials = c->d_part;
    a.N = c->N;
    a.V = c->V;
    a.slot0 = 0;
    a.nslots = nfull;
    a.MU = MU;
    a.dper = c->Dtot;
    a.nblk = nb;
    a.xstride = xstride;
    const FsmPlan fp = fsm_plan(c, MU, nfull, nb);
    if (!fp.ok) return MFB_ERR_UNSUPPORTED;
    a.nsg = fp.nsg;
    a.gbins = fp.gbins;
    a.fb = fp.fb;
    a.fs = fp.fs;
    a.nbc = fp.nbc;
    a.nsc = fp.nsc;
    a.part_row0 = 0;
    a.parts = parts;
    a.scale = 1.0f / 262144.0f;
    const bool w32 = c->segl == 11;
    const long long waves = (long long)nb * a.nbc * a.nsc;
    const int wpb = SegCfg<256>::BLOCK / 64;
    static_assert(SegCfg<256>::BLOCK == SegCfg<2048>::BLOCK || !MFB_SEG_W32, "four independent waves per workgroup");
    const int grid = a.nsg * (int)((waves + wpb - 1) / wpb);
    if (w32) {
        const size_t lds = SegCfg<2048>::LDS_ELEMS * sizeof(cf) + (size_t)wpb * MU * SEG_ACC_STRIDE * sizeof(float);
        return launch_segf_pv<2048, 16, 32>(c, a, grid, lds, pv);
    }
    const size_t lds = SegCfg<256>::LDS_ELEMS * sizeof(cf) + (size_t)wpb * MU * SEG_ACC_STRIDE * sizeof(float);
    return launch_segf_pv<256, 8, 16>(c, a, grid, lds, pv);
}

// Build the per-bin spectra as soon as filters, shifts and the path are known (mfb_set_filters / mfb_set_shifts / mfb_set_search_*),
// not inside the first search: 512 transforms of 2048 points for the 384-tap bank at 64 bins are 10-20 ms of host work, which does
// not belong into a receive loop's first block.  (The search still checks, and builds what is missing.)
static int fsm_prepare_eager(mfb_ctx *c) {
    if (!c->have_filters || !c->have_shifts || c->path != MFB_PATH_SEGMENT || !fsm_enabled()) return MFB_OK;
    const bool span = c->basis == MFB_BASIS_SPAN;
    const int MU = span ? c->MB : c->MU;
    int nfull, ntotal;
    seg_slots(c, &nfull, &ntotal);
    const bool fsm_l = c->segl == 8 || (c->segl == 11 && MFB_SEG_W32 && MU <= 8 && (size_t)c->Dtot * MU * 16384 <= ((size_t)256 << 20));
    if (!(fsm_l && MU <= SEG_MPB_MAX && nfull > 0 && (int)c->h_shifts.size() == c->Dtot && fsm_plan(c, MU, nfull).ok)) return MFB_OK;
    return fsm_prepare(c, span, MU);
}
// how the segment search of the next block will run: filter_side = 1 when the Doppler shift sits on the filters' side (k_segf: one
// forward transform per segment for `bins_per_forward` bins), 0 when every (bin, segment) is mixed and transformed (k_seg)
extern "C" int mfb_get_search_info(mfb_ctx *c, int *filter_side, int *bins_per_forward) {
    if (!c || !filter_side || !bins_per_forward) return MFB_ERR_ARG;
    *filter_side = 0;
    *bins_per_forward = 1;
    if (c->path != MFB_PATH_SEGMENT || !c->have_filters) return MFB_OK;
    const bool span = c->basis == MFB_BASIS_SPAN;
    const int MU = span ? c->MB : c->MU;
    int nfull, ntotal;
    seg_slots(c, &nfull, &ntotal);
    const bool fsm_l = c->segl == 8 || (c->segl == 11 && MFB_SEG_W32 && MU <= 8 && (size_t)c->Dtot * MU * 16384 <= ((size_t)256 << 20));
    if (fsm_l && fsm_enabled() && MU <= SEG_MPB_MAX && nfull > 0 && (int)c->h_shifts.size() == c->Dtot) {
        const FsmPlan p = fsm_plan(c, MU, nfull);
        if (p.ok) {
            *filter_side = 1;
            *bins_per_forward = p.fb;
        }
    }
    return MFB_OK;
}

// The segment-path search of nb blocks in ONE launch (+ the masked tail launch + k_finalize): stream jl of the launch is bin
// jl % Dtot of block jl / Dtot, block b's samples start at x + b * xstride, its doppSum rows at dsum + b * Dtot * M.  nb = 1 is
// the plain search of one block.  A score depends on the block, the shift and the filter only (seg_kernels.hpp), so a
// block's rows come out bit for bit the same whichever batch it is part of.
static int seg_search_enqueue(mfb_ctx *c, int nb, const cf *x, int xstride, float *dsum) {
    // all bins in ONE launch: nothing of length N is written, so there is nothing to chunk.  A second,
    // tiny launch of the masked instantiation covers the slots that hold incomplete segments.
    const bool span = c->basis == MFB_BASIS_SPAN;
    const int MU = span ? c->MB : c->MU;     // filters transformed per bin
    const int rows = nb * c->Dtot;           // Doppler streams of the launch
    int nfull, ntotal;
    seg_slots(c, &nfull, &ntotal);
    const SegPlan pm = plan_seg(c, rows, MU, nfull > 0 ? nfull : 1, c->seg_mpb, true);
    // the tail is a handful of units: spread the filters too (2 per pass) so that it is short
    const SegPlan pt = plan_seg(c, rows, MU, ntotal - nfull > 0 ? ntotal - nfull : 1, c->seg_mpb > 0 && c->seg_mpb < 2 ? c->seg_mpb : 2);
    // one partial per (bin, filter, slot, wave of the team): the index depends on the slot alone (seg_kernels.hpp)
    const int parts = ntotal * seg_geom(c).WPT;
    int rc = reserve_partials(c, (size_t)rows * MU * parts);
    if (rc) return rc;
    // 256-point segments: the shift on the filter side -- one forward transform per segment for all bins (segf_body)
    // (the wave-local 2048-point kernel too: 8 filters per pass there, and per-bin spectra of 16 KiB per (bin, filter) -- up to 256 MiB)
    const bool fsm_l = c->segl == 8 || (c->segl == 11 && MFB_SEG_W32 && MU <= 8 && (size_t)c->Dtot * MU * 16384 <= ((size_t)256 << 20));
    const bool fsm = fsm_l && fsm_enabled() && MU <= SEG_MPB_MAX && nfull > 0 && (int)c->h_shifts.size() == c->Dtot && fsm_plan(c, MU, nfull).ok;
    if (fsm && (rc = fsm_prepare(c, span, MU))) return rc;       // (built by the first block: nothing is allocated inside a graph capture)
    SegArgs am = seg_base(c, pm), at = seg_base(c, pt);
    for (SegArgs *q : {&am, &at}) {
        if (span) {
            q->G = c->d_Gb;
            q->Grows = c->MB;
        }
        q->x = x;
        q->rows = (!span && MU < c->M) ? c->d_uniq : nullptr;
        q->shifts = c->d_shifts;
        q->partials = c->d_part;
        q->MU = MU;
        q->dc = rows;
        q->parts = parts;
        if (nb > 1) {
            q->dper = c->Dtot;
            q->xstride = xstride;
        }
    }
    am.slot0 = 0;
    am.nslots = nfull;
    at.slot0 = nfull;
    at.nslots = ntotal - nfull;
    const int pv = c->V / seg_geom(c).NT;
    prof_mark(c, 0);
    // (both roles inside one grid were tried: the merged kernel ran 8-10 % slower than two launches)
    if (fsm) rc = launch_fsm(c, nb, x, xstride, MU, nfull, parts, pv);
    else if (nfull > 0) rc = launch_seg(c, am, pm.grid, SEG_REDUCE, pv);
    if (!rc && ntotal > nfull) rc = launch_seg(c, at, pt.grid, SEG_REDUCE, -1);
    if (rc) return rc;
    prof_mark(c, 0);
    hipLaunchKernelGGL(k_finalize, dim3(rows), dim3(fin_threads(MU)), 0, c->stream, c->d_part, dsum, rows, c->M, MU,
                       span ? (const int *)nullptr : (const int *)c->d_rep, parts, c->sum_all);
    HIPCHK(hipGetLastError());
    return MFB_OK;
}

extern "C" int mfb_search_async(mfb_ctx *c) {
    if (!c) return MFB_ERR_ARG;
    if (!c->have_filters || !c->have_shifts || !c->have_input) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const int MU = c->MU;
    if (c->search_mode == MFB_SEARCH_ENERGY) {
        // opt-in shortcut (energy_kernels.hpp): no transform, D.N multiply-adds on |X|^2 and the filters' energy
        const int R = c->sum_all ? 1 : c->M;
        const int parts = c->N / EN_CHUNK;
        int rc = reserve_partials(c, (size_t)c->Dtot * R * parts);
        if (rc) return rc;
        if (!c->d_pow) HIPCHK(dev_alloc((void **)&c->d_pow, (size_t)c->N * sizeof(float)));
        if (!c->d_W) HIPCHK(dev_alloc((void **)&c->d_W, (size_t)c->N * (c->sum_all ? 1 : c->M) * sizeof(float)));
        if (!c->W_valid) {
            hipLaunchKernelGGL(k_filter_energy, dim3(c->N / 256, R), dim3(256), 0, c->stream, (const cf *)c->d_masks, c->d_W, c->N, c->M,
                               c->sum_all);
            HIPCHK(hipGetLastError());
            c->W_valid = true;
        }
        prof_mark(c, 0);
        hipLaunchKernelGGL(k_power, dim3(c->N / 256), dim3(256), 0, c->stream, (const cf *)c->d_X, c->d_pow, c->N);
        hipLaunchKernelGGL(k_energy, dim3(parts, R), dim3(256), 0, c->stream, (const float *)c->d_pow, (const float *)c->d_W,
                           (const int *)c->d_shifts, c->d_part, c->N, c->Dtot, R, parts, (float)c->N / 262144.f);
        HIPCHK(hipGetLastError());
        prof_mark(c, 0);
        hipLaunchKernelGGL(k_finalize, dim3(c->Dtot), dim3(fin_threads(R)), 0, c->stream, c->d_part, c->d_sum, c->Dtot, c->M, R, (const int *)nullptr,
                           parts, c->sum_all);
        HIPCHK(hipGetLastError());
        return MFB_OK;
    }
    if (c->path == MFB_PATH_SEGMENT) return seg_search_enqueue(c, 1, c->d_in, 0, c->d_sum);
    const int mpb = c->mpb < MU ? c->mpb : MU;
    const int mgroups = (MU + mpb - 1) / mpb;
    for (int j0 = 0; j0 < c->Dtot; j0 += c->chunk) {
        const int dc = (c->Dtot - j0) < c->chunk ? (c->Dtot - j0) : c->chunk;
        P1Args a = p1_base(c);
        a.X = c->d_X;
        a.shifts = c->d_shifts;
        a.j0 = j0;
        a.dc = dc;
        a.M = MU;
        a.mpb = mpb;
        a.rows = (MU < c->M) ? c->d_uniq : nullptr;
        const int js = c->jsplit < dc ? c->jsplit : dc;
        a.ntiles = c->N2 / TILE;
        a.mgroups = mgroups;
        a.jsplit = js;
        prof_mark(c, 0);
        int rc = launch_p1<KIND_BANK>(c, a, dim3(a.ntiles * mgroups * js, 1, 1));
        prof_mark(c, 0);
        if (rc) return rc;
        P2Args b = p2_base(c);
        b.partials = c->d_part;
        b.part_row0 = j0 * MU;
        prof_mark(c, 1);
        rc = launch_p2<MODE_REDUCE>(c, b, dim3(c->parts, dc * MU, 1));
        prof_mark(c, 1);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_finalize, dim3(c->Dtot), dim3(fin_threads(MU)), 0, c->stream, c->d_part, c->d_sum, c->Dtot, c->M, MU,
                       (const int *)c->d_rep, c->parts, c->sum_all);
    HIPCHK(hipGetLastError());
    return MFB_OK;
}

// device -> host through the page-locked landing zone when it fits (n pieces back to back, ONE synchronisation)
struct BackPiece {
    void *host;
    const void *dev;
    size_t bytes;
};
static int read_back(mfb_ctx *c, const BackPiece *p, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (p[i].bytes + 15) & ~(size_t)15;
    const bool staged = c->h_back && total <= c->back_cap;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (!p[i].bytes) continue;
        HIPCHK(hipMemcpyAsync(staged ? (void *)(c->h_back + off) : p[i].host, p[i].dev, p[i].bytes, hipMemcpyDeviceToHost, c->stream));
        off += (p[i].bytes + 15) & ~(size_t)15;
    }
    HIPCHK(sync_streams(c));
    if (staged) {
        off = 0;
        for (int i = 0; i < n; ++i) {
            if (p[i].bytes) memcpy(p[i].host, c->h_back + off, p[i].bytes);
            off += (p[i].bytes + 15) & ~(size_t)15;
        }
    }
    return MFB_OK;
}

extern "C" int mfb_export_scores_async(mfb_ctx *c, void *dst, int row_offset) {
    if (!c || !dst || row_offset < 0) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync((float *)dst + (size_t)row_offset * c->M, c->d_sum, (size_t)c->Dtot * c->M * sizeof(float),
                          hipMemcpyDeviceToDevice, c->stream));
    return MFB_OK;
}

extern "C" int mfb_export_column_async(mfb_ctx *c, void *dst, int row_offset) {
    if (!c || !dst || row_offset < 0) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    // column 0 of doppSum [Dtot][M] -> dst[row_offset + j]
    HIPCHK(hipMemcpy2DAsync((float *)dst + row_offset, sizeof(float), c->d_sum, (size_t)c->M * sizeof(float), sizeof(float),
                            (size_t)c->Dtot, hipMemcpyDeviceToDevice, c->stream));
    return MFB_OK;
}

extern "C" int mfb_export_rows_async(mfb_ctx *c, void *dst, int dst_row, int first_row, int nrows, int column_only) {
    if (!c || !dst || dst_row < 0 || first_row < 0 || nrows < 0 || first_row + nrows > c->Dtot) return MFB_ERR_ARG;
    if (!nrows) return MFB_OK;
    HIPCHK(hipSetDevice(c->device));
    const float *src = c->d_sum + (size_t)first_row * c->M;
    if (column_only)
        HIPCHK(hipMemcpy2DAsync((float *)dst + dst_row, sizeof(float), src, (size_t)c->M * sizeof(float), sizeof(float), (size_t)nrows,
                                hipMemcpyDeviceToDevice, c->stream));
    else
        HIPCHK(hipMemcpyAsync((float *)dst + (size_t)dst_row * c->M, src, (size_t)nrows * c->M * sizeof(float), hipMemcpyDeviceToDevice,
                              c->stream));
    return MFB_OK;
}

extern "C" int mfb_pick_column(mfb_ctx *c, const void *column, int num, int offset, float res[2]) {
    if (!c || !column || !res || num < 1 || offset < 0) return MFB_ERR_ARG;
    if (!c->sum_all) return MFB_ERR_STATE;   // a single column is the whole table only under SUM_ALL_MASKS
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, c->stream, (const float *)column, c->d_res, num, offset, 1, 1);
    HIPCHK(hipGetLastError());
    const BackPiece bp = {res, c->d_res, 2 * sizeof(float)};
    return read_back(c, &bp, 1);
}

extern "C" int mfb_pick(mfb_ctx *c, const void *scores, int num, int offset, float res[2]) {
    if (!c || !res || num < 1 || offset < 0) return MFB_ERR_ARG;
    if (!scores && (num != c->D || offset != c->Doff)) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const float *in = scores ? (const float *)scores : c->d_sum;
    hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, c->stream, in, c->d_res, num, offset, c->M, c->sum_all);
    HIPCHK(hipGetLastError());
    const BackPiece bp = {res, c->d_res, 2 * sizeof(float)};
    return read_back(c, &bp, 1);
}

extern "C" int mfb_find_carrier(mfb_ctx *c, float res[2]) {
    int rc = mfb_search_async(c);
    if (rc) return rc;
    return mfb_pick(c, nullptr, c->D, c->Doff, res);
}

extern "C" int mfb_get_scores(mfb_ctx *c, float *host) {
    if (!c || !host) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(host, c->d_sum, (size_t)c->Dtot * c->M * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_streams(c));
    return MFB_OK;
}

// doppSum of block `block` of the LAST batch (mfb_receive_blocks_*), as mfb_get_scores gives the one-block table
extern "C" int mfb_get_batch_scores(mfb_ctx *c, int block, float *host) {
    if (!c || !host || block < 0) return MFB_ERR_ARG;
    if (!c->d_sumb || block >= c->bat_cap || block >= c->last_batch_blocks) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));
    HIPCHK(hipMemcpy(host, c->d_sumb + (size_t)block * c->Dtot * c->M, (size_t)c->Dtot * c->M * sizeof(float), hipMemcpyDeviceToHost));
    return MFB_OK;
}

extern "C" int mfb_get_spectrum(mfb_ctx *c, float *host, int start, int count) {
    if (!c || !host || count < 0 || count > c->N) return MFB_ERR_ARG;
    if (!c->have_input) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    start = ((start % c->N) + c->N) % c->N;
    const int first = (start + count <= c->N) ? count : c->N - start;
    if (c->mirror_valid) {            // the spectrum is (or is about to be) on the host already
        HIPCHK(hipEventSynchronize(c->ev_X));
        memcpy(host, c->h_X + start, (size_t)first * sizeof(cf));
        if (first < count) memcpy(host + 2 * (size_t)first, c->h_X, (size_t)(count - first) * sizeof(cf));
        return MFB_OK;
    }
    const BackPiece bp[2] = {{host, c->d_X + start, (size_t)first * sizeof(cf)},
                             {host + 2 * (size_t)first, c->d_X, (size_t)(count - first) * sizeof(cf)}};
    const int rc = read_back(c, bp, 2);
    if (rc) return rc;
    return enable_mirror(c);          // a caller that reads the spectrum once will read it again after every block
}

// Where a block -- or a batch of nb blocks -- lives on the device while it goes through the block path.
struct BlkBufs {
    int nb;              // blocks
    const cf *x;         // samples: block b starts at x + b * xstride
    int xstride;
    cf *X;               // spectra [nb][N]
    float *sum;          // doppSum [nb][Dtot][M]
    float *res;          // picks [nb][2]
    cf *xc;              // matched-filter outputs [nb][M][N]
    float *env;          // symbol-energy envelopes [nb][N]
    cf *P;               // their spectra [nb][N]
    float *cr;           // rate triples [nb][3]
    uint8_t *out;        // result records [nb][rec]
    size_t rec;
    size_t ext;          // offset of the stream-stage outputs inside a record (0: the stages run on the host)
    int parity;          // carry buffer the stream stages read
};
static BlkBufs single_bufs(mfb_ctx *c) {
    return BlkBufs{1, c->d_in, 0, c->d_X, c->d_sum, c->d_res, c->d_xc, c->d_env, c->d_P, c->d_cr, c->d_blkout, 0, 0, 0};
}

// A9 + A10 enqueued on the handle's stream: matched filters at one shift per block (a value, or -- shift_dev != nullptr -- an
// int the device has just computed; for a batch, block b's sits shift_stride ints behind block b - 1's), envelope, its
// spectrum, windowed argmax into cr.  No synchronisation.
static int demod_enqueue(mfb_ctx *c, const BlkBufs &bb, int shift, const int *shift_dev, int shift_stride, int k_offset, int k_len,
                         int spsym_min = 0, int capacity = 0, BlockScalars *scal = nullptr) {
    // A9: matched filters at one shift -> xc[M][N] natural order
    int rc;
    const int nb = bb.nb;
    if (c->path == MFB_PATH_SEGMENT) {
        // the same segment kernel, storing y in natural order instead of reducing it (no transpose)
        int nfull, ntotal;
        seg_slots(c, &nfull, &ntotal);
        // one bin: spread the slots over the chip.  When a team takes all M filters of its segments (M <= 16) it also
        // sums the symbol-energy envelope (sumXCorrBuffMasks, CU:191-205) in the same pass; otherwise the filters are
        // spread too and k_envelope follows
        const bool fused_env = c->M <= SEG_MPB_MAX;
        const SegPlan p = plan_seg(c, nb, c->M, ntotal, fused_env ? c->M : 4);
        SegArgs sa = seg_base(c, p);
        sa.x = bb.x;
        sa.out = bb.xc;
        if (fused_env) {
            sa.env = bb.env;
            sa.env_lo = c->cs_off;
            sa.env_hi = c->M - c->cs_off;
        }
        sa.MU = c->M;
        sa.dc = nb;
        sa.slot0 = 0;
        sa.nslots = ntotal;
        sa.shifts = shift_dev;
        sa.j0 = 0;
        sa.fixed_shift = shift;
        sa.out_off = (c->win_start + c->T - 1) & (c->N - 1);
        if (nb > 1) {
            sa.dper = 1;
            sa.xstride = bb.xstride;
            sa.sstride = shift_stride;
            sa.ostride = (long long)c->M * c->N;
        }
        rc = launch_seg(c, sa, p.grid, SEG_STORE, -1);
        if (rc) return rc;
    } else {
        if (nb != 1) return MFB_ERR_UNSUPPORTED;        // batches of blocks run on the segment path
        P1Args a = p1_base(c);
        a.X = bb.X;
        a.shifts = shift_dev;
        a.j0 = 0;
        a.fixed_shift = shift;
        a.dc = 1;
        const int mgroups = (c->M + c->mpb - 1) / c->mpb;
        a.ntiles = c->N2 / TILE;
        a.mgroups = mgroups;
        a.jsplit = 1;
        rc = launch_p1<KIND_BANK>(c, a, dim3(a.ntiles * mgroups, 1, 1));
        if (rc) return rc;
        P2Args b = p2_base(c);
        p2_store_split(c, b, c->M);
        rc = launch_p2<MODE_STORE>(c, b, dim3(b.parts, c->M, 1));
        if (rc) return rc;
        rc = launch_transpose(c, bb.xc, c->M, 0);
        if (rc) return rc;
    }
    // A10: envelope (unless the matched-filter launch already summed it), spectrum of the envelope, windowed argmax
    if (!(c->path == MFB_PATH_SEGMENT && c->M <= SEG_MPB_MAX)) {
        hipLaunchKernelGGL(k_envelope, dim3(1024, nb), dim3(256), 0, c->stream, (const cf *)bb.xc, bb.env, c->N, c->M, c->cs_off);
        HIPCHK(hipGetLastError());
    }
    rc = forward_fft(c, nullptr, bb.env, bb.P, 1, nb, (size_t)c->N);
    if (rc) return rc;
    if (scal)
        hipLaunchKernelGGL(k_code_rate_block, dim3(nb), dim3(1024), 0, c->stream, (const cf *)bb.P, bb.cr, k_offset, k_len, c->N, spsym_min,
                           capacity, scal, bb.rec);
    else
        hipLaunchKernelGGL(k_code_rate, dim3(1), dim3(1024), 0, c->stream, (const cf *)bb.P, bb.cr, k_offset, k_len);
    HIPCHK(hipGetLastError());
    return MFB_OK;
}

extern "C" int mfb_demodulate(mfb_ctx *c, int shift, int k_offset, int k_len, float res[3]) {
    if (!c || !res) return MFB_ERR_ARG;
    if (!c->have_filters || !c->have_input) return MFB_ERR_STATE;
    if (k_offset < 0 || k_len < 0 || k_offset + k_len > c->N) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    shift = ((shift % c->N) + c->N) % c->N;
    int rc = demod_enqueue(c, single_bufs(c), shift, nullptr, 0, k_offset, k_len);
    if (rc) return rc;
    const BackPiece bp = {res, c->d_cr, 3 * sizeof(float)};
    if ((rc = read_back(c, &bp, 1))) return rc;
    c->have_xc = true;
    return MFB_OK;
}

// ---- one call per block --------------------------------------------------------------------------------------------
// Everything the receive loop does with one block on the device (reference demodulator_base.py:548-632 and 711-1009), as ONE
// stream of launches and ONE synchronisation: [H2D of the pinned input buffer,] forward FFT, Doppler search, pick, shift
// interpolation (k_block_pick), the two spectrum windows of computeSNR, matched filters at that shift, envelope, its
// spectrum, rate/phase argmax, the float64 arithmetic behind it (k_block_rate), symbol centres -- then one packed read-back.
// The same launches take a BATCH of bb.nb consecutive blocks of one window of samples (mfb_receive_blocks_*): every kernel
// gets a block index in its grid, nothing else changes -- a block's numbers are those of the one-block call, bit for bit.
// The launches of one block / batch, from the forward FFT to the ONE device-to-host copy of the result record(s).
// A14 for the blocks of a batch: one packed launch (search + would-be stash edges + ring) where the templates allow it
static void stream_search_launch(mfb_ctx *c, const StreamArgs &sa, int nb, int Tmax) {
    const size_t words = (size_t)((STREAM_PACK_TAPS + sa.T[0] - 1 + STREAM_EDGE_BACK + sa.nOv + sa.nsym + 31) >> 5) + STREAM_PACK_TAPS / 32 + 4;
    static const bool unpacked = getenv("MFB_STREAM_UNPACKED") != nullptr;       // (A/B switch of tools/chain_kernels.sh)
    if (sa.packed && words * 4 <= 60 * 1024 && !unpacked) {
        hipLaunchKernelGGL(k_stream_search, dim3(nb), dim3(STREAM_SEARCH_THREADS), words * 4, c->stream, sa);
        return;
    }
    hipLaunchKernelGGL(k_stream_sync, dim3(sa.K, nb), dim3(256), (size_t)Tmax + SYNC_SEG + Tmax - 1, c->stream, sa);
    hipLaunchKernelGGL(k_stream_ring, dim3(1), dim3(256), 0, c->stream, sa);
    hipLaunchKernelGGL(k_stream_edges, dim3(nb), dim3(256), 0, c->stream, sa);
}
// where the pieces of a result record sit
struct RecPtrs {
    size_t rec;
    uint8_t *d;
    BlockScalars *scal;
    cf *bands;
    int *sym, *cen;
    float *mag;
};
static RecPtrs rec_ptrs(const BlkBufs &bb, int nthreads, int bcap) {
    RecPtrs r;
    r.rec = bb.rec ? bb.rec : blkout_bytes(bcap, nthreads);
    r.d = bb.out;
    r.scal = (BlockScalars *)r.d;
    r.bands = (cf *)(r.d + BLK_HEAD);
    r.sym = (int *)(r.d + BLK_HEAD + align16((size_t)2 * bcap * sizeof(cf)));
    r.cen = (int *)((uint8_t *)r.sym + align16((size_t)nthreads * sizeof(int)));
    r.mag = (float *)((uint8_t *)r.cen + align16((size_t)nthreads * sizeof(int)));
    return r;
}
// Part 1 of a block / batch: forward transform(s), Doppler search, pick (or, at a fixed shift, the cleared scalars).
static int block_enqueue_p1(mfb_ctx *c, const mfb_block_params *p, const BlkBufs &bb, int nthreads, int bcap) {
    int rc;
    const int nb = bb.nb;
    const bool batch = bb.rec != 0;
    if (!batch && p->input == MFB_INPUT_UPLOADED) {
        if (!c->have_input) return MFB_ERR_STATE;
    } else {
        if (!batch) {
            if ((rc = before_fft(c))) return rc;
        }
        if ((rc = forward_fft(c, bb.x, nullptr, bb.X, 1, nb, (size_t)bb.xstride))) return rc;
        if (!batch) {
            if ((rc = after_fft(c))) return rc;
            c->have_input = true;
            c->have_xc = false;
        }
    }
    const RecPtrs r = rec_ptrs(bb, nthreads, bcap);
    if (p->mode == MFB_BLOCK_SEARCH) {
        if (batch) {
            if (c->path != MFB_PATH_SEGMENT || c->search_mode != MFB_SEARCH_TRANSFORMS) return MFB_ERR_UNSUPPORTED;
            if ((rc = seg_search_enqueue(c, nb, bb.x, bb.xstride, bb.sum))) return rc;
        } else if ((rc = mfb_search_async(c))) {
            return rc;
        }
        hipLaunchKernelGGL(k_pick_block, dim3(nb), dim3(64), 0, c->stream, (const float *)bb.sum, bb.res, c->D, c->Doff, c->M, c->sum_all,
                           (const int *)c->d_shifts, c->Dtot, c->N, p->snr_window, (const cf *)bb.X, r.scal, r.bands, bcap, r.rec);
        HIPCHK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_block_clear, dim3(nb), dim3(1), 0, c->stream, r.scal, r.rec);
        HIPCHK(hipGetLastError());
    }
    return MFB_OK;
}
// Part 2: matched filters at the picked (or fixed) shift, envelope, its spectrum, rate / phase, symbol centres, the integer stages
// of a batch, and the ONE device-to-host copy of the record(s).
static int block_enqueue_p2(mfb_ctx *c, const mfb_block_params *p, const BlkBufs &bb, uint8_t *h_dst, int nthreads, int bcap, int capacity,
                            int *shift_out) {
    int rc;
    const int nb = bb.nb;
    const bool batch = bb.rec != 0;
    const RecPtrs r = rec_ptrs(bb, nthreads, bcap);
    const size_t rec = r.rec;
    uint8_t *d = r.d;
    BlockScalars *scal = r.scal;
    int *d_sym = r.sym, *d_cen = r.cen;
    float *d_mag = r.mag;
    const int *shift_dev = nullptr;
    int shift = 0;
    if (p->mode == MFB_BLOCK_SEARCH) shift_dev = &scal->shift;
    else shift = ((p->fixed_shift % c->N) + c->N) % c->N;
    if ((rc = demod_enqueue(c, bb, shift, shift_dev, (int)(rec / sizeof(int)), p->k_offset, p->k_len, p->spsym_min, capacity, scal))) return rc;
    // (entries past nthreads are not part of the record: the kernel's capacity bounds what it writes; the k* == 0 fallback --
    // spSym = 10, DB:737-740, unreachable while the rate window starts above bin 0 -- is completed in block_end)
    hipLaunchKernelGGL(k_centres_block, dim3((nthreads + 255) / 256, nb), dim3(256), 0, c->stream, d_sym, d_cen, d_mag, (const cf *)bb.xc,
                       (const BlockScalars *)scal, c->N, c->M, c->W, p->op, nthreads, rec);
    HIPCHK(hipGetLastError());
    if (batch && bb.ext) {
        // A12 / A13 / A14 of every block of the batch on the device (stream_kernels.hpp); the state they chain on -- the previous
        // batch's tail and bit ring -- sits in carry[cur], this batch leaves its own in carry[1 - cur]
        StreamArgs sa = c->st;
        sa.rec0 = d;
        sa.rec = rec;
        sa.off_sym = (size_t)((uint8_t *)d_sym - d);
        sa.off_cen = (size_t)((uint8_t *)d_cen - d);
        sa.off_mag = (size_t)((uint8_t *)d_mag - d);
        const size_t a1 = align16((size_t)nthreads);
        sa.off_bits = bb.ext;
        sa.off_cenw = bb.ext + a1;
        sa.off_trust = bb.ext + 2 * a1;
        sa.off_post = bb.ext + 3 * a1;
        sa.off_end = sa.off_post + STREAM_POST_MAX;
        sa.off_hits = sa.off_end + STREAM_END_MAX;
        sa.off_edges = sa.off_hits + (size_t)STREAM_MAX_TMPL * 2 * STREAM_MAX_HITS * sizeof(int32_t);
        sa.nb = nb;
        sa.nsym = nthreads;
        sa.carry_in = c->d_carry[bb.parity];
        sa.carry_out = c->d_carry[1 - bb.parity];
        hipLaunchKernelGGL(k_stream_align, dim3(nb), dim3(STREAM_ALIGN_THREADS), 0, c->stream, sa);
        HIPCHK(hipGetLastError());
        if (sa.K > 0) {
            int Tmax = 0;
            for (int t = 0; t < sa.K; ++t) Tmax = sa.T[t] > Tmax ? sa.T[t] : Tmax;
            stream_search_launch(c, sa, nb, Tmax);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipMemcpyAsync(h_dst, d, rec * nb, hipMemcpyDeviceToHost, c->stream));
    *shift_out = shift;
    return MFB_OK;
}
static int block_enqueue(mfb_ctx *c, const mfb_block_params *p, const BlkBufs &bb, uint8_t *h_dst, int nthreads, int bcap, int capacity,
                         int *shift_out) {
    const int rc = block_enqueue_p1(c, p, bb, nthreads, bcap);
    return rc ? rc : block_enqueue_p2(c, p, bb, h_dst, nthreads, bcap, capacity, shift_out);
}

// Host-to-device copy of a page-locked input buffer into its own device copy, on the input stream: it starts as soon as the
// last block that read that device copy is done -- i.e. while the block before this one is still being searched -- and the
// handle's stream picks it up with an event.  (With one device buffer and one stream the copy of an 8 MiB block, 0.15 ms,
// sat between two searches.)
static int input_copy(mfb_ctx *c, const cf *src, cf *dst, size_t samples, hipEvent_t *ev_h2d, hipEvent_t *ev_free, int which) {
    if (!src || !dst) return MFB_ERR_STATE;
    if (!c->in_stream) HIPCHK(hipStreamCreateWithFlags(&c->in_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!ev_h2d[i]) HIPCHK(hipEventCreateWithFlags(&ev_h2d[i], hipEventDisableTiming));
        if (!ev_free[i]) {
            HIPCHK(hipEventCreateWithFlags(&ev_free[i], hipEventDisableTiming));
            HIPCHK(hipEventRecord(ev_free[i], c->stream));
        }
    }
    HIPCHK(hipStreamWaitEvent(c->in_stream, ev_free[which], 0));
    HIPCHK(hipMemcpyAsync(dst, src, samples * sizeof(cf), hipMemcpyHostToDevice, c->in_stream));
    HIPCHK(hipEventRecord(ev_h2d[which], c->in_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, ev_h2d[which], 0));
    return MFB_OK;
}
static int block_input_copy(mfb_ctx *c, int which) {
    return input_copy(c, which ? c->h_in2 : c->h_in, which ? c->d_x2 : c->d_x, (size_t)c->N, c->ev_h2d, c->ev_xfree, which);
}

// A block whose input is one of the handle's two page-locked buffers is the same sequence of launches with the same
// arguments every time (the shift the matched filters run at is read from device memory): the second such block is captured
// into a HIP graph, later ones replay it -- one call into the runtime per block instead of a dozen launches, which is what the
// host side of a stream of small blocks was spending its time on.  Any change of filters, shifts, search settings or stream
// drops the graphs (c->epoch).  MFB_NO_GRAPH=1 in the environment keeps every block on plain launches.
static bool graphs_allowed() {
    static const int off = getenv("MFB_NO_GRAPH") ? atoi(getenv("MFB_NO_GRAPH")) : 0;
    return !off;
}
static void graph_drop(BlockGraph &g) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g.exec = nullptr;
    g.graph = nullptr;
    g.seen = false;
}
static bool same_params(const mfb_block_params &a, const mfb_block_params &b) {
    return a.mode == b.mode && a.input == b.input && a.fixed_shift == b.fixed_shift && a.k_offset == b.k_offset && a.k_len == b.k_len &&
           a.spsym_min == b.spsym_min && a.op == b.op && a.snr_window == b.snr_window && a.max_symbols == b.max_symbols &&
           a.band_capacity == b.band_capacity && a.block_stride == b.block_stride;
}

// Launch the block's (batch's) work: replay its graph, record it (second occurrence of these settings), or plain launches.
template <class Enqueue>
static int graph_or_launch(mfb_ctx *c, BlockGraph &g, const mfb_block_params *p, int nb, bool allowed, Enqueue &&enqueue) {
    int rc;
    if (allowed) {
        if (g.epoch != c->epoch || g.nb != nb || !same_params(g.params, *p)) {
            graph_drop(g);
            g.epoch = c->epoch;
            g.params = *p;
            g.nb = nb;
            g.failed = false;
        }
        if (g.exec) {
            HIPCHK(hipGraphLaunch(g.exec, c->stream));
            return MFB_OK;
        }
        if (g.seen) {
            // second block with these settings: record it (the first one went out as plain launches, so every workspace
            // it needs exists already -- nothing is allocated inside the capture)
            HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
            rc = enqueue();
            hipGraph_t graph = nullptr;
            const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
            if (rc || ce != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                if (rc) return rc;
                g.seen = false;                       // capture not possible here: stay on plain launches
                g.failed = true;
            } else {
                hipGraphExec_t exec = nullptr;
                if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec) {
                    g.graph = graph;
                    g.exec = exec;
                    HIPCHK(hipGraphLaunch(g.exec, c->stream));
                    return MFB_OK;
                }
                (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                g.seen = false;
                g.failed = true;
            }
        } else if (!g.failed) {
            g.seen = true;
        }
    }
    return enqueue();
}

static int check_block_params(const mfb_ctx *c, const mfb_block_params *p) {
    if (!c->have_filters || (p->mode == MFB_BLOCK_SEARCH && !c->have_shifts)) return MFB_ERR_STATE;
    if (p->mode != MFB_BLOCK_SEARCH && p->mode != MFB_BLOCK_FIXED_SHIFT) return MFB_ERR_ARG;
    if (p->k_offset < 0 || p->k_len < 0 || p->k_offset + p->k_len > c->N || p->spsym_min < 2 || p->op < 0 || p->op > 2 ||
        p->max_symbols < 1 || p->snr_window < 0 || p->band_capacity < 0)
        return MFB_ERR_ARG;
    return MFB_OK;
}
// page-locked staging of a flight: at least `need` bytes (the address is baked into the slot's graphs)
static int staging_reserve(mfb_ctx *c, int slot, size_t need) {
    if (need <= c->blk_cap[slot]) return MFB_OK;
    for (auto &row : c->bgraph) graph_drop(row[slot]);
    for (auto &row : c->bgraph2) graph_drop(row[slot]);
    for (auto &row : c->wgraph)
        for (auto &par : row[slot])
            for (auto &g : par) graph_drop(g);
    for (auto &row : c->wgraph2)
        for (auto &par : row[slot])
            for (auto &g : par) graph_drop(g);
    if (c->s2) HIPCHK(hipStreamSynchronize(c->s2));
    if (c->h_blk[slot]) HIPCHK(hipHostFree(c->h_blk[slot]));
    c->h_blk[slot] = nullptr;
    c->blk_cap[slot] = 0;
    HIPCHK(hipHostMalloc((void **)&c->h_blk[slot], need, hipHostMallocDefault));
    c->blk_cap[slot] = need;
    return MFB_OK;
}

static bool batch_split(const mfb_ctx *c);
static int second_stream(mfb_ctx *c, int slot);
// Enqueue: everything up to and including the ONE device-to-host copy into the flight's page-locked staging; no wait.
static int block_begin(mfb_ctx *c, const mfb_block_params *p, int slot) {
    if (!c || !p || slot < 0 || slot > 1) return MFB_ERR_ARG;
    int rc = check_block_params(c, p);
    if (rc) return rc;
    if (p->input != MFB_INPUT_PINNED && p->input != MFB_INPUT_PINNED2 && p->input != MFB_INPUT_DEVICE && p->input != MFB_INPUT_UPLOADED)
        return MFB_ERR_ARG;
    if (p->input == MFB_INPUT_DEVICE && !p->device_block) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    BlockFlight &f = c->flight[slot];
    if (f.active) return MFB_ERR_STATE;          // its results have not been collected
    const int bcap = p->mode == MFB_BLOCK_SEARCH ? p->band_capacity : 0;
    if ((rc = blkout_reserve(c, bcap > c->band_cap ? bcap : c->band_cap))) return rc;
    const int capacity = p->max_symbols < c->cap ? p->max_symbols : c->cap;
    // every symbol the rate window admits (k* < k_offset + k_len  =>  count <= k_offset + k_len), bounded by the capacity
    int nthreads = p->k_offset + p->k_len + 1;
    if (nthreads > capacity) nthreads = capacity;
    if ((rc = staging_reserve(c, slot, blkout_bytes(bcap, nthreads)))) return rc;
    if (!c->ev_blk[slot]) HIPCHK(hipEventCreateWithFlags(&c->ev_blk[slot], hipEventDisableTiming));
    int shift = p->mode == MFB_BLOCK_FIXED_SHIFT ? ((p->fixed_shift % c->N) + c->N) % c->N : 0;
    const bool pinned_in = p->input == MFB_INPUT_PINNED || p->input == MFB_INPUT_PINNED2;
    const int which = p->input == MFB_INPUT_PINNED2 ? 1 : 0;
    if (pinned_in && (rc = block_input_copy(c, which))) return rc;
    // the samples of a page-locked buffer are (being) copied by block_input_copy on the input stream; this stream waits there
    if (pinned_in) c->d_in = which ? c->d_x2 : c->d_x;
    else if (p->input == MFB_INPUT_DEVICE) c->d_in = (const cf *)p->device_block;
    const bool allowed = pinned_in && graphs_allowed() && !c->prof && !c->mirror && (p->input == MFB_INPUT_PINNED || c->h_in2);
    // two parts on two streams (mfb_set_batch_overlap), as for batches: the NEXT block's forward transform and search run beside this
    // block's matched filters, envelope transform, rate, centres and read-back.  Part 2 reads the samples (the STORE kernel), so the
    // input has to be one of the two page-locked buffers' device copies or the caller's device block -- not the handle's upload
    const bool split = batch_split(c) && p->input != MFB_INPUT_UPLOADED && c->path == MFB_PATH_SEGMENT;
    if (!split) {
        rc = graph_or_launch(c, c->bgraph[which][slot], p, 0, allowed,
                             [&]() { return block_enqueue(c, p, single_bufs(c), c->h_blk[slot], nthreads, bcap, capacity, &shift); });
        if (rc) return rc;
        c->have_input = true;
        HIPCHK(hipEventRecord(c->ev_blk[slot], c->stream));
        if (pinned_in) HIPCHK(hipEventRecord(c->ev_xfree[which], c->stream));      // this device copy may be overwritten from here on
    } else {
        if ((rc = second_stream(c, slot))) return rc;
        if (c->z2_rows < 1) {
            HIPCHK(sync_streams(c));
            HIPCHK(dev_alloc((void **)&c->d_Z2, (size_t)c->N * sizeof(cf)));
            c->z2_rows = 1;
        }
        BlkBufs bb = single_bufs(c);
        bb.out = slot ? c->d_blkout2 : c->d_blkout;
        rc = graph_or_launch(c, c->bgraph[which][slot], p, -1, allowed, [&]() { return block_enqueue_p1(c, p, bb, nthreads, bcap); });
        if (rc) return rc;
        c->have_input = true;
        HIPCHK(hipEventRecord(c->ev_p1[slot], c->stream));
        hipStream_t s1 = c->stream;
        c->stream = c->s2;
        c->use_z2 = true;
        c->s2_busy = true;
        hipError_t e = hipStreamWaitEvent(c->s2, c->ev_p1[slot], 0);
        rc = e != hipSuccess ? MFB_ERR_HIP
                             : graph_or_launch(c, c->bgraph2[which][slot], p, -1, allowed,
                                               [&]() { return block_enqueue_p2(c, p, bb, c->h_blk[slot], nthreads, bcap, capacity, &shift); });
        if (!rc) {
            e = hipEventRecord(c->ev_blk[slot], c->s2);
            if (e == hipSuccess && pinned_in) e = hipEventRecord(c->ev_xfree[which], c->s2);
            if (e != hipSuccess) rc = MFB_ERR_HIP;
        }
        c->stream = s1;
        c->use_z2 = false;
        if (rc) return rc;
    }
    const size_t sym_off = BLK_HEAD + align16((size_t)2 * bcap * sizeof(cf)), arr = align16((size_t)nthreads * sizeof(int));
    f.off[0] = 0;
    f.off[4] = BLK_HEAD;
    f.off[1] = sym_off;
    f.off[2] = sym_off + arr;
    f.off[3] = sym_off + 2 * arr;
    f.active = true;
    f.mode = p->mode;
    f.nthreads = nthreads;
    f.bcap = bcap;
    f.shift = shift;
    f.seq = ++c->blk_seq;
    f.op = p->op;
    f.nb = 0;
    f.rec = 0;
    c->have_xc = true;
    return MFB_OK;
}

static void fill_result(mfb_block_result *r, const BlockScalars &hs, int mode, int fixed_shift) {
    r->pick[0] = hs.pick[0];
    r->pick[1] = hs.pick[1];
    r->pick_valid = hs.pick_valid;
    r->shift = mode == MFB_BLOCK_SEARCH ? hs.shift : fixed_shift;
    r->low = hs.low;
    r->high = hs.high;
    r->frac = hs.frac;
    r->cr[0] = hs.cr[0];
    r->cr[1] = hs.cr[1];
    r->cr[2] = hs.cr[2];
    r->spSym = hs.spSym;
    r->codeOffset = hs.codeOffset;
    r->count = hs.count;
    r->rate_fallback = hs.rate_fallback;
    r->band_len[0] = hs.band_len[0];
    r->band_len[1] = hs.band_len[1];
}

// Collect: wait for the flight's event, hand the results out.
static int block_end(mfb_ctx *c, int slot, mfb_block_result *r, int32_t *sym, int32_t *cen, float *mag, float *bands_c64) {
    if (!c || slot < 0 || slot > 1 || !r || !sym || !cen || !mag) return MFB_ERR_ARG;
    BlockFlight &f = c->flight[slot];
    if (!f.active || f.nb) return MFB_ERR_STATE;
    if (f.bcap > 0 && f.mode == MFB_BLOCK_SEARCH && !bands_c64) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_blk[slot]));
    f.active = false;
    const uint8_t *h = c->h_blk[slot];
    BlockScalars hs;
    memcpy(&hs, h + f.off[0], sizeof(hs));
    int n = hs.count;
    if (n > f.nthreads) {
        // rate fallback (k* == 0 -> spSym = 10, DB:737-740; unreachable while the rate window starts above bin 0): the
        // record holds the symbols the window admits; the rest is computed now, if the matched-filter outputs are still
        // this block's
        if (c->blk_seq != f.seq) return MFB_ERR_UNSUPPORTED;
        const int nb = (n + 255) / 256;
        hipLaunchKernelGGL(k_centres, dim3(nb), dim3(256), 0, c->stream, c->d_sym, c->d_cen, c->d_mag, (const cf *)c->d_xc, hs.spSymF,
                           hs.offsetF, c->N, c->M, c->W, f.op, c->cap);
        HIPCHK(hipGetLastError());
        const int more = n - f.nthreads;
        const BackPiece bq[3] = {{sym + f.nthreads, c->d_sym + f.nthreads, (size_t)more * sizeof(int)},
                                 {cen + f.nthreads, c->d_cen + f.nthreads, (size_t)more * sizeof(int)},
                                 {mag + f.nthreads, c->d_mag + f.nthreads, (size_t)more * sizeof(float)}};
        const int rc = read_back(c, bq, 3);
        if (rc) return rc;
        n = f.nthreads;
    }
    memcpy(sym, h + f.off[1], (size_t)n * sizeof(int));
    memcpy(cen, h + f.off[2], (size_t)n * sizeof(int));
    memcpy(mag, h + f.off[3], (size_t)n * sizeof(float));
    if (f.mode == MFB_BLOCK_SEARCH && f.bcap > 0) {
        const int l0 = hs.band_len[0] < f.bcap ? hs.band_len[0] : f.bcap, l1 = hs.band_len[1] < f.bcap ? hs.band_len[1] : f.bcap;
        memcpy(bands_c64, h + f.off[4], (size_t)l0 * sizeof(cf));
        memcpy(bands_c64 + (size_t)2 * f.bcap, h + f.off[4] + (size_t)f.bcap * sizeof(cf), (size_t)l1 * sizeof(cf));
    }
    fill_result(r, hs, f.mode, f.shift);
    return MFB_OK;
}

extern "C" int mfb_receive_block_begin(mfb_ctx *c, const mfb_block_params *p, int slot) { return block_begin(c, p, slot); }
extern "C" int mfb_receive_block_end(mfb_ctx *c, int slot, mfb_block_result *r, int32_t *sym, int32_t *cen, float *mag, float *bands_c64) {
    return block_end(c, slot, r, sym, cen, mag, bands_c64);
}
extern "C" int mfb_receive_block(mfb_ctx *c, const mfb_block_params *p, mfb_block_result *r, int32_t *sym, int32_t *cen, float *mag,
                                 float *bands_c64) {
    if (!c || !p || !r || !sym || !cen || !mag || (p->band_capacity > 0 && p->mode == MFB_BLOCK_SEARCH && !bands_c64)) return MFB_ERR_ARG;
    if (c->flight[0].active) return MFB_ERR_STATE;
    const int rc = block_begin(c, p, 0);
    if (rc) return rc;
    return block_end(c, 0, r, sym, cen, mag, bands_c64);
}

// ---- B consecutive blocks per call ---------------------------------------------------------------------------------------
// The reference's own configurations use blocks of 2^15 ... 2^17 samples with 64 bins (config/base.json:13,33): one such block
// is a few tens of microseconds of device work, and a loop that hands the device one block per turn (DP:284-338) leaves it idle
// most of the time.  Here B consecutive blocks of the stream go to the device as ONE contiguous window of
// B * stride + (N - stride) samples (stride = N - overlap: block b starts at b * stride -- the overlap between neighbours is
// shared storage, nothing is copied twice) and through ONE set of launches: batched forward FFT, one search launch over
// B x D (block, bin) streams, B picks in one launch, one matched-filter launch at the B shifts, batched envelope FFT, B rate
// estimates, the centres of all blocks, one device-to-host copy of B result records.
static int window_samples(const mfb_ctx *c, int nb) { return nb * c->win_stride + (c->N - c->win_stride); }

// the second stream of a block / batch in two parts, and the event of the flight's part 1
static int second_stream(mfb_ctx *c, int slot) {
    if (!c->s2) {
        // (the highest priority: part 2 is a latency chain whose workgroups should get the slots the big kernel frees first)
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&c->s2, hipStreamNonBlocking, greatest));
    }
    if (!c->ev_p1[slot]) HIPCHK(hipEventCreateWithFlags(&c->ev_p1[slot], hipEventDisableTiming));
    return MFB_OK;
}
// A batch on two streams (mfb_set_batch_overlap; include/mfbank.h): the handle's setting, or -- every handle -- MFB_BATCH_SPLIT=0/1 in
// the environment (the A/B switch of profiles/r06_chain.md)
static bool batch_split(const mfb_ctx *c) {
    static const int env = getenv("MFB_BATCH_SPLIT") ? atoi(getenv("MFB_BATCH_SPLIT")) : -1;
    return env >= 0 ? env != 0 : c->batch_overlap;
}
extern "C" int mfb_set_batch_overlap(mfb_ctx *c, int on) {
    if (!c || on < 0 || on > 1) return MFB_ERR_ARG;
    for (int s = 0; s < 2; ++s)
        if (c->flight[s].active && c->flight[s].nb) return MFB_ERR_STATE;      // a batch is in flight
    if (c->batch_overlap != (on != 0)) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(sync_streams(c));
        ++c->epoch;               // the recorded graphs belong to the other arrangement
        c->batch_overlap = on != 0;
    }
    return MFB_OK;
}
static int batch_reserve(mfb_ctx *c, int nb, size_t rec) {
    if (nb > c->bat_cap) {
        HIPCHK(sync_streams(c));
        ++c->epoch;        // recorded graphs hold the old addresses
        void **bufs[] = {(void **)&c->d_Xb, (void **)&c->d_xcb, (void **)&c->d_Pb, (void **)&c->d_envb, (void **)&c->d_sumb, (void **)&c->d_resb,
                         (void **)&c->d_crb};
        for (void **b : bufs) {
            if (*b) HIPCHK(hipFree(*b));
            *b = nullptr;
        }
        c->bat_cap = 0;
        const size_t nbN = (size_t)nb * c->N;
        HIPCHK(dev_alloc((void **)&c->d_Xb, nbN * sizeof(cf)));
        HIPCHK(dev_alloc((void **)&c->d_xcb, nbN * c->M * sizeof(cf)));
        HIPCHK(dev_alloc((void **)&c->d_Pb, nbN * sizeof(cf)));
        HIPCHK(dev_alloc((void **)&c->d_envb, nbN * sizeof(float)));
        HIPCHK(dev_alloc((void **)&c->d_sumb, (size_t)nb * c->Dtot * c->M * sizeof(float)));
        HIPCHK(dev_alloc((void **)&c->d_resb, (size_t)nb * 2 * sizeof(float)));
        HIPCHK(dev_alloc((void **)&c->d_crb, (size_t)nb * 3 * sizeof(float)));
        c->bat_cap = nb;
    }
    if ((size_t)nb > c->z_rows) {       // the plain forward transforms of the batch take one row of the intermediate each
        HIPCHK(sync_streams(c));
        ++c->epoch;
        if (c->d_Z) HIPCHK(hipFree(c->d_Z));
        c->d_Z = nullptr;
        c->z_rows = 0;
        HIPCHK(dev_alloc((void **)&c->d_Z, (size_t)nb * c->N * sizeof(cf)));
        c->z_rows = (size_t)nb;
    }
    if (rec * nb > c->batout_cap) {
        HIPCHK(sync_streams(c));
        ++c->epoch;
        if (c->d_batout) HIPCHK(hipFree(c->d_batout));
        if (c->d_batout2) HIPCHK(hipFree(c->d_batout2));
        c->d_batout = c->d_batout2 = nullptr;
        c->batout_cap = 0;
        HIPCHK(dev_alloc((void **)&c->d_batout, rec * nb));
        HIPCHK(dev_alloc((void **)&c->d_batout2, rec * nb));
        c->batout_cap = rec * nb;
    }
    if (batch_split(c) && (size_t)nb > c->z2_rows) {       // part 2's transforms (the envelopes' spectra) get an intermediate of their own
        HIPCHK(sync_streams(c));
        ++c->epoch;
        if (c->d_Z2) HIPCHK(hipFree(c->d_Z2));
        c->d_Z2 = nullptr;
        c->z2_rows = 0;
        HIPCHK(dev_alloc((void **)&c->d_Z2, (size_t)nb * c->N * sizeof(cf)));
        c->z2_rows = (size_t)nb;
    }
    return MFB_OK;
}

extern "C" int mfb_window_buffer(mfb_ctx *c, int which, int max_blocks, int block_stride, float **host_c64) {
    if (!c || !host_c64 || which < 0 || which > 1 || max_blocks < 1 || max_blocks > 1024 || block_stride < 1 || block_stride > c->N)
        return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->win_blocks != max_blocks || c->win_stride != block_stride) {
        for (int s = 0; s < 2; ++s)
            if (c->flight[s].active && c->flight[s].nb) return MFB_ERR_STATE;      // a batch is reading the windows
        HIPCHK(sync_streams(c));
        if (c->in_stream) HIPCHK(hipStreamSynchronize(c->in_stream));
        ++c->epoch;
        for (int i = 0; i < 2; ++i) {
            if (c->h_win[i]) HIPCHK(hipHostFree(c->h_win[i]));
            if (c->d_win[i]) HIPCHK(hipFree(c->d_win[i]));
            c->h_win[i] = nullptr;
            c->d_win[i] = nullptr;
        }
        c->win_blocks = max_blocks;
        c->win_stride = block_stride;
    }
    if (!c->h_win[which]) {
        const size_t bytes = (size_t)window_samples(c, c->win_blocks) * sizeof(cf);
        HIPCHK(hipHostMalloc((void **)&c->h_win[which], bytes, hipHostMallocDefault));
        memset(c->h_win[which], 0, bytes);
        HIPCHK(dev_alloc((void **)&c->d_win[which], bytes));
    }
    *host_c64 = (float *)c->h_win[which];
    return MFB_OK;
}

extern "C" int mfb_receive_blocks_begin(mfb_ctx *c, const mfb_block_params *p, int nblocks, int slot) {
    if (!c || !p || slot < 0 || slot > 1 || nblocks < 1) return MFB_ERR_ARG;
    int rc = check_block_params(c, p);
    if (rc) return rc;
    const bool win_in = p->input == MFB_INPUT_WINDOW || p->input == MFB_INPUT_WINDOW2;
    if (!win_in && p->input != MFB_INPUT_DEVICE) return MFB_ERR_ARG;
    const int which = p->input == MFB_INPUT_WINDOW2 ? 1 : 0;
    int stride = p->block_stride;
    if (win_in) {
        if (!c->h_win[which] || nblocks > c->win_blocks) return MFB_ERR_STATE;
        if (stride == 0) stride = c->win_stride;
        if (stride != c->win_stride) return MFB_ERR_ARG;
    } else if (!p->device_block || stride < 1 || stride > c->N) {
        return MFB_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    if (c->path != MFB_PATH_SEGMENT || (p->mode == MFB_BLOCK_SEARCH && c->search_mode != MFB_SEARCH_TRANSFORMS)) return MFB_ERR_UNSUPPORTED;
    BlockFlight &f = c->flight[slot];
    if (f.active) return MFB_ERR_STATE;
    const int bcap = p->mode == MFB_BLOCK_SEARCH ? p->band_capacity : 0;
    const int capacity = p->max_symbols < c->cap ? p->max_symbols : c->cap;
    int nthreads = p->k_offset + p->k_len + 1;
    if (nthreads > capacity) nthreads = capacity;
    const size_t core = blkout_bytes(bcap, nthreads);
    const bool stages = c->st_on && p->mode == MFB_BLOCK_SEARCH && nblocks <= 64;
    // stream-stage outputs behind the core record: kept bits | kept centres | trust bytes (uint8[nthreads] each), the block's
    // tail (post, end), the sync hits (idx | score per template)
    const size_t ext_bytes = 3 * align16((size_t)nthreads) + STREAM_POST_MAX + STREAM_END_MAX +
                             (size_t)STREAM_MAX_TMPL * 2 * STREAM_MAX_HITS * sizeof(int32_t) + align16(STREAM_EDGE_CANDS * sizeof(StreamEdge));
    const size_t rec = core + (stages ? ext_bytes : 0);
    if ((rc = batch_reserve(c, nblocks > c->win_blocks ? nblocks : (c->win_blocks > 0 ? c->win_blocks : nblocks), rec))) return rc;
    if ((rc = staging_reserve(c, slot, rec * nblocks))) return rc;
    if (!c->ev_blk[slot]) HIPCHK(hipEventCreateWithFlags(&c->ev_blk[slot], hipEventDisableTiming));
    if (win_in && (rc = input_copy(c, c->h_win[which], c->d_win[which], (size_t)(nblocks * stride + (c->N - stride)), c->ev_wh2d, c->ev_wfree, which)))
        return rc;
    const int parity = c->carry_cur;
    // (the two flights' records live in buffers of their own: the next batch's pick writes into its records while this batch's
    // are still being read on the other stream)
    BlkBufs bb{nblocks, win_in ? (const cf *)c->d_win[which] : (const cf *)p->device_block, stride, c->d_Xb, c->d_sumb, c->d_resb, c->d_xcb,
               c->d_envb, c->d_Pb, c->d_crb, slot ? c->d_batout2 : c->d_batout, rec, stages ? core : 0, parity};
    int shift = p->mode == MFB_BLOCK_FIXED_SHIFT ? ((p->fixed_shift % c->N) + c->N) % c->N : 0;
    const bool allowed = win_in && graphs_allowed() && !c->prof;
    mfb_block_params q = *p;
    q.block_stride = stride;
    const int gi = nblocks <= WG_NB ? nblocks : 0;
    if (!batch_split(c)) {
        rc = graph_or_launch(c, c->wgraph[which][slot][parity][gi], &q, nblocks, allowed,
                             [&]() { return block_enqueue(c, &q, bb, c->h_blk[slot], nthreads, bcap, capacity, &shift); });
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_blk[slot], c->stream));
        if (win_in) HIPCHK(hipEventRecord(c->ev_wfree[which], c->stream));
    } else {
        // part 1 on the handle's stream, part 2 behind it on the second stream: the NEXT batch's part 1 is enqueued behind this
        // one's part 1 only, and runs beside this batch's part 2
        if ((rc = second_stream(c, slot))) return rc;
        rc = graph_or_launch(c, c->wgraph[which][slot][parity][gi], &q, nblocks, allowed,
                             [&]() { return block_enqueue_p1(c, &q, bb, nthreads, bcap); });
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_p1[slot], c->stream));
        hipStream_t s1 = c->stream;
        c->stream = c->s2;            // every launch helper enqueues on c->stream
        c->use_z2 = true;
        c->s2_busy = true;
        hipError_t e = hipStreamWaitEvent(c->s2, c->ev_p1[slot], 0);
        rc = e != hipSuccess ? MFB_ERR_HIP
                             : graph_or_launch(c, c->wgraph2[which][slot][parity][gi], &q, nblocks, allowed,
                                               [&]() { return block_enqueue_p2(c, &q, bb, c->h_blk[slot], nthreads, bcap, capacity, &shift); });
        if (!rc) {
            e = hipEventRecord(c->ev_blk[slot], c->s2);
            if (e == hipSuccess && win_in) e = hipEventRecord(c->ev_wfree[which], c->s2);
            if (e != hipSuccess) rc = MFB_ERR_HIP;
        }
        c->stream = s1;
        c->use_z2 = false;
        if (rc) return rc;
    }
    if (stages) c->carry_cur = 1 - parity;        // this batch's tail and ring are the next batch's start
    const size_t sym_off = BLK_HEAD + align16((size_t)2 * bcap * sizeof(cf)), arr = align16((size_t)nthreads * sizeof(int));
    f.off[0] = 0;
    f.off[4] = BLK_HEAD;
    f.off[1] = sym_off;
    f.off[2] = sym_off + arr;
    f.off[3] = sym_off + 2 * arr;
    f.active = true;
    f.mode = p->mode;
    f.nthreads = nthreads;
    f.bcap = bcap;
    f.shift = shift;
    f.seq = ++c->blk_seq;
    f.op = p->op;
    f.nb = nblocks;
    f.rec = rec;
    f.ext = stages ? core : 0;
    c->last_batch_blocks = p->mode == MFB_BLOCK_SEARCH ? nblocks : 0;
    // the handle's one-block buffers (spectrum, matched-filter outputs) hold nothing of this batch
    c->have_xc = false;
    return MFB_OK;
}

extern "C" int mfb_receive_blocks_end(mfb_ctx *c, int slot, mfb_block_result *results, int32_t *sym, int32_t *cen, float *mag,
                                      int symbol_stride, float *bands_c64) {
    if (!c || slot < 0 || slot > 1 || !results || !sym || !cen || !mag || symbol_stride < 1) return MFB_ERR_ARG;
    BlockFlight &f = c->flight[slot];
    if (!f.active || !f.nb) return MFB_ERR_STATE;
    if (f.bcap > 0 && f.mode == MFB_BLOCK_SEARCH && !bands_c64) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_blk[slot]));
    f.active = false;
    int rc = MFB_OK;
    for (int b = 0; b < f.nb; ++b) {
        const uint8_t *h = c->h_blk[slot] + (size_t)b * f.rec;
        BlockScalars hs;
        memcpy(&hs, h + f.off[0], sizeof(hs));
        int n = hs.count;
        // (the rate fallback k* == 0 -> spSym = 10 could ask for more symbols than the rate window admits; it cannot occur while
        // the window starts above bin 0, and a batch has no second pass to fetch them: the record's symbols are delivered)
        if (n > f.nthreads) {
            n = f.nthreads;
            rc = MFB_ERR_UNSUPPORTED;
        }
        if (n > symbol_stride) return MFB_ERR_ARG;
        memcpy(sym + (size_t)b * symbol_stride, h + f.off[1], (size_t)n * sizeof(int));
        memcpy(cen + (size_t)b * symbol_stride, h + f.off[2], (size_t)n * sizeof(int));
        memcpy(mag + (size_t)b * symbol_stride, h + f.off[3], (size_t)n * sizeof(float));
        if (f.mode == MFB_BLOCK_SEARCH && f.bcap > 0) {
            const int l0 = hs.band_len[0] < f.bcap ? hs.band_len[0] : f.bcap, l1 = hs.band_len[1] < f.bcap ? hs.band_len[1] : f.bcap;
            float *dst = bands_c64 + (size_t)b * 4 * f.bcap;
            memcpy(dst, h + f.off[4], (size_t)l0 * sizeof(cf));
            memcpy(dst + (size_t)2 * f.bcap, h + f.off[4] + (size_t)f.bcap * sizeof(cf), (size_t)l1 * sizeof(cf));
        }
        fill_result(&results[b], hs, f.mode, f.shift);
    }
    return rc;
}

// The whole batch as it came off the device: nb records of `rec` bytes copied into the caller's buffer, and where things are
// inside a record.  The caller (the Python host) reads scalars and arrays in place -- one copy per batch instead of a dozen
// small ones per block.
extern "C" int mfb_receive_blocks_end_record(mfb_ctx *c, int slot, void *dst, size_t capacity, mfb_record_layout *lay) {
    if (!c || slot < 0 || slot > 1 || !dst || !lay) return MFB_ERR_ARG;
    BlockFlight &f = c->flight[slot];
    if (!f.active || !f.nb) return MFB_ERR_STATE;
    if (capacity < f.rec * (size_t)f.nb) {
        // too small: say what the batch needs (nblocks * record_bytes) and leave it in flight for the caller's second attempt
        memset(lay, 0, sizeof(*lay));
        lay->nblocks = f.nb;
        lay->record_bytes = (int64_t)f.rec;
        return MFB_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_blk[slot]));
    f.active = false;
    memcpy(dst, c->h_blk[slot], f.rec * (size_t)f.nb);
    memset(lay, 0, sizeof(*lay));
    lay->nblocks = f.nb;
    lay->record_bytes = (int64_t)f.rec;
    lay->scalars_bytes = (int32_t)sizeof(BlockScalars);
    lay->symbols = f.nthreads;
    lay->band_capacity = f.bcap;
    lay->mode = f.mode;
    lay->fixed_shift = f.shift;
    lay->off_bands = (int64_t)f.off[4];
    lay->off_sym = (int64_t)f.off[1];
    lay->off_cen = (int64_t)f.off[2];
    lay->off_mag = (int64_t)f.off[3];
    if (f.ext) {
        const size_t a1 = align16((size_t)f.nthreads);
        lay->stream_stages = 1;
        lay->off_bits = (int64_t)f.ext;
        lay->off_centres_u8 = (int64_t)(f.ext + a1);
        lay->off_trust = (int64_t)(f.ext + 2 * a1);
        lay->off_post = (int64_t)(f.ext + 3 * a1);
        lay->off_end = lay->off_post + STREAM_POST_MAX;
        lay->off_hits = lay->off_end + STREAM_END_MAX;
        lay->off_edges = lay->off_hits + (int64_t)STREAM_MAX_TMPL * 2 * STREAM_MAX_HITS * (int64_t)sizeof(int32_t);
        lay->edge_candidates = STREAM_EDGE_CANDS;
        lay->edge_hits = STREAM_EDGE_HITS;
        lay->max_hits = STREAM_MAX_HITS;
        lay->templates = c->st.K;
    }
    return MFB_OK;
}

// The integer stages behind the symbol decisions on the device, for batches (stream_kernels.hpp).
extern "C" int mfb_set_stream_stages(mfb_ctx *c, const mfb_stream_params *p) {
    if (!c) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));
    ++c->epoch;
    c->st_on = false;
    if (!p) return MFB_OK;                 // switched off
    // (the LUTs live in the LDS of k_stream_align: 256 rows of a bit LUT, 2048 entries of an NRZ-S LUT -- 2^xcorrMaskSize is 8 ... 32)
    if ((p->lut_mode != 1 && p->lut_mode != 2) || p->lut_rows < 1 || (p->lut_mode == 1 && p->lut_rows > STREAM_LUT8_MAX) ||
        (p->lut_mode == 2 && (p->lut_successors < 1 || (long long)p->lut_rows * 2 * p->lut_successors > STREAM_LUT3_MAX)) || !p->lut || p->overlap_samples < 2 ||
        p->overlap_samples >= c->N || p->overlap_offset < 1 || p->overlap_offset + 1 > STREAM_END_MAX || p->num_templates < 0 ||
        p->num_templates > STREAM_MAX_TMPL || (p->lut_mode == 2 && (p->lut_successors < 1 || p->lut_successors > 64)))
        return MFB_ERR_ARG;
    if (p->num_templates > 0 && (!p->templates || p->bits_overlap < 1 || p->bits_overlap > STREAM_NOV_MAX)) return MFB_ERR_ARG;
    StreamArgs &a = c->st;
    memset(&a, 0, sizeof(a));
    a.N = c->N;
    a.ovw = p->overlap_samples / 2;
    a.o = p->overlap_offset;
    a.thr = p->match_threshold;
    a.err_thr = p->error_threshold;
    a.mode = p->lut_mode;
    a.rows = p->lut_rows;
    a.succ = p->lut_mode == 2 ? p->lut_successors : 0;
    void **olds[] = {(void **)&c->d_lut8, (void **)&c->d_lut3, (void **)&c->d_sttmpl};
    for (void **q : olds) {
        if (*q) HIPCHK(hipFree(*q));
        *q = nullptr;
    }
    if (a.mode == 1) {
        HIPCHK(dev_alloc((void **)&c->d_lut8, (size_t)a.rows));
        HIPCHK(hipMemcpy(c->d_lut8, p->lut, (size_t)a.rows, hipMemcpyHostToDevice));
    } else {
        const size_t n = (size_t)a.rows * 2 * a.succ * sizeof(int);
        HIPCHK(dev_alloc((void **)&c->d_lut3, n));
        HIPCHK(hipMemcpy(c->d_lut3, p->lut, n, hipMemcpyHostToDevice));
    }
    a.lut8 = c->d_lut8;
    a.lut3 = c->d_lut3;
    a.K = p->num_templates;
    a.nOv = p->bits_overlap;
    a.max_hits = STREAM_MAX_HITS;
    size_t taps = 0;
    for (int t = 0; t < a.K; ++t) {
        if (p->template_taps[t] < 1 || p->template_taps[t] > 4096) return MFB_ERR_ARG;
        a.T[t] = p->template_taps[t];
        a.thrs[t] = p->template_thresholds[t];
        a.toff[t] = (int)taps;
        taps += (size_t)a.T[t];
    }
    // the packed search (k_stream_search) takes templates of up to STREAM_PACK_TAPS taps in {-1, 0, +1}
    a.packed = a.K > 0;
    memset(a.P, 0, sizeof(a.P));
    memset(a.Q, 0, sizeof(a.Q));
    for (int t = 0; t < a.K && a.packed; ++t) {
        const int8_t *tp = p->templates + a.toff[t];
        if (a.T[t] > STREAM_PACK_TAPS) a.packed = 0;
        for (int q = 0; q < a.T[t] && a.packed; ++q) {
            const int j = a.T[t] - 1 - q;                   // window element the tap multiplies
            if (tp[q] == 1) a.P[t][j >> 5] |= 1u << (j & 31);
            else if (tp[q] == -1) a.Q[t][j >> 5] |= 1u << (j & 31);
            else if (tp[q] != 0) a.packed = 0;
        }
    }
    if (taps) {
        HIPCHK(dev_alloc((void **)&c->d_sttmpl, taps));
        HIPCHK(hipMemcpy(c->d_sttmpl, p->templates, taps, hipMemcpyHostToDevice));
    }
    a.tmpls = c->d_sttmpl;
    for (int i = 0; i < 2; ++i) {
        if (!c->d_carry[i]) HIPCHK(dev_alloc((void **)&c->d_carry[i], sizeof(StreamCarry)));
        HIPCHK(hipMemset(c->d_carry[i], 0, sizeof(StreamCarry)));       // valid = 0: nothing known until mfb_stream_seed
    }
    if (!c->h_seed) HIPCHK(hipHostMalloc((void **)&c->h_seed, sizeof(StreamCarry), hipHostMallocDefault));
    c->carry_cur = 0;
    c->st_on = true;
    return MFB_OK;
}

extern "C" int mfb_stream_seed(mfb_ctx *c, const uint8_t *post, int npost, const uint8_t *end, int nend, const uint8_t *ring, int ring_len) {
    if (!c || npost < 0 || nend < 0 || ring_len < 0 || (npost && !post) || (nend && !end) || (ring_len && !ring)) return MFB_ERR_ARG;
    if (!c->st_on) return MFB_ERR_STATE;
    if (npost > STREAM_POST_MAX || nend > STREAM_END_MAX || ring_len > STREAM_NOV_MAX) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));          // h_seed may still be on its way from the last seed
    StreamCarry *h = c->h_seed;
    memset(h, 0, sizeof(*h));
    h->valid = 1;
    h->npost = npost;
    h->nend = nend;
    if (npost) memcpy(h->post, post, (size_t)npost);
    if (nend) memcpy(h->end, end, (size_t)nend);
    h->ring_valid = ring_len > 0 && ring_len == c->st.nOv;
    h->ring_len = ring_len;
    if (ring_len) memcpy(h->ring, ring, (size_t)ring_len);
    HIPCHK(hipMemcpyAsync(c->d_carry[c->carry_cur], h, sizeof(*h), hipMemcpyHostToDevice, c->stream));
    return MFB_OK;
}

// Test seam of the stream stages (include/mfbank.h): k_stream_align / _sync / _ring / _edges on INJECTED symbol decisions.
extern "C" int mfb_debug_stream_stages(mfb_ctx *c, int nb, int symbols, const int32_t *counts, const int32_t *sym, const int32_t *cen,
                                       const float *mag, void *dst, size_t capacity, mfb_record_layout *lay) {
    if (!c || nb < 1 || nb > 64 || symbols < 1 || !counts || !sym || !cen || !mag || !dst || !lay) return MFB_ERR_ARG;
    if (!c->st_on) return MFB_ERR_STATE;
    for (int s = 0; s < 2; ++s)
        if (c->flight[s].active) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const size_t core = blkout_bytes(0, symbols), a1 = align16((size_t)symbols);
    const size_t ext_bytes = 3 * a1 + STREAM_POST_MAX + STREAM_END_MAX + (size_t)STREAM_MAX_TMPL * 2 * STREAM_MAX_HITS * sizeof(int32_t) +
                             align16(STREAM_EDGE_CANDS * sizeof(StreamEdge));
    const size_t rec = core + ext_bytes;
    if (capacity < rec * (size_t)nb) return MFB_ERR_ARG;
    int rc = batch_reserve(c, nb > c->bat_cap ? nb : (c->bat_cap > 0 ? c->bat_cap : nb), rec);
    if (rc) return rc;
    if (rec * nb > c->batout_cap) return MFB_ERR_ALLOC;
    std::vector<uint8_t> h(rec * (size_t)nb, 0);
    const size_t off_sym = BLK_HEAD, arr = align16((size_t)symbols * sizeof(int));
    for (int b = 0; b < nb; ++b) {
        if (counts[b] < 0 || counts[b] > symbols) return MFB_ERR_ARG;
        uint8_t *r = h.data() + (size_t)b * rec;
        BlockScalars sc;
        memset(&sc, 0, sizeof(sc));
        sc.count = counts[b];
        memcpy(r, &sc, sizeof(sc));
        memcpy(r + off_sym, sym + (size_t)b * symbols, (size_t)symbols * sizeof(int));
        memcpy(r + off_sym + arr, cen + (size_t)b * symbols, (size_t)symbols * sizeof(int));
        memcpy(r + off_sym + 2 * arr, mag + (size_t)b * symbols, (size_t)symbols * sizeof(float));
    }
    HIPCHK(hipMemcpyAsync(c->d_batout, h.data(), h.size(), hipMemcpyHostToDevice, c->stream));
    StreamArgs sa = c->st;
    sa.rec0 = c->d_batout;
    sa.rec = rec;
    sa.off_sym = off_sym;
    sa.off_cen = off_sym + arr;
    sa.off_mag = off_sym + 2 * arr;
    sa.off_bits = core;
    sa.off_cenw = core + a1;
    sa.off_trust = core + 2 * a1;
    sa.off_post = core + 3 * a1;
    sa.off_end = sa.off_post + STREAM_POST_MAX;
    sa.off_hits = sa.off_end + STREAM_END_MAX;
    sa.off_edges = sa.off_hits + (size_t)STREAM_MAX_TMPL * 2 * STREAM_MAX_HITS * sizeof(int32_t);
    sa.nb = nb;
    sa.nsym = symbols;
    sa.carry_in = c->d_carry[c->carry_cur];
    sa.carry_out = c->d_carry[1 - c->carry_cur];
    hipLaunchKernelGGL(k_stream_align, dim3(nb), dim3(STREAM_ALIGN_THREADS), 0, c->stream, sa);
    if (sa.K > 0) {
        int Tmax = 0;
        for (int t = 0; t < sa.K; ++t) Tmax = sa.T[t] > Tmax ? sa.T[t] : Tmax;
        stream_search_launch(c, sa, nb, Tmax);
    }
    HIPCHK(hipGetLastError());
    c->carry_cur = 1 - c->carry_cur;
    HIPCHK(hipMemcpyAsync(dst, c->d_batout, rec * (size_t)nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_streams(c));
    memset(lay, 0, sizeof(*lay));
    lay->nblocks = nb;
    lay->record_bytes = (int64_t)rec;
    lay->scalars_bytes = (int32_t)sizeof(BlockScalars);
    lay->symbols = symbols;
    lay->mode = MFB_BLOCK_SEARCH;
    lay->off_bands = BLK_HEAD;
    lay->off_sym = (int64_t)sa.off_sym;
    lay->off_cen = (int64_t)sa.off_cen;
    lay->off_mag = (int64_t)sa.off_mag;
    lay->stream_stages = 1;
    lay->off_bits = (int64_t)sa.off_bits;
    lay->off_centres_u8 = (int64_t)sa.off_cenw;
    lay->off_trust = (int64_t)sa.off_trust;
    lay->off_post = (int64_t)sa.off_post;
    lay->off_end = (int64_t)sa.off_end;
    lay->off_hits = (int64_t)sa.off_hits;
    lay->off_edges = (int64_t)sa.off_edges;
    lay->edge_candidates = STREAM_EDGE_CANDS;
    lay->edge_hits = STREAM_EDGE_HITS;
    lay->max_hits = STREAM_MAX_HITS;
    lay->templates = c->st.K;
    return MFB_OK;
}

// Test seam of the one-call path (include/mfbank.h): block_pick_body / block_rate_body on injected device results.
extern "C" int mfb_debug_block_scalars(mfb_ctx *c, int n, const float *picks, const float *triples, int spsym_min, int snr_window,
                                       int max_symbols, mfb_block_result *results, float *launch_args, int32_t *band_pieces) {
    if (!c || n < 1 || !picks || !triples || !results || spsym_min < 2 || snr_window < 0 || max_symbols < 1) return MFB_ERR_ARG;
    if (!c->have_shifts) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    float *d_in = nullptr;
    BlockScalars *d_out = nullptr;
    std::vector<BlockScalars> h((size_t)n);
    int rc = MFB_OK;
    const size_t in_bytes = (size_t)n * 5 * sizeof(float);
    if (hipMalloc((void **)&d_in, in_bytes) != hipSuccess || hipMalloc((void **)&d_out, (size_t)n * sizeof(BlockScalars)) != hipSuccess) {
        rc = MFB_ERR_ALLOC;
    } else if (hipMemcpy(d_in, picks, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(d_in + 2 * (size_t)n, triples, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        rc = MFB_ERR_HIP;
    } else {
        const int capacity = max_symbols < c->cap ? max_symbols : c->cap;
        hipLaunchKernelGGL(k_block_scalars_debug, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, (const float *)d_in,
                           (const float *)(d_in + 2 * (size_t)n), (const int *)c->d_shifts, c->Dtot, c->N, snr_window, spsym_min, capacity,
                           d_out);
        if (hipGetLastError() != hipSuccess || sync_streams(c) != hipSuccess ||
            hipMemcpy(h.data(), d_out, (size_t)n * sizeof(BlockScalars), hipMemcpyDeviceToHost) != hipSuccess)
            rc = MFB_ERR_HIP;
    }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const BlockScalars &hs = h[(size_t)i];
        mfb_block_result &r = results[i];
        r.pick[0] = hs.pick[0];
        r.pick[1] = hs.pick[1];
        r.pick_valid = hs.pick_valid;
        r.shift = hs.shift;
        r.low = hs.low;
        r.high = hs.high;
        r.frac = hs.frac;
        r.cr[0] = hs.cr[0];
        r.cr[1] = hs.cr[1];
        r.cr[2] = hs.cr[2];
        r.spSym = hs.spSym;
        r.codeOffset = hs.codeOffset;
        r.count = hs.count;
        r.rate_fallback = hs.rate_fallback;
        r.band_len[0] = hs.band_len[0];
        r.band_len[1] = hs.band_len[1];
        if (launch_args) {
            launch_args[2 * i] = hs.spSymF;
            launch_args[2 * i + 1] = hs.offsetF;
        }
        if (band_pieces) memcpy(band_pieces + 8 * (size_t)i, &hs.band[0][0][0], 8 * sizeof(int32_t));
    }
    return MFB_OK;
}

extern "C" int mfb_input_buffer2(mfb_ctx *c, float **p) {
    if (!c || !p) return MFB_ERR_ARG;
    if (!c->h_in2) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipHostMalloc((void **)&c->h_in2, (size_t)c->N * sizeof(cf), hipHostMallocDefault));
        memset(c->h_in2, 0, (size_t)c->N * sizeof(cf));
        if (!c->d_x2) HIPCHK(dev_alloc((void **)&c->d_x2, (size_t)c->N * sizeof(cf)));
    }
    *p = (float *)c->h_in2;
    return MFB_OK;
}

extern "C" int mfb_find_centres(mfb_ctx *c, float spSym, float offset, int op, int count, int32_t *sym, int32_t *cen,
                                float *mag) {
    if (!c || !sym || !cen || !mag) return MFB_ERR_ARG;
    if (!c->have_xc) return MFB_ERR_STATE;
    if (!(spSym >= 2.0f) || count < 0 || count > c->cap || op < 0 || op > 2) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    // same launch shape as the reference (ceil(N/spSym/256) blocks of 256, DB:996), bounded by capacity
    int nthreads = (int)ceil((double)c->N / (double)spSym / 256.0) * 256;
    if (nthreads > c->cap) nthreads = c->cap;
    if (nthreads < count) nthreads = count;
    const int nb = (nthreads + 255) / 256;
    // (every one of the nthreads >= count entries is written by k_centres itself: no clearing launches)
    hipLaunchKernelGGL(k_centres, dim3(nb), dim3(256), 0, c->stream, c->d_sym, c->d_cen, c->d_mag, c->d_xc, spSym, offset, c->N,
                       c->M, c->W, op, c->cap);
    HIPCHK(hipGetLastError());
    const BackPiece bp[3] = {{sym, c->d_sym, (size_t)count * sizeof(int)},
                             {cen, c->d_cen, (size_t)count * sizeof(int)},
                             {mag, c->d_mag, (size_t)count * sizeof(float)}};
    return read_back(c, bp, 3);
}

extern "C" int mfb_get_xcorr(mfb_ctx *c, float *host) {
    if (!c || !host) return MFB_ERR_ARG;
    if (!c->have_xc) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(host, c->d_xc, (size_t)c->M * c->N * sizeof(cf), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_streams(c));
    return MFB_OK;
}

extern "C" int mfb_get_envelope(mfb_ctx *c, float *host) {
    if (!c || !host) return MFB_ERR_ARG;
    if (!c->have_xc) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(host, c->d_env, (size_t)c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_streams(c));
    return MFB_OK;
}

// Workspaces of the (handle-less) sync correlator.  Every call borrows a workspace -- device buffers
// that only grow, plus a stream of its own -- from a per-device pool and returns it afterwards, so
// concurrent callers (two decoder threads, two demodulator instances on one device) never share
// buffers or serialise on the null stream.  The pool itself is guarded by a mutex.
struct SyncWs {
    hipStream_t stream = nullptr;
    uint8_t *bits = nullptr;
    int8_t *tmpl = nullptr;
    int32_t *out = nullptr;
    int *seg = nullptr;     // segcnt | segoff | counts
    int32_t *hits = nullptr;  // hit_idx | hit_score
    size_t cap_bits = 0, cap_tmpl = 0, cap_out = 0, cap_seg = 0, cap_hits = 0;
    // small calls (one block's bit stream, a few templates): inputs and results each travel as ONE packed copy
    // through page-locked staging -- every separate copy of pageable memory costs a round trip of its own
    uint8_t *h_stage = nullptr, *d_stage = nullptr;
    size_t cap_hstage = 0, cap_dstage = 0;
};
#define SYNC_STAGE_MAX ((size_t)1 << 20)
#define SYNC_MAX_DEVICES 64
static std::mutex g_sync_mu;
static std::vector<SyncWs *> g_sync_pool[SYNC_MAX_DEVICES];

static SyncWs *ws_acquire(int device) {
    {
        std::lock_guard<std::mutex> lk(g_sync_mu);
        auto &pool = g_sync_pool[device];
        if (!pool.empty()) {
            SyncWs *w = pool.back();
            pool.pop_back();
            return w;
        }
    }
    SyncWs *w = new SyncWs();
    // highest priority: the decoder's small searches run while the demodulator's stream has the next block's kernels queued
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&w->stream, hipStreamNonBlocking, hi) != hipSuccess) {
        delete w;
        return nullptr;
    }
    return w;
}
static void ws_release(int device, SyncWs *w) {
    std::lock_guard<std::mutex> lk(g_sync_mu);
    g_sync_pool[device].push_back(w);
}
struct WsLease {   // returns the workspace on every exit path
    int device;
    SyncWs *w;
    ~WsLease() {
        if (w) ws_release(device, w);
    }
};

template <class T>
static int ws_reserve(T **p, size_t *cap, size_t need) {
    if (*cap >= need) return MFB_OK;
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    HIPCHK(hipMalloc((void **)p, need));
    *cap = need;
    return MFB_OK;
}

static int sync_device_ok(int device) {
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || device >= SYNC_MAX_DEVICES) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(device));
    return MFB_OK;
}

extern "C" int mfb_sync_correlate(int device, const uint8_t *bits, int B, int L, const int8_t *tmpl, int T, int32_t *scores) {
    if (!bits || !tmpl || !scores || B < 1 || L < 1 || T < 1 || T > 4096) return MFB_ERR_ARG;
    int rc = sync_device_ok(device);
    if (rc) return rc;
    const int outLen = L + T - 1;
    WsLease lease{device, ws_acquire(device)};
    if (!lease.w) return MFB_ERR_HIP;
    SyncWs &w = *lease.w;
    if ((rc = ws_reserve(&w.bits, &w.cap_bits, (size_t)B * L))) return rc;
    if ((rc = ws_reserve(&w.tmpl, &w.cap_tmpl, (size_t)T))) return rc;
    if ((rc = ws_reserve(&w.out, &w.cap_out, (size_t)B * outLen * sizeof(int32_t)))) return rc;
    HIPCHK(hipMemcpyAsync(w.bits, bits, (size_t)B * L, hipMemcpyHostToDevice, w.stream));
    HIPCHK(hipMemcpyAsync(w.tmpl, tmpl, (size_t)T, hipMemcpyHostToDevice, w.stream));
    const int bs = 256;
    const size_t lds = (size_t)T + bs + T - 1;
    hipLaunchKernelGGL(k_sync_corr, dim3((outLen + bs - 1) / bs, B), dim3(bs), lds, w.stream, w.bits, w.tmpl, w.out, L, T);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(scores, w.out, (size_t)B * outLen * sizeof(int32_t), hipMemcpyDeviceToHost, w.stream));
    HIPCHK(hipStreamSynchronize(w.stream));
    return MFB_OK;
}


// K templates against the same B bit streams; template t: T[t] taps at tmpls + sum(T[0..t)), threshold thr[t];
// its results sit at counts + t*B, hit_idx / hit_score + t*B*max_hits.
static int sync_find_impl(int device, const uint8_t *bits, int B, int L, const int8_t *tmpls, const int *T, const int *thr, int K,
                          int max_hits, int32_t *hit_idx, int32_t *hit_score, int32_t *counts) {
    if (!bits || !tmpls || !T || !thr || !hit_idx || !hit_score || !counts || B < 1 || L < 1 || K < 1 || K > 16 || max_hits < 1)
        return MFB_ERR_ARG;
    size_t taps = 0;
    int Tmax = 0;
    for (int t = 0; t < K; ++t) {
        if (T[t] < 1 || T[t] > 4096) return MFB_ERR_ARG;
        taps += (size_t)T[t];
        if (T[t] > Tmax) Tmax = T[t];
    }
    int rc = sync_device_ok(device);
    if (rc) return rc;
    const int nseg = (L + Tmax - 1 + SYNC_SEG - 1) / SYNC_SEG;      // segments of the longest output
    WsLease lease{device, ws_acquire(device)};
    if (!lease.w) return MFB_ERR_HIP;
    SyncWs &w = *lease.w;
    const size_t nbits = (size_t)B * L;
    const size_t res_ints = (size_t)K * B * (1 + 2 * (size_t)max_hits);       // counts | idx | score, template-major
    const bool packed = align16(taps) + nbits <= SYNC_STAGE_MAX && res_ints * sizeof(int32_t) <= SYNC_STAGE_MAX;
    if ((rc = ws_reserve(&w.seg, &w.cap_seg, ((size_t)2 * B * nseg) * sizeof(int)))) return rc;
    if ((rc = ws_reserve(&w.hits, &w.cap_hits, res_ints * sizeof(int32_t)))) return rc;
    int32_t *d_counts = w.hits, *d_idx = w.hits + (size_t)K * B, *d_sc = d_idx + (size_t)K * B * max_hits;
    const uint8_t *d_bits;
    const int8_t *d_tmpl;
    if (packed) {
        const size_t in_bytes = align16(taps) + nbits;
        const size_t hneed = in_bytes > res_ints * sizeof(int32_t) ? in_bytes : res_ints * sizeof(int32_t);
        if (w.cap_hstage < hneed) {
            if (w.h_stage) HIPCHK(hipHostFree(w.h_stage));
            w.h_stage = nullptr;
            w.cap_hstage = 0;
            HIPCHK(hipHostMalloc((void **)&w.h_stage, hneed, hipHostMallocDefault));
            w.cap_hstage = hneed;
        }
        if ((rc = ws_reserve(&w.d_stage, &w.cap_dstage, in_bytes))) return rc;
        memcpy(w.h_stage, tmpls, taps);
        memcpy(w.h_stage + align16(taps), bits, nbits);
        HIPCHK(hipMemcpyAsync(w.d_stage, w.h_stage, in_bytes, hipMemcpyHostToDevice, w.stream));
        d_tmpl = (const int8_t *)w.d_stage;
        d_bits = w.d_stage + align16(taps);
    } else {
        if ((rc = ws_reserve(&w.bits, &w.cap_bits, nbits))) return rc;
        if ((rc = ws_reserve(&w.tmpl, &w.cap_tmpl, taps))) return rc;
        HIPCHK(hipMemcpyAsync(w.bits, bits, nbits, hipMemcpyHostToDevice, w.stream));
        HIPCHK(hipMemcpyAsync(w.tmpl, tmpls, taps, hipMemcpyHostToDevice, w.stream));
        d_bits = w.bits;
        d_tmpl = w.tmpl;
    }
    int *segcnt = w.seg, *segoff = w.seg + (size_t)B * nseg;
    size_t toff = 0;
    if ((long long)B * nseg * K <= 64) {
        // the decoder's case: everything in one launch (k_sync_small)
        SyncSmallArgs sa;
        size_t o = 0;
        for (int t = 0; t < K; ++t) {
            sa.T[t] = T[t];
            sa.thr[t] = thr[t];
            sa.toff[t] = (int)o;
            o += (size_t)T[t];
        }
        const size_t lds = (size_t)Tmax + SYNC_SEG + Tmax - 1;
        hipLaunchKernelGGL(k_sync_small, dim3(K, B), dim3(256), lds, w.stream, d_bits, d_tmpl, sa, B, L, max_hits, d_counts, d_idx, d_sc);
        HIPCHK(hipGetLastError());
    } else
    for (int t = 0; t < K; ++t) {
        const int Tt = T[t];
        const int ns = (L + Tt - 1 + SYNC_SEG - 1) / SYNC_SEG;
        const size_t lds = (size_t)Tt + SYNC_SEG + Tt - 1;
        int32_t *ci = d_counts + (size_t)t * B, *ii = d_idx + (size_t)t * B * max_hits, *si = d_sc + (size_t)t * B * max_hits;
        hipLaunchKernelGGL((k_sync_find<false>), dim3(ns, B), dim3(256), lds, w.stream, d_bits, d_tmpl + toff, L, Tt, thr[t], ns, segcnt,
                           (const int *)nullptr, max_hits, (int32_t *)nullptr, (int32_t *)nullptr);
        hipLaunchKernelGGL(k_sync_scan, dim3((B + 63) / 64), dim3(64), 0, w.stream, (const int *)segcnt, segoff, ci, B, ns);
        hipLaunchKernelGGL((k_sync_find<true>), dim3(ns, B), dim3(256), lds, w.stream, d_bits, d_tmpl + toff, L, Tt, thr[t], ns, segcnt,
                           (const int *)segoff, max_hits, ii, si);
        HIPCHK(hipGetLastError());
        toff += (size_t)Tt;
    }
    const size_t nc = (size_t)K * B, nh = (size_t)K * B * max_hits;
    if (packed) {
        HIPCHK(hipMemcpyAsync(w.h_stage, w.hits, res_ints * sizeof(int32_t), hipMemcpyDeviceToHost, w.stream));
        HIPCHK(hipStreamSynchronize(w.stream));
        const int32_t *r = (const int32_t *)w.h_stage;
        memcpy(counts, r, nc * sizeof(int32_t));
        memcpy(hit_idx, r + nc, nh * sizeof(int32_t));
        memcpy(hit_score, r + nc + nh, nh * sizeof(int32_t));
    } else {
        HIPCHK(hipMemcpyAsync(counts, d_counts, nc * sizeof(int32_t), hipMemcpyDeviceToHost, w.stream));
        HIPCHK(hipMemcpyAsync(hit_idx, d_idx, nh * sizeof(int32_t), hipMemcpyDeviceToHost, w.stream));
        HIPCHK(hipMemcpyAsync(hit_score, d_sc, nh * sizeof(int32_t), hipMemcpyDeviceToHost, w.stream));
        HIPCHK(hipStreamSynchronize(w.stream));
    }
    return MFB_OK;
}

extern "C" int mfb_sync_find(int device, const uint8_t *bits, int B, int L, const int8_t *tmpl, int T, int threshold,
                             int max_hits, int32_t *hit_idx, int32_t *hit_score, int32_t *counts) {
    return sync_find_impl(device, bits, B, L, tmpl, &T, &threshold, 1, max_hits, hit_idx, hit_score, counts);
}

extern "C" int mfb_sync_find_multi(int device, const uint8_t *bits, int B, int L, const int8_t *tmpls, const int *T,
                                   const int *thresholds, int ntmpl, int max_hits, int32_t *hit_idx, int32_t *hit_score,
                                   int32_t *counts) {
    return sync_find_impl(device, bits, B, L, tmpls, T, thresholds, ntmpl, max_hits, hit_idx, hit_score, counts);
}

// ---- a decoder's own sync finder -------------------------------------------------------------------------------------------
// The decoder searches the SAME templates in every block's bit stream (DEC:96-113).  A finder object keeps them on the
// device together with a stream, page-locked staging and result buffers of its own, and splits the search into begin
// (copy in, enqueue, return) and end (wait, copy out): the caller overlaps the round trip with its other work.
struct mfb_syncfinder {
    int device, K, max_hits, Tmax;
    std::vector<int> T, thr;
    hipStream_t stream;
    hipEvent_t done;
    int8_t *d_tmpl;
    uint8_t *d_bits, *h_bits;
    int cap_bits;
    int32_t *d_res, *h_res;        // counts[K] | idx[K][max_hits] | score[K][max_hits]
    int *d_seg;                    // segcnt | segoff of the multi-pass form (long streams)
    size_t cap_seg;
    SyncSmallArgs sa;
    bool busy;
};

static int sf_reserve_bits(mfb_syncfinder *f, int L) {
    if (L <= f->cap_bits) return MFB_OK;
    HIPCHK(hipStreamSynchronize(f->stream));
    if (f->d_bits) HIPCHK(hipFree(f->d_bits));
    if (f->h_bits) HIPCHK(hipHostFree(f->h_bits));
    f->d_bits = f->h_bits = nullptr;
    f->cap_bits = 0;
    const int cap = L + L / 2 + 1024;
    HIPCHK(hipMalloc((void **)&f->d_bits, (size_t)cap));
    HIPCHK(hipHostMalloc((void **)&f->h_bits, (size_t)cap, hipHostMallocDefault));
    f->cap_bits = cap;
    return MFB_OK;
}

extern "C" int mfb_syncfinder_destroy(mfb_syncfinder *f) {
    if (!f) return MFB_ERR_ARG;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    if (f->d_tmpl) (void)hipFree(f->d_tmpl);
    if (f->d_bits) (void)hipFree(f->d_bits);
    if (f->h_bits) (void)hipHostFree(f->h_bits);
    if (f->d_res) (void)hipFree(f->d_res);
    if (f->h_res) (void)hipHostFree(f->h_res);
    if (f->d_seg) (void)hipFree(f->d_seg);
    if (f->done) (void)hipEventDestroy(f->done);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
    return MFB_OK;
}

static int sf_create_impl(mfb_syncfinder *f, const int8_t *tmpls, int max_bits) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIPCHK(hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, hi));
    HIPCHK(hipEventCreateWithFlags(&f->done, hipEventDisableTiming));
    size_t taps = 0;
    for (int t = 0; t < f->K; ++t) {
        f->sa.T[t] = f->T[t];
        f->sa.thr[t] = f->thr[t];
        f->sa.toff[t] = (int)taps;
        taps += (size_t)f->T[t];
    }
    HIPCHK(hipMalloc((void **)&f->d_tmpl, taps));
    HIPCHK(hipMemcpy(f->d_tmpl, tmpls, taps, hipMemcpyHostToDevice));
    const size_t res_bytes = (size_t)f->K * (1 + 2 * (size_t)f->max_hits) * sizeof(int32_t);
    HIPCHK(hipMalloc((void **)&f->d_res, res_bytes));
    HIPCHK(hipHostMalloc((void **)&f->h_res, res_bytes, hipHostMallocDefault));
    return sf_reserve_bits(f, max_bits);
}

extern "C" int mfb_syncfinder_create(mfb_syncfinder **out, int device, const int8_t *tmpls, const int *T, const int *thresholds, int ntmpl,
                                     int max_bits, int max_hits) {
    if (!out) return MFB_ERR_ARG;
    *out = nullptr;
    if (!tmpls || !T || !thresholds || ntmpl < 1 || ntmpl > 16 || max_bits < 1 || max_hits < 1) return MFB_ERR_ARG;
    for (int t = 0; t < ntmpl; ++t)
        if (T[t] < 1 || T[t] > 4096) return MFB_ERR_ARG;
    int rc = sync_device_ok(device);
    if (rc) return rc;
    mfb_syncfinder *f = new mfb_syncfinder();
    f->device = device;
    f->K = ntmpl;
    f->max_hits = max_hits;
    f->T.assign(T, T + ntmpl);
    f->thr.assign(thresholds, thresholds + ntmpl);
    f->Tmax = 0;
    for (int t = 0; t < ntmpl; ++t) f->Tmax = T[t] > f->Tmax ? T[t] : f->Tmax;
    f->stream = nullptr;
    f->done = nullptr;
    f->d_tmpl = nullptr;
    f->d_bits = f->h_bits = nullptr;
    f->cap_bits = 0;
    f->d_res = f->h_res = nullptr;
    f->d_seg = nullptr;
    f->cap_seg = 0;
    f->busy = false;
    rc = sf_create_impl(f, tmpls, max_bits);
    if (rc) {
        (void)mfb_syncfinder_destroy(f);
        return rc;
    }
    *out = f;
    return MFB_OK;
}

extern "C" int mfb_syncfinder_begin(mfb_syncfinder *f, const uint8_t *bits, int L) {
    if (!f || !bits || L < 1) return MFB_ERR_ARG;
    if (f->busy) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(f->device));
    int rc = sf_reserve_bits(f, L);
    if (rc) return rc;
    memcpy(f->h_bits, bits, (size_t)L);
    HIPCHK(hipMemcpyAsync(f->d_bits, f->h_bits, (size_t)L, hipMemcpyHostToDevice, f->stream));
    const int K = f->K, mh = f->max_hits;
    int32_t *d_counts = f->d_res, *d_idx = f->d_res + K, *d_sc = d_idx + (size_t)K * mh;
    const int nseg = (L + f->Tmax - 1 + SYNC_SEG - 1) / SYNC_SEG;
    if (nseg <= 16) {
        const size_t lds = (size_t)f->Tmax + SYNC_SEG + f->Tmax - 1;
        hipLaunchKernelGGL(k_sync_small, dim3(K, 1), dim3(256), lds, f->stream, (const uint8_t *)f->d_bits, (const int8_t *)f->d_tmpl, f->sa, 1,
                           L, mh, d_counts, d_idx, d_sc);
    } else {
        const size_t need = (size_t)2 * nseg * sizeof(int);
        if (f->cap_seg < need) {
            HIPCHK(hipStreamSynchronize(f->stream));
            if (f->d_seg) HIPCHK(hipFree(f->d_seg));
            f->d_seg = nullptr;
            f->cap_seg = 0;
            HIPCHK(hipMalloc((void **)&f->d_seg, need));
            f->cap_seg = need;
        }
        int *segcnt = f->d_seg, *segoff = f->d_seg + nseg;
        for (int t = 0; t < K; ++t) {
            const int Tt = f->T[t];
            const int ns = (L + Tt - 1 + SYNC_SEG - 1) / SYNC_SEG;
            const size_t lds = (size_t)Tt + SYNC_SEG + Tt - 1;
            const int8_t *tp = f->d_tmpl + f->sa.toff[t];
            hipLaunchKernelGGL((k_sync_find<false>), dim3(ns, 1), dim3(256), lds, f->stream, (const uint8_t *)f->d_bits, tp, L, Tt, f->thr[t], ns,
                               segcnt, (const int *)nullptr, mh, (int32_t *)nullptr, (int32_t *)nullptr);
            hipLaunchKernelGGL(k_sync_scan, dim3(1), dim3(64), 0, f->stream, (const int *)segcnt, segoff, d_counts + t, 1, ns);
            hipLaunchKernelGGL((k_sync_find<true>), dim3(ns, 1), dim3(256), lds, f->stream, (const uint8_t *)f->d_bits, tp, L, Tt, f->thr[t], ns,
                               segcnt, (const int *)segoff, mh, d_idx + (size_t)t * mh, d_sc + (size_t)t * mh);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(f->h_res, f->d_res, (size_t)K * (1 + 2 * (size_t)mh) * sizeof(int32_t), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipEventRecord(f->done, f->stream));
    f->busy = true;
    return MFB_OK;
}

extern "C" int mfb_syncfinder_end(mfb_syncfinder *f, int32_t *counts, int32_t *hit_idx, int32_t *hit_score) {
    if (!f || !counts || !hit_idx || !hit_score) return MFB_ERR_ARG;
    if (!f->busy) return MFB_ERR_STATE;
    HIPCHK(hipSetDevice(f->device));
    HIPCHK(hipEventSynchronize(f->done));
    f->busy = false;
    const int K = f->K, mh = f->max_hits;
    const int32_t *r = f->h_res;
    memcpy(counts, r, (size_t)K * sizeof(int32_t));
    for (int t = 0; t < K; ++t) {
        const int n = r[t] < mh ? r[t] : mh;
        memcpy(hit_idx + (size_t)t * mh, r + K + (size_t)t * mh, (size_t)n * sizeof(int32_t));
        memcpy(hit_score + (size_t)t * mh, r + K + (size_t)K * mh + (size_t)t * mh, (size_t)n * sizeof(int32_t));
    }
    return MFB_OK;
}

// ---- packed sync correlation (sync_kernels.hpp, second half) -------------------------------------------------------------
// One page-locked buffer per device, grown on demand: callers that produce their packed bit streams straight into it
// (mfb_sync_pinned_buffer) get a true asynchronous host-to-device copy; any other host pointer works too (the runtime stages it).
static std::mutex g_pin_mu;
static uint8_t *g_pin[SYNC_MAX_DEVICES];
static size_t g_pin_cap[SYNC_MAX_DEVICES];

extern "C" int mfb_sync_pinned_buffer(int device, size_t bytes, void **host) {
    if (!host || bytes < 1) return MFB_ERR_ARG;
    int rc = sync_device_ok(device);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_cap[device] < bytes) {
        if (g_pin[device]) HIPCHK(hipHostFree(g_pin[device]));
        g_pin[device] = nullptr;
        g_pin_cap[device] = 0;
        HIPCHK(hipHostMalloc((void **)&g_pin[device], bytes, hipHostMallocDefault));
        g_pin_cap[device] = bytes;
    }
    *host = g_pin[device];
    return MFB_OK;
}

extern "C" int mfb_sync_find_packed(int device, const uint8_t *packed, int B, int L, int row_bytes, const int8_t *tmpl, int T,
                                    int threshold, int max_total, int32_t *hit_idx, int32_t *hit_score, int32_t *counts,
                                    int32_t *total_hits, float *device_ms) {
    if (!packed || !tmpl || !hit_idx || !hit_score || !counts || !total_hits || B < 1 || L < 1 || T < 1 || T > 64 * SYNCP_MAXK ||
        max_total < 1 || row_bytes < (L + 7) / 8)
        return MFB_ERR_ARG;
    const int K = (T + 63) / 64;
    std::vector<unsigned long long> masks(2 * (size_t)K, 0ull);
    for (int t = 0; t < T; ++t) {
        if (tmpl[t] == 1) masks[t / 64] |= 1ull << (t % 64);
        else if (tmpl[t] == -1) masks[K + t / 64] |= 1ull << (t % 64);
        else if (tmpl[t] != 0) return MFB_ERR_UNSUPPORTED;       // the packed form takes taps in {-1, 0, +1} only
    }
    int rc = sync_device_ok(device);
    if (rc) return rc;
    WsLease lease{device, ws_acquire(device)};
    if (!lease.w) return MFB_ERR_HIP;
    SyncWs &w = *lease.w;
    const int nseg = (L + T - 1 + SYNCP_SEG - 1) / SYNCP_SEG;
    const size_t nbytes = (size_t)B * row_bytes;
    if ((rc = ws_reserve(&w.bits, &w.cap_bits, nbytes))) return rc;
    if ((rc = ws_reserve(&w.tmpl, &w.cap_tmpl, masks.size() * sizeof(unsigned long long)))) return rc;
    if ((rc = ws_reserve(&w.seg, &w.cap_seg, ((size_t)2 * B * nseg + 2 * (size_t)B + 1) * sizeof(int)))) return rc;
    if ((rc = ws_reserve(&w.hits, &w.cap_hits, (size_t)2 * max_total * sizeof(int32_t)))) return rc;
    if ((rc = ws_reserve(&w.d_stage, &w.cap_dstage, (size_t)B * nseg * SYNCP_THREADS))) return rc;       // hits per thread, one byte
    int *segcnt = w.seg, *segoff = segcnt + (size_t)B * nseg, *d_counts = segoff + (size_t)B * nseg, *d_streamoff = d_counts + B;
    int32_t *d_idx = w.hits, *d_sc = w.hits + max_total;
    uint8_t *d_tcnt = w.d_stage;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (device_ms) {
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
    }
    HIPCHK(hipMemcpyAsync(w.bits, packed, nbytes, hipMemcpyHostToDevice, w.stream));
    HIPCHK(hipMemcpyAsync(w.tmpl, masks.data(), masks.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, w.stream));
    if (e0) HIPCHK(hipEventRecord(e0, w.stream));
    const unsigned long long *d_masks = (const unsigned long long *)w.tmpl;
#define SYNCP_LAUNCH(WRITE_, KT_)                                                                                                        \
    hipLaunchKernelGGL((k_sync_packed<WRITE_, KT_>), dim3(nseg, B), dim3(SYNCP_THREADS), 0, w.stream, (const uint8_t *)w.bits, row_bytes, L, T, \
                       K, d_masks, threshold, nseg, segcnt, d_tcnt, (const int *)segoff, (const int *)d_streamoff, max_total, d_idx, d_sc)
    if (K == 1) SYNCP_LAUNCH(false, 1);
    else if (K == 2) SYNCP_LAUNCH(false, 2);
    else SYNCP_LAUNCH(false, 0);
    hipLaunchKernelGGL(k_sync_scan, dim3((B + 63) / 64), dim3(64), 0, w.stream, (const int *)segcnt, segoff, d_counts, B, nseg);
    hipLaunchKernelGGL(k_sync_stream_scan, dim3(1), dim3(256), 0, w.stream, (const int *)d_counts, d_streamoff, B);
    if (K == 1) SYNCP_LAUNCH(true, 1);
    else if (K == 2) SYNCP_LAUNCH(true, 2);
    else SYNCP_LAUNCH(true, 0);
#undef SYNCP_LAUNCH
    HIPCHK(hipGetLastError());
    if (e1) HIPCHK(hipEventRecord(e1, w.stream));
    // counts and the grand total first, then exactly the hits
    std::vector<int> hc((size_t)2 * B + 1);
    HIPCHK(hipMemcpyAsync(hc.data(), d_counts, hc.size() * sizeof(int), hipMemcpyDeviceToHost, w.stream));
    HIPCHK(hipStreamSynchronize(w.stream));
    memcpy(counts, hc.data(), (size_t)B * sizeof(int));
    const int total = hc[(size_t)2 * B];
    *total_hits = total;
    const int n = total < max_total ? total : max_total;
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(hit_idx, d_idx, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, w.stream));
        HIPCHK(hipMemcpyAsync(hit_score, d_sc, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, w.stream));
        HIPCHK(hipStreamSynchronize(w.stream));
    }
    if (device_ms) {
        HIPCHK(hipEventElapsedTime(device_ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return MFB_OK;
}

// ---- N4: bit-stream alignment cross-correlation (reference lib/customXCorr.py:5-18) -----------------
// out = ifft( fft(a, N) * conj(fft(b, N)) ) with N the handle's block length: a and b are real
// sequences, truncated or zero-padded to N exactly as np.fft.fft(a, N) does.  Built from the handle's
// forward / inverse transform passes; the handle's spectrum, filter row 0 and matched-filter buffers
// serve as scratch, so this is meant for a handle of its own (M = 1), which is how the Python helper
// uses it.
__global__ void k_scale_c(cf *p, float s, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * s;
}

extern "C" int mfb_xcorr(mfb_ctx *c, const float *a, int Na, const float *b, int Nb, float *out_c64) {
    if (!c || !a || !b || !out_c64 || Na < 1 || Nb < 1) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const int N = c->N;
    auto stage = [&](const float *src, int n) -> int {   // real sequence -> d_env, zero-padded / truncated to N
        const int k = n < N ? n : N;
        HIPCHK(hipMemcpyAsync(c->d_env, src, (size_t)k * sizeof(float), hipMemcpyHostToDevice, c->stream));
        if (k < N) HIPCHK(hipMemsetAsync(c->d_env + k, 0, (size_t)(N - k) * sizeof(float), c->stream));
        return MFB_OK;
    };
    int rc;
    if ((rc = stage(a, Na))) return rc;
    if ((rc = before_fft(c))) return rc;                                        // d_X is scratch here
    if ((rc = forward_fft(c, nullptr, c->d_env, c->d_X, 1))) return rc;        // A = fft(a, N)
    HIPCHK(sync_streams(c));                                    // d_env is reused
    if ((rc = stage(b, Nb))) return rc;
    if ((rc = forward_fft(c, nullptr, c->d_env, c->d_masks, 0))) return rc;    // conj(fft(b, N)) into filter row 0
    // inverse transform of A * conj(B): the two-pass bank kernels at shift 0, one filter row
    P1Args p1 = p1_base(c);
    p1.X = c->d_X;
    p1.shifts = nullptr;
    p1.fixed_shift = 0;
    p1.dc = 1;
    p1.M = 1;
    p1.mpb = 1;
    p1.ntiles = c->N2 / TILE;
    p1.mgroups = 1;
    p1.jsplit = 1;
    if ((rc = launch_p1<KIND_BANK>(c, p1, dim3(p1.ntiles, 1, 1)))) return rc;
    P2Args p2 = p2_base(c);
    p2_store_split(c, p2, 1);
    if ((rc = launch_p2<MODE_STORE>(c, p2, dim3(p2.parts, 1, 1)))) return rc;
    if ((rc = launch_transpose(c, c->d_xc, 1, 0))) return rc;
    hipLaunchKernelGGL(k_scale_c, dim3(256), dim3(256), 0, c->stream, c->d_xc, 1.0f / (float)N, N);   // numpy's ifft is 1/N
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_c64, c->d_xc, (size_t)N * sizeof(cf), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_streams(c));
    c->have_filters = false;   // the scratch use clobbered the bank, the spectrum and the outputs
    c->have_input = false;
    c->have_xc = false;
    return MFB_OK;
}

#ifdef MFB_SEG_TRACE
extern "C" int mfb_debug_read_trace(unsigned long long *out, int nblocks) {
    if (!out || nblocks < 1 || nblocks > 65536) return MFB_ERR_ARG;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_seg_trace), (size_t)nblocks * 3 * sizeof(unsigned long long)));
    return MFB_OK;
}
#endif

extern "C" int mfb_timer_start(mfb_ctx *c) {
    if (!c) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->t0, c->stream));
    return MFB_OK;
}
extern "C" int mfb_timer_stop(mfb_ctx *c, float *ms) {
    if (!c || !ms) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->t1, c->stream));
    HIPCHK(hipEventSynchronize(c->t1));
    HIPCHK(hipEventElapsedTime(ms, c->t0, c->t1));
    return MFB_OK;
}
extern "C" int mfb_profile_enable(mfb_ctx *c, int on) {
    if (!c) return MFB_ERR_ARG;
    c->prof = on != 0;
    return MFB_OK;
}
extern "C" int mfb_profile_read(mfb_ctx *c, int counts[2], float total_ms[2]) {
    if (!c || !counts || !total_ms) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));
    for (int w = 0; w < 2; ++w) {
        counts[w] = (int)(c->ev[w].size() / 2);
        double tot = 0.0;
        for (size_t i = 0; i + 1 < c->ev[w].size(); i += 2) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, c->ev[w][i], c->ev[w][i + 1]));
            tot += ms;
        }
        total_ms[w] = (float)tot;
        for (auto e : c->ev[w]) c->ev_pool.push_back(e);
        c->ev[w].clear();
    }
    return MFB_OK;
}
extern "C" int mfb_sync(mfb_ctx *c) {
    if (!c) return MFB_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_streams(c));
    return MFB_OK;
}

// ---- host copy worker (include/mfbank.h: mfb_hostcopy_*; the worker itself: hostcopy.hpp, plain C++) ----------------------------
extern "C" int mfb_hostcopy_create(mfb_hostcopy **out) {
    if (!out) return MFB_ERR_ARG;
    *out = nullptr;
    mfb_hostcopy *q = new (std::nothrow) mfb_hostcopy();
    if (!q) return MFB_ERR_ALLOC;
    if (!q->start()) {
        delete q;
        return MFB_ERR_ALLOC;
    }
    *out = q;
    return MFB_OK;
}

extern "C" int mfb_hostcopy_submit(mfb_hostcopy *q, void *dst, const void *src, size_t bytes) {
    if (!q || (bytes && (!dst || !src))) return MFB_ERR_ARG;
    if (bytes && !q->submit(dst, src, bytes)) return MFB_ERR_ALLOC;        // (no exception crosses the C ABI)
    return MFB_OK;
}

extern "C" int mfb_hostcopy_drain(mfb_hostcopy *q) {
    if (!q) return MFB_ERR_ARG;
    q->drain();
    return MFB_OK;
}

extern "C" void mfb_hostcopy_destroy(mfb_hostcopy *q) {
    if (!q) return;
    q->shutdown();
    delete q;
}
