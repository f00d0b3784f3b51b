/*
 * mfbank.h -- C ABI of libmfbank.so: the MI355X (gfx950) Doppler matched-filter-bank hot path.
 *
 * This is the drop-in boundary for pyCuSDR's receive hot loop.  Every entry point replaces one
 * device interaction of the reference's host driver; the citation after each prototype names the
 * reference call sites it stands in for (paths relative to pyCuSDR/ in the reference tree,
 *   DB  = demodulator/demodulator_base.py,  CU = demodulator/cuda_kernels.cu,
 *   DEC = decoder.py, CUFFT = lib/cufft.py).
 *
 * Conventions
 *   - plain C: opaque handle, raw pointers, sizes; no C++/torch types.
 *   - complex64 is passed as interleaved float pairs (re, im), exactly numpy's layout.
 *   - every function returns an int status (MFB_OK == 0); mfb_strerror() names it.  The Python
 *     wrapper raises ValueError / TypeError / MemoryError / RuntimeError from these, mirroring
 *     DB:188-190, DB:252-257, DB:306-308 and the cuFFT status table CUFFT:90-116.
 *   - ownership: the caller owns every host array it passes in; the library owns all device
 *     buffers and the pinned input buffer for the lifetime of the handle.
 *   - threading: a handle is not thread-safe; one handle per process, like the reference's one
 *     CUDA context per Demodulator_process (DB:177-181).
 *   - synchronisation: calls that return values to the host synchronise the handle's stream at
 *     the same points the reference blocks on memcpy_dtoh (DB:605, DB:730, DB:1000-1006); the
 *     *_async calls only enqueue.
 */
#ifndef MFBANK_H
#define MFBANK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFB_OK               0
#define MFB_ERR_ARG          1   /* bad argument / shape      -> ValueError  (DB:188-190, 252-255) */
#define MFB_ERR_DTYPE        2   /* wrong element type        -> TypeError   (DB:256-257)          */
#define MFB_ERR_ALLOC        3   /* host/device allocation    -> MemoryError (CUFFT_ALLOC_FAILED)  */
#define MFB_ERR_HIP          4   /* HIP runtime / launch      -> RuntimeError(CUFFT_EXEC_FAILED)   */
#define MFB_ERR_STATE        5   /* filters/shifts/input unset-> RuntimeError(CUFFT_SETUP_FAILED)  */
#define MFB_ERR_UNSUPPORTED  6   /* size outside built plans  -> ValueError  (CUFFT_INVALID_SIZE)  */

typedef struct mfb_ctx mfb_ctx;

/* Operation selector of find_centres (CU:73-76, DB:31-34). */
#define MFB_CENTRES_ABS  0
#define MFB_CENTRES_REAL 1
#define MFB_CENTRES_IMAG 2

const char *mfb_strerror(int status);
/* Library/ABI version, bumped whenever a prototype changes.  No reference counterpart (its kernels are compiled from source
 * at run time, SourceModule DB:214). */
int mfb_abi_version(void);   /* 2: search paths, mfb_xcorr; 3: mfb_set_search_mode, mfb_sync_find_multi; 4: mfb_receive_block,
                              * mfb_export_rows_async, mfb_sync_find_packed; 5: mfb_debug_block_scalars; 6: mfb_receive_blocks_*,
                              * mfb_window_buffer, mfb_block_params.block_stride; 7: mfb_set_stream_stages, mfb_stream_seed,
                              * mfb_receive_blocks_end_record; 8: mfb_hostcopy_*; 9: mfb_set_batch_overlap, mfb_get_batch_scores, mfb_get_search_info, mfb_set_cu_share, mfb_receive_blocks_end_record
                              * reports the size it needs */

/* Create a handle on HIP device `device` for blocks of N = 2^log2N samples, `num_dopplers`
 * Doppler bins plus `doppler_offset` leading noise-reference bins (DB:150-159), M matched
 * filters, findCentres window W (odd), and the two compile-time switches the reference bakes
 * into its kernel header (SUM_ALL_MASKS, CODE_SEARCH_MASK_OFFSET; DB:398-418).
 * Replaces: cuda.Device(...).make_context() DB:177-181, buffer allocation DB:433-476,479-514,
 * FFT plan creation DB:275-338,501 and SourceModule JIT DB:214. */
int mfb_create(mfb_ctx **out, int device, int log2N, int num_dopplers, int doppler_offset,
               int M, int window_width, int sum_all_masks, int code_search_mask_offset);
/* Replaces Demodulator.__del__ (DB:517-530): frees buffers, plans, stream. */
int mfb_destroy(mfb_ctx *ctx);

/* Run all device work of this handle on an existing HIP stream (hipStream_t passed as void*),
 * e.g. torch's current stream so that RCCL collectives order naturally.  NULL restores the
 * handle's own stream.  (The reference's analogue is cufftSetStream, CUFFT:365-375.) */
int mfb_set_stream(mfb_ctx *ctx, void *hip_stream);
/* Tuning knobs (0 keeps the current value): Doppler bins per launch of the two FFT passes (bounds
 * the intermediate buffer; the reference's analogue is CUDA.batchSize, DB:301-313), matched
 * filters handled per workgroup in pass 1, FFT rows per workgroup in pass 2, and the number of
 * workgroups that share the Doppler bins of one (tile, filter group) in pass 1.  All reductions are
 * fixed-order (no float atomics), so results are bit-reproducible run to run; doppler_chunk,
 * masks_per_block and jsplit never change a result bit, rows_per_block regroups the fp32 partial
 * sums (results then agree to rounding, ~1e-7). */
int mfb_set_tuning(mfb_ctx *ctx, int doppler_chunk, int masks_per_block, int rows_per_block, int jsplit);
int mfb_get_tuning(mfb_ctx *ctx, int *doppler_chunk, int *masks_per_block, int *rows_per_block, int *jsplit);

/* Search path.  The Doppler search and the matched filtering at the chosen shift have two
 * implementations with identical results (within fp32 rounding, ~1e-7):
 *   MFB_PATH_SEGMENT  single-pass overlap-save: mfb_set_filters measures the impulse-response support
 *                     T of the bank (every shipped protocol: 48...640 taps); for T <= L/2 the block is cut
 *                     into L-point segments (L = 2^log2L, 256...8192), each mixed to the Doppler shift in
 *                     time, transformed, multiplied by the L-point filter spectra, transformed back and
 *                     reduced -- in registers and LDS, with no length-N intermediate in HBM.
 *   MFB_PATH_TWOPASS  length-N two-pass transforms through an HBM intermediate (any filter).
 * MFB_PATH_AUTO (default) takes the segment path whenever the bank allows it.  log2L = 0 chooses L by
 * a cost model; wg_per_cu (<= 64; workgroups in the grid per CU, default 32) / filters_per_pass = 0 keep the defaults.  Requesting MFB_PATH_SEGMENT for a bank
 * without short support returns MFB_ERR_UNSUPPORTED and leaves the previous setting in force.
 * (The reference's knobs of this kind are CUDA.batchSize / CUDA.streams, DB:171-178, 301-338.) */
#define MFB_PATH_AUTO    0
#define MFB_PATH_TWOPASS 1
#define MFB_PATH_SEGMENT 2
int mfb_set_search_path(mfb_ctx *ctx, int path, int log2L, int wg_per_cu, int filters_per_pass);
/* What is in force: path (TWOPASS or SEGMENT), log2L, taps T of the bank (N if it has no short
 * support), valid outputs per segment and number of segments (0 on the two-pass path).  Any pointer
 * may be NULL.  No reference counterpart (one formulation only: DB:571-591). */
int mfb_get_search_path(mfb_ctx *ctx, int *path, int *log2L, int *taps, int *valid_per_segment, int *segments);
/* Search basis (opt-in; segment path with SUM_ALL_MASKS only).  With SUM_ALL_MASKS the search needs
 * sum_m |y_m[n]|^2 only, which is invariant under any unitary mixing of the filters; the shipped banks (all 2^k
 * bit patterns of a smooth modulation) span far fewer dimensions than they have filters (GMSK 6 of 8, FSK-2 and
 * CC11xx 4 of 8, BPSK 5 of 32; directions below 1e-10 of the energy are dropped).  MFB_BASIS_SPAN makes the search transform an orthogonalised basis F of that span
 * with F F^H = C C^H (C = the taps): the same doppSum to fp32 rounding from rank(C) inverse transforms per
 * segment instead of M.  It is an algorithmic shortcut, so it is OFF by default (MFB_BASIS_FILTERS: every unique
 * filter is transformed, as the reference does, DB:578-588) and bench.py reports it as a separate figure.
 * The demodulation stage always uses the M filters.  Returns MFB_ERR_STATE on a handle created without
 * sum_all_masks (per-filter sums need every filter); on the two-pass path the request stays pending. */
#define MFB_BASIS_FILTERS 0
#define MFB_BASIS_SPAN    1
int mfb_set_search_basis(mfb_ctx *ctx, int basis);
/* Basis in force and the number of filters the search transforms per Doppler bin.  No reference counterpart (it transforms
 * all M, DB:578-588). */
int mfb_get_search_basis(mfb_ctx *ctx, int *basis, int *transformed_filters);
/* Host-only helper: dimension of the span of a bank's impulse responses (M if it has no short support).  No reference
 * counterpart: the reference transforms every length-N filter row (plans DB:292-338). */
int mfb_analyze_rank(const float *masks_c64, int M, int N, int *rank);
/* Search mode (opt-in shortcut).  The search uses only the row sums of |y|^2 over all N outputs of each circular
 * correlation (cuda_kernels.cu:421-480 after DB:578-588), and by Parseval's identity
 *     doppSum[j][m] = N/2^18 . sum_k |X[(k + shift_j) mod N]|^2 . |H_m[k]|^2
 * so MFB_SEARCH_ENERGY computes the table from the power spectrum of the block and the filters' energy spectrum
 * (their sum over m under SUM_ALL_MASKS): D.N multiply-adds, no inverse transform.  Same table to fp32 rounding, any
 * filter bank, either search path for the demodulation stage.  Because it no longer runs the matched-filter bank it
 * is OFF by default (MFB_SEARCH_TRANSFORMS) and bench.py reports it as a separate figure, never as the headline. */
#define MFB_SEARCH_TRANSFORMS 0
#define MFB_SEARCH_ENERGY     1
int mfb_set_search_mode(mfb_ctx *ctx, int mode);
int mfb_get_search_mode(mfb_ctx *ctx, int *mode);
/* Fault injection for tests: the nth device allocation made on behalf of a handle by the calling thread
 * from now on fails as if the device were out of memory (0 disarms).  Lets a test walk the free-on-error
 * path of mfb_create allocation by allocation.  nth < 0: the |nth|-th one raises std::bad_alloc inside the library instead -- what a
 * failed HOST allocation in the filter analysis or the table builders looks like; no C++ exception leaves the library: mfb_create,
 * mfb_set_filters, mfb_set_shifts, mfb_set_search_* and the mfb_analyze_* helpers return MFB_ERR_ALLOC for it, the handle (if any)
 * stays destroyable and takes a new mfb_set_filters.  No reference counterpart (teardown on error: __del__ DB:517-530). */
int mfb_debug_fail_alloc(int nth);
/* Host-only helper (no device work): common circular support window [start, start+len) of the impulse
 * responses ifft(H_m) of a filter bank complex64 [M][N]; len == N when some filter has no short support.
 * This is the analysis mfb_set_filters runs; exported so it can be checked without a GPU.  No reference counterpart: the
 * reference keeps the filters as length-N spectra only (__uploadMaskToGPU DB:246-263). */
int mfb_analyze_filters(const float *masks_c64, int M, int N, int *support_start, int *support_len);

/* Geometry the handle settled on: the two FFT factors N = N1 * N2 and, after mfb_set_filters, the number
 * of filter rows the Doppler search actually transforms (filters that are exact copies or exact
 * negatives of an earlier one are transformed once).  Any pointer may be NULL.  No reference counterpart (its cuFFT plans,
 * DB:292-338, keep their factorisation to themselves). */
int mfb_get_info(mfb_ctx *ctx, int *N1, int *N2, int *unique_filters);

/* Upload the filter bank: host complex64 [M][N], row-major, already conj(fft(template, N)) as
 * protocol.get_filter returns it.  Replaces __uploadMaskToGPU (DB:246-263).  `M`/`N` are what the
 * caller believes the shape is; a mismatch with the handle returns MFB_ERR_ARG. */
int mfb_set_filters(mfb_ctx *ctx, const float *masks_c64, int M, int N);
/* Upload the Doppler shift table int32[count] (count must equal doppler_offset + num_dopplers);
 * every entry must already be wrapped into [0, N).  Replaces memcpy_htod DB:221. */
int mfb_set_shifts(mfb_ctx *ctx, const int32_t *shifts, int count);

/* Page-locked host buffer of N complex64 owned by the handle; the caller fills it in place
 * (overlap carry included).  Replaces pagelocked_empty(...DEVICEMAP) DB:456-457 /
 * get_signalBufferHostPointer DB:1055-1060. */
int mfb_input_buffer(mfb_ctx *ctx, float **host_c64);
/* Copy the pinned buffer to the device and run the forward FFT (unnormalised, sign -1).  MFB_ERR_STATE on a handle without the
 * transforms' intermediate (an mfb_set_filters that failed while re-sizing it: set the filters again).  Replaces uploadToGPU
 * DB:548-558 (cufftExecC2C FORWARD). */
int mfb_upload(mfb_ctx *ctx);
/* Same, from an arbitrary host array of N complex64 (pageable is fine; staged through the pinned
 * buffer): the copy into the page-locked buffer that the reference's caller does itself (DB:555-556 in comments,
 * Demodulator_process DP:256,287) followed by uploadToGPU DB:548-558. */
int mfb_upload_from(mfb_ctx *ctx, const float *host_c64, int N);
/* Same, from N complex64 already resident in device memory (no copy): the forward transform of uploadToGPU DB:557-558 on a
 * caller-owned device buffer.  Used by bench.py so the timed region starts with inputs in HBM, and by the sharded path. */
int mfb_upload_device(mfb_ctx *ctx, const void *dev_c64);

/* Enqueue the Doppler search (shift-multiply, inverse FFT bank, |.|^2 row sums) for this handle's
 * bins; leaves doppSum float32 [count][M] on the device.  Replaces setArrayToZeros +
 * multInputVectorWithShiftedMasksDopp + batched cufftExecC2C INVERSE + blockAbsSumAtomic
 * (DB:571-591; CU:853-857, 339-373, 421-480). */
int mfb_search_async(mfb_ctx *ctx);
/* Copy this handle's doppSum rows into rows [row_offset, row_offset+count) of a device array
 * float32 [*][M] (e.g. the buffer that is then all-reduced across ranks).  Asynchronous.  The table is the reference's
 * GPU_bufDoppSum (filled by blockAbsSumAtomic CU:421-480, read by findDopplerEst CU:502-597); the reference never moves it. */
int mfb_export_scores_async(mfb_ctx *ctx, void *dev_dst, int row_offset);
/* Doppler pick on `dev_scores` float32 [offset+num][M] (NULL = the handle's own doppSum):
 * top-2 weighted index and metric; synchronises and returns res = {idx, metric}.
 * Replaces findDopplerEst + memcpy_dtoh (DB:601-605; CU:502-597). */
int mfb_pick(mfb_ctx *ctx, const void *dev_scores, int num, int offset, float res[2]);
/* The same two steps for the SUM_ALL_MASKS case, in which only column 0 of doppSum is populated
 * (CU:453-464): copy that column into dev_dst float32[*] at [row_offset, row_offset+count), and pick on a
 * device vector float32[offset+num] -- the sharded exchange then moves D floats instead of D*M.
 * mfb_pick_column returns MFB_ERR_STATE on a handle created without sum_all_masks. */
int mfb_export_column_async(mfb_ctx *ctx, void *dev_dst, int row_offset);
/* Row range of either form: rows [first_row, first_row + nrows) of this handle's doppSum -- whole rows of M floats
 * (column_only = 0) or column 0 alone (column_only = 1) -- to rows [dst_row, dst_row + nrows) of dev_dst.
 * Asynchronous.  With a noise-reference bin (doppler_offset > 0; DB:148-159, CU:550-554) a sharded caller moves
 * its bin slice (first_row = doppler_offset) and the replicated noise row (first_row = 0) separately. */
int mfb_export_rows_async(mfb_ctx *ctx, void *dev_dst, int dst_row, int first_row, int nrows, int column_only);
int mfb_pick_column(mfb_ctx *ctx, const void *dev_column, int num, int offset, float res[2]);
/* mfb_search_async + mfb_pick on the handle's own bins: the device part of __findUHF
 * (DB:567-605). */
int mfb_find_carrier(mfb_ctx *ctx, float res[2]);
/* Copy doppSum float32 [count][M] to the host (the reference reads it back only under
 * STORE_BITS_IN_FILE, DB:593-599); synchronises. */
int mfb_get_scores(mfb_ctx *ctx, float *host_scores);
/* Read `count` complex64 spectrum bins starting at `start` (wrapping modulo N), for the host-side
 * SNR estimate.  Replaces the zero-copy reads of GPU_bufSignalFreq_cpu_handle in computeSNR
 * (DB:651-661); synchronises. */
int mfb_get_spectrum(mfb_ctx *ctx, float *host_c64, int start, int count);

/* Matched filters at one shift + symbol-rate/phase estimate.  Runs
 *   multInputVectorWithShiftedMask (CU:174-185; DB:776-781), cufftExecC2C INVERSE batch M
 *   (DB:785), sumXCorrBuffMasks (CU:191-205; DB:717-718), cufftExecR2C (DB:721),
 *   findCodeRateAndPhase (CU:236-320; DB:725-726) over k in [k_offset, k_offset+k_len),
 * then synchronises and returns res = {k*, arg P[k*], |P[k*]|^2} (DB:730). */
int mfb_demodulate(mfb_ctx *ctx, int shift, int k_offset, int k_len, float res[3]);
/* One call per block: everything the receive loop does with a block on the device, as one stream of launches and ONE
 * synchronisation -- uploadToGPU (DB:548-558), __findUHF's search and pick (DB:567-605), its shift interpolation (DB:609-616,
 * float64, on the device), the spectrum windows computeSNR reads (DB:635-667), the matched filters at that shift, envelope,
 * R2C, findCodeRateAndPhase (DB:711-730), the float64 rate/phase arithmetic (DB:733-752), the clamp and symbol count of
 * cudaFindCentres (DB:994-999), findCentres and its three read-backs (DB:996-1006).  Results are bit-identical to the
 * sequence mfb_upload, mfb_find_carrier, mfb_get_spectrum, mfb_demodulate, mfb_find_centres with the reference's host
 * arithmetic in between (tests/test_gpu_block.py). */
enum { MFB_BLOCK_SEARCH = 0, MFB_BLOCK_FIXED_SHIFT = 1 };            /* UHF: Doppler search; STX: shift = IF offset (STX.py:21-24) */
enum { MFB_INPUT_PINNED = 0, MFB_INPUT_DEVICE = 1, MFB_INPUT_UPLOADED = 2, MFB_INPUT_PINNED2 = 3,
       MFB_INPUT_WINDOW = 4, MFB_INPUT_WINDOW2 = 5 };    /* batches: the two page-locked sample windows of mfb_window_buffer */
typedef struct mfb_block_params {
    int32_t mode;            /* MFB_BLOCK_* */
    int32_t input;           /* MFB_INPUT_PINNED(2): H2D of the (second) pinned input buffer first; _DEVICE: N complex64 at device_block;
                              * _UPLOADED: the block was uploaded by an earlier mfb_upload* call */
    const void *device_block;
    int32_t fixed_shift;     /* MFB_BLOCK_FIXED_SHIFT */
    int32_t k_offset, k_len; /* symbol-rate search window (DB:508-512) */
    int32_t spsym_min;       /* clamp of cudaFindCentres (DB:994-995) */
    int32_t op;              /* 0 |.|^2, 1 |re|, 2 |im| (DB:28-31) */
    int32_t snr_window;      /* computeSNR's windowWidth (DB:618: 5) */
    int32_t max_symbols;     /* capacity of sym / cen / mag */
    int32_t band_capacity;   /* complex64 elements per SNR window in bands_c64 [2][band_capacity]; 0 = no windows */
    int32_t block_stride;    /* mfb_receive_blocks_* only: samples from the start of a block to the start of the next (N - overlap);
                              * 0 with MFB_INPUT_WINDOW(2) = what the window was made for */
} mfb_block_params;
typedef struct mfb_block_result {
    float pick[2];           /* {index, metric} of findDopplerEst */
    int32_t pick_valid;      /* 0: NaN index, the block is to be skipped (DB:625-630); shift = 0 was demodulated */
    int32_t shift;           /* dopplerIdxlast */
    int32_t low, high;       /* int(index), ceil(index) */
    double frac;             /* index % 1 */
    float cr[3];             /* {k*, arg P[k*], |P[k*]|^2} */
    double spSym, codeOffset;/* DB:733-752 */
    int32_t count;           /* symbols written to sym / cen / mag */
    int32_t rate_fallback;   /* k* == 0: spSym = 10 (DB:737-740) */
    int32_t band_len[2];     /* elements of the signal / noise window; > band_capacity: not delivered, fetch with mfb_get_spectrum */
} mfb_block_result;
/* One block, one call: uploadAndFindCarrier (UHF.py:7-16 -> DB:548-632) + demodulate (DB:711-859, device stages). */
int mfb_receive_block(mfb_ctx *ctx, const mfb_block_params *params, mfb_block_result *result, int32_t *sym, int32_t *centres,
                      float *magnitude, float *bands_c64);
/* The same in two halves, so that the caller's sequential host stages of block i-1 (and the assembly of block i+1 in the
 * other input buffer) run while the device works on block i: _begin enqueues everything including the read-back into
 * page-locked staging and returns at once; _end waits for that block and hands its results out.  Two blocks may be in
 * flight (slot 0 / 1); they execute in the order they were begun.  mfb_input_buffer2 is the second page-locked input
 * buffer (MFB_INPUT_PINNED2).  The reference has no counterpart: its loop is strictly one block at a time (DP:284-338). */
int mfb_receive_block_begin(mfb_ctx *ctx, const mfb_block_params *params, int slot);
int mfb_receive_block_end(mfb_ctx *ctx, int slot, mfb_block_result *result, int32_t *sym, int32_t *centres, float *magnitude,
                          float *bands_c64);
int mfb_input_buffer2(mfb_ctx *ctx, float **host_c64);
/* B consecutive blocks per call.  The reference's own configurations run blocks of 2^15 ... 2^17 samples over 64 bins
 * (config/base.json:13,33; config/benchmark/bench_base.json:26; config/CC11xx.json:50) and hand the device ONE block per turn of
 * the receive loop (Demodulator_process.run, DP:284-338: popBlock, uploadAndFindCarrier, demodulate, send): a few tens of
 * microseconds of device work per turn.  These calls take `nblocks` consecutive blocks of the stream at once, as ONE contiguous
 * window of nblocks * block_stride + (N - block_stride) complex64 samples (block_stride = N - overlap; block b starts at sample
 * b * block_stride, so neighbouring blocks share their overlap samples as storage -- nothing is copied twice and the caller
 * carries the overlap once per window instead of once per block, DP:256,287,337), and run them through one set of launches:
 * batched forward FFT, ONE search launch over nblocks x bins streams, nblocks picks, ONE matched-filter launch at the nblocks
 * shifts, batched envelope FFT, rate estimates, symbol centres, ONE device-to-host copy of nblocks result records.  Every
 * number of every block equals what mfb_receive_block returns for that block alone, bit for bit (tests/test_gpu_blocks.py).
 *   mfb_window_buffer     page-locked window `which` (0 / 1) for up to max_blocks blocks (owned by the handle, zero-filled,
 *                         with a device copy of its own); asking for another geometry re-allocates both windows.
 *   _begin                params as for mfb_receive_block with input = MFB_INPUT_WINDOW / _WINDOW2 (or MFB_INPUT_DEVICE:
 *                         device_block = the window in device memory, block_stride required); enqueues and returns.  Two
 *                         batches may be in flight (slot 0 / 1, shared with mfb_receive_block_begin).  Segment search path and
 *                         MFB_SEARCH_TRANSFORMS only (else MFB_ERR_UNSUPPORTED: use the one-block calls).
 *   _end                  waits; results[nblocks]; block b's symbols / centres / magnitudes at sym + b * symbol_stride (...),
 *                         its two SNR windows at bands_c64 + b * 4 * band_capacity floats. */
int mfb_window_buffer(mfb_ctx *ctx, int which, int max_blocks, int block_stride, float **host_c64);
int mfb_receive_blocks_begin(mfb_ctx *ctx, const mfb_block_params *params, int nblocks, int slot);
/* Several demodulator instances on ONE device (the reference starts one process and one context per radio, possibly on the same
 * GPU: pyCuSDR.py:245-251, DB:177-181): give this handle's launches part `part` of `parts` equal parts of the compute units (CU i
 * belongs to part i % parts; parts = 1: the whole device again).  Without it two instances share the device without loss but not
 * evenly -- workgroups of the long-filter kernel (76 KiB of LDS) find no room beside the short-filter kernel's (52 KiB): 16 % / 85 % of
 * their stand-alone rates at BASELINE C5 --; with (0, 2) and (1, 2) each runs on its own half, whatever the other does
 * (bench.py: c5_*_shared).  The handle's own stream is re-created with a CU mask (a stream set with mfb_set_stream is the caller's
 * business); nothing may be in flight. */
int mfb_set_cu_share(mfb_ctx *ctx, int part, int parts);
/* How the segment search of the next block will run (no reference counterpart; bench.py's flop count reads it): filter_side = 1 when
 * the Doppler shift sits on the FILTERS' side -- the segment of samples is transformed once for `bins_per_forward` neighbouring bins
 * and every (bin, filter) brings segment spectra of its own, built from the shift table (DB:130-165) when shifts or filters change:
 * 256-point segments always, wave-local 2048-point segments for up to 8 filters and 256 MiB of spectra --, 0 when every (bin, segment)
 * is mixed in time and transformed.  Either way the table is the reference's |IFFT(X[(k + s) mod N] H_m[k])|^2 sums (CU:339-373,
 * 421-480) to fp32 rounding.  MFB_SEG_FSM=0 in the environment keeps the time-side form. */
int mfb_get_search_info(mfb_ctx *ctx, int *filter_side, int *bins_per_forward);
/* The doppSum table (num_dopplers + doppler_offset rows of M floats, as mfb_get_scores: GPU_bufDoppSum, DB:594-601) of block
 * `block` of the batch begun LAST, once that batch has been collected and before the next one is begun.  MFB_ERR_STATE otherwise
 * (no batch, a fixed-shift batch, block beyond it). */
int mfb_get_batch_scores(mfb_ctx *ctx, int block, float *host_scores);
/* A batch as two parts on two streams (no reference counterpart: its loop waits for every stage of a block, DP:284-338).  Part 1 --
 * forward transforms, Doppler search, pick: the launches that fill the chip -- stays on the handle's stream; part 2 -- matched
 * filters at the picked shift, envelope, its spectrum, rate / phase, centres, the integer stages, the read-back: a chain of a dozen
 * small launches -- goes to a second stream (highest priority, transforms through an intermediate of its own, the two flights'
 * records in buffers of their own), so that the NEXT batch's part 1 runs beside this batch's part 2.  Same numbers, bit for bit
 * (the kernels and their order inside a batch do not change).  Worth +12 ... 20 % to a caller that keeps two batches in flight and is
 * not bound by its own per-block work (1.72 -> 1.92 ... 2.0 Gsamples/s at 2^15 x 64 bins x 32 blocks, 2^17 x 8: tools/batch_device_rate.py,
 * profiles/r06_chain.md); it costs a caller that waits for batch k - 1 right after it has begun batch k and is bound by its own
 * work -- the Python receive loop: -7 % -- because batch k - 1's part 2 then shares the chip with batch k's search and finishes later.
 * The one-block calls mfb_receive_block_begin / _end split the same way when it is on (input = a page-locked buffer or a device block):
 * at C2 the next block's search then runs beside this block's 0.16 ms of demodulation stage, 1.578 -> 1.518 ms per block with two
 * blocks in flight (tools/block_device_rate.py).  Off by default (0); the environment's MFB_BATCH_SPLIT=0/1 overrides every handle. */
int mfb_set_batch_overlap(mfb_ctx *ctx, int on);
int mfb_receive_blocks_end(mfb_ctx *ctx, int slot, mfb_block_result *results, int32_t *sym, int32_t *centres, float *magnitude,
                           int symbol_stride, float *bands_c64);
/* The integer stages behind the symbol decisions, on the device, for the blocks of a batch (mfb_receive_blocks_*; UHF search
 * mode).  Replaces, block by block and bit for bit:
 *   A12  extractBits (DB:1012-1023: bits = bitLUT[sym]) -- lut_mode 1, lut = uint8[lut_rows] of 0 / 1 -- or extractBitsNRZs
 *        (DB:1026-1051) -- lut_mode 2, lut = int32[lut_rows][2][lut_successors] = symbolLUT[sym][is-one | is-zero][successors],
 *        impossible transitions -> 0 and counted (lut_rows <= 256 resp. lut_rows * 2 * lut_successors <= 2048: MFB_ERR_ARG beyond);
 *   A13  checkSymbolOverlap (DB:863-988): the symbols whose centre lies in [overlap_samples / 2, N - overlap_samples / 2], the
 *        +-1 repair against the previous block over overlap_offset symbols with match_threshold (DB:97-99, 938-957), skipped
 *        above error_threshold impossible transitions (DB:925), and the uint8 casts of DB:859 (centres mod 256; trust = the raw
 *        bytes of the leading fp32 magnitudes, DB:472,1005-1006);
 *   A14  the decoder's two searches np.where(np.convolve(stream, template) >= threshold) (DEC:96-113) on the stream it stitches
 *        when no candidate is stashed -- the last bits_overlap bits before the block + the block's bits (DEC:89-90) -- for up to
 *        two templates of taps in {-1, 0, +1} (header mask, sync flag), thresholds numOnes - tolerance (DEC:101,113).
 * A block whose case is irregular (a symbol index outside the LUT, no first / last centre, a window or a previous tail too
 * short for the slices numpy would take) is flagged (a13_status 0) and left to the host path, which does whatever the
 * reference does there; the sync hits are flagged invalid from such a block to the end of the batch (and until the caller
 * seeds the state again).  The state a batch starts from -- the previous block's tail (bits behind its window: poswinP, last
 * overlap_offset + 1 bits inside it: posSymEnd, DB:977-979) and the last bits_overlap bits of the stream -- stays on the device
 * from batch to batch; mfb_stream_seed sets it from the host's (start of a stream, after an irregular block, after blocks
 * that went another way).  Batches of more than 64 blocks run without the stages (their records carry layout.stream_stages = 0: the
 * host does A12 ... A14 for them): one workgroup chains the kept-bit counts of a batch's blocks through one wave, 64 lanes.  So do
 * batches at a FIXED shift (MFB_BLOCK_FIXED_SHIFT, the S-band back end demodulator/STX.py:8-24): that back end clips interference
 * peaks out of every block on the host before the block is transformed and tags the symbols next to them in the trust bytes
 * afterwards (DB:670-707, 830-837) -- from sample indices only the host holds, block by block, so the bits of such a block are
 * finished on the host anyway.  p == NULL switches the stages off. */
typedef struct mfb_stream_params {
    int32_t overlap_samples;     /* 2^overlap (config GPU.overlap) */
    int32_t overlap_offset;      /* symbol_check_overlap_offset (DB:19-26): 20 */
    int32_t match_threshold;     /* overlap_offset - symbol_check_match_num_errors_allowed */
    int32_t error_threshold;     /* symbol_check_error_threshold */
    int32_t lut_mode, lut_rows, lut_successors;
    const void *lut;
    int32_t num_templates;       /* 0: no sync search */
    int32_t bits_overlap;        /* protocol.numBitsOverlap */
    int32_t template_taps[2], template_thresholds[2];
    const int8_t *templates;     /* taps back to back, as np.convolve takes them (the flipped +-1 masks) */
} mfb_stream_params;
/* A12 / A13 / A14 on the device (DB:1012-1051, DB:863-988, DEC:89-113: the block comment above); mfb_stream_seed hands over what
 * the host keeps between blocks: poswinP / posSymEnd (DB:977-979) and the tail of bitsOverlapBuf (DEC:90). */
int mfb_set_stream_stages(mfb_ctx *ctx, const mfb_stream_params *params);
int mfb_stream_seed(mfb_ctx *ctx, const uint8_t *post, int npost, const uint8_t *end, int nend, const uint8_t *ring, int ring_len);
/* A finished batch as it came off the device: nblocks records of record_bytes each copied into dst (capacity bytes), and where
 * things are inside a record -- the scalars (struct BlockScalars of csrc/small_kernels.hpp, scalars_bytes long, at offset 0),
 * the two SNR windows, int32 symbols / centres and float32 magnitudes (`symbols` entries each; `count` of them valid), and
 * with the stream stages on: kept bits / centres mod 256 / trust bytes (uint8, a13_nwin valid), the block's tail (post: a13_npost
 * bytes, end: a13_nend), the sync hits (per template: int32 idx[max_hits] | score[max_hits]; sync_count valid).  One copy per
 * batch instead of one call per array and block: what mfb_receive_blocks_end hands out, read in place (DB:1000-1006 reads
 * three arrays per block with three memcpy_dtoh).  capacity < nblocks * record_bytes: MFB_ERR_ARG with layout->nblocks and
 * layout->record_bytes filled in (everything else zero) and the batch STILL in flight -- call again with a buffer of that size. */
typedef struct mfb_record_layout {
    int32_t nblocks, scalars_bytes, symbols, band_capacity, mode, fixed_shift, stream_stages, max_hits, templates, reserved;
    int32_t edge_candidates, edge_hits;   /* per block: the leading T - 1 positions of the stream a FIXED-mode decoder would restart
                                           * at for each of the first header hits (DEC:254-263): int32 {a_rel, valid, n[2],
                                           * idx[2][edge_hits], score[2][edge_hits]} each, at off_edges */
    int64_t record_bytes, off_bands, off_sym, off_cen, off_mag, off_bits, off_centres_u8, off_trust, off_post, off_end, off_hits,
            off_edges;
} mfb_record_layout;
int mfb_receive_blocks_end_record(mfb_ctx *ctx, int slot, void *dst, size_t capacity, mfb_record_layout *layout);
/* Test seam of the stream stages: the same kernels on INJECTED symbol decisions -- counts[nb], and per block `symbols` entries of
 * symbol index / centre / magnitude, as findCentres writes them (DB:996-1006) -- in front of the state the last mfb_stream_seed (or
 * batch) left; the records come back as mfb_receive_blocks_end_record delivers them.  Lets a test drive the alignment through
 * planted +-1 slips, impossible transitions and irregular blocks and compare with the host's checkSymbolOverlap / extractBits*
 * (DB:863-1051) bit for bit (tests/test_gpu_stream_stages.py).  Advances the device-side state like a batch. */
int mfb_debug_stream_stages(mfb_ctx *ctx, int nb, int symbols, const int32_t *counts, const int32_t *sym, const int32_t *centres,
                            const float *magnitude, void *dst, size_t capacity, mfb_record_layout *layout);
/* Test seam of the one-call path.  mfb_receive_block moved two pieces of the reference's float64 HOST arithmetic onto the
 * device: the shift interpolation and the bounds of computeSNR's spectrum windows behind the pick (DB:609-620, 635-667), and
 * samples per symbol / code phase / clamp / symbol count behind the rate argmax (DB:733-752, 994-999).  This call runs exactly
 * those two device stages (the same device functions, one thread per item) on n INJECTED device results instead of the ones
 * the search and the argmax produced -- picks float[n][2] = {index, metric} as findDopplerEst writes them, triples float[n][3]
 * = {k*, arg P[k*], |P[k*]|^2} as findCodeRateAndPhase writes them -- with the handle's shift table and block length.  That is
 * the point at which the reference's own code can be fed the same values (a recording fake of its memcpy_dtoh,
 * tests/golden/make_golden_host.py), so reference-run fixtures gate this arithmetic bit for bit (tests/test_gpu_pins_host.py).
 * results[n]: as mfb_receive_block fills them (no symbols are decided); launch_args float[n][2] (optional): the two float32
 * values findCentres is launched with (DB:997); band_pieces int32[n][2][2][2] (optional): [signal | noise][piece][start, length]
 * of the spectrum windows.  No effect on the handle's state. */
int mfb_debug_block_scalars(mfb_ctx *ctx, int n, const float *picks, const float *triples, int spsym_min, int snr_window,
                            int max_symbols, mfb_block_result *results, float *launch_args, int32_t *band_pieces);

/* Symbol centres on the matched-filter outputs left by mfb_demodulate.  Replaces findCentres
 * (CU:78-146) + the three memcpy_dtoh of cudaFindCentres (DB:996-1006).  Writes `count` =
 * int(N/spSym) entries (must be <= capacity) of symbol index, centre sample and fp32 magnitude. */
int mfb_find_centres(mfb_ctx *ctx, float spSym, float offset, int op, int count,
                     int32_t *host_sym, int32_t *host_centre, float *host_mag);
/* Copy the matched-filter outputs complex64 [M][N] to the host (debug/parity; the reference does
 * this only under STORE_BITS_IN_FILE, DB:849-850). */
int mfb_get_xcorr(mfb_ctx *ctx, float *host_c64);
/* Copy the symbol-energy envelope float32 [N] (output of sumXCorrBuffMasks, CU:191-205; launched at DB:717-718) to the host. */
int mfb_get_envelope(mfb_ctx *ctx, float *host_f32);

/* Batched sync/preamble correlation: for each of B bit streams (uint8 0/1, length L, row-major
 * [B][L]) the full convolution with an integer template int8[T]:
 *   score[b][i] = sum_t tmpl[t] * bits[b][i - t],   i in [0, L+T-1)
 * exactly what np.convolve(bits, tmpl) returns at DEC:96 and DEC:112, as int32 [B][L+T-1].
 * Host pointers in, host pointer out; runs on `device`. */
int mfb_sync_correlate(int device, const uint8_t *bits, int B, int L,
                       const int8_t *tmpl, int T, int32_t *scores);

/* Thresholded form of the same correlation: for every stream the positions i (ascending) with
 * score[b][i] >= threshold and their scores -- exactly np.where(np.convolve(bits, tmpl) >= threshold)
 * as used at DEC:101 and DEC:113 -- without moving the full score arrays to the host.
 * counts[b] receives the TOTAL number of hits of stream b; the first min(counts[b], max_hits) of them
 * are stored in hit_idx[b][..] / hit_score[b][..] (row stride max_hits).  A caller that sees
 * counts[b] > max_hits repeats the call with a larger max_hits (or uses mfb_sync_correlate). */
int mfb_sync_find(int device, const uint8_t *bits, int B, int L, const int8_t *tmpl, int T,
                  int threshold, int max_hits, int32_t *hit_idx, int32_t *hit_score, int32_t *counts);

/* The same for several templates in one call: the decoder correlates every block's bit stream with the header mask
 * AND the sync flag (DEC:96-101 and DEC:112-113).  Template t has T[t] taps at tmpls + T[0] + ... + T[t-1] and threshold
 * thresholds[t]; its results are counts[t*B + b], hit_idx / hit_score[(t*B + b)*max_hits + ..].  ntmpl <= 16.  Small
 * calls (<= 1 MiB of bits and of results) move inputs and results as one packed copy each through page-locked
 * staging: one host-device round trip per block instead of two calls x five copies. */
int mfb_sync_find_multi(int device, const uint8_t *bits, int B, int L, const int8_t *tmpls, const int *T,
                        const int *thresholds, int ntmpl, int max_hits, int32_t *hit_idx, int32_t *hit_score,
                        int32_t *counts);

/* A decoder's own sync finder: the templates and thresholds it searches in every block's bit stream (header mask and sync flag,
 * DEC:96-113) stay on the device with a stream (highest priority), page-locked staging and result buffers of the finder's own.
 * _begin copies the stream in, enqueues the search and returns at once; _end waits and hands out, per template t,
 * counts[t] and the first min(counts[t], max_hits) positions / scores at hit_idx + t*max_hits (ascending).  One search in
 * flight per finder.  The reference computes both correlations with np.convolve on the host, synchronously. */
typedef struct mfb_syncfinder mfb_syncfinder;
int mfb_syncfinder_create(mfb_syncfinder **out, int device, const int8_t *tmpls, const int *T, const int *thresholds, int ntmpl,
                          int max_bits, int max_hits);
int mfb_syncfinder_destroy(mfb_syncfinder *finder);
int mfb_syncfinder_begin(mfb_syncfinder *finder, const uint8_t *bits, int L);
int mfb_syncfinder_end(mfb_syncfinder *finder, int32_t *counts, int32_t *hit_idx, int32_t *hit_score);

/* The same search on PACKED bit streams -- 8 bits per byte in numpy.packbits layout (stream bit i = bit 7 - i%8 of byte
 * i/8), row b at packed + b*row_bytes -- for taps in {-1, 0, +1}: the correlation is popcount(W & P) - popcount(W & Q) on
 * 64-bit windows (exact), an eighth of the bytes cross the host link, and only the hits come back, as ONE flat list: stream
 * b's hits (ascending position) start at sum(counts[0..b)).  total_hits may exceed max_total: the lists then hold the
 * first max_total hits, call again with more room.  device_ms (optional): kernel time between the copies, from HIP
 * events.  Returns MFB_ERR_UNSUPPORTED for other tap values.  Replaces np.convolve + np.where of DEC:96-113 for batches.
 * mfb_sync_pinned_buffer: a page-locked buffer of at least `bytes` owned by the library (per device) that a caller may
 * produce its packed streams into, so that their copy to the device is a plain asynchronous DMA. */
int mfb_sync_find_packed(int device, const uint8_t *packed_bits, int B, int L, int row_bytes, const int8_t *tmpl, int T,
                         int threshold, int max_total, int32_t *hit_idx, int32_t *hit_score, int32_t *counts,
                         int32_t *total_hits, float *device_ms);
int mfb_sync_pinned_buffer(int device, size_t bytes, void **host);

/* Bit-stream alignment cross-correlation of the soft combiner:
 *   out = ifft( fft(a, N) * conj(fft(b, N)) ),  N = the handle's block length,
 * a, b real float32 sequences truncated / zero-padded to N as np.fft.fft(a, N) does; out complex64 [N].
 * Replaces customXCorr (lib/customXCorr.py:5-18; call site softCombiner.py:701-706).  Uses the handle's
 * spectrum / filter / output buffers as scratch: give it a handle of its own (M = 1); filters and input
 * of the handle are invalidated. */
int mfb_xcorr(mfb_ctx *ctx, const float *a, int Na, const float *b, int Nb, float *out_c64);

/* HIP-event stopwatch on the handle's stream (bench.py's live kernel timing).  The reference times blocks with time.time()
 * around the whole call (DP:324-333). */
int mfb_timer_start(mfb_ctx *ctx);
int mfb_timer_stop(mfb_ctx *ctx, float *elapsed_ms);
/* Per-kernel accounting: when enabled, every launch of the search kernels is bracketed by HIP
 * events on the handle's stream; mfb_profile_read synchronises and returns launch counts and
 * summed milliseconds, then clears them: slot 0 = pass 1 (two-pass) or the segment kernel,
 * slot 1 = pass 2 (two-pass; stays 0 on the segment path).  No reference counterpart (the reference logs block rates only,
 * DP:324-333). */
int mfb_profile_enable(mfb_ctx *ctx, int on);
int mfb_profile_read(mfb_ctx *ctx, int counts[2], float total_ms[2]);
/* Block until all work enqueued on the handle's stream has finished (the reference: cuda.Context.synchronize(), DB:651). */
int mfb_sync(mfb_ctx *ctx);

/* A host copy worker for the receive loop: the reference's loop copies every chunk of samples twice on its one thread, into the
 * ring buffer (sigFIFO.py:62-84) and from there into the page-locked input buffer (`raw[ov:] = sigIn.getBlock()`, DP:287,337).
 * Here a chunk is copied once, into the sample window of its batch (mfb_window_buffer) -- and with this worker that copy runs on a
 * thread of its own while the caller's thread does the host stages of the previous batch.  No GPU involved (plain memory; usable
 * before any handle exists).  mfb_hostcopy_submit returns at once; copies run in submission order; source and destination must
 * stay valid and untouched until mfb_hostcopy_drain has returned.  One submitting thread per worker.  (Round 6 tried two and three
 * worker threads: no gain -- 1363 ... 1432 Msamples/s with one, 1360 ... 1419 with two, 1248 ... 1346 with three at 2^15 x 64 x 32 blocks
 * per call --, the loop is bound by its own thread, not by the copies.) */
typedef struct mfb_hostcopy mfb_hostcopy;
int mfb_hostcopy_create(mfb_hostcopy **out);
int mfb_hostcopy_submit(mfb_hostcopy *q, void *dst, const void *src, size_t bytes);
int mfb_hostcopy_drain(mfb_hostcopy *q);           /* returns when every submitted copy has been made (before the batch is begun; DP:287) */
void mfb_hostcopy_destroy(mfb_hostcopy *q);        /* finishes what was submitted, then stops the thread (no reference counterpart) */

#ifdef __cplusplus
}
#endif
#endif /* MFBANK_H */
