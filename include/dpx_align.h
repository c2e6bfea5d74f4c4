/*
 * dpx_align.h -- C ABI of libdpxalign.so, the MI355X (gfx950) pairwise-alignment DP engine.
 *
 * This is the drop-in boundary for ONE hot path of mickgordinier/DPX_GPU_Genomics_Project:
 * the anti-diagonal wavefront matrix fill of a batch of independent pairwise alignments
 * (SURVEY.md section 8).  The reference has no FFI of its own -- its boundary is its C++ headers
 * (c++/SequenceAligner.h, c++/parseInput.h, c++/backtrack.h) and the `<<<grid,block>>>` launches in
 * its cuda/ *.cu mains.  Every entry point below names the reference interface it replaces.
 * Plain pointers and sizes only; no C++/torch types.  All functions return 0 (DPX_OK) or a negative
 * dpx_status; none of them exits the process.
 *
 * Thread-safety: every call is safe from concurrent host threads (the reference's CPU driver runs
 * 20 pthreads, c++/main.cpp:18-19,203); calls on the SAME dpx_batch must be serialised by the caller.
 */
#ifndef DPX_ALIGN_H
#define DPX_ALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round 1.  2: + dpx_batch_create_on, dpx_batch_fill_timed, dpx_batch_last_fill_usec, dpx_batch_output_begin/_end/_take,
 * dpx_text_free, DPX_TUNE_PLACEMENT (round 2).  3: + dpx_pool_reserve, dpx_text_reserve, dpx_batch_last_output_usec, dpx_pack2, dpx_batch_create_packed2; dpx_batch_describe reports the
 * matrix pool (round 3).  Additions only: a caller built against an older version keeps working; dpx_abi_version() >= the version
 * a caller needs is the check. */
#define DPX_ABI_VERSION 3

typedef enum dpx_status {
    DPX_OK = 0,
    DPX_ERR_INVALID = -1,      /* bad argument (NULL, negative size, unknown algo ...) */
    DPX_ERR_NO_DEVICE = -2,    /* no usable gfx950 device / HIP runtime failure at init */
    DPX_ERR_HIP = -3,          /* a HIP call failed; dpx_last_error() has the text */
    DPX_ERR_RANGE = -4,        /* scores of this batch cannot be held in int16 cells */
    DPX_ERR_NOMEM = -5,        /* device or host allocation failed */
    DPX_ERR_NOT_FILLED = -6,   /* results requested before dpx_batch_fill() */
    DPX_ERR_NO_MATRIX = -7,    /* matrix / traceback requested from a DPX_SCORE_ONLY batch */
    DPX_ERR_UNSUPPORTED = -8   /* parameter combination the engine does not implement */
} dpx_status;

/* Algorithm selector.  Replaces the compile-time switch `#define LSW_ENABLE / LNW_ENABLE / ANW_ENABLE`
 * (c++/main.cpp:22-24) and the one-program-per-algorithm split of cuda/ *.cu. */
typedef enum dpx_algo {
    DPX_ALGO_LNW = 0, /* LinearNeedlemanWunsch   c++/LinearNeedlemanWunsch.cpp:89-135, cuda/LNW/ *.cu      */
    DPX_ALGO_LSW = 1, /* LinearSmithWaterman     c++/LinearSmithWaterman.cpp:70-114,  cuda/LinearSmithWaterman.cu */
    DPX_ALGO_ANW = 2, /* AffineNeedlemanWunsch   c++/AffineNeedlemanWunsch.cpp:167-240, cuda/AffineNeedlemanWunsch.cu */
    DPX_ALGO_BSW = 3  /* BandedSmithWaterman     python/LinearBandedSmithWaterman.py:62-104 (C++/CUDA twins are broken) */
} dpx_algo;

/* Identical in layout to the reference's `struct seqPair` (c++/parseInput.h:22-29): byte offsets into the
 * flat `sequences` buffer parseInput() produced, plus lengths. */
typedef struct dpx_seq_pair {
    int32_t referenceIdx;
    int32_t referenceSize;
    int32_t queryIdx;
    int32_t querySize;
} dpx_seq_pair;

/* Scoring parameters: the reference's argv `-match -mismatch -open(-gap) -extend` (c++/main.cpp:133-150,
 * cuda/LinearSmithWaterman.cu:205-216).  LNW/LSW/BSW use gapOpen as THE linear gap (main.cpp:52,77). */
typedef struct dpx_params {
    int32_t algo;      /* dpx_algo */
    int32_t match;
    int32_t mismatch;
    int32_t gapOpen;   /* linear gap for LNW/LSW/BSW; gap-open for ANW */
    int32_t gapExtend; /* ANW only */
    int32_t band;      /* BSW only: cells with |i-j| <= band-1 are computed */
} dpx_params;

/* dpx_batch_create flags */
#define DPX_KEEP_MATRICES 0x0u /* default: write the int16 score matrices (H; H,I,D for ANW) to HBM */
#define DPX_SCORE_ONLY    0x1u /* no matrix writeback (not HBM-bound; never used for the roofline figure) */
#define DPX_TIME_FILLS    0x2u /* bracket every dpx_batch_fill() with HIP events: dpx_batch_last_fill_usec() */
#define DPX_TUNE_PLACEMENT 0x4u /* the batch will be filled many times: time its matrix pool (>= 1 GiB) with hipMemset and shop for a better
                                   one with the batch's OWN FILL on four more candidate pools (the same fill runs 2 - 27 % apart on
                                   two pools of the same construction); every candidate's times go into dpx_batch_describe's pool_* fields */

/* matrix selectors for dpx_batch_matrix */
#define DPX_MAT_H 0 /* scoring matrix   (reference: memo / scoringMemo)            */
#define DPX_MAT_I 1 /* ANW horizontal-gap matrix (queryInsertionMemo)               */
#define DPX_MAT_D 2 /* ANW vertical-gap matrix   (queryDeletionMemo)                */

typedef struct dpx_batch dpx_batch; /* opaque, device-resident batch of pairs */

/* ---- process / device ------------------------------------------------------------------------------ */

/* Bind the calling process to HIP device `device` (one process per GPU; multi-GPU = one rank per device).
 * Replaces cudaGetDeviceCount/cudaGetDeviceProperties(0) at the top of every cuda/ *.cu main
 * (cuda/LinearSmithWaterman.cu:176-190).  Idempotent. */
int dpx_init(int device);
int dpx_device_count(int *count);
/* name (<=255 chars), CU count, HBM bytes of the bound device; any pointer may be NULL */
int dpx_device_info(char *name, size_t nameCap, int *computeUnits, size_t *hbmBytes);
int dpx_shutdown(void);
/* Allocate `count` (1..8; one if `bytes` >= 16 GiB) matrix pools of `bytes` each on the default device and park them for the batches to come (any batch
 * whose matrices fit takes a parked pool instead of allocating), together with two streams per pool for those batches.  Meant for a helper thread while the caller parses its input:
 * the reference sizes its device buffers once, before the batch loop (cuda/LNW/LinearNeedlemanWunschV14.cu:144-213). */
int dpx_pool_reserve(size_t bytes, int count);
/* The same for `count` (1..9) pinned host buffers of `bytes` (<= 1 GiB) each, which the result text of the batches to come is copied
 * into (dpx_batch_output_end / _take; a parked buffer of up to 32 MiB serves any text of 2 MiB or more that fits). */
int dpx_text_reserve(size_t bytes, int count);
const char *dpx_strerror(int status);
const char *dpx_last_error(void); /* thread-local text of the last HIP failure */
int dpx_abi_version(void);

/* ---- batched path (replaces the batched cuda mains, cuda/LNW/LinearNeedlemanWunschV19.cu:422-612) --- */

/* Copy `sequences[0..numBytes)` and pairs[firstPair .. firstPair+numPairs) to HBM and allocate result
 * storage.  Replaces cudaMalloc+cudaMemcpy of sequences/seqPair[] (V19.cu:422-440) and the per-batch matrix
 * pool (V19.cu:488-529).  Sequences are plain bytes ('\0'-separated as parseInput leaves them); any byte
 * value is legal, matching is plain byte equality.  Zero-length sequences are legal. */
int dpx_batch_create(const dpx_params *params, const char *sequences, size_t numBytes, const dpx_seq_pair *pairs,
                     size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out);

/* The same on an explicit device (0 .. dpx_device_count()-1; -1 = the default device): one host process can drive
 * several GPUs, each batch lives on the device it was created on and every call on it runs there.  The class surface
 * uses this to spread the batches it forms from the reference's 20 threads over all visible devices (hostcpp/DpxPair.cpp;
 * rehearsed with several leaders on one GPU only, never run on a multi-GPU box); one process per GPU (dpx_init(rank)) remains
 * the layout of the batched driver and of bench.py.  device >= count is DPX_ERR_INVALID. */
int dpx_batch_create_on(int device, const dpx_params *params, const char *sequences, size_t numBytes, const dpx_seq_pair *pairs,
                        size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out);

/* ---- 2-bit packed input (the input side of c++/parseInput.cpp:78-112) -------------------------------------------
 * The reference keeps one byte per base.  A caller whose sequences use at most four distinct byte values (DNA: "ACGT", the
 * reference's datasets: "0123") may hand the engine four bases per byte: base k sits in bits 2*(k%4) of byte k/4, alphabet[code] is
 * the byte a code stands for, and the pairs keep the reference's struct seqPair -- their indices count bases of the packed buffer
 * exactly as they counted bytes of the flat one.  The engine moves a quarter of the bytes over PCIe and expands them on the device
 * (k_unpack2) into the byte buffer that the fill, traceback and output kernels read: results, matrices and printed text are
 * those of the byte batch.  dpx_pack2() is the host side: it derives the alphabet from the bytes inside the pairs' ranges (order of
 * first appearance) and packs `sequences` (bytes outside every pair, e.g. parseInput's separators, become code 0);
 * DPX_ERR_UNSUPPORTED when the pairs use more than four byte values (the caller keeps dpx_batch_create).  `packed` must hold
 * (numBytes + 3) / 4 bytes. */
int dpx_pack2(const char *sequences, size_t numBytes, const dpx_seq_pair *pairs, size_t numPairs, uint8_t alphabet[4], uint8_t *packed);
int dpx_batch_create_packed2(int device, const dpx_params *params, const uint8_t *packed, size_t numBases, const uint8_t alphabet[4],
                             const dpx_seq_pair *pairs, size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out);

/* Launch the DP fill for every pair of the batch on `stream` (a hipStream_t, or NULL for the batch's own
 * stream).  Asynchronous.  Replaces `needleman_wunsch_kernel<<<BATCH/2,32,smem>>>` (V19.cu:536),
 * `smith_waterman_kernel<<<1,32>>>` (cuda/LinearSmithWaterman.cu:263) and
 * `affine_needleman_wunsch_kernel<<<1,32>>>` (cuda/AffineNeedlemanWunsch.cu:338).  May be called repeatedly.
 * A caller-owned stream must stay valid until the batch has been synchronised or destroyed: dpx_batch_destroy()
 * waits on the stream of the last fill before it parks the batch's buffers for reuse by the next batch.
 * (The inputs of a small batch are still on their way when dpx_batch_create() returns -- one asynchronous copy on the batch's own
 * stream; the first fill on a caller's stream waits for it on the host, fills on the batch's stream are simply ordered behind it.) */
int dpx_batch_fill(dpx_batch *b, void *stream);

/* Run `repeats` fills back-to-back and return the mean device time of one fill in microseconds, measured with
 * hipEvents on the launch stream (the reference's kernel_time accumulator, V19.cu:531-586). */
int dpx_batch_fill_timed(dpx_batch *b, int repeats, double *usecPerFill);

/* Device time of the most recent dpx_batch_fill() of a batch created with DPX_TIME_FILLS, without stalling the launch:
 * the events are recorded on the fill's stream, this call waits only for the second one (the reference accumulates
 * kernel_time the same way around its launches, V19.cu:531-586, but synchronously). */
int dpx_batch_last_fill_usec(dpx_batch *b, double *usec);
/* The same for the device side of the most recent dpx_batch_output_begin(): traceback + text kernels (the reference's
 * backtracking() launch, V19.cu:546-560), without the D2H copies. */
int dpx_batch_last_output_usec(dpx_batch *b, double *usec);

int dpx_batch_sync(dpx_batch *b); /* cudaDeviceSynchronize analogue for this batch's stream */

/* Device pointers of the int32 result arrays (numPairs each) for collectives (RCCL gather of scores).
 * endRow/endCol are the LSW/BSW start cell of the traceback (first strict max in row-major order,
 * c++/LinearSmithWaterman.cpp:145-157); for LNW/ANW they are (m, n). */
int dpx_batch_device_results(dpx_batch *b, void **dScores, void **dEndRow, void **dEndCol);

/* D2H of the per-pair results (any pointer may be NULL).  Replaces cudaMemcpy of similarityScores (V19.cu:590). */
int dpx_batch_results(dpx_batch *b, int32_t *scores, int32_t *endRow, int32_t *endCol);

/* Export one pair's matrix as the reference lays it out: row-major (m+1) x (n+1) int16 including the border
 * row/column (memo[i][j], c++/LinearSmithWaterman.cpp:14-17; cuda scoringMatrix[row*numCols+col]).  `out` is host
 * memory of (m+1)*(n+1) int16.  On the device the matrix lives in the engine's wavefront-tiled layout
 * (DESIGN.md); this call un-tiles it with a device kernel and copies it back. */
int dpx_batch_matrix(dpx_batch *b, size_t pair, int which, int16_t *out);

/* Device traceback of one pair from the stored matrices, with the reference's tie rules.
 * Produces the three lines the reference prints (reference / relation / query; c++/backtrack.cpp:21-356).
 * Each buffer needs m+n+1 bytes; *len receives the alignment length. */
int dpx_batch_traceback(dpx_batch *b, size_t pair, char *refLine, char *relLine, char *qryLine, int32_t *len);

/* The whole batch's result text, formatted on the device exactly as c++/main.cpp prints it: per pair
 *     "<pair number> | <score>\n<reference line>\n<relation line>\n<query line>\n"
 * (three empty lines for a zero-score local alignment, c++/LinearSmithWaterman.cpp:253-257), blocks in batch order, pair
 * numbers counted from `firstPairNumber`.  Replaces the per-pair backtracking + string packing of the batched CUDA mains
 * (cuda/LNW/LinearNeedlemanWunschV15.cu:168-172,372-425: packed variable-length result strings, one D2H of the real
 * bytes; V19.cu:546-579 prints them).  _begin() is asynchronous (device traceback, block lengths, exclusive scan, packed
 * copy, D2H of the offsets on the batch's stream); _end() waits, copies exactly the bytes of the text to pinned host memory
 * and returns it: `*text` (`*bytes` long, also NUL-terminated) and `*offsets` (numPairs + 1 byte offsets of the blocks)
 * stay valid until the batch is filled again or destroyed.  Two batches can be in flight: fill of batch k+1 overlaps
 * traceback + D2H of batch k. */
int dpx_batch_output_begin(dpx_batch *b, uint64_t firstPairNumber);
int dpx_batch_output_end(dpx_batch *b, const char **text, size_t *bytes, const uint64_t **offsets);
/* Like _end(), but the caller takes the (pinned) text buffer over and the batch can be destroyed at once -- its matrix pool
 * is then free for the next batch while a printer thread is still writing the text (the reference prints batch k-1
 * from host strings while batch k runs, V19.cu:546-579).  Give the buffer back with dpx_text_free(). */
int dpx_batch_output_take(dpx_batch *b, char **text, size_t *bytes);
int dpx_text_free(char *text);

/* Sizes: numPairs, total cells (sum refLen*queryLen, the reference's numCells, c++/parseInput.cpp:100),
 * bytes of HBM the matrices occupy, algorithmic bytes of one fill. */
int dpx_batch_info(dpx_batch *b, size_t *numPairs, uint64_t *cells, uint64_t *matrixBytes, uint64_t *algorithmicBytes);

/* One line of text about how the batch will be (was) filled: `algo=LSW kernel_algo=LSW kernel=k_linear_fill_pk dtype=int16
 * rows_per_lane=16 store=1 couples=5000 lane_pairs=0 waves=0 singles=0 streams=0 row_tags=1 pool=vmm pool_bytes=22263365632
 * pool_chunk_mb=1024 pool_kept=0 pool_memset_ms=3.290` (pool_*: how the matrix pool was allocated -- `vmm` = a virtual range
 * backed by physical chunks, `malloc` = one hipMalloc -- and the hipMemset time of every candidate allocation that was timed,
 * `untimed` if none was; pool_kept indexes the one in use).  The reference prints its launch
 * geometry the same way (cuda/LNW/LinearNeedlemanWunschV19.cu:398-409); tests and bench.py read the kernel and the
 * arithmetic type from here instead of guessing the host's choice. */
int dpx_batch_describe(dpx_batch *b, char *buf, size_t cap);

int dpx_batch_destroy(dpx_batch *b);

/* ---- one-shot path (what the SequenceAligner-derived classes call from score_matrix()) ------------- */

/* create + fill + results [+ matrices of pair 0..numPairs-1 into caller buffers] + destroy.
 * H/I/D may be NULL; otherwise they are arrays of numPairs host pointers, each (m+1)*(n+1) int16. */
int dpx_align_batch(const dpx_params *params, const char *sequences, size_t numBytes, const dpx_seq_pair *pairs,
                    size_t numPairs, int32_t *scores, int32_t *endRow, int32_t *endCol, int16_t **H, int16_t **I,
                    int16_t **D);

/* ---- DPX primitive probe (a5: FakeDPX, c++/FakeDPX.hpp:19-126) ----------------------------------- */

/* Evaluate `count` DPX primitives on the device with the CDNA4 instruction mapping used by the kernels
 * (v_max3_i32 / v_pk_max_i16 / v_pk_add_i16 ...).  op numbering follows c++/FakeDPX.hpp declaration order
 * (0 = __vimax3_s32 ... 35 = __viaddmin_s16x2_relu).  pred bit0 = pred / pred_lo, bit1 = pred_hi. */
int dpx_prim_eval(const int32_t *op, const uint32_t *a, const uint32_t *b, const uint32_t *c, size_t count,
                  uint32_t *result, uint32_t *pred);

#ifdef __cplusplus
}
#endif
#endif /* DPX_ALIGN_H */
