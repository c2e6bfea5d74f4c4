/* dpx_kernels.h -- interface between the C-ABI layer (dpx_capi.cpp) and the HIP kernels (dpx_kernels.hip). */
#ifndef DPX_KERNELS_H
#define DPX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dpx_layout.h"

/* same numbering as dpx_algo in include/dpx_align.h */
#define DPX_K_LNW 0
#define DPX_K_LSW 1
#define DPX_K_ANW 2
#define DPX_K_BSW 3

/* one wave per pair; DPX_FILL_THREADS/64 independent waves share a workgroup (no barriers between them) */
#ifndef DPX_FILL_THREADS
#define DPX_FILL_THREADS 256
#endif


typedef struct dpx_fill_args {
    const char *seq;            /* flat sequence bytes (device copy of parseInput's buffer) */
    const dpx_pair_dev *pairs;  /* per-pair geometry + matrix offset */
    const int32_t *order;       /* optional launch order (longest pairs first) or NULL */
    const dpx_wave_desc *waves; /* lane-packed kernels: one descriptor per wave (numPairs = number of waves) */
    int32_t numPairs;
    int32_t match, mismatch, gapOpen, gapExtend, band;
    int16_t *mat;               /* matrix pool (int16, engine layout) or NULL when score-only */
    int32_t *score, *endRow, *endCol;
    uint32_t ldsPerWave;        /* bytes of dynamic LDS per wave */
    uint32_t wavesPerBlock;     /* independent waves per workgroup of the one-wave-per-pair / -couple kernels: 4, or 1 for small launches */
    uint32_t ldsEdge2Off;       /* ANW: offset of the second edge row (D) */
    uint32_t ldsRefOff;         /* offset of the staged reference characters */
    uint32_t ldsQryOff;         /* offset of the staged query characters (rolling multi-stripe path) */
    uint32_t ldsBufStride;      /* split kernel: int16 elements between two edge rows */
    int32_t rowTags;            /* packed SW kernel: 1 = one (score, row-in-lane, column) key per pair and lane (needs max score * R + R-1 <= 65535),
                                   0 = one (score, column) key per pair and row */
    int32_t rampLines;          /* 1: skew-ramp steps store only the 128-byte lines that hold cells (byte-bound batches); 0: whole chunks */
} dpx_fill_args;

hipError_t dpx_launch_fill(const dpx_fill_args &a, int algo, int R, bool store, size_t ldsBytes, hipStream_t stream);
hipError_t dpx_launch_fill_lanes(const dpx_fill_args &a, int algo, int R, bool store, size_t ldsBytes, hipStream_t stream);
/* LDS in front of the staged references of a lane-packed wave (line stage, or the lane scratch when score-only), and
 * waves per workgroup of the kernel for `algo` */
size_t dpx_lanes_stage_bytes(int algo, int R, bool store);
int dpx_lanes_waves_per_block(int algo);
hipError_t dpx_launch_fill_lanes_packed(const dpx_fill_args &a, int algo, size_t ldsBytes, hipStream_t stream);
hipError_t dpx_launch_fill_split(const dpx_fill_args &a, int algo, int R, int waves, size_t ldsBytes, hipStream_t stream);
hipError_t dpx_launch_banded_packed(const dpx_fill_args &a, int C, size_t ldsBytes, hipStream_t stream);
hipError_t dpx_launch_fill_packed(const dpx_fill_args &a, int algo, int R, size_t ldsBytes, hipStream_t stream);
hipError_t dpx_launch_export(const int16_t *mat, const dpx_pair_dev &pr, int algo, int R, int planes, int plane, int gapOpen,
                             int gapExtend, int band, int16_t *out, hipStream_t stream);
/* walk: 0 = one lane per pair, three parallel 2-byte loads per step; 1 = one lane per pair through register-resident 8-row
 * column vectors (LSW / LNW, pays off when the batch is bound by sector requests rather than by load latency); 2 = one WAVE
 * per pair with an LDS window of 32 rows x 64 columns (LSW / LNW on layouts with 8-row vectors; other pairs fall back to 0) */
hipError_t dpx_launch_traceback(const dpx_fill_args &a, int numPairs, int algo, int R, int planes, int walk,
                                const uint64_t *tbOff, char *tb, int32_t *tbLen, hipStream_t stream);
size_t dpx_out_scan_tiles(size_t numPairs);
hipError_t dpx_launch_output(const dpx_pair_dev *pairs, const int32_t *score, const int32_t *tbLen, const uint64_t *tbOff, const char *tb,
                             int numPairs, unsigned long long firstNumber, unsigned long long *tileSums, unsigned long long *outOff, char *out,
                             bool scanOnly, bool compactOnly, hipStream_t stream);
hipError_t dpx_launch_unpack2(const uint32_t *packed, uint32_t alphabet, char *out, size_t numDwords, hipStream_t stream);
hipError_t dpx_launch_prim_eval(const int32_t *op, const uint32_t *a, const uint32_t *b, const uint32_t *c, size_t count,
                                uint32_t *res, uint32_t *pred, hipStream_t stream);

#endif
