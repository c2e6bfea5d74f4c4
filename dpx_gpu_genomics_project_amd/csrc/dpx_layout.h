/*
 * dpx_layout.h -- the engine's device-resident matrix layouts (shared by host and device code).
 *
 * Full-matrix algorithms (LNW / LSW / ANW): "wavefront-tiled" layout.
 *   One wave owns a pair.  Query rows are cut into stripes of 64*R rows; lane l of the wave owns the R
 *   consecutive rows [l*R, l*R+R) of the stripe and sweeps the columns with a skew of one column per lane, and
 *   rolls straight into the next stripe when it reaches the last column (the reference's V2 idea,
 *   cuda/LNW/LinearNeedlemanWunschV2.cu:81-148).  Lane l is therefore on stripe k, column j at the pair's step
 *   T = k*n + (j-1) + l, and what the wave produces in one step is stored contiguously ("chunk" T):
 *
 *       element(i, j, plane) = matOff + T * chunkStride + (plane * 64 + l) * R + r      (R <= 8; R = 16: dpx_tile_off)
 *       with i0 = i-1, k = i0 / (64R), l = (i0 % (64R)) / R, r = i0 % R, T = k*n + (j-1) + l, P = planes (1; 3 for ANW)
 *
 *   i.e. [step][plane][lane][row-in-lane] int16, S*n + 63 chunks per pair (S = stripes).  Every step of a wave is
 *   one fully coalesced 64*R*2-byte store per plane (1 KiB at R = 8).  The reference reached the same idea in
 *   cuda/LNW/LinearNeedlemanWunschV17.cu:106-118 (skewed direction matrix, un-skewed by the back-tracker).  The
 *   only slots that hold no cell are the 63-step skew ramps at the start and end of a pair (63*64*R*2*P bytes,
 *   3 % at 1024x1024); they are written (whole chunks are ~2.5x cheaper than partial ones) but never read.
 *   Border row 0 / column 0 are not stored (they are closed-form); dpx_batch_matrix() re-creates them.
 *
 *   Group interleaving: G pairs that are launched next to each other (64 by default) share one block in which the
 *   chunks of the same step lie side by side: chunk T of member g starts at groupBase + (T*G + g)*chunkElems,
 *   i.e. matOff = groupBase + g*chunkElems and chunkStride = G*chunkElems in the formula above
 *   (element = matOff + T*chunkStride + (plane*64 + l)*R + r).  The G waves then write one compact, forward-moving
 *   window instead of G streams 2 MB apart -- measured +12 % write bandwidth (tools/wbench.hip: 5.45 -> 6.12 TB/s),
 *   the difference between ~7000 and ~200 DRAM pages / TLB entries open at a time.
 *
 * Banded SW: anti-diagonal-major band layout, see dpx_band_index().
 */
#ifndef DPX_LAYOUT_H
#define DPX_LAYOUT_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DPX_HD __host__ __device__ __forceinline__
#else
#define DPX_HD static inline
#endif

#define DPX_WAVE 64
#define DPX_NEG (-(1 << 29)) /* "minus infinity" for the affine gap matrices' virtual borders */

/* per-pair record the kernels read (device copy of dpx_seq_pair + placement) */
typedef struct dpx_pair_dev {
    int32_t refIdx, n;  /* reference offset / length  (columns) */
    int32_t qryIdx, m;  /* query offset / length      (rows)    */
    uint64_t matOff;    /* first int16 element of this pair's chunk 0 */
    uint32_t chunkStride; /* int16 elements between consecutive chunks (steps) of this pair */
    uint16_t lanes;       /* layout tag -- 64: wavefront-tiled (one wave per pair); 16: 8x8 tiles (lane-packed kernels, several pairs per
                             wave); 32: split (k_linear_split, one wave per stripe) */
    uint16_t rows;        /* rows per lane of the kernel that fills this pair (lane-packed and split batches set it); 0 = the batch's */
} dpx_pair_dev;

/* Lane-packed kernels (k_linear_lanes / k_affine_lanes): what one wave aligns.  Up to DPX_WAVE_SLOTS pairs share the 64
 * lanes; slot k owns lanes [first[k], first[k] + num[k]) with num = ceil(m / rows per lane).  Built by the host
 * (dpx_capi.cpp: pack_waves), read through scalar loads. */
#define DPX_WAVE_SLOTS 12 /* (a multiple of 4: the kernels read the descriptor as dwords; 12 pairs of 5 lanes fill a wave of the packed lane kernel) */
typedef struct dpx_wave_desc {
    int32_t pair[DPX_WAVE_SLOTS];    /* batch index of the slot's pair */
    uint8_t first[DPX_WAVE_SLOTS];   /* first lane */
    uint8_t num[DPX_WAVE_SLOTS];     /* lanes (0: slot unused) */
    uint16_t refOff[DPX_WAVE_SLOTS]; /* LDS offset / 16 of the slot's staged reference inside the wave's reference area */
} dpx_wave_desc;

/* number of stripes / elements of one pair's block */
DPX_HD int dpx_tiled_stripes(int m, int R) { return (m + 64 * R - 1) / (64 * R); }
DPX_HD uint64_t dpx_tiled_chunks(int m, int n, int R) { /* steps (chunks) of one pair */
    if (m <= 0 || n <= 0) return 0;
    return (uint64_t)dpx_tiled_stripes(m, R) * (uint64_t)n + 63u;
}
DPX_HD uint32_t dpx_tiled_chunk_elems(int R, int planes) { return (uint32_t)(planes * 64 * R); }
/* Position of (plane, lane l, row-in-lane r) inside a chunk.  A lane's rows are kept in sub-tiles of at most 8 rows
 * (16 B per lane), each sub-tile a contiguous 1 KiB [lane][8] block, so every store instruction of the wave covers
 * whole 64-B sectors: for R = 16 the chunk is [plane][sub-tile 0..1][lane][8] instead of [plane][lane][16]. */
/* (every divisor of the layouts -- R, 64R, sub-tile height, banded C and G -- is a power of two: shifts and masks, the
 * traceback computes three of these offsets per path step) */
DPX_HD int dpx_log2(int v) { return 31 - __builtin_clz((unsigned)v); }
DPX_HD uint32_t dpx_tile_off(int R, int plane, int l, int r) {
    const int sq = R < 8 ? dpx_log2(R) : 3, Q = R >> sq; /* sub-tiles of 2^sq rows */
    return (uint32_t)(((((plane * Q + (r >> sq)) << 6) + l) << sq) + (r & ((1 << sq) - 1)));
}
/* Tile layout (pairs filled by the lane-packed kernels, dpx_pair_dev.lanes == 16): the matrix is cut into 8 x 8 tiles, one
 * 128-byte line each (columns before rows inside the tile: the 8 rows of one column are 16 contiguous bytes -- what a lane
 * produces per step -- and the 8 columns of a row block are one whole line -- what it has produced after 8 steps).  The lines
 * are laid out as the WAVE produces them: a pair's lane l (rows [l*R, l*R+R), R = 8*Q) is lane lambda = first + l of a wave that
 * aligns several pairs (dpx_wave_desc), runs column j in step t = j - 1 + skew with skew = l + (first & 7), and therefore
 * completes column block cb = (j-1)/8 in step t = 8*(cb+1) + skew - 1 -- together with lanes lambda +- 8, +- 16 ... of the same
 * wave, whatever pairs they belong to.  The eight lines of one step are stored side by side:
 *
 *       element(i, j, plane) = matOff + t * (planes*Q*512) + (plane*Q + h) * 512 + (lambda >> 3) * 64 + c*8 + rr
 *       l = (i-1)/R, h = ((i-1)%R)/8, rr = (i-1)%8, c = (j-1)%8, t and lambda as above; matOff = the WAVE's base (shared by its pairs),
 *       chunkStride = first (the pair's first lane)
 *
 * so every step of a wave is one contiguous 1-KiB store per plane and row-block half (as in the wavefront-tiled layout), but only
 * lines that hold cells are ever written: no skew-ramp padding, rows rounded up to 8 instead of to the 64 / 128-row height of a
 * [lane][rows] line (round 1's [step][lane][rows] chunks wrote 1.40x the algorithmic bytes on short reads; this writes 1.05x).
 * The kernels transpose through LDS to get there, see k_linear_lanes. */
DPX_HD uint32_t dpx_tile8_row_blocks(int m) { return (uint32_t)((m + 7) >> 3); }
DPX_HD uint32_t dpx_wtile_step_elems(int Q, int planes) { return (uint32_t)(planes * Q * 512); }
/* steps (1-KiB chunks per plane and half) the wave of this pair needs for it: its last column block completes in step n8 + skew - 1 */
DPX_HD uint64_t dpx_wtile_steps(int m, int n, int R, int first) {
    if (m <= 0 || n <= 0) return 0;
    const int L = (m + R - 1) / R;
    return (uint64_t)(((n + 7) & ~7) + (L - 1) + (first & 7));
}
DPX_HD uint64_t dpx_wtile_index(int i, int j, int plane, int planes, int R, uint32_t first) {
    const int Q = R >> 3, i0 = i - 1, j0 = j - 1;
    const int l = i0 / R, h = (i0 % R) >> 3;
    const int lam = (int)first + l, skew = l + ((int)first & 7);
    const uint64_t t = (uint64_t)(((j0 >> 3) + 1) * 8 + skew - 1);
    return t * (uint64_t)dpx_wtile_step_elems(Q, planes) + (uint64_t)(((plane * Q + h) << 9) + ((lam >> 3) << 6) + ((j0 & 7) << 3) + (i0 & 7));
}
/* offset of cell (i, j) of `plane` relative to the pair's matOff */
DPX_HD uint64_t dpx_tiled_index(int i, int j, int n, int R, int plane, uint32_t chunkStride) {
    const int sr = dpx_log2(R), i0 = i - 1;
    const int k = i0 >> (sr + 6);
    const int l = (i0 >> sr) & 63;
    const int r = i0 & (R - 1);
    uint64_t T = (uint64_t)k * (uint64_t)n + (uint64_t)(j - 1) + (uint64_t)l;
    return T * (uint64_t)chunkStride + (uint64_t)dpx_tile_off(R, plane, l, r);
}
/* Split layout (pairs filled by k_linear_split, dpx_pair_dev.lanes == 32): the stripes of a pair are filled by different
 * waves of one workgroup at the same time, so every stripe owns its chunks -- stripe k runs its n + 63 steps in
 * T = k*SS + (j-1) + l with the stripe stride SS = n + 63 rounded up to a multiple of 16 -- and G = 8/R steps of a lane
 * (R = 2, 4 or 8 rows of 2 bytes) share one 16-byte store:
 *       element(i, j) = matOff + (T / G) * chunkStride + l*8 + (T % G)*R + r          (one plane; 1-KiB chunks) */
DPX_HD int dpx_split_group(int R) { return R >= 8 ? 1 : 8 / R; }
DPX_HD uint32_t dpx_split_stripe_steps(int n, int R) { (void)R; return ((uint32_t)n + 63u + 15u) & ~15u; } /* whole 16-step blocks of the kernel */
DPX_HD uint64_t dpx_split_chunks(int m, int n, int R) {
    if (m <= 0 || n <= 0) return 0;
    return (uint64_t)dpx_tiled_stripes(m, R) * (uint64_t)(dpx_split_stripe_steps(n, R) / (uint32_t)dpx_split_group(R));
}
DPX_HD uint64_t dpx_split_index(int i, int j, int n, int R, uint32_t chunkStride) {
    const int sr = dpx_log2(R), sg = 3 - sr, i0 = i - 1; /* G = 2^sg */
    const int k = i0 >> (sr + 6), l = (i0 >> sr) & 63, r = i0 & (R - 1);
    const uint64_t T = (uint64_t)k * (uint64_t)dpx_split_stripe_steps(n, R) + (uint64_t)(j - 1) + (uint64_t)l;
    return (T >> sg) * (uint64_t)chunkStride + (uint64_t)((l << 3) + (((int)T & ((1 << sg) - 1)) << sr) + r);
}

/* either layout, by the pair's `lanes` tag */
DPX_HD uint64_t dpx_cell_index(int i, int j, int n, int R, int plane, int planes, uint32_t chunkStride, uint32_t lanes) {
    if (lanes == 32) return dpx_split_index(i, j, n, R, chunkStride);
    return lanes == 16 ? dpx_wtile_index(i, j, plane, planes, R, chunkStride) : dpx_tiled_index(i, j, n, R, plane, chunkStride);
}

/*
 * Banded SW (band B: cells with |i-j| <= B-1).  The wave walks anti-diagonals a = i + j (2 .. m+n), step A = a-2.
 * On one anti-diagonal the in-band cells have u = i - j + (B-1) in [0, 2B-2] with u == (a + B - 1) mod 2, so at
 * most B of them exist: slot s = u >> 1 in [0, B).  Lane l holds the C = ceil(B/64) slots [l*C, l*C+C).
 * G = max(1, 8/C) consecutive steps are packed so that every lane writes 16 contiguous bytes:
 *
 *       element(i, j) = (A/G)*chunkStride + l*(G*C) + (A%G)*C + c,   A = i+j-2, s = (i-j+B-1)>>1, l = s/C, c = s%C
 *       (chunkStride = 512 elements when a pair stands alone, GROUP*512 when GROUP pairs are interleaved)
 */
DPX_HD int dpx_band_cpl(int band) { int c = (band + 63) / 64; return c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : 8; } /* cells per lane */
DPX_HD int dpx_band_group(int C) { return C >= 8 ? 1 : 8 / C; }                                                /* steps per 16-B store */
DPX_HD uint64_t dpx_band_chunks(int m, int n, int band) { /* 512-element (1 KiB) chunks of one pair */
    if (m <= 0 || n <= 0) return 0;
    const int C = dpx_band_cpl(band), G = dpx_band_group(C);
    return ((uint64_t)(m + n - 1) + (uint64_t)G - 1) / (uint64_t)G;
}
/* offset of in-band cell (i, j) relative to the pair's matOff (chunks are group-interleaved like the tiled layout) */
DPX_HD uint64_t dpx_band_index(int i, int j, int band, uint32_t chunkStride) {
    const int C = dpx_band_cpl(band), G = dpx_band_group(C);
    const int A = i + j - 2;
    const int s = (i - j + (band - 1)) >> 1;
    const int sc = dpx_log2(C), sg = dpx_log2(G);
    const int l = s >> sc, c = s & (C - 1);
    return (uint64_t)(A >> sg) * (uint64_t)chunkStride + (uint64_t)((l << (sg + sc)) + ((A & (G - 1)) << sc) + c);
}

#endif
