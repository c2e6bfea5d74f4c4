/*
 * dpx_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the pairwise-alignment DP fill.
 *
 * One 64-lane wavefront per alignment pair.  Lane l owns R consecutive query rows of a 64*R-row stripe and
 * sweeps the reference columns with a skew of one column per lane, so the wave is always on one anti-diagonal
 * (of R-row tiles).  Per step and lane:
 *   - `up` of the lane's top row comes from the previous lane's bottom row through ONE DPP move
 *     (v_mov_b32_dpp wave_shr:1 -- the reference's __shfl_up_sync, cuda/LNW/LinearNeedlemanWunschV12.cu:149);
 *     the diagonal is last step's `up`, carried in a register (the V10 trick, V10.cu:157-162);
 *   - the R cells of the lane are chained in registers (left/diag/up never touch memory);
 *   - the stripe's bottom row goes to LDS for the next stripe (V12.cu:110-113,141-143: warpEdgeScore);
 *   - the cell update is v_cmp/v_cndmask + v_add + v_max + v_add + v_max3_i32 (FakeDPX __vibmax/__vimax3,
 *     c++/FakeDPX.cpp:11-13,145-153; the >= predicates are not needed here because the traceback recomputes
 *     directions from the stored scores with the same rule);
 *   - the wave's R x 64 scores of this step leave as one fully coalesced int16 store (dpx_layout.h).
 * No MFMA (integer max-plus recurrence), no LDS transposes, no atomics.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dpx_kernels.h"
#include "dpx_layout.h"
#include "dpx_prims.hpp"

namespace {

using dpx::pack_lo16;
using dpx::wave_shr1;
using dpx::wave_shl1;
using dpx::s16x2;
using dpx::u16x2;
using dpx::as_s16x2;
using dpx::as_u16x2;
using dpx::as_u32;

/* ---- coalesced tile store: R int32 scores -> R int16, 2*R bytes per lane, lanes contiguous ---- */
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

/* matrices are written once and never re-read by the fill: DPX_NT_STORES selects `global_store ... nt` */
#ifndef DPX_NT_STORES
#define DPX_NT_STORES 0
#endif
#ifndef DPX_EXP_NOCOMPUTE
#define DPX_EXP_NOCOMPUTE 0
#endif
#ifndef DPX_EXP_NORAMPSTORE
#define DPX_EXP_NORAMPSTORE 0
#endif
/* ablation builds of the lane-packed kernels (tools/ab_quad.sh): 1 = no global stores, 2 = stores without the LDS round trip */
#ifndef DPX_EXP_QUAD
#define DPX_EXP_QUAD 0
#endif

/* ---- sequence staging: wide loads instead of one byte per lane and instruction ----------------------------------
 * Sequences sit at arbitrary byte offsets of the flat parseInput buffer (c++/parseInput.cpp:78-112 keeps the file's
 * bytes where they are).  stage_bytes() copies a string into LDS with aligned 16-byte loads and stores: the copy
 * starts at the 16-byte block that holds the first character, so the string lands `src & 15` bytes into the
 * (16-byte aligned) LDS buffer; the returned pointer is its first character.  The buffer needs n + 31 bytes.  The
 * device arena is 256-byte aligned and padded, so the blocks read never leave it.  `l` = the calling lane's index
 * among the G lanes that share the copy. */
__device__ __forceinline__ unsigned char *stage_bytes(unsigned char *dst16, const unsigned char *src, const int n, const int l, const int G) {
    const unsigned a = (unsigned)(reinterpret_cast<uintptr_t>(src) & 15u);
    const u32x4 *from = reinterpret_cast<const u32x4 *>(src - a);
    u32x4 *to = reinterpret_cast<u32x4 *>(dst16);
    const int blocks = n > 0 ? (int)((a + (unsigned)n + 15u) >> 4) : 0;
    for (int k = l; k < blocks; k += G) to[k] = from[k];
    return dst16 + a;
}

/* four bytes from an arbitrary address: two aligned dword loads + v_alignbyte_b32 */
__device__ __forceinline__ uint32_t load4(const unsigned char *p) {
    const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3u);
    const uint32_t *al = reinterpret_cast<const uint32_t *>(p - sh);
    return __builtin_amdgcn_alignbyte(al[1], al[0], sh);
}

/* the R query characters of a lane's rows (row0 .. row0+R-1 of `qry`) from R/4 + 1 aligned dword loads and
 * v_alignbyte_b32 instead of R byte loads; rows past the query's end get 0x100 (matches no byte) */
template <int R>
__device__ __forceinline__ void load_query_rows(int (&qc)[R], const unsigned char *qry, const int row0, const int nrows) {
    if (nrows <= 0) {
#pragma unroll
        for (int r = 0; r < R; r++) qc[r] = 0x100;
        return;
    }
    constexpr int W = (R + 3) / 4;
    const unsigned char *at = qry + row0;
    const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(at) & 3u);
    const uint32_t *al = reinterpret_cast<const uint32_t *>(at - sh);
    uint32_t d[W + 1];
#pragma unroll
    for (int k = 0; k <= W; k++) d[k] = al[k];
#pragma unroll
    for (int k = 0; k < W; k++) {
        const uint32_t v = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh); /* bytes 4k .. 4k+3 of the lane's rows */
#pragma unroll
        for (int bq = 0; bq < 4 && 4 * k + bq < R; bq++) {
            const int r = 4 * k + bq;
            qc[r] = (r < nrows) ? (int)((v >> (8 * bq)) & 0xFFu) : 0x100;
        }
    }
}

template <class V>
__device__ __forceinline__ void stream_store(V *dst, V v) {
#if DPX_NT_STORES
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;
#endif
}

template <int R>
__device__ __forceinline__ void store_tile(int16_t *dst, const int (&v)[R]) {
    if constexpr (R == 1) {
        *dst = (int16_t)v[0];
    } else if constexpr (R == 2) {
        stream_store(reinterpret_cast<uint32_t *>(dst), pack_lo16(v[0], v[1]));
    } else if constexpr (R == 4) {
        u32x2 w = {pack_lo16(v[0], v[1]), pack_lo16(v[2], v[3])};
        stream_store(reinterpret_cast<u32x2 *>(dst), w);
    } else {
#pragma unroll
        for (int q = 0; q < R / 8; q++) {
            u32x4 w = {pack_lo16(v[8 * q + 0], v[8 * q + 1]), pack_lo16(v[8 * q + 2], v[8 * q + 3]),
                       pack_lo16(v[8 * q + 4], v[8 * q + 5]), pack_lo16(v[8 * q + 6], v[8 * q + 7])};
            stream_store(reinterpret_cast<u32x4 *>(dst + q * 512), w); /* sub-tile q: its own 1 KiB [lane][8] block */
        }
    }
}

/* 64-lane max-reduction of a 64-bit key (once per pair; cost irrelevant) */
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

/* =====================================================================================================
 * Linear-gap fill: LinearNeedlemanWunsch (LOCAL = false) and LinearSmithWaterman (LOCAL = true).
 *   NW cell (c++/LinearNeedlemanWunsch.cpp:105-128):  H = max(left+g, max(up+g, diag+s))
 *   SW cell (c++/LinearSmithWaterman.cpp:82-103):     H = max(0, max(up+g, left+g, diag+s))
 * ===================================================================================================== */
template <int R, bool LOCAL>
struct LinState {
    int Hl[R];        /* H[row][j-1] of the lane's R rows (the "left" values, then overwritten by H[row][j]) */
    int qc[R];        /* query characters of the lane's rows */
    unsigned key[R];  /* SW: per-row running max of (H << 16 | 0xFFFF - j): max score, then smallest column */
    int dtop;         /* H[row0][j-1]: diagonal of the lane's top row */
};

/* Single-stripe pairs shorter than 64*R rows leave the upper lanes without rows; those lanes need not store.  Returns
 * the number of lanes that do: the row-owning lanes rounded up to whole 128-byte lines of the store (a line written
 * in part costs a read-modify-write at the memory side: rounding to 64-byte sectors measured 25 % slower on the
 * short-read batch, profiles/README.md). */
template <int R>
__device__ __forceinline__ int store_lanes(const int m) {
    constexpr int perLine = 128 / (2 * (R < 8 ? R : 8)); /* lanes per 128-byte line: 32 / 16 / 8 for R = 2 / 4 / >= 8 */
    const int owning = (m + R - 1) / R;
    return min(64, (owning + perLine - 1) / perLine * perLine);
}

/* Skew ramps: in step t of a stripe of n columns only the lanes t-n+1 .. t are on a real cell.  Storing the whole chunk writes
 * 6 % of padding at 1024 x 1024 (12 % at 512 x 512); storing exactly the lanes on a cell leaves partly written 128-byte lines,
 * which cost more than they save (profiles/README.md).  So the ramp steps store the lanes on a cell ROUNDED OUT TO WHOLE LINES
 * (8 lanes x 16 B; 16 / 32 lanes for 4 / 2 rows per lane): nearly all of the padding stays unwritten, every written line is
 * whole.  DPX_EXP_FULLRAMP=1 builds the round-1 behaviour (whole chunks) for A/B runs. */
#ifndef DPX_EXP_FULLRAMP
#define DPX_EXP_FULLRAMP 0
#endif
template <int R>
__device__ __forceinline__ bool ramp_stores(const int lane, const int t, const int n, const int rampLines) {
#if DPX_EXP_FULLRAMP
    return true;
#else
    if (!rampLines) return true; /* small batches are bound by the latency of a step, not by bytes: whole chunks are cheaper there */
    constexpr int perLine = 128 / (2 * (R < 8 ? R : 8));
    const int lo = max(t - n + 1, 0) & ~(perLine - 1), hi = (min(t, 63) + perLine) & ~(perLine - 1);
    return lane >= lo && lane < hi;
#endif
}

/* the R chained cells of one lane for one column (shared by the striped and the rolling schedule) */
template <int R, bool LOCAL, bool KEYS>
__device__ __forceinline__ void lin_cells(LinState<R, LOCAL> &st, const int upin, const int rc, const int j, const int match,
                                          const int mismatch, const int gap) {
    int u = upin, d = st.dtop;
    const unsigned negj = 0xFFFFu - (unsigned)j;
#if DPX_EXP_NOCOMPUTE /* ablation build only (tools/): stores without the recurrence */
    st.Hl[0] += u + d + rc + (int)negj;
#else
    /* the diagonal terms first (row r's diagonal is row r-1's LEFT value), then every row in place: the new H[r] overwrites the old
     * one in its own register.  Written the other way round (diagonal taken from a rotating `d`) the values move one register
     * down per step and the compiler has to move them back at the loop edge: 8 v_mov per step at R = 8 (round 3, ISA of
     * k_linear_lanes: 74 vector instructions per step, of which 48 are the recurrence) */
    int dterm[R];
#pragma unroll
    for (int r = 0; r < R; r++) dterm[r] = ((r == 0) ? d : st.Hl[r - 1]) + ((st.qc[r] == rc) ? match : mismatch);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int g = max(u, st.Hl[r]) + gap;
        int h;
        if constexpr (LOCAL) h = max(max(g, dterm[r]), 0); /* v_max3_i32 */
        else h = max(g, dterm[r]);
        u = h;
        st.Hl[r] = h;
        if constexpr (LOCAL && KEYS) st.key[r] = max(st.key[r], ((unsigned)h << 16) | negj);
    }
#endif
    st.dtop = upin;
}

/* The cell update of the int32 kernels (round 3; k_linear_stream still uses lin_cells).  The lane's state is kept as Hg = H + gap: then
 *     H[r][j] = max3(Hg[r-1][j], Hg[r][j-1], Hg[r-1][j-1] + (s - gap))          (one v_max3_i32, SW: the diagonal term clamped at 0 first)
 * and Hg[r][j] = H[r][j] + gap is the only op left on the chain from one row to the next: 5 vector instructions per NW cell
 * instead of 6 (v_cmp, v_cndmask, v_add, v_max3, v_add), two dependent ones per row instead of three.  The diagonal terms are
 * computed first and every row is updated in place (no register rotation, see lin_cells).  h[] returns the plain scores. */
template <int R, bool LOCAL>
__device__ __forceinline__ void lin_cells_g(LinState<R, LOCAL> &st, const int upinG, const int rc, const unsigned negj, const int matchG,
                                            const int mismatchG, const int gap, int (&h)[R]) {
    int dterm[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        dterm[r] = ((r == 0) ? st.dtop : st.Hl[r - 1]) + ((st.qc[r] == rc) ? matchG : mismatchG);
        if constexpr (LOCAL) dterm[r] = max(dterm[r], 0);
    }
    int ug = upinG;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int v = max(max(ug, st.Hl[r]), dterm[r]); /* v_max3_i32 */
        h[r] = v;
        ug = v + gap;
        st.Hl[r] = ug;
        if constexpr (LOCAL) st.key[r] = max(st.key[r], ((unsigned)v << 16) | negj);
    }
    st.dtop = upinG;
}

/* pack the lane's R scores into R/2 dwords (the store format) */
template <int R, bool LOCAL>
__device__ __forceinline__ void lin_pack(LinState<R, LOCAL> &st, uint32_t (&w)[(R + 1) / 2]) {
#pragma unroll
    for (int q = 0; q < R / 2; q++) w[q] = pack_lo16(st.Hl[2 * q], st.Hl[2 * q + 1]);
}

template <int R>
__device__ __forceinline__ void store_words(int16_t *dst, const uint32_t (&w)[(R + 1) / 2]) {
    if constexpr (R == 2) {
        stream_store(reinterpret_cast<uint32_t *>(dst), w[0]);
    } else if constexpr (R == 4) {
        u32x2 v = {w[0], w[1]};
        stream_store(reinterpret_cast<u32x2 *>(dst), v);
    } else {
#pragma unroll
        for (int q = 0; q < R / 8; q++) {
            u32x4 v = {w[4 * q + 0], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            stream_store(reinterpret_cast<u32x4 *>(dst + q * 512), v); /* sub-tile q: its own 1 KiB [lane][8] block */
        }
    }
}

/* SW: fold the finished stripe's per-row keys into the lane's best (rows ascend with r and with the stripe index,
 * so a strict '>' keeps the first row holding the lane's maximum) */
template <int R, bool LOCAL>
__device__ __forceinline__ void lin_fold_keys(const LinState<R, LOCAL> &st, const int row0, const int nrows, int &bestv,
                                              int &bestrow, int &bestcol) {
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int hv = (int)(st.key[r] >> 16), col = 0xFFFF - (int)(st.key[r] & 0xFFFFu);
        if (r < nrows && hv > bestv) {
            bestv = hv;
            bestrow = row0 + 1 + r;
            bestcol = col;
        }
    }
}

/* WHOLE: store for every lane (the chunk is private to this stripe); otherwise only lanes on a real cell store.
 * SW tracks per-row (score, column) keys in registers, so the start cell needs no second pass over the matrix. */
template <int R, bool LOCAL, bool STORE, bool MASKED, bool WHOLE>
__device__ __forceinline__ void lin_step(LinState<R, LOCAL> &st, const int t, const int lane, const int n,
                                         const bool laneHasRows, const int matchG, const int mismatchG, const int gap,
                                         const int e0, const int rc, int16_t *edge, const bool writeEdge,
                                         int16_t *tileDst, const int storeLanes, const int rampLines) {
    /* the state is H + gap (lin_cells_g); e0 = the plain H of the row above lane 0 (edge row / row-0 border) */
    const int j = t - lane + 1;
    /* cross-lane traffic happens with all lanes enabled: a finished lane must still feed its neighbour */
    const int upin = wave_shr1(st.Hl[R - 1], e0 + gap);
    bool active = true;
    if constexpr (MASKED) active = laneHasRows && (j >= 1) && (j <= n);
    uint32_t w[(R + 1) / 2];
    if (active) {
        int h[R];
        lin_cells_g<R, LOCAL>(st, upin, rc, 0xFFFFu - (unsigned)j, matchG, mismatchG, gap, h);
        if (writeEdge && lane == 63) edge[j] = (int16_t)h[R - 1];
        if constexpr (STORE) {
#pragma unroll
            for (int q = 0; q < R / 2; q++) w[q] = pack_lo16(h[2 * q], h[2 * q + 1]);
            if constexpr (MASKED && !WHOLE) store_words<R>(tileDst, w);
        }
    } else if constexpr (STORE) {
        lin_pack<R, LOCAL>(st, w); /* (a lane that is not on a cell stores into padding that nothing reads: any value) */
    }
    /* Every lane stores, also lanes that are not on a real cell during the skew ramps: the wave then always writes
     * its whole 64*R*2-byte chunk.  Partial chunks (masked stores) measured ~2.5x the cost of full ones -- a chunk
     * with a partly written 64-B sector becomes a read-modify-write at the HBM.  The extra bytes land in the skew
     * padding of the pair's block, which nothing ever reads. */
    if constexpr (STORE && (!MASKED || WHOLE) && !(MASKED && DPX_EXP_NORAMPSTORE)) {
        bool doStore = lane < storeLanes; /* storeLanes == 64 unless the pair is shorter than the stripe */
        if constexpr (MASKED) doStore = doStore && ramp_stores<R>(lane, t, n, rampLines);
        if (doStore) store_words<R>(tileDst, w);
    }
}

template <int R, bool LOCAL, bool STORE>
__global__ void __launch_bounds__(DPX_FILL_THREADS) k_linear_fill(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int p = blockIdx.x * (int)a.wavesPerBlock + wv; /* wave-uniform: everything per-pair lives in SGPRs */
    if (p >= a.numPairs) return;
    if (a.order) p = a.order[p];
    const dpx_pair_dev pr = a.pairs[p];
    const int n = pr.n, m = pr.m;
    const int gap = a.gapOpen, matchG = a.match - gap, mismatchG = a.mismatch - gap; /* the lanes' state is H + gap (lin_cells_g) */

    if (m <= 0 || n <= 0) { /* empty sequence: only borders exist */
        if (lane == 0) {
            a.score[p] = LOCAL ? 0 : (m <= 0 ? n * gap : m * gap);
            a.endRow[p] = LOCAL ? 0 : max(m, 0);
            a.endCol[p] = LOCAL ? 0 : max(n, 0);
        }
        return;
    }
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);
    unsigned char *my = smem + (size_t)wv * a.ldsPerWave;
    int16_t *edge = reinterpret_cast<int16_t *>(my); /* edge[j], j = 0..n+1: H of the row above the current stripe */
    /* refl[64 + (j-1)], 64 bytes of slack either side; staged with 16-byte loads (stage_bytes) */
    const unsigned char *refl = stage_bytes(my + a.ldsRefOff + 64, ref, n, lane, 64) - 64;
    /* row-0 border (LinearNeedlemanWunsch.cpp:38-41; all zero for SW) is the first stripe's "row above" */
    for (int x = lane; x <= n + 1; x += 64) edge[x] = (int16_t)(LOCAL ? 0 : x * gap);

    int16_t *Hp = a.mat + pr.matOff;
    const size_t cs = pr.chunkStride;
    const int S = dpx_tiled_stripes(m, R);

    int bestv = 0, bestrow = 0, bestcol = 0;
    LinState<R, LOCAL> st;

    if (STORE && S >= 2 && n >= 128) { /* (score-only fills are VALU-bound: the plain striped loop is leaner there) */
        /* ---------- rolling schedule: a lane that finishes column n of its stripe starts column 1 of the next one
         * on the following step, so the skew ramp is paid once per pair instead of once per stripe and every
         * chunk between the two ramps is a whole one. ---------- */
        const unsigned char *ql = stage_bytes(my + a.ldsQryOff, qry, m, lane, 64); /* staged query: the stripe switch must not wait on global memory */
        int row0 = lane * R;
        int nrows = min(max(m - row0, 0), R);
        int jl = 1 - lane; /* this lane's column; <= 0: not started yet */
        int kl = 0;        /* this lane's stripe */
        load_query_rows<R>(st.qc, qry, row0, nrows);
#pragma unroll
        for (int r = 0; r < R; r++) {
            st.Hl[r] = (LOCAL ? 0 : (row0 + 1 + r) * gap) + gap;
            st.key[r] = 0u;
        }
        st.dtop = (LOCAL ? 0 : row0 * gap) + gap;
        int j0 = 1;      /* lane 0's column (wave-uniform): index of its `up` in the edge row */
        bool sw = false; /* this lane wrapped at the end of the previous step */
        int rcN = refl[63 + jl];
        int e0N = edge[1];
        int16_t *tile = Hp + (size_t)lane * (R < 8 ? R : 8);
        const int total = S * n + 63;
        auto roll_step = [&](const int T) {
            const int rc = rcN, e0 = e0N;
            const int jn = (jl >= n) ? 1 : jl + 1;
            rcN = refl[63 + jn];
            j0 = (j0 >= n) ? 1 : j0 + 1;
            e0N = edge[j0]; /* read n-64 steps after lane 63 wrote it, 63 steps before lane 63 overwrites it */
            /* the neighbour's bottom row must be taken BEFORE a switching lane resets its registers */
            const int upin = wave_shr1(st.Hl[R - 1], e0 + gap);
            if (sw) { /* stripe switch (one lane per step for 64 steps around each stripe boundary) */
                if constexpr (LOCAL) {
                    lin_fold_keys<R, LOCAL>(st, row0, nrows, bestv, bestrow, bestcol);
                }
                row0 += 64 * R;
                nrows = min(max(m - row0, 0), R);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    st.qc[r] = (r < nrows) ? (int)ql[row0 + r] : 0x100;
                    st.Hl[r] = (LOCAL ? 0 : (row0 + 1 + r) * gap) + gap;
                    st.key[r] = 0u;
                }
                st.dtop = (LOCAL ? 0 : row0 * gap) + gap;
            }
            uint32_t w[(R + 1) / 2];
            if (jl >= 1 && kl < S && nrows > 0) {
                int h[R];
                lin_cells_g<R, LOCAL>(st, upin, rc, 0xFFFFu - (unsigned)jl, matchG, mismatchG, gap, h);
                if (lane == 63 && kl + 1 < S) edge[jl] = (int16_t)h[R - 1];
                if constexpr (STORE) {
#pragma unroll
                    for (int q = 0; q < R / 2; q++) w[q] = pack_lo16(h[2 * q], h[2 * q + 1]);
                }
            } else if constexpr (STORE) {
                lin_pack<R, LOCAL>(st, w);
            }
            if constexpr (STORE) { /* whole chunk every step; on the pair's two ramps only the lines with cells (ramp_stores) */
                if (ramp_stores<R>(lane, T, S * n, a.rampLines)) store_words<R>(tile + (size_t)T * cs, w);
            }
            sw = false;
            if (jl >= n) { jl = 1; kl++; sw = kl < S; }
            else jl++;
        };
        int T = 0;
        for (; T + 1 < total; T += 2) { /* two steps per iteration: lets the register allocator ping-pong the H chain */
            roll_step(T);
            roll_step(T + 1);
        }
        if (T < total) roll_step(T);
        if constexpr (LOCAL) lin_fold_keys<R, LOCAL>(st, row0, nrows, bestv, bestrow, bestcol);
    } else {
        /* ---------- striped schedule (single stripe, or references too short to roll) ---------- */
        const int W = n + 63;
        for (int k = 0; k < S; k++) {
            const int base = k * 64 * R;
            const int row0 = base + lane * R; /* rows above the lane's first row */
            const int nrows = min(max(m - row0, 0), R);
            const bool laneHasRows = nrows > 0;
            const bool hasNext = (k + 1 < S);
            load_query_rows<R>(st.qc, qry, row0, nrows);
#pragma unroll
            for (int r = 0; r < R; r++) {
                st.Hl[r] = (LOCAL ? 0 : (row0 + 1 + r) * gap) + gap; /* column-0 border, LinearNeedlemanWunsch.cpp:31-34 (+ gap: lin_cells_g) */
                st.key[r] = 0u;
            }
            st.dtop = (LOCAL ? 0 : row0 * gap) + gap;
            int16_t *tile = Hp + (size_t)k * (size_t)n * cs + (size_t)lane * (R < 8 ? R : 8);

            /* software pipeline: the LDS reads of step t+1 (reference character of the lane's next column, and
             * lane 0's `up` from the edge row) are issued before the arithmetic of step t */
            const unsigned char *rp = refl + 64 - lane; /* rp[t] = reference character of column j = t - lane + 1 */
            int rcN = rp[0];
            int e0N = edge[1];
#define DPX_LIN_STEP(T_, MASKED_, WHOLE_, HASROWS_)                                                                    \
            {                                                                                                         \
                const int rc = rcN, e0 = e0N;                                                                         \
                rcN = rp[(T_) + 1];                                                                                   \
                e0N = edge[min((T_) + 2, n + 1)];                                                                     \
                lin_step<R, LOCAL, STORE, MASKED_, WHOLE_>(st, (T_), lane, n, HASROWS_, matchG, mismatchG, gap, e0, rc, edge, \
                                                           hasNext, tile + (size_t)(T_) * cs, storeLanes, a.rampLines); \
            }
            const bool fast = (base + 64 * R <= m) && (n >= 64);
            const int storeLanes = (S == 1) ? store_lanes<R>(m) : 64;
            if (S == 1) { /* the ramp chunks belong to this stripe alone: store them whole */
                if (fast) {
                    int t = 0;
                    for (; t < 63; t++) DPX_LIN_STEP(t, true, true, true)
                    for (; t + 1 < n; t += 2) { /* two steps per trip: the left/diagonal registers swap roles instead of moving */
                        DPX_LIN_STEP(t, false, true, true)
                        DPX_LIN_STEP(t + 1, false, true, true)
                    }
                    for (; t < n; t++) DPX_LIN_STEP(t, false, true, true)
                    for (; t < W; t++) DPX_LIN_STEP(t, true, true, true)
                } else {
                    for (int t = 0; t < W; t++) DPX_LIN_STEP(t, true, true, laneHasRows)
                }
            } else { /* several stripes of a short reference share their ramp chunks: masked stores there */
                for (int t = 0; t < W; t++) DPX_LIN_STEP(t, true, false, laneHasRows)
            }
#undef DPX_LIN_STEP
            if constexpr (LOCAL) lin_fold_keys<R, LOCAL>(st, row0, nrows, bestv, bestrow, bestcol);
        }
    }

    if constexpr (LOCAL) {
        /* first strict maximum in row-major order (c++/LinearSmithWaterman.cpp:145-157):
         * max score, then smallest row; then the smallest column of that row */
        const unsigned long long mine = ((unsigned long long)(unsigned)bestv << 32) | (unsigned)(0x7FFFFFFF - bestrow);
        const unsigned long long top = wave_max_u64(mine);
        if ((int)(top >> 32) == 0) {
            if (lane == 0) { a.score[p] = 0; a.endRow[p] = 0; a.endCol[p] = 0; }
        } else if (mine == top) { /* rows are unique per lane, so exactly one lane matches; it holds the row's first column */
            a.score[p] = bestv;
            a.endRow[p] = bestrow;
            a.endCol[p] = bestcol;
        }
    } else {
        /* score = H[m][n] (LinearNeedlemanWunsch.cpp:176): after the last stripe Hl[] holds column n */
        const int lastBase = (S - 1) * 64 * R;
        const int lm = (m - 1 - lastBase) / R, rm = (m - 1 - lastBase) % R;
        if (lane == lm) {
            int v = st.Hl[0];
#pragma unroll
            for (int r = 1; r < R; r++) v = (r == rm) ? st.Hl[r] : v;
            a.score[p] = v - gap; /* (the state is H + gap) */
            a.endRow[p] = m;
            a.endCol[p] = n;
        }
    }
}

/* =====================================================================================================
 * Lane-packed kernels for short and medium queries (the reference's own dataset shape: reads of 80-150 bases,
 * cuda/LNW V12 on bsw/small): SEVERAL pairs per wave.  A pair of m rows takes ceil(m/R) consecutive lanes (R = 8: 13
 * lanes for a 100-row query), the host packs pairs of similar reference length into the 64 lanes of a wave
 * (dpx_wave_desc: up to 8 slots), so nearly every lane owns rows -- one pair per wave kept 13 (or 25) of 64 lanes
 * busy, round 1's four-per-wave quad kernels 13 of 16.  Lane l of a slot owns rows [l*R, l*R+R) and runs column
 * j = t - l - d + 1 in step t (d = the slot's first lane mod 8, see below); `up` moves with v_mov_b32_dpp wave_shr:1,
 * a slot's first lane takes the row-0 border instead; n, m, pointers are per-lane values, the wave runs
 * max(n + lanes + d) steps and every lane masks itself.
 *
 * Writeback: 8 x 8 tile layout (dpx_layout.h), one whole 128-byte line per (row block, column block), through an LDS
 * transpose.  Per step a lane parks its R new scores -- 16 B per row block -- in its own LDS line (ds_write_b128); a
 * lane's line is complete when its column index reaches a multiple of 8.  With the delay d that happens for the lanes
 * lambda = (t+1) mod 8 of every 8-lane group in step t, whatever slots they belong to: eight lines, read back so that
 * lane x of the wave holds piece x%8 of the line of lane 8*(x/8) + (t+1)%8 (ds_read_b128) and written by ONE
 * global_store_dwordx4 of eight whole lines.  The fill is bound by store INSTRUCTIONS (about 104 cycles per
 * global_store_dwordx4 and CU whether 64 or 48 lanes are active, profiles/README.md), so what counts is that every
 * store instruction carries eight lines of real cells: no skew-ramp padding, rows rounded up to 8 instead of 64/128
 * (round 1's [step][lane][rows] chunks wrote 1.40x the algorithmic bytes on short reads), no idle lanes.
 * Where a line goes (pointer, column-block stride, the owner's skew and last column block) sits in the 16 spare bytes
 * of the owner's LDS line; the reading lanes pick it up one step ahead.  LDS lines are 144 bytes apart and column c
 * of lane lambda's line sits in slot (c + lambda) % 8 -- which is t % 8 for every lane of the wave in step t, the skew
 * cancels: conflict-free for the b128 write groups (8 contiguous lanes, banks 4*lambda + const) and for the b128 read
 * groups of MI355X_MICROARCH.md (4 x 16 lanes).  The store of the lines read in step t is issued in step t+1, behind
 * that step's arithmetic.
 * ===================================================================================================== */
constexpr int kStageLine = 144; /* bytes between two LDS lines: 128 of cells + 16 of routing (StagePad) */
constexpr int kLaneScratch = 1024; /* per wave: 16 B per lane for the end-of-pair reduction (aliases the line stage, which is dead by then) */

template <int Q, int PLANES>
struct LineStage {
    static constexpr int kBytes = PLANES * Q * 64 * kStageLine; /* per wave */
    static constexpr int kStepElems = PLANES * Q * 512;         /* int16 elements a wave stores per step (dpx_layout.h) */
    u32x4 pend[PLANES * Q];
    int16_t *pendDst = nullptr; /* this lane's 16 bytes of (plane 0, sub-tile 0) in the step's chunk; plane p, sub-tile h: + (p*Q + h)*512 elements */
    int pendN = 0;              /* sub-tiles of the owner lane that hold rows (0: nothing to store) */
    uint32_t route = 0;         /* routing word of the owner whose lines complete in the NEXT step */

    /* routing word of a lane, written once into the 16 spare bytes of its LDS line: skew | n8 << 8 | valid sub-tiles << 28 */
    static __device__ __forceinline__ void set_route(unsigned char *tile, int lane, int skew, int n8, int nValid) {
        *reinterpret_cast<uint32_t *>(tile + lane * kStageLine + 128) = (uint32_t)skew | ((uint32_t)n8 << 8) | ((uint32_t)nValid << 28);
    }
    /* park the lane's 8 rows of sub-tile h, plane p, in slot t % 8 of its line (t = the wave's step) */
    static __device__ __forceinline__ void put(unsigned char *tile, int lane, int p, int h, int t, u32x4 v) {
#if DPX_EXP_QUAD == 2
        if (t == -5)
#endif
        *reinterpret_cast<u32x4 *>(tile + ((p * Q + h) * 64 + lane) * kStageLine + ((t & 7) << 4)) = v;
    }
    /* write out what fetch() read one step ago */
    __device__ __forceinline__ void store() {
#if DPX_EXP_QUAD == 1
        if (pendN == 77) /* never */
#endif
#pragma unroll
        for (int h = 0; h < Q; h++) {
            if (h < pendN) {
#pragma unroll
                for (int p = 0; p < PLANES; p++) stream_store(reinterpret_cast<u32x4 *>(pendDst + ((p * Q + h) << 9)), pend[p * Q + h]);
            }
        }
    }
    __device__ __forceinline__ void fetch_route(const unsigned char *tile, int lane, int tNext) {
        route = *reinterpret_cast<const uint32_t *>(tile + ((lane & ~7) | ((tNext + 1) & 7)) * kStageLine + 128);
    }
    /* step t: read piece lane%8 of the lines that completed in this step (owner = lane (t+1)%8 of this lane's 8-group), valid
     * if the owner's column t + 1 - skew is one of its column-block ends (routing word fetched during the previous step);
     * they go to chunk t of the wave's stream, at this lane's 16 bytes */
    __device__ __forceinline__ void fetch(const unsigned char *tile, int16_t *waveBase, int lane, int t) {
        const int owner = (lane & ~7) | ((t + 1) & 7);
        const int jg = t + 1 - (int)(route & 0xFFu); /* owner's column: a multiple of 8 */
        const int n8 = (int)((route >> 8) & 0xFFFFFu);
        pendN = (jg >= 8 && jg <= n8) ? (int)(route >> 28) : 0;
        pendDst = waveBase + (size_t)t * kStepElems + (lane << 3);
#pragma unroll
        for (int p = 0; p < PLANES; p++)
#pragma unroll
            for (int h = 0; h < Q; h++)
#if DPX_EXP_QUAD == 2
                pend[p * Q + h].x += (uint32_t)owner;
#else
                pend[p * Q + h] = *reinterpret_cast<const u32x4 *>(tile + ((p * Q + h) * 64 + owner) * kStageLine + (((lane + owner) & 7) << 4));
#endif
    }
};

/* what a lane learns from the wave's descriptor */
struct LaneSlot {
    bool has;
    int p, l, num, d; /* pair, lane inside the slot, lanes of the slot, start delay (first lane mod 8) */
    uint32_t refOff;  /* LDS byte offset of the slot's staged reference inside the wave's reference area */
};
__device__ __forceinline__ LaneSlot find_slot(const dpx_wave_desc *wd, const int lane) {
    LaneSlot s{false, 0, 0, 0, 0, 0u};
    const uint32_t *w = reinterpret_cast<const uint32_t *>(wd); /* wave-uniform address: 2 dwords per slot through scalar loads */
    static_assert(DPX_WAVE_SLOTS % 4 == 0 && sizeof(dpx_wave_desc) == 8 * DPX_WAVE_SLOTS, "descriptor is read as dwords");
    constexpr int kFirst = DPX_WAVE_SLOTS, kNum = DPX_WAVE_SLOTS + DPX_WAVE_SLOTS / 4, kRef = DPX_WAVE_SLOTS + DPX_WAVE_SLOTS / 2; /* dword offsets */
#pragma unroll
    for (int k = 0; k < DPX_WAVE_SLOTS; k++) {
        const int f = (int)((w[kFirst + (k >> 2)] >> (8 * (k & 3))) & 0xFFu), c = (int)((w[kNum + (k >> 2)] >> (8 * (k & 3))) & 0xFFu);
        const uint32_t ro = (w[kRef + (k >> 1)] >> (16 * (k & 1))) & 0xFFFFu;
        if (lane >= f && lane < f + c) { s.has = true; s.p = (int)w[k]; s.l = lane - f; s.num = c; s.d = f & 7; s.refOff = ro << 4; }
    }
    return s;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

template <int R, bool LOCAL, bool STORE>
__global__ void __launch_bounds__(DPX_FILL_THREADS) k_linear_lanes(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int Q = R / 8;
    using Stage = LineStage<Q, 1>;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x * (int)a.wavesPerBlock + wv;
    if (w >= a.numPairs) return; /* wave-uniform; numPairs = number of wave descriptors */
    const LaneSlot sl = find_slot(a.waves + w, lane);
    const bool has = sl.has;
    const int p = sl.p, l = sl.l;
    const dpx_pair_dev pr = a.pairs[p];
    const int n = has ? pr.n : 0, m = has ? pr.m : 0;
    const int gap = a.gapOpen, matchG = a.match - gap, mismatchG = a.mismatch - gap;
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);

    unsigned char *tileL = smem + (size_t)wv * a.ldsPerWave;      /* [line stage | lane scratch][staged references] */
    unsigned char *scratch = tileL;
    unsigned char *refl = tileL + (STORE ? Stage::kBytes : kLaneScratch) + sl.refOff;
    const unsigned char *refs = stage_bytes(refl, ref, n, l, max(sl.num, 1));

    const int row0 = l * R;
    const int nrows = min(max(m - row0, 0), R);
    LinState<R, LOCAL> st; /* Hl = H + gap, dtop = H[row0][j-1] + gap (lin_cells_g) */
    load_query_rows<R>(st.qc, qry, row0, nrows);
#pragma unroll
    for (int r = 0; r < R; r++) {
        st.Hl[r] = (LOCAL ? 0 : (row0 + 1 + r) * gap) + gap;
        st.key[r] = 0u;
    }
    st.dtop = (LOCAL ? 0 : row0 * gap) + gap;

    const int skew = l + sl.d; /* this lane runs column j = t - skew + 1 in step t */
    const int n8 = (n + 7) & ~7;
    const int LB = (int)dpx_tile8_row_blocks(m);
    /* Routing.  A lane's line (8 columns x 8 rows, 128 B) is complete in the steps t = skew + 7, skew + 15, ... <= n8 + skew - 1, and
     * skew = lane (mod 8): in step t the lines of the lanes (t+1) mod 8 of every 8-lane group are complete, whatever pairs they
     * belong to.  Every lane publishes (first step | last step << 16) and the number of its row blocks that hold rows once, in the
     * 16 spare bytes of its LDS line; every lane then keeps the words of the eight lanes of its group in registers -- the loop reads
     * no routing from LDS (round 2 fetched one word per step) and decides with two compares. */
    uint32_t rt[8], nv[8];
    if constexpr (STORE) {
        const int nValid = has ? min(max(LB - l * Q, 0), Q) : 0;
        uint32_t *mine = reinterpret_cast<uint32_t *>(tileL + lane * kStageLine + 128);
        mine[0] = nValid > 0 ? ((uint32_t)(skew + 7) | ((uint32_t)(n8 + skew - 1) << 16)) : 0x00007FFFu; /* (never valid: first > last) */
        mine[1] = (uint32_t)nValid;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t *o = reinterpret_cast<const uint32_t *>(tileL + ((lane & ~7) | k) * kStageLine + 128);
            rt[k] = o[0];
            nv[k] = o[1];
        }
    }
    /* the wave's stream of chunks (dpx_layout.h): every pair of the wave carries the same base; lane 0 always belongs to slot 0 */
    int16_t *waveBase = a.mat + (size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pr.matOff >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)(pr.matOff & 0xFFFFFFFFull)));
    const int steps = __builtin_amdgcn_readfirstlane(wave_max_i32(has ? (STORE ? n8 : n) + skew : 0)); /* uniform: the loop runs on the scalar unit */
    const unsigned char *rp = refs - skew; /* rp[t] = reference character of column j = t - skew + 1 */
    const unsigned nEff = nrows > 0 ? (unsigned)n : 0u; /* the lane is on a cell iff (unsigned)(t - skew) < nEff */
    const int rpLast = n + skew;                        /* rp[rpLast] = refs[n]: inside the 31 bytes of slack of stage_bytes */
    unsigned char *putPtr = tileL + lane * kStageLine;             /* + sub-tile * 64 lines + (t & 7) * 16 */
    const unsigned char *fetchPtr[8];                              /* piece lane % 8 of the line of lane k of this lane's group, rotated by its owner */
#pragma unroll
    for (int k = 0; k < 8; k++) fetchPtr[k] = tileL + ((lane & ~7) | k) * kStageLine + (((lane + k) & 7) << 4);
    u32x4 pend[Q];
    int pendN = 0;
    int16_t *pendDst = nullptr;
    int bordG = (2 - skew) * gap; /* NW, first lane of a slot: H[0][j] + gap = (j + 1) * gap, j = t - skew + 1 */
    int rcN = rp[0];
    auto flush = [&]() __attribute__((always_inline)) { /* write out the lines read one step ago */
#if DPX_EXP_QUAD == 1 /* ablation build (tools/ab_quad.sh): no global stores */
        if (pendN == 77)
#endif
#pragma unroll
        for (int hq = 0; hq < Q; hq++)
            if (hq < pendN) stream_store(reinterpret_cast<u32x4 *>(pendDst + (hq << 9)), pend[hq]);
    };
    auto lane_step = [&](const int t, auto kTag) __attribute__((always_inline)) {
        constexpr int K = decltype(kTag)::value; /* t & 7 */
        const int tms = t - skew;
        const int rc = rcN;
        rcN = rp[min(t + 1, rpLast)]; /* (never past the slot's staged reference: a shorter pair's lane idles to the end of the wave) */
        const int sh = wave_shr1(st.Hl[R - 1], 0);
        const int upinG = (l == 0) ? (LOCAL ? gap : bordG) : sh; /* a slot's first lane: row-0 border of its column (+ gap) */
        if constexpr (!LOCAL) bordG += gap;
        if ((unsigned)tms < nEff) {
            int h[R];
            lin_cells_g<R, LOCAL>(st, upinG, rc, 0xFFFEu - (unsigned)tms, matchG, mismatchG, gap, h);
            if constexpr (STORE) {
#pragma unroll
                for (int hq = 0; hq < Q; hq++) {
                    u32x4 v = {pack_lo16(h[8 * hq + 0], h[8 * hq + 1]), pack_lo16(h[8 * hq + 2], h[8 * hq + 3]),
                               pack_lo16(h[8 * hq + 4], h[8 * hq + 5]), pack_lo16(h[8 * hq + 6], h[8 * hq + 7])};
#if DPX_EXP_QUAD != 2
                    *reinterpret_cast<u32x4 *>(putPtr + hq * 64 * kStageLine + (K << 4)) = v;
#else
                    pend[hq].y += v.x;
#endif
                }
            }
        }
        if constexpr (STORE) {
            flush();
            constexpr int O = (K + 1) & 7; /* the owners of the lines that are complete now */
            const uint32_t r = rt[O];
            const bool valid = (uint32_t)t >= (r & 0xFFFFu) && (uint32_t)t <= (r >> 16);
            pendN = valid ? (int)nv[O] : 0;
#if DPX_EXP_QUAD == 3 /* ablation build: every store instruction full (invalid lines written too): bytes + 50 %, same instruction count */
            pendN = Q;
#endif
            pendDst = waveBase + (size_t)t * Stage::kStepElems + (lane << 3);
#pragma unroll
            for (int hq = 0; hq < Q; hq++)
#if DPX_EXP_QUAD == 2 /* ablation build: stores without the LDS round trip */
                pend[hq].x += (uint32_t)t;
#else
                pend[hq] = *reinterpret_cast<const u32x4 *>(fetchPtr[O] + hq * 64 * kStageLine);
#endif
        }
    };
    {
        int t = 0;
        for (; t + 8 <= steps; t += 8) { /* eight steps per trip: every LDS address of the line stage is a register + an immediate */
            lane_step(t + 0, std::integral_constant<int, 0>{}); lane_step(t + 1, std::integral_constant<int, 1>{});
            lane_step(t + 2, std::integral_constant<int, 2>{}); lane_step(t + 3, std::integral_constant<int, 3>{});
            lane_step(t + 4, std::integral_constant<int, 4>{}); lane_step(t + 5, std::integral_constant<int, 5>{});
            lane_step(t + 6, std::integral_constant<int, 6>{}); lane_step(t + 7, std::integral_constant<int, 7>{});
        }
        if (t + 0 < steps) lane_step(t + 0, std::integral_constant<int, 0>{});
        if (t + 1 < steps) lane_step(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < steps) lane_step(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < steps) lane_step(t + 3, std::integral_constant<int, 3>{});
        if (t + 4 < steps) lane_step(t + 4, std::integral_constant<int, 4>{});
        if (t + 5 < steps) lane_step(t + 5, std::integral_constant<int, 5>{});
        if (t + 6 < steps) lane_step(t + 6, std::integral_constant<int, 6>{});
    }
    if constexpr (STORE) flush();
    if constexpr (LOCAL) {
        /* first strict maximum in row-major order over the slot's lanes (rows ascend with the lane): the slot's first lane
         * scans its lanes' (score, row, column) in the lane scratch */
        int bestv = 0, bestrow = 0, bestcol = 0;
        lin_fold_keys<R, LOCAL>(st, row0, nrows, bestv, bestrow, bestcol);
        int *mine = reinterpret_cast<int *>(scratch + lane * 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); /* (the line stage this aliases has been read to the end) */
        __builtin_amdgcn_wave_barrier();
        mine[0] = bestv; mine[1] = bestrow; mine[2] = bestcol;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); /* the other lanes' entries are read below: keep the order */
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (has && l == 0) {
            for (int k = 1; k < sl.num; k++) {
                const int *o = reinterpret_cast<const int *>(scratch + (lane + k) * 16);
                if (o[0] > bestv) { bestv = o[0]; bestrow = o[1]; bestcol = o[2]; }
            }
            a.score[p] = bestv; a.endRow[p] = bestv > 0 ? bestrow : 0; a.endCol[p] = bestv > 0 ? bestcol : 0;
        }
    } else {
        const int lm = (m - 1) / R, rm = (m - 1) % R; /* owner of row m: its registers hold column n after its last step */
        if (has && l == lm) {
            int v = st.Hl[0];
#pragma unroll
            for (int r = 1; r < R; r++) v = (r == rm) ? st.Hl[r] : v;
            a.score[p] = v - gap; a.endRow[p] = m; a.endCol[p] = n; /* (the state is H + gap) */
        }
    }
}

/* =====================================================================================================
 * Split kernel for SMALL batches (BASELINE configs[1]: 1000 pairs of 512 x 512 -- one wave per pair leaves the chip with
 * one wave per SIMD, and a lone wave issues a vector instruction only every 4 cycles).  One workgroup per pair, one WAVE
 * PER STRIPE: wave w fills rows [w*64R, (w+1)*64R) with R = 4 (or 2), all stripes at the same time, wave w+1 running
 * ~90 columns behind wave w and taking its "row above" from the LDS edge row wave w's lane 63 writes -- the stripe
 * hand-off of cuda/LNW/LinearNeedlemanWunschV12.cu:110-113,141-143 (warpEdgeScore), made concurrent.  Progress is
 * published through LDS every 8 columns (release) and checked every 16 (acquire).  1000 pairs become 2000-4000 waves
 * with half (a quarter) of the dependent chain per step.  Stores stay 16 bytes per lane: G = 8/R steps are collected in
 * registers (dpx_layout.h, split layout); every stripe owns its chunks, so ramp chunks are written whole.
 * ===================================================================================================== */
#define DPX_SPLIT_MAX_WAVES 16
template <int R, bool LOCAL>
__global__ void __launch_bounds__(64 * DPX_SPLIT_MAX_WAVES) k_linear_split(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int G = 8 / R; /* steps per 16-byte store */
    static_assert(R == 2 || R == 4 || R == 8, "rows per lane");
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int p = blockIdx.x;
    if (a.order) p = a.order[p];
    const dpx_pair_dev pr = a.pairs[p];
    const int n = pr.n, m = pr.m; /* host guarantees m > 0, n > 0 */
    const int gap = a.gapOpen, matchG = a.match - gap, mismatchG = a.mismatch - gap; /* the state is H + gap (lin_cells_g) */
    const int W = dpx_tiled_stripes(m, R);
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);
    /* LDS: [control: progress[16], done, pad | per-wave results 16 x 16 B][staged reference][edge rows, one per stripe boundary] */
    int *ctrl = reinterpret_cast<int *>(smem);
    int *result = ctrl + 32;
    if (threadIdx.x < 32) ctrl[threadIdx.x] = 0;
    const unsigned char *refl = stage_bytes(smem + a.ldsRefOff + 64, ref, n, (int)threadIdx.x, (int)blockDim.x) - 64; /* refl[64 + (j-1)] */
    __syncthreads(); /* every wave of the workgroup is still here; surplus waves leave afterwards */
    if (w >= W) return;
    int16_t *edgeMine = reinterpret_cast<int16_t *>(smem + a.ldsQryOff) + (size_t)w * a.ldsBufStride;       /* bottom row of this stripe */
    const int16_t *edgePrev = edgeMine - a.ldsBufStride;                                                        /* ... of the stripe above */

    const int row0 = w * 64 * R + lane * R;
    const int nrows = min(max(m - row0, 0), R);
    LinState<R, LOCAL> st;
    load_query_rows<R>(st.qc, qry, row0, nrows);
#pragma unroll
    for (int r = 0; r < R; r++) {
        st.Hl[r] = (LOCAL ? 0 : (row0 + 1 + r) * gap) + gap;
        st.key[r] = 0u;
    }
    st.dtop = (LOCAL ? 0 : row0 * gap) + gap;
    const bool hasNext = w + 1 < W;
    const int rowsHere = min(m - w * 64 * R, 64 * R);
    const int storeLanes = min(64, (((rowsHere + R - 1) / R) + 7) & ~7); /* row-owning lanes, rounded up to whole 128-byte lines */
    const uint32_t SS = dpx_split_stripe_steps(n, R); /* n + 63 rounded up to whole 16-step blocks */
    const size_t cs = pr.chunkStride;
    int16_t *tile = a.mat + pr.matOff + ((size_t)w * (SS / G)) * cs + (size_t)lane * 8;
    const unsigned char *rp = refl + 64 - lane; /* rp[t] = reference character of column j = t - lane + 1 */
    auto wait_for = [&](const int need) {       /* wave-uniform: the stripe above has published `need` columns of its bottom row */
        while (__hip_atomic_load(&ctrl[w - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
    };
    /* lane 63 writes the stripe's bottom row; the other lanes write the same instruction's worth into a 32-byte dump, so the
     * edge write costs one ds_write_b16 per step and no exec juggling */
    int16_t *dump = reinterpret_cast<int16_t *>(smem + 384) + 16 * 0;
    uint32_t acc[4];
    int hv[R]; /* the plain scores of the lane's last column (outside the ramps every lane has one every step) */
#pragma unroll
    for (int r = 0; r < R; r++) hv[r] = st.Hl[r] - gap;
    /* one block of 16 steps; MASKED: the skew ramps, where a lane may be before column 1 or past column n */
    auto block16 = [&](const int tb, auto maskedTag) {
        constexpr bool MASKED = decltype(maskedTag)::value;
        if (w > 0) wait_for(min(tb + 17, n));
        int16_t *ew = (hasNext && lane == 63) ? edgeMine + (tb - 62) : dump; /* ew[g] = edgeMine[j] of step tb + g */
        const unsigned char *rpb = rp + tb;
        const int16_t *epb = edgePrev + tb;
        int rcN = rpb[0];
        int e0N = ((w == 0) ? (LOCAL ? 0 : (tb + 1) * gap) : (int)epb[1]) + gap; /* "up" of lane 0, + gap like every state value */
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int t = tb + g;
            const int rc = rcN, e0 = e0N;
            rcN = rpb[g + 1];
            e0N = ((w == 0) ? (LOCAL ? 0 : (t + 2) * gap) : (int)epb[g + 2]) + gap; /* (entries past n are never used) */
            const int upin = wave_shr1(st.Hl[R - 1], e0);
            const int j = t - lane + 1;
            const unsigned negj = 0xFFFFu - (unsigned)j;
            if constexpr (MASKED) {
                if (j >= 1 && j <= n) {
                    lin_cells_g<R, LOCAL>(st, upin, rc, negj, matchG, mismatchG, gap, hv);
                    if (hasNext && lane == 63) edgeMine[j] = (int16_t)hv[R - 1];
                }
            } else {
                lin_cells_g<R, LOCAL>(st, upin, rc, negj, matchG, mismatchG, gap, hv);
                ew[g] = (int16_t)hv[R - 1];
            }
            if constexpr (R == 8) {
                acc[0] = pack_lo16(hv[0], hv[1]); acc[1] = pack_lo16(hv[2], hv[3]);
                acc[2] = pack_lo16(hv[4], hv[5]); acc[3] = pack_lo16(hv[6], hv[7]);
            } else if constexpr (R == 4) {
                acc[2 * (g % G)] = pack_lo16(hv[0], hv[1]); acc[2 * (g % G) + 1] = pack_lo16(hv[2], hv[3]);
            } else {
                acc[g % G] = pack_lo16(hv[0], hv[1]);
            }
            if ((g % G) == G - 1 && t - (G - 1) < n + 63) { /* whole lines, also on the skew ramps (see lin_step); nothing past the last step */
                /* (no ramp_stores() here: this kernel runs small batches, which are bound by the latency of a step, not by bytes --
                 * line-rounded ramp masks measured 8 % SLOWER on 1000 x 512 x 512, tools/ab_ramp.sh) */
                const bool doStore = lane < storeLanes;
                if (doStore) {
                    u32x4 v = {acc[0], acc[1], acc[2], acc[3]};
                    stream_store(reinterpret_cast<u32x4 *>(tile + (size_t)(t / G) * cs), v);
                }
            }
            /* publish the bottom row's progress every 8 columns (lane 63 ran column t - 62 in this step) */
            if ((g & 7) == 7 && hasNext && t >= 63) {
                if (lane == 63) __hip_atomic_store(&ctrl[w], min(t - 62, n), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    {
        int tb = 0;
        const int steps = (int)SS;
        for (; tb < steps && tb < 64; tb += 16) block16(tb, std::true_type{});
        for (; tb + 16 <= n; tb += 16) block16(tb, std::false_type{}); /* every lane inside [1, n] for all 16 steps */
        for (; tb < steps; tb += 16) block16(tb, std::true_type{});
    }
    if (hasNext && lane == 63) __hip_atomic_store(&ctrl[w], n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);

    /* results: every wave leaves its candidate in LDS; the wave that arrives last combines them */
    int bestv = 0, bestrow = 0, bestcol = 0;
    if constexpr (LOCAL) {
        lin_fold_keys<R, LOCAL>(st, row0, nrows, bestv, bestrow, bestcol);
        const unsigned long long mine = ((unsigned long long)(unsigned)bestv << 32) | (unsigned)(0x7FFFFFFF - bestrow);
        const unsigned long long top = wave_max_u64(mine);
        if (mine == top && (bestv > 0 ? true : lane == 0)) { result[4 * w] = bestv; result[4 * w + 1] = bestrow; result[4 * w + 2] = bestcol; }
    } else {
        const int lm = (m - 1 - w * 64 * R) / R, rm = (m - 1) % R; /* owner of row m (in the last stripe) */
        if (w == W - 1 && lane == lm) {
            int v = st.Hl[0];
#pragma unroll
            for (int r = 1; r < R; r++) v = (r == rm) ? st.Hl[r] : v;
            a.score[p] = v - gap; a.endRow[p] = m; a.endCol[p] = n; /* (the state is H + gap) */
        }
    }
    if constexpr (LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(&ctrl[16], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived == W - 1 && lane == 0) { /* first strict maximum in row-major order: max score, then the first stripe holding it */
            int bv = 0, br = 0, bc = 0;
            for (int k = 0; k < W; k++) {
                const int v = __hip_atomic_load(&result[4 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v > bv) { bv = v; br = result[4 * k + 1]; bc = result[4 * k + 2]; }
            }
            a.score[p] = bv; a.endRow[p] = bv > 0 ? br : 0; a.endCol[p] = bv > 0 ? bc : 0;
        }
    }
}

/* =====================================================================================================
 * "+Opt": packed two-pairs-per-wave linear fill (the reference's V18/V19 idea, cuda/LNW/LinearNeedlemanWunschV18.cu:
 * 112-341 and the unfinished cuda/LinearSmithWatermanOpt.cu: one warp computes two pairs with __vibmax_s16x2).
 * Every VGPR holds pair A in its high half and pair B in its low half (V19.cu:16-24); the cell update runs on the
 * VOP3P packed-int16 pipe: v_xor (chars), v_pk_min_u16 (0/1 "differs"), v_pk_mad_i16 (s = differs*(mismatch-match)+match),
 * v_pk_max_i16 / v_pk_add_i16.  One DPP move carries both pairs to the next lane.  Both pairs must have the same
 * (m, n); the host couples equal-shaped pairs and sends leftovers to k_linear_fill.  Matrices are written in the
 * same per-pair wavefront-tiled layout, so export and traceback do not care which kernel filled a pair.
 * SW start cell: tracked exactly in the loop, per half: a packed running row maximum and the column at which it was
 * first reached (v_pk_max_u16 / v_pk_sub_u16 / v_pk_min_u16 / v_pk_sub_u16 / v_bfi_b32, see PkState).
 * ===================================================================================================== */
template <int R>
struct PkState {
    uint32_t Hl[R];   /* packed H[row][j-1] */
    uint32_t qc[R];   /* packed query characters (16-bit lanes) */
    uint32_t rmax[R]; /* SW: per-row running maximum of pair A's key (H << 16 | 0xFFFF - column): max score, then smallest column */
    uint32_t rcol[R]; /* SW: the same for pair B */
    uint32_t dtop;
    uint32_t keyA, keyB; /* SW, TAGS: ONE key per pair and lane, (H * R + (R-1 - row in lane)) << 16 | 0xFFFF - column: max score, then
                            smallest row, then smallest column -- the row-major "first strict maximum" of c++/LinearSmithWaterman.cpp:145-157 */
};

/* per half: H * R + tag on the VOP3P pipe (v_pk_mad_u16 with two inline constants: no register for either) */
template <int R, int TAG>
__device__ __forceinline__ uint32_t pk_row_tag(uint32_t h) {
    static_assert(TAG >= 0 && TAG < R && R <= 16, "inline constants");
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "n"(R), "n"(TAG));
    return r;
}
/* row r of the lane (a constant once the row loop is unrolled: the switch folds away) carries tag R-1 - r */
template <int R>
__device__ __forceinline__ uint32_t pk_row_tag_of(uint32_t h, const int r) {
#define DPX_TAG_CASE(k) case k: if constexpr (k < R) return pk_row_tag<R, (k < R ? R - 1 - k : 0)>(h); else return 0u;
    switch (r) {
        DPX_TAG_CASE(0) DPX_TAG_CASE(1) DPX_TAG_CASE(2) DPX_TAG_CASE(3) DPX_TAG_CASE(4) DPX_TAG_CASE(5) DPX_TAG_CASE(6) DPX_TAG_CASE(7)
        DPX_TAG_CASE(8) DPX_TAG_CASE(9) DPX_TAG_CASE(10) DPX_TAG_CASE(11) DPX_TAG_CASE(12) DPX_TAG_CASE(13) DPX_TAG_CASE(14) DPX_TAG_CASE(15)
    default: return 0u;
    }
#undef DPX_TAG_CASE
}

__device__ __forceinline__ uint32_t pk_hi16(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); } /* {lo.hi16, hi.hi16} */

/* SW with gap <= 0 (round 3): max(up, left) - |gap| SATURATING at 0 (v_pk_sub_u16 ... clamp; SW cells are never negative, so the signed
 * maximum of two of them is their unsigned one).  The gap term is then >= 0, so H = max(gap term, diag + s) >= 0 without the extra
 * v_pk_max_i16(H, 0) of c++/LinearSmithWaterman.cpp:103 -- the SW cell costs what the NW cell costs.  The host sends a batch to the
 * kernels that use this only if gap <= 0. */
__device__ __forceinline__ s16x2 pk_gap_sat(const uint32_t up, const uint32_t left, const uint32_t gabsP) {
    return as_s16x2(as_u32(__builtin_elementwise_sub_sat(as_u16x2(as_u32(dpx::pk_max(as_s16x2(up), as_s16x2(left)))), as_u16x2(gabsP))));
}

template <int R>
__device__ __forceinline__ void store_tile_pk(int16_t *dstA, int16_t *dstB, const uint32_t (&v)[R]) {
    /* pair A = high halves, pair B = low halves; R/2 dwords each */
    if constexpr (R == 2) {
        *reinterpret_cast<uint32_t *>(dstA) = pk_hi16(v[0], v[1]);
        *reinterpret_cast<uint32_t *>(dstB) = pack_lo16((int)v[0], (int)v[1]);
    } else if constexpr (R == 4) {
        uint2 a, b;
        a.x = pk_hi16(v[0], v[1]); a.y = pk_hi16(v[2], v[3]);
        b.x = pack_lo16((int)v[0], (int)v[1]); b.y = pack_lo16((int)v[2], (int)v[3]);
        *reinterpret_cast<uint2 *>(dstA) = a;
        *reinterpret_cast<uint2 *>(dstB) = b;
    } else {
#pragma unroll
        for (int q = 0; q < R / 8; q++) {
            uint4 a, b;
            a.x = pk_hi16(v[8 * q + 0], v[8 * q + 1]); a.y = pk_hi16(v[8 * q + 2], v[8 * q + 3]);
            a.z = pk_hi16(v[8 * q + 4], v[8 * q + 5]); a.w = pk_hi16(v[8 * q + 6], v[8 * q + 7]);
            b.x = pack_lo16((int)v[8 * q + 0], (int)v[8 * q + 1]); b.y = pack_lo16((int)v[8 * q + 2], (int)v[8 * q + 3]);
            b.z = pack_lo16((int)v[8 * q + 4], (int)v[8 * q + 5]); b.w = pack_lo16((int)v[8 * q + 6], (int)v[8 * q + 7]);
            *reinterpret_cast<uint4 *>(dstA + q * 512) = a; /* sub-tile q */
            *reinterpret_cast<uint4 *>(dstB + q * 512) = b;
        }
    }
}

template <int R, bool LOCAL, bool MASKED, bool WHOLE, bool TAGS = false, bool PARTIAL = false>
__device__ __forceinline__ void pk_step(PkState<R> &st, const int t, const int lane, const int n, const bool laneHasRows, const int nrows,
                                        const uint32_t matchP, const uint32_t negDeltaP, const uint32_t gapP, const uint32_t e0,
                                        const uint32_t rcP, uint32_t *edge, const bool writeEdge, int16_t *tileA, int16_t *tileB,
                                        const int storeLanes, const int rampLines) {
    const int j = t - lane + 1;
    const uint32_t upin = (uint32_t)wave_shr1((int)st.Hl[R - 1], (int)e0);
    bool active = true;
    if constexpr (MASKED) active = laneHasRows && (j >= 1) && (j <= n);
    if (active) {
        uint32_t u = upin, d = st.dtop;
        const uint32_t onesP = 0x00010001u;
        const uint32_t negj = 0xFFFFu - (uint32_t)j;
        uint32_t colMax = 0u; /* TAGS: per half, max over the lane's rows of this column of (H * R + R-1 - r) */
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t left = st.Hl[r];
            const uint32_t differs = dpx::pk_min_u16_raw(st.qc[r] ^ rcP, onesP);                 /* 0 / 1 per half */
            const s16x2 sc = as_s16x2(dpx::pk_mad_i16_raw(differs, negDeltaP, matchP));         /* match or mismatch per half */
            s16x2 h;
            if constexpr (LOCAL && TAGS) { /* (rowTags batches have gap <= 0: gapP holds |gap| here) */
                h = dpx::pk_max(pk_gap_sat(u, left, gapP), (s16x2)(as_s16x2(d) + sc));
            } else {
                const s16x2 g = dpx::pk_max(as_s16x2(u), as_s16x2(left)) + as_s16x2(gapP);
                h = dpx::pk_max(g, (s16x2)(as_s16x2(d) + sc));
                if constexpr (LOCAL) h = dpx::pk_max(h, as_s16x2(0u));
            }
            d = left;
            u = as_u32(h);
            st.Hl[r] = u;
            if constexpr (LOCAL && TAGS) {
                /* the lane's rows are consecutive, so "max score, then first row" inside the lane is one packed maximum over
                 * (H * R + R-1 - r): v_pk_mad_u16 + v_pk_max_u16 for both pairs (the host guarantees H * R + R-1 <= 65535) */
                uint32_t tg = pk_row_tag_of<R>(u, r);
                if constexpr (PARTIAL) tg = (r < nrows) ? tg : 0u; /* rows past the query's end hold no cells */
                colMax = r == 0 ? tg : dpx::pk_max_u16_raw(colMax, tg);
            } else if constexpr (LOCAL) {
                /* first strict maximum of the row, per pair: the int32 kernels' (score, column) key, one per half -- v_and_or_b32 /
                 * v_lshl_or_b32 build it, v_max_u32 (v_max3_u32 across two unrolled steps) folds it: 3-4 ops per two cells where the
                 * packed running-maximum + column select of round 1 took 5 */
                st.rmax[r] = max(st.rmax[r], (u & 0xFFFF0000u) | negj);
                st.rcol[r] = max(st.rcol[r], (u << 16) | negj);
            }
        }
        if constexpr (LOCAL && TAGS) { /* ... and the column: once per step and pair, not once per row */
            st.keyA = max(st.keyA, (colMax & 0xFFFF0000u) | negj);
            st.keyB = max(st.keyB, (colMax << 16) | negj);
        }
        st.dtop = upin;
        if (writeEdge && lane == 63) edge[j] = st.Hl[R - 1];
        if constexpr (MASKED && !WHOLE) store_tile_pk<R>(tileA, tileB, st.Hl);
    }
    if constexpr (!MASKED || WHOLE) {
        bool doStore = lane < storeLanes; /* whole lines (see lin_step) */
        if constexpr (MASKED) doStore = doStore && ramp_stores<R>(lane, t, n, rampLines);
        if (doStore) store_tile_pk<R>(tileA, tileB, st.Hl);
    }
}

#ifndef DPX_PK_MIN_BLOCKS
#define DPX_PK_MIN_BLOCKS 1
#endif
template <int R, bool LOCAL, bool TAGS>
__global__ void __launch_bounds__(DPX_FILL_THREADS, DPX_PK_MIN_BLOCKS) k_linear_fill_pk(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * (int)a.wavesPerBlock + wv; /* couple index */
    if (c >= a.numPairs) return;
    const int pA = a.order[2 * c], pB = a.order[2 * c + 1];
    const dpx_pair_dev prA = a.pairs[pA], prB = a.pairs[pB];
    const int n = prA.n, m = prA.m; /* host guarantees prB.n == n, prB.m == m, both > 0 */
    const unsigned char *refA = reinterpret_cast<const unsigned char *>(a.seq + prA.refIdx);
    const unsigned char *refB = reinterpret_cast<const unsigned char *>(a.seq + prB.refIdx);
    const unsigned char *qryA = reinterpret_cast<const unsigned char *>(a.seq + prA.qryIdx);
    const unsigned char *qryB = reinterpret_cast<const unsigned char *>(a.seq + prB.qryIdx);
    const int gap = a.gapOpen;
    const uint32_t matchP = ((uint32_t)(uint16_t)a.match << 16) | (uint16_t)a.match;
    const uint32_t negDeltaP = ((uint32_t)(uint16_t)(a.mismatch - a.match) << 16) | (uint16_t)(a.mismatch - a.match);
    const int gapK = (LOCAL && TAGS) ? -gap : gap; /* row-tag batches (gap <= 0): the cell update subtracts |gap| with saturation (pk_gap_sat) */
    const uint32_t gapP = ((uint32_t)(uint16_t)gapK << 16) | (uint16_t)gapK;

    unsigned char *my = smem + (size_t)wv * a.ldsPerWave;
    uint32_t *edge = reinterpret_cast<uint32_t *>(my);                    /* packed edge[j], j = 0..n+1 */
    uint16_t *refl = reinterpret_cast<uint16_t *>(my + a.ldsRefOff);      /* refl[64 + (j-1)] = A char << 8 | B char */
    for (int x = 4 * lane; x < n; x += 256) { /* four columns per lane and trip: 2 + 2 aligned dword loads, one 8-byte LDS store */
        const uint32_t a4 = load4(refA + x), b4 = load4(refB + x);
        uint2 w;
        w.x = __builtin_amdgcn_perm(a4, b4, 0x05010400u); /* {B0, A0, B1, A1}: entry j = A char << 8 | B char */
        w.y = __builtin_amdgcn_perm(a4, b4, 0x07030602u);
        *reinterpret_cast<uint2 *>(refl + 64 + x) = w;    /* refl + 64 is 8-byte aligned (ldsRefOff is a multiple of 16) */
    }
    for (int x = lane; x <= n + 1; x += 64) {
        const uint16_t b = (uint16_t)(LOCAL ? 0 : x * gap);
        edge[x] = ((uint32_t)b << 16) | b;
    }
    int16_t *HpA = a.mat + prA.matOff, *HpB = a.mat + prB.matOff;
    const int W = n + 63;
    const int S = dpx_tiled_stripes(m, R);
    int bestA = 0, browA = 0, bcolA = 0, bestB = 0, browB = 0, bcolB = 0;
    PkState<R> st;

    for (int k = 0; k < S; k++) {
        const int base = k * 64 * R;
        const int row0 = base + lane * R;
        const int nrows = min(max(m - row0, 0), R);
        const bool laneHasRows = nrows > 0;
        const bool hasNext = (k + 1 < S);
        int qa[R], qb[R];
        load_query_rows<R>(qa, qryA, row0, nrows);
        load_query_rows<R>(qb, qryB, row0, nrows);
#pragma unroll
        for (int r = 0; r < R; r++) {
            st.qc[r] = ((uint32_t)qa[r] << 16) | (uint32_t)qb[r]; /* rows past the end: 0x0100 in both halves */
            const uint16_t b = (uint16_t)(LOCAL ? 0 : (row0 + 1 + r) * gap);
            st.Hl[r] = ((uint32_t)b << 16) | b;
            st.rmax[r] = 0u;
            st.rcol[r] = 0u;
        }
        st.keyA = st.keyB = 0u;
        { const uint16_t b = (uint16_t)(LOCAL ? 0 : row0 * gap); st.dtop = ((uint32_t)b << 16) | b; }
        const size_t csA = prA.chunkStride, csB = prB.chunkStride;
        int16_t *tileA = HpA + (size_t)k * (size_t)n * csA + (size_t)lane * (R < 8 ? R : 8);
        int16_t *tileB = HpB + (size_t)k * (size_t)n * csB + (size_t)lane * (R < 8 ? R : 8);
        const uint16_t *rp = refl + 64 - lane;
        uint32_t rcN = rp[0];
        uint32_t e0N = edge[1];
#define DPX_PK_STEP(MASKED_, WHOLE_, HASROWS_, PARTIAL_)                                                                        \
        {                                                                                                             \
            const uint32_t rc16 = rcN, e0 = e0N;                                                                      \
            rcN = rp[t + 1];                                                                                          \
            e0N = edge[min(t + 2, n + 1)];                                                                            \
            const uint32_t rcP = __builtin_amdgcn_perm(0u, rc16, 0x0c010c00u); /* {A char, B char} -> 16-bit lanes */  \
            pk_step<R, LOCAL, MASKED_, WHOLE_, TAGS, PARTIAL_>(st, t, lane, n, HASROWS_, nrows, matchP, negDeltaP, gapP, e0, rcP, edge, hasNext, \
                                       tileA + (size_t)t * csA, tileB + (size_t)t * csB, storeLanes, a.rampLines);   \
        }
        const bool fast = (base + 64 * R <= m) && (n >= 64);
        const int storeLanes = (S == 1) ? store_lanes<R>(m) : 64;
        if (S == 1) {
            if (fast) {
                int t = 0;
                for (; t < 63; t++) DPX_PK_STEP(true, true, true, false)
                for (; t + 1 < n;) { DPX_PK_STEP(false, true, true, false) t++; DPX_PK_STEP(false, true, true, false) t++; } /* two steps per trip: the keys of both fold with v_max3_u32 */
                for (; t < n; t++) DPX_PK_STEP(false, true, true, false)
                for (; t < W; t++) DPX_PK_STEP(true, true, true, false)
            } else {
                for (int t = 0; t < W; t++) DPX_PK_STEP(true, true, laneHasRows, true)
            }
        } else if (fast) { /* stripes share their ramp chunks: masked stores on the ramps */
            int t = 0;
            for (; t < 63; t++) DPX_PK_STEP(true, false, true, false)
            for (; t < n; t++) DPX_PK_STEP(false, false, true, false)
            for (; t < W; t++) DPX_PK_STEP(true, false, true, false)
        } else {
            for (int t = 0; t < W; t++) DPX_PK_STEP(true, false, laneHasRows, true)
        }
#undef DPX_PK_STEP
        if constexpr (LOCAL && TAGS) { /* (stripes ascend: a strict '>' keeps the first row holding the maximum) */
            constexpr int TB = R == 16 ? 4 : R == 8 ? 3 : R == 4 ? 2 : 1;
            const int hA = (int)(st.keyA >> (16 + TB)), hB = (int)(st.keyB >> (16 + TB));
            if (hA > bestA) { bestA = hA; browA = row0 + R - (int)((st.keyA >> 16) & (R - 1)); bcolA = 0xFFFF - (int)(st.keyA & 0xFFFFu); }
            if (hB > bestB) { bestB = hB; browB = row0 + R - (int)((st.keyB >> 16) & (R - 1)); bcolB = 0xFFFF - (int)(st.keyB & 0xFFFFu); }
        } else if constexpr (LOCAL) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int hA = (int)(st.rmax[r] >> 16), hB = (int)(st.rcol[r] >> 16);
                if (r < nrows && hA > bestA) { bestA = hA; browA = row0 + 1 + r; bcolA = 0xFFFF - (int)(st.rmax[r] & 0xFFFFu); }
                if (r < nrows && hB > bestB) { bestB = hB; browB = row0 + 1 + r; bcolB = 0xFFFF - (int)(st.rcol[r] & 0xFFFFu); }
            }
        }
    }

    if constexpr (LOCAL) {
        /* max score, then first row (c++/LinearSmithWaterman.cpp:145-157) ... */
        const unsigned long long mineA = ((unsigned long long)(unsigned)bestA << 32) | (unsigned)(0x7FFFFFFF - browA);
        const unsigned long long mineB = ((unsigned long long)(unsigned)bestB << 32) | (unsigned)(0x7FFFFFFF - browB);
        const unsigned long long topA = wave_max_u64(mineA), topB = wave_max_u64(mineB);
        /* ... and the lane that owns that row already holds the row's first column */
        if ((int)(topA >> 32) == 0) { if (lane == 0) { a.score[pA] = 0; a.endRow[pA] = 0; a.endCol[pA] = 0; } }
        else if (mineA == topA) { a.score[pA] = bestA; a.endRow[pA] = browA; a.endCol[pA] = bcolA; }
        if ((int)(topB >> 32) == 0) { if (lane == 0) { a.score[pB] = 0; a.endRow[pB] = 0; a.endCol[pB] = 0; } }
        else if (mineB == topB) { a.score[pB] = bestB; a.endRow[pB] = browB; a.endCol[pB] = bcolB; }
    } else {
        const int lastBase = (S - 1) * 64 * R;
        const int lm = (m - 1 - lastBase) / R, rm = (m - 1 - lastBase) % R;
        if (lane == lm) {
            uint32_t v = st.Hl[0];
#pragma unroll
            for (int r = 1; r < R; r++) v = (r == rm) ? st.Hl[r] : v;
            a.score[pA] = (int)(int16_t)(v >> 16); a.endRow[pA] = m; a.endCol[pA] = n;
            a.score[pB] = (int)(int16_t)(v & 0xFFFFu); a.endRow[pB] = m; a.endCol[pB] = n;
        }
    }
}

/* =====================================================================================================
 * Packed lane kernel for short and medium queries (round 3): k_linear_lanes' schedule and tile layout with the packed-int16
 * arithmetic of k_linear_fill_pk -- but the two halves of a register do not hold two PAIRS (which would need equal shapes), they
 * hold two ROW BLOCKS OF THE SAME PAIR: a lane owns 16 consecutive rows, rows 0-7 in the high halves and rows 8-15 in the low
 * halves, and the low half runs ONE COLUMN BEHIND the high half.  A physical lane is two virtual lanes of the wavefront (2l and
 * 2l+1): the low half takes its "row above" from the lane's own high half of the previous step, the high half from the previous
 * lane's low half (one DPP move + one v_perm_b32), the diagonal is last step's "row above" as everywhere.  Any mix of shapes
 * packs as before (a pair of m rows takes ceil(m/16) lanes); the cell update costs 7 (NW) / 8 (SW) VOP3P instructions per TWO
 * cells instead of 5-6 per cell, and the per-step bookkeeping is shared by 16 rows: the int32 kernel needs 117 vector
 * instructions per step and 16 rows (2 x 58.5), this one 80.
 *   - Column 0 of the low half (its first step) needs no mask: its state starts at a large negative value, so that
 *     max(up+gap, left+gap, diag+s) = up+gap = the column-0 border (SW: 0) -- the host checks that the value cannot wrap.
 *   - The high half's extra step at column n+1 (while the low half finishes column n) runs only in lanes that own more than
 *     8 rows; its results are garbage that nothing reads (SW: masked out of the start-cell key).
 *   - SW start cell: one (H * 8 + 7 - r, column) key per half (pk_row_tag); rows past the query's end are not masked -- the host
 *     uses this kernel for SW only if gap <= 0 and mismatch <= 0, where such a row can never exceed the real rows above it.
 *   - Matrices: the SAME tile layout as k_linear_lanes<16> (dpx_layout.h, lanes == 16, rows == 16).  A line's address is a
 *     function of (row block, column block) only; this kernel just completes the lines in another order: the line of lane
 *     lambda, block h, that k_linear_lanes<16> writes to chunk T of the wave's stream is complete here in step T + delta + h
 *     (delta = this schedule's skew minus that one's), so two 8-line stores per step go to per-owner chunks instead of one
 *     contiguous 2 KiB.  Export and traceback do not know which kernel filled a pair.
 * ===================================================================================================== */
template <bool LOCAL>
__global__ void __launch_bounds__(DPX_FILL_THREADS) k_linear_lanes_pk(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStageBytes = 2 * 64 * kStageLine; /* [block][lane] lines */
    constexpr int kStepElems = 1024;                 /* int16 elements of one chunk of the wave's stream (two 1-KiB halves) */
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x * (int)a.wavesPerBlock + wv;
    if (w >= a.numPairs) return;
    const LaneSlot sl = find_slot(a.waves + w, lane);
    const bool has = sl.has;
    const int p = sl.p, l = sl.l;
    const dpx_pair_dev pr = a.pairs[p];
    const int n = has ? pr.n : 0, m = has ? pr.m : 0;
    const int gap = a.gapOpen;
    const uint32_t matchP = ((uint32_t)(uint16_t)a.match << 16) | (uint16_t)a.match;
    const uint32_t negDeltaP = ((uint32_t)(uint16_t)(a.mismatch - a.match) << 16) | (uint16_t)(a.mismatch - a.match);
    const int gapK = LOCAL ? -gap : gap; /* SW (gap <= 0, checked by the host): |gap| for the saturating gap term (pk_gap_sat) */
    const uint32_t gapP = ((uint32_t)(uint16_t)gapK << 16) | (uint16_t)gapK;
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);

    unsigned char *tileL = smem + (size_t)wv * a.ldsPerWave; /* [line stage][staged references] */
    unsigned char *scratch = tileL;
    unsigned char *refl = tileL + kStageBytes + sl.refOff;
    const unsigned char *refs = stage_bytes(refl, ref, n, l, max(sl.num, 1));

    const int row0 = l * 16;
    const int nrows = min(max(m - row0, 0), 16);
    /* the smallest value that one more weight cannot wrap: "minus infinity" left of column 0 in the low halves */
    const int wmin = min(min(a.match, a.mismatch), min(gap, 0));
    const uint32_t negInf = (uint16_t)(-32768 - wmin);
    uint32_t Hl[8], qc[8], dtop, keyHi = 0u, keyLo = 0u;
    {
        int qa[16];
        load_query_rows<16>(qa, qry, row0, nrows);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            qc[r] = ((uint32_t)qa[r] << 16) | (uint32_t)qa[8 + r]; /* rows past the end: 0x0100, matches no byte */
            const uint16_t b = (uint16_t)(LOCAL ? 0 : (row0 + 1 + r) * gap);
            Hl[r] = ((uint32_t)b << 16) | negInf; /* high half: column 0 of its rows; low half: not started */
        }
        const uint16_t b = (uint16_t)(LOCAL ? 0 : row0 * gap);
        dtop = ((uint32_t)b << 16) | negInf;
    }
    const int first = lane - l;
    const int dOld = first & 7, dNew = (2 * first) & 7;
    const int skew = 2 * l + dNew; /* the HIGH half runs column j = t - skew + 1 in step t, the low half column j - 1 */
    const int n8 = (n + 7) & ~7;
    const int LB = (int)dpx_tile8_row_blocks(m);
    const int nBlocks = has ? min(max(LB - 2 * l, 0), 2) : 0; /* row blocks of this lane that hold rows */
    /* routing, once: [first | last << 16] completing step of the high block, the same of the low block (one step later), and
     * delta = how many steps later than in k_linear_lanes<16>'s schedule this lane's lines are complete */
    {
        uint32_t *mine = reinterpret_cast<uint32_t *>(tileL + lane * kStageLine + 128);
        mine[0] = nBlocks >= 1 ? ((uint32_t)(skew + 7) | ((uint32_t)(n8 + skew - 1) << 16)) : 0x00007FFFu; /* (never valid: first > last) */
        mine[1] = nBlocks >= 2 ? ((uint32_t)(skew + 8) | ((uint32_t)(n8 + skew) << 16)) : 0x00007FFFu;
        mine[2] = (uint32_t)(skew - (l + dOld));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t rtHi[8], rtLo[8];
    int dl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t *o = reinterpret_cast<const uint32_t *>(tileL + ((lane & ~7) | k) * kStageLine + 128);
        rtHi[k] = o[0]; rtLo[k] = o[1]; dl[k] = (int)o[2];
    }
    int16_t *waveBase = a.mat + (size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pr.matOff >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)(pr.matOff & 0xFFFFFFFFull)));
    const int lastStep = has ? (nBlocks >= 2 ? n8 + skew : n8 + skew - 1) : -1; /* the step in which this lane's last line is complete */
    const int steps = __builtin_amdgcn_readfirstlane(wave_max_i32(lastStep)) + 1;
    const unsigned char *rp = refs - skew; /* rp[t] = reference character of the high half's column */
    const unsigned nEff = nrows > 8 ? (unsigned)n + 1u : (nrows > 0 ? (unsigned)n : 0u); /* on a cell iff (unsigned)(t - skew) < nEff */
    const int rpLast = n + skew; /* rp[rpLast] = refs[n]: inside the slack of stage_bytes (never read past the slot's staged reference) */
    unsigned char *putPtr = tileL + lane * kStageLine;
    const unsigned char *fetchPtr[8]; /* piece lane % 8 of the high-block line of lane k of this lane's group (low block: + 64 lines) */
#pragma unroll
    for (int k = 0; k < 8; k++) fetchPtr[k] = tileL + ((lane & ~7) | k) * kStageLine + (((lane + 2 * k) & 7) << 4);
    const int laneElems = ((lane >> 3) << 6) + ((lane & 7) << 3); /* this lane's 16 bytes inside a 1-KiB half chunk */
    u32x4 pendA, pendB;
    bool okA = false, okB = false;
    int16_t *dstA = nullptr, *dstB = nullptr;
    uint32_t bord = (uint16_t)(-skew * gap); /* NW, first lane of a slot: H[0][j] of the high half's column, j = t - skew + 1 (16 bits; + gap before use) */
    uint32_t rcCur = rp[0], rcPrev = 0u;
    auto flush = [&]() __attribute__((always_inline)) {
        if (okA) stream_store(reinterpret_cast<u32x4 *>(dstA), pendA);
        if (okB) stream_store(reinterpret_cast<u32x4 *>(dstB), pendB);
    };
    auto lane_step = [&](const int t, auto kTag) __attribute__((always_inline)) {
        constexpr int K = decltype(kTag)::value; /* t & 7 */
        const int tms = t - skew;
        const uint32_t rcP = (rcCur << 16) | rcPrev; /* {character of column j, of column j-1} */
        rcPrev = rcCur;
        rcCur = rp[min(t + 1, rpLast)];
        const uint32_t sh = (uint32_t)wave_shr1((int)Hl[7], 0);
        uint32_t upin = __builtin_amdgcn_perm(sh, Hl[7], 0x05040302u); /* {previous lane's low half, this lane's high half} */
        if constexpr (!LOCAL) bord = (uint32_t)(uint16_t)(bord + (uint32_t)gap);
        if (l == 0) upin = (LOCAL ? 0u : (bord << 16)) | (upin & 0xFFFFu); /* a slot's first lane: the row-0 border */
        if ((unsigned)tms < nEff) {
            uint32_t u = upin, colMax = 0u;
            const uint32_t onesP = 0x00010001u;
            uint32_t dsum[8]; /* diagonal terms first, then every row in place (no register rotation across the masked region, see lin_cells) */
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t differs = dpx::pk_min_u16_raw(qc[r] ^ rcP, onesP);
                const uint32_t sc = dpx::pk_mad_i16_raw(differs, negDeltaP, matchP);
                dsum[r] = as_u32((s16x2)(as_s16x2(r == 0 ? dtop : Hl[r - 1]) + as_s16x2(sc)));
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                s16x2 h;
                if constexpr (LOCAL) h = dpx::pk_max(pk_gap_sat(u, Hl[r], gapP), as_s16x2(dsum[r])); /* gap <= 0 (host): gapP = |gap|, no max(H, 0) needed */
                else h = dpx::pk_max(dpx::pk_max(as_s16x2(u), as_s16x2(Hl[r])) + as_s16x2(gapP), as_s16x2(dsum[r]));
                u = as_u32(h);
                Hl[r] = u;
                if constexpr (LOCAL) {
                    const uint32_t tg = pk_row_tag_of<8>(u, r);
                    colMax = r == 0 ? tg : dpx::pk_max_u16_raw(colMax, tg);
                }
            }
            dtop = upin;
            if constexpr (LOCAL) {
                const uint32_t negj = 0xFFFEu - (uint32_t)tms;                 /* 0xFFFF - j of the high half */
                if (tms < n) keyHi = max(keyHi, (colMax & 0xFFFF0000u) | negj); /* (column n+1 does not exist) */
                keyLo = max(keyLo, (colMax << 16) | (negj + 1u));              /* the low half is on column j - 1 (column 0 scores 0) */
            }
            /* park both blocks' 16 bytes in their lines: the high block's column in slot t % 8, the low block's in slot (t-1) % 8 */
            u32x4 vh = {pk_hi16(Hl[0], Hl[1]), pk_hi16(Hl[2], Hl[3]), pk_hi16(Hl[4], Hl[5]), pk_hi16(Hl[6], Hl[7])};
            u32x4 vl = {pack_lo16((int)Hl[0], (int)Hl[1]), pack_lo16((int)Hl[2], (int)Hl[3]), pack_lo16((int)Hl[4], (int)Hl[5]), pack_lo16((int)Hl[6], (int)Hl[7])};
            *reinterpret_cast<u32x4 *>(putPtr + (K << 4)) = vh;
            *reinterpret_cast<u32x4 *>(putPtr + 64 * kStageLine + (((K + 7) & 7) << 4)) = vl;
        }
        flush();
        /* lines that are complete now: odd steps the HIGH blocks of the lanes (t+1)/2 mod 4 (and + 4) of every group, even steps the
         * LOW blocks of the lanes t/2 mod 4 (and + 4) */
        constexpr int H = (K & 1) ? 0 : 1;
        constexpr int OA = (K & 1) ? (((K + 1) >> 1) & 3) : (K >> 1), OB = OA + 4;
        const uint32_t ra = H ? rtLo[OA] : rtHi[OA], rb = H ? rtLo[OB] : rtHi[OB];
        okA = (uint32_t)t >= (ra & 0xFFFFu) && (uint32_t)t <= (ra >> 16);
        okB = (uint32_t)t >= (rb & 0xFFFFu) && (uint32_t)t <= (rb >> 16);
        dstA = waveBase + (ptrdiff_t)(t - dl[OA] - H) * kStepElems + (H * 512 + laneElems);
        dstB = waveBase + (ptrdiff_t)(t - dl[OB] - H) * kStepElems + (H * 512 + laneElems);
        pendA = *reinterpret_cast<const u32x4 *>(fetchPtr[OA] + H * 64 * kStageLine);
        pendB = *reinterpret_cast<const u32x4 *>(fetchPtr[OB] + H * 64 * kStageLine);
    };
    {
        int t = 0;
        for (; t + 8 <= steps; t += 8) {
            lane_step(t + 0, std::integral_constant<int, 0>{}); lane_step(t + 1, std::integral_constant<int, 1>{});
            lane_step(t + 2, std::integral_constant<int, 2>{}); lane_step(t + 3, std::integral_constant<int, 3>{});
            lane_step(t + 4, std::integral_constant<int, 4>{}); lane_step(t + 5, std::integral_constant<int, 5>{});
            lane_step(t + 6, std::integral_constant<int, 6>{}); lane_step(t + 7, std::integral_constant<int, 7>{});
        }
        if (t + 0 < steps) lane_step(t + 0, std::integral_constant<int, 0>{});
        if (t + 1 < steps) lane_step(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < steps) lane_step(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < steps) lane_step(t + 3, std::integral_constant<int, 3>{});
        if (t + 4 < steps) lane_step(t + 4, std::integral_constant<int, 4>{});
        if (t + 5 < steps) lane_step(t + 5, std::integral_constant<int, 5>{});
        if (t + 6 < steps) lane_step(t + 6, std::integral_constant<int, 6>{});
    }
    flush();
    if constexpr (LOCAL) {
        /* the lane's candidate: the high block's rows come first in row-major order, so the low block must be strictly better */
        int bestv = (int)(keyHi >> 19), bestrow = row0 + 8 - (int)((keyHi >> 16) & 7), bestcol = 0xFFFF - (int)(keyHi & 0xFFFFu);
        const int vLo = (int)(keyLo >> 19);
        if (vLo > bestv) { bestv = vLo; bestrow = row0 + 16 - (int)((keyLo >> 16) & 7); bestcol = 0xFFFF - (int)(keyLo & 0xFFFFu); }
        int *mine = reinterpret_cast<int *>(scratch + lane * 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); /* (the line stage this aliases has been read to the end) */
        __builtin_amdgcn_wave_barrier();
        mine[0] = bestv; mine[1] = bestrow; mine[2] = bestcol;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (has && l == 0) {
            for (int k = 1; k < sl.num; k++) {
                const int *o = reinterpret_cast<const int *>(scratch + (lane + k) * 16);
                if (o[0] > bestv) { bestv = o[0]; bestrow = o[1]; bestcol = o[2]; }
            }
            a.score[p] = bestv; a.endRow[p] = bestv > 0 ? bestrow : 0; a.endCol[p] = bestv > 0 ? bestcol : 0;
        }
    } else {
        const int lm = (m - 1) / 16, rr = (m - 1) % 16; /* owner of row m: its registers hold column n after its last step */
        if (has && l == lm) {
            uint32_t v = Hl[0];
#pragma unroll
            for (int r = 1; r < 8; r++) v = (r == (rr & 7)) ? Hl[r] : v;
            a.score[p] = (rr < 8) ? (int)(int16_t)(v >> 16) : (int)(int16_t)(v & 0xFFFFu);
            a.endRow[p] = m; a.endCol[p] = n;
        }
    }
}

/* =====================================================================================================
 * Affine-gap (Gotoh) global fill: AffineNeedlemanWunsch (c++/AffineNeedlemanWunsch.cpp:167-240).
 *   D[i][j] = (i==1) ? H[i-1][j]+o+e : max(H[i-1][j]+o+e, D[i-1][j]+e)      vertical gap   (:185-197)
 *   I[i][j] = (j==1) ? H[i][j-1]+o+e : max(H[i][j-1]+o+e, I[i][j-1]+e)      horizontal gap (:201-213)
 *   H[i][j] = max(I, max(D, H[i-1][j-1]+s))                                                (:229-236)
 * The i==1 / j==1 special cases are expressed by virtual borders D[0][j] = I[i][0] = DPX_NEG, which gives the
 * identical values for every stored cell.  Three int16 planes (H, I, D) are written per step.
 * ===================================================================================================== */
template <int R>
struct AffState {
    int Hl[R], Il[R]; /* H[row][j-1], I[row][j-1] */
    int Dl[R];        /* D[row][j] just computed (needed only for the store and the lane hand-off) */
    int qc[R];
    int dtop;
};

template <int R>
__device__ __forceinline__ void aff_cells(AffState<R> &st, const int upH, const int upD, const int rc, const int match,
                                          const int mismatch, const int oe, const int e) {
    int uH = upH, uD = upD, d = st.dtop;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int lH = st.Hl[r];
        const int s = (st.qc[r] == rc) ? match : mismatch;
        const int Dn = max(uH + oe, uD + e);
        const int In = max(lH + oe, st.Il[r] + e);
        const int h = max(max(Dn, d + s), In); /* v_max3_i32 */
        d = lH;
        uH = h;
        uD = Dn;
        st.Hl[r] = h;
        st.Il[r] = In;
        st.Dl[r] = Dn;
    }
    st.dtop = upH;
}

template <int R, bool STORE, bool MASKED, bool WHOLE>
__device__ __forceinline__ void aff_step(AffState<R> &st, const int t, const int lane, const int n, const bool laneHasRows,
                                         const int match, const int mismatch, const int oe, const int e, const int e0H,
                                         const int e0D, const int rc, int16_t *edgeH, int16_t *edgeD,
                                         const bool writeEdge, int16_t *tileDst, const int storeLanes, const int rampLines) {
    const int j = t - lane + 1;
    const int upH = wave_shr1(st.Hl[R - 1], e0H);
    const int upD = wave_shr1(st.Dl[R - 1], e0D);
    bool active = true;
    if constexpr (MASKED) active = laneHasRows && (j >= 1) && (j <= n);
    if (active) {
        aff_cells<R>(st, upH, upD, rc, match, mismatch, oe, e);
        if (writeEdge && lane == 63) {
            edgeH[j] = (int16_t)st.Hl[R - 1];
            edgeD[j] = (int16_t)st.Dl[R - 1];
        }
        if constexpr (STORE && MASKED && !WHOLE) {
            store_tile<R>(tileDst, st.Hl);
            store_tile<R>(tileDst + 64 * R, st.Il);
            store_tile<R>(tileDst + 128 * R, st.Dl);
        }
    }
    if constexpr (STORE && (!MASKED || WHOLE)) { /* whole lines, also on the skew ramps (see lin_step) */
        bool doStore = lane < storeLanes;
        if constexpr (MASKED) doStore = doStore && ramp_stores<R>(lane, t, n, rampLines);
        if (doStore) {
            store_tile<R>(tileDst, st.Hl);
            store_tile<R>(tileDst + 64 * R, st.Il);
            store_tile<R>(tileDst + 128 * R, st.Dl);
        }
    }
}

template <int R, bool STORE>
__global__ void __launch_bounds__(DPX_FILL_THREADS) k_affine_fill(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int p = blockIdx.x * (int)a.wavesPerBlock + wv;
    if (p >= a.numPairs) return;
    if (a.order) p = a.order[p];
    const dpx_pair_dev pr = a.pairs[p];
    const int n = pr.n, m = pr.m;
    const int match = a.match, mismatch = a.mismatch;
    const int o = a.gapOpen, e = a.gapExtend, oe = o + e;

    if (m <= 0 || n <= 0) {
        if (lane == 0) { /* H[m][n] on the border: 0 at the origin, else o + len*e (AffineNeedlemanWunsch.cpp:43-53) */
            const int len = m <= 0 ? max(n, 0) : m;
            a.score[p] = len <= 0 ? 0 : o + len * e;
            a.endRow[p] = max(m, 0);
            a.endCol[p] = max(n, 0);
        }
        return;
    }
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);
    unsigned char *my = smem + (size_t)wv * a.ldsPerWave;
    int16_t *edgeH = reinterpret_cast<int16_t *>(my);
    int16_t *edgeD = reinterpret_cast<int16_t *>(my + a.ldsEdge2Off);
    const unsigned char *refl = stage_bytes(my + a.ldsRefOff + 64, ref, n, lane, 64) - 64;
    /* row-0 border H[0][j] = o + j*e (AffineNeedlemanWunsch.cpp:50-53); D[0][j] is the virtual DPX_NEG (k == 0 below) */
    for (int x = lane; x <= n + 1; x += 64) { edgeH[x] = (int16_t)(o + x * e); edgeD[x] = 0; }

    int16_t *Mp = a.mat + pr.matOff;
    const int W = n + 63;
    const int S = dpx_tiled_stripes(m, R);
    AffState<R> st;

    if (STORE && S >= 2 && n >= 128) {
        /* ---------- rolling schedule (see k_linear_fill): lanes run straight on into the next stripe ---------- */
        const unsigned char *ql = stage_bytes(my + a.ldsQryOff, qry, m, lane, 64);
        int row0 = lane * R;
        int nrows = min(max(m - row0, 0), R);
        int jl = 1 - lane, kl = 0;
        load_query_rows<R>(st.qc, qry, row0, nrows);
#pragma unroll
        for (int r = 0; r < R; r++) {
            st.Hl[r] = o + (row0 + 1 + r) * e;
            st.Il[r] = DPX_NEG;
            st.Dl[r] = DPX_NEG;
        }
        st.dtop = row0 == 0 ? 0 : o + row0 * e;
        int j0 = 1;
        bool sw = false;
        int rcN = refl[63 + jl];
        int eHN = edgeH[1];
        int eDN = DPX_NEG; /* lane 0 is in stripe 0 first: virtual D[0][j] */
        const size_t cs = pr.chunkStride;
        int16_t *tile = Mp + (size_t)lane * R;
        const int total = S * n + 63;
        auto roll_step = [&](const int T) {
            const int rc = rcN, eH = eHN, eD = eDN;
            const int jn = (jl >= n) ? 1 : jl + 1;
            rcN = refl[63 + jn];
            j0 = (j0 >= n) ? 1 : j0 + 1;
            eHN = edgeH[j0];
            eDN = (T + 1 < n) ? DPX_NEG : (int)edgeD[j0]; /* lane 0 leaves stripe 0 after n steps */
            const int upH = wave_shr1(st.Hl[R - 1], eH);
            const int upD = wave_shr1(st.Dl[R - 1], eD);
            if (sw) {
                row0 += 64 * R;
                nrows = min(max(m - row0, 0), R);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    st.qc[r] = (r < nrows) ? (int)ql[row0 + r] : 0x100;
                    st.Hl[r] = o + (row0 + 1 + r) * e;
                    st.Il[r] = DPX_NEG;
                    st.Dl[r] = DPX_NEG;
                }
                st.dtop = o + row0 * e;
            }
            if (jl >= 1 && kl < S && nrows > 0) {
                aff_cells<R>(st, upH, upD, rc, match, mismatch, oe, e);
                if (lane == 63 && kl + 1 < S) {
                    edgeH[jl] = (int16_t)st.Hl[R - 1];
                    edgeD[jl] = (int16_t)st.Dl[R - 1];
                }
            }
            if constexpr (STORE) {
                if (ramp_stores<R>(lane, T, S * n, a.rampLines)) { /* on the pair's two ramps only the lines with cells */
                    int16_t *dst = tile + (size_t)T * cs;
                    store_tile<R>(dst, st.Hl);
                    store_tile<R>(dst + 64 * R, st.Il);
                    store_tile<R>(dst + 128 * R, st.Dl);
                }
            }
            sw = false;
            if (jl >= n) { jl = 1; kl++; sw = kl < S; }
            else jl++;
        };
        int T = 0;
        for (; T + 1 < total; T += 2) {
            roll_step(T);
            roll_step(T + 1);
        }
        if (T < total) roll_step(T);
    } else {
    for (int k = 0; k < S; k++) {
            const int base = k * 64 * R;
            const int row0 = base + lane * R;
            const int nrows = min(max(m - row0, 0), R);
            const bool laneHasRows = nrows > 0;
            const bool hasNext = (k + 1 < S);
            load_query_rows<R>(st.qc, qry, row0, nrows);
    #pragma unroll
            for (int r = 0; r < R; r++) {
                st.Hl[r] = o + (row0 + 1 + r) * e; /* H[i][0] = o + i*e (:43-46) */
                st.Il[r] = DPX_NEG;                /* virtual I[i][0] */
                st.Dl[r] = DPX_NEG;
            }
            st.dtop = row0 == 0 ? 0 : o + row0 * e; /* H[0][0] = 0 */
            const size_t cs = pr.chunkStride;
            int16_t *tile = Mp + (size_t)k * (size_t)n * cs + (size_t)lane * R;
    
            const unsigned char *rp = refl + 64 - lane;
            int rcN = rp[0];
            int eHN = edgeH[1];
            int eDN = k == 0 ? DPX_NEG : (int)edgeD[1];
    #define DPX_AFF_STEP(MASKED_, WHOLE_, HASROWS_)                                                                               \
            {                                                                                                             \
                const int rc = rcN, eH = eHN, eD = eDN;                                                                   \
                rcN = rp[t + 1];                                                                                          \
                eHN = edgeH[min(t + 2, n + 1)];                                                                           \
                eDN = k == 0 ? DPX_NEG : (int)edgeD[min(t + 2, n + 1)];                                                   \
                aff_step<R, STORE, MASKED_, WHOLE_>(st, t, lane, n, HASROWS_, match, mismatch, oe, e, eH, eD, rc, edgeH, edgeD, \
                                            hasNext, tile + (size_t)t * cs, storeLanes, a.rampLines);                     \
            }
            const bool fast = (base + 64 * R <= m) && (n >= 64);
            const int storeLanes = (S == 1) ? store_lanes<R>(m) : 64;
            if (S == 1) {
                if (fast) {
                    int t = 0;
                    for (; t < 63; t++) DPX_AFF_STEP(true, true, true)
                    for (; t + 1 < n;) { DPX_AFF_STEP(false, true, true) t++; DPX_AFF_STEP(false, true, true) t++; } /* two steps per trip */
                    for (; t < n; t++) DPX_AFF_STEP(false, true, true)
                    for (; t < W; t++) DPX_AFF_STEP(true, true, true)
                } else {
                    for (int t = 0; t < W; t++) DPX_AFF_STEP(true, true, laneHasRows)
                }
            } else if (fast) { /* stripes share their ramp chunks: masked stores on the ramps */
                int t = 0;
                for (; t < 63; t++) DPX_AFF_STEP(true, false, true)
                for (; t + 1 < n;) { DPX_AFF_STEP(false, false, true) t++; DPX_AFF_STEP(false, false, true) t++; }
                for (; t < n; t++) DPX_AFF_STEP(false, false, true)
                for (; t < W; t++) DPX_AFF_STEP(true, false, true)
            } else {
                for (int t = 0; t < W; t++) DPX_AFF_STEP(true, false, laneHasRows)
            }
    #undef DPX_AFF_STEP
        }
}
    const int lastBase = (S - 1) * 64 * R;
    const int lm = (m - 1 - lastBase) / R, rm = (m - 1 - lastBase) % R;
    if (lane == lm) {
        int v = st.Hl[0];
#pragma unroll
        for (int r = 1; r < R; r++) v = (r == rm) ? st.Hl[r] : v;
        a.score[p] = v; /* scoringMemo[m][n] (:365) */
        a.endRow[p] = m;
        a.endCol[p] = n;
    }
}

/* Affine lane-packed kernel: the Gotoh recurrence of k_affine_fill on the several-pairs-per-wave schedule of
 * k_linear_lanes (`up` of H and D through wave_shr:1, a slot's first lane takes the row-0 borders); the three planes H,
 * I, D leave through the same LDS line stage into the 8 x 8 tile layout, three whole-line stores per step.  One wave per
 * workgroup: the stage needs 27 KiB of LDS per wave (three planes), so small workgroups keep five of them on a CU. */
#define DPX_ALANES_THREADS 64
/* Round 3: the cell update keeps its state as H + (o+e), I + e, D + e (aff_cells_g: 9 vector instructions per cell instead of 10, rows
 * in place), eight steps per trip, routing in registers, loop control on the scalar unit -- as in k_linear_lanes.  Tried and dropped:
 * a HALF-LINE stage (4 columns per lane and plane, 15 KiB of LDS per wave instead of 27: ten waves per CU instead of five; the two
 * 64-byte halves of a line stored four steps apart by the same wave) -- bit-exact, 2.42 ms instead of 1.67 on 100k short reads: a
 * 128-byte line that is written in two pieces costs far more than the residency buys. */

template <int R>
struct AffStateG {
    int Hoe[R], Ie[R]; /* H[row][j-1] + (o+e), I[row][j-1] + e */
    int qc[R];
    int dtopOe;        /* H[row0][j-1] + (o+e) */
    int DeLast;        /* D[row0+R][j] + e of the column just computed (the next lane's "D above") */
};

/* D = max(H_up + oe, D_up + e);  I = max(H_left + oe, I_left + e);  H = max3(D, H_diag + s, I)   (c++/AffineNeedlemanWunsch.cpp:185-236) */
template <int R>
__device__ __forceinline__ void aff_cells_g(AffStateG<R> &st, const int upHoe, const int upDe, const int rc, const int matchG, const int mismatchG,
                                            const int oe, const int e, int (&Hv)[R], int (&Iv)[R], int (&Dv)[R]) {
    int dterm[R];
#pragma unroll
    for (int r = 0; r < R; r++) dterm[r] = ((r == 0) ? st.dtopOe : st.Hoe[r - 1]) + ((st.qc[r] == rc) ? matchG : mismatchG); /* (s - oe) */
    int ug = upHoe, ud = upDe;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int Dn = max(ug, ud);
        const int In = max(st.Hoe[r], st.Ie[r]);
        const int h = max(max(Dn, dterm[r]), In); /* v_max3_i32 */
        Hv[r] = h; Iv[r] = In; Dv[r] = Dn;
        ug = h + oe;
        ud = Dn + e;
        st.Hoe[r] = ug;
        st.Ie[r] = In + e;
    }
    st.DeLast = ud;
    st.dtopOe = upHoe;
}

template <int R, bool STORE>
__global__ void __launch_bounds__(DPX_ALANES_THREADS) k_affine_lanes(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(R == 8, "one 8-row block per lane");
    constexpr int kPlane = 64 * kStageLine; /* bytes of one plane's lines */
    constexpr int kStageBytes = 3 * kPlane;
    constexpr int kStepElems = 3 * 512;     /* int16 elements of one chunk of the wave's stream (dpx_layout.h) */
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x * (DPX_ALANES_THREADS / 64) + wv;
    if (w >= a.numPairs) return; /* wave-uniform; numPairs = number of wave descriptors */
    const LaneSlot sl = find_slot(a.waves + w, lane);
    const bool has = sl.has;
    const int p = sl.p, l = sl.l;
    const dpx_pair_dev pr = a.pairs[p];
    const int n = has ? pr.n : 0, m = has ? pr.m : 0;
    const int o = a.gapOpen, e = a.gapExtend, oe = o + e;
    const int matchG = a.match - oe, mismatchG = a.mismatch - oe;
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);

    unsigned char *tileL = smem + (size_t)wv * a.ldsPerWave;
    unsigned char *refl = tileL + (STORE ? kStageBytes : kLaneScratch) + sl.refOff;
    const unsigned char *refs = stage_bytes(refl, ref, n, l, max(sl.num, 1));

    const int row0 = l * R;
    const int nrows = min(max(m - row0, 0), R);
    AffStateG<R> st;
    load_query_rows<R>(st.qc, qry, row0, nrows);
#pragma unroll
    for (int r = 0; r < R; r++) {
        st.Hoe[r] = o + (row0 + 1 + r) * e + oe; /* H[i][0] = o + i*e (AffineNeedlemanWunsch.cpp:43-46) */
        st.Ie[r] = DPX_NEG + e;                  /* virtual I[i][0] */
    }
    st.dtopOe = (row0 == 0 ? 0 : o + row0 * e) + oe; /* H[0][0] = 0 */
    st.DeLast = DPX_NEG + e;

    const int skew = l + sl.d; /* this lane runs column j = t - skew + 1 in step t; skew = lane (mod 8) */
    const int n8 = (n + 7) & ~7;
    const int LB = (int)dpx_tile8_row_blocks(m);
    /* routing, once: a lane's lines are complete in the steps skew + 7, skew + 15, ... <= n8 + skew - 1 (first | last << 16); every lane
     * keeps the words of the eight lanes of its group in registers */
    uint32_t rt[8];
    if constexpr (STORE) {
        const bool rowsHere = has && (LB - l) > 0;
        uint32_t *mine = reinterpret_cast<uint32_t *>(tileL + lane * kStageLine + 128);
        mine[0] = rowsHere ? ((uint32_t)(skew + 7) | ((uint32_t)(n8 + skew - 1) << 16)) : 0x00007FFFu; /* (never valid: first > last) */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 8; k++) rt[k] = *reinterpret_cast<const uint32_t *>(tileL + ((lane & ~7) | k) * kStageLine + 128);
    }
    int16_t *waveBase = a.mat + (size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pr.matOff >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)(pr.matOff & 0xFFFFFFFFull)));
    const int steps = __builtin_amdgcn_readfirstlane(wave_max_i32(has ? (STORE ? n8 : n) + skew : 0));
    const unsigned char *rp = refs - skew;
    const unsigned nEff = nrows > 0 ? (unsigned)n : 0u;
    const int rpLast = n + skew; /* rp[rpLast] = refs[n]: inside the slack of stage_bytes */
    unsigned char *putPtr = tileL + lane * kStageLine; /* + plane * kPlane + (t & 7) * 16 */
    const unsigned char *fetchPtr[8]; /* piece lane % 8 of the line of lane k of this lane's group, rotated by its owner (see k_linear_lanes) */
#pragma unroll
    for (int k = 0; k < 8; k++) fetchPtr[k] = tileL + ((lane & ~7) | k) * kStageLine + (((lane + k) & 7) << 4);
    u32x4 pend[3];
    bool pendOk = false;
    int16_t *pendDst = nullptr;
    int bordOe = o + (1 - skew) * e + oe; /* first lane of a slot: H[0][j] + (o+e), j = t - skew + 1 */
    int rcN = rp[0];
    auto flush = [&]() __attribute__((always_inline)) {
        if (pendOk) {
#pragma unroll
            for (int pl = 0; pl < 3; pl++) stream_store(reinterpret_cast<u32x4 *>(pendDst + (pl << 9)), pend[pl]);
        }
    };
    auto pack8 = [](const int (&v)[R]) __attribute__((always_inline)) -> u32x4 {
        u32x4 x = {pack_lo16(v[0], v[1]), pack_lo16(v[2], v[3]), pack_lo16(v[4], v[5]), pack_lo16(v[6], v[7])};
        return x;
    };
    auto lane_step = [&](const int t, auto kTag) __attribute__((always_inline)) {
        constexpr int K = decltype(kTag)::value; /* t & 7 */
        const int tms = t - skew;
        const int rc = rcN;
        rcN = rp[min(t + 1, rpLast)];
        const int shH = wave_shr1(st.Hoe[R - 1], 0), shD = wave_shr1(st.DeLast, 0);
        const int upHoe = (l == 0) ? bordOe : shH;          /* row-0 border H[0][j] = o + j*e (:50-53) */
        const int upDe = (l == 0) ? (DPX_NEG + e) : shD;    /* virtual D[0][j] */
        bordOe += e;
        if ((unsigned)tms < nEff) {
            int Hv[R], Iv[R], Dv[R];
            aff_cells_g<R>(st, upHoe, upDe, rc, matchG, mismatchG, oe, e, Hv, Iv, Dv);
            if constexpr (STORE) {
                *reinterpret_cast<u32x4 *>(putPtr + 0 * kPlane + (K << 4)) = pack8(Hv);
                *reinterpret_cast<u32x4 *>(putPtr + 1 * kPlane + (K << 4)) = pack8(Iv);
                *reinterpret_cast<u32x4 *>(putPtr + 2 * kPlane + (K << 4)) = pack8(Dv);
            }
        }
        if constexpr (STORE) {
            flush();
            constexpr int O = (K + 1) & 7; /* the owners of the lines that are complete now */
            const uint32_t r = rt[O];
            pendOk = (uint32_t)t >= (r & 0xFFFFu) && (uint32_t)t <= (r >> 16);
            pendDst = waveBase + (size_t)t * kStepElems + (lane << 3);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) pend[pl] = *reinterpret_cast<const u32x4 *>(fetchPtr[O] + pl * kPlane);
        }
    };
    {
        int t = 0;
        for (; t + 8 <= steps; t += 8) {
            lane_step(t + 0, std::integral_constant<int, 0>{}); lane_step(t + 1, std::integral_constant<int, 1>{});
            lane_step(t + 2, std::integral_constant<int, 2>{}); lane_step(t + 3, std::integral_constant<int, 3>{});
            lane_step(t + 4, std::integral_constant<int, 4>{}); lane_step(t + 5, std::integral_constant<int, 5>{});
            lane_step(t + 6, std::integral_constant<int, 6>{}); lane_step(t + 7, std::integral_constant<int, 7>{});
        }
        if (t + 0 < steps) lane_step(t + 0, std::integral_constant<int, 0>{});
        if (t + 1 < steps) lane_step(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < steps) lane_step(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < steps) lane_step(t + 3, std::integral_constant<int, 3>{});
        if (t + 4 < steps) lane_step(t + 4, std::integral_constant<int, 4>{});
        if (t + 5 < steps) lane_step(t + 5, std::integral_constant<int, 5>{});
        if (t + 6 < steps) lane_step(t + 6, std::integral_constant<int, 6>{});
    }
    if constexpr (STORE) flush();
    const int lm = (m - 1) / R, rm = (m - 1) % R;
    if (has && l == lm) {
        int v = st.Hoe[0];
#pragma unroll
        for (int r = 1; r < R; r++) v = (r == rm) ? st.Hoe[r] : v;
        a.score[p] = v - oe; /* scoringMemo[m][n] (:365); the state is H + (o+e) */
        a.endRow[p] = m;
        a.endCol[p] = n;
    }
}

/* =====================================================================================================
 * Banded Smith-Waterman (python/LinearBandedSmithWaterman.py:62-104): the LSW recurrence restricted to
 * |i-j| <= B-1; every cell outside the band (and the borders) reads as 0.
 *
 * The band is walked by anti-diagonals a = i+j.  One anti-diagonal holds at most B in-band cells
 * (slot s = (i-j+B-1)>>1); lane l owns the C = ceil(B/64) slots [l*C, l*C+C), so all 64 lanes work on every
 * step -- no stripes, no skew, no LDS edge row.  With p = (a+B-1)&1 the neighbours on the previous
 * anti-diagonal are   p=1: up = prev[s], left = prev[s+1]      p=0: up = prev[s-1], left = prev[s]
 * and the diagonal is prev2[s]; the one value that crosses a lane boundary moves with a single DPP
 * (wave_shl:1 / wave_shr:1).  Query / reference characters ride along in registers: entering a p=1 step the
 * query window slides one slot down, entering a p=0 step the reference window slides one slot up.
 * ===================================================================================================== */
template <int C>
struct BandState {
    int prev[C], prev2[C]; /* anti-diagonals a-1 and a-2 */
    int qch[C], rch[C];    /* query / reference character of each slot's cell */
    unsigned key[C];       /* running max of (H << 16 | 0xFFFF - A): max score, then earliest step */
    int inBand[2][C];      /* ~0 / 0: is slot s inside the band on a step of parity p (s <= B-1-p)? */
};

/* INTERIOR: every in-band slot of this anti-diagonal lies inside the matrix, so validity is the per-lane constant
 * inBand mask (one v_and) instead of two compares against the step's slot window */
template <int C, bool P1, bool INTERIOR>
__device__ __forceinline__ void band_step(BandState<C> &st, const int A, int &i0, int &j0, const int lane, const int m,
                                          const int n, const int B, const int match, const int mismatch, const int gap,
                                          const unsigned char *qL, const unsigned char *rL, int *out) {
    const int p = P1 ? 1 : 0;
    /* (i0, j0) = row / column of slot 0 on this anti-diagonal, kept incrementally (wave-uniform): entering a p=1 step
     * the row advances, entering a p=0 step the column does.  i0 = (A+2 + p - (B-1)) >> 1 may be <= 0. */
    if constexpr (P1) i0++; else j0++;
    const int smin = INTERIOR ? 0 : max(max(1 - i0, j0 - n), 0);
    const int smax = INTERIOR ? 0 : min(min(m - i0, j0 - 1), B - 1 - p);
    int up[C], left[C];
    if constexpr (P1) {
        /* interior: 1 <= i0 and i0 + (B-1-p) <= m, so the index stays below m + 64 (the staged query has 64 B of slack) */
        const int newq = INTERIOR ? qL[i0 + 64 * C - 2] : qL[min(max(i0 + 64 * C - 2, 0), m - 1)];
        const int tq = wave_shl1(st.qch[0], newq);
#pragma unroll
        for (int c = 0; c < C - 1; c++) st.qch[c] = st.qch[c + 1];
        st.qch[C - 1] = tq;
        const int nb = wave_shl1(st.prev[0], 0);
#pragma unroll
        for (int c = 0; c < C; c++) { up[c] = st.prev[c]; left[c] = (c < C - 1) ? st.prev[c + 1] : nb; }
    } else {
        const int newr = INTERIOR ? rL[j0 - 1] : rL[min(max(j0 - 1, 0), n - 1)]; /* interior: 1 <= j0 <= n */
        const int tr = wave_shr1(st.rch[C - 1], newr);
#pragma unroll
        for (int c = C - 1; c > 0; c--) st.rch[c] = st.rch[c - 1];
        st.rch[0] = tr;
        const int nb = wave_shr1(st.prev[C - 1], 0);
#pragma unroll
        for (int c = 0; c < C; c++) { left[c] = st.prev[c]; up[c] = (c > 0) ? st.prev[c - 1] : nb; }
    }
    const unsigned negA = 0xFFFFu - (unsigned)A;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int sc = (st.qch[c] == st.rch[c]) ? match : mismatch;
        int h = max(max(max(up[c], left[c]) + gap, st.prev2[c] + sc), 0);
        if constexpr (INTERIOR) {
            h &= st.inBand[P1 ? 1 : 0][c];
        } else {
            const int s = lane * C + c;
            h = ((s >= smin) && (s <= smax)) ? h : 0;
        }
        st.key[c] = max(st.key[c], ((unsigned)h << 16) | negA);
        st.prev2[c] = st.prev[c];
        st.prev[c] = h;
        out[c] = h;
    }
}

template <int C, bool PB, bool STORE>
__global__ void __launch_bounds__(DPX_FILL_THREADS) k_banded_fill(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int G = (C >= 8) ? 1 : 8 / C; /* steps per 16-byte store */
    constexpr int GG = (G < 2) ? 2 : G;     /* steps per loop iteration (parity pattern repeats every 2) */
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int p = blockIdx.x * (int)a.wavesPerBlock + wv;
    if (p >= a.numPairs) return;
    if (a.order) p = a.order[p];
    const dpx_pair_dev pr = a.pairs[p];
    const int n = pr.n, m = pr.m, B = a.band;
    const int match = a.match, mismatch = a.mismatch, gap = a.gapOpen;
    if (m <= 0 || n <= 0) {
        if (lane == 0) { a.score[p] = 0; a.endRow[p] = 0; a.endCol[p] = 0; }
        return;
    }
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);
    unsigned char *my = smem + (size_t)wv * a.ldsPerWave;
    const unsigned char *qL = stage_bytes(my, qry, m, lane, 64);            /* 16-byte loads; each buffer has 16 spare bytes for the shift */
    const unsigned char *rL = stage_bytes(my + a.ldsRefOff, ref, n, lane, 64);

    BandState<C> st;
    { /* character windows of the virtual anti-diagonal a = 1 (the first real step then slides one of them) */
        const int p1 = B & 1; /* (1 + B - 1) & 1 */
        const int i0 = (1 + p1 - (B - 1)) >> 1;
        const int j0 = 1 - i0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int s = lane * C + c;
            st.qch[c] = qL[min(max(i0 + s - 1, 0), m - 1)];
            st.rch[c] = rL[min(max(j0 - s - 1, 0), n - 1)];
            st.prev[c] = 0;
            st.prev2[c] = 0;
            st.key[c] = 0u;
            st.inBand[0][c] = (s <= B - 1) ? -1 : 0;
            st.inBand[1][c] = (s <= B - 2) ? -1 : 0;
        }
    }
    /* is every in-band slot of anti-diagonal A inside the matrix?  (true for one contiguous range of A) */
    auto interior = [&](const int A) -> bool {
        const int aa = A + 2, pp = (aa + B - 1) & 1;
        const int i0 = (aa + pp - (B - 1)) >> 1, j0 = aa - i0, top = B - 1 - pp;
        return i0 >= 1 && i0 + top <= m && j0 - top >= 1 && j0 <= n;
    };
    const int NS = m + n - 1;               /* anti-diagonals a = 2 .. m+n */
    const int numGroups = (NS + G - 1) / G;
    int16_t *Hp = a.mat + pr.matOff + (size_t)lane * 8u;
    const size_t cs = pr.chunkStride;
    int acc[8];
    /* slot 0's (row, column) on the virtual anti-diagonal a = 1; band_step advances them */
    int i0 = (1 + (B & 1) - (B - 1)) >> 1;
    int j0 = 1 - i0;
#define DPX_BAND_BODY(INTERIOR_)                                                                                          \
    _Pragma("unroll") for (int g = 0; g < GG; g += 2) {                                                                  \
        band_step<C, PB, INTERIOR_>(st, A0 + g, i0, j0, lane, m, n, B, match, mismatch, gap, qL, rL, &acc[(g % G) * C]);  \
        if constexpr (STORE && G == 1) {                                                                                  \
            if (INTERIOR_ || A0 + g < numGroups) store_tile<8>(Hp + (size_t)(A0 + g) * cs, acc);                         \
        }                                                                                                                 \
        band_step<C, !PB, INTERIOR_>(st, A0 + g + 1, i0, j0, lane, m, n, B, match, mismatch, gap, qL, rL,                 \
                                     &acc[((g + 1) % G) * C]);                                                            \
        if constexpr (STORE) {                                                                                            \
            if (((g + 1) % G) == G - 1) {                                                                                 \
                const int grp = (A0 + g + 1) / G;                                                                         \
                if (INTERIOR_ || grp < numGroups) store_tile<8>(Hp + (size_t)grp * cs, acc);                             \
            }                                                                                                             \
        }                                                                                                                 \
    }
    /* parity of step A is (A + B + 1) & 1; A0 is even, so even steps have parity PB and odd steps !PB.
     * Three phases: head (some slots outside the matrix), interior, tail. */
    int A0 = 0;
    for (; A0 < NS && !(interior(A0) && interior(A0 + GG - 1)); A0 += GG) { DPX_BAND_BODY(false) }
    for (; A0 + GG <= NS && interior(A0 + GG - 1); A0 += GG) { DPX_BAND_BODY(true) }
    for (; A0 < NS; A0 += GG) { DPX_BAND_BODY(false) }
#undef DPX_BAND_BODY
    /* candidates: every slot's first maximum; rows/cols recovered from (step, slot).  Within a slot cells arrive in
     * row-major order, so the earliest step is the slot's first maximum; across slots pick max score, min row, min col */
    unsigned long long mine = 0ull;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int hv = (int)(st.key[c] >> 16);
        if (hv > 0) {
            const int A = 0xFFFF - (int)(st.key[c] & 0xFFFFu);
            const int aa = A + 2;
            const int pp = (aa + B - 1) & 1;
            const int u = 2 * (lane * C + c) + pp;
            const int i = (aa + u - (B - 1)) >> 1;
            const int j = aa - i;
            const unsigned long long k = ((unsigned long long)(unsigned)hv << 40) | ((unsigned long long)(0xFFFFFu - (unsigned)i) << 20) |
                                         (unsigned long long)(0xFFFFFu - (unsigned)j);
            mine = k > mine ? k : mine;
        }
    }
    const unsigned long long top = wave_max_u64(mine);
    if (lane == 0) {
        const int hv = (int)(top >> 40);
        a.score[p] = hv;
        a.endRow[p] = hv > 0 ? (int)(0xFFFFFu - (unsigned)((top >> 20) & 0xFFFFFu)) : 0;
        a.endCol[p] = hv > 0 ? (int)(0xFFFFFu - (unsigned)(top & 0xFFFFFu)) : 0;
    }
}

/* =====================================================================================================
 * "+Opt" for the band: two equal-shaped pairs per wave, pair A in the high and pair B in the low half of every register,
 * the cell update on the packed-int16 pipe as in k_linear_fill_pk (the reference's V18/V19 idea, cuda/LNW/
 * LinearNeedlemanWunschV18.cu:112-341), the anti-diagonal schedule of k_banded_fill unchanged: one DPP move carries both
 * pairs' neighbour cell, the per-lane in-band masks and the head / tail window tests are the same for both (same shape).
 * The one-pair kernel needs ~36 vector instructions per step for its 2 cells per lane (band 128) and sits between the
 * VALU and the write limit; two pairs per instruction leave it to the stores.  SW start cell per half: packed running
 * maximum + the step at which it was first reached (5 packed ops per slot and step, see PkState).
 * ===================================================================================================== */
template <int C>
struct BandStatePk {
    uint32_t prev[C], prev2[C]; /* anti-diagonals a-1 and a-2, {A, B} */
    uint32_t qch[C], rch[C];    /* query / reference characters in 16-bit lanes, {A, B} */
    uint32_t rmax[C], rstep[C]; /* per slot: running maximum of the key (H << 16 | 0xFFFF - step) of pair A / of pair B */
    uint32_t inBand[2][C];      /* ~0 / 0: is slot s inside the band on a step of parity p (s <= B-1-p)? */
};

__device__ __forceinline__ uint32_t pk_chars(const uint16_t both) { return __builtin_amdgcn_perm(0u, (uint32_t)both, 0x0c010c00u); } /* A<<8|B -> {A, B} */

template <int C, bool P1, bool INTERIOR>
__device__ __forceinline__ void band_step_pk(BandStatePk<C> &st, const int A, int &i0, int &j0, const int lane, const int m, const int n,
                                             const int B, const uint32_t matchP, const uint32_t negDeltaP, const uint32_t gapP,
                                             const uint16_t *qL, const uint16_t *rL, uint32_t *out) {
    const int p = P1 ? 1 : 0;
    if constexpr (P1) i0++; else j0++;
    const int smin = INTERIOR ? 0 : max(max(1 - i0, j0 - n), 0);
    const int smax = INTERIOR ? 0 : min(min(m - i0, j0 - 1), B - 1 - p);
    uint32_t up[C], left[C];
    if constexpr (P1) {
        const uint32_t newq = pk_chars(INTERIOR ? qL[i0 + 64 * C - 2] : qL[min(max(i0 + 64 * C - 2, 0), m - 1)]);
        const uint32_t tq = (uint32_t)wave_shl1((int)st.qch[0], (int)newq);
#pragma unroll
        for (int c = 0; c < C - 1; c++) st.qch[c] = st.qch[c + 1];
        st.qch[C - 1] = tq;
        const uint32_t nb = (uint32_t)wave_shl1((int)st.prev[0], 0);
#pragma unroll
        for (int c = 0; c < C; c++) { up[c] = st.prev[c]; left[c] = (c < C - 1) ? st.prev[c + 1] : nb; }
    } else {
        const uint32_t newr = pk_chars(INTERIOR ? rL[j0 - 1] : rL[min(max(j0 - 1, 0), n - 1)]);
        const uint32_t tr = (uint32_t)wave_shr1((int)st.rch[C - 1], (int)newr);
#pragma unroll
        for (int c = C - 1; c > 0; c--) st.rch[c] = st.rch[c - 1];
        st.rch[0] = tr;
        const uint32_t nb = (uint32_t)wave_shr1((int)st.prev[C - 1], 0);
#pragma unroll
        for (int c = 0; c < C; c++) { left[c] = st.prev[c]; up[c] = (c > 0) ? st.prev[c - 1] : nb; }
    }
    const uint32_t onesP = 0x00010001u;
    const uint32_t negA = 0xFFFFu - (uint32_t)A;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const uint32_t differs = dpx::pk_min_u16_raw(st.qch[c] ^ st.rch[c], onesP);
        const s16x2 sc = as_s16x2(dpx::pk_mad_i16_raw(differs, negDeltaP, matchP));
        uint32_t h = as_u32(dpx::pk_max(pk_gap_sat(up[c], left[c], gapP), (s16x2)(as_s16x2(st.prev2[c]) + sc))); /* gap <= 0 (host): gapP = |gap| */
        if constexpr (INTERIOR) {
            h &= st.inBand[P1 ? 1 : 0][c];
        } else {
            const int s = lane * C + c;
            h = ((s >= smin) && (s <= smax)) ? h : 0u;
        }
        st.rmax[c] = max(st.rmax[c], (h & 0xFFFF0000u) | negA);  /* (score, earliest step) keys as in k_banded_fill, one per half */
        st.rstep[c] = max(st.rstep[c], (h << 16) | negA);
        st.prev2[c] = st.prev[c];
        st.prev[c] = h;
        out[c] = h;
    }
}

template <int C, bool PB>
__global__ void __launch_bounds__(DPX_FILL_THREADS) k_banded_fill_pk(const dpx_fill_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int G = (C >= 8) ? 1 : 8 / C;
    constexpr int GG = (G < 2) ? 2 : G;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cpl = blockIdx.x * (int)a.wavesPerBlock + wv; /* couple index */
    if (cpl >= a.numPairs) return;
    const int pA = a.order[2 * cpl], pB = a.order[2 * cpl + 1];
    const dpx_pair_dev prA = a.pairs[pA], prB = a.pairs[pB];
    const int n = prA.n, m = prA.m, B = a.band; /* host guarantees equal shapes, both > 0 */
    const uint32_t matchP = ((uint32_t)(uint16_t)a.match << 16) | (uint16_t)a.match;
    const uint32_t negDeltaP = ((uint32_t)(uint16_t)(a.mismatch - a.match) << 16) | (uint16_t)(a.mismatch - a.match);
    const uint32_t gapP = ((uint32_t)(uint16_t)(-a.gapOpen) << 16) | (uint16_t)(-a.gapOpen); /* |gap| (gap <= 0, checked by the host): pk_gap_sat */
    const unsigned char *refA = reinterpret_cast<const unsigned char *>(a.seq + prA.refIdx), *refB = reinterpret_cast<const unsigned char *>(a.seq + prB.refIdx);
    const unsigned char *qryA = reinterpret_cast<const unsigned char *>(a.seq + prA.qryIdx), *qryB = reinterpret_cast<const unsigned char *>(a.seq + prB.qryIdx);
    unsigned char *my = smem + (size_t)wv * a.ldsPerWave;
    uint16_t *qL = reinterpret_cast<uint16_t *>(my);               /* entry i: query A char << 8 | query B char */
    uint16_t *rL = reinterpret_cast<uint16_t *>(my + a.ldsRefOff); /* entry j: reference A char << 8 | reference B char */
    auto stage2 = [&](uint16_t *dst, const unsigned char *sa, const unsigned char *sb, const int len) {
        for (int x = 4 * lane; x < len; x += 256) { /* four entries per lane and trip (see k_linear_fill_pk) */
            const uint32_t a4 = load4(sa + x), b4 = load4(sb + x);
            uint2 w;
            w.x = __builtin_amdgcn_perm(a4, b4, 0x05010400u);
            w.y = __builtin_amdgcn_perm(a4, b4, 0x07030602u);
            *reinterpret_cast<uint2 *>(dst + x) = w;
        }
    };
    stage2(qL, qryA, qryB, m);
    stage2(rL, refA, refB, n);

    BandStatePk<C> st;
    {
        const int p1 = B & 1;
        const int i0 = (1 + p1 - (B - 1)) >> 1;
        const int j0 = 1 - i0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int s = lane * C + c;
            st.qch[c] = pk_chars(qL[min(max(i0 + s - 1, 0), m - 1)]);
            st.rch[c] = pk_chars(rL[min(max(j0 - s - 1, 0), n - 1)]);
            st.prev[c] = 0u; st.prev2[c] = 0u; st.rmax[c] = 0u; st.rstep[c] = 0u;
            st.inBand[0][c] = (s <= B - 1) ? ~0u : 0u;
            st.inBand[1][c] = (s <= B - 2) ? ~0u : 0u;
        }
    }
    auto interior = [&](const int A) -> bool {
        const int aa = A + 2, pp = (aa + B - 1) & 1;
        const int i0 = (aa + pp - (B - 1)) >> 1, j0 = aa - i0, top = B - 1 - pp;
        return i0 >= 1 && i0 + top <= m && j0 - top >= 1 && j0 <= n;
    };
    const int NS = m + n - 1;
    const int numGroups = (NS + G - 1) / G;
    int16_t *HpA = a.mat + prA.matOff + (size_t)lane * 8u, *HpB = a.mat + prB.matOff + (size_t)lane * 8u;
    const size_t csA = prA.chunkStride, csB = prB.chunkStride;
    uint32_t acc[8];
    int i0 = (1 + (B & 1) - (B - 1)) >> 1;
    int j0 = 1 - i0;
    auto store_group = [&](const int grp) {
        uint4 va, vb;
        va.x = pk_hi16(acc[0], acc[1]); va.y = pk_hi16(acc[2], acc[3]); va.z = pk_hi16(acc[4], acc[5]); va.w = pk_hi16(acc[6], acc[7]);
        vb.x = pack_lo16((int)acc[0], (int)acc[1]); vb.y = pack_lo16((int)acc[2], (int)acc[3]);
        vb.z = pack_lo16((int)acc[4], (int)acc[5]); vb.w = pack_lo16((int)acc[6], (int)acc[7]);
        *reinterpret_cast<uint4 *>(HpA + (size_t)grp * csA) = va;
        *reinterpret_cast<uint4 *>(HpB + (size_t)grp * csB) = vb;
    };
#define DPX_BANDPK_BODY(INTERIOR_)                                                                                          \
    _Pragma("unroll") for (int g = 0; g < GG; g += 2) {                                                                    \
        band_step_pk<C, PB, INTERIOR_>(st, A0 + g, i0, j0, lane, m, n, B, matchP, negDeltaP, gapP, qL, rL, &acc[(g % G) * C]); \
        if constexpr (G == 1) {                                                                                             \
            if (INTERIOR_ || A0 + g < numGroups) store_group(A0 + g);                                                      \
        }                                                                                                                   \
        band_step_pk<C, !PB, INTERIOR_>(st, A0 + g + 1, i0, j0, lane, m, n, B, matchP, negDeltaP, gapP, qL, rL,             \
                                        &acc[((g + 1) % G) * C]);                                                           \
        if (((g + 1) % G) == G - 1) {                                                                                       \
            const int grp = (A0 + g + 1) / G;                                                                               \
            if (INTERIOR_ || grp < numGroups) store_group(grp);                                                            \
        }                                                                                                                   \
    }
    int A0 = 0;
    for (; A0 < NS && !(interior(A0) && interior(A0 + GG - 1)); A0 += GG) { DPX_BANDPK_BODY(false) }
    for (; A0 + GG <= NS && interior(A0 + GG - 1); A0 += GG) { DPX_BANDPK_BODY(true) }
    for (; A0 < NS; A0 += GG) { DPX_BANDPK_BODY(false) }
#undef DPX_BANDPK_BODY
    /* candidates per half: every slot's first maximum; rows / columns recovered from (step, slot) as in k_banded_fill */
    unsigned long long mine[2] = {0ull, 0ull};
#pragma unroll
    for (int c = 0; c < C; c++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const uint32_t key = half ? st.rstep[c] : st.rmax[c];
            const int hv = (int)(key >> 16);
            if (hv > 0) {
                const int A = 0xFFFF - (int)(key & 0xFFFFu);
                const int aa = A + 2;
                const int pp = (aa + B - 1) & 1;
                const int u = 2 * (lane * C + c) + pp;
                const int i = (aa + u - (B - 1)) >> 1;
                const int j = aa - i;
                const unsigned long long k = ((unsigned long long)(unsigned)hv << 40) | ((unsigned long long)(0xFFFFFu - (unsigned)i) << 20) |
                                             (unsigned long long)(0xFFFFFu - (unsigned)j);
                mine[half] = k > mine[half] ? k : mine[half];
            }
        }
    }
    const unsigned long long topA = wave_max_u64(mine[0]), topB = wave_max_u64(mine[1]);
    if (lane == 0) {
        const int hA = (int)(topA >> 40), hB = (int)(topB >> 40);
        a.score[pA] = hA;
        a.endRow[pA] = hA > 0 ? (int)(0xFFFFFu - (unsigned)((topA >> 20) & 0xFFFFFu)) : 0;
        a.endCol[pA] = hA > 0 ? (int)(0xFFFFFu - (unsigned)(topA & 0xFFFFFu)) : 0;
        a.score[pB] = hB;
        a.endRow[pB] = hB > 0 ? (int)(0xFFFFFu - (unsigned)((topB >> 20) & 0xFFFFFu)) : 0;
        a.endCol[pB] = hB > 0 ? (int)(0xFFFFFu - (unsigned)(topB & 0xFFFFFu)) : 0;
    }
}

/* =====================================================================================================
 * Export: un-tile one pair's plane into the reference's row-major (m+1) x (n+1) layout, borders included.
 * ===================================================================================================== */
__global__ void k_export_matrix(const int16_t *mat, dpx_pair_dev pr, int algo, int R, int planes, int plane, int gapOpen,
                                int gapExtend, int band, int16_t *out) {
    const int n = pr.n, m = pr.m;
    if (pr.rows) R = pr.rows;
    const size_t total = (size_t)(m + 1) * (size_t)(n + 1);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / (size_t)(n + 1));
        const int j = (int)(idx % (size_t)(n + 1));
        int v;
        if (i == 0 || j == 0) {
            const int len = i + j; /* one of them is 0 */
            if (plane != 0) v = 0;                                   /* I / D are zero-initialised (ANW.cpp:24-27) */
            else if (algo == DPX_K_LNW) v = len * gapOpen;           /* LNW.cpp:31-41 */
            else if (algo == DPX_K_ANW) v = len == 0 ? 0 : gapOpen + len * gapExtend; /* ANW.cpp:43-53 */
            else v = 0;                                              /* LSW / BSW */
        } else if (algo == DPX_K_BSW) {
            const int dlt = i - j;
            v = (dlt <= band - 1 && -dlt <= band - 1) ? mat[pr.matOff + dpx_band_index(i, j, band, pr.chunkStride)] : 0;
        } else {
            v = mat[pr.matOff + dpx_cell_index(i, j, n, R, plane, planes, pr.chunkStride, pr.lanes)];
        }
        out[idx] = (int16_t)v;
    }
}

/* =====================================================================================================
 * Device traceback (the reference's on-device backtracking(), cuda/LNW/
 * LinearNeedlemanWunschV19.cu:26-110, and host back-trackers c++/backtrack.cpp:21-356).
 * One lane per pair walks back from the end cell.  Directions are not stored: they are recomputed from the
 * int16 score matrices with the reference's own tie rules (a >= b wins for the FIRST argument of __vibmax):
 *   LSW (c++/LinearSmithWaterman.cpp:106-108): UPPER, then LEFT, then CORNER; stop when the next cell is 0 (:222)
 *   LNW (c++/LinearNeedlemanWunsch.cpp:122-125): INSERTION if left >= max(up, diag), else DELETION if up >= diag
 *   ANW (c++/AffineNeedlemanWunsch.cpp:185-233, :258-360): 3-state walk over H / I / D, GAP_OPEN wins ties
 * The three lines (reference / relation / query) are written right-aligned into the pair's buffer.
 * ===================================================================================================== */
struct TbView {
    const int16_t *mat;
    uint64_t off;
    uint32_t cs, lanes;
    int n, m, R, planes, algo, band, gapOpen, gapExtend;
    __device__ __forceinline__ int get(int i, int j, int plane) const {
        if (i == 0 || j == 0) { /* closed-form borders, as in k_export_matrix */
            const int len = i + j;
            if (plane != 0) return 0;
            if (algo == DPX_K_LNW) return len * gapOpen;
            if (algo == DPX_K_ANW) return len == 0 ? 0 : gapOpen + len * gapExtend;
            return 0;
        }
        if (algo == DPX_K_BSW) {
            const int dlt = i - j;
            if (dlt > band - 1 || -dlt > band - 1) return 0;
            return mat[off + dpx_band_index(i, j, band, cs)];
        }
        return mat[off + dpx_cell_index(i, j, n, R, plane, planes, cs, lanes)];
    }
};

/* A byte string read back to front through one register: four characters per aligned dword load (the walk reads
 * query[i-1] and reference[j-1] on every step; with ~100k lanes in flight neither L1 nor L2 keeps a sector between two
 * steps of the same lane, so every byte load was an HBM sector fetch -- profiles/README.md). */
struct CharWin {
    const unsigned char *s;
    uintptr_t at = 1; /* address of the dword held in `w` (1 = none) */
    uint32_t w = 0;
    __device__ __forceinline__ int get(int x) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(s + x), al = a & ~(uintptr_t)3;
        if (al != at) { at = al; w = *reinterpret_cast<const uint32_t *>(al); } /* never leaves the 256-byte aligned arena */
        return (int)((w >> (8 * (int)(a & 3))) & 0xFFu);
    }
};

/* The H plane seen through two registers-resident column vectors.  In the wavefront-tiled and the tile layout with R >= 8 the 8 rows
 * (i0 & ~7 .. +7) of one column are 16 contiguous, 16-byte aligned bytes: `cur` holds them for column j, `prev` for
 * column j-1.  A path step then needs at most ONE new 16-byte load (the next column on a left / diagonal move; two when
 * the walk climbs into the 8-row group above) instead of three 2-byte loads that each miss every cache. */
struct TileWalker {
    const int16_t *mat;
    uint64_t off;
    uint32_t cs, lanes;
    int n, sr, border; /* sr = log2(rows per lane); border: value of H on row 0 / column 0 is border * (i + j) */
    int i = 0, j = 0;
    u32x4 cur, prev;

    __device__ __forceinline__ u32x4 column(int i0, int jj) const { /* the 8-row group of row i0 (0-based), column jj >= 1 */
        const int r = i0 & ((1 << sr) - 1), sub = r >> 3;
        uint64_t T, tile;
        if (lanes == 16) { /* tile layout: the column's 8 rows are one 16-byte piece of a line of the wave's stream (cs = first lane) */
            return *reinterpret_cast<const u32x4 *>(mat + off + (dpx_wtile_index(i0 + 1, jj, 0, 1, 1 << sr, cs) & ~(uint64_t)7));
        } else {
            const int k = i0 >> (sr + 6), l = (i0 >> sr) & 63;
            T = (uint64_t)k * (uint64_t)n + (uint64_t)(jj - 1) + (uint64_t)l;
            tile = (uint64_t)(((sub << 6) + l) << 3);
        }
        return *reinterpret_cast<const u32x4 *>(mat + off + T * (uint64_t)cs + tile);
    }
    static __device__ __forceinline__ int elem(const u32x4 &v, int rr) { /* int16 number rr (0..7) of the vector */
        const uint32_t d = rr < 4 ? (rr < 2 ? v.x : v.y) : (rr < 6 ? v.z : v.w);
        return (int)(int16_t)(d >> (16 * (rr & 1)));
    }
    __device__ __forceinline__ void load_both() {
        if (i >= 1 && j >= 1) cur = column(i - 1, j);
        if (i >= 1 && j >= 2) prev = column(i - 1, j - 1);
    }
    __device__ __forceinline__ void start(int i_, int j_) { i = i_; j = j_; load_both(); }
    /* a single cell outside the two cached columns' 8-row group (the row above the group): plain 2-byte load */
    __device__ __forceinline__ int single(int ii, int jj) const {
        if (ii == 0 || jj == 0) return border * (ii + jj);
        return mat[off + dpx_cell_index(ii, jj, n, 1 << sr, 0, 1, cs, lanes)];
    }
    __device__ __forceinline__ int here() const { return (i == 0 || j == 0) ? border * (i + j) : elem(cur, (i - 1) & 7); }
    __device__ __forceinline__ int up() const {
        if (i <= 1 || j == 0) return border * (i - 1 + j);
        const int rr = (i - 1) & 7;
        return rr ? elem(cur, rr - 1) : single(i - 1, j);
    }
    __device__ __forceinline__ int left() const {
        if (j <= 1 || i == 0) return border * (i + j - 1);
        return elem(prev, (i - 1) & 7);
    }
    __device__ __forceinline__ int diag() const {
        if (i <= 1 || j <= 1) return border * (i - 1 + j - 1);
        const int rr = (i - 1) & 7;
        return rr ? elem(prev, rr - 1) : single(i - 1, j - 1);
    }
    __device__ __forceinline__ void move(bool rowUp, bool colLeft) {
        const bool newGroup = rowUp && (((i - 1) & 7) == 0); /* leaving the 8-row group through its top row */
        i -= rowUp ? 1 : 0;
        j -= colLeft ? 1 : 0;
        if (i == 0 || j == 0) return; /* on the border: closed form from here on */
        if (newGroup) {
            load_both();
        } else if (colLeft) {
            cur = prev;
            if (j >= 2) prev = column(i - 1, j - 1);
        }
    }
};

/* one lane walks one pair */
__device__ void tb_walk_lane(const dpx_fill_args &a, const int p, int algo, int R, int planes, int cachedWalk, const int32_t *endRow,
                             const int32_t *endCol, const uint64_t *tbOff, char *tb, int32_t *tbLen) {
    const dpx_pair_dev pr = a.pairs[p];
    const int n = pr.n, m = pr.m;
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);
    const int cap = (m + n + 1 + 3) & ~3; /* line capacity, dword-aligned like tbOff[] (see EMIT) */
    char *lr = tb + tbOff[p], *lx = lr + cap, *lq = lx + cap;
    int pos = cap; /* lines grow from the back */
    uint32_t accR = 0, accX = 0, accQ = 0; /* the last <= 4 characters of each line, earliest in the highest byte */
    const int match = a.match, mismatch = a.mismatch;
    TbView v{a.mat, pr.matOff, pr.chunkStride, pr.lanes, n, m, pr.rows ? (int)pr.rows : R, planes, algo, a.band, a.gapOpen, a.gapExtend};
    /* Characters are produced back to front; four of them are collected per line and written as one aligned dword
     * (three scattered dword stores per four path steps instead of twelve byte stores). */
#define EMIT(rc_, xc_, qc_)                                                                      \
    {                                                                                            \
        --pos;                                                                                   \
        accR = (accR << 8) | (uint32_t)(unsigned char)(rc_);                                     \
        accX = (accX << 8) | (uint32_t)(unsigned char)(xc_);                                     \
        accQ = (accQ << 8) | (uint32_t)(unsigned char)(qc_);                                     \
        if ((pos & 3) == 0) {                                                                    \
            *reinterpret_cast<uint32_t *>(lr + pos) = accR;                                      \
            *reinterpret_cast<uint32_t *>(lx + pos) = accX;                                      \
            *reinterpret_cast<uint32_t *>(lq + pos) = accQ;                                      \
        }                                                                                        \
    }
    int i = endRow[p], j = endCol[p];
    CharWin qw{qry}, rw{ref};
    /* register-cached columns: LSW / LNW on the wavefront-tiled or the tile layout with 8-row sub-tiles (rows per lane >= 8), for
     * batches with enough lanes in flight that the walk is bound by sector requests (measured: 100k short pairs -25 %);
     * smaller batches are bound by the latency of one dependent load per step instead, and there the plain
     * three-loads-in-parallel step is the shorter chain (5000 x 1024^2: cached +40 %, 20k x 300^2: +10 %).  The host decides. */
    const bool cached = (algo == DPX_K_LSW || algo == DPX_K_LNW) && v.R >= 8 && cachedWalk && pr.lanes != 32;
    if (cached) {
        const int g = a.gapOpen;
        TileWalker w{a.mat, pr.matOff, pr.chunkStride, pr.lanes, n, dpx_log2(v.R), algo == DPX_K_LNW ? g : 0};
        w.start(i, j);
        if (algo == DPX_K_LSW) {
            int h = (i > 0 && j > 0) ? w.here() : 0;
            while (h > 0) {
                const int up = w.up(), left = w.left();
                if (up + g == h) { EMIT('_', ' ', qw.get(w.i - 1)); w.move(true, false); h = up; }
                else if (left + g == h) { EMIT(rw.get(w.j - 1), ' ', '_'); w.move(false, true); h = left; }
                else {
                    const int dg = w.diag(), qc = qw.get(w.i - 1), rc = rw.get(w.j - 1);
                    EMIT(rc, qc == rc ? '*' : '|', qc);
                    w.move(true, true);
                    h = dg;
                }
            }
        } else {
            while (w.i != 0 || w.j != 0) {
                if (w.i == 0) { EMIT(rw.get(w.j - 1), ' ', '_'); w.move(false, true); continue; }  /* row-0 border: QUERY_INSERTION */
                if (w.j == 0) { EMIT('_', ' ', qw.get(w.i - 1)); w.move(true, false); continue; }  /* column-0 border: QUERY_DELETION */
                const int qc = qw.get(w.i - 1), rc = rw.get(w.j - 1);
                const bool eq = qc == rc;
                const int mm = w.diag() + (eq ? match : mismatch);
                const int del = w.up() + g, ins = w.left() + g;
                const int vmax = max(del, mm);
                if (ins >= vmax) { EMIT(rc, ' ', '_'); w.move(false, true); }
                else if (del >= mm) { EMIT('_', ' ', qc); w.move(true, false); }
                else { EMIT(rc, eq ? '*' : '|', qc); w.move(true, true); }
            }
        }
    } else if (algo == DPX_K_LSW || algo == DPX_K_BSW) {
        const int g = a.gapOpen;
        int h = (i > 0 && j > 0) ? v.get(i, j, 0) : 0;
        while (h > 0) {
            const int up = v.get(i - 1, j, 0), left = v.get(i, j - 1, 0), dg = v.get(i - 1, j - 1, 0);
            if (up + g == h) { EMIT('_', ' ', qw.get(i - 1)); i--; h = up; }
            else if (left + g == h) { EMIT(rw.get(j - 1), ' ', '_'); j--; h = left; }
            else { const int qc = qw.get(i - 1), rc = rw.get(j - 1); EMIT(rc, qc == rc ? '*' : '|', qc); i--; j--; h = dg; }
        }
    } else if (algo == DPX_K_LNW) {
        const int g = a.gapOpen;
        while (i != 0 || j != 0) {
            if (i == 0) { EMIT(rw.get(j - 1), ' ', '_'); j--; continue; }  /* row-0 border: QUERY_INSERTION */
            if (j == 0) { EMIT('_', ' ', qw.get(i - 1)); i--; continue; }  /* column-0 border: QUERY_DELETION */
            const int qc = qw.get(i - 1), rc = rw.get(j - 1);
            const bool eq = qc == rc;
            const int mm = v.get(i - 1, j - 1, 0) + (eq ? match : mismatch);
            const int del = v.get(i - 1, j, 0) + g, ins = v.get(i, j - 1, 0) + g;
            const int vmax = max(del, mm);
            if (ins >= vmax) { EMIT(rc, ' ', '_'); j--; }
            else if (del >= mm) { EMIT('_', ' ', qc); i--; }
            else { EMIT(rc, eq ? '*' : '|', qc); i--; j--; }
        }
    } else { /* ANW */
        const int o = a.gapOpen, e = a.gapExtend;
        int cur = 0; /* 0 SCORING, 1 INSERTION, 2 DELETION */
        while (i != 0 && j != 0) {
            if (cur == 0) {
                const bool eq = qw.get(i - 1) == rw.get(j - 1);
                const int mm = v.get(i - 1, j - 1, 0) + (eq ? match : mismatch);
                const int D = v.get(i, j, 2), I = v.get(i, j, 1);
                const int vmax = max(D, mm);
                if (I >= vmax) cur = 1;
                else if (D >= mm) cur = 2;
                else { EMIT(rw.get(j - 1), eq ? '*' : '|', qw.get(i - 1)); i--; j--; }
            } else if (cur == 1) {
                const bool open = (j == 1) || (v.get(i, j - 1, 0) + o + e >= v.get(i, j - 1, 1) + e);
                if (open) cur = 0;
                EMIT(rw.get(j - 1), ' ', '_'); j--;
            } else {
                const bool open = (i == 1) || (v.get(i - 1, j, 0) + o + e >= v.get(i - 1, j, 2) + e);
                if (open) cur = 0;
                EMIT('_', ' ', qw.get(i - 1)); i--;
            }
        }
        while (i > 0) { EMIT('_', ' ', qw.get(i - 1)); i--; }
        while (j > 0) { EMIT(rw.get(j - 1), ' ', '_'); j--; }
    }
#undef EMIT
    if (pos & 3) { /* the 1-3 newest characters have not filled a dword: the newest sits in the lowest byte, at `pos` */
        const int left = 4 - (pos & 3);
        for (int t = 0; t < left; t++) {
            lr[pos + t] = (char)(accR >> (8 * t)); lx[pos + t] = (char)(accX >> (8 * t)); lq[pos + t] = (char)(accQ >> (8 * t));
        }
    }
    tbLen[p] = cap - pos;
}

__global__ void k_traceback(const dpx_fill_args a, int numPairs, int algo, int R, int planes, int cachedWalk, const int32_t *endRow,
                            const int32_t *endCol, const uint64_t *tbOff, char *tb, int32_t *tbLen) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= numPairs) return;
    tb_walk_lane(a, p, algo, R, planes, cachedWalk, endRow, endCol, tbOff, tb, tbLen);
}

/* -----------------------------------------------------------------------------------------------------
 * Wave-cooperative traceback (LSW / LNW / ANW / banded SW, every matrix layout of the fill kernels).
 * The lane-per-pair walk above pays one dependent HBM round trip per path step (1100 of them on a 1024 x 1024 pair).  Here
 * one WAVE owns a pair: lane c fetches column cLo + c of a window of the matrix around the walker -- 64 rows (ANW: 48 rows of
 * every plane) x 64 columns, the 8 rows of a row group with one 16-byte load (layouts with fewer than 8 rows per lane: 8- or
 * 4-byte pieces) -- into LDS (one 144-byte line per column and plane, borders included as ordinary cells), and the walk runs
 * until it leaves the window through its top or its left edge: one HBM round trip per ~64 steps of a diagonal path.
 * Round 4: the walk takes RUNS, not steps.  Round 3 walked on the scalar unit, one cell per trip: an LDS round trip for three
 * scores and two characters, five v_readfirstlane and the branches on them -- about 340 issue cycles per path step, and the
 * traceback of a batch cost as much as its fill (10 000 pairs of 1024 x 1024: 4.4 ms against 3.2 ms).  Which way the path leaves a
 * cell depends on that cell's neighbourhood only, so the 64 lanes each decide ONE cell of the line the path would follow next --
 * lane l the cell of the walker's diagonal in column l; in ANW's gap states the cells of the walker's row (INSERTION) or column
 * (DELETION) -- from four LDS reads at their own address, and a ballot gives the number of steps the path really follows that
 * line: the length of the run of "diagonal" decisions below the walker's lane (LSW: up / left / diagonal / stop at H = 0; LNW:
 * INSERTION over DELETION over diagonal, borders included; ANW: the SCORING state's choice, "the gap was opened here" for the two
 * gap states, c++/backtrack.cpp:214-356).  The lanes of the run store their own three characters.  Alignments worth computing
 * are mostly long diagonal runs: a 1024 x 1024 pair of the benchmark is ~90 trips instead of ~2050 -- ~50 since a trip also takes the
 * step that ENDS its run (the lane below the run has decided that cell in the same pass: a diagonal run and the single gap step
 * after it are one trip).  Measured per wave (s_memrealtime around the phases, 1024 x 1024 LSW): 19 windows of ~1.3 us load latency
 * on an idle chip (4 us with 1700 waves loading at once: ~200 distinct lines per wave and window against the CUs' miss queues) and
 * ~50 trips of 0.43 us; the batch's kernel time is its SLOWEST pair -- the benchmark's 1 % unrelated pairs, whose alignments under
 * +3 / -1 / -2 are 300 short runs: 0.19-0.25 ms against 0.06-0.13 for the others.  A prefetch of the next window along the diagonal
 * was built and measured (19 of 21 windows adopted): no gain with 1700 waves in flight (the loads are queued, not late), removed.
 * Banded SW (round 4, BAND): the same walk with LSW's rule; the window is gathered from the anti-diagonal-major band layout with
 * 2-byte loads (issue<3, .>), cells outside the band are 0.  4000 pairs of 4096 x 4096, band 128: 1.7 ms per batch of 1712 against
 * 10 ms for one lane per pair.
 * ----------------------------------------------------------------------------------------------------- */
template <int PLANES> struct TbWin {
    static constexpr int G = PLANES == 3 ? 6 : 8;      /* row groups of a window */
    static constexpr int WR = 8 * G;                   /* rows R0+1 .. R0+WR; columns cLo .. cLo+63, one per lane */
    static constexpr int CS = WR + 8;                  /* int16 elements between two columns in LDS: 16-byte aligned lines, banks spread */
    static constexpr int kBytes = PLANES * 64 * CS * 2;
};
size_t dpx_traceback_wave_lds(int planes) { return planes == 3 ? (size_t)TbWin<3>::kBytes : (size_t)TbWin<1>::kBytes; }

/* the 8 rows 8*grp+1 .. 8*grp+8 of column jc (>= 1) of `plane`, whatever the layout: one 16-byte piece where a lane owns >= 8 rows
 * (wavefront-tiled with 8-row sub-tiles, tile layout), otherwise the pieces of 8 / Rr neighbouring lanes (they sit in different chunks) */
__device__ __forceinline__ u32x4 tb_load_group8(const int16_t *base, const dpx_pair_dev &pr, const int Rr, const int planes, const int plane,
                                                const int grp, const int jc, const int n) {
    const int i1 = grp * 8 + 1;
    if (Rr >= 8) return *reinterpret_cast<const u32x4 *>(base + dpx_cell_index(i1, jc, n, Rr, plane, planes, pr.chunkStride, pr.lanes));
    if (Rr == 4) {
        const uint2 lo = *reinterpret_cast<const uint2 *>(base + dpx_cell_index(i1, jc, n, 4, plane, planes, pr.chunkStride, pr.lanes));
        const uint2 hi = *reinterpret_cast<const uint2 *>(base + dpx_cell_index(i1 + 4, jc, n, 4, plane, planes, pr.chunkStride, pr.lanes));
        return u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    u32x4 v;
    v.x = *reinterpret_cast<const uint32_t *>(base + dpx_cell_index(i1, jc, n, 2, plane, planes, pr.chunkStride, pr.lanes));
    v.y = *reinterpret_cast<const uint32_t *>(base + dpx_cell_index(i1 + 2, jc, n, 2, plane, planes, pr.chunkStride, pr.lanes));
    v.z = *reinterpret_cast<const uint32_t *>(base + dpx_cell_index(i1 + 4, jc, n, 2, plane, planes, pr.chunkStride, pr.lanes));
    v.w = *reinterpret_cast<const uint32_t *>(base + dpx_cell_index(i1 + 6, jc, n, 2, plane, planes, pr.chunkStride, pr.lanes));
    return v;
}

template <int PLANES, bool BAND, int ALGO>
__global__ void __launch_bounds__(64) k_traceback_wave(const dpx_fill_args a, int numPairs, int R, const int32_t *endRow,
                                                       const int32_t *endCol, const uint64_t *tbOff, char *tb, int32_t *tbLen) {
    using W = TbWin<PLANES>;
    /* BAND: a banded SW matrix (anti-diagonal-major band layout, dpx_band_index) walked by LSW's rule; cells outside the band read 0
     * (TbView::get, c++/BandedSmithWaterman.cpp's back-tracker sees the zero-initialised matrix there) */
    constexpr int algo = BAND ? DPX_K_LSW : ALGO; /* (compile-time: the walk of one algorithm carries no branches for the others) */
    constexpr int G = W::G, WR = W::WR, CS = W::CS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemTb[];
    int16_t *win = reinterpret_cast<int16_t *>(smemTb); /* win[(plane * 64 + (jj - cLo)) * CS + (ii - R0 - 1)] = plane[ii][jj] */
    const int p = blockIdx.x;
    const int lane = threadIdx.x;
    if (p >= numPairs) return;
    const dpx_pair_dev pr = a.pairs[p];
    const int n = pr.n, m = pr.m;
    const int Rr = pr.rows ? (int)pr.rows : R;
    /* (the host launches this kernel for LSW / LNW / banded SW with one plane and ANW with three; rows per lane are
     * 2, 4, 8 or 16 in every full-matrix layout; empty sequences walk along a border or not at all) */
    const unsigned char *ref = reinterpret_cast<const unsigned char *>(a.seq + pr.refIdx);
    const unsigned char *qry = reinterpret_cast<const unsigned char *>(a.seq + pr.qryIdx);
    const int16_t *base = a.mat + pr.matOff;
    const int cap = (m + n + 1 + 3) & ~3;
    char *lr = tb + tbOff[p], *lx = lr + cap, *lq = lx + cap;
    int pos = cap;
    const int match = a.match, mismatch = a.mismatch, g = a.gapOpen, ext = a.gapExtend;
    /* H on row 0 / column 0, `len` cells from the corner (TbView::get): LNW len * gap, ANW open + len * extend (0 in the corner), LSW 0 */
    auto bval = [&](const int len) -> int { return algo == DPX_K_LNW ? len * g : (algo == DPX_K_ANW ? (len ? g + len * ext : 0) : 0); };
    /* where the 16-byte piece (8 rows of a row group, one column) lies: shifts only for the two layouts whose lanes own >= 8 rows */
    const int Q = Rr >> 3, lgQ = Q == 2 ? 1 : 0;
    const int kind = Rr >= 8 ? (pr.lanes == 64 ? 0 : pr.lanes == 16 ? 1 : 2) : 2;
    const uint32_t cs = pr.chunkStride; /* (tile layout: the pair's first lane) */
    const int swBand = a.band, bandSc = BAND ? dpx_log2(dpx_band_cpl(a.band)) : 0; /* banded SW: band width, log2(cells per lane) */
    /* where the 16-byte piece (rows 8*grp+1 .. +8 of column jc, plane pl) lies, for the two layouts whose lanes own >= 8 rows */
    auto piece_at = [&](auto kindC, const int pl, const int grp, const int jc) -> const int16_t * {
        if constexpr (decltype(kindC)::value == 0) { /* wavefront-tiled (dpx_tiled_index + dpx_tile_off) */
            const int l = (grp >> lgQ) & 63, kk = grp >> (lgQ + 6), sub = grp & (Q - 1);
            const size_t T = (size_t)kk * (size_t)n + (size_t)(jc - 1) + (size_t)l;
            return base + T * cs + (size_t)(((((pl << lgQ) + sub) << 6) + l) << 3);
        } else { /* tile layout of the lane-packed kernels (dpx_wtile_index) */
            const int l = grp >> lgQ, h = grp & (Q - 1), lam = (int)cs + l, skew = l + ((int)cs & 7), j0 = jc - 1;
            const size_t t = (size_t)(((j0 >> 3) + 1) * 8 + skew - 1);
            return base + t * (size_t)(PLANES * Q * 512) + (size_t)((((pl << lgQ) + h) << 9) + ((lam >> 3) << 6) + ((j0 & 7) << 3));
        }
    };
    int i = __builtin_amdgcn_readfirstlane(endRow[p]), j = __builtin_amdgcn_readfirstlane(endCol[p]);
    int R0 = 1 << 28, cLo = 1 << 28;
    int diag0 = 0;        /* LSW / LNW: i - j of the cell the window was anchored on */
    bool banded = false;  /* ... and whether only the band around that diagonal was fetched */
    bool wantFull = false; /* the walk left the last band sideways (a long gap): fetch whole columns next time */
    uint32_t chR = 0u, chQ = 0u; /* reference character of this lane's column; query character of window row `lane` */
    /* window with (ii, jj) in its bottom-right corner region */
    /* Window with (ii, jj) in its bottom-right corner region.  Only a BAND of it is fetched unless the walk left the last one sideways
     * -- the three row groups of every column (and plane) around the diagonal through the anchor (rows d-8 .. d+7 at least, d = the
     * diagonal's row in that column).  The walk follows that diagonal or leaves it by a few gap steps; need_window() re-anchors when it
     * is more than 7 rows above / 6 below.  In the wavefront-tiled layouts every column's piece lies in its own 64-byte sector, so 3
     * instead of 8 pieces per column are 3/8 of the traffic and of the requests (the traceback of 10 000 pairs of 1024 x 1024 read
     * ~5 GB for paths that touch ~0.3 GB).
     * fill_window<KIND, CNT>: the loads of one layout (0 wavefront-tiled, 1 tile layout, 2 any layout through dpx_cell_index, 3 band
     * layout) and one group count, straight-line -- round 4 found a window's loads spending ~3000 cycles in the nest of uniform
     * branches that one loop over "whatever layout, however many groups" compiled to (a taken branch is a fetch bubble).  Cells
     * without storage (rows past m, columns outside 1..n, outside the band) load the pool's first bytes instead of branching and are
     * masked to 0 afterwards. */
    constexpr int GL = 3; /* row groups of a banded column */
    /* issue<KIND, CNT>: the loads of a window -- CNT row groups from group gBase + gFirst (per lane) of column jc, the query characters of
     * rows r0 + 1 + lane and the reference character of the column -- into `raw`, nothing else: no value is touched, so every load of the window is
     * in flight before the wave waits for the first. */
    auto issue = [&](auto kindC, auto cntC, auto &raw, uint32_t &rawQ, uint32_t &rawR, const int gBase, const int gFirst, const int jc, const int r0) {
        constexpr int KIND = decltype(kindC)::value, CNT = decltype(cntC)::value;
        if constexpr (KIND == 3) {
            /* Band layout (dpx_band_index, its shifts hoisted): the rows of a column lie on consecutive anti-diagonals -- 2-byte loads,
             * 24 per lane in a banded window (neighbouring columns and rows share 64-byte sectors: ~60 distinct ones for the wave). */
            const int sg = 3 - bandSc;
#pragma unroll
            for (int gi = 0; gi < CNT; gi++) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int ii2 = (gBase + gFirst + gi) * 8 + 1 + e, dlt = ii2 - jc, A = ii2 + jc - 2, sl = (dlt + swBand - 1) >> 1;
                    const bool ok = ii2 >= 1 && ii2 <= m && jc >= 1 && jc <= n && dlt < swBand && -dlt < swBand;
                    const size_t off = (size_t)(A >> sg) * cs + (size_t)(((sl >> bandSc) << 3) + ((A & ((1 << sg) - 1)) << bandSc) + (sl & ((1 << bandSc) - 1)));
                    const int16_t *at = ok ? base + off : a.mat;
                    raw[gi * 8 + e] = (uint32_t)*reinterpret_cast<const uint16_t *>(at);
                }
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < PLANES; pl++) {
#pragma unroll
                for (int gi = 0; gi < CNT; gi++) {
                    const int grp = gBase + gFirst + gi;
                    const bool ok = grp >= 0 && grp * 8 < m && jc >= 1 && jc <= n;
                    u32x4 x;
                    if constexpr (KIND == 2) {
                        x = u32x4{0u, 0u, 0u, 0u};
                        if (ok) x = tb_load_group8(base, pr, Rr, PLANES, pl, grp, jc, n);
                    } else {
                        const int16_t *at = ok ? piece_at(kindC, pl, grp, jc) : a.mat;
                        x = *reinterpret_cast<const u32x4 *>(at);
                    }
                    const int k = (pl * CNT + gi) * 4;
                    raw[k] = x.x; raw[k + 1] = x.y; raw[k + 2] = x.z; raw[k + 3] = x.w;
                }
            }
        }
        const int qi = r0 + lane;
        const unsigned char *atQ = (lane < WR && qi >= 0 && qi < m) ? qry + qi : reinterpret_cast<const unsigned char *>(a.seq);
        const unsigned char *atR = (jc >= 1 && jc <= n) ? ref + (jc - 1) : reinterpret_cast<const unsigned char *>(a.seq);
        rawQ = *atQ;
        rawR = *atR;
    };
    /* commit<KIND, CNT>: what issue() loaded becomes the window in LDS -- cells without storage masked to 0, the borders of H added as
     * ordinary cells -- and the lane's two characters */
    auto commit = [&](auto kindC, auto cntC, const auto &raw, const uint32_t rawQ, const uint32_t rawR, const int gBase, const int gFirst, const int jc, const int r0) {
        constexpr int KIND = decltype(kindC)::value, CNT = decltype(cntC)::value;
        u32x4 v[PLANES][CNT];
        if constexpr (KIND == 3) {
#pragma unroll
            for (int gi = 0; gi < CNT; gi++) {
                uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int ii2 = (gBase + gFirst + gi) * 8 + 1 + e, dlt = ii2 - jc;
                    const bool ok = ii2 >= 1 && ii2 <= m && jc >= 1 && jc <= n && dlt < swBand && -dlt < swBand;
                    d[e >> 1] |= (ok ? raw[gi * 8 + e] : 0u) << ((e & 1) * 16);
                }
                v[0][gi] = u32x4{d[0], d[1], d[2], d[3]};
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < PLANES; pl++) {
#pragma unroll
                for (int gi = 0; gi < CNT; gi++) {
                    const int grp = gBase + gFirst + gi, k = (pl * CNT + gi) * 4;
                    const bool ok = grp >= 0 && grp * 8 < m && jc >= 1 && jc <= n;
                    v[pl][gi] = ok ? u32x4{raw[k], raw[k + 1], raw[k + 2], raw[k + 3]} : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        { const int qi = r0 + lane; chQ = (lane < WR && qi >= 0 && qi < m) ? rawQ : 0u; } /* query character of row r0 + 1 + lane */
        chR = (jc >= 1 && jc <= n) ? rawR : 0u;
        if (algo != DPX_K_LSW) { /* the borders of H as cells: column 0 and row 0 (LSW: zeros, as loaded) */
#pragma unroll
            for (int gi = 0; gi < CNT; gi++) {
                const int grp = gBase + gFirst + gi;
                if (jc == 0 || grp < 0) {
                    uint32_t d[4];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const int r0b = grp * 8 + 1 + e, r1b = r0b + 1;
                        const int v0 = jc == 0 ? (r0b >= 0 ? bval(r0b) : 0) : (r0b == 0 && jc > 0 ? bval(jc) : 0);
                        const int v1 = jc == 0 ? (r1b >= 0 ? bval(r1b) : 0) : (r1b == 0 && jc > 0 ? bval(jc) : 0);
                        d[e >> 1] = ((uint32_t)(uint16_t)v1 << 16) | (uint32_t)(uint16_t)v0;
                    }
                    v[0][gi] = u32x4{d[0], d[1], d[2], d[3]};
                }
            }
        }
#pragma unroll
        for (int pl = 0; pl < PLANES; pl++)
#pragma unroll
            for (int gi = 0; gi < CNT; gi++) *reinterpret_cast<u32x4 *>(win + (pl * 64 + lane) * CS + (gFirst + gi) * 8) = v[pl][gi];
    };
    /* first fetched row group (relative to the window's first) of this lane's column in a banded window whose last column holds the
     * diagonal's row iiDiag */
    auto band_first = [&](const int iiDiag, const int r0) -> int {
        const int dl = (iiDiag - r0 - 1) - 63 + lane;
        return min(max((dl - 8) >> 3, 0), G - GL);
    };
    constexpr int kRawFull = BAND ? G * 8 : PLANES * G * 4, kRawBand = BAND ? GL * 8 : PLANES * GL * 4;
    using std::integral_constant;
    auto load_window = [&](const int ii, const int jj) {
        const int gBase = ((ii - 1) >> 3) - (G - 1);
        R0 = gBase * 8;
        cLo = jj - 63;
        const int jc = cLo + lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); /* the previous window's reads are done before it is overwritten (LDS only: the walk's byte stores are not waited for) */
        __builtin_amdgcn_wave_barrier();
        diag0 = ii - jj;
        banded = !wantFull;
        const int gFirst = banded ? band_first(ii, R0) : 0;
        uint32_t rq, rr;
        auto both = [&](auto kindC, auto cntC, auto &raw) {
            issue(kindC, cntC, raw, rq, rr, gBase, gFirst, jc, R0);
            commit(kindC, cntC, raw, rq, rr, gBase, gFirst, jc, R0);
        };
        if (banded) {
            uint32_t raw[kRawBand];
            if constexpr (BAND) both(integral_constant<int, 3>{}, integral_constant<int, GL>{}, raw);
            else if (kind == 0) both(integral_constant<int, 0>{}, integral_constant<int, GL>{}, raw);
            else if (kind == 1) both(integral_constant<int, 1>{}, integral_constant<int, GL>{}, raw);
            else both(integral_constant<int, 2>{}, integral_constant<int, GL>{}, raw);
        } else {
            uint32_t raw[kRawFull];
            if constexpr (BAND) both(integral_constant<int, 3>{}, integral_constant<int, G>{}, raw);
            else if (kind == 0) both(integral_constant<int, 0>{}, integral_constant<int, G>{}, raw);
            else if (kind == 1) both(integral_constant<int, 1>{}, integral_constant<int, G>{}, raw);
            else both(integral_constant<int, 2>{}, integral_constant<int, G>{}, raw);
        }
        /* the two characters are needed HERE: without this the compiler waits for them (s_waitcnt vmcnt(0)) where the walk first uses them --
         * in every trip of the loop, where the wait also covers the byte stores of the previous trips (3 us per trip, measured) */
        asm volatile("" : "+v"(chQ), "+v"(chR));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    /* rows i-1, i and columns j-1, j must lie inside the window (borders are cells of it) */
    auto need_window = [&]() -> bool {
        if (i - 1 <= R0 || i > R0 + WR || j - 1 < cLo || j > cLo + 63) return true;
        if (banded) { const int dev = (i - j) - diag0; if (dev < -7 || dev > 6) { wantFull = true; return true; } } /* outside the fetched band */
        return false;
    };
    auto cell = [&](const int pl, const int col, const int row) -> int { return (int)win[(pl * 64 + col) * CS + row]; };
    /* number of lanes that continue a run which starts at lane `from` and goes DOWN the lanes while `on` holds (lane 0 is never on) */
    auto run_down = [&](const bool on, const int from) -> int {
        const unsigned long long inv = ~__builtin_amdgcn_ballot_w64(on) << (63 - from);
        return inv ? __builtin_clzll(inv) : 64;
    };
    /* Decision of window cell (rq, cq) (>= 1 each: callers clamp), rc = the reference character of its column: 0 diagonal, 1 up, 2 left,
     * 3 stop / other (ANW: the SCORING state's 1 = to INSERTION, 2 = to DELETION).  qcOut = the query character of its row. */
    auto decide_cell = [&](const int rq, const int cq, const uint32_t rc, int &qcOut) -> uint32_t {
        const int qc = __builtin_amdgcn_ds_bpermute(rq << 2, (int)chQ);
        qcOut = qc;
        const int ii = R0 + 1 + rq, jc = cLo + cq;
        uint32_t d;
        if constexpr (PLANES == 1) {
            const int h = cell(0, cq, rq), up = cell(0, cq, rq - 1), left = cell(0, cq - 1, rq);
            if (algo == DPX_K_LSW) {
                d = h <= 0 ? 3u : (up + g == h ? 1u : (left + g == h ? 2u : 0u)); /* UPPER, LEFT, CORNER; stop at 0 (c++/backtrack.cpp:21-97) */
            } else {
                const int dg = cell(0, cq - 1, rq - 1);
                const int mm = dg + ((uint32_t)qc == rc ? match : mismatch);
                const int del = up + g, ins = left + g;
                d = ins >= max(del, mm) ? 2u : (del >= mm ? 1u : 0u); /* INSERTION over DELETION over the diagonal */
                if (ii == 0) d = 2u;       /* row 0: QUERY_INSERTION to the corner */
                else if (jc == 0) d = 1u;  /* column 0: QUERY_DELETION */
            }
        } else {
            const int dg = cell(0, cq - 1, rq - 1), I = cell(1, cq, rq), D = cell(2, cq, rq);
            const int mm = dg + ((uint32_t)qc == rc ? match : mismatch);
            d = I >= max(D, mm) ? 1u : (D >= mm ? 2u : 0u);
            if (ii <= 0 || jc <= 0) d = 3u;
        }
        return d;
    };
    /* (r, c) = the walker's window cell; lane l decides the cell of the walker's diagonal in its own column (3 outside the usable part) */
    auto decide_diag = [&](const int r, const int c, int &qcOut) -> uint32_t {
        const int rr = r - (c - lane);
        const bool usable = lane <= c && lane >= 1 && rr >= 1;
        const uint32_t d = decide_cell(usable ? rr : 1, usable ? lane : 1, chR, qcOut); /* (clamped: every lane reads inside the window) */
        return usable ? d : 3u;
    };
    /* LSW / LNW gaps in runs as well (round 4: the end gaps of a global alignment of short reads are 30 steps along one row): the cells to the
     * left of the walker on its row (lane l: column l) or above it in its column (lane k: k rows up) that decide like it.  In a banded
     * window a run ends where it leaves the fetched band (`dev` = the walker's distance from the anchor's diagonal). */
    auto left_run = [&](const int r, const int c, const int dev) -> int {
        int qc;
        const bool usable = lane <= c && lane >= 1 && (!banded || dev + (c - lane) <= 6);
        const uint32_t d = decide_cell(r, usable ? lane : 1, chR, qc);
        return run_down(usable && d == 2u, c);
    };
    auto up_run = [&](const int r, const int c, const int dev) -> int {
        int qc;
        const bool usable = r - lane >= 1 && (!banded || dev - lane >= -7);
        const uint32_t d = decide_cell(usable ? r - lane : 1, c, (uint32_t)__builtin_amdgcn_readlane((int)chR, c), qc);
        const unsigned long long ends = __builtin_amdgcn_ballot_w64(!(usable && d == 1u));
        return ends ? __builtin_ctzll(ends) : 64;
    };
    /* the lanes c, c-1, ... c-len+1 store the characters of a diagonal run (their own column's reference character, their row's query character) */
    auto emit_diag = [&](const int c, const int len, const int qc) {
        const int k = c - lane;
        if (k >= 0 && k < len) {
            const int at = pos - 1 - k;
            lr[at] = (char)chR; lx[at] = ((uint32_t)qc == chR) ? '*' : '|'; lq[at] = (char)qc;
        }
        pos -= len;
    };
    /* `len` steps to the left from column c: the lanes store their reference characters against gaps */
    auto emit_left = [&](const int c, const int len) {
        const int k = c - lane;
        if (k >= 0 && k < len) { const int at = pos - 1 - k; lr[at] = (char)chR; lx[at] = ' '; lq[at] = '_'; }
        pos -= len;
    };
    /* `len` steps up from row r: lane k stores the query character of row r - k */
    auto emit_up = [&](const int r, const int len) {
        const int qc = __builtin_amdgcn_ds_bpermute(max(r - lane, 0) << 2, (int)chQ);
        if (lane < len) { const int at = pos - 1 - lane; lr[at] = '_'; lx[at] = ' '; lq[at] = (char)qc; }
        pos -= len;
    };
    int cur = 0; /* ANW: 0 SCORING, 1 INSERTION, 2 DELETION */
    for (;;) {
        if (algo == DPX_K_LSW ? !(i > 0 && j > 0) : !(i != 0 || j != 0)) break;
        if (need_window()) { load_window(max(i, 1), max(j, 1)); wantFull = false; } /* (a window anchored on row 1 / column 1 also serves row 0 / column 0) */
        const int r = i - R0 - 1, c = j - cLo;
        if constexpr (PLANES == 1) {
            if (algo != DPX_K_LSW && (i == 0 || j == 0)) { /* LNW on a border: to the corner in runs (row 0: QUERY_INSERTION, column 0: QUERY_DELETION) */
                if (i == 0) { const int len = min(j, c); emit_left(c, len); j -= len; }
                else { const int len = min(i, r); emit_up(r, len); i -= len; }
                continue;
            }
            int qc;
            const uint32_t d = decide_diag(r, c, qc);
            const int run = run_down(d == 0u, c);
            if (run) {
                /* ... and the step that ends the run with it: the lane below the run has decided that cell already (paths of related
                 * sequences are diagonal runs separated by single gap steps: half the trips) */
                emit_diag(c, run, qc); i -= run; j -= run;
                const int cx = c - run, rx = r - run;
                if (cx >= 1 && rx >= 1 && i > 0 && j > 0) {
                    const uint32_t dx = (uint32_t)__builtin_amdgcn_readlane((int)d, cx);
                    if (dx == 1u) { emit_up(rx, 1); i--; }
                    else if (dx == 2u) { emit_left(cx, 1); j--; }
                    else if (algo == DPX_K_LSW) break; /* H = 0: the local alignment starts here */
                }
                continue;
            }
            const uint32_t dc = (uint32_t)__builtin_amdgcn_readlane((int)d, c);
            if (algo == DPX_K_LSW && dc == 3u) break;
            const int dev = (i - j) - diag0;
            if (dc == 1u) { const int len = max(up_run(r, c, dev), 1); emit_up(r, len); i -= len; }
            else { const int len = max(left_run(r, c, dev), 1); emit_left(c, len); j -= len; }
        } else {
            if (i == 0 || j == 0) { /* along a border to the corner: first up column 0, then left along row 0 (c++/backtrack.cpp:214-356) */
                if (i > 0) { const int len = min(i, r); emit_up(r, len); i -= len; }   /* (rows r, r-1, ... 1 of the window hold query rows) */
                else { const int len = min(j, c); emit_left(c, len); j -= len; }
                continue;
            }
            if (cur == 0) {
                int qc;
                const uint32_t d = decide_diag(r, c, qc);
                const int run = run_down(d == 0u, c);
                if (run) {
                    emit_diag(c, run, qc); i -= run; j -= run;
                    const int cx = c - run, rx = r - run; /* the cell that ends the run has been decided with it */
                    if (cx >= 1 && rx >= 1 && i > 0 && j > 0) { const int dx = __builtin_amdgcn_readlane((int)d, cx); if (dx == 1 || dx == 2) cur = dx; }
                    continue;
                }
                cur = __builtin_amdgcn_readlane((int)d, c); /* 1: to INSERTION, 2: to DELETION */
            } else if (cur == 1) {
                /* INSERTION: steps to the left along row r until (and including) the cell where the gap was opened; lane l decides the cell in column l */
                const int cq = max(lane, 1), jc = cLo + cq;
                const bool opened = jc == 1 || cell(0, cq - 1, r) + g + ext >= cell(1, cq - 1, r) + ext;
                const bool usable = lane <= c && lane >= 1 && jc >= 1 && (!banded || (i - j) - diag0 + (c - lane) <= 6); /* (column l-1, row r inside the band) */
                const int cont = run_down(usable && !opened, c);          /* cells the gap passes through */
                const bool stops = c - cont >= 1 && cLo + c - cont >= 1 && (!banded || (i - j) - diag0 + cont <= 6); /* ... and then a usable cell that opened it (else: the window's / band's edge) */
                const int len = cont + (stops ? 1 : 0);
                emit_left(c, len); j -= len;
                if (stops) cur = 0;
            } else {
                /* DELETION: steps up along column c; lane k decides the cell k rows above the walker */
                const int rq = max(r - lane, 1), ii = R0 + 1 + rq;
                const bool opened = ii == 1 || cell(0, c, rq - 1) + g + ext >= cell(2, c, rq - 1) + ext;
                const bool usable = r - lane >= 1 && ii >= 1 && (!banded || (i - j) - diag0 - lane >= -7); /* (column c, row r-k-1 inside the band) */
                const unsigned long long m64 = __builtin_amdgcn_ballot_w64(!(usable && !opened)); /* first lane that ends the run: opened, or not usable */
                const int cont = m64 ? __builtin_ctzll(m64) : 64;
                const bool stops = r - cont >= 1 && R0 + 1 + r - cont >= 1 && (!banded || (i - j) - diag0 - cont >= -7);
                const int len = cont + (stops ? 1 : 0);
                emit_up(r, len); i -= len;
                if (stops) cur = 0;
            }
        }
    }
    if (lane == 0) tbLen[p] = cap - pos;
}

/* =====================================================================================================
 * Output path: packed, variable-length result text built on the device -- the reference's V15 idea
 * (cuda/LNW/LinearNeedlemanWunschV15.cu:168-172,372-425: per-pair string lengths, prefix offsets, one packed buffer, one
 * D2H of the real bytes) taken one step further: the device writes each pair's block exactly as c++/main.cpp prints it,
 *     "<pair number> | <score>\n<reference line>\n<relation line>\n<query line>\n"
 * (three empty lines for a zero-score local alignment, c++/LinearSmithWaterman.cpp:253-257), so the host's share of the
 * output is one fwrite per batch.  Three small kernels after k_traceback: block lengths, an exclusive scan, the copy.
 * ===================================================================================================== */
__device__ __forceinline__ int dec_digits(unsigned long long v) {
    int d = 1;
    while (v >= 10ull) { v /= 10ull; d++; }
    return d;
}
__device__ __forceinline__ unsigned long long block_len(unsigned long long number, int score, int len) {
    const unsigned us = score < 0 ? 0u - (unsigned)score : (unsigned)score;
    return (unsigned long long)(dec_digits(number) + 3 + (score < 0 ? 1 : 0) + dec_digits(us) + 1) + 3ull * (unsigned long long)(len + 1);
}

constexpr int kScanThreads = 256, kScanPerThread = 8, kScanTile = kScanThreads * kScanPerThread;

/* inclusive scan of one value per thread across the workgroup; returns the exclusive prefix, *total = sum over the group */
__device__ __forceinline__ unsigned long long group_exclusive(unsigned long long v, unsigned long long *lds, unsigned long long *total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(x, off, 64);
        if (lane >= off) x += o;
    }
    if (lane == 63) lds[wv] = x;
    __syncthreads();
    unsigned long long base = 0, all = 0;
    for (int k = 0; k < kScanThreads / 64; k++) { const unsigned long long w = lds[k]; if (k < wv) base += w; all += w; }
    __syncthreads();
    *total = all;
    return base + x - v;
}

/* phase 1: per-tile totals of the block lengths; phase 3 (FINAL): per-pair offsets = tile base + exclusive prefix */
template <bool FINAL>
__global__ void __launch_bounds__(kScanThreads) k_out_scan(const int32_t *score, const int32_t *tbLen, int numPairs, unsigned long long firstNumber,
                                                           unsigned long long *tileSums, unsigned long long *outOff) {
    __shared__ unsigned long long lds[kScanThreads / 64];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanPerThread;
    unsigned long long v[kScanPerThread], mine = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) {
        const size_t p = base + k;
        v[k] = p < (size_t)numPairs ? block_len(firstNumber + p, score[p], tbLen[p]) : 0ull;
        mine += v[k];
    }
    unsigned long long total;
    unsigned long long ex = group_exclusive(mine, lds, &total);
    if constexpr (!FINAL) {
        if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
    } else {
        ex += tileSums[blockIdx.x]; /* exclusive prefix of the tiles, from k_out_scan_tiles */
#pragma unroll
        for (int k = 0; k < kScanPerThread; k++) {
            const size_t p = base + k;
            if (p < (size_t)numPairs) outOff[p] = ex;
            ex += v[k];
        }
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanThreads - 1) outOff[numPairs] = ex; /* total bytes */
    }
}

/* phase 2: exclusive scan of the tile totals, in place (one workgroup; numTiles is small: pairs / 2048) */
__global__ void __launch_bounds__(kScanThreads) k_out_scan_tiles(unsigned long long *tileSums, int numTiles) {
    __shared__ unsigned long long lds[kScanThreads / 64];
    unsigned long long carry = 0;
    for (int t0 = 0; t0 < numTiles; t0 += kScanThreads) {
        const int t = t0 + (int)threadIdx.x;
        const unsigned long long v = t < numTiles ? tileSums[t] : 0ull;
        unsigned long long total;
        const unsigned long long ex = group_exclusive(v, lds, &total);
        if (t < numTiles) tileSums[t] = carry + ex;
        carry += total;
    }
}

/* one pair's block, written by one wave: header by lane 0, the three right-aligned lines of k_traceback copied 64 bytes per instruction */
__device__ __forceinline__ void out_write_block(const dpx_pair_dev *pairs, const int32_t *score, const int32_t *tbLen, const uint64_t *tbOff,
                                                const char *tb, const int p, const int lane, const unsigned long long firstNumber,
                                                const unsigned long long dstOff, char *out) {
    const int len = tbLen[p], sc = score[p];
    const int cap = (pairs[p].m + pairs[p].n + 1 + 3) & ~3; /* line capacity, as in k_traceback */
    char *dst = out + dstOff;
    const unsigned long long number = firstNumber + (unsigned long long)p;
    const unsigned us = sc < 0 ? 0u - (unsigned)sc : (unsigned)sc;
    const int dn = dec_digits(number), ds = dec_digits(us), neg = sc < 0 ? 1 : 0;
    const int hdr = dn + 3 + neg + ds + 1;
    if (lane == 0) {
        unsigned long long v = number;
        for (int k = dn - 1; k >= 0; k--) { dst[k] = (char)('0' + (int)(v % 10ull)); v /= 10ull; }
        dst[dn] = ' '; dst[dn + 1] = '|'; dst[dn + 2] = ' ';
        if (neg) dst[dn + 3] = '-';
        unsigned u = us;
        for (int k = ds - 1; k >= 0; k--) { dst[dn + 3 + neg + k] = (char)('0' + (int)(u % 10u)); u /= 10u; }
        dst[hdr - 1] = '\n';
    }
    const char *src = tb + tbOff[p] + (cap - len);
#pragma unroll
    for (int line = 0; line < 3; line++) {
        const char *s = src + (size_t)line * cap;
        char *d = dst + hdr + (size_t)line * (len + 1);
        for (int x = lane; x < len; x += 64) d[x] = s[x];
        if (lane == 0) d[len] = '\n';
    }
}

__global__ void __launch_bounds__(256) k_out_compact(const dpx_pair_dev *pairs, const int32_t *score, const int32_t *tbLen, const uint64_t *tbOff,
                                                      const char *tb, int numPairs, unsigned long long firstNumber,
                                                      const unsigned long long *outOff, char *out) {
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (p >= numPairs) return;
    out_write_block(pairs, score, tbLen, tbOff, tb, p, lane, firstNumber, outOff[p], out);
}

/* Small batches (up to kOutSmallPairs pairs: the class-per-pair drivers send 20 per device round trip): lengths, scan and copy in ONE
 * workgroup -- one launch instead of four in a chain of dependent kernels whose launches are most of the round trip. */
constexpr int kOutSmallPairs = 256;
static_assert(kOutSmallPairs <= kScanThreads, "one pair per thread");
__global__ void __launch_bounds__(kScanThreads) k_out_small(const dpx_pair_dev *pairs, const int32_t *score, const int32_t *tbLen, const uint64_t *tbOff,
                                                            const char *tb, int numPairs, unsigned long long firstNumber, unsigned long long *outOff,
                                                            char *out) {
    __shared__ unsigned long long lds[kScanThreads / 64];
    __shared__ unsigned long long offs[kOutSmallPairs];
    const int p = (int)threadIdx.x; /* one pair per thread (kScanThreads >= kOutSmallPairs) */
    const unsigned long long mine = p < numPairs ? block_len(firstNumber + (unsigned long long)p, score[p], tbLen[p]) : 0ull;
    unsigned long long total;
    const unsigned long long ex = group_exclusive(mine, lds, &total);
    if (p < numPairs) { outOff[p] = ex; offs[p] = ex; }
    if (p == 0) outOff[numPairs] = total;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); q < numPairs; q += kScanThreads / 64)
        out_write_block(pairs, score, tbLen, tbOff, tb, q, lane, firstNumber, offs[q], out);
}

/* =====================================================================================================
 * 2-bit packed input (dpx_batch_create_packed2; the input side of c++/parseInput.cpp:78-112): the host sends four bases
 * per byte (base k in bits 2*(k%4) of byte k/4) and a 4-entry alphabet; one thread expands one dword = 16 bases into the 16 bytes
 * of the byte buffer that every fill, traceback and output kernel reads (one aligned 16-byte store).  The fills keep matching
 * plain bytes (the reference's contract: any alphabet), so a packed batch and a byte batch are the same batch after this kernel.
 * ===================================================================================================== */
__global__ void __launch_bounds__(256) k_unpack2(const uint32_t *packed, const uint32_t alphabet, uint4 *out, const size_t numDwords) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= numDwords) return;
    const uint32_t w = packed[i];
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { /* eight bits of w = four bases = one output dword: v_perm_b32 picks alphabet bytes by selector */
        const uint32_t c = (w >> (8 * k)) & 0xFFu;
        const uint32_t sel = (c & 3u) | (((c >> 2) & 3u) << 8) | (((c >> 4) & 3u) << 16) | ((c >> 6) << 24);
        o[k] = __builtin_amdgcn_perm(0u, alphabet, sel);
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

/* =====================================================================================================
 * DPX primitive probe (dpx_prim_eval): runs the CDNA4 mappings of dpx_prims.hpp on the device.
 * ===================================================================================================== */
__global__ void k_prim_eval(const int32_t *op, const uint32_t *a, const uint32_t *b, const uint32_t *c, size_t count,
                            uint32_t *res, uint32_t *pred) {
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= count) return;
    const uint32_t A = a[x], B = b[x], C = c[x];
    const int sa = (int)A, sb = (int)B, sc = (int)C;
    bool p = false, ph = false, pl = false;
    uint32_t r = 0;
    switch (op[x]) {
    case 0: r = (uint32_t)dpx::vimax3_s32(sa, sb, sc); break;
    case 1: r = dpx::vimax3_s16x2(A, B, C); break;
    case 2: r = dpx::vimax3_u32(A, B, C); break;
    case 3: r = dpx::vimax3_u16x2(A, B, C); break;
    case 4: r = (uint32_t)dpx::vimin3_s32(sa, sb, sc); break;
    case 5: r = dpx::vimin3_s16x2(A, B, C); break;
    case 6: r = dpx::vimin3_u32(A, B, C); break;
    case 7: r = dpx::vimin3_u16x2(A, B, C); break;
    case 8: r = (uint32_t)dpx::vimax_s32_relu(sa, sb); break;
    case 9: r = dpx::vimax_s16x2_relu(A, B); break;
    case 10: r = (uint32_t)dpx::vimin_s32_relu(sa, sb); break;
    case 11: r = dpx::vimin_s16x2_relu(A, B); break;
    case 12: r = (uint32_t)dpx::vimax3_s32_relu(sa, sb, sc); break;
    case 13: r = dpx::vimax3_s16x2_relu(A, B, C); break;
    case 14: r = (uint32_t)dpx::vimin3_s32_relu(sa, sb, sc); break;
    case 15: r = dpx::vimin3_s16x2_relu(A, B, C); break;
    case 16: r = (uint32_t)dpx::vibmax_s32(sa, sb, &p); break;
    case 17: r = dpx::vibmax_u32(A, B, &p); break;
    case 18: r = (uint32_t)dpx::vibmin_s32(sa, sb, &p); break;
    case 19: r = dpx::vibmin_u32(A, B, &p); break;
    case 20: r = dpx::vibmax_s16x2(A, B, &ph, &pl); break;
    case 21: r = dpx::vibmax_u16x2(A, B, &ph, &pl); break;
    case 22: r = dpx::vibmin_s16x2(A, B, &ph, &pl); break;
    case 23: r = dpx::vibmin_u16x2(A, B, &ph, &pl); break;
    case 24: r = (uint32_t)dpx::viaddmax_s32(sa, sb, sc); break;
    case 25: r = dpx::viaddmax_u32(A, B, C); break;
    case 26: r = dpx::viaddmax_s16x2(A, B, C); break;
    case 27: r = dpx::viaddmax_u16x2(A, B, C); break;
    case 28: r = (uint32_t)dpx::viaddmin_s32(sa, sb, sc); break;
    case 29: r = dpx::viaddmin_u32(A, B, C); break;
    case 30: r = dpx::viaddmin_s16x2(A, B, C); break;
    case 31: r = dpx::viaddmin_u16x2(A, B, C); break;
    case 32: r = (uint32_t)dpx::viaddmax_s32_relu(sa, sb, sc); break;
    case 33: r = dpx::viaddmax_s16x2_relu(A, B, C); break;
    case 34: r = (uint32_t)dpx::viaddmin_s32_relu(sa, sb, sc); break;
    case 35: r = dpx::viaddmin_s16x2_relu(A, B, C); break;
    default: break;
    }
    res[x] = r;
    const int o = op[x];
    pred[x] = (o >= 20 && o <= 23) ? (uint32_t)((ph ? 2 : 0) | (pl ? 1 : 0)) : (uint32_t)(p ? 1 : 0);
}

template <class K>
hipError_t launch_fill_kernel(K kernel, const dpx_fill_args &a, dim3 grid, size_t lds, hipStream_t s) {
    if (lds > 64u * 1024u) { /* opt in to more than the default 64 KiB of dynamic LDS (160 KiB per CU on gfx950) */
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    /* workgroups of a.wavesPerBlock independent waves (4, or 1 for small launches: see dpx_fill_waves_per_block); `lds` is the request of a
     * four-wave workgroup */
    const unsigned wpb = a.wavesPerBlock;
    hipLaunchKernelGGL(kernel, grid, dim3(64u * wpb), lds / 4u * wpb, s, a);
    return hipGetLastError();
}

template <int R>
hipError_t launch_linear_R(const dpx_fill_args &a, bool local, bool store, dim3 grid, size_t lds, hipStream_t s) {
    if (local) return store ? launch_fill_kernel(k_linear_fill<R, true, true>, a, grid, lds, s)
                            : launch_fill_kernel(k_linear_fill<R, true, false>, a, grid, lds, s);
    return store ? launch_fill_kernel(k_linear_fill<R, false, true>, a, grid, lds, s)
                 : launch_fill_kernel(k_linear_fill<R, false, false>, a, grid, lds, s);
}

template <int R>
hipError_t launch_linear_pk_R(const dpx_fill_args &a, bool local, dim3 grid, size_t lds, hipStream_t s) {
    if (!local) return launch_fill_kernel(k_linear_fill_pk<R, false, false>, a, grid, lds, s);
    return a.rowTags ? launch_fill_kernel(k_linear_fill_pk<R, true, true>, a, grid, lds, s)
                     : launch_fill_kernel(k_linear_fill_pk<R, true, false>, a, grid, lds, s);
}

template <int C>
hipError_t launch_banded_C(const dpx_fill_args &a, bool store, dim3 grid, size_t lds, hipStream_t s) {
    const bool pb = ((a.band + 1) & 1) != 0; /* parity of step A = 0 */
    if (pb) return store ? launch_fill_kernel(k_banded_fill<C, true, true>, a, grid, lds, s)
                         : launch_fill_kernel(k_banded_fill<C, true, false>, a, grid, lds, s);
    return store ? launch_fill_kernel(k_banded_fill<C, false, true>, a, grid, lds, s)
                 : launch_fill_kernel(k_banded_fill<C, false, false>, a, grid, lds, s);
}

template <int R>
hipError_t launch_affine_R(const dpx_fill_args &a, bool store, dim3 grid, size_t lds, hipStream_t s) {
    return store ? launch_fill_kernel(k_affine_fill<R, true>, a, grid, lds, s)
                 : launch_fill_kernel(k_affine_fill<R, false>, a, grid, lds, s);
}

} // namespace

/* ---- host-callable launchers (used by dpx_capi.cpp) ---- */

hipError_t dpx_launch_fill(const dpx_fill_args &a, int algo, int R, bool store, size_t ldsBytes, hipStream_t stream) {
    if (a.numPairs <= 0) return hipSuccess;
    const int wavesPerBlock = (int)a.wavesPerBlock;
    dim3 grid((unsigned)((a.numPairs + wavesPerBlock - 1) / wavesPerBlock));
    if (algo == DPX_K_LNW || algo == DPX_K_LSW) {
        const bool local = algo == DPX_K_LSW;
        switch (R) {
        case 2: return launch_linear_R<2>(a, local, store, grid, ldsBytes, stream);
        case 4: return launch_linear_R<4>(a, local, store, grid, ldsBytes, stream);
        case 8: return launch_linear_R<8>(a, local, store, grid, ldsBytes, stream);
        case 16: return launch_linear_R<16>(a, local, store, grid, ldsBytes, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (algo == DPX_K_BSW) { /* here R carries the cells-per-lane count C = dpx_band_cpl(band) */
        switch (R) {
        case 1: return launch_banded_C<1>(a, store, grid, ldsBytes, stream);
        case 2: return launch_banded_C<2>(a, store, grid, ldsBytes, stream);
        case 4: return launch_banded_C<4>(a, store, grid, ldsBytes, stream);
        case 8: return launch_banded_C<8>(a, store, grid, ldsBytes, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (algo == DPX_K_ANW) {
        switch (R) {
        case 2: return launch_affine_R<2>(a, store, grid, ldsBytes, stream);
        case 4: return launch_affine_R<4>(a, store, grid, ldsBytes, stream);
        case 8: return launch_affine_R<8>(a, store, grid, ldsBytes, stream);
        default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;
}

/* lane-packed kernels (short / medium queries): a.waves = one descriptor per wave, a.numPairs = number of waves.
 * ldsBytes is per workgroup: (dpx_lanes_stage_bytes() + reference area) x waves per workgroup (4 for the linear kernels,
 * 1 for the affine one). */
size_t dpx_lanes_stage_bytes(int algo, int R, bool store) {
    if (!store) return (size_t)kLaneScratch;
    return (size_t)(algo == DPX_K_ANW ? 3 : 1) * (size_t)(R / 8) * 64u * (size_t)kStageLine;
}
int dpx_lanes_waves_per_block(int algo) { return algo == DPX_K_ANW ? DPX_ALANES_THREADS / 64 : DPX_FILL_THREADS / 64; } /* (the most: small launches of the linear kernels use 1) */

template <class K>
static hipError_t launch_lanes_kernel(K kernel, const dpx_fill_args &a, dim3 grid, int threads, size_t lds, hipStream_t s) {
    if (lds > 64u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, a);
    return hipGetLastError();
}

hipError_t dpx_launch_fill_lanes(const dpx_fill_args &a, int algo, int R, bool store, size_t ldsBytes, hipStream_t stream) {
    if (a.numPairs <= 0) return hipSuccess;
    const int wpb = algo == DPX_K_ANW ? DPX_ALANES_THREADS / 64 : (int)a.wavesPerBlock; /* (ldsBytes = per wave x this) */
    dim3 grid((unsigned)((a.numPairs + wpb - 1) / wpb));
    if (algo == DPX_K_ANW) {
        const int th = DPX_ALANES_THREADS;
        if (R == 8) return store ? launch_lanes_kernel(k_affine_lanes<8, true>, a, grid, th, ldsBytes, stream)
                                 : launch_lanes_kernel(k_affine_lanes<8, false>, a, grid, th, ldsBytes, stream);
        return hipErrorInvalidValue;
    }
    const bool local = algo == DPX_K_LSW;
    const int th = 64 * wpb;
#define DPX_LANES_CASE(R_)                                                                                              \
    case R_:                                                                                                           \
        if (local) return store ? launch_lanes_kernel(k_linear_lanes<R_, true, true>, a, grid, th, ldsBytes, stream)    \
                                : launch_lanes_kernel(k_linear_lanes<R_, true, false>, a, grid, th, ldsBytes, stream);  \
        return store ? launch_lanes_kernel(k_linear_lanes<R_, false, true>, a, grid, th, ldsBytes, stream)              \
                     : launch_lanes_kernel(k_linear_lanes<R_, false, false>, a, grid, th, ldsBytes, stream);
    switch (R) {
        DPX_LANES_CASE(8)
        DPX_LANES_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef DPX_LANES_CASE
}

/* packed lane kernel (16 rows per lane, two row blocks of one pair in the two halves): a.waves / a.numPairs as dpx_launch_fill_lanes */
hipError_t dpx_launch_fill_lanes_packed(const dpx_fill_args &a, int algo, size_t ldsBytes, hipStream_t stream) {
    if (a.numPairs <= 0) return hipSuccess;
    const int wpb = (int)a.wavesPerBlock;
    dim3 grid((unsigned)((a.numPairs + wpb - 1) / wpb));
    return algo == DPX_K_LSW ? launch_lanes_kernel(k_linear_lanes_pk<true>, a, grid, 64 * wpb, ldsBytes, stream)
                             : launch_lanes_kernel(k_linear_lanes_pk<false>, a, grid, 64 * wpb, ldsBytes, stream);
}

/* split kernel (small batches): one workgroup of `waves` waves per pair, a.numPairs workgroups */
hipError_t dpx_launch_fill_split(const dpx_fill_args &a, int algo, int R, int waves, size_t ldsBytes, hipStream_t stream) {
    if (a.numPairs <= 0) return hipSuccess;
    if (waves < 1 || waves > DPX_SPLIT_MAX_WAVES) return hipErrorInvalidValue;
    const bool local = algo == DPX_K_LSW;
    dim3 grid((unsigned)a.numPairs);
#define DPX_SPLIT_CASE(R_)                                                                                                   \
    case R_: return local ? launch_lanes_kernel(k_linear_split<R_, true>, a, grid, 64 * waves, ldsBytes, stream)             \
                          : launch_lanes_kernel(k_linear_split<R_, false>, a, grid, 64 * waves, ldsBytes, stream);
    switch (R) {
        DPX_SPLIT_CASE(2)
        DPX_SPLIT_CASE(4)
    default: return hipErrorInvalidValue; /* (8 rows per lane would spill under the 1024-thread launch bound) */
    }
#undef DPX_SPLIT_CASE
}

/* packed two-pairs-per-wave linear fill: a.order = couples (2 ints each), a.numPairs = number of couples */
hipError_t dpx_launch_fill_packed(const dpx_fill_args &a, int algo, int R, size_t ldsBytes, hipStream_t stream) {
    if (a.numPairs <= 0) return hipSuccess;
    const int wavesPerBlock = (int)a.wavesPerBlock;
    dim3 grid((unsigned)((a.numPairs + wavesPerBlock - 1) / wavesPerBlock));
    const bool local = algo == DPX_K_LSW;
    switch (R) {
    case 2: return launch_linear_pk_R<2>(a, local, grid, ldsBytes, stream);
    case 4: return launch_linear_pk_R<4>(a, local, grid, ldsBytes, stream);
    case 8: return launch_linear_pk_R<8>(a, local, grid, ldsBytes, stream);
    case 16: return launch_linear_pk_R<16>(a, local, grid, ldsBytes, stream);
    default: return hipErrorInvalidValue;
    }
}

/* packed banded fill: a.order = couples (2 ints each), a.numPairs = number of couples; C = cells per lane */
template <int C>
static hipError_t launch_banded_pk_C(const dpx_fill_args &a, dim3 grid, size_t lds, hipStream_t s) {
    const bool pb = ((a.band + 1) & 1) != 0; /* parity of step A = 0 */
    return pb ? launch_fill_kernel(k_banded_fill_pk<C, true>, a, grid, lds, s) : launch_fill_kernel(k_banded_fill_pk<C, false>, a, grid, lds, s);
}
hipError_t dpx_launch_banded_packed(const dpx_fill_args &a, int C, size_t ldsBytes, hipStream_t stream) {
    if (a.numPairs <= 0) return hipSuccess;
    const int wavesPerBlock = (int)a.wavesPerBlock;
    dim3 grid((unsigned)((a.numPairs + wavesPerBlock - 1) / wavesPerBlock));
    switch (C) {
    case 1: return launch_banded_pk_C<1>(a, grid, ldsBytes, stream);
    case 2: return launch_banded_pk_C<2>(a, grid, ldsBytes, stream);
    case 4: return launch_banded_pk_C<4>(a, grid, ldsBytes, stream);
    case 8: return launch_banded_pk_C<8>(a, grid, ldsBytes, stream);
    default: return hipErrorInvalidValue;
    }
}


hipError_t dpx_launch_export(const int16_t *mat, const dpx_pair_dev &pr, int algo, int R, int planes, int plane, int gapOpen,
                             int gapExtend, int band, int16_t *out, hipStream_t stream) {
    const size_t total = (size_t)(pr.m + 1) * (size_t)(pr.n + 1);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(k_export_matrix, dim3(blocks), dim3(256), 0, stream, mat, pr, algo, R, planes, plane, gapOpen,
                       gapExtend, band, out);
    return hipGetLastError();
}

hipError_t dpx_launch_traceback(const dpx_fill_args &a, int numPairs, int algo, int R, int planes, int walk,
                                const uint64_t *tbOff, char *tb, int32_t *tbLen, hipStream_t stream) {
    if (numPairs <= 0) return hipSuccess;
    const bool cachedWalk = walk == 1;
    if (walk == 2) { /* one wave per pair with an LDS window (k_traceback_wave) */
        const size_t lds = dpx_traceback_wave_lds(algo == DPX_K_ANW ? 3 : 1);
        if (algo == DPX_K_ANW)
            hipLaunchKernelGGL((k_traceback_wave<3, false, DPX_K_ANW>), dim3((unsigned)numPairs), dim3(64), lds, stream, a, numPairs, R, a.endRow, a.endCol, tbOff, tb, tbLen);
        else if (algo == DPX_K_BSW)
            hipLaunchKernelGGL((k_traceback_wave<1, true, DPX_K_LSW>), dim3((unsigned)numPairs), dim3(64), lds, stream, a, numPairs, R, a.endRow, a.endCol, tbOff, tb, tbLen);
        else if (algo == DPX_K_LNW)
            hipLaunchKernelGGL((k_traceback_wave<1, false, DPX_K_LNW>), dim3((unsigned)numPairs), dim3(64), lds, stream, a, numPairs, R, a.endRow, a.endCol, tbOff, tb, tbLen);
        else
            hipLaunchKernelGGL((k_traceback_wave<1, false, DPX_K_LSW>), dim3((unsigned)numPairs), dim3(64), lds, stream, a, numPairs, R, a.endRow, a.endCol, tbOff, tb, tbLen);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_traceback, dim3((unsigned)((numPairs + 63) / 64)), dim3(64), 0, stream, a, numPairs, algo, R, planes,
                       cachedWalk ? 1 : 0, a.endRow, a.endCol, tbOff, tb, tbLen);
    return hipGetLastError();
}

/* result text of a batch: lengths -> exclusive scan (tileSums: ceil(numPairs / 2048) + 1 entries of scratch) -> packed copy.
 * outOff[numPairs] receives the total number of bytes. */
size_t dpx_out_scan_tiles(size_t numPairs) { return (numPairs + kScanTile - 1) / kScanTile; }
hipError_t dpx_launch_output(const dpx_pair_dev *pairs, const int32_t *score, const int32_t *tbLen, const uint64_t *tbOff, const char *tb,
                             int numPairs, unsigned long long firstNumber, unsigned long long *tileSums, unsigned long long *outOff, char *out,
                             bool scanOnly, bool compactOnly, hipStream_t stream) {
    if (numPairs <= 0) return hipSuccess;
    const int tiles = (int)dpx_out_scan_tiles((size_t)numPairs);
    if (!compactOnly && !scanOnly && numPairs <= kOutSmallPairs) {
        hipLaunchKernelGGL(k_out_small, dim3(1), dim3(kScanThreads), 0, stream, pairs, score, tbLen, tbOff, tb, numPairs, firstNumber, outOff, out);
        return hipGetLastError();
    }
    if (!compactOnly) {
        hipLaunchKernelGGL(k_out_scan<false>, dim3(tiles), dim3(kScanThreads), 0, stream, score, tbLen, numPairs, firstNumber, tileSums, outOff);
        hipLaunchKernelGGL(k_out_scan_tiles, dim3(1), dim3(kScanThreads), 0, stream, tileSums, tiles);
        hipLaunchKernelGGL(k_out_scan<true>, dim3(tiles), dim3(kScanThreads), 0, stream, score, tbLen, numPairs, firstNumber, tileSums, outOff);
    }
    if (!scanOnly)
        hipLaunchKernelGGL(k_out_compact, dim3((unsigned)((numPairs + 3) / 4)), dim3(256), 0, stream, pairs, score, tbLen, tbOff, tb, numPairs, firstNumber,
                           outOff, out);
    return hipGetLastError();
}

/* expand numDwords x 16 bases (2 bits each) into bytes: out must be 16-byte aligned and hold 16 * numDwords bytes */
hipError_t dpx_launch_unpack2(const uint32_t *packed, uint32_t alphabet, char *out, size_t numDwords, hipStream_t stream) {
    if (numDwords == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpack2, dim3((unsigned)((numDwords + 255) / 256)), dim3(256), 0, stream, packed, alphabet, reinterpret_cast<uint4 *>(out), numDwords);
    return hipGetLastError();
}

hipError_t dpx_launch_prim_eval(const int32_t *op, const uint32_t *a, const uint32_t *b, const uint32_t *c, size_t count,
                                uint32_t *res, uint32_t *pred, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prim_eval, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, op, a, b, c, count, res,
                       pred);
    return hipGetLastError();
}
