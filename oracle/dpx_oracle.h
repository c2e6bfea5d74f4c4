/*
 * dpx_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's pairwise-alignment DP path
 * (mickgordinier/DPX_GPU_Genomics_Project, c++/ CPU classes).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call
 * this.  The shipped product (libdpxalign.so, the hostcpp/ classes) never does.
 *
 * Parity status: PINNED.  Checked against (a) the reference classes themselves
 * compiled from /root/reference into oracle/_ref/ (tests/test_oracle_vs_ref.py,
 * tests/golden/ made by tests/golden/make_golden.py), (b) the known-answer
 * vectors of c++/testFakeDPX.cpp, (c) python/testing.py's LNW example.
 * Banded SW is pinned only by python/LinearBandedSmithWaterman.py (the C++/CUDA
 * banded sources are broken upstream) -- see DESIGN.md.
 *
 * Conventions: m = strlen(query) = rows, n = strlen(reference)
 * = cols; matrices are (m+1) x (n+1) row-major int32; s(i,j) = match if
 * query[i-1]==reference[j-1] else mismatch (plain byte compare).
 */
#ifndef DPX_ORACLE_H
#define DPX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* direction encodings: c++/backtrack.h:14-27 */
enum { ORC_NONE_MAIN = 0, ORC_MATCH = 1, ORC_MISMATCH = 2, ORC_QUERY_INSERTION = 3, ORC_QUERY_DELETION = 4 };
enum { ORC_NONE_INDEL = 0, ORC_GAP_OPEN = 1, ORC_GAP_EXTEND = 2 };

/* ---- DP fills.  H/I/D/dir pointers may be NULL when that output is not wanted. ---- */

/* c++/LinearSmithWaterman.cpp:70-114 (fill) + :145-157 (first strict max, row-major).
 * dir uses backtrack.h values: UPPER_GAP->QUERY_DELETION, LEFT_GAP->QUERY_INSERTION. dir is (m+1)x(n+1). */
void orc_lsw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gap,
                  int32_t *H, uint8_t *dir, int32_t *score, int32_t *endRow, int32_t *endCol);

/* c++/LinearNeedlemanWunsch.cpp:9-42 (borders) + :89-135 (fill). score = H[m][n]. */
void orc_lnw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gap,
                  int32_t *H, uint8_t *dir, int32_t *score);

/* c++/AffineNeedlemanWunsch.cpp:12-54 (borders) + :167-240 (fill).  I = queryInsertionMemo,
 * D = queryDeletionMemo; dirI/dirD are the GAP_OPEN/GAP_EXTEND matrices. */
void orc_anw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gapOpen,
                  int gapExtend, int32_t *H, int32_t *I, int32_t *D, uint8_t *dirH, uint8_t *dirI,
                  uint8_t *dirD, int32_t *score);

/* python/LinearBandedSmithWaterman.py:62-104: LSW restricted to |i-j| <= band-1, out-of-band cells are 0. */
void orc_bsw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gap, int band,
                  int32_t *H, uint8_t *dir, int32_t *score, int32_t *endRow, int32_t *endCol);

/* ---- tracebacks: write three NUL-terminated strings (reference line, relation line, query line).
 * Each out buffer must hold m+n+2 bytes.  Return the alignment length. ---- */

/* c++/LinearSmithWaterman.cpp:116-228 (single start cell, stop when next cell's H == 0) */
int orc_lsw_traceback(const char *ref, int n, const char *qry, int m, const int32_t *H, const uint8_t *dir,
                      int endRow, int endCol, char *refOut, char *relOut, char *qryOut);
/* c++/LinearNeedlemanWunsch.cpp:137-199 */
int orc_lnw_traceback(const char *ref, int n, const char *qry, int m, const uint8_t *dir, char *refOut,
                      char *relOut, char *qryOut);
/* c++/AffineNeedlemanWunsch.cpp:242-377 */
int orc_anw_traceback(const char *ref, int n, const char *qry, int m, const uint8_t *dirH, const uint8_t *dirI,
                      const uint8_t *dirD, char *refOut, char *relOut, char *qryOut);

/* ---- FakeDPX primitives (c++/FakeDPX.cpp), scalar + packed s16x2/u16x2 ----
 * op codes for orc_dpx(): the 36 entry points of c++/FakeDPX.hpp:19-126. */
enum {
    ORC_VIMAX3_S32 = 0, ORC_VIMAX3_S16X2, ORC_VIMAX3_U32, ORC_VIMAX3_U16X2,
    ORC_VIMIN3_S32, ORC_VIMIN3_S16X2, ORC_VIMIN3_U32, ORC_VIMIN3_U16X2,
    ORC_VIMAX_S32_RELU, ORC_VIMAX_S16X2_RELU, ORC_VIMIN_S32_RELU, ORC_VIMIN_S16X2_RELU,
    ORC_VIMAX3_S32_RELU, ORC_VIMAX3_S16X2_RELU, ORC_VIMIN3_S32_RELU, ORC_VIMIN3_S16X2_RELU,
    ORC_VIBMAX_S32, ORC_VIBMAX_U32, ORC_VIBMIN_S32, ORC_VIBMIN_U32,
    ORC_VIBMAX_S16X2, ORC_VIBMAX_U16X2, ORC_VIBMIN_S16X2, ORC_VIBMIN_U16X2,
    ORC_VIADDMAX_S32, ORC_VIADDMAX_U32, ORC_VIADDMAX_S16X2, ORC_VIADDMAX_U16X2,
    ORC_VIADDMIN_S32, ORC_VIADDMIN_U32, ORC_VIADDMIN_S16X2, ORC_VIADDMIN_U16X2,
    ORC_VIADDMAX_S32_RELU, ORC_VIADDMAX_S16X2_RELU, ORC_VIADDMIN_S32_RELU, ORC_VIADDMIN_S16X2_RELU,
    ORC_DPX_NUM_OPS
};
/* Evaluate one primitive.  pred bit0 = pred (or pred_lo), bit1 = pred_hi; 0 for ops without predicates.
 * Packed s16x2 results are the mathematically correct per-halfword values (the reference's
 * __vimax3_s16x2 omits '& 0xFFFF' on a negative low half, c++/FakeDPX.cpp:28 -- a bug, not restated). */
uint32_t orc_dpx(int op, uint32_t a, uint32_t b, uint32_t c, uint32_t *pred);

/* ---- CPU baseline helper for bench.py: fill `numPairs` pairs (same flat layout as the reference's
 * parseInput: sequences buffer + 4-int seqPair records) on `threads` pthreads; returns seconds.
 * algo: 0 LNW, 1 LSW, 2 ANW, 3 BSW.  scores[numPairs] out. ---- */
double orc_fill_batch_timed(int algo, const char *sequences, const int32_t *pairs4, int numPairs, int match,
                            int mismatch, int gapOpen, int gapExtend, int band, int threads, int32_t *scores);

#ifdef __cplusplus
}
#endif
#endif
