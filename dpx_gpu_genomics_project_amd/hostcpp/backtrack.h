// backtrack.h -- host back-trackers over flat row-major direction matrices, the output-side plumbing the
// reference's CUDA mains call after copying their matrices back (c++/backtrack.h:14-54; used at
// cuda/LinearSmithWaterman.cu:316, cuda/AffineNeedlemanWunsch.cu:391, cuda/LNW/LinearNeedlemanWunschV7.cu:209).
// Enum values are the reference's.  Every routine prints the three result lines (reference with '_' gaps,
// relation '*' match / '|' mismatch / ' ' gap, query).
//
// The MI355X engine stores int16 scores, not directions; dpxDirectionsFromScores() rebuilds the reference's
// direction matrices from exported score matrices with the same tie rules, so callers of these routines keep
// working.  (The engine's own fast path is the device traceback, dpx_batch_traceback.)
#pragma once
#include <iostream>
#include <string>
#include "printLock.h"

enum directionMain { NONE_MAIN, MATCH, MISMATCH, QUERY_INSERTION, QUERY_DELETION };
enum directionIndel { NONE_INDEL, GAP_OPEN, GAP_EXTEND };
enum currentMatrixPosition { SCORING, INSERTION, DELETION };

void printMatrix(const int *memo, const int referenceLength, const int queryLength);
void printBacktrackMatrix(const directionMain *memo, const int referenceLength, const int queryLength);

void backtrackNW(const directionMain *backtrackMemo, const char *referenceString, const int referenceLength,
                 const char *queryString, const int queryLength);
void backtrackMultiNW(const directionMain *backtrackMemo, const char *referenceString, const int referenceLength,
                      const char *queryString, const int queryLength, const int pairNum, const int score);
void backtrackSW(int currentMemoRow, int currentMemoCol, const int numCols, const directionMain *backtrackMemo,
                 const char *referenceString, const char *queryString);
void backtrackANW(const directionMain *scoringBacktrack, const directionIndel *queryInsertionBacktrack,
                  const directionIndel *queryDeletionBacktrack, const char *referenceString, const int referenceLength,
                  const char *queryString, const int queryLength);

// Rebuild direction matrices ((queryLength+1) x (referenceLength+1), row-major) from int16 score matrices as
// dpx_batch_matrix() exports them.  algo: 0 LNW, 1 LSW, 2 ANW (dpx_algo).  I/D and dirI/dirD only for ANW.
void dpxDirectionsFromScores(int algo, const short *H, const short *I, const short *D, const char *referenceString,
                             const int referenceLength, const char *queryString, const int queryLength, int match,
                             int mismatch, int gapOpen, int gapExtend, directionMain *dirH, directionIndel *dirI,
                             directionIndel *dirD);
