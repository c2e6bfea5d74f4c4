#include "AffineNeedlemanWunsch.h"

#include <cstdio>

void AffineNeedlemanWunsch::init_matrix() { gpu = DpxPairResult(); }

void AffineNeedlemanWunsch::print_matrix() {
    if (gpu.H.empty())
        dpxAlignPair(2, reference_str, query_str, matchWeight, mismatchWeight, gapOpenWeight, gapExtendWeight, 0, true, gpu);
    printf("[Scoring Matrix]\n");
    dpxPrintScoreMatrix(reference_str, query_str, gpu.H);
    printf("[Query Insertion Matrix]\n");
    dpxPrintScoreMatrix(reference_str, query_str, gpu.I);
    printf("[Query Deletion Matrix]\n");
    dpxPrintScoreMatrix(reference_str, query_str, gpu.D);
}

void AffineNeedlemanWunsch::score_matrix() {
#ifdef PRINT_MATRIX
    const bool wantMatrix = true;
#else
    const bool wantMatrix = false;
#endif
    dpxAlignPair(2 /* DPX_ALGO_ANW */, reference_str, query_str, matchWeight, mismatchWeight, gapOpenWeight, gapExtendWeight, 0,
                 wantMatrix, gpu);
}

void AffineNeedlemanWunsch::backtrack() {
#ifdef USE_THREADS
    printLock();
#endif
    printf("%d | %d\n%s\n%s\n%s\n", pairNum, gpu.score, gpu.refLine.c_str(), gpu.relLine.c_str(), gpu.qryLine.c_str());
#ifdef USE_THREADS
    printUnlock(); // (no flush per block: stdio orders printf and the drivers' cout lines by itself, and 4000 one-block write() calls were 8 % of the run)
#endif
}

void AffineNeedlemanWunsch::align() {
    init_matrix();
#ifdef PRINT_MATRIX
    print_matrix();
#endif
    score_matrix();
#ifdef PRINT_MATRIX
    print_matrix();
#endif
    backtrack();
}

void AffineNeedlemanWunsch::print_results() {}
