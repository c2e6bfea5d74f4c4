#include "LinearNeedlemanWunsch.h"

#include <cstdio>

void LinearNeedlemanWunsch::init_matrix() { gpu = DpxPairResult(); }

void LinearNeedlemanWunsch::print_matrix() {
    if (gpu.H.empty()) dpxAlignPair(0, reference_str, query_str, match_weight, mismatch_weight, gap_weight, 0, 0, true, gpu);
    dpxPrintScoreMatrix(reference_str, query_str, gpu.H);
}

void LinearNeedlemanWunsch::score_matrix() {
#ifdef PRINT_MATRIX
    const bool wantMatrix = true;
#else
    const bool wantMatrix = false;
#endif
    dpxAlignPair(0 /* DPX_ALGO_LNW */, reference_str, query_str, match_weight, mismatch_weight, gap_weight, 0, 0, wantMatrix, gpu);
}

void LinearNeedlemanWunsch::backtrack() {
#ifdef USE_THREADS
    printLock();
#endif
#ifdef PRINT_MATRIX
    print_matrix();
#endif
    printf("%d | %d\n%s\n%s\n%s\n", pairNum, gpu.score, gpu.refLine.c_str(), gpu.relLine.c_str(), gpu.qryLine.c_str());
#ifdef USE_THREADS
    printUnlock(); // (no flush per block: stdio orders printf and the drivers' cout lines by itself, and 4000 one-block write() calls were 8 % of the run)
#endif
}

void LinearNeedlemanWunsch::align() {
    init_matrix();
#ifdef PRINT_MATRIX
    print_matrix();
#endif
    score_matrix();
    backtrack();
}

void LinearNeedlemanWunsch::print_results() {} // the reference leaves this empty too (LinearNeedlemanWunsch.cpp:233-262)
