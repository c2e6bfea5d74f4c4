#include "LinearSmithWaterman.h"

#include <cstdio>

void LinearSmithWaterman::init_matrix() {
    gpu = DpxPairResult();
    max_score = 0;
}

void LinearSmithWaterman::print_matrix() {
    if (gpu.H.empty()) dpxAlignPair(1, reference_str, query_str, match_weight, mismatch_weight, gap_weight, 0, 0, true, gpu);
    dpxPrintScoreMatrix(reference_str, query_str, gpu.H);
}

void LinearSmithWaterman::score_matrix() {
#ifdef PRINT_MATRIX
    const bool wantMatrix = true;
#else
    const bool wantMatrix = false;
#endif
    dpxAlignPair(1 /* DPX_ALGO_LSW */, reference_str, query_str, match_weight, mismatch_weight, gap_weight, 0, 0, wantMatrix, gpu);
}

void LinearSmithWaterman::backtrack() {
    // the device traceback already produced the lines together with the fill; only the score is latched here,
    // exactly where the reference latches max_score (c++/LinearSmithWaterman.cpp:147-151)
    max_score = gpu.score;
}

void LinearSmithWaterman::align() {
    init_matrix();
#ifdef PRINT_MATRIX
    print_matrix();
#endif
    score_matrix();
    backtrack();
    print_results();
}

void LinearSmithWaterman::print_results() {
#ifdef USE_THREADS
    printLock();
#endif
#ifdef PRINT_MATRIX
    printf("[Scored Matrix]\n");
    print_matrix();
#endif
    if (max_score == 0) {
        printf("%d | 0\n\n\n\n", pairNum); // score 0: three empty lines (c++/LinearSmithWaterman.cpp:253-257)
    } else {
        printf("%d | %d\n%s\n%s\n%s\n", pairNum, max_score, gpu.refLine.c_str(), gpu.relLine.c_str(), gpu.qryLine.c_str());
    }
#ifdef USE_THREADS
    printUnlock(); // (no flush per block: stdio orders printf and the drivers' cout lines by itself, and 4000 one-block write() calls were 8 % of the run)
#endif
}
