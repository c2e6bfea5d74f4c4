#include "BandedSmithWaterman.h"

#include <cstdio>

void BandedSmithWaterman::init_matrix() {
    gpu = DpxPairResult();
    max_score = 0;
}

void BandedSmithWaterman::print_matrix() {
    if (gpu.H.empty()) dpxAlignPair(3, reference_str, query_str, match_weight, mismatch_weight, gap_weight, 0, band_width, true, gpu);
    dpxPrintScoreMatrix(reference_str, query_str, gpu.H);
}

void BandedSmithWaterman::score_matrix() {
#ifdef PRINT_MATRIX
    const bool wantMatrix = true;
#else
    const bool wantMatrix = false;
#endif
    dpxAlignPair(3 /* DPX_ALGO_BSW */, reference_str, query_str, match_weight, mismatch_weight, gap_weight, 0, band_width, wantMatrix, gpu);
}

void BandedSmithWaterman::backtrack() { max_score = gpu.score; }

void BandedSmithWaterman::align() {
    init_matrix();
    score_matrix();
    backtrack();
    print_results();
}

void BandedSmithWaterman::print_results() {
    PrintGuard stdoutIsMine; // lock ... unlock, as the other aligners do by hand
    if (max_score == 0) printf("%d | 0\n\n\n\n", pairNum);
    else printf("%d | %d\n%s\n%s\n%s\n", pairNum, max_score, gpu.refLine.c_str(), gpu.relLine.c_str(), gpu.qryLine.c_str());
}
