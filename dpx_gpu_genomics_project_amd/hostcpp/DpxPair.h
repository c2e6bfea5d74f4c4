// DpxPair.h -- the one place where the C++ host mirror touches the engine: run a pair through the C ABI
// (include/dpx_align.h).  Used by the SequenceAligner-derived classes' score_matrix().  Thread-safe and re-entrant, as
// the reference's 20-pthread driver requires (c++/main.cpp:18-19,203); calls that arrive from several threads at about
// the same time are combined into one device batch (DpxPair.cpp), which is what makes the per-pair classes usable.
// Errors follow the reference's convention: message on stderr, exit(1) (c++/parseInput.cpp:12-15, cuda handleErrs()).
#pragma once
#include <string>
#include <vector>

struct DpxPairResult {
    int score = 0;
    int endRow = 0, endCol = 0;             // start cell of the traceback
    std::string refLine, relLine, qryLine;  // the three printed lines (empty for a zero-score local alignment)
    std::vector<short> H, I, D;             // (m+1) x (n+1) row-major, only when matrices were requested
};

// algo: dpx_algo (0 LNW, 1 LSW, 2 ANW, 3 BSW)
void dpxAlignPair(int algo, const std::string &reference, const std::string &query, int match, int mismatch, int gapOpen,
                  int gapExtend, int band, bool wantMatrices, DpxPairResult &out);

// shared "Matrix Dim / header / rows" dump used by every print_matrix()
void dpxPrintScoreMatrix(const std::string &reference, const std::string &query, const std::vector<short> &M);
