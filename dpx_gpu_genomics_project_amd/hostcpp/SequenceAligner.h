// SequenceAligner.h -- abstract aligner interface of the MI355X engine's C++ host mirror.
//
// Same surface as the reference's base class (c++/SequenceAligner.h:6-28) so that its drivers
// (c++/main.cpp:62,87,113) compile against this directory unchanged: protected reference_str / query_str /
// pairNum, and the six pure virtuals init_matrix, print_matrix, score_matrix, backtrack, align, print_results.
// The derived classes here do not run the DP on the host: score_matrix() hands the pair to the HIP engine
// through the C ABI (include/dpx_align.h).
#pragma once
#include <string>

class SequenceAligner {
  protected:
    std::string reference_str; // columns of the DP matrix
    std::string query_str;     // rows of the DP matrix
    int pairNum;               // index printed in front of the score

  public:
    SequenceAligner(const std::string input_reference, const std::string input_query, const int pairNum)
        : reference_str(input_reference), query_str(input_query), pairNum(pairNum) {}
    virtual ~SequenceAligner() {}

    virtual void init_matrix() = 0;   // reset per-pair state
    virtual void print_matrix() = 0;  // dump the score matrix (debug)
    virtual void score_matrix() = 0;  // DP fill -- runs on the GPU
    virtual void backtrack() = 0;     // recover the alignment strings
    virtual void align() = 0;         // init + score + backtrack + print
    virtual void print_results() = 0; // "<pair> | <score>" + three lines
};
