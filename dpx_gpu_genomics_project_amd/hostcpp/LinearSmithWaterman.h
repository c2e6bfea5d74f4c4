// LinearSmithWaterman.h -- local alignment, linear gap penalty, on the MI355X engine.
// Class name, constructor and methods mirror c++/LinearSmithWaterman.h:11-70 (ctor :51).  The recurrence
// (c++/LinearSmithWaterman.cpp:70-114) runs in the HIP kernel k_linear_fill<R, LOCAL=true>; the start cell is the
// first strict maximum in row-major order (:145-157) and the walk stops at the first zero cell (:222).
#pragma once
#include <deque>
#include <iomanip>
#include <iostream>
#include <vector>
#include "SequenceAligner.h"
#include "debug.h"
#include "printLock.h"
#include "DpxPair.h"

// #define BACKTRACK_ALL   (the reference's enumerate-every-optimal-path mode is not part of the GPU hot path)

class LinearSmithWaterman : public SequenceAligner {
  private:
    int match_weight;
    int mismatch_weight;
    int gap_weight;
    int max_score;
    DpxPairResult gpu; // score, start cell, alignment lines (+ H when a matrix dump is compiled in)

  public:
    LinearSmithWaterman(const std::string input_reference, const std::string input_query, const int pairNum,
                        const int match_weight, const int mismatch_weight, const int gap_weight)
        : SequenceAligner(input_reference, input_query, pairNum), match_weight(match_weight),
          mismatch_weight(mismatch_weight), gap_weight(gap_weight), max_score(0) {}

    void init_matrix();
    void print_matrix();
    void score_matrix();
    void backtrack();
    void align();
    void print_results();
};
