// LinearNeedlemanWunsch.h -- global alignment, linear gap penalty, on the MI355X engine.
// Mirrors c++/LinearNeedlemanWunsch.h:12-59 (ctor :42-43).  Borders i*gap / j*gap (c++/LinearNeedlemanWunsch.cpp:31-41)
// and the __vibmax cell update (:105-128, tie priority INSERTION > DELETION > diagonal) run in the HIP kernel
// k_linear_fill<R, LOCAL=false>; the score is H[m][n].
#pragma once
#include <deque>
#include <iomanip>
#include <iostream>
#include <vector>
#include "SequenceAligner.h"
#include "debug.h"
#include "printLock.h"
#include "DpxPair.h"

class LinearNeedlemanWunsch : public SequenceAligner {
  private:
    int match_weight;
    int mismatch_weight;
    int gap_weight;
    DpxPairResult gpu;

  public:
    LinearNeedlemanWunsch(const std::string input_reference, const std::string input_query, const int pairNum,
                          const int match_weight, const int mismatch_weight, const int gap_weight)
        : SequenceAligner(input_reference, input_query, pairNum), match_weight(match_weight),
          mismatch_weight(mismatch_weight), gap_weight(gap_weight) {}

    void init_matrix();
    void print_matrix();
    void score_matrix();
    void backtrack(); // prints the result block, as the reference's LNW does (LinearNeedlemanWunsch.cpp:199-221)
    void align();
    void print_results();
};
