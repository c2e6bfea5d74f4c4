// AffineNeedlemanWunsch.h -- global alignment with affine (Gotoh) gaps on the MI355X engine.
// Mirrors c++/AffineNeedlemanWunsch.h:12-78 (ctor :59-61).  The three-matrix recurrence
// (c++/AffineNeedlemanWunsch.cpp:167-240) runs in the HIP kernel k_affine_fill; the three-state walk (:242-360)
// runs in the device traceback.
#pragma once
#include <deque>
#include <iomanip>
#include <iostream>
#include <vector>
#include "SequenceAligner.h"
#include "debug.h"
#include "printLock.h"
#include "DpxPair.h"

class AffineNeedlemanWunsch : public SequenceAligner {
  private:
    int matchWeight;
    int mismatchWeight;
    int gapOpenWeight;
    int gapExtendWeight;
    DpxPairResult gpu;

  public:
    AffineNeedlemanWunsch(const std::string inputReference, const std::string inputQuery, const int pairNum,
                          const int matchWeight, const int mismatchWeight, const int gapOpenWeight, const int gapExtendWeight)
        : SequenceAligner(inputReference, inputQuery, pairNum), matchWeight(matchWeight), mismatchWeight(mismatchWeight),
          gapOpenWeight(gapOpenWeight), gapExtendWeight(gapExtendWeight) {}

    void init_matrix();
    void print_matrix();
    void score_matrix();
    void backtrack(); // prints the result block (AffineNeedlemanWunsch.cpp:361-384)
    void align();
    void print_results();
};
