// BandedSmithWaterman.h -- banded local alignment on the MI355X engine.
// Constructor order mirrors c++/BandedSmithWaterman.h:51 (weights first, pairNum last).  Upstream never initialises
// band_width (.h:16) and its fill underflows (BandedSmithWaterman.cpp:80), so the executable semantics are those of
// python/LinearBandedSmithWaterman.py:62-104: cells with |i-j| <= band-1, everything else reads as 0.  The band is an
// optional trailing constructor argument here (default DPX_DEFAULT_BAND).  Runs in the HIP kernel k_banded_fill.
#pragma once
#include <deque>
#include <iomanip>
#include <iostream>
#include <vector>
#include "SequenceAligner.h"
#include "debug.h"
#include "printLock.h"
#include "DpxPair.h"

#ifndef DPX_DEFAULT_BAND
#define DPX_DEFAULT_BAND 128
#endif

class BandedSmithWaterman : public SequenceAligner {
  private:
    int match_weight;
    int mismatch_weight;
    int gap_weight;
    int band_width;
    int max_score;
    DpxPairResult gpu;

  public:
    BandedSmithWaterman(const std::string input_reference, const std::string input_query, const int match_weight,
                        const int mismatch_weight, const int gap_weight, const int pairNum, const int band_width = DPX_DEFAULT_BAND)
        : SequenceAligner(input_reference, input_query, pairNum), match_weight(match_weight), mismatch_weight(mismatch_weight),
          gap_weight(gap_weight), band_width(band_width), max_score(0) {}

    void init_matrix();
    void print_matrix();
    void score_matrix();
    void backtrack();
    void align();
    void print_results();
};
