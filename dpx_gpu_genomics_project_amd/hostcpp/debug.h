// debug.h -- compile-time print switches.  The names are the reference's (c++/debug.h:1-9) because its sources and
// main.cpp test them with #ifdef; all are off by default, turn one on with -D<NAME> or by uncommenting it here.
#ifndef DPX_HOSTCPP_DEBUG_H
#define DPX_HOSTCPP_DEBUG_H

// #define PRINT_MATRIX   // aligners print the score matrix before and after the fill (dpxPrintScoreMatrix)
// #define PRINT_EXTRA    // extra per-pair chatter of the reference's mains
// #define DEBUG          // enables DEBUG_PRINT(label, value)

#ifdef DEBUG
#include <iostream>
#define DEBUG_PRINT(label, value) (std::cout << (label) << " | " << (value) << std::endl)
#endif
#endif
