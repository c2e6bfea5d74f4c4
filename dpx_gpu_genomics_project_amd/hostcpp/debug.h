// debug.h -- compile-time print switches, same names as the reference's c++/debug.h:1-9.
#pragma once

// #define DEBUG
#ifdef DEBUG
#define DEBUG_PRINT(x, y) std::cout << x << " | " << y << std::endl;
#endif

// #define PRINT_MATRIX
// #define PRINT_EXTRA
