// timing.h -- microsecond wall-clock helpers (reference: c++/timing.h:5-9).  One global start mark, as there.
#pragma once
#include <stdint.h>
#include <sys/time.h>
#include <cstddef>

uint64_t start_timer();      // remember "now" and return it (usec since the epoch)
uint64_t get_time();         // usec since the epoch
uint64_t get_elapsed_time(); // usec since start_timer()
