#include "timing.h"

#include <atomic>

namespace {
std::atomic<uint64_t> g_start{0}; // the reference keeps a plain global (timing.cpp:3-4); atomic makes reads race-free

uint64_t now_usec() {
    timeval tv;
    gettimeofday(&tv, nullptr);
    return (uint64_t)tv.tv_sec * 1000000ull + (uint64_t)tv.tv_usec;
}
} // namespace

uint64_t start_timer() {
    const uint64_t t = now_usec();
    g_start.store(t);
    return t;
}

uint64_t get_time() { return now_usec(); }

uint64_t get_elapsed_time() { return now_usec() - g_start.load(); }
