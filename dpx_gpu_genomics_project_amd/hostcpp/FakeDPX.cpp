#include "FakeDPX.hpp"

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/dpx_align.h"

namespace {
// op numbers follow the declaration order above (== dpx_prim_eval numbering)
uint32_t ev(int op, uint32_t a, uint32_t b, uint32_t c, uint32_t *pred = nullptr) {
    int32_t o = op;
    uint32_t r = 0, p = 0;
    const int rc = dpx_prim_eval(&o, &a, &b, &c, 1, &r, &p);
    if (rc != DPX_OK) {
        fprintf(stderr, "FakeDPX: dpx_prim_eval failed: %s (%s)\n", dpx_strerror(rc), dpx_last_error());
        exit(1);
    }
    if (pred) *pred = p;
    return r;
}
uint32_t ev1(int op, uint32_t a, uint32_t b, bool *pred) {
    uint32_t p;
    const uint32_t r = ev(op, a, b, 0, &p);
    *pred = (p & 1u) != 0;
    return r;
}
uint32_t ev2(int op, uint32_t a, uint32_t b, bool *hi, bool *lo) {
    uint32_t p;
    const uint32_t r = ev(op, a, b, 0, &p);
    *hi = (p & 2u) != 0;
    *lo = (p & 1u) != 0;
    return r;
}
} // namespace

typedef unsigned int U;
int FakeDPX::__vimax3_s32(const int a, const int b, const int c) { return (int)ev(0, (U)a, (U)b, (U)c); }
U FakeDPX::__vimax3_s16x2(const U a, const U b, const U c) { return ev(1, a, b, c); }
U FakeDPX::__vimax3_u32(const U a, const U b, const U c) { return ev(2, a, b, c); }
U FakeDPX::__vimax3_u16x2(const U a, const U b, const U c) { return ev(3, a, b, c); }
int FakeDPX::__vimin3_s32(const int a, const int b, const int c) { return (int)ev(4, (U)a, (U)b, (U)c); }
U FakeDPX::__vimin3_s16x2(const U a, const U b, const U c) { return ev(5, a, b, c); }
U FakeDPX::__vimin3_u32(const U a, const U b, const U c) { return ev(6, a, b, c); }
U FakeDPX::__vimin3_u16x2(const U a, const U b, const U c) { return ev(7, a, b, c); }
int FakeDPX::__vimax_s32_relu(const int a, const int b) { return (int)ev(8, (U)a, (U)b, 0); }
U FakeDPX::__vimax_s16x2_relu(const U a, const U b) { return ev(9, a, b, 0); }
int FakeDPX::__vimin_s32_relu(const int a, const int b) { return (int)ev(10, (U)a, (U)b, 0); }
U FakeDPX::__vimin_s16x2_relu(const U a, const U b) { return ev(11, a, b, 0); }
int FakeDPX::__vimax3_s32_relu(const int a, const int b, const int c) { return (int)ev(12, (U)a, (U)b, (U)c); }
U FakeDPX::__vimax3_s16x2_relu(const U a, const U b, const U c) { return ev(13, a, b, c); }
int FakeDPX::__vimin3_s32_relu(const int a, const int b, const int c) { return (int)ev(14, (U)a, (U)b, (U)c); }
U FakeDPX::__vimin3_s16x2_relu(const U a, const U b, const U c) { return ev(15, a, b, c); }
int FakeDPX::__vibmax_s32(const int a, const int b, bool *const pred) { return (int)ev1(16, (U)a, (U)b, pred); }
U FakeDPX::__vibmax_u32(const U a, const U b, bool *const pred) { return ev1(17, a, b, pred); }
int FakeDPX::__vibmin_s32(const int a, const int b, bool *const pred) { return (int)ev1(18, (U)a, (U)b, pred); }
U FakeDPX::__vibmin_u32(const U a, const U b, bool *const pred) { return ev1(19, a, b, pred); }
U FakeDPX::__vibmax_s16x2(const U a, const U b, bool *const hi, bool *const lo) { return ev2(20, a, b, hi, lo); }
U FakeDPX::__vibmax_u16x2(const U a, const U b, bool *const hi, bool *const lo) { return ev2(21, a, b, hi, lo); }
U FakeDPX::__vibmin_s16x2(const U a, const U b, bool *const hi, bool *const lo) { return ev2(22, a, b, hi, lo); }
U FakeDPX::__vibmin_u16x2(const U a, const U b, bool *const hi, bool *const lo) { return ev2(23, a, b, hi, lo); }
int FakeDPX::__viaddmax_s32(const int a, const int b, const int c) { return (int)ev(24, (U)a, (U)b, (U)c); }
U FakeDPX::__viaddmax_u32(const U a, const U b, const U c) { return ev(25, a, b, c); }
U FakeDPX::__viaddmax_s16x2(const U a, const U b, const U c) { return ev(26, a, b, c); }
U FakeDPX::__viaddmax_u16x2(const U a, const U b, const U c) { return ev(27, a, b, c); }
int FakeDPX::__viaddmin_s32(const int a, const int b, const int c) { return (int)ev(28, (U)a, (U)b, (U)c); }
U FakeDPX::__viaddmin_u32(const U a, const U b, const U c) { return ev(29, a, b, c); }
U FakeDPX::__viaddmin_s16x2(const U a, const U b, const U c) { return ev(30, a, b, c); }
U FakeDPX::__viaddmin_u16x2(const U a, const U b, const U c) { return ev(31, a, b, c); }
int FakeDPX::__viaddmax_s32_relu(const int a, const int b, const int c) { return (int)ev(32, (U)a, (U)b, (U)c); }
U FakeDPX::__viaddmax_s16x2_relu(const U a, const U b, const U c) { return ev(33, a, b, c); }
int FakeDPX::__viaddmin_s32_relu(const int a, const int b, const int c) { return (int)ev(34, (U)a, (U)b, (U)c); }
U FakeDPX::__viaddmin_s16x2_relu(const U a, const U b, const U c) { return ev(35, a, b, c); }
