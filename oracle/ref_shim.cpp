/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.  Builds into oracle/_ref/ (git-ignored).
 *
 * Thin C-ABI window onto the REAL reference classes, compiled from the sources
 * where they lie under $(REF)/c++ (never copied into this repo).  Used to pin
 * oracle/dpx_oracle.c and to generate tests/golden/.  `private` is opened up in
 * THIS translation unit only so the score / direction matrices can be read out
 * (SURVEY.md section 8c caveat 3).  Nothing here is shipped or measured as the
 * product.
 */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include <pthread.h>
#include <unistd.h>

#define private public
#define protected public
#include "LinearSmithWaterman.h"
#include "LinearNeedlemanWunsch.h"
#include "AffineNeedlemanWunsch.h"
#undef private
#undef protected
#include "FakeDPX.hpp"

namespace {

/* Capture everything written to fd 1 while `fn` runs (the classes print with cout + printf). */
template <class F>
std::string capture_stdout(F fn) {
    fflush(stdout);
    std::cout.flush();
    FILE *tmp = tmpfile();
    int saved = dup(1);
    dup2(fileno(tmp), 1);
    fn();
    fflush(stdout);
    std::cout.flush();
    dup2(saved, 1);
    close(saved);
    std::string out;
    rewind(tmp);
    char buf[4096];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, tmp)) > 0) out.append(buf, k);
    fclose(tmp);
    return out;
}

void put(char *dst, size_t cap, const std::string &s) {
    if (!dst || !cap) return;
    size_t k = std::min(cap - 1, s.size());
    memcpy(dst, s.data(), k);
    dst[k] = 0;
}

} // namespace

extern "C" {

/* All matrices are written row-major (m+1) x (n+1); pointers may be NULL.  `text` receives the
 * class's own stdout for the pair (score line + 3 alignment lines). */

int ref_lsw(const char *ref, const char *qry, int pairNum, int match, int mismatch, int gap, int32_t *H,
            uint8_t *dir, int32_t *score, char *text, size_t textCap) {
    LinearSmithWaterman a(ref, qry, pairNum, match, mismatch, gap);
    a.init_matrix();
    a.score_matrix();
    size_t m = a.query_str.size(), n = a.reference_str.size();
    if (H)
        for (size_t i = 0; i <= m; i++)
            for (size_t j = 0; j <= n; j++) H[i * (n + 1) + j] = a.memo[i][j];
    if (dir) {
        memset(dir, 0, (m + 1) * (n + 1));
        for (size_t i = 1; i <= m; i++)
            for (size_t j = 1; j <= n; j++) dir[i * (n + 1) + j] = (uint8_t)a.backtrack_memo[i - 1][j - 1];
    }
    a.backtrack();
    if (score) *score = a.max_score;
    std::string s = capture_stdout([&] { a.print_results(); });
    put(text, textCap, s);
    return 0;
}

int ref_lnw(const char *ref, const char *qry, int pairNum, int match, int mismatch, int gap, int32_t *H,
            uint8_t *dir, int32_t *score, char *text, size_t textCap) {
    LinearNeedlemanWunsch a(ref, qry, pairNum, match, mismatch, gap);
    a.init_matrix();
    a.score_matrix();
    size_t m = a.query_str.size(), n = a.reference_str.size();
    for (size_t i = 0; i <= m; i++)
        for (size_t j = 0; j <= n; j++) {
            if (H) H[i * (n + 1) + j] = a.memo[i][j];
            if (dir) dir[i * (n + 1) + j] = (uint8_t)a.backtrack_memo[i][j];
        }
    if (score) *score = a.memo[m][n];
    std::string s = capture_stdout([&] { a.backtrack(); }); /* LNW prints from backtrack() */
    put(text, textCap, s);
    return 0;
}

int ref_anw(const char *ref, const char *qry, int pairNum, int match, int mismatch, int gapOpen, int gapExtend,
            int32_t *H, int32_t *I, int32_t *D, uint8_t *dirH, uint8_t *dirI, uint8_t *dirD, int32_t *score,
            char *text, size_t textCap) {
    AffineNeedlemanWunsch a(ref, qry, pairNum, match, mismatch, gapOpen, gapExtend);
    a.init_matrix();
    a.score_matrix();
    size_t m = a.query_str.size(), n = a.reference_str.size();
    for (size_t i = 0; i <= m; i++)
        for (size_t j = 0; j <= n; j++) {
            size_t k = i * (n + 1) + j;
            if (H) H[k] = a.scoringMemo[i][j];
            if (I) I[k] = a.queryInsertionMemo[i][j];
            if (D) D[k] = a.queryDeletionMemo[i][j];
            if (dirH) dirH[k] = (uint8_t)a.scoringBacktrack[i][j];
            if (dirI) dirI[k] = (uint8_t)a.queryInsertionBacktrack[i][j];
            if (dirD) dirD[k] = (uint8_t)a.queryDeletionBacktrack[i][j];
        }
    if (score) *score = a.scoringMemo[m][n];
    std::string s = capture_stdout([&] { a.backtrack(); });
    put(text, textCap, s);
    return 0;
}

/* FakeDPX window: same op numbering as oracle/dpx_oracle.h.  pred bit0 = pred/pred_lo, bit1 = pred_hi. */
uint32_t ref_dpx(int op, uint32_t a, uint32_t b, uint32_t c, uint32_t *pred) {
    bool p = false, ph = false, pl = false;
    uint32_t r = 0;
    int sa = (int)a, sb = (int)b, sc = (int)c;
    switch (op) {
    case 0: r = FakeDPX::__vimax3_s32(sa, sb, sc); break;
    case 1: r = FakeDPX::__vimax3_s16x2(a, b, c); break;
    case 2: r = FakeDPX::__vimax3_u32(a, b, c); break;
    case 3: r = FakeDPX::__vimax3_u16x2(a, b, c); break;
    case 4: r = FakeDPX::__vimin3_s32(sa, sb, sc); break;
    case 5: r = FakeDPX::__vimin3_s16x2(a, b, c); break;
    case 6: r = FakeDPX::__vimin3_u32(a, b, c); break;
    case 7: r = FakeDPX::__vimin3_u16x2(a, b, c); break;
    case 8: r = FakeDPX::__vimax_s32_relu(sa, sb); break;
    case 9: r = FakeDPX::__vimax_s16x2_relu(a, b); break;
    case 10: r = FakeDPX::__vimin_s32_relu(sa, sb); break;
    case 11: r = FakeDPX::__vimin_s16x2_relu(a, b); break;
    case 12: r = FakeDPX::__vimax3_s32_relu(sa, sb, sc); break;
    case 13: r = FakeDPX::__vimax3_s16x2_relu(a, b, c); break;
    case 14: r = FakeDPX::__vimin3_s32_relu(sa, sb, sc); break;
    case 15: r = FakeDPX::__vimin3_s16x2_relu(a, b, c); break;
    case 16: r = FakeDPX::__vibmax_s32(sa, sb, &p); break;
    case 17: r = FakeDPX::__vibmax_u32(a, b, &p); break;
    case 18: r = FakeDPX::__vibmin_s32(sa, sb, &p); break;
    case 19: r = FakeDPX::__vibmin_u32(a, b, &p); break;
    case 20: r = FakeDPX::__vibmax_s16x2(a, b, &ph, &pl); break;
    case 21: r = FakeDPX::__vibmax_u16x2(a, b, &ph, &pl); break;
    case 22: r = FakeDPX::__vibmin_s16x2(a, b, &ph, &pl); break;
    case 23: r = FakeDPX::__vibmin_u16x2(a, b, &ph, &pl); break;
    case 24: r = FakeDPX::__viaddmax_s32(sa, sb, sc); break;
    case 25: r = FakeDPX::__viaddmax_u32(a, b, c); break;
    case 26: r = FakeDPX::__viaddmax_s16x2(a, b, c); break;
    case 27: r = FakeDPX::__viaddmax_u16x2(a, b, c); break;
    case 28: r = FakeDPX::__viaddmin_s32(sa, sb, sc); break;
    case 29: r = FakeDPX::__viaddmin_u32(a, b, c); break;
    case 30: r = FakeDPX::__viaddmin_s16x2(a, b, c); break;
    case 31: r = FakeDPX::__viaddmin_u16x2(a, b, c); break;
    case 32: r = FakeDPX::__viaddmax_s32_relu(sa, sb, sc); break;
    case 33: r = FakeDPX::__viaddmax_s16x2_relu(a, b, c); break;
    case 34: r = FakeDPX::__viaddmin_s32_relu(sa, sb, sc); break;
    case 35: r = FakeDPX::__viaddmin_s16x2_relu(a, b, c); break;
    default: break;
    }
    if (pred) *pred = (op >= 20 && op <= 23) ? (uint32_t)((ph ? 2 : 0) | (pl ? 1 : 0)) : (uint32_t)(p ? 1 : 0);
    return r;
}

} /* extern "C" */
