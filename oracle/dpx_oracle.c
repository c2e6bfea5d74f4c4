/*
 * dpx_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see dpx_oracle.h).
 * Plain-C restatement of the reference CPU path; every function cites the
 * reference file:line it follows.  Never linked into the shipped library.
 */
#include "dpx_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define AT(M, i, j) (M)[(size_t)(i) * (size_t)(n + 1) + (size_t)(j)]

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* __vibmax_s32: max with pred = (a >= b).  c++/FakeDPX.cpp:145-153 */
static inline int vibmax_s32(int a, int b, int *pred) {
    if (a >= b) { *pred = 1; return a; }
    *pred = 0;
    return b;
}

/* ------------------------------------------------------------------ LSW */
/* Shared by orc_lsw_fill (band <= 0: unbanded) and orc_bsw_fill. */
static void sw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gap, int band,
                    int32_t *H, uint8_t *dir, int32_t *score, int32_t *endRow, int32_t *endCol) {
    int32_t *own = NULL;
    if (!H) { own = (int32_t *)calloc((size_t)(m + 1) * (n + 1), sizeof(int32_t)); H = own; }
    else memset(H, 0, (size_t)(m + 1) * (n + 1) * sizeof(int32_t)); /* LinearSmithWaterman.cpp:14-17: all zero */
    if (dir) memset(dir, 0, (size_t)(m + 1) * (n + 1));

    for (int i = 1; i <= m; i++) {
        int jlo = 1, jhi = n;
        if (band > 0) { /* LinearBandedSmithWaterman.py:71 */
            jlo = 1 + imax(0, i - band);
            jhi = imin(i + band, n + 1) - 1;
        }
        for (int j = jlo; j <= jhi; j++) {
            /* LinearSmithWaterman.cpp:82-97 */
            int up = AT(H, i - 1, j) + gap;
            int left = AT(H, i, j - 1) + gap;
            int eq = (qry[i - 1] == ref[j - 1]);
            int corner = AT(H, i - 1, j - 1) + (eq ? match : mismatch);
            int t = imax(up, imax(left, corner)); /* :100 */
            int h = imax(0, t);                   /* :103 ReLU */
            AT(H, i, j) = h;
            if (dir) { /* :106-109: NONE iff t<0, else UPPER, then LEFT, then CORNER */
                uint8_t d = ORC_NONE_MAIN;
                if (t >= 0) {
                    if (up == h) d = ORC_QUERY_DELETION;        /* UPPER_GAP */
                    else if (left == h) d = ORC_QUERY_INSERTION; /* LEFT_GAP */
                    else d = eq ? ORC_MATCH : ORC_MISMATCH;
                }
                AT(dir, i, j) = d;
            }
        }
    }
    /* LinearSmithWaterman.cpp:145-157: first strictly-greater cell in a row-major scan, starting at 0 */
    int best = 0, br = 0, bc = 0;
    for (int i = 0; i <= m; i++)
        for (int j = 0; j <= n; j++)
            if (AT(H, i, j) > best) { best = AT(H, i, j); br = i; bc = j; }
    if (score) *score = best;
    if (endRow) *endRow = br;
    if (endCol) *endCol = bc;
    free(own);
}

void orc_lsw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gap, int32_t *H,
                  uint8_t *dir, int32_t *score, int32_t *endRow, int32_t *endCol) {
    sw_fill(ref, n, qry, m, match, mismatch, gap, 0, H, dir, score, endRow, endCol);
}

void orc_bsw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gap, int band,
                  int32_t *H, uint8_t *dir, int32_t *score, int32_t *endRow, int32_t *endCol) {
    sw_fill(ref, n, qry, m, match, mismatch, gap, band > 0 ? band : 1, H, dir, score, endRow, endCol);
}

/* ------------------------------------------------------------------ LNW */
void orc_lnw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gap, int32_t *H,
                  uint8_t *dir, int32_t *score) {
    int32_t *own = NULL;
    if (!H) { own = (int32_t *)malloc((size_t)(m + 1) * (n + 1) * sizeof(int32_t)); H = own; }
    AT(H, 0, 0) = 0;
    if (dir) AT(dir, 0, 0) = ORC_NONE_MAIN;
    /* LinearNeedlemanWunsch.cpp:31-41 */
    for (int i = 1; i <= m; i++) { AT(H, i, 0) = i * gap; if (dir) AT(dir, i, 0) = ORC_QUERY_DELETION; }
    for (int j = 1; j <= n; j++) { AT(H, 0, j) = j * gap; if (dir) AT(dir, 0, j) = ORC_QUERY_INSERTION; }
    for (int i = 1; i <= m; i++) {
        for (int j = 1; j <= n; j++) {
            /* :105-128 */
            int eq = (qry[i - 1] == ref[j - 1]);
            int mm = AT(H, i - 1, j - 1) + (eq ? match : mismatch);
            uint8_t d = eq ? ORC_MATCH : ORC_MISMATCH;
            int del = AT(H, i - 1, j) + gap;
            int ins = AT(H, i, j - 1) + gap;
            int pred;
            int v = vibmax_s32(del, mm, &pred);
            if (pred) d = ORC_QUERY_DELETION;
            v = vibmax_s32(ins, v, &pred);
            if (pred) d = ORC_QUERY_INSERTION;
            AT(H, i, j) = v;
            if (dir) AT(dir, i, j) = d;
        }
    }
    if (score) *score = AT(H, m, n);
    free(own);
}

/* ------------------------------------------------------------------ ANW */
void orc_anw_fill(const char *ref, int n, const char *qry, int m, int match, int mismatch, int gapOpen,
                  int gapExtend, int32_t *H, int32_t *I, int32_t *D, uint8_t *dirH, uint8_t *dirI, uint8_t *dirD,
                  int32_t *score) {
    size_t cells = (size_t)(m + 1) * (n + 1);
    int32_t *ownH = NULL, *ownI = NULL, *ownD = NULL;
    if (!H) { ownH = (int32_t *)malloc(cells * sizeof(int32_t)); H = ownH; }
    if (!I) { ownI = (int32_t *)malloc(cells * sizeof(int32_t)); I = ownI; }
    if (!D) { ownD = (int32_t *)malloc(cells * sizeof(int32_t)); D = ownD; }
    /* AffineNeedlemanWunsch.cpp:24-27: all three zero-initialised */
    memset(H, 0, cells * sizeof(int32_t));
    memset(I, 0, cells * sizeof(int32_t));
    memset(D, 0, cells * sizeof(int32_t));
    if (dirH) memset(dirH, 0, cells);
    if (dirI) memset(dirI, 0, cells);
    if (dirD) memset(dirD, 0, cells);
    /* :43-53 -- note H[0][0] stays 0 */
    for (int i = 1; i <= m; i++) { AT(H, i, 0) = gapOpen + i * gapExtend; if (dirH) AT(dirH, i, 0) = ORC_QUERY_DELETION; }
    for (int j = 1; j <= n; j++) { AT(H, 0, j) = gapOpen + j * gapExtend; if (dirH) AT(dirH, 0, j) = ORC_QUERY_INSERTION; }

    for (int i = 1; i <= m; i++) {
        for (int j = 1; j <= n; j++) {
            int pred;
            /* :185-197 deletion matrix (vertical gap) */
            if (i == 1) {
                AT(D, i, j) = AT(H, i - 1, j) + gapOpen + gapExtend;
                if (dirD) AT(dirD, i, j) = ORC_GAP_OPEN;
            } else {
                AT(D, i, j) = vibmax_s32(AT(H, i - 1, j) + gapOpen + gapExtend, AT(D, i - 1, j) + gapExtend, &pred);
                if (dirD) AT(dirD, i, j) = pred ? ORC_GAP_OPEN : ORC_GAP_EXTEND;
            }
            /* :201-213 insertion matrix (horizontal gap) */
            if (j == 1) {
                AT(I, i, j) = AT(H, i, j - 1) + gapOpen + gapExtend;
                if (dirI) AT(dirI, i, j) = ORC_GAP_OPEN;
            } else {
                AT(I, i, j) = vibmax_s32(AT(H, i, j - 1) + gapOpen + gapExtend, AT(I, i, j - 1) + gapExtend, &pred);
                if (dirI) AT(dirI, i, j) = pred ? ORC_GAP_OPEN : ORC_GAP_EXTEND;
            }
            /* :216-236 */
            int eq = (qry[i - 1] == ref[j - 1]);
            int mm = AT(H, i - 1, j - 1) + (eq ? match : mismatch);
            uint8_t d = eq ? ORC_MATCH : ORC_MISMATCH;
            int v = vibmax_s32(AT(D, i, j), mm, &pred);
            if (pred) d = ORC_QUERY_DELETION;
            v = vibmax_s32(AT(I, i, j), v, &pred);
            if (pred) d = ORC_QUERY_INSERTION;
            AT(H, i, j) = v;
            if (dirH) AT(dirH, i, j) = d;
        }
    }
    if (score) *score = AT(H, m, n);
    free(ownH); free(ownI); free(ownD);
}

/* ------------------------------------------------------------------ tracebacks */
/* The reference prepends to std::strings; here we fill from the back of a scratch
 * buffer and memmove to the front -- same resulting strings. */
struct tb { char *r, *x, *q; int cap, pos; };
static void tb_init(struct tb *t, char *r, char *x, char *q, int cap) { t->r = r; t->x = x; t->q = q; t->cap = cap; t->pos = cap; }
static void tb_push(struct tb *t, char rc, char xc, char qc) { t->pos--; t->r[t->pos] = rc; t->x[t->pos] = xc; t->q[t->pos] = qc; }
static int tb_finish(struct tb *t) {
    int len = t->cap - t->pos;
    memmove(t->r, t->r + t->pos, (size_t)len); t->r[len] = 0;
    memmove(t->x, t->x + t->pos, (size_t)len); t->x[len] = 0;
    memmove(t->q, t->q + t->pos, (size_t)len); t->q[len] = 0;
    return len;
}

int orc_lsw_traceback(const char *ref, int n, const char *qry, int m, const int32_t *H, const uint8_t *dir, int endRow,
                      int endCol, char *refOut, char *relOut, char *qryOut) {
    struct tb t; tb_init(&t, refOut, relOut, qryOut, m + n + 1);
    int i = endRow, j = endCol;
    if (AT(H, i, j) <= 0) { refOut[0] = relOut[0] = qryOut[0] = 0; return 0; } /* score 0: no path (cpp:253-257) */
    for (;;) {
        /* LinearSmithWaterman.cpp:170-209 */
        switch (AT(dir, i, j)) {
        case ORC_MATCH: tb_push(&t, ref[j - 1], '*', qry[i - 1]); i--; j--; break;
        case ORC_MISMATCH: tb_push(&t, ref[j - 1], '|', qry[i - 1]); i--; j--; break;
        case ORC_QUERY_INSERTION: tb_push(&t, ref[j - 1], ' ', '_'); j--; break;     /* LEFT_GAP */
        case ORC_QUERY_DELETION: tb_push(&t, '_', ' ', qry[i - 1]); i--; break;       /* UPPER_GAP */
        default: return -1;
        }
        if (AT(H, i, j) == 0) break; /* :222 */
    }
    return tb_finish(&t);
}

int orc_lnw_traceback(const char *ref, int n, const char *qry, int m, const uint8_t *dir, char *refOut, char *relOut,
                      char *qryOut) {
    struct tb t; tb_init(&t, refOut, relOut, qryOut, m + n + 1);
    int i = m, j = n;
    while (i != 0 || j != 0) { /* LinearNeedlemanWunsch.cpp:153-197 */
        switch (AT(dir, i, j)) {
        case ORC_MATCH: tb_push(&t, ref[j - 1], '*', qry[i - 1]); i--; j--; break;
        case ORC_MISMATCH: tb_push(&t, ref[j - 1], '|', qry[i - 1]); i--; j--; break;
        case ORC_QUERY_DELETION: tb_push(&t, '_', ' ', qry[i - 1]); i--; break;
        case ORC_QUERY_INSERTION: tb_push(&t, ref[j - 1], ' ', '_'); j--; break;
        default: return -1;
        }
    }
    return tb_finish(&t);
}

int orc_anw_traceback(const char *ref, int n, const char *qry, int m, const uint8_t *dirH, const uint8_t *dirI,
                      const uint8_t *dirD, char *refOut, char *relOut, char *qryOut) {
    struct tb t; tb_init(&t, refOut, relOut, qryOut, m + n + 1);
    int i = m, j = n;
    enum { SCORING, INSERTION, DELETION } cur = SCORING;
    while (i != 0 && j != 0) { /* AffineNeedlemanWunsch.cpp:258-346 */
        if (cur == SCORING) {
            switch (AT(dirH, i, j)) {
            case ORC_MATCH: tb_push(&t, ref[j - 1], '*', qry[i - 1]); i--; j--; break;
            case ORC_MISMATCH: tb_push(&t, ref[j - 1], '|', qry[i - 1]); i--; j--; break;
            case ORC_QUERY_DELETION: cur = DELETION; break;
            case ORC_QUERY_INSERTION: cur = INSERTION; break;
            default: return -1;
            }
        } else if (cur == INSERTION) {
            switch (AT(dirI, i, j)) {
            case ORC_GAP_OPEN: cur = SCORING; break;
            case ORC_GAP_EXTEND: cur = INSERTION; break;
            default: return -1;
            }
            tb_push(&t, ref[j - 1], ' ', '_'); j--;
        } else {
            switch (AT(dirD, i, j)) {
            case ORC_GAP_OPEN: cur = SCORING; break;
            case ORC_GAP_EXTEND: cur = DELETION; break;
            default: return -1;
            }
            tb_push(&t, '_', ' ', qry[i - 1]); i--;
        }
    }
    while (i > 0) { tb_push(&t, '_', ' ', qry[i - 1]); i--; }   /* :348-353 */
    while (j > 0) { tb_push(&t, ref[j - 1], ' ', '_'); j--; }   /* :355-360 */
    return tb_finish(&t);
}

/* ------------------------------------------------------------------ FakeDPX */
static inline int16_t hi16(uint32_t v) { return (int16_t)(v >> 16); }
static inline int16_t lo16(uint32_t v) { return (int16_t)(v & 0xFFFF); }
static inline uint16_t uhi16(uint32_t v) { return (uint16_t)(v >> 16); }
static inline uint16_t ulo16(uint32_t v) { return (uint16_t)(v & 0xFFFF); }
static inline uint32_t pack16(uint32_t hi, uint32_t lo) { return ((hi & 0xFFFF) << 16) | (lo & 0xFFFF); }
static inline int s_max(int a, int b) { return a > b ? a : b; }
static inline int s_min(int a, int b) { return a < b ? a : b; }
static inline uint32_t u_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline uint32_t u_min(uint32_t a, uint32_t b) { return a < b ? a : b; }

uint32_t orc_dpx(int op, uint32_t a, uint32_t b, uint32_t c, uint32_t *pred) {
    uint32_t p = 0, r = 0;
    int sa = (int)a, sb = (int)b, sc = (int)c;
    switch (op) {
    /* FakeDPX.cpp:11-92 three-input min/max */
    case ORC_VIMAX3_S32: r = (uint32_t)s_max(s_max(sa, sb), sc); break;
    case ORC_VIMAX3_S16X2: r = pack16((uint32_t)s_max(s_max(hi16(a), hi16(b)), hi16(c)), (uint32_t)s_max(s_max(lo16(a), lo16(b)), lo16(c))); break;
    case ORC_VIMAX3_U32: r = u_max(u_max(a, b), c); break;
    case ORC_VIMAX3_U16X2: r = pack16(u_max(u_max(uhi16(a), uhi16(b)), uhi16(c)), u_max(u_max(ulo16(a), ulo16(b)), ulo16(c))); break;
    case ORC_VIMIN3_S32: r = (uint32_t)s_min(s_min(sa, sb), sc); break;
    case ORC_VIMIN3_S16X2: r = pack16((uint32_t)s_min(s_min(hi16(a), hi16(b)), hi16(c)), (uint32_t)s_min(s_min(lo16(a), lo16(b)), lo16(c))); break;
    case ORC_VIMIN3_U32: r = u_min(u_min(a, b), c); break;
    case ORC_VIMIN3_U16X2: r = pack16(u_min(u_min(uhi16(a), uhi16(b)), uhi16(c)), u_min(u_min(ulo16(a), ulo16(b)), ulo16(c))); break;
    /* :97-118 two-input + ReLU */
    case ORC_VIMAX_S32_RELU: r = (uint32_t)s_max(s_max(sa, sb), 0); break;
    case ORC_VIMAX_S16X2_RELU: r = pack16((uint32_t)s_max(s_max(hi16(a), hi16(b)), 0), (uint32_t)s_max(s_max(lo16(a), lo16(b)), 0)); break;
    case ORC_VIMIN_S32_RELU: r = (uint32_t)s_max(s_min(sa, sb), 0); break;
    case ORC_VIMIN_S16X2_RELU: r = pack16((uint32_t)s_max(s_min(hi16(a), hi16(b)), 0), (uint32_t)s_max(s_min(lo16(a), lo16(b)), 0)); break;
    /* :122-138 three-input + ReLU */
    case ORC_VIMAX3_S32_RELU: r = (uint32_t)s_max(s_max(s_max(sa, sb), sc), 0); break;
    case ORC_VIMAX3_S16X2_RELU: r = pack16((uint32_t)s_max(s_max(s_max(hi16(a), hi16(b)), hi16(c)), 0), (uint32_t)s_max(s_max(s_max(lo16(a), lo16(b)), lo16(c)), 0)); break;
    case ORC_VIMIN3_S32_RELU: r = (uint32_t)s_max(s_min(s_min(sa, sb), sc), 0); break;
    /* reference :137 nests two ReLU'd mins: max(min(max(min(a,b),0),c),0) */
    case ORC_VIMIN3_S16X2_RELU: r = pack16((uint32_t)s_max(s_min(s_max(s_min(hi16(a), hi16(b)), 0), hi16(c)), 0), (uint32_t)s_max(s_min(s_max(s_min(lo16(a), lo16(b)), 0), lo16(c)), 0)); break;
    /* :143-181 predicate-returning scalar */
    case ORC_VIBMAX_S32: p = sa >= sb; r = p ? a : b; break;
    case ORC_VIBMAX_U32: p = a >= b; r = p ? a : b; break;
    case ORC_VIBMIN_S32: p = sa <= sb; r = p ? a : b; break;
    case ORC_VIBMIN_U32: p = a <= b; r = p ? a : b; break;
    /* :185-291 predicate-returning packed: bit1 = pred_hi, bit0 = pred_lo */
    case ORC_VIBMAX_S16X2: { int ph = hi16(a) >= hi16(b), pl = lo16(a) >= lo16(b); p = (uint32_t)(ph << 1 | pl); r = pack16(ph ? uhi16(a) : uhi16(b), pl ? ulo16(a) : ulo16(b)); break; }
    case ORC_VIBMAX_U16X2: { int ph = uhi16(a) >= uhi16(b), pl = ulo16(a) >= ulo16(b); p = (uint32_t)(ph << 1 | pl); r = pack16(ph ? uhi16(a) : uhi16(b), pl ? ulo16(a) : ulo16(b)); break; }
    case ORC_VIBMIN_S16X2: { int ph = hi16(a) <= hi16(b), pl = lo16(a) <= lo16(b); p = (uint32_t)(ph << 1 | pl); r = pack16(ph ? uhi16(a) : uhi16(b), pl ? ulo16(a) : ulo16(b)); break; }
    case ORC_VIBMIN_U16X2: { int ph = uhi16(a) <= uhi16(b), pl = ulo16(a) <= ulo16(b); p = (uint32_t)(ph << 1 | pl); r = pack16(ph ? uhi16(a) : uhi16(b), pl ? ulo16(a) : ulo16(b)); break; }
    /* :296-366 add-then-min/max (16-bit adds wrap, as the reference's short arithmetic does) */
    case ORC_VIADDMAX_S32: r = (uint32_t)s_max(sa + sb, sc); break;
    case ORC_VIADDMAX_U32: r = u_max(a + b, c); break;
    case ORC_VIADDMAX_S16X2: r = pack16((uint32_t)s_max((int16_t)(hi16(a) + hi16(b)), hi16(c)), (uint32_t)s_max((int16_t)(lo16(a) + lo16(b)), lo16(c))); break;
    case ORC_VIADDMAX_U16X2: r = pack16(u_max((uint16_t)(uhi16(a) + uhi16(b)), uhi16(c)), u_max((uint16_t)(ulo16(a) + ulo16(b)), ulo16(c))); break;
    case ORC_VIADDMIN_S32: r = (uint32_t)s_min(sa + sb, sc); break;
    case ORC_VIADDMIN_U32: r = u_min(a + b, c); break;
    case ORC_VIADDMIN_S16X2: r = pack16((uint32_t)s_min((int16_t)(hi16(a) + hi16(b)), hi16(c)), (uint32_t)s_min((int16_t)(lo16(a) + lo16(b)), lo16(c))); break;
    case ORC_VIADDMIN_U16X2: r = pack16(u_min((uint16_t)(uhi16(a) + uhi16(b)), uhi16(c)), u_min((uint16_t)(ulo16(a) + ulo16(b)), ulo16(c))); break;
    /* :371-404 add-then-min/max + ReLU */
    case ORC_VIADDMAX_S32_RELU: r = (uint32_t)s_max(s_max(sa + sb, sc), 0); break;
    case ORC_VIADDMAX_S16X2_RELU: r = pack16((uint32_t)s_max(s_max((int16_t)(hi16(a) + hi16(b)), hi16(c)), 0), (uint32_t)s_max(s_max((int16_t)(lo16(a) + lo16(b)), lo16(c)), 0)); break;
    case ORC_VIADDMIN_S32_RELU: r = (uint32_t)s_max(s_min(sa + sb, sc), 0); break;
    case ORC_VIADDMIN_S16X2_RELU: r = pack16((uint32_t)s_max(s_min((int16_t)(hi16(a) + hi16(b)), hi16(c)), 0), (uint32_t)s_max(s_min((int16_t)(lo16(a) + lo16(b)), lo16(c)), 0)); break;
    default: break;
    }
    if (pred) *pred = p;
    return r;
}

/* ------------------------------------------------------------------ timed CPU batch (bench.py cpu_baseline, kind "port") */
struct job {
    int algo; const char *seq; const int32_t *pairs; int lo, hi;
    int match, mismatch, gapOpen, gapExtend, band; int32_t *scores;
};

static void *job_run(void *arg) {
    struct job *jb = (struct job *)arg;
    for (int p = jb->lo; p < jb->hi; p++) {
        const int32_t *sp = jb->pairs + 4 * (size_t)p; /* seqPair: referenceIdx, referenceSize, queryIdx, querySize */
        const char *ref = jb->seq + sp[0]; int n = sp[1];
        const char *qry = jb->seq + sp[2]; int m = sp[3];
        int32_t sc = 0;
        switch (jb->algo) {
        case 0: orc_lnw_fill(ref, n, qry, m, jb->match, jb->mismatch, jb->gapOpen, NULL, NULL, &sc); break;
        case 1: orc_lsw_fill(ref, n, qry, m, jb->match, jb->mismatch, jb->gapOpen, NULL, NULL, &sc, NULL, NULL); break;
        case 2: orc_anw_fill(ref, n, qry, m, jb->match, jb->mismatch, jb->gapOpen, jb->gapExtend, NULL, NULL, NULL, NULL, NULL, NULL, &sc); break;
        default: orc_bsw_fill(ref, n, qry, m, jb->match, jb->mismatch, jb->gapOpen, jb->band, NULL, NULL, &sc, NULL, NULL); break;
        }
        jb->scores[p] = sc;
    }
    return NULL;
}

double orc_fill_batch_timed(int algo, const char *sequences, const int32_t *pairs4, int numPairs, int match, int mismatch,
                            int gapOpen, int gapExtend, int band, int threads, int32_t *scores) {
    if (threads < 1) threads = 1;
    if (threads > numPairs) threads = numPairs > 0 ? numPairs : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    struct job *jobs = (struct job *)malloc(sizeof(struct job) * (size_t)threads);
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    for (int t = 0; t < threads; t++) {
        struct job j = {algo, sequences, pairs4, (int)((long long)numPairs * t / threads),
                        (int)((long long)numPairs * (t + 1) / threads), match, mismatch, gapOpen, gapExtend, band, scores};
        jobs[t] = j;
        pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    gettimeofday(&t1, NULL);
    free(th); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_usec - t0.tv_usec);
}
