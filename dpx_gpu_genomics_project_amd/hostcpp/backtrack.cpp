// backtrack.cpp -- see backtrack.h.  Lines are assembled back-to-front in three flat buffers (an alignment is at
// most referenceLength + queryLength long) instead of prepending to std::strings; the printed text is the same.
#include "backtrack.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

struct Lines {
    std::vector<char> r, x, q;
    size_t pos;
    explicit Lines(size_t cap) : r(cap + 1), x(cap + 1), q(cap + 1), pos(cap) { r[cap] = x[cap] = q[cap] = '\0'; }
    void push(char rc, char xc, char qc) { --pos; r[pos] = rc; x[pos] = xc; q[pos] = qc; }
    void print() const { printf("%s\n%s\n%s\n", &r[pos], &x[pos], &q[pos]); }
};

// one step of a linear-gap walk; returns false on a direction that cannot be followed
inline bool stepMain(directionMain d, Lines &out, const char *ref, const char *qry, int &row, int &col) {
    switch (d) {
    case MATCH: out.push(ref[col - 1], '*', qry[row - 1]); --row; --col; return true;
    case MISMATCH: out.push(ref[col - 1], '|', qry[row - 1]); --row; --col; return true;
    case QUERY_DELETION: out.push('_', ' ', qry[row - 1]); --row; return true;
    case QUERY_INSERTION: out.push(ref[col - 1], ' ', '_'); --col; return true;
    default: return false;
    }
}

} // namespace

void printMatrix(const int *memo, const int cols, const int rows) {
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < cols; c++) printf(" %4d ", memo[(size_t)r * cols + c]);
        printf("\n");
    }
}

void printBacktrackMatrix(const directionMain *memo, const int cols, const int rows) {
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < cols; c++) printf(" %4d ", (int)memo[(size_t)r * cols + c]);
        printf("\n");
    }
}

void backtrackNW(const directionMain *dir, const char *ref, const int refLen, const char *qry, const int qryLen) {
    const int numCols = refLen + 1;
    int row = qryLen, col = refLen;
    Lines out((size_t)refLen + qryLen);
    while (row != 0 || col != 0)
        if (!stepMain(dir[(size_t)row * numCols + col], out, ref, qry, row, col)) exit(1);
    out.print();
}

void backtrackSW(int row, int col, const int numCols, const directionMain *dir, const char *ref, const char *qry) {
    Lines out((size_t)row + col);
    while (row > 0 && col > 0 && dir[(size_t)row * numCols + col] != NONE_MAIN)
        if (!stepMain(dir[(size_t)row * numCols + col], out, ref, qry, row, col)) exit(1);
    out.print();
}

void backtrackMultiNW(const directionMain *dir, const char *ref, const int refLen, const char *qry, const int qryLen,
                      const int pairNum, const int score) {
    const int numCols = refLen + 1;
    int row = qryLen, col = refLen;
    Lines out((size_t)refLen + qryLen);
    while (row != 0 || col != 0) {
        if (!stepMain(dir[(size_t)row * numCols + col], out, ref, qry, row, col)) {
            printLock();
            printf("Exiting(1) backtrack: %d\n", pairNum);
            printUnlock();
            exit(1);
        }
    }
    printLock();
    printf("%d | %d\n", pairNum, score);
    out.print();
    printUnlock();
}

void backtrackANW(const directionMain *dirH, const directionIndel *dirI, const directionIndel *dirD, const char *ref,
                  const int refLen, const char *qry, const int qryLen) {
    const int numCols = refLen + 1;
    int row = qryLen, col = refLen;
    currentMatrixPosition where = SCORING;
    Lines out((size_t)refLen + qryLen);
    while (row != 0 && col != 0) {
        const size_t k = (size_t)row * numCols + col;
        if (where == SCORING) {
            const directionMain d = dirH[k];
            if (d == QUERY_DELETION) where = DELETION;        // the gap itself is emitted from the gap matrix
            else if (d == QUERY_INSERTION) where = INSERTION;
            else if (!stepMain(d, out, ref, qry, row, col)) exit(1);
        } else if (where == INSERTION) {
            if (dirI[k] == GAP_OPEN) where = SCORING;
            else if (dirI[k] != GAP_EXTEND) exit(1);
            out.push(ref[col - 1], ' ', '_');
            --col;
        } else {
            if (dirD[k] == GAP_OPEN) where = SCORING;
            else if (dirD[k] != GAP_EXTEND) exit(1);
            out.push('_', ' ', qry[row - 1]);
            --row;
        }
    }
    for (; row > 0; --row) out.push('_', ' ', qry[row - 1]);
    for (; col > 0; --col) out.push(ref[col - 1], ' ', '_');
    out.print();
}

void dpxDirectionsFromScores(int algo, const short *H, const short *I, const short *D, const char *ref, const int refLen,
                             const char *qry, const int qryLen, int match, int mismatch, int gapOpen, int gapExtend,
                             directionMain *dirH, directionIndel *dirI, directionIndel *dirD) {
    const int n = refLen, m = qryLen;
    const size_t nc = (size_t)n + 1;
    for (int i = 0; i <= m; i++) {
        for (int j = 0; j <= n; j++) {
            const size_t k = (size_t)i * nc + j;
            if (dirI) dirI[k] = NONE_INDEL;
            if (dirD) dirD[k] = NONE_INDEL;
            if (i == 0 || j == 0) { // global aligners walk the borders; local ones never read them
                dirH[k] = (algo == 1 || (i == 0 && j == 0)) ? NONE_MAIN : (i == 0 ? QUERY_INSERTION : QUERY_DELETION);
                continue;
            }
            const bool eq = qry[i - 1] == ref[j - 1];
            const int diag = H[k - nc - 1] + (eq ? match : mismatch);
            const directionMain corner = eq ? MATCH : MISMATCH;
            if (algo == 1) { // LSW: NONE at a zero cell, else UPPER, LEFT, CORNER (c++/LinearSmithWaterman.cpp:106-108)
                const int h = H[k];
                dirH[k] = h <= 0 ? NONE_MAIN : (H[k - nc] + gapOpen == h ? QUERY_DELETION : (H[k - 1] + gapOpen == h ? QUERY_INSERTION : corner));
            } else if (algo == 0) { // LNW: left >= max(up, diag) ? INSERTION : up >= diag ? DELETION : corner
                const int del = H[k - nc] + gapOpen, ins = H[k - 1] + gapOpen;
                dirH[k] = ins >= std::max(del, diag) ? QUERY_INSERTION : (del >= diag ? QUERY_DELETION : corner);
            } else { // ANW (c++/AffineNeedlemanWunsch.cpp:185-233)
                dirH[k] = I[k] >= std::max((int)D[k], diag) ? QUERY_INSERTION : (D[k] >= diag ? QUERY_DELETION : corner);
                if (dirD) dirD[k] = (i == 1 || H[k - nc] + gapOpen + gapExtend >= D[k - nc] + gapExtend) ? GAP_OPEN : GAP_EXTEND;
                if (dirI) dirI[k] = (j == 1 || H[k - 1] + gapOpen + gapExtend >= I[k - 1] + gapExtend) ? GAP_OPEN : GAP_EXTEND;
            }
        }
    }
}
