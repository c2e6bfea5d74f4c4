#include "DpxPair.h"

#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>

#include "../../include/dpx_align.h"

namespace {
[[noreturn]] void fail(const char *what, int rc) {
    fprintf(stderr, "DPX ENGINE ERROR: %s: %s (%s)\n", what, dpx_strerror(rc), dpx_last_error());
    exit(1);
}
} // namespace

void dpxAlignPair(int algo, const std::string &reference, const std::string &query, int match, int mismatch, int gapOpen,
                  int gapExtend, int band, bool wantMatrices, DpxPairResult &out) {
    // flat buffer in parseInput layout: reference '\0' query '\0'
    std::string flat = reference;
    flat.push_back('\0');
    const int qryOff = (int)flat.size();
    flat += query;
    flat.push_back('\0');
    dpx_seq_pair sp{0, (int)reference.size(), qryOff, (int)query.size()};
    dpx_params prm{algo, match, mismatch, gapOpen, gapExtend, band};

    dpx_batch *b = nullptr;
    int rc = dpx_batch_create(&prm, flat.data(), flat.size(), &sp, 0, 1, DPX_KEEP_MATRICES, &b);
    if (rc != DPX_OK) fail("dpx_batch_create", rc);
    if ((rc = dpx_batch_fill(b, nullptr)) != DPX_OK) fail("dpx_batch_fill", rc);
    int32_t score = 0, er = 0, ec = 0;
    if ((rc = dpx_batch_results(b, &score, &er, &ec)) != DPX_OK) fail("dpx_batch_results", rc);
    out.score = score;
    out.endRow = er;
    out.endCol = ec;
    const size_t cap = reference.size() + query.size() + 2;
    std::vector<char> l0(cap), l1(cap), l2(cap);
    int32_t len = 0;
    if ((rc = dpx_batch_traceback(b, 0, l0.data(), l1.data(), l2.data(), &len)) != DPX_OK) fail("dpx_batch_traceback", rc);
    out.refLine.assign(l0.data(), (size_t)len);
    out.relLine.assign(l1.data(), (size_t)len);
    out.qryLine.assign(l2.data(), (size_t)len);
    if (wantMatrices) {
        const size_t cells = (reference.size() + 1) * (query.size() + 1);
        out.H.resize(cells);
        if ((rc = dpx_batch_matrix(b, 0, DPX_MAT_H, out.H.data())) != DPX_OK) fail("dpx_batch_matrix(H)", rc);
        if (algo == DPX_ALGO_ANW) {
            out.I.resize(cells);
            out.D.resize(cells);
            if ((rc = dpx_batch_matrix(b, 0, DPX_MAT_I, out.I.data())) != DPX_OK) fail("dpx_batch_matrix(I)", rc);
            if ((rc = dpx_batch_matrix(b, 0, DPX_MAT_D, out.D.data())) != DPX_OK) fail("dpx_batch_matrix(D)", rc);
        }
    }
    dpx_batch_destroy(b);
}

void dpxPrintScoreMatrix(const std::string &reference, const std::string &query, const std::vector<short> &M) {
    using std::cout;
    const size_t rows = query.size() + 1, cols = reference.size() + 1;
    const int w = 2;
    cout << "Reference: " << reference << " Size: " << reference.size() << "\n";
    cout << "Query: " << query << " Size: " << query.size() << "\n";
    cout << "Matrix Dim: [ " << rows << " x " << cols << " ]\n";
    cout << "    " << std::setw(w) << " " << " ";
    for (char c : reference) cout << "  " << std::setw(w) << c << " ";
    cout << "\n";
    for (size_t i = 0; i < rows; i++) {
        if (i == 0) cout << "  ";
        else cout << query[i - 1] << " ";
        cout << "[";
        for (size_t j = 0; j < cols; j++) {
            cout << " " << std::setw(w) << M[i * cols + j];
            if (j + 1 != cols) cout << ", ";
        }
        cout << "],\n";
    }
    cout << std::endl;
}
