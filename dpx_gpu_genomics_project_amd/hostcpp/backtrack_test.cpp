// backtrack_test.cpp -- exercises the host back-trackers of backtrack.h (the output-side plumbing the reference's CUDA
// mains call after their D2H: cuda/LinearSmithWaterman.cu:316, cuda/AffineNeedlemanWunsch.cu:391,
// cuda/LNW/LinearNeedlemanWunschV7.cu:209) the way such a main would use them on top of the engine:
//   one device batch for the file -> dpx_batch_matrix() exports H (and I, D) row-major ->
//   dpxDirectionsFromScores() rebuilds the reference's direction matrices -> backtrackSW / NW / MultiNW / ANW print.
// stdout is "<pair> | <score>" + three lines per pair, the format of c++/main.cpp, so tests can compare it with the
// reference's output byte for byte.   -dump K additionally prints pair K's matrices with printMatrix /
// printBacktrackMatrix (cuda/LNW/LinearNeedlemanWunschV8.cu:601-603).
//
//   backtrack_test -pairs <file> -algo LSW|LNW|MULTINW|ANW [-match 3 -mismatch -1 -open -2 -extend -1] [-dump K]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dpx_align.h"
#include "backtrack.h"
#include "parseInput.h"

static void die(const char *what, int rc) {
    fprintf(stderr, "%s: %s (%s)\n", what, dpx_strerror(rc), dpx_last_error());
    exit(1);
}

int main(int argc, char *argv[]) {
    const char *file = nullptr;
    std::string algoName = "LSW";
    int match = 3, mismatch = -1, gapOpen = -2, gapExtend = -1, dump = -1;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "-pairs")) file = argv[i + 1];
        else if (!strcmp(argv[i], "-algo")) algoName = argv[i + 1];
        else if (!strcmp(argv[i], "-match")) match = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-mismatch")) mismatch = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-open")) gapOpen = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-extend")) gapExtend = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-dump")) dump = atoi(argv[i + 1]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (!file) { fprintf(stderr, "need -pairs <file>\n"); return 2; }
    const bool multi = algoName == "MULTINW";
    const int algo = algoName == "LSW" ? DPX_ALGO_LSW : (algoName == "LNW" || multi) ? DPX_ALGO_LNW : algoName == "ANW" ? DPX_ALGO_ANW : -1;
    if (algo < 0) { fprintf(stderr, "unknown -algo\n"); return 2; }

    seqPair *idx;
    char *seq;
    const inputInfo info = parseInput(file, idx, seq);
    const dpx_params prm{algo, match, mismatch, gapOpen, gapExtend, 0};
    dpx_batch *b = nullptr;
    int rc = dpx_batch_create(&prm, seq, info.numBytes, reinterpret_cast<const dpx_seq_pair *>(idx), 0, info.numPairs, DPX_KEEP_MATRICES, &b);
    if (rc != DPX_OK) die("dpx_batch_create", rc);
    if ((rc = dpx_batch_fill(b, nullptr)) != DPX_OK) die("dpx_batch_fill", rc);
    std::vector<int32_t> score(info.numPairs), er(info.numPairs), ec(info.numPairs);
    if ((rc = dpx_batch_results(b, score.data(), er.data(), ec.data())) != DPX_OK) die("dpx_batch_results", rc);

    std::vector<short> H, I, D;
    std::vector<directionMain> dirH;
    std::vector<directionIndel> dirI, dirD;
    for (size_t p = 0; p < info.numPairs; p++) {
        const char *ref = seq + idx[p].referenceIdx, *qry = seq + idx[p].queryIdx;
        const int n = idx[p].referenceSize, m = idx[p].querySize;
        const size_t cells = (size_t)(m + 1) * (size_t)(n + 1);
        H.resize(cells);
        dirH.resize(cells);
        if ((rc = dpx_batch_matrix(b, p, DPX_MAT_H, H.data())) != DPX_OK) die("dpx_batch_matrix(H)", rc);
        if (algo == DPX_ALGO_ANW) {
            I.resize(cells); D.resize(cells); dirI.resize(cells); dirD.resize(cells);
            if ((rc = dpx_batch_matrix(b, p, DPX_MAT_I, I.data())) != DPX_OK) die("dpx_batch_matrix(I)", rc);
            if ((rc = dpx_batch_matrix(b, p, DPX_MAT_D, D.data())) != DPX_OK) die("dpx_batch_matrix(D)", rc);
        }
        dpxDirectionsFromScores(algo, H.data(), algo == DPX_ALGO_ANW ? I.data() : nullptr, algo == DPX_ALGO_ANW ? D.data() : nullptr, ref, n,
                                qry, m, match, mismatch, gapOpen, gapExtend, dirH.data(), algo == DPX_ALGO_ANW ? dirI.data() : nullptr,
                                algo == DPX_ALGO_ANW ? dirD.data() : nullptr);
        if ((int)p == dump) {
            std::vector<int> wide(H.begin(), H.end());
            printMatrix(wide.data(), n + 1, m + 1);
            printBacktrackMatrix(dirH.data(), n + 1, m + 1);
            continue;
        }
        if (multi) { // prints its own "<pair> | <score>" header under the print lock (c++/backtrack.cpp:205-210)
            backtrackMultiNW(dirH.data(), ref, n, qry, m, (int)p, score[p]);
            continue;
        }
        printf("%zu | %d\n", p, score[p]);
        if (algo == DPX_ALGO_LSW) backtrackSW(er[p], ec[p], n + 1, dirH.data(), ref, qry);
        else if (algo == DPX_ALGO_LNW) backtrackNW(dirH.data(), ref, n, qry, m);
        else backtrackANW(dirH.data(), dirI.data(), dirD.data(), ref, n, qry, m);
    }
    dpx_batch_destroy(b);
    cleanupParsedFile(idx, seq);
    return 0;
}
