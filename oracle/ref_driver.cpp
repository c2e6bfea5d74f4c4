/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.  Builds into oracle/_ref/ref_driver (git-ignored).
 *
 * A deterministic driver around the REAL reference classes (compiled from $(REF)/c++ where they
 * lie).  It exists because the reference's own c++/main.cpp (a) hard-codes the algorithm with a
 * #define (main.cpp:22-24), (b) silently drops pairs past the last full 400 (main.cpp:169) and
 * (c) prints in nondeterministic thread order.  Two modes:
 *
 *   ref_driver align <LSW|LNW|ANW> <file> <match> <mismatch> <open> [extend]
 *       -> stdout: for every pair, in order, exactly what the class's align() prints
 *          (golden text for tests/golden/).
 *   ref_driver time  <LSW|LNW|ANW> <file> <match> <mismatch> <open> <extend> <maxPairs>
 *       -> times the reference fill (init_matrix + score_matrix only) and the whole align()
 *          (stdout -> /dev/null) with the reference's threading shape: batches of
 *          THREADS_PER_BATCH=20 pthreads x PAIRS_PER_THREAD=20 pairs (main.cpp:18-19,166-232).
 *          Prints one JSON line.  This is bench.py's cpu_baseline with kind "reference".
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <pthread.h>
#include <unistd.h>

#include "LinearSmithWaterman.h"
#include "LinearNeedlemanWunsch.h"
#include "AffineNeedlemanWunsch.h"
#include "parseInput.h"
#include "timing.h"

namespace {

enum Algo { LSW, LNW, ANW };

struct Work {
    Algo algo;
    bool fillOnly;
    const char *sequences;
    const seqPair *idx;
    int lo, hi;
    int match, mismatch, open, extend;
};

void run_pair(const Work &w, int i) {
    const char *ref = &w.sequences[w.idx[i].referenceIdx];
    const char *qry = &w.sequences[w.idx[i].queryIdx];
    switch (w.algo) {
    case LSW: {
        LinearSmithWaterman a(ref, qry, i, w.match, w.mismatch, w.open);
        if (w.fillOnly) { a.init_matrix(); a.score_matrix(); } else a.align();
        break;
    }
    case LNW: {
        LinearNeedlemanWunsch a(ref, qry, i, w.match, w.mismatch, w.open);
        if (w.fillOnly) { a.init_matrix(); a.score_matrix(); } else a.align();
        break;
    }
    case ANW: {
        AffineNeedlemanWunsch a(ref, qry, i, w.match, w.mismatch, w.open, w.extend);
        if (w.fillOnly) { a.init_matrix(); a.score_matrix(); } else a.align();
        break;
    }
    }
}

void *thread_main(void *p) {
    Work *w = (Work *)p;
    for (int i = w->lo; i < w->hi; i++) run_pair(*w, i);
    return NULL;
}

/* the reference's batch shape, but with the tail handled (no dropped / out-of-range pairs) */
double run_batched(Work base, int numPairs) {
    const int pairsPerThread = 20, threadsPerBatch = 20;
    uint64_t t0 = get_time();
    for (int start = 0; start < numPairs; start += pairsPerThread * threadsPerBatch) {
        pthread_t th[threadsPerBatch];
        Work w[threadsPerBatch];
        int nth = 0;
        for (int t = 0; t < threadsPerBatch; t++) {
            int lo = start + t * pairsPerThread;
            if (lo >= numPairs) break;
            w[t] = base;
            w[t].lo = lo;
            w[t].hi = std::min(numPairs, lo + pairsPerThread);
            pthread_create(&th[t], NULL, thread_main, &w[t]);
            nth++;
        }
        for (int t = 0; t < nth; t++) pthread_join(th[t], NULL);
    }
    return (double)(get_time() - t0) * 1e-6;
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: ref_driver align|time LSW|LNW|ANW file match mismatch open [extend [maxPairs]]\n");
        return 2;
    }
    bool timing = strcmp(argv[1], "time") == 0;
    Algo algo = strcmp(argv[2], "LSW") == 0 ? LSW : strcmp(argv[2], "LNW") == 0 ? LNW : ANW;
    seqPair *idx;
    char *sequences;
    inputInfo info = parseInput(argv[3], idx, sequences);
    Work base;
    base.algo = algo;
    base.fillOnly = false;
    base.sequences = sequences;
    base.idx = idx;
    base.match = atoi(argv[4]);
    base.mismatch = atoi(argv[5]);
    base.open = atoi(argv[6]);
    base.extend = argc > 7 ? atoi(argv[7]) : -1;
    int numPairs = (int)info.numPairs;
    if (argc > 8) numPairs = std::min(numPairs, atoi(argv[8]));

    if (!timing) {
        base.lo = 0;
        base.hi = numPairs;
        thread_main(&base);
        fflush(stdout);
    } else {
        double cells = 0;
        for (int i = 0; i < numPairs; i++) cells += (double)idx[i].referenceSize * (double)idx[i].querySize;
        base.fillOnly = true;
        double tFill = run_batched(base, numPairs);
        /* whole align(): the classes print; send that to /dev/null for the duration */
        fflush(stdout);
        int saved = dup(1);
        FILE *devnull = fopen("/dev/null", "w");
        dup2(fileno(devnull), 1);
        base.fillOnly = false;
        double tAlign = run_batched(base, numPairs);
        fflush(stdout);
        dup2(saved, 1);
        close(saved);
        fclose(devnull);
        long cores = sysconf(_SC_NPROCESSORS_ONLN);
        printf("{\"pairs\": %d, \"cells\": %.0f, \"fill_sec\": %.6f, \"align_sec\": %.6f, \"fill_gcups\": %.6f, "
               "\"align_gcups\": %.6f, \"threads\": 20, \"cores\": %ld}\n",
               numPairs, cells, tFill, tAlign, cells / tFill / 1e9, cells / tAlign / 1e9, cores);
    }
    cleanupParsedFile(idx, sequences);
    return 0;
}
