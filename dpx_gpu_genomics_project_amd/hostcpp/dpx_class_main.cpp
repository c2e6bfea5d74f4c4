// dpx_class_main.cpp -- the reference's c++/main.cpp shape on the MI355X engine: batches of THREADS_PER_BATCH pthreads,
// each aligning PAIRS_PER_THREAD consecutive pairs through the SequenceAligner-derived classes (c++/main.cpp:18-19,
// 166-232), same argv, same stdout lines.  Differences, all fixes of upstream defects that never change a printed
// block: the algorithm is a run-time flag instead of a #define (main.cpp:22-24), pairs past the last full 400 are
// not dropped (main.cpp:169 floors twice) and the per-thread loop is clamped to numPairs (main.cpp:61 is not).
//
//   dpx_class_main -pairs <file> -match M -mismatch X -open O [-extend E] [-algo LSW|LNW|ANW|BSW] [-band B]
#include <pthread.h>

#include <cassert>
#include <cstring>
#include <string>

#include "AffineNeedlemanWunsch.h"
#include "BandedSmithWaterman.h"
#include "LinearNeedlemanWunsch.h"
#include "LinearSmithWaterman.h"
#include "parseInput.h"
#include "printLock.h"
#include "timing.h"

#define PAIRS_PER_THREAD 20
#define THREADS_PER_BATCH 20

namespace {
enum Algo { LSW, LNW, ANW, BSW };

struct thread_arg {
    Algo algo;
    int firstPair, endPair;
    const char *sequences;
    const seqPair *sequenceIdxs;
    int matchWeight, mismatchWeight, gapOpenWeight, gapExtendWeight, band;
};

void *threadCompute(void *tmp) {
    const thread_arg *a = static_cast<const thread_arg *>(tmp);
    for (int i = a->firstPair; i < a->endPair; i++) {
        const char *ref = &a->sequences[a->sequenceIdxs[i].referenceIdx];
        const char *qry = &a->sequences[a->sequenceIdxs[i].queryIdx];
        switch (a->algo) {
        case LSW: { LinearSmithWaterman x(ref, qry, i, a->matchWeight, a->mismatchWeight, a->gapOpenWeight); x.align(); break; }
        case LNW: { LinearNeedlemanWunsch x(ref, qry, i, a->matchWeight, a->mismatchWeight, a->gapOpenWeight); x.align(); break; }
        case ANW: { AffineNeedlemanWunsch x(ref, qry, i, a->matchWeight, a->mismatchWeight, a->gapOpenWeight, a->gapExtendWeight); x.align(); break; }
        case BSW: { BandedSmithWaterman x(ref, qry, a->matchWeight, a->mismatchWeight, a->gapOpenWeight, i, a->band); x.align(); break; }
        }
    }
    return nullptr;
}
} // namespace

int main(int argc, char *argv[]) {
    if (argc < 3) {
        fprintf(stderr, "usage: dpx_class_main -pairs <InSeqFile> -match <matchWeight> -mismatch <mismatchWeight> -open <gapWeight> "
                        "[-extend <gapExtend>] [-algo LSW|LNW|ANW|BSW] [-band <B>]\n");
        exit(EXIT_FAILURE);
    }
    const char *pairFileName = nullptr;
    thread_arg proto{};
    proto.algo = LSW;
    proto.matchWeight = 3; proto.mismatchWeight = -1; proto.gapOpenWeight = -4; proto.gapExtendWeight = -1; // main.cpp:128-132
    proto.band = DPX_DEFAULT_BAND;
    for (int i = 1; i + 1 < argc; i += 2) {
        const char *f = argv[i], *v = argv[i + 1];
        if (!strcmp(f, "-pairs")) pairFileName = v;
        else if (!strcmp(f, "-match")) proto.matchWeight = atoi(v);
        else if (!strcmp(f, "-mismatch")) proto.mismatchWeight = atoi(v);
        else if (!strcmp(f, "-open") || !strcmp(f, "-gap")) proto.gapOpenWeight = atoi(v);
        else if (!strcmp(f, "-extend")) proto.gapExtendWeight = atoi(v);
        else if (!strcmp(f, "-band")) proto.band = atoi(v);
        else if (!strcmp(f, "-algo")) proto.algo = !strcmp(v, "LNW") ? LNW : !strcmp(v, "ANW") ? ANW : !strcmp(v, "BSW") ? BSW : LSW;
    }
    if (!pairFileName) { fprintf(stderr, "need -pairs <file>\n"); exit(EXIT_FAILURE); }

    printf("Parsing input file: %s\n", pairFileName);
    seqPair *sequenceIdxs;
    char *sequences;
    inputInfo fileInfo = parseInput(pairFileName, sequenceIdxs, sequences);
    proto.sequences = sequences;
    proto.sequenceIdxs = sequenceIdxs;

    start_timer();
    printf("Pair # | Score\n");
    const int numPairs = (int)fileInfo.numPairs;
    for (int batchStart = 0; batchStart < numPairs; batchStart += PAIRS_PER_THREAD * THREADS_PER_BATCH) {
        pthread_t threads[THREADS_PER_BATCH];
        thread_arg args[THREADS_PER_BATCH];
        int started = 0;
        for (int t = 0; t < THREADS_PER_BATCH; t++) {
            const int first = batchStart + t * PAIRS_PER_THREAD;
            if (first >= numPairs) break;
            args[t] = proto;
            args[t].firstPair = first;
            args[t].endPair = std::min(numPairs, first + PAIRS_PER_THREAD);
            const int ret = pthread_create(&threads[t], nullptr, threadCompute, &args[t]);
            assert(ret == 0);
            (void)ret;
            started++;
        }
        for (int t = 0; t < started; t++) pthread_join(threads[t], nullptr);
    }
    const uint64_t elapsed_time = get_elapsed_time();
    printf("Elapsed time (usec): %llu\n", (unsigned long long)elapsed_time);
    printf("Cleaning up\n");
    cleanupParsedFile(sequenceIdxs, sequences);
    return 0;
}
