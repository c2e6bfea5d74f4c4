// parseInput.h -- flat batch input of the reference (c++/parseInput.h:9-35): a pairs file has 3 lines per pair
// (seed/score line -- ignored --, reference, query).  parseInput() loads the whole file into one malloc'ed buffer,
// turns every '\n' into '\0' and returns, per pair, byte offsets + lengths into that buffer.  The struct layouts are
// the reference's (seqPair is also the C ABI's dpx_seq_pair).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define PRINT_PARSED_PAIRS

struct seqPair { // one alignment pair: where its two strings start in the flat buffer and how long they are
    int referenceIdx, referenceSize;
    int queryIdx, querySize;
};
static_assert(sizeof(seqPair) == 16, "seqPair is also the C ABI's dpx_seq_pair");

struct inputInfo { // what the loader saw
    size_t numPairs, numBytes;
    size_t numCells;                                  // sum of referenceSize * querySize: the denominator of GCUPS
    size_t minReferenceLength, minQueryLength;
    size_t maxReferenceLength, maxQueryLength;
    double avgReferenceLength, avgQueryLength;
};

inputInfo parseInput(const char *pairFileName, seqPair *&sequence_indices, char *&sequences);

// Shard loader for one-process-per-GPU runs (not in the reference).  Maps the file, walks its newlines once
// and materialises ONLY pairs [firstPair, firstPair + info.numPairs) = the rank's contiguous ceil(N/world)-sized range:
// `sequences` holds just those lines ('\n' -> '\0'), the seqPair offsets are relative to it, the statistics in the
// returned inputInfo describe the shard, `totalPairs` is the pair count of the whole file.  Same failure modes as
// parseInput (message on stderr + exit(1)); world == 1 yields parseInput's records.
inputInfo parseInputShard(const char *pairFileName, int rank, int world, seqPair *&sequence_indices, char *&sequences,
                          size_t &firstPair, size_t &totalPairs);
void printParsedFile(const size_t numPairs, const seqPair *sequence_indices, const char *sequences);
void cleanupParsedFile(seqPair *sequence_indices, char *sequences);
