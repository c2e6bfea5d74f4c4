// parse_tool.cpp -- prints what parseInput / parseInputShard make of a pairs file, as JSON (no GPU involved; used by
// tests/test_abi_and_host.py to hold the C++ loaders against the Python restatement of the file format).
//   parse_tool <file> [rank world]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "parseInput.h"

int main(int argc, char *argv[]) {
    if (argc != 2 && argc != 4) { fprintf(stderr, "usage: parse_tool <file> [rank world]\n"); return 2; }
    seqPair *idx = nullptr;
    char *seq = nullptr;
    size_t first = 0, total = 0;
    inputInfo info;
    if (argc == 4) info = parseInputShard(argv[1], atoi(argv[2]), atoi(argv[3]), idx, seq, first, total);
    else { info = parseInput(argv[1], idx, seq); total = info.numPairs; }
    unsigned long long sum = 1469598103934665603ull; // FNV-1a over the bytes of every indexed sequence, in order
    for (size_t p = 0; p < info.numPairs; p++)
        for (int part = 0; part < 2; part++) {
            const char *s = seq + (part ? idx[p].queryIdx : idx[p].referenceIdx);
            const int n = part ? idx[p].querySize : idx[p].referenceSize;
            for (int k = 0; k < n; k++) { sum ^= (unsigned char)s[k]; sum *= 1099511628211ull; }
            if (s[n] != '\0') { fprintf(stderr, "sequence %zu/%d is not NUL-terminated\n", p, part); return 1; }
        }
    printf("{\"numPairs\": %zu, \"totalPairs\": %zu, \"firstPair\": %zu, \"numBytes\": %zu, \"numCells\": %zu, "
           "\"minRef\": %zu, \"maxRef\": %zu, \"minQry\": %zu, \"maxQry\": %zu, \"avgRef\": %.6f, \"avgQry\": %.6f, \"fnv\": \"%llx\", \"sizes\": [",
           info.numPairs, total, first, info.numBytes, info.numCells, info.numPairs ? info.minReferenceLength : 0, info.maxReferenceLength,
           info.numPairs ? info.minQueryLength : 0, info.maxQueryLength, info.avgReferenceLength, info.avgQueryLength, sum);
    for (size_t p = 0; p < info.numPairs; p++) printf("%s[%d, %d]", p ? ", " : "", idx[p].referenceSize, idx[p].querySize);
    printf("]}\n");
    cleanupParsedFile(idx, seq);
    return 0;
}
